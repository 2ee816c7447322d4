"""shape_from_shading J^T(J v): the marching kernel (energy_sfs.hip k_march) against the LDS-tiled k_fused<1> through the C-ABI shim.
   python tools/sfs_probe.py check           bitwise / rounding comparison on ragged sizes, slabs, the three variants (plain, sums, LM diagonal)
   python tools/sfs_probe.py time [W H]      launch times (HIP events, 50 launches) of both kernels and a sweep of the marching grid"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from thallo_amd import api, synthetic as syn

L = api.lib()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
L.thallo_hip_sfs_march_debug_set.argtypes = [C.c_int, C.c_int]; L.thallo_hip_sfs_march_debug_set.restype = None


class Inst:
    def __init__(self, W, H, ra=None, rb=None, yoff=0, Hg=None, seed=3):
        self.W, self.H = W, H
        self.ra, self.rb = (0 if ra is None else ra), (H if rb is None else rb)
        self.yoff, self.Hg = yoff, (H if Hg is None else Hg)
        p = syn.shape_from_shading(W, H)
        self.hp = (C.c_float * 16)(*[float(x) for x in p[:16]])
        X, D, Im, mR, mC = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p[16:21]]
        self.X, self.D = X, D
        N = W * H
        self.Im, self.mR, self.mC = Im, mR, mC
        L.thallo_hip_sfs_march_debug_set(6, 0)            # the legacy planes (float4 G, float2 Wt, byte flags) and their kernels first
        self.G, self.Wt, self.fl = self.precompute(0)
        G1, Wt1, fl1 = self.precompute(1)
        # the marching precompute against k_precompute: flags and row weights bitwise, G to rounding (same expression tree, separate instantiations)
        gs = float(self.G.abs().max())
        self.pre_ok = bool(torch.equal(self.fl[:N], fl1[:N]) and torch.equal(self.Wt, Wt1) and float((self.G - G1).abs().max()) <= 1e-5 * gs)
        # planes + computeCost in one launch against k_precompute + k_cost
        cp = torch.zeros(1024, device="cuda"); cq = torch.zeros(1024, device="cuda")
        L.thallo_hip_sfs_march_debug_set(4, 1)
        G2 = torch.empty_like(self.G); Wt2 = torch.empty_like(self.Wt); fl2 = torch.full((N + 4,), 9, dtype=torch.uint8, device="cuda")
        nb1 = L.thallo_hip_sfs_precompute_cost(W, H, 0, H, self.yoff, self.Hg, self.hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G2), vp(Wt2), vp(fl2), self.ra, self.rb, vp(cp), None)
        nb0 = L.thallo_hip_sfs_cost(W, H, self.ra, self.rb, self.yoff, self.Hg, self.hp, vp(X), vp(D), vp(self.G), vp(self.Wt), vp(self.fl), vp(cq), None)
        torch.cuda.synchronize()
        assert nb1 > 0 and nb0 > 0, (nb1, nb0)
        ca, cb = float(cp[:nb1].double().sum()), float(cq[:nb0].double().sum())
        same_planes = bool(torch.equal(G2, G1) and torch.equal(Wt2, Wt1) and torch.equal(fl2[:N], fl1[:N]))
        self.pre_ok = self.pre_ok and same_planes and abs(ca - cb) <= 2e-6 * abs(cb)
        self.pre_msg = f"cost in the precompute launch {ca:.8g} vs k_cost {cb:.8g} (rel {abs(ca - cb) / abs(cb):.1e}), its planes bitwise the plain launch's {same_planes}; precompute marching vs k_precompute: flags equal {torch.equal(self.fl[:N], fl1[:N])}, Wt equal {torch.equal(self.Wt, Wt1)}, G max diff / max {float((self.G - G1).abs().max()) / gs:.2e}"
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        self.p = torch.randn(N, device="cuda", generator=g) * 1e-3
        self.r = torch.randn(N, device="cuda", generator=g) * 1e-3
        self.ctc = torch.rand(N, device="cuda", generator=g) * 50.0
        self.U = torch.empty(2 * N, dtype=torch.float32, device="cuda"); self.R = torch.empty(3 * N, dtype=torch.float32, device="cuda")
        self.pair_planes()
        L.thallo_hip_sfs_march_debug_set(6, 0)

    def pair_planes(self):
        """round 6: the PACKED planes of the pixel-pair kernels (Gx | Gy | Gz | BI planar in the G buffer; flags | maskR << 8 | maskC << 16 per pixel in the Wt buffer), written by the
        closed-form precompute -- against the legacy planes: flags and mask bytes exactly, BI to rounding, the three partials to 2e-5 of the largest (forward-mode duals vs closed form)"""
        W, H, N = self.W, self.H, self.W * self.H
        self.pair = False
        L.thallo_hip_sfs_march_debug_set(6, 1)
        if W % 2 or not L.thallo_hip_sfs_planes_layout(W, H):
            self.pair_msg = "pair layout: not for this size"; self.pair_ok = True
            return
        self.pair = True
        X, D, Im, mR, mC = self.X, self.D, self.Im, self.mR, self.mC
        self.Gp = torch.full((4 * N,), 3.0, dtype=torch.float32, device="cuda"); self.Fw = torch.full((2 * N,), 3.0, dtype=torch.float32, device="cuda")
        flx = torch.full((N + 4,), 9, dtype=torch.uint8, device="cuda")
        rc = L.thallo_hip_sfs_precompute(W, H, 0, H, self.yoff, self.Hg, self.hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(self.Gp), vp(self.Fw), vp(flx), None)
        assert rc == 0, rc
        Gp2 = torch.empty_like(self.Gp); Fw2 = torch.empty_like(self.Fw); cp = torch.zeros(1024, device="cuda"); cq = torch.zeros(1024, device="cuda")
        nb1 = L.thallo_hip_sfs_precompute_cost(W, H, 0, H, self.yoff, self.Hg, self.hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(Gp2), vp(Fw2), vp(flx), self.ra, self.rb, vp(cp), None)
        L.thallo_hip_sfs_march_debug_set(6, 0)
        nb0 = L.thallo_hip_sfs_cost(W, H, self.ra, self.rb, self.yoff, self.Hg, self.hp, vp(X), vp(D), vp(self.G), vp(self.Wt), vp(self.fl), vp(cq), None)
        torch.cuda.synchronize()
        assert nb1 > 0 and nb0 > 0, (nb1, nb0)
        ca, cb = float(cp[:nb1].double().sum()), float(cq[:nb0].double().sum())
        g4 = self.G.view(N, 4)
        gs = float(g4[:, :3].abs().max())
        gd = max(float((self.Gp[k * N:(k + 1) * N] - g4[:, k]).abs().max()) for k in range(3)) / gs
        bs = max(float(g4[:, 3].abs().max()), 1e-30)
        bd = float((self.Gp[3 * N:] - g4[:, 3]).abs().max()) / bs
        fw = self.Fw[:N].view(torch.int32)
        wg = float(np.sqrt(np.float32(self.hp[2])))
        flags_eq = bool(torch.equal((fw & 0xff).to(torch.uint8), self.fl[:N]))
        w2 = self.Wt.view(N, 2)
        wx = (((fw >> 8) & 0xff).float() * np.float32(wg)); wy = (((fw >> 16) & 0xff).float() * np.float32(wg))
        w_eq = bool(torch.equal(wx, w2[:, 0]) and torch.equal(wy, w2[:, 1]))
        same = bool(torch.equal(Gp2, self.Gp) and torch.equal(Fw2[:N].view(torch.int32), fw))
        self.pair_ok = flags_eq and w_eq and same and gd < 2e-5 and bd < 2e-6 and abs(ca - cb) <= 2e-6 * abs(cb)
        self.pair_msg = (f"pair planes: flags equal {flags_eq}, row weights equal {w_eq}, partials max diff / max {gd:.2e}, BI max diff / max {bd:.2e}; cost in the precompute launch "
                         f"{ca:.8g} vs k_cost {cb:.8g} (rel {abs(ca - cb) / abs(cb):.1e}), its planes bitwise the plain launch's {same}")

    def precompute(self, march):
        N = self.W * self.H
        G = torch.full((4 * N,), 3.0, dtype=torch.float32, device="cuda"); Wt = torch.full((2 * N,), 3.0, dtype=torch.float32, device="cuda")
        fl = torch.full((N + 4,), 9, dtype=torch.uint8, device="cuda")
        L.thallo_hip_sfs_march_debug_set(4, march)
        rc = L.thallo_hip_sfs_precompute(self.W, self.H, 0, self.H, self.yoff, self.Hg, self.hp, vp(self.X), vp(self.D), vp(self.Im), vp(self.mR), vp(self.mC), vp(G), vp(Wt), vp(fl), None)
        L.thallo_hip_sfs_march_debug_set(4, 1)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return G, Wt, fl

    def apply(self, variant, Ap, aD, s3, pair=False):
        L.thallo_hip_sfs_march_debug_set(6, 1 if pair else 0)
        G, Wt = (self.Gp, self.Fw) if pair else (self.G, self.Wt)
        a = (self.W, self.H, self.ra, self.rb, self.yoff, self.Hg, self.hp, vp(G), vp(Wt), vp(self.fl), vp(self.U), vp(self.R), vp(self.p))
        if variant == "plain":
            return L.thallo_hip_sfs_apply_jtj(*a, vp(Ap), vp(aD), None)
        if variant == "sums":
            return L.thallo_hip_sfs_apply_jtj_sums(*a, vp(Ap), vp(aD), vp(self.r), vp(s3), None)
        return L.thallo_hip_sfs_apply_jtj_lm(*a, vp(self.ctc), vp(Ap), vp(aD), None, None)


def run_init(inst, march, diag_by_march=True, pair=False):
    """PCGInit1's J^T F pass: r = -J^T F, z = r, p_prev = 0, delta = 0, alphaN partials; and the raw LM diagonal diag(J^T J)"""
    L.thallo_hip_sfs_march_debug_set(2, 1 if march else 0); L.thallo_hip_sfs_march_debug_set(3, 1 if diag_by_march else 0); L.thallo_hip_sfs_march_debug_set(6, 1 if pair else 0)
    N = inst.W * inst.H
    o = [torch.full((N,), 7.0, device="cuda") for _ in range(5)]
    aN = torch.zeros(1024, device="cuda")
    G, Wt = (inst.Gp, inst.Fw) if pair else (inst.G, inst.Wt)
    nb = L.thallo_hip_sfs_pcg_init(inst.W, inst.H, inst.ra, inst.rb, inst.yoff, inst.Hg, inst.hp, vp(inst.X), vp(inst.D), vp(G), vp(Wt), vp(inst.fl),
                                   vp(inst.U), vp(inst.R), vp(o[0]), vp(o[1]), vp(o[2]), vp(o[3]), vp(o[4]), vp(aN), None)
    assert nb > 0, nb
    torch.cuda.synchronize()
    L.thallo_hip_sfs_march_debug_set(3, 1); L.thallo_hip_sfs_march_debug_set(6, 0)
    return [t.cpu().numpy() for t in o], float(aN[:nb].double().sum())


def run(inst, variant, march, pair=False):
    L.thallo_hip_sfs_march_debug_set(2, 1 if march else 0)
    N = inst.W * inst.H
    Ap = torch.full((N,), 7.0, device="cuda"); aD = torch.zeros(1024, device="cuda"); s3 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
    nb = inst.apply(variant, Ap, aD, s3, pair)
    L.thallo_hip_sfs_march_debug_set(6, 0)
    assert nb > 0, nb
    torch.cuda.synchronize()
    return Ap.cpu().numpy(), float(aD[:nb].double().sum()), s3[:3 * nb].view(-1, 3).sum(0).cpu().numpy()


def check_pair_lm(inst):
    """round 6, packed planes: (1) PCGInit1 with PCGFinalizeDiagonal riding along (thallo_hip_sfs_pcg_init_lm) against thallo_hip_sfs_pcg_init + thallo_hip_lm_finalize_diagonal on
    the same planes: r, delta, p_prev, b, SSq bitwise; CtC, M^-1, z to rounding (the same expressions in another kernel); alphaN to summation order.  (2) the LM model cost in one launch
    (thallo_hip_sfs_lm_model_cost) against thallo_hip_lm_owed_delta + thallo_hip_sfs_apply_jtj + thallo_hip_dot: delta bitwise, the two sums to summation order."""
    W, H, N = inst.W, inst.H, inst.W * inst.H
    if inst.ra != 0 or inst.rb != H:
        return True, "pair LM launches: whole images only"
    L.thallo_hip_sfs_march_debug_set(2, 1); L.thallo_hip_sfs_march_debug_set(6, 1)
    def f(v=7.0):      # (the flat kernels work on float4s: the padding behind the N unknowns is zero, as in the solver's vectors)
        t = torch.zeros(N + 64, device="cuda"); t[:N] = v; return t
    r0, z0, pp0, d0, dg0 = f(), f(), f(), f(), f()
    aN0 = torch.zeros(1024, device="cuda")
    nb = L.thallo_hip_sfs_pcg_init(W, H, 0, H, inst.yoff, inst.Hg, inst.hp, vp(inst.X), vp(inst.D), vp(inst.Gp), vp(inst.Fw), vp(inst.fl), vp(inst.U), vp(inst.R),
                                   vp(r0), vp(z0), vp(pp0), vp(d0), vp(dg0), vp(aN0), None)
    assert nb > 0, nb
    ssq0, ctc0, pre0, b0 = f(), f(), f(), f()
    radius, lo, hi = C.c_float(1e4), C.c_float(1e-6), C.c_float(1e32)
    L.thallo_hip_lm_finalize_diagonal.argtypes = [C.c_void_p] * 7 + [C.c_long, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    nbf = L.thallo_hip_lm_finalize_diagonal(vp(dg0), vp(ssq0), vp(ctc0), vp(pre0), vp(r0), vp(b0), vp(z0), N, radius, lo, hi, 1, 0, vp(aN0), None)
    assert nbf > 0, nbf
    torch.cuda.synchronize()
    a0 = float(aN0[:nbf].double().sum())
    r1, z1, pp1, d1 = f(), f(), f(), f()
    ssq1, ctc1, pre1, b1 = f(), f(), f(), f()
    aN1 = torch.zeros(1024, device="cuda")
    L.thallo_hip_sfs_pcg_init_lm.argtypes = [C.c_int] * 6 + [C.c_void_p] * 14 + [C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    nb1 = L.thallo_hip_sfs_pcg_init_lm(W, H, 0, H, inst.yoff, inst.Hg, inst.hp, vp(inst.X), vp(inst.D), vp(inst.Gp), vp(inst.Fw), vp(inst.fl), vp(r1), vp(z1), vp(pp1), vp(d1),
                                       vp(ssq1), vp(ctc1), vp(pre1), vp(b1), radius, lo, hi, 1, vp(aN1), None)
    assert nb1 > 0, nb1
    torch.cuda.synchronize()
    a1 = float(aN1[:nb1].double().sum())
    eq = lambda u, v: bool(torch.equal(u[:N], v[:N]))
    close = lambda u, v: float((u[:N] - v[:N]).abs().max()) <= 2e-6 * float(v[:N].abs().max())
    ok1 = eq(r1, r0) and eq(d1, d0) and eq(pp1, pp0) and eq(b1, b0) and eq(ssq1, ssq0) and close(ctc1, ctc0) and close(pre1, pre0) and close(z1, z0) and abs(a1 - a0) <= 1e-5 * abs(a0)
    msg = (f"init + finalize in one launch: r / delta / p_prev / b / SSq bitwise {eq(r1, r0) and eq(d1, d0) and eq(pp1, pp0) and eq(b1, b0) and eq(ssq1, ssq0)}, CtC / M^-1 / z bitwise "
           f"{eq(ctc1, ctc0)} {eq(pre1, pre0)} {eq(z1, z0)} (to rounding {close(ctc1, ctc0) and close(pre1, pre0) and close(z1, z0)}), alphaN rel {abs(a1 - a0) / abs(a0):.1e}")
    # model cost: state says "not stopped", L = 3: kl = 2 -> p_even
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    delta = torch.randn(N + 64, device="cuda", generator=g) * 1e-3; pe = torch.randn(N + 64, device="cuda", generator=g) * 1e-3; po = torch.randn(N + 64, device="cuda", generator=g) * 1e-3
    delta[N:] = 0; pe[N:] = 0; po[N:] = 0
    words = torch.tensor([2.0, 3.0, 5.0, 7.0, 0.75, 1.5, 9.0, 9.0], device="cuda")       # alphaN_k at [2k], alphaD_k at [2k + 1]: alpha_2 = 0.5
    state = torch.zeros(8, device="cuda")
    ok2, msgs = True, []
    for gate, done, Lm in ((0, 0, 3), (1, 2, 3), (0, 0, 0)):
        state.zero_(); st = state.view(torch.int32); st[1] = gate; st[2] = done
        dA = delta.clone()
        L.thallo_hip_lm_owed_delta.argtypes = [C.c_void_p] * 3 + [C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        assert L.thallo_hip_lm_owed_delta(vp(dA), vp(pe), vp(po), N, vp(words), vp(words[1:]), 2, vp(state), Lm, None) == 0
        Ad = f(); pa = torch.zeros(1024, device="cuda"); pb = torch.zeros(1024, device="cuda")
        na = L.thallo_hip_sfs_apply_jtj(W, H, 0, H, inst.yoff, inst.Hg, inst.hp, vp(inst.Gp), vp(inst.Fw), vp(inst.fl), vp(inst.U), vp(inst.R), vp(dA), vp(Ad), vp(pa), None)
        L.thallo_hip_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        nd = L.thallo_hip_dot(vp(dA), vp(b0), N, vp(pb), None)
        assert na > 0 and nd > 0, (na, nd)
        torch.cuda.synchronize()
        jj0, db0 = float(pa[:na].double().sum()), float(pb[:nd].double().sum())
        dB = f(); qa = torch.zeros(1024, device="cuda"); qb = torch.zeros(1024, device="cuda")
        L.thallo_hip_sfs_lm_model_cost.argtypes = [C.c_int] * 6 + [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        Xc = inst.X.clone().reshape(-1); pX = torch.full((N,), 7.0, device="cuda")           # (... with savePreviousUnknowns + PCGLinearUpdate riding along)
        nm = L.thallo_hip_sfs_lm_model_cost(W, H, 0, H, inst.yoff, inst.Hg, inst.hp, vp(inst.Gp), vp(inst.Fw), vp(inst.fl), vp(delta), vp(dB), vp(pe), vp(po), vp(b0), vp(words), vp(words[1:]), 2,
                                            vp(state), Lm, vp(qa), vp(qb), vp(Xc), vp(pX), None)
        assert nm > 0, nm
        torch.cuda.synchronize()
        x_ok = bool(torch.equal(pX, inst.X.reshape(-1)) and torch.equal(Xc, inst.X.reshape(-1) + dA[:N]))
        jj1, db1 = float(qa[:nm].double().sum()), float(qb[:nm].double().sum())
        rel = lambda u, v: abs(u - v) / max(abs(v), 1e-30)
        good = eq(dB, dA) and x_ok and rel(jj1, jj0) < 1e-4 and rel(db1, db0) < 1e-4
        ok2 = ok2 and good
        msgs.append(f"(gate {gate}, done {done}, L {Lm}) delta bitwise {eq(dB, dA)}, prevX / X + delta bitwise {x_ok}, dJJd rel {rel(jj1, jj0):.1e}, db rel {rel(db1, db0):.1e}")
    L.thallo_hip_sfs_march_debug_set(2, -1); L.thallo_hip_sfs_march_debug_set(6, 0)
    return ok1 and ok2, msg + "; model cost in one launch: " + "; ".join(msgs)


def check():
    cases = [(130, 67, {}), (64, 48, {}), (3, 3, {}), (2, 2, {}), (61, 5, {}), (126, 9, {}), (200, 131, dict(ra=7, rb=90, yoff=50, Hg=400)), (121, 40, dict(ra=2, rb=38, yoff=0, Hg=40)),
             (250, 40, dict(ra=2, rb=38, yoff=0, Hg=40)), (640, 480, {}), (512, 512, {}), (2048, 2048, {})]
    ok = True
    for W, H, kw in cases:
        inst = Inst(W, H, **kw)
        print(f"{W}x{H} {kw} {inst.pre_msg}", flush=True)
        print(f"{W}x{H} {kw} {inst.pair_msg}", flush=True)
        ok = ok and inst.pre_ok and inst.pair_ok
        (ra_, za, pa, da, ga), na = run_init(inst, False)
        (rb_, zb, pb, db, gb), nb_ = run_init(inst, True)
        (_, _, _, _, gc), _ = run_init(inst, True, diag_by_march=False)
        md = np.abs(ra_ - rb_).max() / np.abs(ra_).max()
        dd = np.abs(ga - gb).max() / np.abs(ga).max()
        good = md < 2e-6 and dd < 2e-6 and np.array_equal(ga, gc) and np.array_equal(rb_, zb) and np.array_equal(pa, pb) and np.array_equal(da, db) and abs(na - nb_) <= 1e-5 * abs(na)
        print(f"{W}x{H} {kw} init: max diff / max {md:.2e}, z == r {np.array_equal(rb_, zb)}, p_prev / delta equal {np.array_equal(pa, pb) and np.array_equal(da, db)}, "
              f"alphaN rel {abs(na - nb_) / abs(na):.1e}; LM diagonal by the marching kernel vs k_diag: max diff / max {dd:.2e}, bitwise {np.array_equal(ga, gb)}", flush=True)
        ok = ok and good
        if inst.pair:
            (rp_, zp, pp, dp, gp), np_ = run_init(inst, True, pair=True)
            mdp = np.abs(ra_ - rp_).max() / np.abs(ra_).max(); ddp = np.abs(ga - gp).max() / np.abs(ga).max()
            goodp = mdp < 2e-5 and ddp < 2e-5 and np.array_equal(rp_, zp) and np.array_equal(pa, pp) and np.array_equal(da, dp) and abs(na - np_) <= 2e-5 * abs(na)
            print(f"{W}x{H} {kw} init, pixel pairs on packed planes: max diff / max {mdp:.2e}, z == r {np.array_equal(rp_, zp)}, alphaN rel {abs(na - np_) / abs(na):.1e}, LM diagonal max diff / max {ddp:.2e}", flush=True)
            okl, msgl = check_pair_lm(inst)
            print(f"{W}x{H} {kw} {msgl}", flush=True)
            ok = ok and goodp and okl
        for variant in ("plain", "sums", "lm"):
            a, ad_a, s_a = run(inst, variant, False)
            if inst.pair:      # the pixel-pair kernel on the packed planes against the tile kernel on the legacy planes: the partials in G differ in their last digits (closed form vs duals)
                c, ad_c, s_c = run(inst, variant, True, pair=True)
                rows = slice(inst.ra * W, inst.rb * W)
                relp = lambda u, v: abs(u - v) / max(abs(u), 1e-30)
                mdc = np.abs(a[rows] - c[rows]).max() / np.abs(a[rows]).max()
                unt = np.array_equal(a[:inst.ra * W], c[:inst.ra * W]) and np.array_equal(a[inst.rb * W:], c[inst.rb * W:])
                line = f"{W}x{H} {kw} {variant}, pixel pairs: max diff / max {mdc:.2e}, outside rows equal {unt}, alphaD rel {relp(ad_a, ad_c):.1e}"
                if variant == "sums":
                    line += ", sums rel " + " ".join(f"{relp(u, v):.1e}" for u, v in zip(s_a, s_c))
                    ok = ok and all(relp(u, v) < 2e-5 for u, v in zip(s_a[1:], s_c[1:])) and relp(s_a[0], s_c[0]) < 1e-12
                print(line, flush=True)
                ok = ok and mdc < 2e-5 and unt and relp(ad_a, ad_c) < 1e-4 and np.isfinite(c).all()
            b, ad_b, s_b = run(inst, variant, True)
            rows = slice(inst.ra * W, inst.rb * W)
            same = np.array_equal(a[rows].view(np.uint32), b[rows].view(np.uint32))
            untouched = np.array_equal(a[:inst.ra * W], b[:inst.ra * W]) and np.array_equal(a[inst.rb * W:], b[inst.rb * W:])
            scale = np.abs(a[rows]).max()
            md = np.abs(a[rows] - b[rows]).max() / scale
            rel = lambda u, v: abs(u - v) / max(abs(u), 1e-30)
            line = f"{W}x{H} {kw} {variant}: bitwise {same}, max diff / max {md:.2e}, outside rows equal {untouched}, alphaD rel {rel(ad_a, ad_b):.1e}"
            if variant == "sums":
                line += ", sums rel " + " ".join(f"{rel(u, v):.1e}" for u, v in zip(s_a, s_b))
                ok = ok and all(rel(u, v) < 2e-6 for u, v in zip(s_a[1:], s_b[1:])) and rel(s_a[0], s_b[0]) < 1e-12      # (N = sum r.r: exact products, double sums)
            print(line, flush=True)
            ok = ok and md < 2e-6 and untouched and rel(ad_a, ad_b) < 1e-4 and np.isfinite(b).all()
    L.thallo_hip_sfs_march_debug_set(2, -1); L.thallo_hip_sfs_march_debug_set(6, -1)
    print("CHECK", "OK" if ok else "FAILED")
    return 0 if ok else 1


def time_one(inst, variant, march, reps=50):
    L.thallo_hip_sfs_march_debug_set(2, 1 if march else 0)
    N = inst.W * inst.H
    Ap = torch.empty(N, device="cuda"); aD = torch.zeros(1024, device="cuda"); s3 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
    for _ in range(5):
        nb = inst.apply(variant, Ap, aD, s3)
    if nb <= 0:
        return None, nb
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        inst.apply(variant, Ap, aD, s3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, nb


def timing(W, H):
    inst = Inst(W, H)
    for variant in ("plain", "sums", "lm"):
        t, nb = time_one(inst, variant, False)
        print(f"{W}x{H} {variant}: tile kernel {t:.1f} us ({nb} workgroups)", flush=True)
        L.thallo_hip_sfs_march_debug_set(0, 0); L.thallo_hip_sfs_march_debug_set(1, 0)
        t, nb = time_one(inst, variant, True)
        print(f"{W}x{H} {variant}: marching kernel (default grid) {t:.1f} us ({nb} workgroups) = {33 * W * H / t / 1e6:.2f} TB/s of the 33 B/pixel", flush=True)
    cp = torch.zeros(1024, device="cuda")
    for fused in (0, 1):
        G = torch.empty(4 * W * H, device="cuda"); Wt = torch.empty(2 * W * H, device="cuda"); fl = torch.empty(W * H + 4, dtype=torch.uint8, device="cuda")
        def call():
            if fused:
                return L.thallo_hip_sfs_precompute_cost(W, H, 0, H, 0, H, inst.hp, vp(inst.X), vp(inst.D), vp(inst.Im), vp(inst.mR), vp(inst.mC), vp(G), vp(Wt), vp(fl), 0, H, vp(cp), None)
            L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, inst.hp, vp(inst.X), vp(inst.D), vp(inst.Im), vp(inst.mR), vp(inst.mC), vp(G), vp(Wt), vp(fl), None)
            return L.thallo_hip_sfs_cost(W, H, 0, H, 0, H, inst.hp, vp(inst.X), vp(inst.D), vp(G), vp(Wt), vp(fl), vp(cp), None)
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        print(f"{W}x{H} precompute + computeCost: {'one launch' if fused else 'two launches'} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    for march in (0, 1):
        for _ in range(3):
            inst.precompute(march)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.thallo_hip_sfs_march_debug_set(4, march)
        G = torch.empty(4 * W * H, device="cuda"); Wt = torch.empty(2 * W * H, device="cuda"); fl = torch.empty(W * H + 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, inst.hp, vp(inst.X), vp(inst.D), vp(inst.Im), vp(inst.mR), vp(inst.mC), vp(G), vp(Wt), vp(fl), None)
        e1.record(); torch.cuda.synchronize()
        L.thallo_hip_sfs_march_debug_set(4, 1)
        print(f"{W}x{H} precompute: {'marching kernel' if march else 'k_precompute'} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    for march, dg, lab in ((False, 0, "tile kernel"), (True, 0, "marching kernel"), (True, 1, "marching kernel + k_diag (LM)"), (True, 2, "marching kernel incl. the LM diagonal")):
        L.thallo_hip_sfs_march_debug_set(2, 1 if march else 0); L.thallo_hip_sfs_march_debug_set(3, 1 if dg == 2 else 0)
        o = [torch.empty(W * H, device="cuda") for _ in range(5)]; aN = torch.zeros(1024, device="cuda")
        call = lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, inst.hp, vp(inst.X), vp(inst.D), vp(inst.G), vp(inst.Wt), vp(inst.fl), vp(inst.U), vp(inst.R),
                                                 vp(o[0]), vp(o[1]), vp(o[2]), vp(o[3]), vp(o[4]) if dg else None, vp(aN), None)
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        print(f"{W}x{H} PCGInit1 J^T F: {lab} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    L.thallo_hip_sfs_march_debug_set(3, 1)
    for wgcu in (1, 2, 3, 4):
        L.thallo_hip_sfs_march_debug_set(0, 0); L.thallo_hip_sfs_march_debug_set(1, wgcu)
        t, nb = time_one(inst, "sums", True)
        print(f"  sums, grid for {wgcu} workgroups / CU: {t:.1f} us ({nb} workgroups)", flush=True)
    L.thallo_hip_sfs_march_debug_set(1, 0)
    for rows in (8, 12, 16, 24, 32, 48, 64, 128):
        L.thallo_hip_sfs_march_debug_set(0, rows)
        t, nb = time_one(inst, "sums", True)
        print(f"  sums, {rows} rows per wave: " + (f"{t:.1f} us ({nb} workgroups)" if t else f"not launchable ({nb})"), flush=True)
    L.thallo_hip_sfs_march_debug_set(0, 0); L.thallo_hip_sfs_march_debug_set(2, -1)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(check())
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048)
    timing(W, H)
