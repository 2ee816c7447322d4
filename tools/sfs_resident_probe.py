"""sfs_resident_probe.py -- shape_from_shading's resident PCG loop against one launch per iteration with the same rows per wave: where do the unknowns differ?
    python tools/sfs_resident_probe.py [W H nit lit]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import thallo_amd
from thallo_amd import api, synthetic as syn
from helpers import copy_params, to_device, to_host
W, H, nit, lit = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (640, 480, 3, 10))]
L = thallo_amd.lib(); L.thallo_hip_sfs_resident_rows.restype = C.c_int
R = L.thallo_hip_sfs_resident_rows(W, H); print("R", R)
p = syn.shape_from_shading(W, H)
os.environ["THALLO_DELTA_PLANES"] = "0"
def run(resident, n):
    os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
    L.thallo_hip_sfs_march_debug_set(0, R)
    dev = to_device(copy_params(p))
    lm = os.environ.get("SRP_LM") == "1"            # the LM step's resident launch instead of the GN loop's
    s = api.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), **({"solverkind": "levenberg_marquardt"} if lm else {}))
    if lm: s.enable_lm()
    s.set_solver_parameters(nIterations=n, lIterations=lit, **({"q_tolerance": 0.0} if lm else {}))
    prm = s.make_params(dev); s.init(prm)
    tr = []
    while s.step(prm): tr.append(s.alpha_beta_trace())
    c = s.current_cost(); s.close()
    L.thallo_hip_sfs_march_debug_set(0, 0)
    return to_host(dev[16]).copy(), tr, c
for n in (range(1, nit + 1) if not (len(sys.argv) > 5 and sys.argv[5] in ("kernel", "stamps", "ab", "soak")) else []):
    for rep in range(2):
        xa, ta, ca = run(True, n); xb, tb, cb = run(False, n)
        d = np.argwhere(xa != xb)
        print("steps", n, "rep", rep, "cost", ca, cb, "differing pixels", len(d), "first", d[:6].tolist(), "trace equal", ta == tb,
              "max rel", float(np.abs(xa - xb).max() / np.abs(xb).max()))
        if len(d):
            ys = np.unique(d[:, 0]); xs = np.unique(d[:, 1])
            print("   rows", ys[:20].tolist(), "... cols", xs[:20].tolist(), "rows mod R", np.unique(ys % R).tolist())


def kernel_level(state_steps=2):
    """One iteration at kernel level from the solver's state after `state_steps` GN steps: resident L = 1 against thallo_hip_sfs_pcg_iter(first) on the same planes."""
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    class FinT(C.Structure): _fields_ = [("alphaN", api.SumT), ("tickets", C.c_void_p), ("aD", C.c_void_p), ("bN", C.c_void_p)]
    x2, _, _ = run(False, state_steps) if state_steps else (p[16], None, None)
    N = W * H
    hp = (C.c_float * 16)(*[float(v) for v in p[:16]])
    X = torch.from_numpy(np.ascontiguousarray(x2)).cuda()
    D, Im, mR, mC = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p[17:21]]
    Gp = torch.zeros(4 * N, device="cuda"); Fw = torch.zeros(2 * N, device="cuda"); flx = torch.zeros(N + 4, dtype=torch.uint8, device="cuda")
    L.thallo_hip_sfs_march_debug_set(6, 1)
    assert L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(Gp), vp(Fw), vp(flx), None) == 0
    f = lambda: torch.zeros(N + 64, device="cuda")
    r0, z0, pp0, d0 = f(), f(), f(), f(); aN0 = torch.zeros(1024, device="cuda")
    nb = L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Gp), vp(Fw), vp(flx), None, None, vp(r0), vp(z0), vp(pp0), vp(d0), None, vp(aN0), None)
    assert nb > 0, nb
    # marching, first iteration, rows per wave = R
    L.thallo_hip_sfs_march_debug_set(0, R)
    r1, A1, p1 = f(), f(), f(); aD = torch.zeros(1024, device="cuda"); s3 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
    S = lambda t, n: api.SumT(t.data_ptr(), n)
    L.thallo_hip_sfs_pcg_iter.argtypes = [C.c_int] * 6 + [C.c_void_p] * 11 + [C.c_int, api.SumT, api.SumT, api.SumT, C.c_void_p, C.c_void_p, FinT, C.c_void_p]
    fin = FinT(S(aN0, nb), None, None, None)
    nbm = L.thallo_hip_sfs_pcg_iter(W, H, 0, H, 0, H, hp, vp(Gp), vp(Fw), vp(flx), vp(r0), vp(r1), vp(A1), vp(A1), vp(pp0), vp(p1), vp(d0), 1, S(aN0, nb), S(aN0, nb), S(aN0, nb), vp(aD), vp(s3), fin, None)
    L.thallo_hip_sfs_march_debug_set(0, 0)
    assert nbm > 0, nbm
    torch.cuda.synchronize()
    # resident, L = 1
    L.thallo_hip_sfs_resident_bytes.restype = C.c_long
    xb = torch.zeros(L.thallo_hip_sfs_resident_bytes(W, H) // 4 + 16, dtype=torch.int32, device="cuda")
    r2, A2, p2, d2 = r0.clone(), f(), f(), f(); words = torch.zeros(64, device="cuda")
    L.thallo_hip_sfs_pcg_resident.argtypes = [C.c_int] * 3 + [C.c_void_p] * 9 + [api.SumT, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    nbr = L.thallo_hip_sfs_pcg_resident(W, H, 0, hp, vp(Gp), vp(Fw), vp(r2), vp(pp0), vp(r2), vp(A2), vp(p2), vp(d2), S(aN0, nb), vp(words), None, vp(xb), 1, None)
    assert nbr > 0, nbr
    torch.cuda.synchronize()
    print("workgroups", nbm, nbr, "p equal", bool(torch.equal(p1[:N], p2[:N])), "r equal", bool(torch.equal(r1[:N], r2[:N])))
    dA = torch.nonzero(A1[:N] != A2[:N]).flatten().cpu().numpy()
    print("A p differs at", len(dA), "pixels", [(int(i) // W, int(i) % W, float(A1[i]), float(A2[i])) for i in dA[:12]])
    if len(dA):
        ys, xs = dA // W, dA % W
        print("  rows mod R", np.unique(ys % R).tolist(), "cols mod 124", np.unique(xs % 124).tolist()[:40])
    rec = xb.view(torch.int64)[32:32 + 1024 * 8].view(1024, 8).cpu().numpy()
    ad_r = (rec[:nbr, 0] & 0xffffffff).astype(np.uint32).view(np.float32)
    hi_lo = lambda a, b: ((rec[:nbr, a] & 0xffffffff).astype(np.uint64) << np.uint64(32) | (rec[:nbr, b] & 0xffffffff).astype(np.uint64)).view(np.float64)
    s3h = s3[:3 * nbm].cpu().numpy().reshape(-1, 3)
    print("per-workgroup alphaD partials equal", bool((ad_r == aD[:nbm].cpu().numpy()).all()), "N / S1 / S2 equal", bool((hi_lo(1, 2) == s3h[:, 0]).all()), bool((hi_lo(3, 4) == s3h[:, 1]).all()), bool((hi_lo(5, 6) == s3h[:, 2]).all()))
    bad = np.nonzero(ad_r != aD[:nbm].cpu().numpy())[0]; print("  differing slots", bad[:10].tolist(), [(float(ad_r[i]), float(aD[i])) for i in bad[:4]])
    wm = torch.zeros(8, device="cuda")
    L.thallo_hip_pcg_scalars_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, api.SumT, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.thallo_hip_pcg_scalars_finish(vp(aD), vp(s3), nbm, S(aN0, nb), vp(wm), vp(wm[1:]), None) >= 0
    torch.cuda.synchronize(); print("marching words (scalars_finish)", float(wm[0]), float(wm[1]))
    ad_m = aD[:nbm].cpu().numpy(); print("alphaD marching (lane-strided float sum)", float(ad_m.sum()), "resident word", float(words[0]), "betaN", float(words[1]))


if len(sys.argv) > 5 and sys.argv[5] == "kernel":
    kernel_level(int(sys.argv[6]) if len(sys.argv) > 6 else 2)


def stamps():
    """needs the sweep build: make -C thallo_amd/csrc VARIANT=sweep; THALLO_LIB=tools/ab/libThallo_sweep.so"""
    nw = 1024 * 4
    st = torch.zeros(nw * 4 * 8, dtype=torch.int64, device="cuda")
    assert L.thallo_hip_debug_stamps_sfs_resident(C.c_void_p(st.data_ptr())) == 0
    run(True, 2)
    torch.cuda.synchronize()
    t = st.cpu().numpy().reshape(nw, 4, 8).astype(np.float64)
    t = t[t[:, 0, 0] > 0]
    names = ["global poll", "columns via LDS + scalars exchange", "r / p / delta update", "stencil + sums", "publish"]
    d = np.diff(t[:, :, :6], axis=2) / 100.0
    print("waves", len(t), "iteration us", round(float(((t[:, 1:, 0] - t[:, :-1, 0]) / 100.0).mean()), 3))
    for i, n in enumerate(names): print("  %-40s mean %.3f  max-wave %.3f" % (n, d[:, :, i].mean(), d[:, :, i].mean(axis=1).max()))
    print("  entry skew us", round(float((t[:, 1, 0].max() - t[:, 1, 0].min()) / 100.0), 3), "stencil-end skew", round(float((t[:, 1, 4].max() - t[:, 1, 4].min()) / 100.0), 3))
    L.thallo_hip_debug_stamps_sfs_resident(None)


if len(sys.argv) > 5 and sys.argv[5] == "stamps":
    stamps()


def ab_time():
    """GN and LM steps at A/B settings of the exchange layout (thallo_hip_sfs_resident_debug_set(2, bits)), one box: us per PCG iteration through Thallo_ProblemStep"""
    import time
    for lm in (0, 1):
        for bits in [int(b) for b in os.environ.get("SRP_BITS", "0,1,2,3,0").split(",")]:
            L.thallo_hip_sfs_resident_debug_set(2, bits)
            dev = to_device(copy_params(p))
            s = api.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), timing_level=0, **({"solverkind": "levenberg_marquardt"} if lm else {}))
            if lm: s.enable_lm()
            s.set_solver_parameters(nIterations=40, lIterations=lit, **({"q_tolerance": 0.0} if lm else {}))
            prm = s.make_params(dev); s.init(prm)
            for _ in range(3): s.step(prm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): s.step(prm)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("LM" if lm else "GN", "ab bits", bits, "us per PCG iteration %.2f" % (dt / 30 / lit * 1e6), "cost", s.current_cost(), flush=True)
            s.close()
    L.thallo_hip_sfs_resident_debug_set(2, 0)


if len(sys.argv) > 5 and sys.argv[5] == "ab":
    ab_time()


def soak(nsteps=150):
    """many steps: the resident loops against themselves (determinism) and against the launches with the same rows per wave, GN and LM"""
    for lm in (0, 1):
        os.environ["SRP_LM"] = str(lm)
        xa, ta, ca = run(True, nsteps); xb, tb, cb = run(True, nsteps); xc, tc, cc = run(False, nsteps)
        print("LM" if lm else "GN", "steps run", len(ta), len(tb), len(tc), "resident twice equal", bool((xa == xb).all()) and ta == tb, "resident vs launches equal", bool((xa == xc).all()) and ta == tc,
              "costs", ca, cb, cc, "error string", (api.last_error() or "")[:80], flush=True)


if len(sys.argv) > 5 and sys.argv[5] == "soak":
    soak(int(sys.argv[6]) if len(sys.argv) > 6 else 150)
