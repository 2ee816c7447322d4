"""shape_from_shading, round 6: the pixel-pair marching kernels on the packed planes against the one-pixel-per-lane kernels (energy_sfs.hip) through the C-ABI shim.
Launch times (HIP events around back-to-back launches) of the GN iteration (thallo_hip_sfs_pcg_iter, delta left to the ring), the LM iteration, precompute (+ cost),
PCGInit1 (+ the LM diagonal, + FinalizeDiagonal) and the LM model cost, with a sweep of the pair kernel's grid (workgroups per CU, rows per wave) and prefetch depth.
Runs on the GPU box:  python tools/sfs_pair_time.py [W H]"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from thallo_amd import api, synthetic as syn

L = api.lib()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
L.thallo_hip_sfs_march_debug_set.argtypes = [C.c_int, C.c_int]; L.thallo_hip_sfs_march_debug_set.restype = None


class Sum(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("count", C.c_int)]


class Fin(C.Structure):
    _fields_ = [("alphaN", Sum), ("tickets", C.c_void_p), ("alphaD_word", C.c_void_p), ("betaN_word", C.c_void_p)]


def timed(fn, reps=40, warm=5):
    for _ in range(warm):
        rc = fn()
        assert rc is None or rc >= 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


_trash = None


def timed_cold(fn, reps=12):
    """one launch at a time behind ~1 GB of unrelated traffic (the 256 MB Infinity Cache holds nothing of the kernel's planes: what a launch sees inside a solver step), events
    recorded around the single launch (dispatch gap included)"""
    global _trash
    if _trash is None:
        _trash = (torch.empty(128 << 20, device="cuda"), torch.empty(128 << 20, device="cuda"))
    fn(); tot = 0.0
    for _ in range(reps):
        _trash[1].copy_(_trash[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main(W, H):
    N = W * H
    p = syn.shape_from_shading(W, H)
    hp = (C.c_float * 16)(*[float(x) for x in p[:16]])
    X, D, Im, mR, mC = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p[16:21]]
    G = torch.empty(4 * N + 16, device="cuda"); Wt = torch.empty(2 * N + 16, device="cuda"); fl = torch.empty(N + 256, dtype=torch.uint8, device="cuda")
    f = lambda: torch.zeros(N + 64, device="cuda")
    r = [f(), f()]; A = [f(), f()]; P = [f(), f()]; delta = f(); ctc = torch.rand(N + 64, device="cuda") * 50; b = f(); pre = torch.rand(N + 64, device="cuda") + 0.5
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    r[0].copy_(torch.randn(N + 64, device="cuda", generator=g) * 1e-3); P[0].copy_(r[0]); b.copy_(r[0])
    aN = torch.ones(1024, device="cuda"); aD = torch.ones(1024, device="cuda"); bN = torch.ones(1024, device="cuda")
    s3 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda"); q3 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
    cost = torch.zeros(1024, device="cuda"); z = f(); pp = f(); dg = f(); ssq = f(); U = torch.empty(2 * N, device="cuda"); R = torch.empty(3 * N, device="cuda")
    tick = torch.zeros(1024, dtype=torch.int32, device="cuda"); words = torch.ones(8, device="cuda"); state = torch.zeros(8, device="cuda")
    one = Sum(vp(aN).value, 1)
    nofin = Fin(Sum(None, 0), None, None, None)
    out = {}

    def planes(pair):
        L.thallo_hip_sfs_march_debug_set(6, 1 if pair else 0)
        rc = L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), None)
        assert rc == 0, rc

    L.thallo_hip_sfs_pcg_iter.argtypes = [C.c_int] * 6 + [C.c_void_p] * 11 + [C.c_int, Sum, Sum, Sum, C.c_void_p, C.c_void_p, Fin, C.c_void_p]
    L.thallo_hip_sfs_pcg_iter_lm.argtypes = [C.c_int] * 6 + [C.c_void_p] * 14 + [C.c_int, Sum, Sum, Sum, C.c_void_p, C.c_void_p, C.c_void_p, Fin, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    cur = [0]

    def gn_iter(with_delta=False):
        c = cur[0]; cur[0] ^= 1
        return L.thallo_hip_sfs_pcg_iter(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(r[c]), vp(r[c ^ 1]), vp(A[c]), vp(A[c ^ 1]), vp(P[c]), vp(P[c ^ 1]), vp(delta) if with_delta else None, 0,
                                         one, one, one, vp(aD), vp(s3), nofin, None)

    def lm_iter():
        c = cur[0]; cur[0] ^= 1
        fin = nofin      # (partials only: with tickets the launch's last workgroup would apply the zeta test to this tool's meaningless sums, set the gate, and every later launch would return at once)
        return L.thallo_hip_sfs_pcg_iter_lm(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(r[c]), vp(r[c ^ 1]), vp(A[c]), vp(A[c ^ 1]), vp(P[c]), vp(P[c ^ 1]), vp(delta), vp(ctc), vp(b), vp(pre), 0,
                                            one, one, one, vp(aD), vp(s3), vp(q3), fin, vp(state), 0, C.c_float(0.0), None)

    def set_tune(rows=0, wgcu=0, depth=0):
        L.thallo_hip_sfs_march_debug_set(0, rows); L.thallo_hip_sfs_march_debug_set(1, wgcu); L.thallo_hip_sfs_march_debug_set(7, depth)

    for pair in (0, 1):
        if pair and not (W % 2 == 0):
            break
        planes(pair)
        tag = "pair" if pair else "legacy"
        set_tune()
        out[tag] = {
            "gn_iter_ring_us": timed(gn_iter), "gn_iter_with_delta_us": timed(lambda: gn_iter(True)), "lm_iter_us": timed(lm_iter),
            "precompute_us": timed(lambda: L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), None), 20),
            "precompute_cost_us": timed(lambda: L.thallo_hip_sfs_precompute_cost(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), 0, H, vp(cost), None), 20),
            "init_us": timed(lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(r[0]), vp(z), vp(pp), vp(delta), None, vp(aN), None), 20),
            "init_diag_us": timed(lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(r[0]), vp(z), vp(pp), vp(delta), vp(dg), vp(aN), None), 20),
            "apply_plain_us": timed(lambda: L.thallo_hip_sfs_apply_jtj(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(P[0]), vp(A[0]), vp(aD), None)),
        }
        print(tag, json.dumps({k: round(v, 2) for k, v in out[tag].items()}), flush=True)
        cold = {
            "gn_iter_ring_us": timed_cold(gn_iter), "lm_iter_us": timed_cold(lm_iter),
            "precompute_us": timed_cold(lambda: L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), None)),
            "precompute_cost_us": timed_cold(lambda: L.thallo_hip_sfs_precompute_cost(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), 0, H, vp(cost), None)),
            "init_us": timed_cold(lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(r[0]), vp(z), vp(pp), vp(delta), None, vp(aN), None)),
            "init_diag_us": timed_cold(lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(r[0]), vp(z), vp(pp), vp(delta), vp(dg), vp(aN), None)),
            "apply_plain_us": timed_cold(lambda: L.thallo_hip_sfs_apply_jtj(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(P[0]), vp(A[0]), vp(aD), None)),
        }
        out[tag + "_cold"] = cold
        print(tag, "COLD (one launch behind 1 GB of other traffic, dispatch gap included)", json.dumps({k: round(v, 2) for k, v in cold.items()}), flush=True)
        if pair:
            for wgcu in (1, 2, 3, 4):
                set_tune(0, wgcu, 0)
                print("  pair COLD, grids for %d workgroups / CU: precompute %.2f, precompute + cost %.2f, init %.2f, GN iteration %.2f, LM iteration %.2f us" % (wgcu,
                      timed_cold(lambda: L.thallo_hip_sfs_precompute(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), None)),
                      timed_cold(lambda: L.thallo_hip_sfs_precompute_cost(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(Im), vp(mR), vp(mC), vp(G), vp(Wt), vp(fl), 0, H, vp(cost), None)),
                      timed_cold(lambda: L.thallo_hip_sfs_pcg_init(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(U), vp(R), vp(r[0]), vp(z), vp(pp), vp(delta), None, vp(aN), None)),
                      timed_cold(gn_iter), timed_cold(lm_iter)), flush=True)
            set_tune()
        if pair:
            L.thallo_hip_sfs_pcg_init_lm.argtypes = [C.c_int] * 6 + [C.c_void_p] * 14 + [C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
            L.thallo_hip_sfs_lm_model_cost.argtypes = [C.c_int] * 6 + [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            t1 = timed(lambda: L.thallo_hip_sfs_pcg_init_lm(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(r[0]), vp(z), vp(pp), vp(delta), vp(ssq), vp(ctc), vp(pre), vp(b),
                                                            C.c_float(1e4), C.c_float(1e-6), C.c_float(1e32), 1, vp(aN), None), 20)
            t2 = timed(lambda: L.thallo_hip_sfs_lm_model_cost(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(delta), vp(z), vp(P[0]), vp(P[1]), vp(b), vp(words), vp(words[1:]), 2, vp(state), 3,
                                                              vp(aD), vp(bN), vp(X), vp(ssq), None), 20)
            print("pair init + finalize diagonal in one launch %.2f us; model cost in one launch %.2f us" % (t1, t2), flush=True)
            t1c = timed_cold(lambda: L.thallo_hip_sfs_pcg_init_lm(W, H, 0, H, 0, H, hp, vp(X), vp(D), vp(G), vp(Wt), vp(fl), vp(r[0]), vp(z), vp(pp), vp(delta), vp(ssq), vp(ctc), vp(pre), vp(b),
                                                                   C.c_float(1e4), C.c_float(1e-6), C.c_float(1e32), 1, vp(aN), None))
            t2c = timed_cold(lambda: L.thallo_hip_sfs_lm_model_cost(W, H, 0, H, 0, H, hp, vp(G), vp(Wt), vp(fl), vp(delta), vp(z), vp(P[0]), vp(P[1]), vp(b), vp(words), vp(words[1:]), 2, vp(state), 3,
                                                                    vp(aD), vp(bN), vp(X), vp(ssq), None))
            print("  COLD: init + finalize %.2f us; model cost %.2f us" % (t1c, t2c), flush=True)
            out["pair"]["init_fin_us"] = t1; out["pair"]["model_cost_us"] = t2
            ctc.uniform_(0, 50); pre.uniform_(0.5, 1.5)
            for depth in ((3, 6) if os.environ.get("SP_DEPTH6") else (3,)):
                for wgcu in (1, 2, 3):
                    set_tune(0, wgcu, depth)
                    print("  pair GN iteration (ring), depth %d, grid for %d workgroups / CU: %.2f us; LM iteration %.2f us" % (depth, wgcu, timed(gn_iter), timed(lm_iter)), flush=True)
            for rows in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                for depth in ((3, 6) if os.environ.get("SP_DEPTH6") else (3,)):
                    set_tune(rows, 0, depth)
                    try:
                        t = timed(gn_iter)
                    except AssertionError:
                        continue
                    print("  pair GN iteration (ring), %d rows per wave, depth %d: %.2f us" % (rows, depth, t), flush=True)
            set_tune()
    L.thallo_hip_sfs_march_debug_set(6, -1)
    by = {"legacy": 49.0, "pair": 40.0}
    for tag in out:
        t = out[tag]["gn_iter_ring_us"]
        if tag.endswith("_cold"): continue
        print("%s: GN iteration %.2f us = %.2f TB/s on its own %.0f B/pixel (%.3f of 8 TB/s); on the legacy formulation's 49 B/pixel %.2f TB/s" % (
            tag, t, by[tag] * N / t / 1e6, by[tag], by[tag] * N / t / 1e6 / 8.0, 49.0 * N / t / 1e6))
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 2048)
    main(W, H)
