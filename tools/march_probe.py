"""GPU probe (run through gpurun): the marching one-kernel PCG iteration (thallo_hip_iw_pcg_iter_march) against the LDS-tiled
k_iter on the same inputs -- bitwise comparison of every output plane, then a timing sweep over rows-per-segment / prefetch depth /
non-temporal mask.  Needs the sweep build for the knobs to take effect:  make -C thallo_amd/csrc SWEEP=1 -B
Not part of the product or the tests."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import thallo_amd
from thallo_amd import api, synthetic as syn

W = int(os.environ.get("MB_W", "2048")); H = int(os.environ.get("MB_H", str(W)))
REPS = int(os.environ.get("MB_REPS", "20"))
FIN = os.environ.get("MB_FIN", "1") != "0"       # 0: timing chains without the in-kernel finish (plain partial stores, nothing adds them up)
L = thallo_amd.lib()
L.thallo_hip_vector_elems.restype = C.c_long; L.thallo_hip_vector_elems.argtypes = [C.c_long]
p = syn.image_warping(W, H)
N = W * H; n = 3 * N; na = L.thallo_hip_vector_elems(n)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in p]
f = lambda: torch.zeros(na, dtype=torch.float32, device="cuda")
r0, pre, z, p0, delta0, A0 = [f() for _ in range(6)]
cs = torch.zeros(2 * N, dtype=torch.float32, device="cuda"); flags = torch.zeros(N + 256, dtype=torch.uint8, device="cuda")
parts = torch.zeros(8 * 1024, dtype=torch.float32, device="cuda")
s12 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
irregular = torch.zeros(16, dtype=torch.int32, device="cuda")
tickets = torch.zeros(528, dtype=torch.int32, device="cuda")
words = torch.zeros(16, dtype=torch.float32, device="cuda")
scal = torch.tensor([1.0, 2.0, 0.3, 1.5, 2.5], dtype=torch.float32, device="cuda")     # aN, aD, bN, aN2, aD2 -> alpha .5, beta .3, alpha2 .6
vp, fl = C.c_void_p, C.c_float
S = lambda i: api.SumT(scal.data_ptr() + 4 * i, 1)
nb0 = L.thallo_hip_iw_pcg_init(W, H, 0, H, vp(dev[0].data_ptr()), vp(dev[1].data_ptr()), vp(dev[2].data_ptr()), vp(dev[3].data_ptr()),
                               vp(dev[4].data_ptr()), fl(p[5]), fl(p[6]), vp(r0.data_ptr()), vp(pre.data_ptr()), vp(z.data_ptr()),
                               vp(p0.data_ptr()), vp(delta0.data_ptr()), vp(cs.data_ptr()), vp(flags.data_ptr()), None, vp(irregular.data_ptr()), vp(parts.data_ptr()), None)
assert nb0 > 0
torch.cuda.synchronize()
assert int(irregular[0].item()) == 0


def run(kind, mode, r_in, A_in, p_in, dl, fin=True):
    r_out, A_out, p_out = f(), f(), f()
    d = dl.clone()
    if ((mode >> 1) & 3) == 2:           # p_out holds p_{k-2} before it is overwritten
        p_out.copy_(pk2)
    args_tail = (S(0), S(1), S(2), S(3), S(4), vp(irregular.data_ptr()), vp(parts.data_ptr() + 4096), vp(s12.data_ptr()),
                 vp(tickets.data_ptr()) if fin else None, vp(words.data_ptr()) if fin else None, vp(words.data_ptr() + 4) if fin else None, None)
    if kind == "tile":
        nb = L.thallo_hip_iw_pcg_iter(W, H, 0, H, vp(cs.data_ptr()), vp(dev[2].data_ptr()), vp(flags.data_ptr()), vp(pre.data_ptr()), fl(p[5]), fl(p[6]),
                                      vp(r_in.data_ptr()), vp(r_out.data_ptr()), vp(A_in.data_ptr()), vp(A_out.data_ptr()), vp(p_in.data_ptr()), vp(p_out.data_ptr()),
                                      vp(d.data_ptr()), mode, *args_tail)
    else:
        nb = L.thallo_hip_iw_pcg_iter_march(W, H, 0, H, vp(cs.data_ptr()), vp(flags.data_ptr()), fl(p[5]), fl(p[6]),
                                            vp(r_in.data_ptr()), vp(r_out.data_ptr()), vp(A_in.data_ptr()), vp(A_out.data_ptr()), vp(p_in.data_ptr()), vp(p_out.data_ptr()),
                                            vp(d.data_ptr()), mode, *args_tail)
    assert nb > 0, nb
    torch.cuda.synchronize()
    aD = float(parts[1024:1024 + nb].double().sum().item())
    ss = s12[:3 * nb].view(nb, 3).sum(0).tolist()
    return dict(r=r_out, A=A_out, p=p_out, d=d, aD=aD, s=ss, nb=nb, words=words[:8].tolist())


res = {"W": W, "H": H}
# iteration 0 with the tile kernel gives realistic r1 / Ap1 / p1
it0 = run("tile", 1, r0, A0, p0, delta0)
pk2 = it0["p"].clone()
it1 = run("tile", 2, it0["r"], it0["A"], it0["p"], delta0)       # mode 2: deferred (none)
cmp = {}
for mode, (ri, Ai, pi, di) in {1: (r0, A0, p0, delta0), 2: (it0["r"], it0["A"], it0["p"], delta0), 4: (it1["r"], it1["A"], it1["p"], delta0),
                               0: (it0["r"], it0["A"], it0["p"], delta0)}.items():
    a = run("tile", mode, ri, Ai, pi, di)
    b = run("march", mode, ri, Ai, pi, di)
    e = {}
    for k in ("r", "A", "p", "d"):
        x, y = a[k][:n], b[k][:n]
        e[k + "_mismatch"] = int((x.view(torch.int32) != y.view(torch.int32)).sum().item())
        e[k + "_maxabs"] = float((x - y).abs().max().item())
        e[k + "_ref_maxabs"] = float(x.abs().max().item())
    e["aD_rel"] = abs(a["aD"] - b["aD"]) / max(abs(a["aD"]), 1e-30)
    e["s_rel"] = [abs(u - v) / max(abs(u), 1e-30) for u, v in zip(a["s"], b["s"])]
    e["nb"] = [a["nb"], b["nb"]]
    e["words"] = [a["words"][:2] + a["words"][4:6], b["words"][:2] + b["words"][4:6]]
    cmp[f"mode{mode}"] = e
res["compare"] = cmp
print(json.dumps(res, indent=1)); sys.stdout.flush()


def timeit(kind, reps=REPS, modes=(4, 2), pingpong=True, hook=None):
    """alternate modes 4 / 2 like the PCG loop does, ping-ponging the buffers (pingpong=False: always a -> b, so no launch reads what the previous one wrote)"""
    bufs = [[it0["r"].clone(), it0["A"].clone(), it0["p"].clone()], [f(), f(), f()]]
    d = delta0.clone()
    fn = L.thallo_hip_iw_pcg_iter if kind == "tile" else L.thallo_hip_iw_pcg_iter_march

    def one(k):
        if hook is not None:
            hook(k)
        a, b = (bufs[k & 1], bufs[(k & 1) ^ 1]) if pingpong else (bufs[0], bufs[1])
        mode = modes[k % len(modes)]
        tail = (vp(a[0].data_ptr()), vp(b[0].data_ptr()), vp(a[1].data_ptr()), vp(b[1].data_ptr()), vp(a[2].data_ptr()), vp(b[2].data_ptr()), vp(d.data_ptr()), mode,
                S(0), S(1), S(2), S(3), S(4), vp(irregular.data_ptr()), vp(parts.data_ptr() + 4096), vp(s12.data_ptr()),
                vp(tickets.data_ptr()) if FIN else None, vp(words.data_ptr()) if FIN else None, vp(words.data_ptr() + 4) if FIN else None, None)
        if kind == "tile":
            rc = fn(W, H, 0, H, vp(cs.data_ptr()), vp(dev[2].data_ptr()), vp(flags.data_ptr()), vp(pre.data_ptr()), fl(p[5]), fl(p[6]), *tail)
        else:
            rc = fn(W, H, 0, H, vp(cs.data_ptr()), vp(flags.data_ptr()), fl(p[5]), fl(p[6]), *tail)
        assert rc > 0, rc
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(2 * reps):
        one(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


def time_stream(with_delta, ntm, per_cu, reps=REPS, pingpong=True, span=1, alternate=False):
    a, b = [it0["r"].clone(), it0["A"].clone(), it0["p"].clone()], [f(), f(), f()]
    d = delta0.clone()

    def one(k):
        x, y = ((a, b) if k & 1 else (b, a)) if pingpong else (a, b)
        rc = L.thallo_hip_iw_stream_ref(W, H, vp(cs.data_ptr()), vp(flags.data_ptr()), vp(x[0].data_ptr()), vp(y[0].data_ptr()), vp(x[1].data_ptr()), vp(y[1].data_ptr()),
                                        vp(x[2].data_ptr()), vp(y[2].data_ptr()), vp(d.data_ptr()), with_delta, ntm, per_cu, span + (1000 if alternate and (k & 1) else 0), None)
        assert rc == 0, rc
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(2 * reps):
        one(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


def cfg(depth, nt, occ, dbg, rows=0, mp=0):
    for what, v in ((0, rows), (1, depth), (2, nt), (3, occ), (4, dbg), (5, mp)):
        L.thallo_hip_march_debug_set(what, v)


if os.environ.get("MB_MODE") == "rows":          # rows-per-segment sweep at the current size (small images / slabs)
    out = {"tile_us": round(timeit("tile"), 2)}
    for rows in (0, 2, 3, 4, 5, 6, 8, 12, 16, 24, 36):
        L.thallo_hip_march_debug_set(0, rows)
        try:
            out[f"march_rows{rows}_us"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2)]
        except AssertionError as e:
            out[f"march_rows{rows}_us"] = "launch refused"
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "stamps":        # stamps build (THALLO_LIB=tools/ab/libThallo_stamps.so, see csrc/Makefile): phase time stamps (tile kernel: thread 0 of each workgroup; marching kernel: lane 0 of each wave), last launch of a chain
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    assert L.thallo_hip_debug_stamps(vp(stamps.data_ptr())) == 0 and L.thallo_hip_debug_stamps_march(vp(stamps.data_ptr())) == 0
    if os.environ.get("MB_ROWS"):
        L.thallo_hip_march_debug_set(0, int(os.environ["MB_ROWS"]))
    NAMES = {"tile": ["entry", "loads issued", "scalars known", "tile published", "barrier 1", "stencil done", "tail done"],
             "march": ["entry", "lut barrier", "first trip (loads issued)", "scalars known", "second trip done", "rows done", "tail done"]}
    for kind in ("tile", "march"):
        for modes in ((2,), (4,)):
            stamps.zero_()
            timeit(kind, modes=modes)
            torch.cuda.synchronize()
            st = stamps.cpu().numpy().reshape(-1, 8)
            st = st[(st[:, 0] > 0) & (st[:, 6] > 0)].astype(np.float64)
            for k in range(1, 7):                     # waves without rows skip the loop stamps
                st[:, k] = np.where(st[:, k] > 0, st[:, k], st[:, k - 1])
            t0 = st[:, 0].min()
            out = {"kind": kind, "modes": modes, "stamped": int(st.shape[0]), "us_per_launch": round(timeit(kind, modes=modes), 2)}
            for k, nm in enumerate(NAMES[kind]):
                v = (st[:, k] - t0) * 0.01
                out[nm] = [round(float(np.min(v)), 2), round(float(np.median(v)), 2), round(float(np.max(v)), 2)]
            if kind == "tile":
                out["clock64_ticks_per_us"] = round(float(np.median(st[:, 7] / np.maximum((st[:, 6] - st[:, 0]) * 0.01, 1e-3))), 1)
            d = (st[:, 1:7] - st[:, 0:6]) * 0.01
            out["phase_median_us"] = [round(float(x), 2) for x in np.median(d, 0)]
            print(json.dumps(out))
    sys.exit(0)

if os.environ.get("MB_MODE") == "updown":         # sweep build: every other launch marches its segments bottom-up (reads first what the previous launch wrote last)
    out = {"W": W, "H": H}
    for mode, (ri, Ai, pi, di) in {2: (it0["r"], it0["A"], it0["p"], delta0), 4: (it1["r"], it1["A"], it1["p"], delta0), 0: (it0["r"], it0["A"], it0["p"], delta0)}.items():
        cfg(2, 5, 2, 0); a = run("march", mode, ri, Ai, pi, di)
        cfg(2, 5, 2, 4); b = run("march", mode, ri, Ai, pi, di)
        out[f"up_vs_down_mode{mode}"] = {k: int((a[k][:n].view(torch.int32) != b[k][:n].view(torch.int32)).sum().item()) for k in ("r", "A", "p", "d")}
        out[f"up_vs_down_mode{mode}"]["aD_rel"] = abs(a["aD"] - b["aD"]) / abs(a["aD"])
    for rep in range(2):
        for nt in (5, 0, 1, 4, 16):
            cfg(2, nt, 2, 0)
            out[f"nt{nt}_down_only_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
            alt = lambda k: L.thallo_hip_march_debug_set(4, 4 if (k & 1) else 0)
            out[f"nt{nt}_alternating_us_{rep}"] = [round(timeit("march", hook=alt), 2), round(timeit("march", modes=(2,), hook=alt), 2), round(timeit("march", modes=(4,), hook=alt), 2)]
            out[f"nt{nt}_alternating_by_pairs_us_{rep}"] = [round(timeit("march", hook=lambda k: L.thallo_hip_march_debug_set(4, 4 if (k & 2) else 0)), 2)]
            L.thallo_hip_march_debug_set(4, 0)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_CAP"):      # force the workgroup budget the grid is sized for (256 on an MI355X = round 2's one workgroup per CU, whatever the width)
    L.thallo_hip_march_debug_set(6, int(os.environ["MB_CAP"]))

if os.environ.get("MB_MODE") == "ab":             # any build: the marching kernel at its default configuration (product vs sweep build: the sweep build's stamp checks drain the loads every row)
    out = {"W": W, "H": H, "lib": os.environ.get("THALLO_LIB", "product")}
    for rep in range(3):
        out[f"march_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "occ":            # sweep build: workgroups per CU -- the register budget the kernel is compiled for (2: 256 VGPRs ... 4: 128, prefetch depth 1) x the grid budget
    out = {"W": W, "H": H}
    for rep in range(2):
        for cap in (0, 512, 768, 1024):
            L.thallo_hip_march_debug_set(6, cap)
            for depth, occ in ((2, 2), (2, 3), (1, 3), (1, 4)):
                cfg(depth, 1, occ, 0)
                try:
                    out[f"cap{cap}_depth{depth}_occ{occ}_us_{rep}"] = [round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(0,)), 2)]
                except AssertionError:
                    out[f"cap{cap}_depth{depth}_occ{occ}_us_{rep}"] = "refused"
        L.thallo_hip_march_debug_set(6, 0)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "nohalo":         # sweep build: what more waves per CU would give if halo rows cost nothing (dbg 3 = aligned strips, no halo rows / lanes, no arithmetic) -- the
    out = {"W": W, "H": H}                        # upper bound of "waves marching in alternating directions hand their boundary rows over through LDS" 
    for rep in range(2):
        for cap in (0, 512, 1024):
            L.thallo_hip_march_debug_set(6, cap)
            for dbg in (3, 0):
                cfg(2, 5, 2, dbg)
                out[f"cap{cap}_dbg{dbg}_us_{rep}"] = [round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
        L.thallo_hip_march_debug_set(6, 0)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "depth":          # sweep build: rows of prefetch (1, 2, 3) at cache policy 1
    out = {"W": W, "H": H}
    for rep in range(2):
        for depth in (1, 2, 3):
            cfg(depth, 1, 2, 0)
            out[f"depth{depth}_us_{rep}"] = [round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "xcd":            # sweep build: with (map 0) and without (map 3) the XCD-aware placement of the workgroups
    out = {"W": W, "H": H}
    for rep in range(2):
        for mp in (0, 3):
            for dbg in (0, 3):
                cfg(2, 5, 2, dbg, 0, mp)
                out[f"map{mp}_dbg{dbg}_us_{rep}"] = [round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "batched":        # sweep build: dbg 7 = the three rows of a trip taken together, 24 refills back to back, then arithmetic and 18 stores (mode 2 only: no delta variant)
    out = {"W": W, "H": H}
    a = run("march", 2, it0["r"], it0["A"], it0["p"], delta0)
    cfg(2, 5, 2, 7); b = run("march", 2, it0["r"], it0["A"], it0["p"], delta0); cfg(2, 5, 2, 0)
    out["batched_vs_product_mismatches"] = {k: int((a[k][:n].view(torch.int32) != b[k][:n].view(torch.int32)).sum().item()) for k in ("r", "A", "p")}
    for rep in range(3):
        for nt in (5, 0):
            for dbg in (0, 7):
                cfg(2, nt, 2, dbg)
                out[f"nt{nt}_dbg{dbg}_us_{rep}"] = round(timeit("march", modes=(2,)), 2)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "fixed":          # sweep build: dbg 3 (the halo-free skeleton) against dbg 8 (the same without the iteration's scalars and without the reduction tail) and the streaming reference on strips
    out = {"W": W, "H": H}
    for rep in range(3):
        for dbg in (0, 3, 8):
            cfg(2, 5, 2, dbg)
            out[f"dbg{dbg}_us_{rep}"] = round(timeit("march", modes=(2,)), 2)
        out[f"stream_strips_percu1_us_{rep}"] = round(time_stream(0, 5, 1, span=-1), 2)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "policy":         # sweep build: the cache-policy masks at the current size (the product's 5 was chosen at 2048^2, where the Infinity Cache holds much of the working set)
    out = {"W": W, "H": H}
    for rep in range(2):
        for nt in (5, 0, 1, 4, 11, 17, 21, 31, 43, 63):
            cfg(2, nt, 2, 0)
            out[f"nt{nt}_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
        out[f"stream_nt5_percu4_us_{rep}"] = [round(time_stream(0, 5, 4, span=0), 2), round(time_stream(1, 5, 4, span=0), 2)]
        out[f"stream_nt11_percu4_us_{rep}"] = [round(time_stream(0, 11, 4, span=0), 2), round(time_stream(1, 11, 4, span=0), 2)]
        out[f"stream_nt0_percu4_us_{rep}"] = [round(time_stream(0, 0, 4, span=0), 2), round(time_stream(1, 0, 4, span=0), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "uncond":         # sweep build: dbg 3 (aligned strips, no halo, no arithmetic) against dbg 6 (the same with every store unconditional: the compiler's
    out = {"W": W, "H": H}                        # s_waitcnt vmcnt(N) then count the stores too -- 41 / 40 / 37 / 36 / 34 instead of 23 / 22 / 19 / 18 / 16)
    for rep in range(3):
        for nt in (5, 11):
            for dbg in (0, 3, 6):
                cfg(2, nt, 2, dbg)
                out[f"nt{nt}_dbg{dbg}_us_{rep}"] = [round(timeit("march", modes=(2,)), 2)]
        out[f"stream_strips_nt5_percu1_us_{rep}"] = round(time_stream(0, 5, 1, span=-1), 2)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "strips":         # sweep build: the streaming reference in linear chunks vs walking column strips down like the marching kernel (no halo, no window, no arithmetic)
    out = {"W": W, "H": H}
    for rep in range(3):
        for nt in (5, 0, 11):
            for per_cu in (1, 2, 4):
                out[f"linear_nt{nt}_percu{per_cu}_us_{rep}"] = [round(time_stream(0, nt, per_cu, span=0), 2), round(time_stream(1, nt, per_cu, span=0), 2)]
                out[f"strips_nt{nt}_percu{per_cu}_us_{rep}"] = [round(time_stream(0, nt, per_cu, span=-1), 2), round(time_stream(1, nt, per_cu, span=-1), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "pages":          # sweep build: the 4 waves of a workgroup stacked (0), side by side (1), side by side with a barrier per three rows (2)
    out = {"W": W, "H": H}
    for rep in range(2):
        for nt in (5, 0):
            for mp in (0, 1, 2):
                for rows in (0, 72, 144):
                    cfg(2, nt, 2, 0, rows, mp)
                    try:
                        out[f"nt{nt}_map{mp}_rows{rows}_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
                    except AssertionError:
                        out[f"nt{nt}_map{mp}_rows{rows}_us_{rep}"] = "refused"
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "layout":         # sweep build: streaming reference with the solver vectors as planes (Offset | Angle) vs as one 6-floats-per-pixel-pair stream each
    out = {"W": W, "H": H}
    for rep in range(3):
        for nt in (0, 5, 11):
            out[f"planes_nt{nt}_us_{rep}"] = [round(time_stream(0, nt, 2, span=0), 2), round(time_stream(1, nt, 2, span=0), 2)]
            out[f"fused_nt{nt}_us_{rep}"] = [round(time_stream(0, 100 + nt, 2, span=0), 2), round(time_stream(1, 100 + nt, 2, span=0), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "dsums":          # sweep build: what the three double sums cost (dbg 2 = the kernel without them; cache policy 1)
    out = {"W": W, "H": H}
    for rep in range(3):
        for dbg in (0, 2):
            cfg(2, 1, 2, dbg)
            out[f"nt1_dbg{dbg}_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "dbg":            # sweep build: what the halo and the arithmetic cost (dbg 1 = no stencil arithmetic, 2 = no double sums, 3 = aligned strips without halo rows / lanes)
    out = {"W": W, "H": H}
    for rep in range(2):
        for nt, dbg in ((5, 0), (5, 1), (5, 3), (11, 0), (11, 1), (11, 3), (0, 0), (0, 3), (1, 0), (1, 3)):
            cfg(2, nt, 2, dbg)
            out[f"nt{nt}_dbg{dbg}_us_{rep}"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
        out[f"stream_nt0_us_{rep}"] = [round(time_stream(0, 0, 2, span=0), 2), round(time_stream(1, 0, 2, span=0), 2)]
        out[f"stream_nt11_us_{rep}"] = [round(time_stream(0, 11, 2, span=0), 2), round(time_stream(1, 11, 2, span=0), 2)]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "order":          # sweep build: traversal order x cache policy (workgroup shape map 0 / 1, streaming reference span 0 / 1 / 8)
    out = {"W": W, "H": H}
    for nt in (5, 11):
        for dbg in (0, 3):
            for mp in (0, 1):
                cfg(2, nt, 2, dbg, 0, mp)
                out[f"march_nt{nt}_dbg{dbg}_map{mp}_us"] = [round(timeit("march", modes=(2,)), 2), round(timeit("march", modes=(4,)), 2)]
    for nt in (0, 11):
        for per_cu in (1, 2, 4):
            for span in (0, 1, 8):
                out[f"stream_nt{nt}_percu{per_cu}_span{span}_us"] = round(time_stream(0, nt, per_cu, span=span), 2)
    for nt in (0, 11):                             # does the non-temporal gain need the previous launch's writes?  (pingpong off: a -> b every time)
        out[f"stream_nt{nt}_percu2_nopingpong_us"] = round(time_stream(0, nt, 2, pingpong=False, span=0), 2)
        out[f"march_nt{nt if nt else 5}_dbg3_nopingpong_us"] = (cfg(2, nt if nt else 5, 2, 3), round(timeit("march", modes=(2,), pingpong=False), 2))[1]
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "reverse":        # sweep build: does reading what the previous launch touched LAST first let the Infinity Cache serve it?  (every other launch reversed)
    out = {"W": W, "H": H}
    for per_cu in (2, 4):
        for span in (0, 1, 8):
            for nt in (0, 1, 11, 43, 63):
                for wd in (0, 1):
                    out[f"percu{per_cu}_span{span}_nt{nt}_delta{wd}_fwd_us"] = round(time_stream(wd, nt, per_cu, span=span), 2)
                    out[f"percu{per_cu}_span{span}_nt{nt}_delta{wd}_alt_us"] = round(time_stream(wd, nt, per_cu, span=span, alternate=True), 2)
    print(json.dumps(out)); sys.exit(0)

if os.environ.get("MB_MODE") == "pmcsmall":      # under rocprofv3 --pmc (tools/small_pmc.sh): the two kernels and the streaming reference, one byte mix
    timeit("tile", modes=(2,)); timeit("march", modes=(2,)); time_stream(0, 0, 1)
    sys.exit(0)

if os.environ.get("MB_MODE") == "floor":         # what a launch costs at this size: empty-ish launches, the streaming reference, the two real kernels
    def chain(fn, reps=200):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps * 1e3, 2)
    one_word = torch.zeros(64, device="cuda"); plane = torch.zeros(3 * N, device="cuda"); plane2 = torch.zeros(3 * N, device="cuda")
    out = {"W": W, "H": H, "tiny_add_us": chain(lambda: one_word.add_(1.0)), "plane_add_us": chain(lambda: plane.add_(1.0)),
           "plane_copy_us": chain(lambda: plane2.copy_(plane))}
    for per_cu in (1, 2, 4):
        for span in (0, 1):
            for nt in (0, 11):
                out[f"stream_percu{per_cu}_span{span}_nt{nt}_us"] = round(time_stream(0, nt, per_cu, span=span), 2)
    out["tile_us"] = [round(timeit("tile"), 2), round(timeit("tile", modes=(2,)), 2)]
    out["march_us"] = [round(timeit("march"), 2), round(timeit("march", modes=(2,)), 2)]
    print(json.dumps(out)); sys.exit(0)

if __name__ == "pmc":
    cfg(2, 5, 2, 0); timeit("march", modes=(2,)); timeit("march", modes=(4,))
    cfg(2, 11, 2, 0); timeit("march", modes=(2,)); timeit("march", modes=(4,))      # the streaming reference's best policy (loads non-temporal, stores default) on the marching kernel
    cfg(2, 5, 2, 3); timeit("march", modes=(2,))
    time_stream(0, 11, 2, span=0); time_stream(0, 0, 2, span=0); time_stream(1, 11, 2, span=0)
    timeit("tile", modes=(2,))
    sys.exit(0)

tm = {}
for rep in range(2):
    tm[f"tile_us_{rep}"] = round(timeit("tile"), 2)
    for rows in (0, 27, 36, 48):
        for nt in (5, 1, 0, 11, 9, 3, 33, 43, 17, 21):
            cfg(2, nt, 2, 0, rows)
            a, m2, m4 = timeit("march"), timeit("march", modes=(2,)), timeit("march", modes=(4,))
            tm[f"march_nt{nt}_rows{rows}_us_{rep}"] = [round(a, 2), round(m2, 2), round(m4, 2)]
        print(json.dumps(tm)); sys.stdout.flush()
res["timing"] = tm
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"march_probe4_{W}x{H}.json"), "w"), indent=1)
