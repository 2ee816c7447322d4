"""arap_resident_probe.py -- GPU probe: ARAP 102,400 vertices (BASELINE config 2's size), us per PCG iteration through Thallo_ProblemStep with the resident PCG loop
(one launch per GN step) against PCGUpdate + applyJTJ per iteration (THALLO_RESIDENT=0), alternating."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 320; nv = int(sys.argv[2]) if len(sys.argv) > 2 else nu
p = syn.arap_mesh(nu, nv)
dims = (p[2].shape[0], p[6].shape[0])
L = 100
def run(resident, steps=20, warm=3, reorder=1):
    os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
    thallo_amd.lib().thallo_hip_arap_debug_reorder(reorder)
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(warm): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = sorted(s.kernel_stats()); c = s.current_cost(); s.close()
    thallo_amd.lib().thallo_hip_arap_debug_reorder(1)
    return {"resident": resident, "us_per_pcg_iter": round(dt / (steps * L) * 1e6, 2), "ms_per_gn_step": round(dt / steps * 1e3, 3), "cost": c, "kernels": names}
out = {"vertices": dims[0], "edges": dims[1], "runs": [run(True), run(False), run(True), run(False)], "callers_numbering": [run(True, reorder=0), run(False, reorder=0)]}
print(json.dumps(out))
if os.environ.get("ARAP_STAMPS"):          # needs a library built with `make VARIANT=arapstamps EXTRA=-DARAP_STAMPS` (THALLO_LIB points at tools/ab/libThallo_arapstamps.so)
    import ctypes as C
    os.environ["THALLO_RESIDENT"] = "1"
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=3, lIterations=L)
    prm = s.make_params(dev); s.init(prm); s.step(prm); s.step(prm); torch.cuda.synchronize()
    Lb = thallo_amd.lib()
    Lb.thallo_hip_arap_debug_last_xbuf.restype = C.c_void_p
    t = (C.c_uint * 64)()
    C.CDLL("libamdhip64.so").hipMemcpy(t, C.c_void_p(Lb.thallo_hip_arap_debug_last_xbuf()), 256, 2)
    for name, o in (("wg0", 16), ("wg200", 28)):
        a = [t[o + i] for i in range(8)]
        print(name, "stamps (10 ns units, deltas): start>apply+store>record>polled>barrier>sums>update+ghosts>barrier", [a[i + 1] - a[i] for i in range(7)])
    nb = Lb.thallo_hip_arap_resident_bytes
    nb.restype = C.c_long; nb.argtypes = [C.c_int]
    total = nb(dims[0])
    st = (C.c_uint * 2048)()
    C.CDLL("libamdhip64.so").hipMemcpy(st, C.c_void_p(Lb.thallo_hip_arap_debug_last_xbuf() + total - 8192), 8192, 2)
    nwg = (dims[0] + 255) // 256
    a = np.array(st[:], dtype=np.int64).reshape(4, 512)[:, :nwg]
    t0 = a[0].min()
    for name, row in zip(("iteration start", "record stored", "totals in", "update done"), a):
        r = (row - t0) / 100.0
        print(f"{name:16s} us after the first workgroup's start: min {r.min():6.2f}  median {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  max {r.max():6.2f}  (last: wg {int(r.argmax())})")
    late = np.argsort(a[1])[-8:]
    print("the 8 latest records: workgroups", late.tolist(), "their iteration start - t0:", ((a[0][late] - t0) / 100.0).round(2).tolist(), "apply+record us:", ((a[1][late] - a[0][late]) / 100.0).round(2).tolist())
