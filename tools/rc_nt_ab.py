"""GPU probe (sweep build: make -C thallo_amd/csrc VARIANT=sweep, THALLO_LIB=tools/ab/libThallo_sweep.so): the cache-policy mask of the no-A p-plane marching kernel under
the ring schedule (bits: 1 delta, 2 r loads, 4 r stores, 8 p loads, 16 p stores, 32 cs / flags; product 5): ms per GN step at 2048^2, one process."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("THALLO_LIB", os.path.join(ROOT, "tools", "ab", "libThallo_sweep.so"))
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
L = thallo_amd.lib()
W = H = 2048
p = syn.image_warping(W, H)
def run(nt, steps=12):
    L.thallo_hip_march_rc_debug_set(2, nt)
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=steps + 3, lIterations=100)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(3): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = s.current_cost(); s.close()
    return {"nt_mask": nt, "ms_per_gn_step": round(dt / steps * 1e3, 4), "cost": c}
print(json.dumps([run(m) for m in (5, 21, 4, 13, 7, 63, 0, 5, 21)], indent=0))
