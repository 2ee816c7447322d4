"""GPU probe: the headline schedule (marching launch per iteration, ring of p planes) with the marching grid sized for 1 / 2 / 3 workgroups per CU (thallo_hip_march_debug_set(6, cap)):
ms per GN step at 2048^2, alternating in one process.  python tools/march_occ_ab.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
L = thallo_amd.lib()
W = H = 2048
p = syn.image_warping(W, H)
def run(cap, steps=12):
    L.thallo_hip_march_debug_set(6, cap)
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=steps + 3, lIterations=100)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(3): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s.close(); L.thallo_hip_march_debug_set(6, 0)
    return {"workgroup_budget": cap or "default (1 per CU)", "rows_per_wave": L.thallo_hip_iw_march_rows(W, H) if not cap else None, "ms_per_gn_step": round(dt / steps * 1e3, 4)}
print(json.dumps([run(0), run(512), run(768), run(0), run(512)], indent=1))
