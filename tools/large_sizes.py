import os, sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
def run(w, h, L=100, steps=4):
    p = syn.image_warping(w, h)
    d = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((w, h), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    pr = s.make_params(d); s.init(pr)
    for _ in range(2): s.step(pr)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(steps): s.step(pr)
    torch.cuda.synchronize(); us = (time.perf_counter() - t1) / (steps * L) * 1e6
    c = s.current_cost(); s.close()
    free, tot = torch.cuda.mem_get_info()
    return {"us_per_pcg_iter": round(us, 1), "pixels_M": round(w * h / 1e6, 1), "loop_TBps_at_69.84_B_per_px": round(69.84 * w * h / us / 1e6, 2), "cost": c}
out = {}
for (w, h) in ((2048, 2048), (4096, 4096), (8192, 8192), (16384, 8192)):
    try: out[f"{w}x{h}"] = run(w, h)
    except Exception as e: out[f"{w}x{h}"] = repr(e)[:200]
    print(json.dumps(out), flush=True)
