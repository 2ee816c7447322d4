import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
def run(w, h, resident, lm=False, lit=10):
    os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
    q = syn.shape_from_shading(w, h)
    d = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in q]
    s = thallo_amd.ThalloSolver((w, h), thallo_amd.energy_file("shape_from_shading"), timing_level=0, **({"solverkind": "levenberg_marquardt"} if lm else {}))
    if lm: s.enable_lm()
    s.set_solver_parameters(nIterations=1 << 30, lIterations=lit, **({"q_tolerance": 0.0} if lm else {}))
    p = s.make_params(d); s.init(p)
    for _ in range(3): s.step(p)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): s.step(p)
    torch.cuda.synchronize(); us = (time.perf_counter() - t) / (20 * lit) * 1e6
    names = sorted(k for k, v in s.kernel_stats().items())
    s.close(); return round(us, 2), ("resident" if any("Resident" in n for n in names) else "launches")
for (w, h) in ((640, 480), (800, 600), (1024, 768), (1280, 720), (1280, 960), (1600, 900), (1920, 1080)):
    print(w, h, "GN", run(w, h, True), run(w, h, False), "LM", run(w, h, True, True), run(w, h, False, True), flush=True)
