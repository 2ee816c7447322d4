"""GPU probe (sweep build: THALLO_LIB=tools/ab/libThallo_sweep.so): prefetch depth 2 against 4 of the marching PCG iteration at sizes with few rows per wave
(launch per iteration, resident loop off): us per PCG iteration.  Decides MARCH_RC_DEEP_ROWS."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("THALLO_LIB", os.path.join(ROOT, "tools", "ab", "libThallo_sweep.so"))
os.environ["THALLO_RESIDENT"] = "0"
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
L = thallo_amd.lib()
def run(w, h, depth, Lit=100, steps=10):
    L.thallo_hip_march_rc_debug_set(0, depth)
    p = syn.image_warping(w, h)
    d = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((w, h), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=Lit)
    pr = s.make_params(d); s.init(pr)
    for _ in range(3): s.step(pr)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(steps): s.step(pr)
    torch.cuda.synchronize(); us = (time.perf_counter() - t1) / (steps * Lit) * 1e6
    s.close(); return round(us, 2)
out = {}
for (w, h) in ((2048, 2048), (2048, 1536), (2048, 1280), (2048, 1024), (1280, 1024), (1024, 768), (2048, 512), (2048, 256), (512, 512)):
    out[f"{w}x{h}"] = {"depth2": [run(w, h, 2), run(w, h, 2)], "depth4": [run(w, h, 4), run(w, h, 4)]}
    print(json.dumps(out), flush=True)
