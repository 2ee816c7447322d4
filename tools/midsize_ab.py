"""[RESEARCH build: make -C thallo_amd/csrc VARIANT=research (stamps: EXTRA with -DTHALLO_RESEARCH), run with THALLO_LIB=tools/ab/libThallo_research.so -- the loop this probes is not in the product library since round 6]
GPU probe: mid-size images (too large for the resident kernel's registers, smaller than the benchmark) through Thallo_ProblemStep: us per PCG iteration with a marching
launch per iteration (default) and with the persistent marching loop (THALLO_AB=persist=1).  python tools/midsize_ab.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
def run(W, H, ab, steps=10, L=100):
    if ab: os.environ["THALLO_AB"] = ab
    else: os.environ.pop("THALLO_AB", None)
    p = syn.image_warping(W, H)
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=steps + 2, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(2): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = sorted(s.kernel_stats()); s.close()
    return round(dt / steps / L * 1e6, 2), [n for n in names if n.startswith("PCG")]
out = {}
for (W, H) in ((2048, 1024), (1536, 1536), (2048, 1536), (1280, 1024)):
    out[f"{W}x{H}"] = {"launch_per_iteration": run(W, H, None), "persistent": run(W, H, "persist=1"), "rows_per_wave": thallo_amd.lib().thallo_hip_iw_march_rows(W, H)}
print(json.dumps(out, indent=1))
