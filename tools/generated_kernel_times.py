"""Per-kernel times of the GENERATED image_warping kernels at the benchmark size (THALLO_FRONTEND=generate), from the library's own hipEvent timer
(timingLevel 2: every launch).  python tools/generated_kernel_times.py [size]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["THALLO_FRONTEND"] = "generate"
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = syn.image_warping(W, H)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=2)
print(s.energy_name, "|", s.schedule_name)
s.solve(dev, nIterations=2, lIterations=20)
ks = s.kernel_stats()
print(json.dumps({k: {"launches": v["launches"], "mean_us": round(1e3 * v["total_ms"] / max(1, v["samples"]), 1)} for k, v in ks.items()}, indent=1))
