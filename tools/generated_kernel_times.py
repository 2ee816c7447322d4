"""Per-kernel times of the GENERATED kernels of a stencil energy at the benchmark size (THALLO_FRONTEND=generate), from the library's own hipEvent timer
(timingLevel 2: every launch).  python tools/generated_kernel_times.py [image_warping|shape_from_shading] [size]
A/B: THALLO_AB=frontend_preload=0 (every residual instance of the merged gather kernel loads for itself: round 4's lowering)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["THALLO_FRONTEND"] = "generate"
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "image_warping"
W = H = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2048
p = getattr(syn, name)(W, H)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x if isinstance(x, (bytes, bytearray)) else float(x) for x in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file(name), timing_level=2)
print(s.energy_name, "|", s.schedule_name)
s.solve(dev, nIterations=2, lIterations=20)
ks = s.kernel_stats()
print(json.dumps({k: {"launches": v["launches"], "mean_us": round(1e3 * v["total_ms"] / max(1, v["samples"]), 1)} for k, v in ks.items()}, indent=1))
print("cost", s.current_cost())
