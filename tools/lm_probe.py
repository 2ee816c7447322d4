"""GPU probe: per-kernel times of the Levenberg-Marquardt branch on shape_from_shading 2048^2 (BASELINE config 4) and bundle adjustment
(ladybug-1723 shape, config 5).  python tools/lm_probe.py [sfs|ba]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn

which = sys.argv[1] if len(sys.argv) > 1 else "sfs"
if which == "sfs":
    W = H = int(os.environ.get("PW", "2048")); p = syn.shape_from_shading(W, H); dims, name, nit, lit = (W, H), "shape_from_shading", 6, 10
else:
    p = syn.bundle_adjustment(); dims, name, nit, lit = (1723, 156502, 678718), "bundle_adjustment", 5, 150
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file(name), timing_level=0)
s.enable_lm()
s.set_solver_parameters(nIterations=nit, lIterations=lit, q_tolerance=0.0)
params = s.make_params(dev)
s.init(params); s.step(params)                      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while s.step(params):
    n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"config": f"{name} LM {nit}x{lit}", "gn_steps_timed": n, "ms_per_gn_iter": dt / max(n, 1) * 1e3, "us_per_pcg_iter": dt / max(n, 1) / lit * 1e6}
s2 = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file(name), timing_level=0)
s2.enable_lm(); s2.set_solver_parameters(nIterations=3, lIterations=lit, q_tolerance=0.0); s2.set_kernel_sampling(1)
dev2 = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
pr = s2.make_params(dev2); s2.init(pr)
while s2.step(pr):
    pass
out["kernel_mean_us"] = {k: round(v["mean_ms"] * 1e3, 2) for k, v in s2.kernel_stats().items() if v["mean_ms"]}
print(json.dumps(out, indent=1))
