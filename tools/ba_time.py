import json, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/thallo_amd") else os.getcwd())
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
p = syn.bundle_adjustment(); dims = (1723, 156502, 678718); L = 150
def run(lm, steps=4):
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("bundle_adjustment"), timing_level=0)
    if lm: s.enable_lm()
    s.set_solver_parameters(nIterations=steps + 1, lIterations=L, q_tolerance=0.0)
    prm = s.make_params(dev); s.init(prm)
    s.step(prm); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    costs = [s.current_cost()]
    while s.step(prm): n += 1; costs.append(s.current_cost())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s.close()
    return {"lm": lm, "us_per_pcg_iter": round(dt / max(n, 1) / L * 1e6, 2), "costs": costs}
if os.environ.get("BA_TIME_ONLY") == "gn":
    print(json.dumps([run(False), run(False)])); sys.exit(0)
if os.environ.get("BA_TIME_ONLY") == "lm":      # under rocprofv3: the LM loop's kernels only
    print(json.dumps([run(True), run(True)])); sys.exit(0)
out = [run(False), run(True), run(False), run(True)]
os.environ["THALLO_AB"] = "lm_fold_p=0"           # the reference-shaped LM loop (A/B)
out += [dict(run(True), lm_fold_p=0), dict(run(True), lm_fold_p=0)]
print(json.dumps(out))
