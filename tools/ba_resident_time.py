"""[RESEARCH build: make -C thallo_amd/csrc VARIANT=research (stamps: EXTRA with -DTHALLO_RESEARCH), run with THALLO_LIB=tools/ab/libThallo_research.so -- the loop this probes is not in the product library since round 6]
ba_resident_time.py -- GPU probe: bundle adjustment (ladybug-1723 shape) GN, us per PCG iteration through Thallo_ProblemStep, the resident PCG loop (one launch per GN
step) against three launches per iteration (THALLO_RESIDENT=0), alternating in one process.  python tools/ba_resident_time.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
p = syn.bundle_adjustment(); dims = (1723, 156502, 678718); L = 150
def run(res, steps=4):
    os.environ["THALLO_RESIDENT"] = res
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("bundle_adjustment"), timing_level=0)
    s.set_solver_parameters(nIterations=steps + 1, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    s.step(prm); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    costs = [s.current_cost()]
    while s.step(prm): n += 1; costs.append(s.current_cost())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = sorted(s.kernel_stats())
    s.close()
    return {"resident": res, "us_per_pcg_iter": round(dt / max(n, 1) / L * 1e6, 2), "final_cost": costs[-1], "kernels": names, "error": thallo_amd.last_error()}
out = [run("0"), run("2"), run("0"), run("2")]
for wg in (256, 512, 768):
    thallo_amd.lib().thallo_hip_ba_resident_debug_set(0, wg)
    r = run("2"); r["workgroups"] = wg; out.append(r)
thallo_amd.lib().thallo_hip_ba_resident_debug_set(0, 0)
print(json.dumps([{k: v for k, v in r.items() if k != "kernels"} for r in out], indent=1))
