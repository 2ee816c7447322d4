"""shape_from_shading 2048^2 per PCG iteration through Thallo_ProblemStep, no kernel timers: Gauss-Newton and LM (10 PCG iterations per step, the reference's budget), with the
finish of iteration k-1 deferred into the launch of iteration k (default) and with the in-kernel finish (THALLO_AB=fin_in_kernel=1), alternating in one process.  python tools/sfs_time.py"""
import json, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/thallo_amd") else os.getcwd())
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
W = int(os.environ.get("PW", "2048")); H = int(os.environ.get("PH", str(W)))
p = syn.shape_from_shading(W, H)


def run(lm, fin, steps=12, L=10):
    if fin is None: os.environ.pop("THALLO_AB", None)
    else: os.environ["THALLO_AB"] = "fin_in_kernel=" + fin
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), timing_level=0)
    if lm: s.enable_lm()
    s.set_solver_parameters(nIterations=steps + 1, lIterations=L, q_tolerance=0.0)
    prm = s.make_params(dev); s.init(prm)
    s.step(prm); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while s.step(prm): n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = s.current_cost(); s.close()
    return {"lm": lm, "finish": "deferred" if fin is None else "in-kernel", "us_per_pcg_iter": round(dt / max(n, 1) / L * 1e6, 2), "cost": c}


out = []
for rep in range(2):
    for lm in (False, True):
        for fin in (None, "1"):
            out.append(run(lm, fin))
sys.stdout.flush()
print("JSON " + json.dumps(out))
