"""Generated graph energies at the benchmark sizes (THALLO_FRONTEND=generate): the unknown-wise lowering through the Sparse maps (per-owner instance lists, useAutoscheduler = 1)
against the residual-wise atomic scatter (useAutoscheduler = 0).  Per-kernel means from the library's hipEvent timer (timingLevel 2) and the wall time per PCG iteration
without it.  python tools/generated_graph_times.py"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["THALLO_FRONTEND"] = "generate"
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn


def run(name, dims, p, auto, L):
    out = {}
    for timing in (2, 0):
        dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
        s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file(name), timing_level=timing, autoschedule=auto)
        out["schedule"] = s.schedule_name
        s.set_solver_parameters(nIterations=4, lIterations=L)
        prm = s.make_params(dev); s.init(prm)
        s.step(prm); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        while s.step(prm): n += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if timing:
            ks = s.kernel_stats()
            out["kernel_mean_us"] = {k: round(1e3 * v["total_ms"] / max(1, v["samples"]), 1) for k, v in ks.items() if v["launches"]}
        else:
            out["us_per_pcg_iter"] = round(dt / max(n, 1) / L * 1e6, 1); out["cost"] = s.current_cost()
        s.close()
    return out


res = {}
p = syn.arap_mesh(320, 320)
dims = (p[2].shape[0], p[6].shape[0])
for auto in (1, 0): res["arap %d v / %d e, autoschedule=%d" % (dims[0], dims[1], auto)] = run("arap_mesh_deformation", dims, p, auto, 50)
p = syn.bundle_adjustment()
dims = (1723, 156502, 678718)
for auto in (1, 0): res["bundle_adjustment ladybug-1723 shape, autoschedule=%d" % auto] = run("bundle_adjustment", dims, p, auto, 50)
sys.stdout.flush()
print("JSON " + json.dumps(res))        # (the library's own "Initial cost" lines share stdout: the result is the line that starts with JSON)
