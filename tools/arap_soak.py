"""GPU soak: ARAP 102,400 vertices, 120 GN steps x 60 PCG iterations, resident PCG loop (plan numbering) against PCGUpdate + applyJTJ per iteration: every cost and the unknowns bit for bit."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
p = syn.arap_mesh(320, 320, n_handles=64, angle_amp=0.3); dims = (p[2].shape[0], p[6].shape[0])
out = {}
res = []
for resident in ("1", "0"):
    os.environ["THALLO_RESIDENT"] = resident
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=120, lIterations=60)
    prm = s.make_params(dev); s.init(prm)
    costs = [s.current_cost()]
    while s.step(prm): costs.append(s.current_cost())
    res.append((costs, dev[2].clone(), dev[3].clone())); s.close()
print(json.dumps({"steps": len(res[0][0]) - 1, "costs_equal": res[0][0] == res[1][0], "unknowns_equal": bool(torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])), "first": res[0][0][0], "last": res[0][0][-1], "err": thallo_amd.last_error()}))
