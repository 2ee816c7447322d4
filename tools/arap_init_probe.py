import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
p = syn.arap_mesh(320, 320); dims = (p[2].shape[0], p[6].shape[0])
out = {}
for reorder in (1, 0):
    thallo_amd.lib().thallo_hip_arap_debug_reorder(reorder)
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=2, lIterations=10)
    prm = s.make_params(dev)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.init(prm); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    out["init_ms_reorder_%d" % reorder] = ts
    s.close()
thallo_amd.lib().thallo_hip_arap_debug_reorder(1)
print(json.dumps(out))
