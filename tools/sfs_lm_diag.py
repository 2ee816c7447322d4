"""GPU probe: shape_from_shading 2048^2 LM, device vs row oracle per LM step: costs, PCG iterations run (the zeta early exit), with the default q_tolerance and with
the early exit disabled (q_tolerance = 0)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import api, synthetic as syn
from oracle import oracle as orc

W = H = int(os.environ.get("PW", "2048")); NS = int(os.environ.get("NS", "8"))
p = syn.shape_from_shading(W, H)
cp = lambda q: [x.copy() if isinstance(x, np.ndarray) else x for x in q]
orc.set_threads(max(1, min(64, os.cpu_count() or 1)))
for qtol in (1e-4, 0.0):
    co, _ = orc.Problem(orc.SFS, (W, H), cp(p)).solve(nIterations=NS, lIterations=10, use_lm=1, q_tolerance=qtol)
    want = orc.last_pcg_counts()
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = api.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), solverkind="levenberg_marquardt")
    s.enable_lm(); s.set_solver_parameters(nIterations=NS, lIterations=10, q_tolerance=qtol)
    params = s.make_params(dev); s.init(params)
    costs, iters = [s.current_cost()], []
    while s.step(params):
        costs.append(s.current_cost()); iters.append(len(s.alpha_beta_trace()))
    s.close()
    m = min(len(costs), len(co))
    print(json.dumps({"q_tolerance": qtol, "oracle_pcg_iters": want, "device_pcg_iters": iters,
                      "rel_err": [float(abs(a - b) / abs(b)) for a, b in zip(costs[:m], co[:m])]}))
