"""sfs_lm_budget.py -- GPU probe: shape_from_shading 2048^2, LM, the reference budget 60 x 10, against tests/golden/oracle_trajectories.json (per-step relative cost error)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import thallo_amd
from thallo_amd import api, synthetic as syn
from helpers import to_device
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_trajectories.json")))
W = H = 2048
p = syn.shape_from_shading(W, H)
dev = to_device(p)
s = api.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), solverkind="levenberg_marquardt")
s.enable_lm()
final, costs = s.solve(dev, profiled=True, nIterations=60, lIterations=10)
co = np.array(fx["sfs2048_lm_60x10"]["double"]); cf = np.array(fx["sfs2048_lm_float_order_12x10"]["float_order"])
costs = np.array(costs)
m = min(len(costs), len(co))
err = np.abs(costs[:m] - co[:m]) / np.abs(co[:m])
print(json.dumps({"n": [len(costs), len(co)], "err": [float(f"{e:.3g}") for e in err], "gpu_final": float(costs[-1]), "oracle_final": float(co[-1]),
                  "spread12": [float(f"{e:.3g}") for e in np.abs(cf - co[:13]) / np.abs(co[:13])], "pcg_counts_oracle": fx["sfs2048_lm_60x10"]["pcg_counts"]}))
