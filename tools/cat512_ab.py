import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, thallo_amd
from thallo_amd import api, formats as F
g='tests/golden'
mask = F.read_png(os.path.join(g, "cat512_mask.png"))[:, :, 0].astype(np.float32)
H, W = mask.shape
cons = F.add_border_constraints(F.read_constraints(os.path.join(g, "cat512.constraints")), W, H)
yy, xx = np.mgrid[0:H, 0:W]
ur = np.stack([xx, yy], axis=2).astype(np.float32)
wf, wr = float(np.sqrt(np.float32(100.0))), float(np.sqrt(np.float32(0.01)))
c_img = F.constraint_image(cons, mask, np.float32(1) / np.float32(19))
cd = np.array([6.14875977e+03,7.29465067e-01,4.31712300e-01,3.31736743e-01,2.73316592e-01,2.35340923e-01,2.06320405e-01,1.84857458e-01,1.68697983e-01])
dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else x for x in [ur, np.zeros((H, W), np.float32), ur, c_img, mask, wf, wr]]
s = api.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"))
final, costs = s.solve(dev, profiled=True, nIterations=8, lIterations=100)
print(os.environ.get("TAG"), np.array(costs), "rel", np.abs(np.array(costs)-cd)/cd)
