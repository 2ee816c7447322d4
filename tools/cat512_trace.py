"""A/B probe: alpha/beta traces of one GN step (100 PCG iterations) on the cat512 instance -- marching kernel vs tile kernel vs the CPU port."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, thallo_amd
from thallo_amd import api, formats as F
from oracle import oracle as orc
g = 'tests/golden'
mask = F.read_png(os.path.join(g, "cat512_mask.png"))[:, :, 0].astype(np.float32)
H, W = mask.shape
cons = F.add_border_constraints(F.read_constraints(os.path.join(g, "cat512.constraints")), W, H)
yy, xx = np.mgrid[0:H, 0:W]
ur = np.stack([xx, yy], axis=2).astype(np.float32)
wf, wr = float(np.sqrt(np.float32(100.0))), float(np.sqrt(np.float32(0.01)))
c_img = F.constraint_image(cons, mask, np.float32(1) / np.float32(19))
P = lambda: [ur.copy(), np.zeros((H, W), np.float32), ur.copy(), c_img.copy(), mask.copy(), wf, wr]
ref = orc.cpu_port_image_warping(W, H, P(), 1, 100, want_trace=True)
out = {}
for tag, env in (("march", {}), ("tile", {"THALLO_MARCH": "0"}), ("two", {"THALLO_AB": "one_kernel=0"})):
    for k in ("THALLO_MARCH", "THALLO_AB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in P()]
    s = api.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"))
    final, costs = s.solve(dev, profiled=True, nIterations=1, lIterations=100)
    out[tag] = (np.array(s.alpha_beta_trace()), costs)
    s.close()
r = ref["trace"]
np.set_printoptions(linewidth=200, precision=2)
for tag in out:
    t = out[tag][0]
    print(tag, "costs", out[tag][1], "port costs", ref["costs"])
    print(" rel alpha err vs port, every 5th iteration:", (np.abs(t[:, 0] - r[:, 0]) / np.abs(r[:, 0]))[::5])
print("march vs tile rel alpha diff:", (np.abs(out["march"][0][:, 0] - out["tile"][0][:, 0]) / np.abs(out["tile"][0][:, 0]))[::5])
