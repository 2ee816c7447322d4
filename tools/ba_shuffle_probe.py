import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
p0 = syn.bundle_adjustment(); dims = (1723, 156502, 678718); L = 150
def run(p, steps=3):
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("bundle_adjustment"), timing_level=0)
    s.set_solver_parameters(nIterations=steps + 1, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    s.step(prm); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while s.step(prm): n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = s.current_cost(); s.close()
    return round(dt / max(n, 1) / L * 1e6, 2), c
print("banded (as generated):", run(p0), run(p0))
rng = np.random.default_rng(1)
perm = rng.permutation(dims[1])            # new id of old point j = perm[j]
inv = np.argsort(perm)
p1 = [p0[0], np.ascontiguousarray(p0[1][inv]), p0[2], p0[3], np.ascontiguousarray(perm[p0[4]].astype(np.int32))]
print("points shuffled:", run(p1), run(p1))
# observations shuffled too
po = rng.permutation(dims[2])
p2 = [p1[0], p1[1], np.ascontiguousarray(p1[2][po]), np.ascontiguousarray(p1[3][po]), np.ascontiguousarray(p1[4][po])]
print("points and observations shuffled:", run(p2), run(p2))
