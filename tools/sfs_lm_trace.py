"""GPU probe: one process, shape_from_shading 2048^2 LM 6 x 10 -- meant to be run under rocprofv3 --kernel-trace to see the launch sequence of an LM step (per-step launches and host gaps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
W = H = 2048
p = syn.shape_from_shading(W, H)
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), timing_level=0)
s.enable_lm()
s.set_solver_parameters(nIterations=6, lIterations=10, q_tolerance=0.0)
prm = s.make_params(dev); s.init(prm)
while s.step(prm): pass
torch.cuda.synchronize()
