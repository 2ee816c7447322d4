"""shape_from_shading 2048^2, two Gauss-Newton steps of ten PCG iterations -- the launches tools/sfs_pmc.sh counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
W = H = 2048
p = syn.shape_from_shading(W, H)
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), timing_level=0)
s.solve(dev, nIterations=2, lIterations=10)
torch.cuda.synchronize()
