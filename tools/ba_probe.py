import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import thallo_amd
from thallo_amd import api, synthetic as syn
from oracle import oracle as orc
from helpers import to_device, copy_params
for (C_,P_,O_) in [(12,60,300),(64,4000,20000)]:
  for lit in (5,10,20,40,80):
    p = syn.bundle_adjustment(C=C_, P=P_, O=O_, band=8)
    po = copy_params(p)
    co,_ = orc.Problem(orc.BUNDLE_ADJUST,(C_,P_,O_),po).solve(nIterations=3,lIterations=lit)
    cf,_ = orc.Problem(orc.BUNDLE_ADJUST,(C_,P_,O_),copy_params(p)).solve(nIterations=3,lIterations=lit, float_sums=1)
    dev = to_device(p)
    s = api.ThalloSolver((C_,P_,O_), thallo_amd.energy_file("bundle_adjustment"))
    _, costs = s.solve(dev, profiled=True, nIterations=3, lIterations=lit)
    print(C_, lit, 'gpu-vs-oracle', np.abs(np.array(costs)-co)/co, ' oracle float-vs-double sums', np.abs(cf-co)/co)
