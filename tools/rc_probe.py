"""rc_probe.py -- A/B of the marching PCG iteration without the A p plane (energy_image_warping_march_rc.hip) against the stored-plane kernel, through
Thallo_ProblemStep on one box: ms per GN step and us per PCG iteration (the library's `Linear Solve` event pair / lIterations), alternating runs.

  THALLO_LIB=tools/ab/libThallo_sweep.so python tools/rc_probe.py [size] [steps]        env RC_CFGS="depth:occ,..." RC_NT="mask,..." RC_ROWS="rows per segment,..."
(the knobs need the sweep build: make -C thallo_amd/csrc VARIANT=sweep; the product has one configuration)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import thallo_amd
from thallo_amd import api, synthetic as syn

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Lit = 100
L = thallo_amd.lib()
p = syn.image_warping(W, H)


def run(form, depth=2, occ=2, nt=5, warm=3):
    os.environ["THALLO_MARCH"] = form
    os.environ["THALLO_RESIDENT"] = "0"
    L.thallo_hip_march_rc_debug_set(0, depth); L.thallo_hip_march_rc_debug_set(1, occ); L.thallo_hip_march_rc_debug_set(2, nt)
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = api.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=1)
    s.set_solver_parameters(nIterations=warm + steps, lIterations=Lit)
    params = s.make_params(dev)
    s.init(params)
    for _ in range(warm):
        s.step(params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        s.step(params)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    cost = s.current_cost()
    s.close()
    return {"ms_per_gn_step": round(ms, 4), "us_per_pcg_iter": round(1000 * ms / Lit, 2), "cost": cost}


cfgs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("RC_CFGS", "2:2,1:2,2:1,4:1").split(",")]
nts = [int(v) for v in os.environ.get("RC_NT", "").split(",") if v]           # cache-policy masks (sweep build)
rows = [int(v) for v in os.environ.get("RC_ROWS", "0").split(",")]          # rows per wave segment (0 = automatic: one workgroup per CU)
out = {"W": W, "H": H, "steps": steps, "lib": os.environ.get("THALLO_LIB", "product"), "runs": []}
for rep in range(2):
    for R in rows:
        L.thallo_hip_march_debug_set(0, R)
        out["runs"].append({"form": "stored", "rows": R, **run("3")})
        for d, o in cfgs:
            out["runs"].append({"form": "rc", "rows": R, "depth": d, "occ": o, **run("1", d, o)})
        for nt in nts:
            out["runs"].append({"form": "rc", "rows": R, "nt": nt, **run("1", int(os.environ.get("RC_NT_DEPTH", "2")), 2, nt)})
print(json.dumps(out))
