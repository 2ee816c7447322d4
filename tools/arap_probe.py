"""ARAP applyJTJ: the unrolled ELL form (energy_graph.hip k_arap_apply_ell) against the loop form -- same solve twice, bitwise comparison of the
cost trajectory and the unknowns, time per PCG iteration.  python tools/arap_probe.py [nx ny]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import thallo_amd
from thallo_amd import api, synthetic as syn

L = api.lib()
L.thallo_hip_arap_debug_set.argtypes = [C.c_int, C.c_int]; L.thallo_hip_arap_debug_set.restype = None


def solve(p, unrolled, nit=5, lit=100):
    L.thallo_hip_arap_debug_set(0, unrolled)
    dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((p[2].shape[0], p[6].shape[0]), thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=nit + 1, lIterations=lit)
    prm = s.make_params(dev)
    s.init(prm); s.step(prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    costs = []
    for _ in range(nit):
        s.step(prm); costs.append(s.current_cost())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = [d.clone() for d in dev if hasattr(d, "clone")]
    s.close()
    L.thallo_hip_arap_debug_set(0, 1)
    return costs, out, dt / (nit * lit) * 1e6


def main():
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 320)
    p = syn.arap_mesh(nx, ny)
    ca, ua, ta = solve(p, 0)
    cb, ub, tb = solve(p, 1)
    same = ca == cb and all(torch.equal(a, b) for a, b in zip(ua, ub))
    print(f"ARAP {p[2].shape[0]} vertices / {p[6].shape[0]} edges: loop form {ta:.2f} us per PCG iteration (incl. cost read-backs), unrolled ELL form {tb:.2f} us; "
          f"costs and unknowns bitwise equal: {same}")
    print("costs", ca[:3], cb[:3])
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
