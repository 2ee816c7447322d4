"""ARAP applyJTJ: the unrolled ELL form (energy_graph.hip k_arap_apply_ell) against the loop form -- same solve twice, bitwise comparison of the
cost trajectory and the unknowns, time per PCG iteration -- and the form that recomputes G_e (k_arap_apply_rc, round 3, the default) against both (same formulas,
equal to rounding).  python tools/arap_probe.py [nx ny]"""
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


def solve(p, unrolled, nit=5, lit=100, recompute=0):
    L.thallo_hip_arap_debug_set(0, unrolled); L.thallo_hip_arap_debug_set(1, recompute)
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
    L.thallo_hip_arap_debug_set(0, 1); L.thallo_hip_arap_debug_set(1, 1)
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
    cc, uc, tc = solve(p, 1, recompute=1)
    rel = max(abs(a - b) / abs(b) for a, b in zip(cc, cb))
    import json
    print(json.dumps({"vertices": int(p[2].shape[0]), "edges": int(p[6].shape[0]), "loop_us": round(ta, 2), "ell_stored_G_us": round(tb, 2), "ell_recompute_G_us": round(tc, 2),
                      "stored_forms_bitwise_equal": bool(same), "recompute_vs_stored_max_rel_cost_diff": rel}))
    return 0 if same and rel < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
