"""Probe (1 GPU, world size 1, the library's own RCCL really issuing its all-gathers): per-PCG-iteration time of ONE rank's shape_from_shading slab (what one of 8
ranks owns at 2048^2 -> 2048 x 256 + ghost rows) through the library's flat slab path (csrc/solver_dist.cpp: dist_gn_flat / step_lm) with (a) pack + ncclAllGather +
unpack per exchange and (b) the device-side exchange (thallo_hip_dist_xrows, one launch per exchange); Gauss-Newton and Levenberg-Marquardt.  Gives the per-rank
kernel + protocol cost without xGMI latency.   PW / PH: slab size."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from thallo_amd import synthetic as syn
from thallo_amd.distributed_sfs import PlanSfsSlabSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29536")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
W = int(os.environ.get("PW", "2048")); H = int(os.environ.get("PH", "256")); L = 10; STEPS = 20
p = syn.shape_from_shading(W, H)
out = {"size": [W, H], "l_iters": L, "gn_steps": STEPS}
for lm in (False, True):
    for name, p2p in (("rccl", False), ("p2p", True)):
        s = PlanSfsSlabSolver(p, W, H, 0, 1, L, lm=lm, device_exchange=p2p, force_rccl=True)
        s.solver.set_solver_parameters(nIterations=1 << 30, lIterations=L, **({"q_tolerance": 0.0} if lm else {}))
        s.solver.init(s.params)
        for _ in range(3):
            s.solver.step(s.params)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for _ in range(STEPS):
            if not s.solver.step(s.params):
                break
            n += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        key = ("lm_" if lm else "gn_") + name
        out[key + "_us_per_pcg_iter"] = round(dt / max(n, 1) / L * 1e6, 2)
        out[key + "_steps"] = n
        out[key + "_cost"] = s.solver.current_cost()
        out[key + "_exchange"] = s.solver.distributed_info()["exchange"]
        s.solver.close()
print(json.dumps(out))
dist.destroy_process_group()
