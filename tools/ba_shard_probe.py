"""Probe (1 GPU, world size 1): bundle adjustment (ladybug-1723 shape: 156,502 points, padded to a multiple of 4 by BaShardLayout) through the library's shard path
(csrc/solver_dist.cpp: dist_gn_shard) with the device-side scalar exchange (thallo_hip_dist_xscalars_shard) and with pack + all-gather + scalars; at world size 1 the
all-reduce itself has nothing to add and is skipped on both paths (the device-side one is exercised by 2-3 ranks sharing the GPU in tests/test_gpu_distributed.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from thallo_amd import synthetic as syn
from thallo_amd.distributed_ba import PlanBaShardSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
p = syn.bundle_adjustment()
L, STEPS = 150, 4
out = {"shape": "C=1723 P=156502 O=678718", "l_iters": L}
for name, dx in (("collectives", False), ("device", True)):
    s = PlanBaShardSolver(p, 0, 1, L, device_exchange=dx)
    s.solver.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    s.solver.init(s.params)
    s.solver.step(s.params)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        s.solver.step(s.params)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name + "_us_per_pcg_iter"] = round(dt / STEPS / L * 1e6, 2)
    out[name + "_exchange"] = s.solver.distributed_info()["exchange"]
    out[name + "_cost"] = s.solver.current_cost()
    s.solver.close()
print(json.dumps(out))
dist.destroy_process_group()
