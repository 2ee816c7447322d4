"""Probe (1 GPU, backend nccl, world_size 1 with the all-gather really issued): per-PCG-iteration time of ONE rank's slab
(W x H = what one of 8 ranks owns at 2048^2 -> 2048 x 256) through the library's slab path (csrc/solver_dist.cpp) with (a) the RCCL
all-gather transport, eager and graph-replayed, and (b) the device-side exchange, eager and graph-replayed.  Gives the per-rank kernel +
protocol cost without xGMI latency.   PW / PH: slab size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from thallo_amd import synthetic as syn
from thallo_amd.distributed import PlanSlabSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
W = int(os.environ.get("PW", "2048")); H = int(os.environ.get("PH", "256")); L = 100
p = syn.image_warping(W, H)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n / L * 1e6


out = {}
for name, p2p in (("rccl", False), ("p2p", True)):
    s = PlanSlabSolver(p, W, H, 0, 1, L, device_exchange=p2p, force_allgather=True)
    out[name + "_cost0"] = s.cost()
    print(name, s.info)
    out[name + "_eager_us"] = timeit(s.gn_step)
    ok = s.capture()
    print(name, "captured:", ok, getattr(s, "_graph_error", None))
    if ok:
        out[name + "_graph_us"] = timeit(s.gn_step_fast)
    if p2p:
        print("p2p error word:", s.solver.distributed_error())
    s.drop_graph()
    out[name + "_cost"] = s.cost()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    s.solver.distributed_kernel_only(5); t0.record(); s.solver.distributed_kernel_only(50); t1.record(); torch.cuda.synchronize()
    out[name + "_kernel_only_us"] = t0.elapsed_time(t1) / 50 * 1e3
print({k: round(v, 2) for k, v in out.items()})
dist.destroy_process_group()
