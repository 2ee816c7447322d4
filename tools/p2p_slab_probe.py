"""Probe (1 GPU, backend nccl, world_size 1 with the collective code path forced): per-PCG-iteration time of ONE rank's slab
(W x H = what one of 8 ranks owns at 2048^2 -> 2048 x 256) through (a) RCCL collectives, eager and graph-replayed, and
(b) the device-side exchange, eager and graph-replayed.  Gives the per-rank kernel + protocol cost without xGMI latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from thallo_amd import synthetic as syn
from thallo_amd.distributed import make_hip_solver

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
if 'THALLO_ITER_PER_CU' in os.environ:
    from thallo_amd import api
    api.lib().thallo_hip_debug_set(8, int(os.environ['THALLO_ITER_PER_CU']))
if 'THALLO_ITER_NT' in os.environ:
    from thallo_amd import api
    api.lib().thallo_hip_debug_set(7, int(os.environ['THALLO_ITER_NT']))
for name, p2p in (("rccl", False), ("p2p", True)):
    s, _ = make_hip_solver(p, W, H, 0, 1, L, ipc=p2p)
    s.use_dist = True
    if p2p:
        print("p2p enabled:", s.try_enable_p2p(), s.p2p_check)
    step = s.gn_step_p2p if p2p else s.gn_step
    out[name + "_eager_us"] = timeit(lambda: step(L))
    ok = s.capture_gn_step(L)
    print(name, "captured:", ok, getattr(s, "_graph_error", None))
    if ok:
        out[name + "_graph_us"] = timeit(lambda: s.gn_step_fast(L))
    if p2p:
        print("p2p error word:", s.be.p2p_error())
    out[name + "_cost"] = s.cost()
print({k: round(v, 2) for k, v in out.items()})
dist.destroy_process_group()
