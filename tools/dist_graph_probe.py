"""Probe (1 GPU): does hipGraph capture of the slab GN step work with RCCL collectives inside?  world_size 1 over
backend nccl exercises the capture path of ProcessGroupNCCL; results are compared with the eager path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from thallo_amd import synthetic as syn
from thallo_amd.distributed import make_hip_solver, SlabSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
W = H = int(os.environ.get("SZ", "1024")); L = 50
p = syn.image_warping(W, H)
a, _ = make_hip_solver(p, W, H, 0, 1, L)
b, _ = make_hip_solver(p, W, H, 0, 1, L)
# force the collective code path even at world 1
a.use_dist = True; b.use_dist = True
ca = [a.cost()]
for _ in range(3):
    a.gn_step(L); ca.append(a.cost())
ok = b.capture_gn_step(L)
print("captured:", ok, getattr(b, "_graph_error", None))
# b has run 2 steps (warm-up + capture).  replay a third
cb2 = b.cost()
b.gn_step_fast(L); cb3 = b.cost()
print("eager  costs", ca)
print("graph  cost after 2 steps", cb2, "after 3", cb3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): a.gn_step(L)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5): b.gn_step_fast(L)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
print(f"eager {te*1e3:.2f} ms/step ({te/L*1e6:.1f} us/iter)   graph {tg*1e3:.2f} ms/step ({tg/L*1e6:.1f} us/iter)")
dist.destroy_process_group()
