"""[RESEARCH build: make -C thallo_amd/csrc VARIANT=research (stamps: EXTRA with -DTHALLO_RESEARCH), run with THALLO_LIB=tools/ab/libThallo_research.so -- the loop this probes is not in the product library since round 6]
Where an iteration of bundle adjustment's resident PCG loop spends its time: the stamps build (make VARIANT=bstamps EXTRA=-DBRES_STAMPS, loaded through THALLO_LIB).
Runs on the GPU box.  python tools/ba_resident_probe.py [workgroups]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("THALLO_LIB", os.path.join(ROOT, "tools", "ab", "libThallo_bstamps.so"))
os.environ["THALLO_RESIDENT"] = "2"
import torch, thallo_amd
from thallo_amd import synthetic as syn
L = thallo_amd.lib()
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if wg: L.thallo_hip_ba_resident_debug_set(0, wg)
buf = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device="cuda")
L.thallo_hip_debug_stamps_ba.argtypes = [C.c_void_p]
assert L.thallo_hip_debug_stamps_ba(C.c_void_p(buf.data_ptr())) == 0
p = syn.bundle_adjustment(); dims = (1723, 156502, 678718)
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("bundle_adjustment"), timing_level=0)
s.set_solver_parameters(nIterations=3, lIterations=30)
prm = s.make_params(dev); s.init(prm)
while s.step(prm): pass
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(1024, 4, 8).astype(np.float64) / 100.0
live = st[:, 1, 0] > 0
t = st[live][:, 1, :]          # iteration 9
names = ["U", "barrier 1", "A (camera)", "barrier 2", "B (point)", "barrier 3", "totals"]
t0 = t[:, 0].min()
print(f"workgroups {live.sum()}: iteration period {np.mean(st[live][:, 2, 0] - st[live][:, 1, 0]):.2f} us")
d = np.diff(t, axis=1)
for i, nme in enumerate(names): print(f"  {nme:12s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f}")
print("  phase ends (max over workgroups, from the earliest start):", ", ".join(f"{t[:, i].max() - t0:.2f}" for i in range(8)))
