"""Debug probe (GPU box): world-1 device-side exchange vs the plain slab path; prints both alpha/beta traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thallo_amd import synthetic as syn
from thallo_amd.distributed import make_hip_solver

W, H, L = 64, 48, 6
p = syn.image_warping(W, H, n_markers=8)
solver, lay = make_hip_solver(p, W, H, 0, 1, 10, ipc=True)
be = solver.be
be.enable_p2p(None)
X0, A0 = be.offset.clone(), be.angle.clone()
solver.gn_step(L)
ref = be.S[2:2 + 2 * L + 1].cpu().numpy()
be.offset.copy_(X0); be.angle.copy_(A0)
solver.gn_step_p2p(L)
got = be.S[2:2 + 2 * L + 1].cpu().numpy()
print("ref", ref); print("got", got); print("err", be.p2p_error())
mail = torch.as_tensor(type("R", (), {"__cuda_array_interface__": {"shape": (2 * (2 * L + 4),), "typestr": "<u4", "data": (be.p2p.mail, False), "version": 2}})(), device="cuda")
print("mail", mail.cpu().numpy().view(np.uint32).reshape(-1, 2)[:2 * L + 4])
print("ctl", be.ctl.cpu().numpy()[:4])
# ---- isolate step2: same inputs through the plain and the dist kernel
import ctypes as C
from thallo_amd import api
be.offset.copy_(X0); be.angle.copy_(A0)
be.init(0)
solver._gather_sum_and_rows(2)
be.p2p_begin(2)
be.step1_p2p(0, True, 2, 3, 2, 3)
torch.cuda.synchronize()
r0 = be.r.clone(); Ap = be.Ap.clone()
be.p2p_collect(3, 1); torch.cuda.synchronize()
print("S[2], S[3]", be.S[2].item(), be.S[3].item())
be.step2_p2p(2, 3, 4); torch.cuda.synchronize()
r_dist = be.r.clone(); print('dbg', be.parts[1024+600:1024+607].cpu().numpy())
be.p2p_collect(4, 1); torch.cuda.synchronize()
bN_dist = be.S[4].item()
be.r.copy_(r0)
be.step2(2, 3); torch.cuda.synchronize()
bN_plain = be.parts[:be.nb].double().sum().item()
r_plain = be.r.clone()
m = Ap.abs() > 1e-3
print("alpha dist ", ((r0 - r_dist)[m] / Ap[m]).median().item(), " plain", ((r0 - r_plain)[m] / Ap[m]).median().item())
print("bN dist", bN_dist, "plain", bN_plain, " r diff", (r_dist - r_plain).abs().max().item())
