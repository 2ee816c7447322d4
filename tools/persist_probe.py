"""[RESEARCH build: make -C thallo_amd/csrc VARIANT=research (stamps: EXTRA with -DTHALLO_RESEARCH), run with THALLO_LIB=tools/ab/libThallo_research.so -- the loop this probes is not in the product library since round 6]
Where an iteration of the persistent marching loop spends its time: the stamps build (make VARIANT=pstamps EXTRA=-DPST_STAMPS, loaded through THALLO_LIB).
Runs on the GPU box.  python tools/persist_probe.py  ->  gpurun_out/persist_stamps.txt"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("THALLO_LIB", os.path.join(ROOT, "tools", "ab", "libThallo_pstamps.so"))
os.environ["THALLO_AB"] = "persist=1"
import torch, thallo_amd
from thallo_amd import synthetic as syn
W = H = int(os.environ.get("PP_SIZE", 2048))
L = thallo_amd.lib()
nwg = 256
buf = torch.zeros(nwg * 4 * 4 * 8, dtype=torch.int64, device="cuda")
L.thallo_hip_debug_stamps_persist.argtypes = [C.c_void_p]
assert L.thallo_hip_debug_stamps_persist(C.c_void_p(buf.data_ptr())) == 0
if "PP_ACQ" in os.environ: L.thallo_hip_iw_march_persist_debug_set(0, int(os.environ["PP_ACQ"]))
if "PP_RES" in os.environ: L.thallo_hip_iw_march_persist_debug_set(1, int(os.environ["PP_RES"]))
p = syn.image_warping(W, H)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
s.set_solver_parameters(nIterations=6, lIterations=100)
params = s.make_params(dev); s.init(params)
for _ in range(4): s.step(params)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(nwg, 4, 4, 8).astype(np.float64) / 100.0        # us
live = st[:, :, :, 0].min(axis=2) > 0
names = ["poll", "lds+barrier", "sums", "march", "drain", "barrier2", "publish->next top"]
out = []
for it in range(1, 4):
    t = st[:, :, it, :][live]                  # waves x 8
    tprev = st[:, :, it - 1, :][live]
    t0 = t[:, 0].min()
    out.append(f"iteration {it}: top min {0:.2f} max {t[:,0].max()-t0:.2f} | poll done mean {np.mean(t[:,1]-t0):.2f} max {np.max(t[:,1]-t0):.2f} | alpha ready mean {np.mean(t[:,3]-t0):.2f} max {np.max(t[:,3]-t0):.2f} | "
               f"march end min {np.min(t[:,4]-t0):.2f} mean {np.mean(t[:,4]-t0):.2f} max {np.max(t[:,4]-t0):.2f} | published max {np.max(t[:,6]-t0):.2f}")
    d = np.diff(t[:, :7], axis=1)
    out.append("   per wave phase means (us): " + ", ".join(f"{n} {np.mean(d[:, i]):.2f}" for i, n in enumerate(names[:6])))
    out.append(f"   march duration per wave: min {np.min(t[:,4]-t[:,3]):.2f} mean {np.mean(t[:,4]-t[:,3]):.2f} max {np.max(t[:,4]-t[:,3]):.2f}; iteration period (top to top, mean) {np.mean(t[:,0]-tprev[:,0]):.2f}")
open(os.path.join(ROOT, "gpurun_out", "persist_stamps.txt"), "a").write("PP_RES=%s PP_ACQ=%s\n" % (os.environ.get("PP_RES"), os.environ.get("PP_ACQ")) + "\n".join(out) + "\n")
print("\n".join(out))
# ---- who is slow: the march time per workgroup as a (band of 4 segments) x (strip) map, and per XCD group (blockIdx % 8)
nstrips = (W + 123) // 124
R = int(os.environ.get("PP_R", 35)); nseg = (H + R - 1) // R; nwgrow = (nseg + 3) // 4; total = nstrips * nwgrow
grid = (total + 7) // 8 * 8
m = np.full((nwgrow, nstrips), np.nan); xcd = [[] for _ in range(8)]
it = 2
for b in range(grid):
    grp, l = b % 8, b // 8
    lo, hi = total * grp // 8, total * (grp + 1) // 8
    if lo + l >= hi: continue
    idn = lo + l
    strip, band = idn % nstrips, idn // nstrips
    d = st[b, :, it, 4] - st[b, :, it, 3]
    m[band, strip] = d.max()
    xcd[grp].append(d.max())
lines = ["march time of the slowest wave per workgroup, rows = bands of 4 segments, columns = strips (us):"]
for r in range(nwgrow): lines.append(" ".join(f"{v:5.1f}" for v in m[r]))
lines.append("per blockIdx % 8 group: " + ", ".join(f"{np.mean(x):.1f}/{np.max(x):.1f}" for x in xcd))
w = np.stack([st[:, wv, it, 4] - st[:, wv, it, 3] for wv in range(4)], axis=1)
lines.append("per wave index in the workgroup (mean over workgroups with rows): " + ", ".join(f"{np.mean(w[w[:, 0] > 1, i]):.1f}" for i in range(4)))
open(os.path.join(ROOT, "gpurun_out", "persist_stamps.txt"), "a").write("\n".join(lines) + "\n")
print("\n".join(lines))
