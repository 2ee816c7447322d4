"""rc_slab_check.py -- GPU probe: the slab form of the marching iteration without the A p plane (thallo_hip_iw_pcg_iter_march_rc with row0 > 0 / row1 < H)
against the stored-plane kernel on the same slab and inputs: r, p, delta of owned + ghost rows, the boundary rows of A p, the partial sums -- bit for bit."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import thallo_amd
from thallo_amd import api, synthetic as syn

W = int(os.environ.get("MB_W", "128")); H = int(os.environ.get("MB_H", "50"))
row0, row1 = int(os.environ.get("MB_ROW0", "1")), int(os.environ.get("MB_ROW1", str(H - 1)))
L = thallo_amd.lib()
L.thallo_hip_vector_elems.restype = C.c_long; L.thallo_hip_vector_elems.argtypes = [C.c_long]
p = syn.image_warping(W, H, n_markers=8)
N = W * H; n = 3 * N; na = L.thallo_hip_vector_elems(n)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in p]
f = lambda: torch.zeros(na, dtype=torch.float32, device="cuda")
r0, pre, z, p0, delta0, A0 = [f() for _ in range(6)]
cs = torch.zeros(2 * N, dtype=torch.float32, device="cuda"); flags = torch.zeros(N + 256, dtype=torch.uint8, device="cuda")
parts = torch.zeros(8 * 1024, dtype=torch.float32, device="cuda")
s12 = torch.zeros(3 * 1024, dtype=torch.float64, device="cuda")
irregular = torch.zeros(16, dtype=torch.int32, device="cuda")
scal = torch.tensor([1.0, 2.0, 0.3, 1.5, 2.5], dtype=torch.float32, device="cuda")     # aN, aD, bN, aN2, aD2 -> alpha .5, beta .3, alpha2 .6
vp, fl = C.c_void_p, C.c_float
S = lambda i: api.SumT(scal.data_ptr() + 4 * i, 1)
nb0 = L.thallo_hip_iw_pcg_init(W, H, 0, H, vp(dev[0].data_ptr()), vp(dev[1].data_ptr()), vp(dev[2].data_ptr()), vp(dev[3].data_ptr()),
                               vp(dev[4].data_ptr()), fl(p[5]), fl(p[6]), vp(r0.data_ptr()), vp(pre.data_ptr()), vp(z.data_ptr()),
                               vp(p0.data_ptr()), vp(delta0.data_ptr()), vp(cs.data_ptr()), vp(flags.data_ptr()), None, vp(irregular.data_ptr()), vp(parts.data_ptr()), None)
assert nb0 > 0
torch.cuda.synchronize()


def run(kind, mode, r_in, A_in, p_in, dl, pk2=None):
    r_out, A_out, p_out = f(), f(), f()
    d = dl.clone()
    if ((mode >> 1) & 3) == 2:
        p_out.copy_(pk2)
    parts.zero_(); s12.zero_()
    fn = L.thallo_hip_iw_pcg_iter_march if kind == "stored" else L.thallo_hip_iw_pcg_iter_march_rc
    nb = fn(W, H, row0, row1, vp(cs.data_ptr()), vp(flags.data_ptr()), fl(p[5]), fl(p[6]),
            vp(r_in.data_ptr()), vp(r_out.data_ptr()), vp(A_in.data_ptr()), vp(A_out.data_ptr()), vp(p_in.data_ptr()), vp(p_out.data_ptr()),
            vp(d.data_ptr()), mode, S(0), S(1), S(2), S(3), S(4), vp(irregular.data_ptr()), vp(parts.data_ptr() + 4096), vp(s12.data_ptr()), None, None, None, None)
    assert nb > 0, nb
    torch.cuda.synchronize()
    return dict(r=r_out, A=A_out, p=p_out, d=d, aD=parts[1024:1024 + nb].clone(), s=s12[:3 * nb].clone(), nb=nb)


def rows_of(v, rows):       # the pixels of the given rows in a solver vector [Offset 2N | Angle N]
    o = v[:2 * N].view(H, W, 2)[rows]; a = v[2 * N:3 * N].view(H, W)[rows]
    return torch.cat([o.reshape(-1), a.reshape(-1)])


it0 = run("stored", 1, r0, A0, p0, delta0)                    # the first iteration (always the stored-plane kernel)
it1 = run("stored", 2, it0["r"], it0["A"], it0["p"], delta0)  # mode 2 = none deferred
out = {"W": W, "H": H, "row0": row0, "row1": row1}
allrows = list(range(H)); bnd = [row0, row1 - 1]
for mode, (ri, Ai, pi, pk2) in {2: (it0["r"], it0["A"], it0["p"], None), 0: (it0["r"], it0["A"], it0["p"], None), 4: (it1["r"], it1["A"], it1["p"], it0["p"])}.items():
    a = run("stored", mode, ri, Ai, pi, delta0, pk2)
    b = run("rc", mode, ri, Ai, pi, delta0, pk2)
    e = {}
    for k in ("r", "p", "d"):
        x, y = rows_of(a[k], allrows), rows_of(b[k], allrows)
        bad = (x.view(torch.int32) != y.view(torch.int32))
        e[k + "_mismatch"] = int(bad.sum().item())
        if e[k + "_mismatch"]:
            xo = a[k][:2 * N].view(H, W, 2); yo = b[k][:2 * N].view(H, W, 2)
            rows_bad = sorted(set(torch.nonzero((xo != yo).any(2).any(1)).flatten().tolist()))
            e[k + "_bad_rows"] = rows_bad[:12]
    x, y = rows_of(a["A"], bnd), rows_of(b["A"], bnd)
    e["A_boundary_mismatch"] = int((x.view(torch.int32) != y.view(torch.int32)).sum().item())
    e["nb"] = [a["nb"], b["nb"]]
    e["aD_equal"] = bool(torch.equal(a["aD"], b["aD"])); e["s_equal"] = bool(torch.equal(a["s"], b["s"]))
    out[f"mode{mode}"] = e
print(json.dumps(out))
if os.environ.get("MB_DUMP"):
    mode = 2
    a = run("stored", mode, it0["r"], it0["A"], it0["p"], delta0); b = run("rc", mode, it0["r"], it0["A"], it0["p"], delta0)
    rr = int(os.environ["MB_DUMP"])
    for k in ("r", "p"):
        xa = a[k][:2 * N].view(H, W, 2)[rr, :6].flatten().tolist(); xb = b[k][:2 * N].view(H, W, 2)[rr, :6].flatten().tolist(); xi = it0[k][:2 * N].view(H, W, 2)[rr, :6].flatten().tolist()
        print(k, "in", xi); print(k, "stored", xa); print(k, "rc", xb)
    print("A_in row", it0["A"][:2 * N].view(H, W, 2)[rr, :6].flatten().tolist())
    L.thallo_hip_iw_march_rows.restype = C.c_int
    print("R", L.thallo_hip_iw_march_rows(W, row1 - row0))
