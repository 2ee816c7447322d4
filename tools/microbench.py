"""Kernel micro-benchmarks on one MI355X (run through gpurun).  Not part of the product or the tests:
times individual shim kernels at the benchmark size with HIP events, plus diagnostic variants."""
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

W = H = int(os.environ.get("MB_SIZE", "2048"))
REPS = int(os.environ.get("MB_REPS", "30"))
L = thallo_amd.lib()
L.thallo_hip_vector_elems.restype = C.c_long; L.thallo_hip_vector_elems.argtypes = [C.c_long]
p = syn.image_warping(W, H)
N = W * H; n = 3 * N; na = L.thallo_hip_vector_elems(n)
dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in p]
f = lambda: torch.zeros(na, dtype=torch.float32, device="cuda")
r, pre, z, p0, p1, delta, Ap = [f() for _ in range(7)]
cs = torch.zeros(2 * N, dtype=torch.float32, device="cuda"); flags = torch.zeros(N + 256, dtype=torch.uint8, device="cuda")
parts = torch.zeros(16 * 1024, dtype=torch.float32, device="cuda")
irregular = torch.zeros(16, dtype=torch.int32, device="cuda")
vp, fl = C.c_void_p, C.c_float
PB = parts.data_ptr()
nb0 = L.thallo_hip_iw_pcg_init(W, H, 0, H, vp(dev[0].data_ptr()), vp(dev[1].data_ptr()), vp(dev[2].data_ptr()), vp(dev[3].data_ptr()),
                               vp(dev[4].data_ptr()), fl(p[5]), fl(p[6]), vp(r.data_ptr()), vp(pre.data_ptr()), vp(z.data_ptr()),
                               vp(p0.data_ptr()), vp(delta.data_ptr()), vp(cs.data_ptr()), vp(flags.data_ptr()), None, vp(irregular.data_ptr()), vp(PB), None)
s_aN = api.SumT(PB, nb0)


def timeit(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def step1_fused(first=0, zfree=True):
    nb = L.thallo_hip_iw_pcg_step1(W, H, 0, H, vp(cs.data_ptr()), vp(dev[2].data_ptr()), vp(flags.data_ptr()), fl(p[5]), fl(p[6]),
                                   vp(z.data_ptr()), vp(p0.data_ptr()), vp(p1.data_ptr()), vp(delta.data_ptr()), vp(Ap.data_ptr()),
                                   first, s_aN, s_aN, s_aN, s_aN, s_aN, vp(irregular.data_ptr()), vp(r.data_ptr()) if zfree else None, vp(PB + 4096), None)
    assert nb > 0
    return nb


def step1_plain():
    nb = L.thallo_hip_iw_apply_jtj(W, H, 0, H, vp(cs.data_ptr()), vp(dev[2].data_ptr()), vp(flags.data_ptr()), fl(p[5]), fl(p[6]),
                                   vp(z.data_ptr()), vp(Ap.data_ptr()), vp(irregular.data_ptr()), vp(PB + 4096), None)
    assert nb > 0
    return nb


nbD = step1_fused(1)
s_aD = api.SumT(PB + 4096, nbD)


def step2():
    assert L.thallo_hip_pcg_step2(vp(r.data_ptr()), vp(Ap.data_ptr()), vp(pre.data_ptr()), vp(z.data_ptr()), C.c_long(n), s_aN, s_aD, vp(PB + 8192), None) > 0


def step2_iw():
    assert L.thallo_hip_iw_pcg_step2(W, H, 0, H, vp(flags.data_ptr()), fl(p[5]), fl(p[6]), vp(r.data_ptr()), vp(Ap.data_ptr()), vp(pre.data_ptr()),
                                     vp(z.data_ptr()), s_aN, s_aD, vp(irregular.data_ptr()), vp(PB + 8192), None) > 0


res = {}
MB = 1e-6
for nogrid in (0, 1):
    L.thallo_hip_debug_set(4, nogrid)
    for mode in (0, 1):
        L.thallo_hip_debug_set(0, mode)
        res[f"grid{1-nogrid}_fused_dbg{mode}_us"] = round(timeit(step1_fused), 2)
        res[f"grid{1-nogrid}_plain_dbg{mode}_us"] = round(timeit(step1_plain), 2)
L.thallo_hip_debug_set(0, 0); L.thallo_hip_debug_set(4, 0)
res["step1_fused_dbg0_us"] = timeit(step1_fused); res["step1_plain_dbg0_us"] = timeit(step1_plain)
res["step2_us"] = timeit(step2)
res["step2_iw_zfree_us"] = timeit(step2_iw)
res["step1_fused_general_us"] = timeit(lambda: step1_fused(0, False))
res["step2_iw_zfree_alg37_GBs"] = 37 * N / res["step2_iw_zfree_us"] * 1e-3
# HBM references with torch: copy (read 50 MB + write 50 MB) and 3-read/2-write elementwise
a = torch.randn(na, device="cuda"); b = torch.randn(na, device="cuda"); c = torch.empty(na, device="cuda")
res["torch_copy_us"] = timeit(lambda: c.copy_(a))
res["torch_copy_GBs"] = 2 * na * 4 / res["torch_copy_us"] * 1e-3
big_a = torch.randn(64 * 1024 * 1024, device="cuda"); big_c = torch.empty_like(big_a)
t = timeit(lambda: big_c.copy_(big_a), 10)
res["torch_copy_256MB_GBs"] = 2 * big_a.numel() * 4 / t * 1e-3
res["step1_fused_alg96_GBs"] = 96 * N / res["step1_fused_dbg0_us"] * 1e-3
res["step1_fused_actual89_GBs"] = 89 * N / res["step1_fused_dbg0_us"] * 1e-3
res["step1_plain_alg48_GBs"] = 48 * N / res["step1_plain_dbg0_us"] * 1e-3
res["step2_actual60_GBs"] = 60 * N / res["step2_us"] * 1e-3
print(json.dumps(res, indent=1))
