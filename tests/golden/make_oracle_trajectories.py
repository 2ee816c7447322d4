"""make_oracle_trajectories.py -- full-size cost trajectories of the CPU oracle (oracle/thallo_oracle.c) for the BASELINE configurations, as a committed fixture.

The GPU suite used to run these oracle solves live on the GPU box's host cores: 2 x 45 s for the ladybug-shaped bundle adjustment LM 5 x 150, 2 x 40 s for
shape_from_shading 2048^2 LM, 20 s for cat512 -- 40 % of a suite that has to finish inside the driver's time limit, and the reference budget of
shape_from_shading (60 x 10, examples/shape_from_shading/src/main.cpp:44-53) was out of reach altogether.  They are inputs-in / costs-out of deterministic seeded
instances, i.e. golden vectors: this script (run once, in the build container: `python tests/golden/make_oracle_trajectories.py [name ...]`) writes
tests/golden/oracle_trajectories.json, and tests/test_gpu_parity.py compares the GPU trajectories with it.  Every entry carries a checksum of the instance's input
arrays; a test whose inputs do not reproduce it (another numpy / libm) runs the oracle live instead, as before.

What is stored per instance: `double` = the oracle's default mode (double accumulators; threaded, i.e. the atomics' order is not fixed: two runs differ by ~3e-7 on
bundle adjustment -- stored as `rerun_spread` where measured), `float_order` = the serial float summation order of the reference's CPU mode (cpu_cuda.t:265-301),
`pcg_counts` = PCG iterations per LM step (the zeta test, gauss_newton.t:1666-1686).
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as orc
from thallo_amd import synthetic as syn
from thallo_amd import formats as F
from helpers import copy_params

OUT = os.path.join(ROOT, "tests", "golden", "oracle_trajectories.json")
GOLD = os.path.join(ROOT, "tests", "golden")


def checksum(params):
    h = hashlib.sha256()
    for a in params:
        if isinstance(a, np.ndarray):
            h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(np.ascontiguousarray(a).tobytes())
        else:
            h.update(np.float64(a).tobytes())
    return h.hexdigest()[:32]


def solve(kind, dims, p, threads, **kw):
    prev = orc.set_threads(threads)
    t0 = time.time()
    try:
        c, _ = orc.Problem(kind, dims, copy_params(p)).solve(**kw)
    finally:
        orc.set_threads(prev)
    return [float(x) for x in c], [int(x) for x in orc.last_pcg_counts()], time.time() - t0


def cat512_params():
    mask = F.read_png(os.path.join(GOLD, "cat512_mask.png"))[:, :, 0].astype(np.float32)
    H, W = mask.shape
    cons = F.add_border_constraints(F.read_constraints(os.path.join(GOLD, "cat512.constraints")), W, H)
    yy, xx = np.mgrid[0:H, 0:W]
    ur = np.stack([xx, yy], axis=2).astype(np.float32)
    wf, wr = float(np.sqrt(np.float32(100.0))), float(np.sqrt(np.float32(0.01)))
    c_img = F.constraint_image(cons, mask, np.float32(1) / np.float32(19))
    return (W, H), [ur.copy(), np.zeros((H, W), dtype=np.float32), ur.copy(), c_img, mask, wf, wr]


def main():
    want = set(sys.argv[1:])
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    nthreads = max(1, os.cpu_count() or 1)

    def todo(name):
        return not want or name in want

    if todo("ba_ladybug_lm_5x150"):
        p = syn.bundle_adjustment()
        dims = (p[0].shape[0], p[1].shape[0], p[2].shape[0])
        c1, n1, t1 = solve(orc.BUNDLE_ADJUST, dims, p, nthreads, nIterations=5, lIterations=150, use_lm=1)
        c2, n2, t2 = solve(orc.BUNDLE_ADJUST, dims, p, nthreads, nIterations=5, lIterations=150, use_lm=1)
        den = np.maximum(np.array(c1), 1e-3 * c1[0])
        data["ba_ladybug_lm_5x150"] = {"dims": list(dims), "input_checksum": checksum(p), "double": c1, "pcg_counts": n1,
                                       "rerun_spread": [float(x) for x in np.abs(np.array(c2) - np.array(c1)) / den], "seconds": [t1, t2], "threads": nthreads}
        json.dump(data, open(OUT, "w"), indent=1)
    if todo("cat512_8x100"):
        (W, H), p = cat512_params()
        c1, n1, t1 = solve(orc.IMAGE_WARPING, (W, H), p, nthreads, nIterations=8, lIterations=100)
        cf, nf, tf = solve(orc.IMAGE_WARPING, (W, H), p, 1, nIterations=8, lIterations=100, float_sums=1)
        data["cat512_8x100"] = {"dims": [W, H], "input_checksum": checksum(p), "double": c1, "float_order": cf, "seconds": [t1, tf], "threads": nthreads}
        json.dump(data, open(OUT, "w"), indent=1)
    if todo("sfs2048_gn_2x10"):
        W = H = 2048
        p = syn.shape_from_shading(W, H)
        c1, n1, t1 = solve(orc.SFS, (W, H), p, nthreads, nIterations=2, lIterations=10)
        data["sfs2048_gn_2x10"] = {"dims": [W, H], "input_checksum": checksum(p), "double": c1, "seconds": [t1], "threads": nthreads}
        json.dump(data, open(OUT, "w"), indent=1)
    if todo("sfs2048_lm_60x10"):
        W = H = 2048
        p = syn.shape_from_shading(W, H)
        c1, n1, t1 = solve(orc.SFS, (W, H), p, nthreads, nIterations=60, lIterations=10, use_lm=1)
        data["sfs2048_lm_60x10"] = {"dims": [W, H], "input_checksum": checksum(p), "double": c1, "pcg_counts": n1, "seconds": [t1], "threads": nthreads,
                                    "budget": "examples/shape_from_shading/src/main.cpp:44-53: 60 nonlinear x 10 linear iterations"}
        json.dump(data, open(OUT, "w"), indent=1)
    if todo("sfs2048_lm_float_order_12x10"):
        W = H = 2048
        p = syn.shape_from_shading(W, H)
        cf, nf, tf = solve(orc.SFS, (W, H), p, 1, nIterations=12, lIterations=10, use_lm=1, float_sums=1)
        data["sfs2048_lm_float_order_12x10"] = {"dims": [W, H], "input_checksum": checksum(p), "float_order": cf, "pcg_counts": nf, "seconds": [tf], "threads": 1}
        json.dump(data, open(OUT, "w"), indent=1)
    print(json.dumps({k: {kk: (vv if not isinstance(vv, list) or len(vv) < 14 else vv[:6] + ["..."]) for kk, vv in v.items()} for k, v in data.items()}, indent=1))


if __name__ == "__main__":
    main()
