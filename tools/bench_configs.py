"""Secondary configs of BASELINE.json (not the headline bench): ms per GN iteration / PCG it/s on one MI355X for
image_warping 512^2, ARAP 102,400 vertices, shape_from_shading 2048^2, bundle adjustment ladybug-1723 shape.
Prints one JSON object; kernel means from the library's sampled HIP events."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import thallo_amd
from thallo_amd import synthetic as syn


def run(name, fname, dims, params, nit, lit, warm=1, lm=False):
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in params]
    # BC_TIMING (default 0 = Thallo.h's "No timing recorded"): at level 1 every step records eight coarse events, each a barrier packet between two launches -- ~30 us per GN
    # step, which only the small configurations notice
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file(fname), timing_level=int(os.environ.get("BC_TIMING", "0")), **({"solverkind": "levenberg_marquardt"} if lm else {}))
    if lm:
        s.enable_lm()
    s.set_solver_parameters(nIterations=2 * nit + warm, lIterations=lit, **({"q_tolerance": 0.0} if lm else {}))
    prm = s.make_params(dev)
    s.init(prm)
    c0 = s.current_cost()
    for _ in range(warm):
        s.step(prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()            # the timed pass: no events of any kind in the stream
    for _ in range(nit):
        s.step(prm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.reset_kernel_stats(); s.set_kernel_sampling(4)
    for _ in range(nit):                # a second pass for the per-kernel means (HIP events around every fourth launch of a name)
        s.step(prm)
    torch.cuda.synchronize()
    s.set_kernel_sampling(0)
    ks = {k: round(v["mean_ms"] * 1e3, 2) for k, v in s.kernel_stats().items() if v["mean_ms"]}
    return {"config": name, "ms_per_gn_iter": dt / nit * 1e3, "pcg_iters_per_sec": nit * lit / dt, "us_per_pcg_iter": dt / (nit * lit) * 1e6,
            "cost0": c0, "cost": s.current_cost(), "kernel_mean_us": ks}


def cat512():
    """BASELINE.json configs[1] on the reference's own data: cat512 mask + markers + pinned border (tests/golden fixtures),
    first step of the harness' continuation (targets at 1/19 of the way)."""
    from thallo_amd import formats as F
    g = os.path.join(ROOT, "tests", "golden")
    mask = F.read_png(os.path.join(g, "cat512_mask.png"))[:, :, 0].astype(np.float32)
    H, W = mask.shape
    cons = F.add_border_constraints(F.read_constraints(os.path.join(g, "cat512.constraints")), W, H)
    yy, xx = np.mgrid[0:H, 0:W]
    ur = np.stack([xx, yy], axis=2).astype(np.float32)
    return [ur.copy(), np.zeros((H, W), np.float32), ur.copy(), F.constraint_image(cons, mask, np.float32(1.0 / 19.0)), mask,
            float(np.sqrt(np.float32(100.0))), float(np.sqrt(np.float32(0.01)))]


only = sys.argv[1] if len(sys.argv) > 1 else ""      # e.g. "sfs", "ba", "image_warping", "arap": the configurations whose name contains it
want = lambda name: only in name
out = []
if want("image_warping cat512"): out.append(run("image_warping cat512 (reference data) GN 8x100", "image_warping", (512, 512), cat512(), 8, 100))
if want("image_warping 512"): out.append(run("image_warping 512x512 synthetic GN 8x100", "image_warping", (512, 512), syn.image_warping(512, 512), 8, 100))
if want("arap"):
    p = syn.arap_mesh(320, 320)
    out.append(run("arap_mesh 102400 v / 614400 e GN 20x100", "arap_mesh_deformation", (p[2].shape[0], p[6].shape[0]), p, 5, 100))
if want("shape_from_shading") or only == "sfs":
    out.append(run("shape_from_shading 2048x2048 GN x10", "shape_from_shading", (2048, 2048), syn.shape_from_shading(2048, 2048), 6, 10))
    out.append(run("shape_from_shading 2048x2048 LM x10 (BASELINE config 4's solver)", "shape_from_shading", (2048, 2048), syn.shape_from_shading(2048, 2048), 5, 10, lm=True))
if only == "sfs640":      # (not in the default set: tools/profile_configs.sh averages per kernel NAME, so every profiled configuration of an energy has one size)
    out.append(run("shape_from_shading 640x480 GN x10 (the size of the reference's data set)", "shape_from_shading", (640, 480), syn.shape_from_shading(640, 480), 12, 10))
    out.append(run("shape_from_shading 640x480 LM x10", "shape_from_shading", (640, 480), syn.shape_from_shading(640, 480), 10, 10, lm=True))
if want("bundle_adjustment") or only == "ba":
    p = syn.bundle_adjustment()
    out.append(run("bundle_adjustment C=1723 P=156502 O=678718 LM x150 (BASELINE config 5's solver)", "bundle_adjustment", (p[0].shape[0], p[1].shape[0], p[2].shape[0]), p, 3, 150, lm=True))
    out.append(run("bundle_adjustment C=1723 P=156502 O=678718 GN x150", "bundle_adjustment", (p[0].shape[0], p[1].shape[0], p[2].shape[0]), p, 3, 150))
print(json.dumps(out, indent=1))
