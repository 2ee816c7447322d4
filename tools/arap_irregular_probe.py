"""GPU probe: ARAP on a 102,400-vertex mesh of irregular degree (the torus + chords: degrees 6 .. 6 + 2 * chords, like a real triangle mesh), us per PCG iteration with the
resident loop (edge slots beyond 6 through memory) against one launch per iteration (degrees <= 8: the recomputing applyJTJ; above: the stored-block kernel)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
nu = nv = 320; L = 100
def mesh(chords, seed=9):
    p = syn.arap_mesh(nu, nv); N = p[2].shape[0]
    rng = np.random.default_rng(seed); v0, v1 = [p[6]], [p[7]]
    idx = np.arange(N); iu, iv = idx % nu, idx // nu
    for c in range(chords):
        pick = rng.random(N) < 0.3
        tgt = (iv * nu + (iu + 2) % nu) if c % 2 == 0 else (((iv + 2) % nv) * nu + iu)
        a, b = idx[pick].astype(p[6].dtype), tgt[pick].astype(p[6].dtype)
        v0 += [a, b]; v1 += [b, a]
    p[6] = np.ascontiguousarray(np.concatenate(v0)); p[7] = np.ascontiguousarray(np.concatenate(v1))
    return p
def run(p, resident, steps=15, warm=3):
    os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
    dims = (p[2].shape[0], p[6].shape[0])
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver(dims, thallo_amd.energy_file("arap_mesh_deformation"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(warm): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = sorted(s.kernel_stats()); s.close()
    return {"us_per_pcg_iter": round(dt / (steps * L) * 1e6, 2), "kernels": names}
out = {}
for chords in (0, 1, 2, 3):
    p = mesh(chords); deg = np.bincount(p[6], minlength=p[2].shape[0])
    out["chords_%d" % chords] = {"max_degree": int(deg.max()), "edges": int(p[6].shape[0]), "resident": run(p, True), "launch_per_iteration": run(p, False)}
print(json.dumps(out))
