"""GPU probe: one GN step of image_warping (Thallo_ProblemStep, 100 PCG iterations) launched eagerly against the same step captured into a HIP graph and replayed -- what the
launch path costs per PCG iteration at the benchmark size and at mid sizes.  python tools/graph_probe.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, thallo_amd
from thallo_amd import synthetic as syn
os.environ.setdefault("THALLO_RESIDENT", "0")
def run(w, h, L=100, steps=20):
    p = syn.image_warping(w, h)
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((w, h), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    prm = s.make_params(dev); s.init(prm)
    for _ in range(3): s.step(prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.step(prm)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / (steps * L) * 1e6
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph(); out = {"eager_us_per_pcg_iter": round(eager, 2)}
    try:
        s.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            s.step(prm); side.synchronize()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                s.step(prm)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): g.replay()
        torch.cuda.synchronize(); out["graph_us_per_pcg_iter"] = round((time.perf_counter() - t0) / (steps * L) * 1e6, 2)
    except Exception as e:
        out["graph_error"] = repr(e)[:200] + " | " + thallo_amd.last_error()[:200]
    out["cost"] = s.current_cost()
    return out
print("JSON " + json.dumps({f"{w}x{h}": run(w, h) for (w, h) in ((2048, 2048), (2048, 1024), (1280, 1024), (1024, 768))}))
