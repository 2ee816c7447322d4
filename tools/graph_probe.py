"""GPU probe: does hipGraph replay of a whole single-GPU Thallo_ProblemStep shorten the launch-bound configurations?  Eager vs replayed,
image_warping at PW x PW (default 512), 100 PCG iterations per GN step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn
W = int(os.environ.get("PW", "512")); H = int(os.environ.get("PH", str(W))); L = 100
if "MARCH_ROWS" in os.environ:      # rows per wave segment of the marching kernel (tools-only knob)
    thallo_amd.lib().thallo_hip_march_debug_set(0, int(os.environ["MARCH_ROWS"]))
p = syn.image_warping(W, H)
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
params = s.make_params(dev)
s.init(params)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n / L * 1e6


eager = timeit(lambda: s.step(params))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
s.set_stream(side.cuda_stream)
with torch.cuda.stream(side):
    s.step(params)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
    s.step(params)
torch.cuda.synchronize()
replay = timeit(g.replay)
print({"size": (W, H), "march": os.environ.get("THALLO_MARCH", "auto"), "rows": os.environ.get("MARCH_ROWS", "auto"), "eager_us_per_pcg_iter": round(eager, 2), "graph_us_per_pcg_iter": round(replay, 2)})
