"""GPU probe: the resident PCG loop (thallo_hip_iw_pcg_resident, one launch per GN step) against one launch per PCG iteration of the marching
kernel with the same rows per segment -- bitwise comparison of costs, alpha / beta and unknowns after NGN Gauss-Newton steps, then the time
per PCG iteration of both (and of the default launch-per-iteration schedule at that size).  PW / PH: image size (default 512 x 512)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thallo_amd
from thallo_amd import synthetic as syn

W = int(os.environ.get("PW", "512")); H = int(os.environ.get("PH", str(W))); L = int(os.environ.get("PL", "100")); NGN = int(os.environ.get("NGN", "2"))
Lb = thallo_amd.lib()
Lb.thallo_hip_iw_resident_rows.restype = int
R = Lb.thallo_hip_iw_resident_rows(W, H)
p = syn.image_warping(W, H)
out = {"size": [W, H], "L": L, "resident_rows": R}


def run(env, rows=None, steps=NGN, time_it=False):
    for k in ("THALLO_RESIDENT", "THALLO_MARCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    Lb.thallo_hip_march_debug_set(0, rows or 0)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=0)
    s.set_solver_parameters(nIterations=1 << 30, lIterations=L)
    params = s.make_params(dev)
    s.init(params)
    costs, traces = [s.current_cost()], []
    for _ in range(steps):
        s.step(params); costs.append(s.current_cost()); traces.append(s.alpha_beta_trace())
    us = None
    if time_it:
        s.step(params); torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): s.step(params)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / n / L * 1e6
        c = s.current_cost()
        assert np.isfinite(c), thallo_amd.last_error()
    res = (costs, traces, dev[0].clone(), dev[1].clone(), us)
    s.close(); Lb.thallo_hip_march_debug_set(0, 0)
    return res


if R > 0:
    a = run({}, time_it=True)
    b = run({"THALLO_RESIDENT": "0", "THALLO_MARCH": "2"}, rows=R, time_it=True)
    out["costs_equal"] = a[0] == b[0]
    out["alpha_beta_equal"] = a[1] == b[1]
    out["unknowns_equal"] = bool(torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]))
    if not out["costs_equal"]:
        out["costs"] = [a[0], b[0]]
    if not out["alpha_beta_equal"]:
        for i, (ta, tb) in enumerate(zip(a[1], b[1])):
            bad = [k for k, (x, y) in enumerate(zip(ta, tb)) if x != y]
            if bad:
                out["first_alpha_beta_mismatch"] = [i, bad[0], ta[bad[0]], tb[bad[0]]]; break
    out["resident_us_per_pcg_iter"] = round(a[4], 2)
    out["march_same_rows_us_per_pcg_iter"] = round(b[4], 2)
if os.environ.get("RP_STAMPS") and R > 0:        # needs the sweep build: THALLO_LIB=tools/ab/libThallo_sweep.so
    import ctypes as C
    nw = 1024 * 4
    st = torch.zeros(nw * 4 * 16, dtype=torch.int64, device="cuda")
    assert Lb.thallo_hip_debug_stamps_resident(C.c_void_p(st.data_ptr())) == 0
    run({}, steps=1)
    torch.cuda.synchronize()
    t = st.cpu().numpy().reshape(nw, 4, 16).astype(np.float64)
    live = t[:, 0, 0] > 0
    t = t[live]
    names = ["global poll (sums quarter + row + columns)", "LDS exchange of the sums + butterflies", "LDS rows / columns in", "r / p / delta update", "stencil + halo stores", "wave sums + publish"]
    d = np.diff(t[:, :, :7], axis=2) / 100.0          # us
    out["stamps_waves"] = int(live.sum())
    out["phase_us_mean"] = {n: round(float(d[:, :, i].mean()), 3) for i, n in enumerate(names)}
    out["phase_us_max_wave"] = {n: round(float(d[:, :, i].mean(axis=1).max()), 3) for i, n in enumerate(names)}
    tw = t[t[:, 1, 4] - t[:, 1, 3] > 30]            # waves with rows (their update phase takes time)
    out["working_waves"] = int(len(tw))
    out["working_phase_us"] = {"poll": round(float(((tw[:, :, 1] - tw[:, :, 0]) / 100).mean()), 3), "halo in (LDS)": round(float(((tw[:, :, 8] - tw[:, :, 1]) / 100).mean()), 3),
                               "sums to LDS + wait siblings": round(float(((tw[:, :, 9] - tw[:, :, 8]) / 100).mean()), 3), "butterflies + scalars": round(float(((tw[:, :, 10] - tw[:, :, 9]) / 100).mean()), 3),
                               "update": round(float(((tw[:, :, 4] - tw[:, :, 3]) / 100).mean()), 3), "stencil": round(float(((tw[:, :, 5] - tw[:, :, 4]) / 100).mean()), 3), "publish": round(float(((tw[:, :, 6] - tw[:, :, 5]) / 100).mean()), 3)}
    out["poll_passes_lane0_mean"] = round(float(t[:, :, 7].mean()), 2)
    out["iteration_us"] = round(float(((t[:, 1:, 0] - t[:, :-1, 0]) / 100.0).mean()), 3)
    # skew: how far apart the waves enter an iteration, and leave their stencil
    out["entry_skew_us"] = round(float((t[:, 1, 0].max() - t[:, 1, 0].min()) / 100.0), 3)
    out["stencil_end_skew_us"] = round(float((t[:, 1, 5].max() - t[:, 1, 5].min()) / 100.0), 3)
    Lb.thallo_hip_debug_stamps_resident(None)
c = run({"THALLO_RESIDENT": "0"}, time_it=True)
out["default_launch_per_iteration_us"] = round(c[4], 2)
if R > 0:
    out["max_rel_cost_diff_vs_default"] = float(np.max(np.abs(np.array(a[0]) - np.array(c[0])) / np.abs(np.array(c[0]))))
print(json.dumps(out))
