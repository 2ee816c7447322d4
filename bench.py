#!/usr/bin/env python3
"""bench.py -- PCG iterations/s and ms per Gauss-Newton iteration, image_warping 2048^2 on MI355X.

A "step" = one Gauss-Newton iteration (one Thallo_ProblemStep: PCGInit + lIterations=100 PCG
iterations + PCGLinearUpdate) on one synthetic 2048x2048 image_warping instance already resident
in HBM (reference workload: examples/image_warping/src/main.cpp:131-149, 8 GN x 100 PCG).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 2048] [--liters 100]

Prints ONE JSON line (rank 0).  `value` = whole-job PCG iterations per second.
`roofline` = the dominant kernel (PCGIteration: one launch = a whole PCG iteration; since round 5 p_k goes into a ring of planes, 57.1 B/pixel, and the delta update is a
launch of its own next to the loop) -- its OWN duration (begin-to-end timestamps of two HIP events handed to the launch itself, hipExtLaunchKernelGGL) sampled at every
53rd launch of it INSIDE the timed region (`roofline.method_version` 2; version 1, rounds 1-4: events recorded around the launch, dispatch gap included, on 74.8 B/pixel --
kept as `avg_launch_ms_between_recorded_events`; the two versions' `frac` are not comparable).  `roofline.traffic` is NOT measured in this run
(`traffic_measured_this_run` false): it is the committed PMC figure of tools/profile.sh's passes over this same command.  The reference
formulation's 180 B/pixel figure is kept under `reference_formulation`; `roofline.applyjtj_standalone` = the plain applyJTJ kernel (48 B/pixel) timed
back-to-back after the timed region.  `--gpus N` (N > 1) without a launcher: this process starts the N ranks itself; a device-side transport that does not come up is
reported per rank on stderr and in `transport_fallback` (THALLO_DIST_TRANSPORT=device: exit 3 instead).  `cpu_baseline` = oracle/cpu_port_image_warping.c (OpenMP port
of the same algorithm; the reference ships no runnable CPU path) on the host cores, rank 0, N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy ceiling)
ALG_BYTES_APPLYJTJ = 48         # read p 12 + Angle 4 + UrShape 8 + Mask 4 + Constraints 8, write Ap_X 12
ALG_BYTES_FUSED_STEP1 = 96      # applyJTJ 48 + PCGStep3 (read z 12, p counted once, write p 12) + delta update (r/w 24)
ALG_BYTES_PCG_ITER = 180        # applyJTJ 48 + PCGStep2 96 + PCGStep3 36 (the reference's three-kernel formulation)
# What the one-kernel schedule has to move per pixel and PCG iteration, each array once.
# Stored-plane kernel (rounds 2-3; THALLO_MARCH=3): read r 12, Ap 12, p 12, cs 8, flags 1; write r 12, Ap 12, p 12; the deferred delta update (read delta 12 +
# p_{k-2} 12, write delta 12) every other iteration = 18  ->  99.
# Round 4 (energy_image_warping_march_rc.hip): no A p plane -- read r 12, p 12, cs 8, flags 1; write r 12, p 12 = 57 on odd iterations, 93 on even ones (the two
# deferred delta updates), and the FIRST iteration of a GN step on the stored-plane kernel without its A p read = 69: fused_bytes_per_iter(L) below (74.76 at L = 100).
FUSED_BYTES_PCG_ITER_STORED = 99


def fused_bytes_per_iter(L, ring=False):
    from thallo_amd.api import iw_fused_bytes_per_iter
    return iw_fused_bytes_per_iter(L, ring=ring)


FUSED_BYTES_STEP1 = 75          # two-kernel schedule (THALLO_AB=one_kernel=0, A/B): PCGStep1 = read z 12, p 12, cs 8, flags 1; write p 12, Ap 12 + the delta update every other launch 18
SAMPLE_PERIOD = 53             # HIP events at every 53rd launch of the dominant kernel INSIDE the timed region (handed to the launch itself -- the kernel's own begin / end
                               # timestamps -- and, for comparison, recorded around it): ~38 samples in 20 steps whose
                               # place in the GN step rotates (53 does not divide 100 or the delta update's period of 16: a launch next to that update takes ~8 us longer, and
                               # one fixed place per step measured 56 or 67 us depending on the place); every 16th launch cost 0.9 % of the rate, every 53rd ~0.3 %


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--liters", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small", action="store_true")              # skip the extra 512^2 / 2048x256 timings (resident PCG loop vs launch per iteration)
    ap.add_argument("--sample-period", type=int, default=-1)      # HIP events around every N-th launch of each kernel inside the timed region (default 53; 0: none, roofline from the loop's event pair)
    return ap.parse_args()


def standalone_applyjtj(torch, W, H, p_np, reps=50):
    """Time the plain applyJTJ kernel (48 B/px algorithmic) back-to-back with HIP events."""
    import thallo_amd
    from thallo_amd import api
    L = thallo_amd.lib()
    L.thallo_hip_vector_elems.restype = C.c_long; L.thallo_hip_vector_elems.argtypes = [C.c_long]
    N = W * H; n = 3 * N; na = L.thallo_hip_vector_elems(n)
    dev = [torch.from_numpy(x).cuda() if hasattr(x, "dtype") else x for x in p_np]
    f = lambda: torch.zeros(na, dtype=torch.float32, device="cuda")
    r, pre, z, p0, delta, Ap = f(), f(), f(), f(), f(), f()
    cs = torch.zeros(2 * N, dtype=torch.float32, device="cuda")
    flags = torch.zeros(N + 256, dtype=torch.uint8, device="cuda")
    parts = torch.zeros(4 * 1024, dtype=torch.float32, device="cuda")
    irregular = torch.zeros(16, dtype=torch.int32, device="cuda")
    vp, fl = C.c_void_p, C.c_float
    L.thallo_hip_iw_pcg_init(W, H, 0, H, vp(dev[0].data_ptr()), vp(dev[1].data_ptr()), vp(dev[2].data_ptr()), vp(dev[3].data_ptr()),
                             vp(dev[4].data_ptr()), fl(p_np[5]), fl(p_np[6]), vp(r.data_ptr()), vp(pre.data_ptr()), vp(z.data_ptr()),
                             vp(p0.data_ptr()), vp(delta.data_ptr()), vp(cs.data_ptr()), vp(flags.data_ptr()), None, vp(irregular.data_ptr()), vp(parts.data_ptr()), None)

    def launch():
        return L.thallo_hip_iw_apply_jtj(W, H, 0, H, vp(cs.data_ptr()), vp(dev[2].data_ptr()), vp(flags.data_ptr()), fl(p_np[5]), fl(p_np[6]),
                                         vp(z.data_ptr()), vp(Ap.data_ptr()), vp(irregular.data_ptr()), vp(parts.data_ptr() + 4096), None)
    for _ in range(5):
        assert launch() > 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # kernels run on the NULL stream = torch's current stream
    e0.record()
    for _ in range(reps):
        launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) before this process has made any GPU
    call, pass rank 0's JSON line through and exit with the first failure's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:          # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if int(env_world or "1") != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
    import numpy as np
    import torch
    import torch.distributed as dist
    import thallo_amd
    from thallo_amd import synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("THALLO_DIST_BACKEND", "nccl")
    if os.environ.get("THALLO_BENCH_DRY") == "1":
        # plumbing check without a GPU (tests/test_bench_spawn.py): the ranks rendezvous, agree on the world size and rank 0 prints a line
        dist.init_process_group("gloo", rank=rank, world_size=world) if world > 1 else None
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
            assert dist.get_world_size() == args.gpus
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": int(t.item()), "steps": args.steps, "warmup": args.warmup}))
        if world > 1:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        # (RCCL refuses two ranks on one device; never fall back to a silent 1-GPU run)
        sys.exit(f"bench.py: --gpus {world} needs {world} GPUs on this node, found {ndev}")
    torch.cuda.set_device(local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  THALLO_DIST_BACKEND=gloo exists only to exercise this leg on a 1-GPU box; it is never the
        # measured configuration.
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    W = H = args.size
    L_it = args.liters
    K, Wm = args.steps, args.warmup
    p = syn.image_warping(W, H)

    if world > 1:
        from thallo_amd import distributed as tdist
        res = tdist.bench_image_warping(p, W, H, L_it, K, Wm, rank, world)
        if rank == 0:
            print(json.dumps(res))
        dist.destroy_process_group()
        return

    _L = thallo_amd.lib()      # tools knobs of the persistent marching loop (tools/persist_ab.sh); defaults are the product settings
    if "THALLO_PERSIST_ACQ" in os.environ: _L.thallo_hip_iw_march_persist_debug_set(0, int(os.environ["THALLO_PERSIST_ACQ"]))
    if "THALLO_PERSIST_RES" in os.environ: _L.thallo_hip_iw_march_persist_debug_set(1, int(os.environ["THALLO_PERSIST_RES"]))
    if "THALLO_PERSIST_OCC" in os.environ: _L.thallo_hip_iw_march_persist_debug_set(2, int(os.environ["THALLO_PERSIST_OCC"]))
    dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"), timing_level=1)      # (1: the coarse events -- eight per GN step of L PCG iterations -- that performance_summary() below reads; level 0 records nothing since round 6)
    s.set_solver_parameters(nIterations=K + Wm, lIterations=L_it)
    params = s.make_params(dev)
    s.init(params)
    for _ in range(Wm):
        s.step(params)
    torch.cuda.synchronize()
    # one kernel per PCG iteration (thallo_hip_iw_pcg_iter, the default) vs PCGStep1 + PCGStep2 (THALLO_AB=one_kernel=0, A/B)
    one_kernel = "one_kernel=0" not in os.environ.get("THALLO_AB", "")
    s.reset_kernel_stats()
    # the dominant kernel's own launch duration: HIP events around every 53rd launch, live inside the timed region (round 5: the PCG loop of a GN step is no longer
    # L launches of one kernel and nothing else -- the delta updates of the ring of p planes run next to it on a second stream -- so the loop's event pair / L is
    # the loop's figure, `pcg_loop_ms_per_iteration`, not the kernel's)
    s.set_kernel_sampling(args.sample_period if args.sample_period >= 0 else SAMPLE_PERIOD)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        assert s.step(params) == 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.set_kernel_sampling(0)
    ks = s.kernel_stats()
    final_cost = s.current_cost()
    ring = "PCGDeltaUpdate" in ks
    persistent = "PCGLoopPersistent" in ks
    assert s.step(params) == 0                 # budget used up: finalises the plan and its performance summary
    perf = s.performance_summary()

    npx = W * H
    dom = "PCGIteration" if one_kernel else "PCGStep1"
    dom_bytes = fused_bytes_per_iter(L_it, ring) if one_kernel else FUSED_BYTES_STEP1          # what the kernel has to move (= its PMC traffic)
    ref_bytes = ALG_BYTES_PCG_ITER if one_kernel else ALG_BYTES_FUSED_STEP1        # the reference formulation of the same work
    timing = "HIP events on the kernel's stream around every 53rd launch of it, inside the timed region"
    if persistent:      # the loop is a few launches of many iterations each: the library's event pair around it ("Linear Solve") / L, delta updates included
        step1_ms = perf["linearSolve"]["meanMS"] / L_it
        n_samples = perf["linearSolve"]["count"] * L_it
    elif ks.get(dom, {}).get("own_samples"):      # the sampled launches carried their own start / stop events: the kernel's begin-to-end time (what rocprofv3 calls its duration)
        step1_ms = ks[dom]["own_mean_ms"]
        n_samples = ks[dom]["own_samples"]
        timing = ("HIP events handed to the sampled launches themselves (hipExtLaunchKernelGGL start / stop: the kernel's own begin and end timestamps), every "
                  "53rd launch of it on its stream, inside the timed region")
    elif ks.get(dom, {}).get("samples"):
        step1_ms = ks[dom]["mean_ms"]
        n_samples = ks[dom]["samples"]
    else:               # --sample-period 0: the loop's figure stands in
        step1_ms = perf["linearSolve"]["meanMS"] / L_it
        n_samples = perf["linearSolve"]["count"] * L_it
    ach = dom_bytes * npx / (step1_ms * 1e-3) / 1e9
    # the loop's bytes per iteration: the marching launch + 12 per pending term + delta read and written once per PCGDeltaUpdate launch
    sched_bytes = dom_bytes + 12.0 + 24.0 * (ks.get("PCGDeltaUpdate", {}).get("launches", 0) / K) / L_it
    sa_ms = standalone_applyjtj(torch, W, H, p)
    sa_gbs = ALG_BYTES_APPLYJTJ * npx / (sa_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    for rel in (os.path.join("profiles", "traffic_latest.json"),):
        tf = os.path.join(ROOT, rel)
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(dom + "_bytes_per_launch")
                traffic_source = rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/profile.sh; not measured in this run)"
            except Exception:
                traffic = None

    out = {
        "metric": "pcg_iters_per_sec", "value": K * L_it / dt, "unit": "PCG iterations/s",
        "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"examples/image_warping {W}x{H} ARAP, GN + matrix-free PCG, {L_it} PCG iterations per GN step",
                   "width": W, "height": H, "unknowns": 3 * npx, "l_iterations": L_it, "parallelism": "1 GPU"},
        "ms_per_gn_iter": dt / K * 1e3, "us_per_pcg_iter": dt / (K * L_it) * 1e6,
        "final_cost": final_cost,
        "roofline": {"bound": "hbm",
                     "kernel": ("PCGIteration (one launch = PCGStep2 of iteration k-1 + PCGStep3 + applyJTJ of iteration k; p_k into a ring of planes)" if one_kernel and ring
                                else "PCGIteration (one launch = PCGStep2 of iteration k-1 + PCGStep3 + delta update + applyJTJ of iteration k)"
                                if one_kernel else "PCGStep1 (fused PCGStep3 + delta update + applyJTJ)"),
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "traffic_measured_this_run": False,
                     "method_version": 2,      # 2 (round 5 on): the kernel's own begin-to-end duration, on the kernel's own byte count; 1: events recorded around the launch
                     "bytes_per_pixel": dom_bytes, "pixels_per_launch": npx,
                     "avg_launch_ms": step1_ms, "samples": n_samples,
                     "timing": timing,
                     # the same sampled launches between two events recorded around them: kernel + the dispatch gap in front of it
                     "avg_launch_ms_between_recorded_events": ks.get(dom, {}).get("mean_ms"),
                     "pcg_loop_ms_per_iteration": perf["linearSolve"]["meanMS"] / L_it,
                     # the whole schedule's HBM rate: the loop's bytes per iteration (the marching launch + the delta update's 12 per term + 24 per launch) over
                     # the loop's time per iteration -- information next to the contract's per-kernel figure above
                     "schedule": ({"bytes_per_pixel_per_iteration": sched_bytes, "achieved": sched_bytes * npx / (perf["linearSolve"]["meanMS"] / L_it * 1e-3) / 1e9,
                                   "frac": sched_bytes * npx / (perf["linearSolve"]["meanMS"] / L_it * 1e-3) / 1e9 / HBM_PEAK_GBS} if ring and one_kernel and not persistent else None),
                     "note": "achieved = bytes_per_pixel x pixels / avg launch time: the bytes this fused kernel has to move, each array once "
                             "(DESIGN.md section 4).  Round 4 removed the A p plane from the iteration (99 -> 74.8 B/pixel); round 5 took the delta update out of "
                             "it (-> 57.1 B/pixel: p_k goes into a ring of planes and delta takes 32 of them per PCGDeltaUpdate launch, 12.75 B/pixel/iteration "
                             "next to the loop).  frac is quoted on the kernel's OWN, smaller byte count: a lower frac at a higher PCG rate is a faster "
                             "schedule, not a slower kernel",
                     "delta_update": ({"kernel": "PCGDeltaUpdate (thallo_hip_linear_update_n)", "launches_per_gn_step": ks["PCGDeltaUpdate"]["launches"] / K,
                                       "avg_launch_ms": ks["PCGDeltaUpdate"]["mean_ms"]} if ring and one_kernel and ks.get("PCGDeltaUpdate", {}).get("samples") else None),
                     # the same launch priced with SURVEY.md 8d's bytes of the reference's three-kernel formulation -- a speed-up figure, not a
                     # roofline fraction (it exceeds the HBM peak because the schedule removes 45 % of those bytes)
                     "reference_formulation": {"bytes_per_pixel": ref_bytes, "equivalent_GBps": ref_bytes * npx / (step1_ms * 1e-3) / 1e9},
                     "applyjtj_standalone": {"algorithmic_bytes_per_pixel": ALG_BYTES_APPLYJTJ, "avg_launch_ms": sa_ms,
                                             "achieved": sa_gbs, "frac": sa_gbs / HBM_PEAK_GBS}},
    }
    if not one_kernel:      # z-free two-kernel schedule: read r 12, Ap 12, flags 1; write r 12
        step2_ms = ks["PCGStep2"]["mean_ms"]
        out["roofline"]["pcg_step2"] = {"algorithmic_bytes_per_pixel": 37, "avg_launch_ms": step2_ms, "achieved": 37 * npx / (step2_ms * 1e-3) / 1e9}
    # BASELINE config 1's size and one rank's slab of the 8-GPU run: working sets that fit the chip's registers run the whole PCG loop of a GN step in ONE
    # launch (thallo_hip_iw_pcg_resident); the launch-per-iteration schedule of the same plan beside it (THALLO_RESIDENT=0).  Not the headline metric: extra keys.
    if not args.no_small:
        def small(w, h, resident):
            os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
            try:
                q = syn.image_warping(w, h)
                d2 = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in q]
                s2 = thallo_amd.ThalloSolver((w, h), thallo_amd.energy_file("image_warping"), timing_level=0)
                s2.set_solver_parameters(nIterations=1 << 30, lIterations=L_it)
                p2 = s2.make_params(d2)
                s2.init(p2)
                for _ in range(3):
                    s2.step(p2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    s2.step(p2)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t1) / (20 * L_it) * 1e6
                names = sorted(s2.kernel_stats())
                c = s2.current_cost()
                s2.close()
                assert c == c, thallo_amd.last_error()
                return {"us_per_pcg_iter": us, "kernels": names}
            finally:
                os.environ.pop("THALLO_RESIDENT", None)
        out["small_working_sets"] = {f"{w}x{h}": {"resident_loop": small(w, h, True), "launch_per_iteration": small(w, h, False)} for (w, h) in ((512, 512), (2048, 256), (2048, 512))}
        # (2048 x 512, the 1/4 slab of a 4-GPU run: 9 rows per wave, round 4.)  The 1/2 slab of a 2-GPU run has more rows per wave than the resident kernel's
        # registers hold (R = 18 > 10): one marching launch per iteration
        out["small_working_sets"]["2048x1024"] = {"resident_loop": None, "launch_per_iteration": small(2048, 1024, False)}
        # round 6: shape_from_shading at the size of the reference's data set (640 x 480, 10 PCG iterations per step as its example runs): the PCG loop of a GN step / a whole
        # LM step's loop + model cost + update in one resident launch, against one launch per PCG iteration (THALLO_RESIDENT=0)
        def small_sfs(resident, lm):
            os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
            try:
                q = syn.shape_from_shading(640, 480)
                d2 = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else float(x) for x in q]
                s2 = thallo_amd.ThalloSolver((640, 480), thallo_amd.energy_file("shape_from_shading"), timing_level=0, **({"solverkind": "levenberg_marquardt"} if lm else {}))
                if lm: s2.enable_lm()
                s2.set_solver_parameters(nIterations=1 << 30, lIterations=10, **({"q_tolerance": 0.0} if lm else {}))
                p2 = s2.make_params(d2)
                s2.init(p2)
                for _ in range(3):
                    s2.step(p2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    s2.step(p2)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t1) / (20 * 10) * 1e6
                names = sorted(s2.kernel_stats())
                c = s2.current_cost()
                s2.close()
                assert c == c, thallo_amd.last_error()
                return {"us_per_pcg_iter": us, "kernels": names}
            finally:
                os.environ.pop("THALLO_RESIDENT", None)
        out["small_working_sets"]["shape_from_shading_640x480"] = {"gn": {"resident_loop": small_sfs(True, False), "launch_per_iteration": small_sfs(False, False)},
                                                                    "lm": {"resident_loop": small_sfs(True, True), "launch_per_iteration": small_sfs(False, True)}}
    if not args.no_cpu_baseline:
        from oracle import oracle as orc
        q = syn.image_warping(W, H)
        sample_l = L_it
        res = orc.cpu_port_image_warping(W, H, q, 1, sample_l, want_costs=False)
        out["cpu_baseline"] = {"value": sample_l / res["seconds_pcg"], "unit": "PCG iterations/s", "cores": res["threads"],
                               "kind": "port", "ms_per_gn_iter": res["seconds_total"] * 1e3,
                               "placement": ("one thread pinned per CPU (sched_setaffinity), vectors first-touched by their threads" if res.get("pinned")
                                             else "threads not pinned (sched_setaffinity refused)"),
                               "note": "baseline only (an OpenMP restatement, not a tuned CPU solver); the GPU / CPU ratio says nothing about kernel quality",
                               "sample": f"1 GN step x {sample_l} PCG iterations of the same {W}x{H} instance, OpenMP port of the "
                                         "reference algorithm (oracle/cpu_port_image_warping.c)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
