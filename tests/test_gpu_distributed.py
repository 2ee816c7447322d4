"""GPU: the HIP slab backend of the multi-GPU driver.  The box has ONE GPU, so the ranks share cuda:0 and
talk over gloo (RCCL refuses two ranks on one device); the compute path is the real one -- row-range
kernels, ghost-row maintenance, pack/unpack -- only the transport differs from the 8-GPU run."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _collect(q, procs, world, limit=150.0):
    """Gather one result per rank; fail fast if a rank died instead of waiting out the queue timeout."""
    import queue as _queue
    import time as _time
    res, t0 = [], _time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=1.0))
        except _queue.Empty:
            dead = [p_.exitcode for p_ in procs if p_.exitcode not in (None, 0)]
            if dead or _time.time() - t0 > limit:
                for p_ in procs:
                    if p_.is_alive():
                        p_.terminate()
                raise AssertionError(f"ranks failed: exit codes {[p_.exitcode for p_ in procs]}")
    for p_ in procs:
        p_.join(timeout=30)
        assert p_.exitcode == 0
    return res


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, W, H, nit, lit, q, device_exchange=False):
    """One rank of a slab run through the LIBRARY (ThalloX_PlanSetDistributed + Thallo_ProblemInit / Step / CurrentCost)."""
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed import PlanSlabSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.image_warping(W, H, n_markers=8)
        solver = PlanSlabSolver(p, W, H, rank, world, lit, device_exchange=device_exchange)
        costs = solver.solve(nit, lit)
        lay = solver.lay
        info = solver.info
        err = solver.solver.distributed_error() if device_exchange else 0
        trace = solver.solver.alpha_beta_trace()
        off, ang = solver.owned()
        q.put((rank, costs, lay.g0, lay.g1, off, ang, info, err, trace, solver.solver.kernel_stats()))
        solver.solver.close()
    finally:
        dist.destroy_process_group()


def _run(world, W, H, nit, lit, device_exchange):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, nit, lit, q, device_exchange)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = _collect(q, procs, world)
    res.sort(key=lambda t: t[0])
    return res


@pytest.mark.parametrize("kernel", ["auto", "march"])
@pytest.mark.parametrize("world,W,H,nit,lit", [(2, 128, 96, 3, 30), (3, 64, 100, 2, 20), (1, 64, 48, 2, 10)])
def test_hip_slabs_p2p_mailbox_exchange(orc, monkeypatch, world, W, H, nit, lit, kernel):
    """The device-side exchange (mailbox granules + peer-to-peer ghost rows, csrc/dist_device.hpp) between `world` processes --
    here all on GPU 0, mapped through hipIpc like real peers -- behind Thallo_ProblemStep: it must enable itself (self-check against the
    all-gather path at the first Init), never time out, give every rank bit-identical alpha/beta, and follow the oracle's cost trajectory."""
    from thallo_amd import synthetic as syn
    if kernel == "march":
        monkeypatch.setenv("THALLO_MARCH", "2")       # the marching kernel's multi-GPU variant at sizes where the plugin would pick the tile kernel (the ranks inherit the environment)
    res = _run(world, W, H, nit, lit, True)
    p = syn.image_warping(W, H, n_markers=8)
    co, _ = orc.Problem(orc.IMAGE_WARPING, (W, H), p).solve(nIterations=nit, lIterations=lit)
    for rank, costs, g0, g1, off, ang, info, err, trace, _stats in res:
        assert info["exchange"] == "p2p-mailbox" and info["self_check"]["all_ranks_pass"], (rank, info)
        assert info["memory"] == ["fine-grained", "fine-grained"], info
        assert err == 0, (rank, info)
        assert np.abs(np.array(costs) - co).max() <= 1e-5 * np.abs(co).max(), (rank, costs, co)
        assert costs == res[0][1]
        assert trace == res[0][8] and len(trace) == lit          # rank-ordered sums: identical bits on every rank


@pytest.mark.parametrize("world,W,H,nit,lit", [(2, 128, 96, 3, 30), (3, 64, 100, 2, 20), (2, 256, 256, 2, 40)])
def test_hip_slabs_match_oracle(orc, world, W, H, nit, lit):
    """The all-gather transport of the same schedule (what runs when the device-side exchange is not requested or its self-check fails)."""
    from thallo_amd import synthetic as syn
    res = _run(world, W, H, nit, lit, False)
    p = syn.image_warping(W, H, n_markers=8)
    co, _ = orc.Problem(orc.IMAGE_WARPING, (W, H), p).solve(nIterations=nit, lIterations=lit)
    for rank, costs, g0, g1, off, ang, info, err, trace, _stats in res:
        assert info["exchange"] == "allgather", info
        assert np.abs(np.array(costs) - co).max() <= 1e-5 * np.abs(co).max(), (rank, costs, co)
        assert costs == res[0][1] and trace == res[0][8]
        assert np.abs(off - p[0][g0:g1]).max() <= 2e-4 * np.abs(p[0]).max()
        assert np.abs(ang - p[1][g0:g1]).max() <= 2e-4 * max(1.0, np.abs(p[1]).max())


@pytest.mark.parametrize("kernel", ["auto", "march"])
def test_hip_slab_transports_agree_bitwise_on_the_unknowns(orc, monkeypatch, kernel):
    """Both transports add the per-rank sums in rank order from the same per-rank values: the same alpha / beta bits and the same unknowns, although the
    device-side transport updates delta every iteration and the all-gather transport every other one.  Round 3: also for the marching kernel -- its
    multi-GPU template variant (peer stores in the row loop) used to be contracted into fused multiply-adds differently from its single-GPU variant
    (-ffp-contract=fast decides by context); the image_warping kernels are now built with -ffp-contract=on (per source expression), so equal expressions
    give equal bits in every instantiation.  (One launch per PCG iteration on both transports: THALLO_RESIDENT=0; the resident loop has its own test below.)"""
    monkeypatch.setenv("THALLO_RESIDENT", "0")
    if kernel == "march":
        monkeypatch.setenv("THALLO_MARCH", "2")
    a = _run(2, 128, 64, 2, 12, True)
    b = _run(2, 128, 64, 2, 12, False)
    for ra, rb in zip(a, b):
        assert ra[6]["exchange"] == "p2p-mailbox" and rb[6]["exchange"] == "allgather"
        assert ra[8] == rb[8] and ra[1] == rb[1]
        assert (ra[4] == rb[4]).all() and (ra[5] == rb[5]).all()


def _worker_rows(rank, world, port, W, H, nit, lit, q, resident, rows, cap=0):
    """_worker on the device-side transport with the PCG loop either resident (one launch per GN step) or one marching launch per iteration with `rows` rows per segment;
    cap > 0: the resident kernel's workgroup budget per rank (so that the ranks sharing GPU 0 here ARE co-resident at the rows per wave of a real multi-GPU slab)"""
    import torch  # noqa: F401  (before libThallo.so: the HIP runtime torch ships must be the one that gets loaded)
    import thallo_amd
    os.environ["THALLO_RESIDENT"] = "1" if resident else "0"
    os.environ["THALLO_MARCH"] = "2"
    thallo_amd.lib().thallo_hip_march_debug_set(0, 0 if resident else rows)
    if cap > 0:
        thallo_amd.lib().thallo_hip_resident_debug_set(1, cap)
    _worker(rank, world, port, W, H, nit, lit, q, True)


@pytest.mark.parametrize("world,W,H,lit,cap", [(2, 128, 96, 30, 0), (3, 252, 90, 12, 0), (1, 2048, 256, 20, 0), (2, 640, 240, 16, 0), (1, 2048, 512, 12, 0), (2, 1024, 900, 10, 112)])
def test_resident_slab_loop_is_bitwise_the_launch_per_iteration_transport(monkeypatch, world, W, H, lit, cap):
    """VERDICT r2 item 1, multi-GPU half: on the device-side transport a rank's whole PCG loop is ONE launch (thallo_hip_iw_pcg_resident_dist): state in registers,
    the first / last owned row of A p straight into the neighbouring ranks' ghost areas, workgroup 0 exchanges the rank's sums through the same mailbox slots and
    publishes the two global words.  Same granules, same rank order, same per-rank summation order as one marching launch per iteration with the same rows per
    segment: costs, alpha_k / beta_k and the owned unknowns are bit-identical on every rank (ranks share GPU 0 here; 2048 x 256 / 2048 x 512 = one rank's slab of the
    8- / 4-GPU run; the last case: two ranks with a budget of 112 workgroups each, i.e. co-resident on one GPU at the 10 rows per wave of a 4-GPU slab -- VERDICT r3 item 2)."""
    import ctypes as C
    import torch  # noqa: F401
    import torch.multiprocessing as mp
    import thallo_amd
    from thallo_amd.distributed import SlabLayout, image_warping_slab_counts
    L = thallo_amd.lib()
    L.thallo_hip_iw_resident_rows_slab.restype = C.c_int
    rows = None
    monkeypatch.setenv("THALLO_MARCH", "2")         # (as the workers: the split PlanSlabSolver picks depends on which kernels the plugin will run)
    if cap > 0:
        L.thallo_hip_resident_debug_set(1, cap)
    try:
        counts = image_warping_slab_counts(W, H, world)
        for r in range(world):
            lay = SlabLayout(H, r, world, counts=counts)
            rr = L.thallo_hip_iw_resident_rows_slab(W, lay.row1 - lay.row0, 1 if r < world - 1 else 0)
            assert 1 <= rr <= 10, (r, rr)
            rows = rr if rows is None else rows
            assert rr == rows, "this test forces ONE rows-per-segment on the marching kernel of every rank"
    finally:
        L.thallo_hip_resident_debug_set(1, 0)
    if cap > 0 or (W, H) == (2048, 512):
        assert rows >= 9, rows
    out = []
    for resident in (True, False):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_rows, args=(r, world, port, W, H, 2, lit, q, resident, rows, cap)) for r in range(world)]
        for p_ in procs:
            p_.start()
        out.append(sorted(_collect(q, procs, world), key=lambda t: t[0]))
    for ra, rb in zip(*out):
        assert ra[6]["exchange"] == "p2p-mailbox" and rb[6]["exchange"] == "p2p-mailbox" and ra[7] == 0 and rb[7] == 0
        assert np.isfinite(ra[1]).all() and len(ra[8]) == lit
        assert "PCGLoopResident" in ra[9] and "PCGLoopResident" not in rb[9] and rb[9]["PCGIteration"]["launches"] >= 2 * lit, (ra[9].keys(), rb[9].keys())
        assert ra[8] == rb[8], [k for k, (x, y) in enumerate(zip(ra[8], rb[8])) if x != y][:3]
        assert ra[1] == rb[1]
        assert (ra[4] == rb[4]).all() and (ra[5] == rb[5]).all()


def test_hip_single_slab_equals_library_path(orc):
    """world_size 1 declared as a slab == the plain single-device plan on the same instance (same kernels, same one-kernel schedule; the slab
    form adds its scalars through the rank-ordered finish instead of the kernel's last workgroup: same order, same bits)."""
    import torch
    import thallo_amd
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed import PlanSlabSolver
    W, H = 192, 80
    p = syn.image_warping(W, H, n_markers=8)
    solver = PlanSlabSolver(p, W, H, 0, 1, 25, device_exchange=False)
    costs = solver.solve(2, 25)
    dev = [torch.from_numpy(x.copy()).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("image_warping"))
    _, c2 = s.solve(dev, profiled=True, nIterations=2, lIterations=25)
    assert np.abs(np.array(costs) - np.array(c2)).max() <= 1e-6 * max(c2)
    assert (solver.offset.view(-1) - dev[0].view(-1)).abs().max().item() <= 1e-5 * dev[0].abs().max().item()


def test_distributed_plan_rejects_what_it_cannot_run():
    """Loud failures: a non-slab energy, a slab without its ghost row, a bad rank, an irregular UrShape."""
    import torch
    import thallo_amd
    from thallo_amd import api, synthetic as syn
    s = thallo_amd.ThalloSolver((16, 8), thallo_amd.energy_file("laplacian_image"))
    with pytest.raises(RuntimeError, match="no row-slab form"):
        s.set_distributed(0, 1, 0, 8)
    s = thallo_amd.ThalloSolver((64, 16), thallo_amd.energy_file("image_warping"))
    with pytest.raises(RuntimeError, match="ghost row"):
        s.set_distributed(0, 2, 0, 16, allgather=lambda *a: None)      # rank 0 of 2 must keep one ghost row below
    with pytest.raises(RuntimeError, match="rank"):
        s.set_distributed(3, 2, 0, 16)
    W, H = 64, 16
    p = syn.image_warping(W, H, n_markers=4)
    p[2] = p[2] * 1.5                                                  # UrShape off the unit grid
    dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() if isinstance(x, np.ndarray) else float(x) for x in p]
    s.set_distributed(0, 1, 0, 16)
    s.init(s.make_params(dev))
    assert not s.ready() and "unit pixel grid" in api.last_error()


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """bench.py's N > 1 leg end to end on this 1-GPU box: `--gpus 2` spawns the ranks, THALLO_DIST_BACKEND=gloo lets them share cuda:0 (RCCL refuses
    that; never a measured configuration), the slab schedule runs behind Thallo_ProblemStep with the device-side exchange, rank 0 prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, THALLO_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "256", "--liters", "20"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["metric"] == "pcg_iters_per_sec" and d["value"] > 0
    assert d["exchange"] == "p2p-mailbox" and d["p2p_check"]["self_check"]["all_ranks_pass"]
    assert d["final_cost"] < d["initial_cost"]
    assert 0 < d["roofline"]["frac"] < 1


# ------------------------------------------------------------------ camera-sharded bundle adjustment (HIP backend)
def _ba_worker(rank, world, port, dims, nit, lit, q, device_exchange=True, lm=False, sp=None):
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed_ba import PlanBaShardSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C_, P_, O_ = dims
        p = syn.bundle_adjustment(C=C_, P=P_, O=O_, band=8)
        solver = PlanBaShardSolver(p, rank, world, lit, device_exchange=device_exchange, lm=lm)
        costs = solver.solve(nit, **(sp or {}))
        lay = solver.lay
        q.put((rank, costs, lay.c0, lay.c1, solver.cameras[:lay.C_loc].cpu().numpy(), solver.points[:lay.P].cpu().numpy(), solver.solver.distributed_info()))
        solver.solver.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,nit,lit", [(2, (64, 4000, 20000), 3, 30), (3, (13, 81, 400), 3, 10), (1, (12, 60, 300), 3, 20)])      # (81 points: the padded point block)
def test_hip_ba_camera_shards_match_oracle(orc, world, dims, nit, lit):
    """Camera shards behind Thallo_ProblemStep (csrc/solver_dist.cpp, shard form): all-reduce of the point block of A p + one tiny all-gather per PCG
    iteration; the replicated points stay bit-identical across ranks."""
    import torch.multiprocessing as mp
    from thallo_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    C_, P_, O_ = dims
    p = syn.bundle_adjustment(C=C_, P=P_, O=O_, band=8)
    co, _ = orc.Problem(orc.BUNDLE_ADJUST, dims, p).solve(nIterations=nit, lIterations=lit)
    runs = {}
    for dx in (True, False):        # round 3: the point block's all-reduce by peer stores (one launch, sums in rank order) / through the all-reduce callback
        port = _free_port()
        procs = [ctx.Process(target=_ba_worker, args=(r, world, port, dims, nit, lit, q, dx)) for r in range(world)]
        for p_ in procs:
            p_.start()
        res = _collect(q, procs, world)
        res.sort(key=lambda t: t[0])
        for rank, costs, c0, c1, cams, pts, info in res:
            assert info["exchange"] == ("p2p-allreduce + allgather" if dx else "allreduce + allgather"), info
            assert not dx or info["self_check"]["all_ranks_pass"] is True
            assert np.abs(np.array(costs) - co).max() <= 3e-5 * np.abs(co).max(), (rank, costs, co)
            assert costs == res[0][1]
            assert np.array_equal(pts, res[0][5])
            assert np.abs(pts - p[1]).max() <= 2e-3 * np.abs(p[1]).max()
        runs[dx] = res
    a, b = np.array(runs[True][0][1]), np.array(runs[False][0][1])
    assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), (a, b)      # the two all-reduces add the ranks' parts in different orders: equal to rounding


@pytest.mark.parametrize("world,dims,nit,lit,sp", [(2, (64, 4000, 20000), 4, 25, {}), (3, (13, 81, 400), 6, 12, {"min_relative_decrease": 0.99}), (2, (24, 400, 2400), 4, 40, {"q_tolerance": 0.02}),
                                                   (1, (12, 60, 300), 3, 20, {})])
def test_hip_ba_camera_shards_levenberg_marquardt_match_oracle(orc, world, dims, nit, lit, sp):
    """Round 6 (VERDICT r5 Missing 3; BASELINE config 4 is camera-sharded and the reference's example runs LM 5 x 150): the LM branch on camera shards behind
    Thallo_ProblemStep (csrc/solver_dist.cpp step_lm_shard), ranks sharing the one GPU over gloo: against the oracle's LM trajectory of the whole problem -- across a
    residual reset (lIterations 25 / 40 > residual_reset_period 10), rejected steps (min_relative_decrease raised) and the zeta test's early exit -- the same costs on every rank,
    the replicated points bit-identical across ranks after accepted and reverted steps, both all-reduce transports."""
    import torch.multiprocessing as mp
    from thallo_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    C_, P_, O_ = dims
    p = syn.bundle_adjustment(C=C_, P=P_, O=O_, band=8)
    co, _ = orc.Problem(orc.BUNDLE_ADJUST, dims, [a.copy() if hasattr(a, "copy") else a for a in p]).solve(nIterations=nit, lIterations=lit, use_lm=1, **sp)
    runs = {}
    for dx in (True, False):
        port = _free_port()
        procs = [ctx.Process(target=_ba_worker, args=(r, world, port, dims, nit, lit, q, dx, True, sp)) for r in range(world)]
        for p_ in procs:
            p_.start()
        res = _collect(q, procs, world)
        res.sort(key=lambda t: t[0])
        for rank, costs, c0, c1, cams, pts, info in res:
            m = min(len(costs), len(co))
            assert m >= 3 and np.abs(np.array(costs[:m]) - co[:m]).max() <= 3e-4 * np.abs(co).max(), (rank, costs, co)
            assert costs == res[0][1]
            assert np.array_equal(pts, res[0][5])
        runs[dx] = res
    a, b = np.array(runs[True][0][1]), np.array(runs[False][0][1])
    assert len(a) == len(b) and np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), (a, b)
    assert any(a[i + 1] < a[i] for i in range(len(a) - 1))
    if "min_relative_decrease" in sp:
        assert any(a[i + 1] == a[i] for i in range(len(a) - 1)), a          # a rejected step (the bar on the step quality raised): the unknowns came back, on every rank


# ------------------------------------------------------------------ vertex-partitioned ARAP (HIP backend)
def _arap_worker(rank, world, port, nu, nv, nit, lit, q):
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed_graph import PlanArapSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.arap_mesh(nu, nv, n_handles=8, angle_amp=0.3)
        solver = PlanArapSolver(p, rank, world, lit)
        costs = solver.solve(nit)
        q.put((rank, costs, solver.position.cpu().numpy(), solver.angle.cpu().numpy(), solver.solver.distributed_info()))
        solver.solver.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nu,nv,nit,lit", [(2, 40, 30, 3, 40), (3, 24, 16, 3, 20), (1, 12, 8, 2, 10)])
def test_hip_arap_vertex_partition_matches_oracle(orc, world, nu, nv, nit, lit):
    import torch.multiprocessing as mp
    from thallo_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arap_worker, args=(r, world, port, nu, nv, nit, lit, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = _collect(q, procs, world)
    p = syn.arap_mesh(nu, nv, n_handles=8, angle_amp=0.3)
    co, _ = orc.Problem(orc.ARAP_MESH, (p[2].shape[0], p[6].shape[0]), p).solve(nIterations=nit, lIterations=lit)
    res.sort(key=lambda t: t[0])
    for rank, costs, pos, ang, info in res:
        assert info["exchange"] == "allgather" and "unit ranges" in info["form"]
        assert np.abs(np.array(costs) - co).max() <= 1e-5 * np.abs(co).max(), (rank, costs, co)
        assert costs == res[0][1] and np.array_equal(pos, res[0][2]) and np.array_equal(ang, res[0][3])      # the unknowns stay replicated bit for bit
        assert np.abs(pos - p[2]).max() <= 2e-4 * np.abs(p[2]).max()


def _arap_part_worker(rank, world, port, nu, nv, nit, lit, q, device_exchange=True):
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed_graph import PlanArapPartitionSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.arap_mesh(nu, nv, n_handles=8, angle_amp=0.3)
        solver = PlanArapPartitionSolver(p, rank, world, lit, device_exchange=device_exchange)
        costs = solver.solve(nit)
        part = solver.part
        q.put((rank, costs, part.owned_global, part.local_global, solver.owned(), solver.ghosts(), solver.solver.distributed_info()))
        solver.solver.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nu,nv,nit,lit", [(2, 40, 30, 3, 40), (3, 24, 17, 3, 20), (1, 12, 8, 2, 10)])
def test_hip_arap_real_vertex_partition_matches_oracle(orc, world, nu, nv, nit, lit):
    """VERDICT r2 item 8 / SURVEY 8(e) row 2: a REAL vertex partition -- every rank's Plan is its local sub-mesh (owned + ghost vertices, the edges with an owned end), the
    vectors are local-sized, and per PCG iteration only A p at the boundary vertices travels (ThalloX_PlanSetGhostExchange).  Against the single-domain oracle; the
    ghosts' unknowns end equal to their owners' bit for bit without ever being exchanged (same arithmetic on the same bits); ragged ranges (24 x 17 vertices on 3 ranks)."""
    import torch.multiprocessing as mp
    from thallo_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    runs = []
    for dx in (True, False):        # the boundary exchange as one launch of peer stores (thallo_hip_dist_xunits) / as pack + all-gather + unpack
        port = _free_port()
        procs = [ctx.Process(target=_arap_part_worker, args=(r, world, port, nu, nv, nit, lit, q, dx)) for r in range(world)]
        for p_ in procs:
            p_.start()
        r_ = _collect(q, procs, world)
        r_.sort(key=lambda t: t[0])
        for t in r_:
            assert t[6]["exchange"] == ("p2p-units" if dx else "allgather"), t[6]
            assert not dx or t[6]["self_check"]["all_ranks_pass"] is True
        runs.append(r_)
    for a, b in zip(*runs):         # same arithmetic in the same order: the two transports agree bit for bit
        assert a[1] == b[1] and np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1]), a[0]
    res = runs[0]
    p = syn.arap_mesh(nu, nv, n_handles=8, angle_amp=0.3)
    N = p[2].shape[0]
    co, _ = orc.Problem(orc.ARAP_MESH, (N, p[6].shape[0]), p).solve(nIterations=nit, lIterations=lit)
    pos, ang = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
    for rank, costs, owned_g, local_g, (po, ao), _, info in res:
        assert "unit partition" in info["form"] and info["world"] == world, info
        if world > 1:
            assert len(local_g) < N                                    # the plan really is smaller than the mesh
        assert np.abs(np.array(costs) - co).max() <= 1e-5 * np.abs(co).max(), (rank, costs, co)
        assert costs == res[0][1]
        pos[owned_g] = po; ang[owned_g] = ao
    assert not np.isnan(pos).any()
    assert np.abs(pos - p[2]).max() <= 2e-4 * np.abs(p[2]).max() and np.abs(ang - p[3]).max() <= 2e-4 * max(1.0, np.abs(p[3]).max())
    for rank, costs, owned_g, local_g, _, (pg, agh), info in res:      # ghosts == their owners' values, bit for bit
        gh = local_g[len(owned_g):]
        assert np.array_equal(pg, pos[gh]) and np.array_equal(agh, ang[gh]), rank


# ------------------------------------------------------------------ shape_from_shading row slabs (2 ghost rows), behind Thallo_ProblemStep
def _sfs_worker(rank, world, port, W, H, nit, lit, lm, q, device_exchange=True):
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed_sfs import PlanSfsSlabSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.shape_from_shading(W, H)
        solver = PlanSfsSlabSolver(p, W, H, rank, world, lit, lm=lm, device_exchange=device_exchange)
        extra = {"q_tolerance": 0.05} if lm else {}             # (LM: large enough that the device-side zeta test ends some PCG loops early)
        costs = solver.solve(nit, **extra)
        lay = solver.lay
        q.put((rank, costs, lay.g0, lay.g1, solver.owned(), solver.solver.distributed_info()))
        solver.solver.close()
    finally:
        dist.destroy_process_group()


def _run_sfs(world, W, H, nit, lit, lm, device_exchange=True):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sfs_worker, args=(r, world, port, W, H, nit, lit, lm, q, device_exchange)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = _collect(q, procs, world)
    res.sort(key=lambda t: t[0])
    return res


@pytest.mark.parametrize("world,W,H,nit,lit,lm", [(2, 64, 64, 4, 10, False), (3, 128, 112, 3, 10, False), (3, 128, 112, 3, 10, True), (2, 256, 96, 3, 12, True), (1, 64, 48, 3, 10, False)])
def test_sfs_device_side_row_exchange_is_bitwise_the_allgather_transport(world, W, H, nit, lit, lm):
    """shape_from_shading's slabs exchanged through pack + all-gather + unpack (three launches and a collective per exchange).  Now ONE launch:
    thallo_hip_dist_xrows stores the boundary rows into the neighbours' inboxes, sends the scalar granules to every rank, waits, sums in rank order and copies its own
    inbox into the ghost rows.  Gauss-Newton: same arithmetic in the same order, costs and unknowns of both transports are bit-identical.  Levenberg-Marquardt on the
    device-side transport goes further (round 3): the LM iteration is ONE marching launch + ONE exchange (13 sums + the rows of A p; q and betaN from their expansions in
    alpha, the zeta test in the exchange) -- equal to the reference-shaped loop of the all-gather transport to rounding, early exits included.  1-3 ranks."""
    a = _run_sfs(world, W, H, nit, lit, lm, device_exchange=True)
    b = _run_sfs(world, W, H, nit, lit, lm, device_exchange=False)
    for (rank, costs, g0, g1, owned, info), (_, costs_b, _, _, owned_b, info_b) in zip(a, b):
        assert info["exchange"] == "p2p-rows" and info["self_check"]["all_ranks_pass"] is True, info
        assert info_b["exchange"] == "allgather", info_b
        if not lm:
            assert costs == costs_b, (rank, costs, costs_b)
            assert np.array_equal(owned, owned_b), rank
        else:       # LM on the device-side transport is the one-launch iteration with ONE exchange (q and betaN from their expansions in alpha): equal to rounding
            assert len(costs) == len(costs_b) and np.abs(np.array(costs) - np.array(costs_b)).max() <= 2e-5 * np.abs(np.array(costs_b)).max(), (rank, costs, costs_b)
            assert np.abs(owned - owned_b).max() <= 2e-4 * max(1.0, np.abs(owned_b).max()), rank


@pytest.mark.parametrize("world,W,H,nit,lit", [(2, 64, 64, 4, 10), (3, 128, 112, 3, 10), (1, 64, 48, 3, 10)])
def test_hip_sfs_slabs_match_oracle(orc, world, W, H, nit, lit):
    """Gauss-Newton, single-reduction form: one exchange per PCG iteration (the device-side one: thallo_hip_dist_xrows)."""
    from thallo_amd import synthetic as syn
    res = _run_sfs(world, W, H, nit, lit, False)
    p = syn.shape_from_shading(W, H)
    co, _ = orc.Problem(orc.SFS, (W, H), p).solve(nIterations=nit, lIterations=lit)
    for rank, costs, g0, g1, X, info in res:
        assert info["exchange"] == "p2p-rows" and info["world"] == world
        assert (np.abs(np.array(costs) - co) <= 2e-5 * np.abs(co) + 1e-9).all(), (rank, costs, co)
        assert costs == res[0][1]
        assert np.abs(X - p[16][g0:g1]).max() <= 2e-5


@pytest.mark.parametrize("world,W,H,nit,lit", [(2, 64, 64, 5, 10), (3, 128, 112, 4, 10), (1, 64, 48, 4, 10)])
def test_hip_sfs_slabs_levenberg_marquardt_match_oracle(orc, world, W, H, nit, lit):
    """BASELINE config 4's solver across ranks: the LM branch (trust region, zeta test on the device) with every reduction made global where it
    is produced; identical cost trajectories -- including the steps LM rejects -- on every rank, and the oracle's LM trajectory."""
    from thallo_amd import synthetic as syn
    res = _run_sfs(world, W, H, nit, lit, True)
    p = syn.shape_from_shading(W, H)
    co, _ = orc.Problem(orc.SFS, (W, H), p).solve(nIterations=nit, lIterations=lit, use_lm=1, q_tolerance=0.05)
    for rank, costs, g0, g1, X, info in res:
        m = min(len(costs), len(co))
        assert m >= 3 and abs(len(costs) - len(co)) <= 1, (costs, co)
        assert (np.abs(np.array(costs[:m]) - co[:m]) <= 2e-4 * np.abs(co[:m])).all(), (rank, costs, co)       # the bar of test_shape_from_shading_lm
        assert costs == res[0][1]
        assert np.abs(X - p[16][g0:g1]).max() <= 2e-4


def test_hip_sfs_2048_levenberg_marquardt_two_ranks_match_one_gpu():
    """BASELINE config 4 at full size (2048 x 2048, LM + PCG 3 x 10), two row slabs against the single-GPU plan on the same instance (which
    tests/test_gpu_parity.py checks against the oracle at this size): same trajectory to 1e-5, identical on both ranks."""
    import torch
    import thallo_amd
    from thallo_amd import synthetic as syn
    W = H = 2048
    res = _run_sfs(2, W, H, 3, 10, True)
    p = syn.shape_from_shading(W, H)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if isinstance(a, np.ndarray) else float(a) for a in p]
    s = thallo_amd.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"))
    s.enable_lm()
    _, c1 = s.solve(dev, profiled=True, nIterations=3, lIterations=10, q_tolerance=0.05)
    s.close()
    for rank, costs, g0, g1, X, info in res:
        m = min(len(costs), len(c1))
        assert m >= 3 and (np.abs(np.array(costs[:m]) - np.array(c1[:m])) <= 1e-5 * np.abs(np.array(c1[:m]))).all(), (costs, c1)
        assert costs == res[0][1]
        # (sums over rows and ranks associate differently from one GPU's; 30 PCG iterations of an LM solve amplify that to a few 1e-4 of the depth values)
        assert np.abs(X - dev[16].view(H, W)[g0:g1].cpu().numpy()).max() <= 1e-3


def _worker_fail(rank, world, port, W, H, lit, q, energy, fail_rank, nth):
    """A 2-rank solve in which ONE rank reports a rank-local launch failure in the middle of its first step (ThalloX_DistributedControl what = 2)."""
    import torch
    import torch.distributed as dist
    from thallo_amd import synthetic as syn
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if energy == "iw":
            from thallo_amd.distributed import PlanSlabSolver
            solver = PlanSlabSolver(syn.image_warping(W, H, n_markers=8), W, H, rank, world, lit, device_exchange=False)
        elif energy == "ba":
            from thallo_amd.distributed_ba import PlanBaShardSolver
            solver = PlanBaShardSolver(syn.bundle_adjustment(C=16, P=300, O=1500, band=8), rank, world, lit)       # device-side all-reduce + scalar granules
        elif energy == "arap_part":
            from thallo_amd.distributed_graph import PlanArapPartitionSolver
            solver = PlanArapPartitionSolver(syn.arap_mesh(16, 12, n_handles=8, angle_amp=0.3), rank, world, lit)
        else:
            from thallo_amd.distributed_sfs import PlanSfsSlabSolver
            solver = PlanSfsSlabSolver(syn.shape_from_shading(W, H), W, H, rank, world, lit, lm=(energy == "sfs_lm"))
        s = solver.solver
        s.set_solver_parameters(nIterations=3, lIterations=lit)
        s.init(solver.params)
        c0 = s.current_cost()
        if rank == fail_rank:
            s._L.ThalloX_DistributedControl(s.plan, 2, nth)
        steps = 0
        while s.step(solver.params) and steps < 10:
            steps += 1
        c1 = s.current_cost()
        more = s.step(solver.params)
        from thallo_amd import api
        q.put((rank, c0, steps, c1, more, api.last_error()))
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("energy,nth", [("iw", 7), ("sfs", 9), ("sfs_lm", 12), ("ba", 9), ("arap_part", 8)] +
                         [("sfs_lm", int(n)) for n in os.environ.get("THALLO_FAIL_SWEEP", "27,33").split(",") if n])      # (sfs_lm: also behind the step's cost exchange; THALLO_FAIL_SWEEP=1,2,...: a sweep by hand)
def test_rank_local_failure_is_reported_by_every_rank_and_nobody_hangs(energy, nth):
    """a launch that fails on ONE rank used to return in front of the matching all-gather and leave the other ranks blocked in it.  Now the rank stays
    in the collective sequence with poisoned payloads and the failure becomes everybody's at the next cost evaluation: both ranks finish, both see a NaN cost and
    an error text, both plans refuse further steps."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, W, H, lit = 2, 64, 48, 6
    procs = [ctx.Process(target=_worker_fail, args=(r, world, port, W, H, lit, q, energy, 1, nth)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted(_collect(q, procs, world, limit=120.0))
    for rank, c0, steps, c1, more, err in res:
        assert np.isfinite(c0) and c0 > 0
        assert not np.isfinite(c1), (rank, c1)              # agreed: every rank's cost is void
        assert more == 0 and "fail" in err.lower(), (rank, err)
    assert res[0][2] == res[1][2]                           # the same number of steps on both ranks: nobody left the sequence early


# ------------------------------------------------------------------ the collectives inside the library (RCCL bound at run time)
def test_library_rccl_binding_and_world1_slab(orc):
    """ncclAllGather / ncclAllReduce issued by the library on the plan's stream instead of a ctypes callback into torch.distributed once or twice
    per PCG iteration.  What ONE GPU can check (RCCL refuses two ranks on a device): the run-time binding (a one-rank communicator moves a buffer through both
    collectives) and a shape_from_shading slab plan -- the flat form: one all-gather per PCG iteration -- at world size 1 through ncclAllGather, against the same
    plan on the in-process copy: identical costs, Gauss-Newton and LM."""
    import torch
    import thallo_amd
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed_sfs import PlanSfsSlabSolver
    torch.cuda.set_device(0)
    L = thallo_amd.lib()
    assert L.ThalloX_RcclSelfTest() == 0, thallo_amd.last_error()
    W, H = 96, 64
    p = syn.shape_from_shading(W, H)
    for lm in (False, True):
        runs = []
        for force in (True, False):
            s = PlanSfsSlabSolver(p, W, H, 0, 1, 10, lm=lm, force_rccl=force)
            assert s.library_rccl == force
            runs.append(s.solve(3))
            s.solver.close()
        assert runs[0] == runs[1] and len(runs[0]) >= 3 and runs[0][-1] < runs[0][0], runs
