"""Soak test of the device-side exchange (GPU box): `world` processes on GPU 0 over real IPC mappings run many GN steps of the slab
schedule behind Thallo_ProblemStep; every rank must finish without a mailbox timeout, hold bit-identical alpha/beta traces, and land on
the cost of the all-gather transport.  python tools/p2p_soak.py [world] [gn_steps] [l_iters] [W] [H]
SOAK_DOMAIN=sfs | sfs_lm | arap | ba: the same for the other device-side exchanges (row inboxes + granules, boundary units, the peer-store all-reduce): many steps, no
timeout, every rank the same cost trajectory, the all-gather transport's final cost to rounding."""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mask_devices():
    # SOAK_MASK=1: the per-rank visibility mask of a real launcher (HIP_VISIBLE_DEVICES=$LOCAL_RANK): every rank sees exactly ONE device and calls it device 0.  On a
    # one-GPU box that is the same card for everybody, but handles are exported and opened between processes whose device enumerations are private -- as far as one GPU
    # can emulate "rank r's device 0 is not rank s's device 0" (VERDICT r4 item 5d).  Set before torch / HIP are loaded.
    if os.environ.get("SOAK_MASK") == "1": os.environ["HIP_VISIBLE_DEVICES"] = "0"


def worker(rank, world, port, W, H, steps, L, q):
    mask_devices()
    import torch, torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed import PlanSlabSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.image_warping(W, H, n_markers=8)
        ref = PlanSlabSolver(p, W, H, rank, world, L, device_exchange=False)
        c_ref = ref.solve(steps, L)
        s = PlanSlabSolver(p, W, H, rank, world, L, device_exchange=True)
        c0 = s.cost()
        t0 = time.time()
        for i in range(steps):
            s.gn_step()
        torch.cuda.synchronize()
        dt = time.time() - t0
        err = s.solver.distributed_error()
        q.put((rank, s.info, err, c_ref[-1], [c0, s.cost()], s.solver.alpha_beta_trace(), dt))
    finally:
        dist.destroy_process_group()


def worker_other(rank, world, port, W, H, steps, L, q):
    mask_devices()
    import torch, torch.distributed as dist
    from thallo_amd import synthetic as syn
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dom = os.environ.get("SOAK_DOMAIN")
    try:
        def make(dx):
            if dom in ("sfs", "sfs_lm"):
                from thallo_amd.distributed_sfs import PlanSfsSlabSolver
                return PlanSfsSlabSolver(syn.shape_from_shading(W, H), W, H, rank, world, L, lm=(dom == "sfs_lm"), device_exchange=dx)
            if dom == "arap":
                from thallo_amd.distributed_graph import PlanArapPartitionSolver
                return PlanArapPartitionSolver(syn.arap_mesh(W // 4, H // 4, n_handles=8, angle_amp=0.3), rank, world, L, device_exchange=dx)
            from thallo_amd.distributed_ba import PlanBaShardSolver
            return PlanBaShardSolver(syn.bundle_adjustment(C=48, P=3001, O=15000, band=8), rank, world, L, device_exchange=dx)
        c_ref = make(False).solve(steps)
        s = make(True)
        t0 = time.time()
        c = s.solve(steps)
        dt = time.time() - t0
        q.put((rank, s.solver.distributed_info(), s.solver.distributed_error(), c_ref[-1], [c[0], c[-1]], c, dt))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 192
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    dom = os.environ.get("SOAK_DOMAIN")
    procs = [ctx.Process(target=worker_other if dom else worker, args=(r, world, port, W, H, steps, L, q)) for r in range(world)]
    for p_ in procs: p_.start()
    res = []
    t0 = time.time()
    while len(res) < world and time.time() - t0 < 900:
        try:
            res.append(q.get(timeout=1.0))
        except Exception:
            if any(p_.exitcode not in (None, 0) for p_ in procs): break
    for p_ in procs: p_.join(timeout=20)
    res.sort(key=lambda t: t[0])
    ok = len(res) == world
    for rank, info, err, c_ref, c, trace, dt in res:
        print(f"rank {rank}: exchange={info.get('exchange')} err={err} cost {c[0]:.6g} -> {c[1]:.6g} (all-gather path: {c_ref:.6g}) {dt:.2f} s")
        want = {None: "p2p-mailbox", "sfs": "p2p-rows", "sfs_lm": "p2p-rows", "arap": "p2p-units", "ba": "p2p-allreduce + allgather"}[dom]
        tol = 1e-5 if dom != "sfs_lm" else 5e-2      # (LM: the two transports run different schedules; §5 on how fast LM trajectories of this energy drift apart)
        ok = ok and info.get("exchange") == want and err == 0 and trace == res[0][5] and abs(c[1] - c_ref) <= tol * abs(c_ref)
    print("SOAK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
