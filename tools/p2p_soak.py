"""Soak test of the device-side exchange (GPU box): `world` processes on GPU 0 over real IPC mappings run many GN steps of the slab
schedule behind Thallo_ProblemStep; every rank must finish without a mailbox timeout, hold bit-identical alpha/beta traces, and land on
the cost of the all-gather transport.  python tools/p2p_soak.py [world] [gn_steps] [l_iters] [W] [H]"""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, W, H, steps, L, q):
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
    procs = [ctx.Process(target=worker, args=(r, world, port, W, H, steps, L, q)) for r in range(world)]
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
        ok = ok and info.get("exchange") == "p2p-mailbox" and err == 0 and trace == res[0][5] and abs(c[1] - c_ref) <= 1e-5 * abs(c_ref)
    print("SOAK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
