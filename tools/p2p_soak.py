"""Soak test of the device-side exchange (GPU box): `world` processes on GPU 0 over real IPC mappings run many GN steps of the
one-kernel schedule; every rank must finish without a mailbox timeout, hold bit-identical alpha/beta traces, and land on the cost
of the collective path.  python tools/p2p_soak.py [world] [gn_steps] [l_iters] [W] [H]"""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def worker(rank, world, port, W, H, steps, L, q):
    import torch, torch.distributed as dist
    from thallo_amd import synthetic as syn
    from thallo_amd.distributed import make_hip_solver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = syn.image_warping(W, H, n_markers=8)
        ref, _ = make_hip_solver(p, W, H, rank, world, L)
        c_ref = ref.solve(3, L)
        s, lay = make_hip_solver(p, W, H, rank, world, L, ipc=True)
        on = s.try_enable_p2p(l_iters=min(6, L))
        c0 = s.cost()
        t0 = time.time()
        traces = []
        for i in range(steps):
            s.gn_step_p2p(L)
            if i < 3 or i == steps - 1:
                traces.append(s.be.S[2:2 + 2 * L + 1].cpu().numpy().copy())
        torch.cuda.synchronize()
        dt = time.time() - t0
        err = s.be.p2p_error()
        q.put((rank, on, s.p2p_check, err, getattr(s.be, "p2p_post_mortem", None), c_ref, [c0, s.cost()], np.concatenate(traces), dt))
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
    for r in res:
        print("rank", r[0], "p2p", r[1], r[2], "timeout", r[3], r[4], "collective costs", [round(c, 5) for c in r[5]], "p2p costs", [round(c, 6) for c in r[6]], "%.1f s" % r[8])
        ok = ok and r[1] and r[3] == 0 and (r[7] == res[0][7]).all()
    print("exchanges per rank:", steps * L, "  SOAK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
