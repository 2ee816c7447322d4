"""CPU statement of the ROW-SLAB schedules of the multi-GPU Gauss-Newton step (thallo_amd/csrc/solver_dist.cpp Plan::dist_gn / dist_gn_flat) --
TEST INFRASTRUCTURE: a host-side driver over a pluggable compute backend (tests/slab_numpy_backend.py, tests/sfs_scipy_backend.py) and
torch.distributed / gloo, run by tests/test_distributed_cpu.py with world sizes 2 and 3.
  * the H rows of the image are split into contiguous slabs, one per rank; a rank's local image carries ghost rows above / below;
  * per PCG iteration the exchanges are dictated by the algorithm (gauss_newton.t:1641-1665): the scalars and the ghost rows of the vector the
    stencil is applied to -- in the one-exchange form (backends with iter_collective) ONE all-gather of [alphaD, N, S1, S2 | boundary rows of Ap],
    otherwise an all-reduce of alphaD plus one all-gather of [betaN | boundary rows of z];
  * every rank adds the gathered partial sums in rank order, so alpha and beta are bit-identical on all ranks and the replicated host logic
    cannot diverge;
  * once per GN step the ghost rows of the unknowns are refreshed the same way.
"""
import torch
import torch.distributed as dist

from thallo_amd import api


def _segs(pairs):
    s = api.SegsT()
    for k, (o, l) in enumerate(pairs):
        s.off[k] = o
        s.len[k] = l
    s.n = len(pairs)
    return s


class SlabSolver:
    """Gauss-Newton + PCG over row slabs; replicated host logic, rank-ordered sums (gauss_newton.t:1545-1785)."""

    def __init__(self, backend, layout, group=None, force_collectives=False):
        self.be, self.lay, self.group = backend, layout, group
        self.world = layout.world
        self.use_dist = self.world > 1 or force_collectives     # force: issue the collectives even at world size 1 (probes)

    # -- collectives
    def _allreduce(self, idx):
        if self.use_dist:
            dist.all_reduce(self.be.S[idx:idx + 1], group=self.group)

    def _gather_sum_and_rows(self, out_idx):
        be = self.be
        be.pack()
        if self.use_dist:
            dist.all_gather_into_tensor(be.gath, be.send, group=self.group)
            be.unpack(out_idx, be.gath)
        else:
            be.unpack(out_idx, be.send)

    def _exchange_unknown_ghosts(self):
        """once per GN step: ghost rows of the unknowns <- neighbours' boundary rows (backend packs / unpacks)"""
        if not self.use_dist:
            return
        send = self.be.pack_unknowns()
        gath = torch.empty(self.world * send.numel(), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(gath, send, group=self.group)
        self.be.unpack_unknowns(gath.view(self.world, -1))

    # -- solver
    def cost(self):
        self.be.cost_local(0)
        self._allreduce(0)
        return self.be.scalar(0)

    def gn_step(self, l_iters):
        """One Gauss-Newton iteration: PCGInit + l_iters PCG iterations + linear update (no host sync)."""
        be = self.be
        B, L = 2, l_iters
        cur = 0
        batched = getattr(be, "batches_delta", False)
        be.init(cur)                                   # local alphaN partials, z, ...
        if self.use_dist and hasattr(be, "pack_grid_info"):
            # every rank must pick the same PCG schedule (z-free iff UrShape is the pixel grid everywhere), and the ghost rows
            # need their owner's flags byte (M^-1 of a ghost pixel depends on rows this rank does not hold)
            send = be.pack_grid_info()
            gath = torch.empty(self.world * send.numel(), dtype=send.dtype, device=send.device)
            dist.all_gather_into_tensor(gath, send, group=self.group)
            be.unpack_grid_info(gath.view(self.world, -1))
        self._gather_sum_and_rows(B)                   # S[B] = alphaN_0 (global); ghost rows of r and z
        one_kernel = getattr(be, "one_kernel_collective", False)
        if one_kernel and hasattr(be, "irregular"):
            if not getattr(self, "_grid_checked", False):
                # the HIP one-kernel schedule needs UrShape on the pixel grid on every rank (z-free); checked once per solver (one host sync)
                self._grid_ok = int(be.irregular[0].item()) == 0
                self._grid_checked = True
            one_kernel = self._grid_ok
        ag = (lambda send, recv: dist.all_gather_into_tensor(recv, send, group=self.group)) if self.use_dist else None
        for k in range(L if one_kernel else 0):        # one kernel + ONE all-gather per PCG iteration
            jN, jD, jB = B + 2 * k, B + 2 * k + 1, B + 2 * k + 2
            mode = (1 if k == 0 else 2 if k & 1 else 4) if batched else (1 if k == 0 else 0)
            be.iter_collective(cur, mode, jN - 2 if k else jN, jD - 2 if k else jD, jN, jD, jB,
                               jN - 4 if k > 1 else jN, jD - 4 if k > 1 else jD, ag)
            cur ^= 1
        for k in range(0 if one_kernel else L):
            jN, jD, jB = B + 2 * k, B + 2 * k + 1, B + 2 * k + 2
            if batched:                                # every other delta update deferred (thallo_hip.h THALLO_IW_STEP1_MODE)
                be.step1(cur, 1 if k == 0 else 2 if k & 1 else 4, jN - 2 if k else jN, jD - 2 if k else jD, jN, jD,
                         jN - 4 if k > 1 else jN, jD - 4 if k > 1 else jD)
            else:
                be.step1(cur, k == 0, jN - 2 if k else jN, jD - 2 if k else jD, jN, jD)
            self._allreduce(jD)                        # alphaD_k
            cur ^= 1
            be.step2(jN, jD)
            self._gather_sum_and_rows(jB)              # betaN_k ; ghost rows of z
        if L > 1 and batched and (L - 1) & 1:
            be.linear_update2(cur, B + 2 * (L - 2), B + 2 * (L - 2) + 1, B + 2 * (L - 1), B + 2 * (L - 1) + 1)
        elif L > 0:
            be.linear_update(cur, B + 2 * (L - 1), B + 2 * (L - 1) + 1, True)
        else:
            be.linear_update(cur, B, B, False)
        self._exchange_unknown_ghosts()

    def solve(self, n_iters, l_iters):
        costs = [self.cost()]
        for _ in range(n_iters):
            self.gn_step(l_iters)
            costs.append(self.cost())
        return costs
