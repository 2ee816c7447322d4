"""Camera-sharded multi-GPU runs of bundle adjustment on the materialized sparse-J path, behind Thallo_ProblemStep
(csrc/solver_dist.cpp, shard form).

Rank k owns a contiguous range of cameras and every observation (row pair of J) of those cameras; the points are replicated.  Then
  * J p is local (an observation needs its own camera and its point, both present);
  * the camera block of J^T(Jp) is complete locally; the POINT block is a partial sum over the rank's observations -> one all-reduce of 3P floats per
    PCG iteration (and of the point blocks of J^T F and diag(J^T J) once per GN step);
  * after that every rank holds identical point blocks of Ap, r, p, delta and updates them redundantly, so the point unknowns never need a broadcast;
  * the scalars: the ranks' camera parts of [alphaD | N, S1, S2] in one tiny all-gather (added in rank order) + the point parts every rank computes
    for itself after the all-reduce;
  * round 6, lm=True: the Levenberg-Marquardt branch on the same layout (solver_dist.cpp step_lm_shard): the element-wise LM kernels run on the camera block and the
    point block separately, the camera sums travel in the tiny all-gather, the point sums are added by every rank for itself; accept / revert and the trust region are
    replicated host logic on identical scalars;
  * round 3, device_exchange=True: the all-reduce is ONE launch of peer stores (thallo_hip_dist_allreduce: reduce-scatter into the chunk owners' inboxes, sums in
    rank order, all-gather into every rank's second inbox) after a self-check at Init -- no ncclAllReduce in the PCG loop; distributed_info() says which.
The kernels are the single-GPU ones run on the local sub-instance [cameras of this rank (padded to a multiple of 4) | all points].  This module is
set-up only: the shard (BaShardLayout) and the two callbacks over torch.distributed.
"""
import numpy as np
import torch

from . import api
from .distributed import library_rccl, torch_allgather, torch_allreduce


class BaShardLayout:
    def __init__(self, C_total, rank, world):
        per, rem = divmod(C_total, world)
        counts = [per + (1 if r < rem else 0) for r in range(world)]
        starts = np.concatenate([[0], np.cumsum(counts)])
        self.rank, self.world, self.C_total = rank, world, C_total
        self.c0, self.c1 = int(starts[rank]), int(starts[rank + 1])
        if self.c1 <= self.c0:
            raise ValueError(f"rank {rank} of {world} owns no cameras")
        self.C_loc = self.c1 - self.c0
        self.C_pad = (self.C_loc + 3) // 4 * 4          # 16-byte alignment of the point block in the flat vectors

    def shard(self, params):
        """local sub-instance of a global params list [cameras, points, observations, oToC, oToP]"""
        cams, pts, obs, oc, op = params
        sel = np.nonzero((oc >= self.c0) & (oc < self.c1))[0]
        cl = np.zeros((self.C_pad, 9), np.float32)
        cl[: self.C_loc] = cams[self.c0:self.c1]
        # the shared (point) block is all-reduced and updated with 16-byte accesses: 3 P floats must be a multiple of 4 -> pad with points nobody observes
        # (no rows, zero gradient, they never move; ladybug-1723 has 156,502 points)
        self.P, self.P_pad = pts.shape[0], (pts.shape[0] + 3) // 4 * 4
        pl = np.zeros((self.P_pad, 3), np.float32); pl[: self.P] = pts
        pts = pl
        return [cl, np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(obs[sel]),
                np.ascontiguousarray(oc[sel] - self.c0, np.int32), np.ascontiguousarray(op[sel], np.int32)]


class PlanBaShardSolver:
    def __init__(self, params_global, rank, world, l_iters, group=None, device_exchange=True, lm=False):
        self.lay = lay = BaShardLayout(params_global[0].shape[0], rank, world)
        local = lay.shard(params_global)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.tensors = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in local]
        self.cameras, self.points = self.tensors[0], self.tensors[1]
        dims = (lay.C_pad, local[1].shape[0], local[2].shape[0])
        self.solver = api.ThalloSolver(dims, api.energy_file("bundle_adjustment"), timing_level=0, **({"solverkind": "levenberg_marquardt"} if lm else {}))
        if lm:          # round 6: the LM branch on camera shards (csrc/solver_dist.cpp step_lm_shard): the reference's example runs this energy as LM 5 x 150
            self.solver.enable_lm()
        self.solver.set_solver_parameters(nIterations=1 << 30, lIterations=l_iters)
        self.library_rccl = library_rccl(self.solver, rank, world, group)      # ranks on GPUs of their own: all-gather and all-reduce run inside the library (no callback)
        ag = torch_allgather(group, dev) if world > 1 and not self.library_rccl else None
        ar = torch_allreduce(group, dev) if world > 1 and not self.library_rccl else None
        self.solver.set_distributed(rank, world, lay.c0, lay.c1, allgather=ag, device_exchange=device_exchange, allreduce=ar)
        self.params = self.solver.make_params(self.tensors)

    def solve(self, n_iters, **solver_params):
        self.solver.set_solver_parameters(nIterations=n_iters, **solver_params)
        self.solver.init(self.params)
        if not self.solver.ready():
            raise RuntimeError("Thallo_ProblemInit failed: " + api.last_error())
        costs = [self.solver.current_cost()]
        while self.solver.step(self.params):
            costs.append(self.solver.current_cost())
        return costs
