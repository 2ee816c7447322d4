"""Vertex-partitioned multi-GPU runs of the graph-edge domain (ARAP mesh deformation), behind Thallo_ProblemStep
(csrc/solver_dist.cpp, range form).

Rank k owns the contiguous vertex range [n0,n1) (equal ranges).  A graph that fits one GPU's cache hierarchy many times over (102,400
vertices = 2.4 MB per solver vector) does not need ghost-vertex bookkeeping: every rank holds the whole problem and FULL-length vectors,
runs the gather kernels for its own vertices only (the n0,n1 arguments of thallo_hip_arap_*) and the energy-independent vector update for
ALL unknowns -- redundantly, same inputs, same bits.  What travels per PCG iteration is ONE all-gather of [alphaD | N, S1, S2 | the owned
slice of A p]; the unknowns stay replicated without any exchange.  This module is set-up only.
"""
import numpy as np
import torch

from . import api
from .distributed import library_rccl, torch_allgather


class VertexPartition:
    def __init__(self, N, rank, world):
        if N % (4 * world):
            raise ValueError(f"vertex-partitioned path needs N % (4*world) == 0 (N={N}, world={world})")
        self.N, self.rank, self.world = N, rank, world
        self.chunk = N // world
        self.n0, self.n1 = rank * self.chunk, (rank + 1) * self.chunk


class PlanArapSolver:
    def __init__(self, params, rank, world, l_iters, group=None):
        w_fit, w_reg, pos, ang, orig, cons, v0, v1 = params
        N, E = pos.shape[0], v0.shape[0]
        self.part = part = VertexPartition(N, rank, world)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.position, self.angle = t(pos), t(ang)
        self._const = [t(orig), t(cons), t(v0.astype(np.int32)), t(v1.astype(np.int32))]
        self.solver = api.ThalloSolver((N, E), api.energy_file("arap_mesh_deformation"), timing_level=0)
        self.solver.set_solver_parameters(nIterations=1 << 30, lIterations=l_iters)
        self.library_rccl = library_rccl(self.solver, rank, world, group)      # ranks on GPUs of their own: the all-gather runs inside the library (no callback)
        ag = torch_allgather(group, dev) if world > 1 and not self.library_rccl else None
        self.solver.set_distributed(rank, world, part.n0, part.n1, allgather=ag, device_exchange=False)     # (row0, row1 = the owned vertex range)
        self.params = self.solver.make_params([float(w_fit), float(w_reg), self.position, self.angle] + self._const)

    def solve(self, n_iters, **solver_params):
        self.solver.set_solver_parameters(nIterations=n_iters, **solver_params)
        self.solver.init(self.params)
        if not self.solver.ready():
            raise RuntimeError("Thallo_ProblemInit failed: " + api.last_error())
        costs = [self.solver.current_cost()]
        while self.solver.step(self.params):
            costs.append(self.solver.current_cost())
        return costs
