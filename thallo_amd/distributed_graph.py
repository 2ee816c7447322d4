"""Vertex-partitioned multi-GPU runs of the graph-edge domain (ARAP mesh deformation), behind Thallo_ProblemStep
(csrc/solver_dist.cpp, range form).

Rank k owns the contiguous vertex range [n0,n1) (equal ranges).  A graph that fits one GPU's cache hierarchy many times over (102,400
vertices = 2.4 MB per solver vector) does not need ghost-vertex bookkeeping: every rank holds the whole problem and FULL-length vectors,
runs the gather kernels for its own vertices only (the n0,n1 arguments of thallo_hip_arap_*) and the energy-independent vector update for
ALL unknowns -- redundantly, same inputs, same bits.  What travels per PCG iteration is ONE all-gather of [alphaD | N, S1, S2 | the owned
slice of A p]; the unknowns stay replicated without any exchange (PlanArapSolver).

Round 3 adds the REAL partition (PlanArapPartitionSolver, GhostPartition): a rank's Plan is its local sub-mesh -- owned vertices first, then the ghost vertices its
owned ones share an edge with; all directed edges with an owned end -- so memory and the vector updates are local-sized, and what travels per PCG iteration is
[alphaD | N, S1, S2 | A p at the BOUNDARY vertices] (ThalloX_PlanSetGhostExchange; solver_dist.cpp, partition form).  A ghost's r, p, delta follow from its owner's
A p by the same arithmetic on the same bits, so the ghosts' unknowns stay equal to their owners' without an exchange of their own.  This module is set-up only.
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


class GhostPartition:
    """Contiguous owned ranges [n0, n1) of the GLOBAL vertex ids (equal chunks; the last rank takes the remainder) + the ghost bookkeeping every rank derives from
    the global directed edge list (v0 -> v1): local ids = owned first (ascending global id), then ghosts (ascending global id)."""

    def __init__(self, N, v0, v1, rank, world):
        v0 = np.asarray(v0, dtype=np.int64); v1 = np.asarray(v1, dtype=np.int64)
        self.N, self.rank, self.world = N, rank, world
        chunk = N // world
        bounds = [r * chunk for r in range(world)] + [N]
        self.bounds = bounds
        owner = np.minimum(np.arange(N) // max(chunk, 1), world - 1)
        self.n0, self.n1 = bounds[rank], bounds[rank + 1]
        cross = owner[v0] != owner[v1]
        # boundary list of every rank: its owned vertices that have an edge (either direction) to a vertex owned elsewhere, ascending
        self.boundary_global = []
        for r in range(world):
            b = np.unique(np.concatenate([v0[cross & (owner[v0] == r)], v1[cross & (owner[v1] == r)]]))
            self.boundary_global.append(b)
        mine0, mine1 = owner[v0] == rank, owner[v1] == rank
        keep = mine0 | mine1                                            # every directed edge with an owned end
        ghosts = np.unique(np.concatenate([v1[mine0 & ~mine1], v0[mine1 & ~mine0]]))
        self.owned_global = np.arange(self.n0, self.n1)
        self.local_global = np.concatenate([self.owned_global, ghosts])     # local id -> global id
        lut = np.full(N, -1, np.int64); lut[self.local_global] = np.arange(self.local_global.size)
        self.n_own, self.n_loc = self.owned_global.size, self.local_global.size
        self.edge_ids = np.nonzero(keep)[0]
        self.v0_local, self.v1_local = lut[v0[keep]].astype(np.int32), lut[v1[keep]].astype(np.int32)
        self.boundary_units = lut[self.boundary_global[rank]].astype(np.int32)
        self.ghost_units = np.arange(self.n_own, self.n_loc, dtype=np.int32)
        self.ghost_src_rank = owner[ghosts].astype(np.int32)
        self.ghost_src_pos = np.array([np.searchsorted(self.boundary_global[owner[g]], g) for g in ghosts], dtype=np.int32)
        for g, r, p_ in zip(ghosts, self.ghost_src_rank, self.ghost_src_pos):      # every ghost is in its owner's boundary list (it has an edge to one of MY vertices)
            assert self.boundary_global[r][p_] == g


class PlanArapPartitionSolver:
    """arap_mesh_deformation over a real vertex partition: the Plan sees only the local sub-mesh."""

    def __init__(self, params, rank, world, l_iters, group=None, device_exchange=True):
        w_fit, w_reg, pos, ang, orig, cons, v0, v1 = params
        N = pos.shape[0]
        self.part = part = GhostPartition(N, v0, v1, rank, world)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        lg = part.local_global
        self.position, self.angle = t(pos[lg]), t(ang[lg])
        self._const = [t(orig[lg]), t(cons[lg]), t(part.v0_local), t(part.v1_local)]
        self.solver = api.ThalloSolver((part.n_loc, part.v0_local.shape[0]), api.energy_file("arap_mesh_deformation"), timing_level=0)
        self.solver.set_solver_parameters(nIterations=1 << 30, lIterations=l_iters)
        self.library_rccl = library_rccl(self.solver, rank, world, group)
        ag = torch_allgather(group, dev) if world > 1 and not self.library_rccl else None
        self.solver.set_ghost_exchange(part.boundary_units, part.ghost_units, part.ghost_src_rank, part.ghost_src_pos)
        self.solver.set_distributed(rank, world, 0, part.n_own, allgather=ag, device_exchange=device_exchange)      # device_exchange: the boundary values + scalars in ONE launch of peer stores
        self.params = self.solver.make_params([float(w_fit), float(w_reg), self.position, self.angle] + self._const)

    solve = PlanArapSolver.solve

    def owned(self):
        n = self.part.n_own
        return self.position[:n].cpu().numpy(), self.angle[:n].cpu().numpy()

    def ghosts(self):
        n = self.part.n_own
        return self.position[n:].cpu().numpy(), self.angle[n:].cpu().numpy()
