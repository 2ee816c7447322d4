"""CPU mirror of the RANGE form of the multi-GPU Gauss-Newton step (thallo_amd/csrc/solver_dist.cpp Plan::dist_gn_range) for the vertex-partitioned
graph energies -- TEST INFRASTRUCTURE (oracle CSR + scipy, torch.distributed / gloo).  The executable statement of that schedule:
every rank holds the whole problem and FULL-length vectors, applies J^T J for its own vertex range only, does the vector update for ALL unknowns
redundantly, and per PCG iteration ONE all-gather of [alphaD | N, S1, S2 | the owned slice of A p] travels; partial sums are added in rank order.
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

from oracle import oracle as orc

F = np.float32


def _allgather(vec, world):
    t = torch.from_numpy(np.ascontiguousarray(vec))
    out = [torch.empty_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(out, t)
    else:
        out = [t]
    return [o.numpy() for o in out]


class ArapRangeMirror:
    def __init__(self, part, params):
        self.part = part
        self.params = [a.copy() if isinstance(a, np.ndarray) else a for a in params]
        N, E = self.params[2].shape[0], self.params[6].shape[0]
        self.N, self.E, self.n = N, E, 6 * N
        own = np.zeros(self.n, bool)
        own[3 * part.n0:3 * part.n1] = True; own[3 * N + 3 * part.n0:3 * N + 3 * part.n1] = True
        self.own = own
        v0 = self.params[6]
        rows_own = np.zeros(3 * N + 3 * E, bool)
        rows_own[3 * part.n0:3 * part.n1] = True                                     # fit rows of owned vertices
        rows_own[3 * N:] = np.repeat((v0 >= part.n0) & (v0 < part.n1), 3)            # reg rows of edges leaving owned vertices
        self.rows_own = rows_own

    def _problem(self):
        return orc.Problem(orc.ARAP_MESH, (self.N, self.E), self.params)

    def _replicate(self, vec):
        """every rank's owned slices of `vec` (two planes) into place; returns the full vector"""
        N, w = self.N, self.part.world
        got = _allgather(vec[self.own], w)
        full = np.empty_like(vec)
        for r, g in enumerate(got):
            c = self.part.chunk
            full[3 * c * r:3 * c * (r + 1)] = g[:3 * c]; full[3 * N + 3 * c * r:3 * N + 3 * c * (r + 1)] = g[3 * c:]
        return full

    def cost(self):
        res = self._problem().csr()[3].astype(np.float64)
        mine = np.array([0.5 * (res[self.rows_own] ** 2).sum()], np.float64)
        return float(sum(F(g[0]) for g in _allgather(mine, self.part.world)))

    def gn_step(self, L):
        w, own = self.part.world, self.own
        rp, col, val, res = self._problem().csr()
        J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), self.n))
        r = (-(J.T @ res.astype(np.float64))).astype(F)
        d = np.asarray(J.multiply(J).sum(0)).ravel().astype(F)
        pre = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        z = pre * r
        aN = F(sum(F(g[0]) for g in _allgather(np.array([(r[own].astype(np.float64) * z[own]).sum()]), w)))       # rank-ordered
        r, pre = self._replicate(r), self._replicate(pre)                            # (identical here; on the GPU only the owned slices are computed)
        p = np.zeros(self.n, F); delta = np.zeros(self.n, F); Ap = np.zeros(self.n, F)
        alpha = beta = F(0)
        for k in range(L):
            if k:                                                                    # pcg_update over ALL unknowns, redundantly
                r = (r - alpha * Ap).astype(F); delta = (delta + alpha * p).astype(F)
            p = (pre * r + beta * p).astype(F)
            ap = (J.T @ (J @ p.astype(np.float64))).astype(F)
            m64, r64, a64, p64 = pre[own].astype(np.float64), r[own].astype(np.float64), ap[own].astype(np.float64), p[own].astype(np.float64)
            msg = np.concatenate([[(p64 * a64).sum(), (m64 * r64 * r64).sum(), (m64 * r64 * a64).sum(), (m64 * a64 * a64).sum()], ap[own].astype(np.float64)])
            got = _allgather(msg, w)                                                 # ONE exchange per PCG iteration
            aD = F(sum(F(g[0]) for g in got)); n_, s1, s2 = (sum(g[i] for g in got) for i in (1, 2, 3))
            c, N = self.part.chunk, self.N
            for rk, g in enumerate(got):
                Ap[3 * c * rk:3 * c * (rk + 1)] = g[4:4 + 3 * c]; Ap[3 * N + 3 * c * rk:3 * N + 3 * c * (rk + 1)] = g[4 + 3 * c:]
            alpha = aN / aD if aD != 0 else F(0)
            bN = F(max(n_ - 2.0 * float(alpha) * s1 + float(alpha) ** 2 * s2, 0.0))
            beta = bN / aN if aN != 0 else F(0)
            aN = bN
        if L:
            delta = (delta + alpha * p).astype(F)
        N = self.N
        self.params[2].reshape(-1)[:] += delta[:3 * N]                               # every rank updates every unknown: they stay replicated
        self.params[3].reshape(-1)[:] += delta[3 * N:]

    def solve(self, nit, lit):
        costs = [self.cost()]
        for _ in range(nit):
            self.gn_step(lit)
            costs.append(self.cost())
        return costs


class ArapPartitionMirror:
    """CPU mirror of the PARTITION form (solver_dist.cpp, D.part; thallo_amd/distributed_graph.py GhostPartition): a rank holds only its local sub-mesh -- owned vertices
    [0, n_own), then ghosts; the edges with an owned end -- and local-sized vectors.  Per GN step the ghosts' r and M^-1 come from their owners; per PCG iteration ONE
    all-gather of [alphaD | N, S1, S2 | A p at my boundary vertices] fills the ghosts' A p, after which a ghost's r, p, delta follow from the same arithmetic as its
    owner's.  J restricted to the rows of owned fit terms and of edges with an owned SOURCE reproduces exactly the owned rows of the global J^T J p."""

    def __init__(self, gp, params):
        self.gp = gp
        w_fit, w_reg, pos, ang, orig, cons, v0, v1 = params
        lg = gp.local_global
        self.local = [w_fit, w_reg, pos[lg].copy(), ang[lg].copy(), orig[lg].copy(), cons[lg].copy(), gp.v0_local.copy(), gp.v1_local.copy()]
        self.Nl, self.El = len(lg), len(gp.v0_local)
        self.n = 6 * self.Nl
        no = gp.n_own
        own = np.zeros(self.n, bool); own[:3 * no] = True; own[3 * self.Nl:3 * self.Nl + 3 * no] = True
        self.own = own
        rows = np.zeros(3 * self.Nl + 3 * self.El, bool)
        rows[:3 * no] = True                                                           # fit rows of owned vertices
        rows[3 * self.Nl:] = np.repeat(gp.v0_local < no, 3)                            # reg rows of edges LEAVING an owned vertex (each directed edge counted by its source's owner)
        self.rows_cost = rows
        rows_any = np.zeros_like(rows); rows_any[:3 * no] = True; rows_any[3 * self.Nl:] = True      # J^T J p at an owned vertex needs every local edge (in- and out-)
        self.rows_apply = rows_any
        self.maxb = None

    def _unit_floats(self, units):
        u = np.asarray(units, np.int64)
        return np.concatenate([(3 * u[:, None] + np.arange(3)).ravel(), (3 * self.Nl + 3 * u[:, None] + np.arange(3)).ravel()]) if len(u) else np.zeros(0, np.int64)

    def _exchange(self, header, vec):
        """all-gather of [header | vec at my boundary units (padded)]; returns every rank's header and fills my ghosts"""
        gp, w = self.gp, self.gp.world
        if self.maxb is None:
            self.maxb = max(int(g[0]) for g in _allgather(np.array([float(len(gp.boundary_units))]), w))
        body = np.zeros(6 * self.maxb, np.float64)
        idx = self._unit_floats(gp.boundary_units)
        nb = len(gp.boundary_units)
        body[:3 * nb] = vec[idx[:3 * nb]]; body[3 * self.maxb:3 * self.maxb + 3 * nb] = vec[idx[3 * nb:]]
        got = _allgather(np.concatenate([np.asarray(header, np.float64), body]), w)
        h = len(header)
        for g, r, pos in zip(gp.ghost_units, gp.ghost_src_rank, gp.ghost_src_pos):
            vec[3 * g:3 * g + 3] = got[r][h + 3 * pos:h + 3 * pos + 3]
            vec[3 * self.Nl + 3 * g:3 * self.Nl + 3 * g + 3] = got[r][h + 3 * self.maxb + 3 * pos:h + 3 * self.maxb + 3 * pos + 3]
        return [g[:h] for g in got]

    def _csr(self):
        return orc.Problem(orc.ARAP_MESH, (self.Nl, self.El), self.local).csr()

    def cost(self):
        res = self._csr()[3].astype(np.float64)
        mine = np.array([0.5 * (res[self.rows_cost] ** 2).sum()], np.float64)
        return float(sum(F(g[0]) for g in _allgather(mine, self.gp.world)))

    def gn_step(self, L):
        own = self.own
        rp, col, val, res = self._csr()
        J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), self.n))
        Ja = sp.diags(self.rows_apply.astype(np.float64)) @ J                        # (all local rows: a ghost's own rows are incomplete, its values come from its owner)
        r = (-(Ja.T @ res.astype(np.float64))).astype(F)
        d = np.asarray(Ja.multiply(Ja).sum(0)).ravel().astype(F)
        pre = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        aN_loc = (r[own].astype(np.float64) * (pre[own] * r[own])).sum()
        hs = self._exchange([aN_loc], r); aN = F(sum(F(h[0]) for h in hs))
        self._exchange([], pre)
        p = np.zeros(self.n, F); delta = np.zeros(self.n, F); Ap = np.zeros(self.n, F)
        alpha = beta = F(0)
        for k in range(L):
            if k:
                r = (r - alpha * Ap).astype(F); delta = (delta + alpha * p).astype(F)
            p = (pre * r + beta * p).astype(F)
            Ap = (Ja.T @ (Ja @ p.astype(np.float64))).astype(F)                      # right on the owned entries; the ghosts' come with the exchange
            m64, r64, a64, p64 = pre[own].astype(np.float64), r[own].astype(np.float64), Ap[own].astype(np.float64), p[own].astype(np.float64)
            hs = self._exchange([(p64 * a64).sum(), (m64 * r64 * r64).sum(), (m64 * r64 * a64).sum(), (m64 * a64 * a64).sum()], Ap)
            aD = F(sum(F(h[0]) for h in hs)); n_, s1, s2 = (sum(h[i] for h in hs) for i in (1, 2, 3))
            alpha = aN / aD if aD != 0 else F(0)
            bN = F(max(n_ - 2.0 * float(alpha) * s1 + float(alpha) ** 2 * s2, 0.0))
            beta = bN / aN if aN != 0 else F(0)
            aN = bN
        if L:
            delta = (delta + alpha * p).astype(F)
        self.local[2].reshape(-1)[:] += delta[:3 * self.Nl]                           # owned AND ghost unknowns: the ghosts follow their owners without an exchange
        self.local[3].reshape(-1)[:] += delta[3 * self.Nl:]

    def solve(self, nit, lit):
        costs = [self.cost()]
        for _ in range(nit):
            self.gn_step(lit)
            costs.append(self.cost())
        return costs
