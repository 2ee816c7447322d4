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
