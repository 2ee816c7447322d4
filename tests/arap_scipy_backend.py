"""CPU compute backend for thallo_amd.distributed_graph.GraphPartSolver -- TEST INFRASTRUCTURE (oracle CSR + scipy)."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import oracle as orc

F = np.float32


class ScipyArapPartBackend:
    def __init__(self, part, params, max_l_iters):
        self.part = part
        self.params = [a.copy() if isinstance(a, np.ndarray) else a for a in params]
        N, E = self.params[2].shape[0], self.params[6].shape[0]
        self.N, self.E, self.n = N, E, 6 * N
        z = lambda: torch.zeros(self.n, dtype=torch.float32)
        self.r, self.pre, self.z, self.delta, self.Ap = z(), z(), z(), z(), z()
        self.p = [z(), z()]
        self.S = torch.zeros(2 * max_l_iters + 8, dtype=torch.float32)
        self.position, self.angle = torch.from_numpy(self.params[2]), torch.from_numpy(self.params[3])
        self.rng = (3 * part.n0, 3 * part.chunk, 3 * N + 3 * part.n0, 3 * part.chunk)
        own = np.zeros(self.n, bool)
        own[self.rng[0]:self.rng[0] + self.rng[1]] = True; own[self.rng[2]:self.rng[2] + self.rng[3]] = True
        self.own = own
        v0 = self.params[6]
        rows_own = np.zeros(3 * N + 3 * E, bool)
        rows_own[3 * part.n0:3 * part.n1] = True                                     # fit rows of owned vertices
        rows_own[3 * N:] = np.repeat((v0 >= part.n0) & (v0 < part.n1), 3)            # reg rows of edges leaving owned vertices
        self.rows_own = rows_own

    def _problem(self):
        return orc.Problem(orc.ARAP_MESH, (self.N, self.E), self.params)

    def cost_local(self, out_idx):
        res = self._problem().csr()[3].astype(np.float64)
        self.S[out_idx] = float(0.5 * (res[self.rows_own] ** 2).sum())

    def init(self, cur, out_idx):
        rp, col, val, res = self._problem().csr()
        self.J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), self.n))
        own = self.own
        r = (-(self.J.T @ res.astype(np.float64))).astype(F)
        d = np.asarray(self.J.multiply(self.J).sum(0)).ravel().astype(F)
        m = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        self.r.numpy()[own] = r[own]; self.pre.numpy()[own] = m[own]; self.z.numpy()[own] = (m * r)[own]
        self.p[cur].zero_(); self.delta.zero_()
        self.S[out_idx] = float((r[own].astype(np.float64) * (m * r)[own]).sum())

    def _ab(self, first, iN, iD, iB):
        if first:
            return F(0), F(0)
        aN, aD, bN = F(self.S[iN]), F(self.S[iD]), F(self.S[iB])
        return (aN / aD if aD != 0 else F(0)), (bN / aN if aN != 0 else F(0))

    def pupdate(self, cur, first, iN, iD, iB):
        alpha, beta = self._ab(first, iN, iD, iB)
        own = self.own
        pin = self.p[cur].numpy()
        if not first:
            self.delta.numpy()[own] += alpha * pin[own]
        self.p[cur ^ 1].numpy()[own] = self.z.numpy()[own] + beta * pin[own]

    def owned_slices(self, vec):
        o0, l0, o1, l1 = self.rng
        return vec[o0:o0 + l0], vec[o1:o1 + l1]

    def full_slices(self, vec):
        return vec[:3 * self.N], vec[3 * self.N:6 * self.N]

    def apply(self, cur, out_idx):
        pv = self.p[cur].numpy().astype(np.float64)
        ap = (self.J.T @ (self.J @ pv)).astype(F)
        self.Ap.numpy()[self.own] = ap[self.own]
        self.S[out_idx] = float((pv[self.own] * ap[self.own]).sum())

    def step2(self, iN, iD, out_idx):
        aN, aD = F(self.S[iN]), F(self.S[iD])
        alpha = aN / aD if aD != 0 else F(0)
        own = self.own
        self.r.numpy()[own] -= alpha * self.Ap.numpy()[own]
        self.z.numpy()[own] = self.pre.numpy()[own] * self.r.numpy()[own]
        self.S[out_idx] = float((self.z.numpy()[own].astype(np.float64) * self.r.numpy()[own]).sum())

    def linear_update(self, cur, iN, iD, with_p):
        own = self.own
        d = self.delta.numpy().copy()
        if with_p:
            aN, aD = F(self.S[iN]), F(self.S[iD])
            d += (aN / aD if aD != 0 else F(0)) * self.p[cur].numpy()
        N = self.N
        self.params[2].reshape(-1)[own[:3 * N]] += d[:3 * N][own[:3 * N]]
        self.params[3].reshape(-1)[own[3 * N:]] += d[3 * N:][own[3 * N:]]

    def unknown_views(self):
        return self.position.view(-1), self.angle.view(-1)

    def scalar(self, idx):
        return float(self.S[idx])
