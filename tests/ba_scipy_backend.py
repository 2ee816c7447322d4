"""CPU compute backend for thallo_amd.distributed_ba.BaShardSolver -- TEST INFRASTRUCTURE.
Uses the oracle's CSR export of the rank's sub-instance and scipy sparse products (float64 accumulate, float32 state),
so the sharding / all-reduce logic can run under gloo without a GPU."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import oracle as orc

F = np.float32


class ScipyBaShardBackend:
    def __init__(self, layout, local_params, max_l_iters):
        self.lay = layout
        self.params = [a.copy() for a in local_params]
        cams, pts, obs, oc, op = self.params
        self.Cp, self.P, self.O = cams.shape[0], pts.shape[0], obs.shape[0]
        self.cameras, self.points = torch.from_numpy(cams), torch.from_numpy(pts)       # views: updated in place
        self.nc, self.n = 9 * self.Cp, 9 * self.Cp + 3 * self.P
        self.slot = (self.n + 3) // 4 * 4
        na = self.slot + 8
        z = lambda: torch.zeros(na, dtype=torch.float32)
        self.r, self.pre, self.z, self.delta, self.Ap, self.diag = z(), z(), z(), z(), z(), z()
        self.p = [z(), z()]
        self.S = torch.zeros(2 * max_l_iters + 16, dtype=torch.float32)
        self.T = torch.zeros(8, dtype=torch.float32)

    def _problem(self):
        return orc.Problem(orc.BUNDLE_ADJUST, (self.Cp, self.P, self.O), self.params)

    def cost_local(self, out_idx):
        self.S[out_idx] = self._problem().cost() if self.O else 0.0

    def init_partial(self, cur):
        rp, col, val, res = self._problem().csr()
        self.J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), self.n))
        n = self.n
        self.r.numpy()[:n] = (-(self.J.T @ res.astype(np.float64))).astype(F)
        self.diag.numpy()[:n] = np.asarray(self.J.multiply(self.J).sum(0)).ravel().astype(F)
        self.p[cur].zero_(); self.delta.zero_()

    def point_block(self, vec, with_slot=False):
        return vec[self.nc: (self.slot + 1) if with_slot else self.n]

    def _parts(self, a, b):
        a, b = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
        self.T[0] = float(a[:self.nc] @ b[:self.nc]); self.T[1] = float(a[self.nc:self.n] @ b[self.nc:self.n])

    def init_finish(self):
        n = self.n
        d = self.diag.numpy()[:n]
        m = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        self.pre.numpy()[:n] = m
        self.z.numpy()[:n] = m * self.r.numpy()[:n]
        self._parts(self.r, self.z)

    def _ab(self, first, iN, iD, iB):
        if first:
            return F(0), F(0)
        aN, aD, bN = F(self.S[iN]), F(self.S[iD]), F(self.S[iB])
        return (aN / aD if aD != 0 else F(0)), (bN / aN if aN != 0 else F(0))

    def pupdate(self, cur, first, iN, iD, iB):
        alpha, beta = self._ab(first, iN, iD, iB)
        n = self.n
        pin = self.p[cur].numpy()[:n]
        if not first:
            self.delta.numpy()[:n] += alpha * pin
        self.p[cur ^ 1].numpy()[:n] = self.z.numpy()[:n] + beta * pin

    def apply_partial(self, cur):
        n = self.n
        pv = self.p[cur].numpy()[:n].astype(np.float64)
        ap = (self.J.T @ (self.J @ pv)).astype(F)
        self.Ap.numpy()[:n] = ap
        self.Ap[self.slot] = float(pv[:self.nc] @ ap[:self.nc].astype(np.float64))

    def apply_finish(self, cur, out_idx):
        pv, ap = self.p[cur].numpy().astype(np.float64), self.Ap.numpy().astype(np.float64)
        self.S[out_idx] = float(F(self.Ap[self.slot]) + F(pv[self.nc:self.n] @ ap[self.nc:self.n]))

    def step2(self, iN, iD):
        aN, aD = F(self.S[iN]), F(self.S[iD])
        alpha = aN / aD if aD != 0 else F(0)
        n = self.n
        self.r.numpy()[:n] -= alpha * self.Ap.numpy()[:n]
        self.z.numpy()[:n] = self.pre.numpy()[:n] * self.r.numpy()[:n]
        self._parts(self.z, self.r)

    def linear_update(self, cur, iN, iD, with_p):
        n, nc = self.n, self.nc
        d = self.delta.numpy()[:n].copy()
        if with_p:
            aN, aD = F(self.S[iN]), F(self.S[iD])
            d += (aN / aD if aD != 0 else F(0)) * self.p[cur].numpy()[:n]
        self.params[0].reshape(-1)[:] += d[:nc]
        self.params[1].reshape(-1)[:] += d[nc:]

    def scalar(self, idx):
        return float(self.S[idx])
