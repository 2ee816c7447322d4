"""CPU compute backend for the shape_from_shading slab path (tests/slab_schedule_mirror.py SlabSolver, ghost width 2) --
TEST INFRASTRUCTURE.  The rank's local image (owned rows + 2 ghost rows) is handed to the oracle as a stand-alone
problem with u_y shifted by the slab's row offset; rows/columns of its Jacobian that belong to the owned pixels are
exact (their stencils stay inside the local image), everything else is masked out."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import oracle as orc

F = np.float32


class ScipySfsSlabBackend:
    def __init__(self, W, layout, local_params, H_global, max_l_iters):
        self.W, self.lay = W, layout
        self.Hl, self.row0, self.row1 = layout.Hl, layout.row0, layout.row1
        yoff = layout.g0 - layout.top
        self.params = [a.copy() if isinstance(a, np.ndarray) else a for a in local_params]
        self.params[6] = float(self.params[6]) - yoff                 # u_y in local row coordinates
        self.X = torch.from_numpy(self.params[16])
        N = W * self.Hl
        self.N = N
        z = lambda: torch.zeros(N, dtype=torch.float32)
        self.r, self.z, self.delta, self.Ap = z(), z(), z(), z()
        self.p = [z(), z()]
        self.S = torch.zeros(2 * max_l_iters + 8, dtype=torch.float32)
        g = layout.ghost
        self.msg = 1 + 2 * g * W
        self.send = torch.zeros(self.msg, dtype=torch.float32)
        self.gath = torch.zeros(layout.world * self.msg, dtype=torch.float32)
        own = np.zeros((self.Hl, W), bool); own[self.row0:self.row1] = True
        self.own = own.reshape(-1)
        self.rows_own = np.repeat(self.own, 6)                         # 6 residual rows per pixel
        ext = np.zeros((self.Hl, W), bool); ext[self.row0 - layout.top:self.row1 + layout.bot] = True
        self.ext = ext.reshape(-1)
        self.local_sum = F(0)

    def _problem(self):
        return orc.Problem(orc.SFS, (self.W, self.Hl), self.params)

    def cost_local(self, out_idx):
        res = self._problem().csr()[3].astype(np.float64)
        self.S[out_idx] = float(0.5 * (res[self.rows_own] ** 2).sum())

    def init(self, cur):
        rp, col, val, res = self._problem().csr()
        self.J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), self.N))
        r = (-(self.J.T @ res.astype(np.float64))).astype(F)
        own = self.own
        self.r.numpy()[own] = r[own]; self.z.numpy()[own] = r[own]
        self.p[cur].zero_(); self.delta.zero_()
        self.local_sum = F((r[own].astype(np.float64) ** 2).sum())

    def _ab(self, first, iN, iD, iB):
        if first:
            return F(0), F(0)
        aN, aD, bN = F(self.S[iN]), F(self.S[iD]), F(self.S[iB])
        return (aN / aD if aD != 0 else F(0)), (bN / aN if aN != 0 else F(0))

    def step1(self, cur, first, iN, iD, iB, out_idx):
        alpha, beta = self._ab(first, iN, iD, iB)
        ext, own = self.ext, self.own
        pin = self.p[cur].numpy()
        if not first:
            self.delta.numpy()[ext] += alpha * pin[ext]
        self.p[cur ^ 1].numpy()[ext] = self.z.numpy()[ext] + beta * pin[ext]
        pv = self.p[cur ^ 1].numpy().astype(np.float64)
        ap = (self.J.T @ (self.J @ pv)).astype(F)
        self.Ap.numpy()[own] = ap[own]
        self.S[out_idx] = float((pv[own] * ap[own]).sum())

    def step2(self, iN, iD):
        aN, aD = F(self.S[iN]), F(self.S[iD])
        alpha = aN / aD if aD != 0 else F(0)
        own = self.own
        self.r.numpy()[own] -= alpha * self.Ap.numpy()[own]
        self.z.numpy()[own] = self.r.numpy()[own]
        self.local_sum = F((self.r.numpy()[own].astype(np.float64) ** 2).sum())

    def pack(self):
        g, W = self.lay.ghost, self.W
        zz = self.z.view(self.Hl, W)
        m = self.send.numpy()
        m[0] = self.local_sum
        m[1:1 + g * W] = zz[self.row0:self.row0 + g].reshape(-1).numpy(); m[1 + g * W:] = zz[self.row1 - g:self.row1].reshape(-1).numpy()

    def unpack(self, out_idx, gathered):
        lay, g, W = self.lay, self.lay.ghost, self.W
        gv = gathered.numpy().reshape(-1, self.msg)
        tot = F(0)
        for r_ in range(gv.shape[0]):
            tot = F(tot + gv[r_, 0])
        self.S[out_idx] = float(tot)
        zz = self.z.view(self.Hl, W).numpy()
        if lay.top:
            zz[self.row0 - g:self.row0] = gv[lay.rank - 1, 1 + g * W:].reshape(g, W)
        if lay.bot:
            zz[self.row1:self.row1 + g] = gv[lay.rank + 1, 1:1 + g * W].reshape(g, W)

    def linear_update(self, cur, iN, iD, with_p):
        own = self.own
        d = self.delta.numpy().copy()
        if with_p:
            aN, aD = F(self.S[iN]), F(self.S[iD])
            d += (aN / aD if aD != 0 else F(0)) * self.p[cur].numpy()
        self.params[16].reshape(-1)[own] += d[own]

    def pack_unknowns(self):
        g, X = self.lay.ghost, self.X.view(self.Hl, self.W)
        return torch.cat([X[self.row0:self.row0 + g].reshape(-1), X[self.row1 - g:self.row1].reshape(-1)])

    def unpack_unknowns(self, gath):
        lay, g, W = self.lay, self.lay.ghost, self.W
        X = self.X.view(self.Hl, W)
        gv = gath.view(lay.world, 2, g * W)
        if lay.top:
            X[self.row0 - g:self.row0].copy_(gv[lay.rank - 1, 1].view(g, W))
        if lay.bot:
            X[self.row1:self.row1 + g].copy_(gv[lay.rank + 1, 0].view(g, W))

    def scalar(self, idx):
        return float(self.S[idx])
