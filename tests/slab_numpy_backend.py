"""CPU compute backend for tests/slab_schedule_mirror.py SlabSolver -- TEST INFRASTRUCTURE.

Implements the slab kernel contract of include/thallo_hip.h (owned rows [row0,row1) of a local image
with ghost rows; fused PCGStep1 keeps p current on ghost rows; pack/unpack message layout) in vectorised
numpy float32, so the partition / halo-exchange / rank-ordered-sum logic of the multi-GPU driver can be
exercised under gloo with world_size 2 on a machine without GPUs.  A third, independent statement of
the image_warping gather formulas (after the C row-form oracle and the HIP kernels)."""
import numpy as np
import torch

F = np.float32


class NumpySlabBackend:
    def __init__(self, W, layout, local_params, max_l_iters, one_kernel=False):
        # one_kernel: mirror the shipped multi-GPU schedule (thallo_hip_iw_pcg_iter + ONE all-gather per PCG iteration of
        # [alphaD | N, S1, S2 | boundary rows of Ap]); else the two-kernel / two-collective form
        self.one_kernel_collective = bool(one_kernel)
        self.W, self.lay = W, layout
        self.Hl, self.row0, self.row1 = layout.Hl, layout.row0, layout.row1
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy())
        self.offset, self.angle, self.urshape, self.constraints, self.mask = [t(a) for a in local_params[:5]]
        self.w_fit, self.w_reg = F(local_params[5]), F(local_params[6])
        N = W * self.Hl
        self.N, self.n = N, 3 * N
        z = lambda: torch.zeros(self.n, dtype=torch.float32)
        self.r, self.pre, self.z, self.delta, self.Ap = z(), z(), z(), z(), z()
        self.p = [z(), z()]
        self.S = torch.zeros(2 * max_l_iters + 8, dtype=torch.float32)
        self.msg = 1 + 6 * W
        self.send = torch.zeros(self.msg, dtype=torch.float32)
        self.gath = torch.zeros(layout.world * self.msg, dtype=torch.float32)
        self.local_sum = F(0)
        self.Ap_pp = [self.Ap, z()]                      # Ap ping-pongs in the one-kernel schedule
        self.msg_iter = 7 + 6 * W
        self.send_iter = torch.zeros(self.msg_iter, dtype=torch.float32)
        self.gath_iter = torch.zeros(layout.world * self.msg_iter, dtype=torch.float32)

    # flat vector <-> planes (views into the torch storage)
    def _planes(self, v):
        a = v.numpy()
        return a[:2 * self.N].reshape(self.Hl, self.W, 2), a[2 * self.N:].reshape(self.Hl, self.W)

    def _consts(self):
        act = (self.mask.numpy() == 0)
        a = self.angle.numpy()
        return act, np.cos(a).astype(F), np.sin(a).astype(F), self.urshape.numpy()

    @staticmethod
    def _shift(a, dy, dx, fill=0):
        """b[y,x] = a[y+dy, x+dx] (fill outside)"""
        out = np.full_like(a, fill)
        H, W = a.shape[:2]
        ys = slice(max(0, -dy), min(H, H - dy)); xs = slice(max(0, -dx), min(W, W - dx))
        yd = slice(max(0, dy), min(H, H + dy)); xd = slice(max(0, dx), min(W, W + dx))
        out[ys, xs] = a[yd, xd]
        return out

    def _owned(self, a):
        return a[self.row0:self.row1]

    def cost_local(self, out_idx):
        act, c, s, u = self._consts()
        o = self.offset.numpy(); cons = self.constraints.numpy()
        tot = np.zeros((self.Hl, self.W), F)
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            v = act & self._shift(act, dy, dx, False)
            du = u - self._shift(u, dy, dx); do = o - self._shift(o, dy, dx)
            ex = self.w_reg * (do[..., 0] - (c * du[..., 0] - s * du[..., 1]))
            ey = self.w_reg * (do[..., 1] - (s * du[..., 0] + c * du[..., 1]))
            tot += np.where(v, ex * ex + ey * ey, 0).astype(F)
        vf = act & (cons[..., 0] >= 0) & (cons[..., 1] >= 0)
        f = self.w_fit * (o - cons)
        tot += np.where(vf, (f * f).sum(-1), 0).astype(F)
        self.S[out_idx] = float(np.sum(self._owned(0.5 * tot), dtype=np.float64))

    def init(self, cur):
        act, c, s, u = self._consts()
        o = self.offset.numpy(); cons = self.constraints.numpy()
        wr2, wf2 = self.w_reg * self.w_reg, self.w_fit * self.w_fit
        jx = np.zeros((self.Hl, self.W), F); jy = jx.copy(); ja = jx.copy(); dgo = jx.copy(); dga = jx.copy()
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            v = act & self._shift(act, dy, dx, False)
            du = u - self._shift(u, dy, dx); do = o - self._shift(o, dy, dx)
            cj, sj = self._shift(c, dy, dx), self._shift(s, dy, dx)
            eix = do[..., 0] - (c * du[..., 0] - s * du[..., 1]); eiy = do[..., 1] - (s * du[..., 0] + c * du[..., 1])
            ejx = -do[..., 0] + (cj * du[..., 0] - sj * du[..., 1]); ejy = -do[..., 1] + (sj * du[..., 0] + cj * du[..., 1])
            gix = -s * du[..., 0] - c * du[..., 1]; giy = c * du[..., 0] - s * du[..., 1]
            jx += np.where(v, eix - ejx, 0); jy += np.where(v, eiy - ejy, 0); ja -= np.where(v, gix * eix + giy * eiy, 0)
            dgo += np.where(v, F(2), 0); dga += np.where(v, gix * gix + giy * giy, 0)
        jx *= wr2; jy *= wr2; ja *= wr2; dgo *= wr2; dga *= wr2
        vf = act & (cons[..., 0] >= 0) & (cons[..., 1] >= 0)
        jx += np.where(vf, wf2 * (o[..., 0] - cons[..., 0]), 0); jy += np.where(vf, wf2 * (o[..., 1] - cons[..., 1]), 0)
        dgo += np.where(vf, wf2, 0)
        inv = lambda d: (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        mo = np.where(act, inv(dgo), 0).astype(F); ma = np.where(act, inv(dga), 0).astype(F)
        ro, ra = self._planes(self.r); po_, pa_ = self._planes(self.pre); zo, za = self._planes(self.z)
        R0, R1 = self.row0, self.row1
        ro[R0:R1, :, 0] = np.where(act, -jx, 0)[R0:R1]; ro[R0:R1, :, 1] = np.where(act, -jy, 0)[R0:R1]; ra[R0:R1] = np.where(act, -ja, 0)[R0:R1]
        po_[R0:R1, :, 0] = mo[R0:R1]; po_[R0:R1, :, 1] = mo[R0:R1]; pa_[R0:R1] = ma[R0:R1]
        zo[R0:R1] = po_[R0:R1] * ro[R0:R1]; za[R0:R1] = pa_[R0:R1] * ra[R0:R1]
        self.p[cur].zero_(); self.delta.zero_()
        self._fitvalid = vf
        self.local_sum = F(np.sum((ro[R0:R1] * zo[R0:R1]).sum(-1) + ra[R0:R1] * za[R0:R1], dtype=np.float64))

    def _alpha_beta(self, first, iN, iD, iB):
        if first:
            return F(0), F(0)
        aN, aD, bN = F(self.S[iN]), F(self.S[iD]), F(self.S[iB])
        alpha = aN / aD if aD != 0 else F(0)
        beta = bN / aN if aN != 0 else F(0)
        return alpha, beta

    def step1(self, cur, first, iN, iD, iB, out_idx):
        alpha, beta = self._alpha_beta(first, iN, iD, iB)
        R0, R1 = self.row0, self.row1
        zo, za = self._planes(self.z); po, pa = self._planes(self.p[cur]); qo, qa = self._planes(self.p[cur ^ 1])
        do_, da_ = self._planes(self.delta); Ao, Aa = self._planes(self.Ap)
        if not first:
            do_[R0:R1] += alpha * po[R0:R1]; da_[R0:R1] += alpha * pa[R0:R1]
        qo[:] = zo + beta * po; qa[:] = za + beta * pa          # all rows incl. ghosts: p stays current on ghost rows
        act, c, s, u = self._consts()
        wr2, wf2 = self.w_reg * self.w_reg, self.w_fit * self.w_fit
        ax = np.zeros((self.Hl, self.W), F); ay = ax.copy(); aa = ax.copy()
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            v = act & self._shift(act, dy, dx, False)
            du = u - self._shift(u, dy, dx)
            cj, sj = self._shift(c, dy, dx), self._shift(s, dy, dx)
            gix = -s * du[..., 0] - c * du[..., 1]; giy = c * du[..., 0] - s * du[..., 1]
            gjx = sj * du[..., 0] + cj * du[..., 1]; gjy = -cj * du[..., 0] + sj * du[..., 1]
            dp = qo - self._shift(qo, dy, dx); paj = self._shift(qa, dy, dx)
            ex = dp[..., 0] - gix * qa; ey = dp[..., 1] - giy * qa
            ax += np.where(v, dp[..., 0] + ex + gjx * paj, 0); ay += np.where(v, dp[..., 1] + ey + gjy * paj, 0)
            aa -= np.where(v, gix * ex + giy * ey, 0)
        ax *= wr2; ay *= wr2; aa *= wr2
        ax += np.where(self._fitvalid, wf2 * qo[..., 0], 0); ay += np.where(self._fitvalid, wf2 * qo[..., 1], 0)
        Ao[R0:R1, :, 0] = np.where(act, ax, 0)[R0:R1]; Ao[R0:R1, :, 1] = np.where(act, ay, 0)[R0:R1]; Aa[R0:R1] = np.where(act, aa, 0)[R0:R1]
        self.S[out_idx] = float(np.sum((qo[R0:R1] * Ao[R0:R1]).sum(-1) + qa[R0:R1] * Aa[R0:R1], dtype=np.float64))

    def step2(self, iN, iD):
        aN, aD = F(self.S[iN]), F(self.S[iD])
        alpha = aN / aD if aD != 0 else F(0)
        R0, R1 = self.row0, self.row1
        ro, ra = self._planes(self.r); Ao, Aa = self._planes(self.Ap); mo, ma = self._planes(self.pre); zo, za = self._planes(self.z)
        ro[R0:R1] -= alpha * Ao[R0:R1]; ra[R0:R1] -= alpha * Aa[R0:R1]
        zo[R0:R1] = mo[R0:R1] * ro[R0:R1]; za[R0:R1] = ma[R0:R1] * ra[R0:R1]
        self.local_sum = F(np.sum((zo[R0:R1] * ro[R0:R1]).sum(-1) + za[R0:R1] * ra[R0:R1], dtype=np.float64))

    def pack(self):
        zo, za = self._planes(self.z)
        W = self.W
        m = self.send.numpy()
        m[0] = self.local_sum
        m[1:1 + 2 * W] = zo[self.row0].reshape(-1); m[1 + 2 * W:1 + 3 * W] = za[self.row0]
        m[1 + 3 * W:1 + 5 * W] = zo[self.row1 - 1].reshape(-1); m[1 + 5 * W:] = za[self.row1 - 1]

    def unpack(self, out_idx, gathered):
        g = gathered.numpy().reshape(-1, self.msg)
        tot = F(0)
        for r in range(g.shape[0]):
            tot = F(tot + g[r, 0])
        self.S[out_idx] = float(tot)
        zo, za = self._planes(self.z)
        W, lay = self.W, self.lay
        if lay.top:
            src = g[lay.rank - 1, 1 + 3 * W:]
            zo[self.row0 - 1] = src[:2 * W].reshape(W, 2); za[self.row0 - 1] = src[2 * W:]
        if lay.bot:
            src = g[lay.rank + 1, 1:1 + 3 * W]
            zo[self.row1] = src[:2 * W].reshape(W, 2); za[self.row1] = src[2 * W:]

    # -- once per GN step (one-kernel schedule): the ghost rows need their owner's M^-1 (it depends on rows this rank does not hold);
    #    the HIP backend ships the flags byte instead and recomputes M^-1 from it
    def pack_grid_info(self):
        """[first owned row | last owned row] x [M^-1 (3W) | r_0 (3W)]"""
        po_, pa_ = self._planes(self.pre); ro, ra = self._planes(self.r)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1).copy())
        return torch.cat([t(x[y]) for y in (self.row0, self.row1 - 1) for x in (po_, pa_, ro, ra)])

    def unpack_grid_info(self, g):
        W, lay = self.W, self.lay
        po_, pa_ = self._planes(self.pre); ro, ra = self._planes(self.r)
        g = g.numpy().reshape(lay.world, 2, 6 * W)
        for cond, ghost, src in ((lay.top, self.row0 - 1, (lay.rank - 1, 1)), (lay.bot, self.row1, (lay.rank + 1, 0))):
            if cond:
                m = g[src[0], src[1]]
                po_[ghost] = m[:2 * W].reshape(W, 2); pa_[ghost] = m[2 * W:3 * W]
                ro[ghost] = m[3 * W:5 * W].reshape(W, 2); ra[ghost] = m[5 * W:]

    def iter_collective(self, cur, mode, iN, iD, iB, jD, jB, iN2, iD2, allgather):
        """The one-kernel PCG iteration (csrc/energy_image_warping.hip k_iter) + its single exchange, in numpy:
        r = r - alpha Ap (owned AND ghost rows: Ap's ghost rows came with the previous exchange), p = M^-1 r + beta p, delta update,
        Ap' = J^T J p on the owned rows, message [alphaD | N, S1, S2 as (hi, lo) words | first, last owned row of Ap'],
        betaN = N - 2 alpha S1 + alpha^2 S2 from the rank-ordered sums."""
        first = bool(int(mode) & 1)
        assert (int(mode) >> 1) == 0, "the numpy mirror updates delta every iteration"
        alpha, beta = self._alpha_beta(first, iN, iD, iB)
        R0, R1 = self.row0, self.row1
        ro, ra = self._planes(self.r); mo, ma = self._planes(self.pre)
        Ai_o, Ai_a = self._planes(self.Ap_pp[cur]); Ao, Aa = self._planes(self.Ap_pp[cur ^ 1])
        po, pa = self._planes(self.p[cur]); qo, qa = self._planes(self.p[cur ^ 1]); do_, da_ = self._planes(self.delta)
        if not first:
            ro -= alpha * Ai_o; ra -= alpha * Ai_a
            do_[R0:R1] += alpha * po[R0:R1]; da_[R0:R1] += alpha * pa[R0:R1]
        qo[:] = mo * ro + beta * po; qa[:] = ma * ra + beta * pa
        act, c, s, u = self._consts()
        wr2, wf2 = self.w_reg * self.w_reg, self.w_fit * self.w_fit
        ax = np.zeros((self.Hl, self.W), F); ay = ax.copy(); aa = ax.copy()
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            v = act & self._shift(act, dy, dx, False)
            du = u - self._shift(u, dy, dx)
            cj, sj = self._shift(c, dy, dx), self._shift(s, dy, dx)
            gix = -s * du[..., 0] - c * du[..., 1]; giy = c * du[..., 0] - s * du[..., 1]
            gjx = sj * du[..., 0] + cj * du[..., 1]; gjy = -cj * du[..., 0] + sj * du[..., 1]
            dp = qo - self._shift(qo, dy, dx); paj = self._shift(qa, dy, dx)
            ex = dp[..., 0] - gix * qa; ey = dp[..., 1] - giy * qa
            ax += np.where(v, dp[..., 0] + ex + gjx * paj, 0); ay += np.where(v, dp[..., 1] + ey + gjy * paj, 0)
            aa -= np.where(v, gix * ex + giy * ey, 0)
        ax *= wr2; ay *= wr2; aa *= wr2
        ax += np.where(self._fitvalid, wf2 * qo[..., 0], 0); ay += np.where(self._fitvalid, wf2 * qo[..., 1], 0)
        Ao[R0:R1, :, 0] = np.where(act, ax, 0)[R0:R1]; Ao[R0:R1, :, 1] = np.where(act, ay, 0)[R0:R1]; Aa[R0:R1] = np.where(act, aa, 0)[R0:R1]
        d = np.float64
        own = lambda a: a[R0:R1].astype(d)
        aD = F(np.sum((own(qo) * own(Ao)).sum(-1) + own(qa) * own(Aa)))
        q3 = [np.sum((own(mo) * own(x) * own(y)).sum(-1) + own(ma) * own(xa) * own(ya))
              for (x, xa, y, ya) in ((ro, ra, ro, ra), (ro, ra, Ao, Aa), (Ao, Aa, Ao, Aa))]
        W, lay, msg = self.W, self.lay, self.msg_iter
        m = self.send_iter.numpy()
        m[0] = aD
        m[1:7] = np.array(q3, dtype=np.float64).view(np.uint32).reshape(3, 2)[:, ::-1].reshape(-1).view(np.float32)      # (hi, lo) words
        m[7:7 + 2 * W] = Ao[R0].reshape(-1); m[7 + 2 * W:7 + 3 * W] = Aa[R0]
        m[7 + 3 * W:7 + 5 * W] = Ao[R1 - 1].reshape(-1); m[7 + 5 * W:] = Aa[R1 - 1]
        if allgather is not None:
            allgather(self.send_iter, self.gath_iter)
            g = self.gath_iter.numpy().reshape(-1, msg)
        else:
            g = m.reshape(1, msg)
        gad, gq = F(0), np.zeros(3, np.float64)
        for r in range(g.shape[0]):
            gad = F(gad + g[r, 0])
            w = g[r, 1:7].view(np.uint32).astype(np.uint64).reshape(3, 2)
            gq += ((w[:, 0] << np.uint64(32)) | w[:, 1]).view(np.float64)
        aN = F(self.S[iB])
        al = aN / gad if gad != 0 else F(0)
        bn = gq[0] - 2.0 * float(al) * gq[1] + float(al) * float(al) * gq[2]
        self.S[jD] = float(gad); self.S[jB] = float(F(max(bn, 0.0)))
        if lay.top:
            src = g[lay.rank - 1, 7 + 3 * W:]
            Ao[R0 - 1] = src[:2 * W].reshape(W, 2); Aa[R0 - 1] = src[2 * W:]
        if lay.bot:
            src = g[lay.rank + 1, 7:7 + 3 * W]
            Ao[R1] = src[:2 * W].reshape(W, 2); Aa[R1] = src[2 * W:]

    def linear_update(self, cur, iN, iD, with_p):
        R0, R1 = self.row0, self.row1
        do_, da_ = self._planes(self.delta)
        o = self.offset.numpy(); a = self.angle.numpy()
        if with_p:
            aN, aD = F(self.S[iN]), F(self.S[iD])
            alpha = aN / aD if aD != 0 else F(0)
            po, pa = self._planes(self.p[cur])
            o[R0:R1] += do_[R0:R1] + alpha * po[R0:R1]; a[R0:R1] += da_[R0:R1] + alpha * pa[R0:R1]
        else:
            o[R0:R1] += do_[R0:R1]; a[R0:R1] += da_[R0:R1]

    def pack_unknowns(self):
        o, a = self.offset.view(self.Hl, 2 * self.W), self.angle.view(self.Hl, self.W)
        return torch.cat([o[self.row0], a[self.row0], o[self.row1 - 1], a[self.row1 - 1]])

    def unpack_unknowns(self, g):
        W, lay = self.W, self.lay
        o, a = self.offset.view(self.Hl, 2 * W), self.angle.view(self.Hl, W)
        g = g.view(lay.world, 2, 3 * W)
        if lay.top:
            o[self.row0 - 1].copy_(g[lay.rank - 1, 1, :2 * W]); a[self.row0 - 1].copy_(g[lay.rank - 1, 1, 2 * W:])
        if lay.bot:
            o[self.row1].copy_(g[lay.rank + 1, 0, :2 * W]); a[self.row1].copy_(g[lay.rank + 1, 0, 2 * W:])

    def scalar(self, idx):
        return float(self.S[idx])
