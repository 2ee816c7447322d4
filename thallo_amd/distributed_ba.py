"""Camera-sharded multi-GPU driver for bundle adjustment on the materialized sparse-J path.

Rank k owns a contiguous range of cameras and every observation (row pair of J) of those cameras; the points are
replicated.  Then
  * J p is local (an observation needs its own camera and its point, both present);
  * the camera block of J^T(Jp) is complete locally; the POINT block is a partial sum over the rank's observations
    -> one all-reduce of 3P floats (+1 piggy-backed scalar: the rank's camera part of p.Ap) per PCG iteration;
  * after that every rank holds identical point blocks of Ap, r, z, p, delta and updates them redundantly, so the point
    unknowns never need a broadcast; the second per-iteration collective is the 1-float all-reduce of the camera part
    of betaN.
The kernels are the single-GPU ones (include/thallo_hip.h: thallo_hip_ba_*) run on the local sub-instance
[cameras of this rank (padded to a multiple of 4) | all points].  Transport: torch.distributed (nccl = RCCL on GPUs,
gloo in the CPU tests, which drive this class with a scipy compute backend).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import api


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
        return [cl, np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(obs[sel]),
                np.ascontiguousarray(oc[sel] - self.c0, np.int32), np.ascontiguousarray(op[sel], np.int32)]


class HipBaShardBackend:
    """The gfx950 BA kernels on this rank's sub-instance."""

    def __init__(self, layout, local_params, max_l_iters):
        self.L = api.lib()
        L = self.L
        vp, ci, cl, fl = C.c_void_p, C.c_int, C.c_long, C.c_float
        L.thallo_hip_ba_cost.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        L.thallo_hip_ba_compute_j.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.thallo_hip_ba_pcg_init.argtypes = [ci, ci] + [vp] * 15
        L.thallo_hip_ba_apply_jtj.argtypes = [ci, ci] + [vp] * 10
        L.thallo_hip_pcg_init_finish.argtypes = [vp, vp, vp, vp, cl, ci, vp, vp]
        L.thallo_hip_dot.argtypes = [vp, vp, cl, vp, vp]
        L.thallo_hip_pcg_pupdate.argtypes = [vp, vp, vp, vp, cl, ci, api.SumT, api.SumT, api.SumT, vp]
        self.lay = layout
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        cams, pts, obs, oc, op = local_params
        self.Cp, self.P, self.O = cams.shape[0], pts.shape[0], obs.shape[0]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.cameras, self.points, self.obs, self.oToC, self.oToP = t(cams), t(pts), t(obs), t(oc), t(op)
        # incidence lists of the local sub-instance (same construction as BundleAdjustmentPlugin::prepare)
        O, Cp, P = self.O, self.Cp, self.P
        order = np.argsort(oc, kind="stable")
        cam_ptr = np.concatenate([[0], np.cumsum(np.bincount(oc, minlength=Cp))]).astype(np.int32)
        pos = np.empty(O, np.int64); pos[order] = np.arange(O)
        q_cam, q_pt = oc[order].astype(np.int32), op[order].astype(np.int32)
        porder = np.argsort(op, kind="stable")
        pt_ptr = np.concatenate([[0], np.cumsum(np.bincount(op, minlength=P))]).astype(np.int32)
        pt_pos = pos[porder].astype(np.int32)
        pad = lambda a: np.concatenate([a, np.zeros(4, a.dtype)])
        self.cam_ptr, self.cam_obs, self.q_cam, self.q_pt = t(pad(cam_ptr)), t(pad(order.astype(np.int32))), t(pad(q_cam)), t(pad(q_pt))
        self.pt_ptr, self.pt_pos = t(pad(pt_ptr)), t(pad(pt_pos))
        self.Jb = torch.zeros(24 * O + 64, dtype=torch.float32, device=dev)
        self.F = torch.zeros(2 * O + 64, dtype=torch.float32, device=dev)
        self.nc, self.n = 9 * Cp, 9 * Cp + 3 * P
        # scalar slot piggy-backed on the point-block all-reduce: the first 16-byte-aligned index at/after n, which no
        # float4 kernel running over n elements ever touches (the floats between n and it stay 0)
        self.slot = (self.n + 3) // 4 * 4
        na = (self.slot + 4 + 255) // 256 * 256
        z = lambda: torch.zeros(na, dtype=torch.float32, device=dev)
        self.r, self.pre, self.z, self.delta, self.Ap, self.diag = z(), z(), z(), z(), z(), z()
        self.p = [z(), z()]
        self.parts = torch.zeros(4 * 1024, dtype=torch.float32, device=dev)
        self.S = torch.zeros(2 * max_l_iters + 16, dtype=torch.float32, device=dev)
        self.T = torch.zeros(8, dtype=torch.float32, device=dev)          # scratch scalars: [cam part, point part, ...]

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _sum(self, idx):
        return api.SumT(self.S.data_ptr() + 4 * idx, 1)

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed with hipError {-rc}")
        return rc

    def _dot_to(self, a, b, off, length, out_tensor, out_idx):
        """out_tensor[out_idx] = sum a[off:off+length] * b[off:off+length]"""
        vp = C.c_void_p
        nb = self._chk(self.L.thallo_hip_dot(vp(a.data_ptr() + 4 * off), vp(b.data_ptr() + 4 * off), C.c_long(length), vp(self.parts.data_ptr()), self._st()), "dot")
        self._chk(self.L.thallo_hip_finish_sum(api.SumT(self.parts.data_ptr(), nb), vp(out_tensor.data_ptr() + 4 * out_idx), self._st()), "finish_sum")

    # ---- phases
    def cost_local(self, out_idx):
        vp = C.c_void_p
        nb = self._chk(self.L.thallo_hip_ba_cost(self.Cp, self.P, self.O, vp(self.cameras.data_ptr()), vp(self.points.data_ptr()), vp(self.obs.data_ptr()),
                                                 vp(self.oToC.data_ptr()), vp(self.oToP.data_ptr()), vp(self.parts.data_ptr()), self._st()), "ba_cost")
        self._chk(self.L.thallo_hip_finish_sum(api.SumT(self.parts.data_ptr(), nb), vp(self.S.data_ptr() + 4 * out_idx), self._st()), "finish_sum")

    def init_partial(self, cur):
        """J blocks; r = -J^T F and raw diag (point blocks: partial sums over this rank's observations); p = delta = 0"""
        vp = C.c_void_p
        self._chk(self.L.thallo_hip_ba_compute_j(self.O, vp(self.cameras.data_ptr()), vp(self.points.data_ptr()), vp(self.obs.data_ptr()),
                                                 vp(self.cam_obs.data_ptr()), vp(self.q_cam.data_ptr()), vp(self.q_pt.data_ptr()),
                                                 vp(self.Jb.data_ptr()), vp(self.F.data_ptr()), self._st()), "ba_compute_j")
        self._chk(self.L.thallo_hip_ba_pcg_init(self.Cp, self.P, vp(self.cam_ptr.data_ptr()), vp(self.q_pt.data_ptr()), vp(self.pt_ptr.data_ptr()),
                                                vp(self.pt_pos.data_ptr()), vp(self.q_cam.data_ptr()), vp(self.Jb.data_ptr()), vp(self.F.data_ptr()),
                                                vp(self.r.data_ptr()), vp(self.pre.data_ptr()), vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()),
                                                vp(self.delta.data_ptr()), vp(self.diag.data_ptr()), vp(self.parts.data_ptr()), self._st()), "ba_pcg_init")

    def point_block(self, vec, with_slot=False):
        return vec[self.nc: (self.slot + 1) if with_slot else self.n]

    def init_finish(self):
        """after the point blocks of r and diag were all-reduced: pre, z; T[0] = camera part of r.z, T[1] = point part"""
        vp = C.c_void_p
        self._chk(self.L.thallo_hip_pcg_init_finish(vp(self.r.data_ptr()), vp(self.diag.data_ptr()), vp(self.pre.data_ptr()), vp(self.z.data_ptr()),
                                                    C.c_long(self.n), 1, vp(self.parts.data_ptr() + 4096 * 2), self._st()), "pcg_init_finish")
        self._dot_to(self.r, self.z, 0, self.nc, self.T, 0)
        self._dot_to(self.r, self.z, self.nc, 3 * self.P, self.T, 1)

    def pupdate(self, cur, first, iN, iD, iB):
        vp = C.c_void_p
        self._chk(self.L.thallo_hip_pcg_pupdate(vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()), vp(self.p[cur ^ 1].data_ptr()), vp(self.delta.data_ptr()),
                                                C.c_long(self.n), 1 if first else 0, self._sum(iN), self._sum(iD), self._sum(iB), self._st()), "pcg_pupdate")

    def apply_partial(self, cur):
        """Ap = J^T(J p) of this rank's observations; Ap[slot] (behind the point block) = camera part of p.Ap"""
        vp = C.c_void_p
        self._chk(self.L.thallo_hip_ba_apply_jtj(self.Cp, self.P, vp(self.cam_ptr.data_ptr()), vp(self.q_pt.data_ptr()), vp(self.pt_ptr.data_ptr()),
                                                 vp(self.pt_pos.data_ptr()), vp(self.q_cam.data_ptr()), vp(self.Jb.data_ptr()), vp(self.p[cur].data_ptr()),
                                                 vp(self.Ap.data_ptr()), vp(self.parts.data_ptr()), self._st()), "ba_apply_jtj")
        self._dot_to(self.p[cur], self.Ap, 0, self.nc, self.Ap, self.slot)

    def apply_finish(self, cur, out_idx):
        """after the all-reduce of Ap[nc : slot+1]: S[out] = (sum of camera parts) + p_pt.Ap_pt"""
        self._dot_to(self.p[cur], self.Ap, self.nc, 3 * self.P, self.T, 2)
        torch.add(self.Ap[self.slot], self.T[2], out=self.S[out_idx])

    def step2(self, iN, iD):
        """r -= alpha Ap; z = pre r; T[0] = camera part of z.r, T[1] = point part"""
        vp = C.c_void_p
        api.lib().thallo_hip_pcg_step2(vp(self.r.data_ptr()), vp(self.Ap.data_ptr()), vp(self.pre.data_ptr()), vp(self.z.data_ptr()), C.c_long(self.n),
                                       self._sum(iN), self._sum(iD), vp(self.parts.data_ptr() + 4096 * 2), self._st())
        self._dot_to(self.z, self.r, 0, self.nc, self.T, 0)
        self._dot_to(self.z, self.r, self.nc, 3 * self.P, self.T, 1)

    def linear_update(self, cur, iN, iD, with_p):
        vp = C.c_void_p
        for X, off, ln in ((self.cameras, 0, self.nc), (self.points, self.nc, 3 * self.P)):
            p_ptr = vp(self.p[cur].data_ptr() + 4 * off) if with_p else None
            self._chk(self.L.thallo_hip_linear_update(vp(X.data_ptr()), vp(self.delta.data_ptr() + 4 * off), p_ptr, C.c_long(ln),
                                                      self._sum(iN), self._sum(iD), self._st()), "linear_update")

    def scalar(self, idx):
        return float(self.S[idx].item())


class BaShardSolver:
    """GN + PCG over camera shards; the same recurrences as the single-GPU driver (gauss_newton.t:1545-1785)."""

    def __init__(self, backend, layout, group=None):
        self.be, self.lay, self.group = backend, layout, group
        self.world = layout.world

    def _ar(self, t):
        if self.world > 1:
            dist.all_reduce(t, group=self.group)

    def cost(self):
        self.be.cost_local(0)
        self._ar(self.be.S[0:1])
        return self.be.scalar(0)

    def _scalar_from_T(self, out_idx):
        """S[out] = all-reduce(camera part T[0]) + point part T[1] (identical on every rank)"""
        be = self.be
        self._ar(be.T[0:1])
        torch.add(be.T[0], be.T[1], out=be.S[out_idx])

    def gn_step(self, l_iters):
        be = self.be
        B, L = 2, l_iters
        cur = 0
        be.init_partial(cur)
        self._ar(be.point_block(be.r)); self._ar(be.point_block(be.diag))
        be.init_finish()
        self._scalar_from_T(B)                                  # alphaN_0
        for k in range(L):
            jN, jD, jB = B + 2 * k, B + 2 * k + 1, B + 2 * k + 2
            be.pupdate(cur, k == 0, jN - 2 if k else jN, jD - 2 if k else jD, jN)
            cur ^= 1
            be.apply_partial(cur)
            self._ar(be.point_block(be.Ap, with_slot=True))     # 3P floats + the piggy-backed camera dot
            be.apply_finish(cur, jD)                            # alphaD_k
            be.step2(jN, jD)
            self._scalar_from_T(jB)                             # betaN_k
        if L > 0:
            be.linear_update(cur, B + 2 * (L - 1), B + 2 * (L - 1) + 1, True)
        else:
            be.linear_update(cur, B, B, False)

    def solve(self, n_iters, l_iters):
        costs = [self.cost()]
        for _ in range(n_iters):
            self.gn_step(l_iters)
            costs.append(self.cost())
        return costs


def make_hip_ba_solver(params_global, rank, world, max_l_iters):
    lay = BaShardLayout(params_global[0].shape[0], rank, world)
    be = HipBaShardBackend(lay, lay.shard(params_global), max_l_iters)
    return BaShardSolver(be, lay), lay
