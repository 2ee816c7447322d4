"""Vertex-partitioned multi-GPU driver for the graph-edge domain (ARAP mesh deformation).

Rank k owns the contiguous vertex range [n0,n1).  A graph that fits one GPU's cache hierarchy many times over
(102,400 vertices = 2.4 MB per solver vector) does not need ghost-vertex bookkeeping: every rank keeps FULL-length
vectors, runs the gather kernels for its own vertex range only (include/thallo_hip.h: the n0,n1 arguments of
thallo_hip_arap_*), and the one thing a neighbour needs -- the CG direction p at the other end of an edge -- is
delivered by an all-gather of the owned slices of p (2 x 3N floats in total) per PCG iteration, next to the two scalar
all-reduces every PCG iteration has.  The unknowns are re-replicated the same way once per GN step, and every rank
recomputes the per-edge F/G blocks it touches.  Equal-size partitions (N divisible by 4*world) keep the all-gather
regular and the flat ranges 16-byte aligned.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import api


class VertexPartition:
    def __init__(self, N, rank, world):
        if N % (4 * world):
            raise ValueError(f"vertex-partitioned path needs N % (4*world) == 0 (N={N}, world={world})")
        self.N, self.rank, self.world = N, rank, world
        self.chunk = N // world
        self.n0, self.n1 = rank * self.chunk, (rank + 1) * self.chunk


class HipArapPartBackend:
    def __init__(self, part, params, max_l_iters):
        self.L = api.lib()
        L = self.L
        vp, ci, cl, fl = C.c_void_p, C.c_int, C.c_long, C.c_float
        L.thallo_hip_arap_cost.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, fl, fl, vp, C.c_long, vp]
        L.thallo_hip_arap_precompute.argtypes = [ci, vp, vp, vp, vp, vp, fl, vp, vp, C.c_long, vp]
        L.thallo_hip_arap_pcg_init.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, fl, fl, vp, vp, vp, vp, vp, vp, vp, C.c_long, vp]
        L.thallo_hip_arap_apply_jtj.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, fl, fl, vp, vp, vp, C.c_long, vp]
        L.thallo_hip_pcg_pupdate_ranges.argtypes = [vp, vp, vp, vp, cl, cl, cl, cl, ci, api.SumT, api.SumT, api.SumT, vp]
        self.part = part
        dev = torch.device("cuda", torch.cuda.current_device())
        w_fit, w_reg, pos, ang, orig, cons, v0, v1 = params
        self.w_fit, self.w_reg = float(w_fit), float(w_reg)
        N, E = pos.shape[0], v0.shape[0]
        self.N, self.E = N, E
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.position, self.angle, self.original, self.constraints = t(pos), t(ang), t(orig), t(cons)
        # incidence lists (same construction as GraphIncidence::build)
        order = np.argsort(v0, kind="stable")
        out_ptr = np.concatenate([[0], np.cumsum(np.bincount(v0, minlength=N))]).astype(np.int32)
        pos_of = np.empty(E, np.int64); pos_of[order] = np.arange(E)
        out_v1 = v1[order].astype(np.int32)
        iorder = np.argsort(v1, kind="stable")
        in_ptr = np.concatenate([[0], np.cumsum(np.bincount(v1, minlength=N))]).astype(np.int32)
        in_edge, in_src = pos_of[iorder].astype(np.int32), v0[iorder].astype(np.int32)
        pad = lambda a: np.concatenate([a, np.zeros(4, a.dtype)])
        self.out_ptr, self.out_v1, self.in_ptr, self.in_edge, self.in_src = (t(pad(a)) for a in (out_ptr, out_v1, in_ptr, in_edge, in_src))
        self.F = torch.zeros(3 * E + 64, dtype=torch.float32, device=dev)
        self.G = torch.zeros(9 * E + 64, dtype=torch.float32, device=dev)
        self.n = 6 * N
        na = (self.n + 255) // 256 * 256
        z = lambda: torch.zeros(na, dtype=torch.float32, device=dev)
        self.r, self.pre, self.z, self.delta, self.Ap = z(), z(), z(), z(), z()
        self.p = [z(), z()]
        self.parts = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.nb = 1
        self.S = torch.zeros(2 * max_l_iters + 8, dtype=torch.float32, device=dev)
        # owned flat ranges: Position part [3 n0, 3 n1), Angle part [3N + 3 n0, 3N + 3 n1)
        self.rng = (3 * part.n0, 3 * part.chunk, 3 * N + 3 * part.n0, 3 * part.chunk)

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _sum(self, idx):
        return api.SumT(self.S.data_ptr() + 4 * idx, 1)

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed with hipError {-rc}")
        return rc

    def _finish(self, out_idx):
        self._chk(self.L.thallo_hip_finish_sum(api.SumT(self.parts.data_ptr(), self.nb), C.c_void_p(self.S.data_ptr() + 4 * out_idx), self._st()), "finish_sum")

    def _g(self):
        vp = C.c_void_p
        return vp(self.out_ptr.data_ptr()), vp(self.out_v1.data_ptr()), vp(self.in_ptr.data_ptr()), vp(self.in_edge.data_ptr()), vp(self.in_src.data_ptr())

    def cost_local(self, out_idx):
        vp, fl = C.c_void_p, C.c_float
        op, ov, ip, ie, isr = self._g()
        self.nb = self._chk(self.L.thallo_hip_arap_cost(self.N, self.part.n0, self.part.n1, op, ov, vp(self.position.data_ptr()), vp(self.angle.data_ptr()),
                                                        vp(self.original.data_ptr()), vp(self.constraints.data_ptr()), fl(self.w_fit), fl(self.w_reg),
                                                        vp(self.parts.data_ptr()), C.c_long(0), self._st()), "arap_cost")
        self._finish(out_idx)

    def init(self, cur, out_idx):
        vp, fl = C.c_void_p, C.c_float
        op, ov, ip, ie, isr = self._g()
        self._chk(self.L.thallo_hip_arap_precompute(self.N, op, ov, vp(self.position.data_ptr()), vp(self.angle.data_ptr()), vp(self.original.data_ptr()),
                                                    fl(self.w_reg), vp(self.F.data_ptr()), vp(self.G.data_ptr()), C.c_long(0), self._st()), "arap_precompute")
        self.p[cur].zero_()
        self.nb = self._chk(self.L.thallo_hip_arap_pcg_init(self.N, self.part.n0, self.part.n1, op, ip, ie, vp(self.position.data_ptr()), vp(self.constraints.data_ptr()),
                                                            vp(self.F.data_ptr()), vp(self.G.data_ptr()), fl(self.w_fit), fl(self.w_reg),
                                                            vp(self.r.data_ptr()), vp(self.pre.data_ptr()), vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()),
                                                            vp(self.delta.data_ptr()), None, vp(self.parts.data_ptr()), C.c_long(0), self._st()), "arap_pcg_init")
        self._finish(out_idx)

    def pupdate(self, cur, first, iN, iD, iB):
        vp, cl = C.c_void_p, C.c_long
        o0, l0, o1, l1 = self.rng
        self._chk(self.L.thallo_hip_pcg_pupdate_ranges(vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()), vp(self.p[cur ^ 1].data_ptr()), vp(self.delta.data_ptr()),
                                                       cl(o0), cl(l0), cl(o1), cl(l1), 1 if first else 0, self._sum(iN), self._sum(iD), self._sum(iB), self._st()),
                  "pcg_pupdate_ranges")

    def owned_slices(self, vec):
        o0, l0, o1, l1 = self.rng
        return vec[o0:o0 + l0], vec[o1:o1 + l1]

    def full_slices(self, vec):
        return vec[:3 * self.N], vec[3 * self.N:6 * self.N]

    def apply(self, cur, out_idx):
        vp, fl = C.c_void_p, C.c_float
        op, ov, ip, ie, isr = self._g()
        self.nb = self._chk(self.L.thallo_hip_arap_apply_jtj(self.N, self.part.n0, self.part.n1, op, ov, ip, ie, isr, vp(self.constraints.data_ptr()),
                                                             vp(self.G.data_ptr()), fl(self.w_fit), fl(self.w_reg), vp(self.p[cur].data_ptr()),
                                                             vp(self.Ap.data_ptr()), vp(self.parts.data_ptr()), C.c_long(0), self._st()), "arap_apply_jtj")
        self._finish(out_idx)

    def step2(self, iN, iD, out_idx):
        vp, cl = C.c_void_p, C.c_long
        o0, l0, o1, l1 = self.rng
        self.nb = self._chk(self.L.thallo_hip_pcg_step2_ranges(vp(self.r.data_ptr()), vp(self.Ap.data_ptr()), vp(self.pre.data_ptr()), vp(self.z.data_ptr()),
                                                               cl(o0), cl(l0), cl(o1), cl(l1), self._sum(iN), self._sum(iD), vp(self.parts.data_ptr()), self._st()),
                            "pcg_step2_ranges")
        self._finish(out_idx)

    def linear_update(self, cur, iN, iD, with_p):
        vp = C.c_void_p
        o0, l0, o1, l1 = self.rng
        for X, off, xo in ((self.position, o0, 3 * self.part.n0), (self.angle, o1, 3 * self.part.n0)):
            p_ptr = vp(self.p[cur].data_ptr() + 4 * off) if with_p else None
            self._chk(self.L.thallo_hip_linear_update(vp(X.data_ptr() + 4 * xo), vp(self.delta.data_ptr() + 4 * off), p_ptr, C.c_long(l0),
                                                      self._sum(iN), self._sum(iD), self._st()), "linear_update")

    def unknown_views(self):
        return self.position.view(-1), self.angle.view(-1)

    def scalar(self, idx):
        return float(self.S[idx].item())


class GraphPartSolver:
    """GN + PCG over a vertex partition; recurrences of gauss_newton.t:1545-1785, unfused graph schedule."""

    def __init__(self, backend, part, group=None):
        self.be, self.part, self.group = backend, part, group
        self.world = part.world

    def _ar(self, idx):
        if self.world > 1:
            dist.all_reduce(self.be.S[idx:idx + 1], group=self.group)

    def _replicate(self, full, owned):
        """all-gather the owned slice of every rank into the full-length plane (equal chunks, rank order)"""
        if self.world > 1:
            dist.all_gather_into_tensor(full, owned.clone(), group=self.group)

    def cost(self):
        self.be.cost_local(0)
        self._ar(0)
        return self.be.scalar(0)

    def gn_step(self, l_iters):
        be = self.be
        B, L = 2, l_iters
        cur = 0
        be.init(cur, B)
        self._ar(B)
        for k in range(L):
            jN, jD, jB = B + 2 * k, B + 2 * k + 1, B + 2 * k + 2
            be.pupdate(cur, k == 0, jN - 2 if k else jN, jD - 2 if k else jD, jN)
            cur ^= 1
            for full, owned in zip(be.full_slices(be.p[cur]), be.owned_slices(be.p[cur])):
                self._replicate(full, owned)                    # neighbours' p for the gather
            be.apply(cur, jD)
            self._ar(jD)
            be.step2(jN, jD, jB)
            self._ar(jB)
        if L > 0:
            be.linear_update(cur, B + 2 * (L - 1), B + 2 * (L - 1) + 1, True)
        else:
            be.linear_update(cur, B, B, False)
        n0, n1 = self.part.n0, self.part.n1
        for X in be.unknown_views():                            # re-replicate the unknowns for the next precompute
            self._replicate(X, X[3 * n0:3 * n1])

    def solve(self, n_iters, l_iters):
        costs = [self.cost()]
        for _ in range(n_iters):
            self.gn_step(l_iters)
            costs.append(self.cost())
        return costs


def make_hip_arap_solver(params_global, rank, world, max_l_iters):
    part = VertexPartition(params_global[2].shape[0], rank, world)
    be = HipArapPartBackend(part, params_global, max_l_iters)
    return GraphPartSolver(be, part), part
