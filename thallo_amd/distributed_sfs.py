"""Row-slab multi-GPU backend for shape_from_shading (image-stencil row; BASELINE config 4).

Same driver as image_warping (thallo_amd.distributed.SlabSolver) with TWO ghost rows per interior side: the chain
B_I(c) -> shading row(q) -> J^T gather(i) has radius 2.  Per PCG iteration: all-reduce of alphaD, and one all-gather of
[betaN_local | first 2 owned rows of z | last 2 owned rows of z]; p on the ghost rows is maintained locally by running
the p update over the owned rows +-2.  Pixel coordinates and the image-border guard of the shading rows are global
(yoff / Hg arguments of thallo_hip_sfs_*).
"""
import ctypes as C

import numpy as np
import torch

from . import api
from .distributed import SlabLayout, SlabSolver, _segs


class HipSfsSlabBackend:
    def __init__(self, W, layout, local_params, H_global, max_l_iters):
        self.L = api.lib()
        L = self.L
        vp, ci, cl, fl = C.c_void_p, C.c_int, C.c_long, C.c_float
        L.thallo_hip_sfs_precompute.argtypes = [ci] * 6 + [vp] * 10
        L.thallo_hip_sfs_cost.argtypes = [ci] * 6 + [vp] * 8
        L.thallo_hip_sfs_pcg_init.argtypes = [ci] * 6 + [vp] * 15
        L.thallo_hip_sfs_apply_jtj.argtypes = [ci] * 6 + [vp] * 10
        L.thallo_hip_pcg_pupdate_ranges.argtypes = [vp, vp, vp, vp, cl, cl, cl, cl, ci, api.SumT, api.SumT, api.SumT, vp]
        self.W, self.lay, self.Hg = W, layout, H_global
        self.Hl, self.row0, self.row1 = layout.Hl, layout.row0, layout.row1
        self.yoff = layout.g0 - layout.top
        dev = torch.device("cuda", torch.cuda.current_device())
        self.hp = (C.c_float * 16)(*[float(v) for v in local_params[:16]])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.X, self.D, self.Im, self.mR, self.mC = (t(a) for a in local_params[16:21])
        N = W * self.Hl
        self.N = self.n = N
        na = (N + 255) // 256 * 256
        z = lambda: torch.zeros(na, dtype=torch.float32, device=dev)
        self.r, self.z, self.delta, self.Ap = z(), z(), z(), z()
        self.p = [z(), z()]
        self.G = torch.zeros(4 * N + 64, dtype=torch.float32, device=dev)
        self.Wt = torch.zeros(2 * N + 64, dtype=torch.float32, device=dev)
        self.fl = torch.zeros(N + 256, dtype=torch.uint8, device=dev)
        self.U = torch.zeros(2 * N + 64, dtype=torch.float32, device=dev)
        self.R = torch.zeros(3 * N + 64, dtype=torch.float32, device=dev)
        self.parts = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.nb = 1
        self.S = torch.zeros(2 * max_l_iters + 8, dtype=torch.float32, device=dev)
        g = layout.ghost
        self.msg = 1 + 2 * g * W
        self.send = torch.zeros(self.msg, dtype=torch.float32, device=dev)
        self.gath = torch.zeros(layout.world * self.msg, dtype=torch.float32, device=dev)
        rows = lambda y, k: [(W * y, W * k)]
        self.seg_first_last = _segs(rows(self.row0, g) + rows(self.row1 - g, g))
        self.seg_top_ghost = _segs(rows(self.row0 - g, g)) if layout.top else _segs([])
        self.seg_bot_ghost = _segs(rows(self.row1, g)) if layout.bot else _segs([])
        if W % 4:
            raise ValueError("the slab path needs W % 4 == 0")

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _sum(self, idx):
        return api.SumT(self.S.data_ptr() + 4 * idx, 1)

    def _local(self):
        return api.SumT(self.parts.data_ptr(), self.nb)

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed with hipError {-rc}")
        return rc

    def _precompute(self):
        vp = C.c_void_p
        self._chk(self.L.thallo_hip_sfs_precompute(self.W, self.Hl, 0, self.Hl, self.yoff, self.Hg, self.hp, vp(self.X.data_ptr()), vp(self.D.data_ptr()),
                                                   vp(self.Im.data_ptr()), vp(self.mR.data_ptr()), vp(self.mC.data_ptr()), vp(self.G.data_ptr()),
                                                   vp(self.Wt.data_ptr()), vp(self.fl.data_ptr()), self._st()), "sfs_precompute")

    def cost_local(self, out_idx):
        vp = C.c_void_p
        self._precompute()
        self.nb = self._chk(self.L.thallo_hip_sfs_cost(self.W, self.Hl, self.row0, self.row1, self.yoff, self.Hg, self.hp, vp(self.X.data_ptr()), vp(self.D.data_ptr()),
                                                       vp(self.G.data_ptr()), vp(self.Wt.data_ptr()), vp(self.fl.data_ptr()), vp(self.parts.data_ptr()), self._st()), "sfs_cost")
        self._chk(self.L.thallo_hip_finish_sum(self._local(), vp(self.S.data_ptr() + 4 * out_idx), self._st()), "finish_sum")

    def init(self, cur):
        vp = C.c_void_p
        self._precompute()
        self.p[cur].zero_(); self.delta.zero_()
        self.nb = self._chk(self.L.thallo_hip_sfs_pcg_init(self.W, self.Hl, self.row0, self.row1, self.yoff, self.Hg, self.hp, vp(self.X.data_ptr()), vp(self.D.data_ptr()),
                                                           vp(self.G.data_ptr()), vp(self.Wt.data_ptr()), vp(self.fl.data_ptr()), vp(self.U.data_ptr()), vp(self.R.data_ptr()),
                                                           vp(self.r.data_ptr()), vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()), vp(self.delta.data_ptr()),
                                                           None, vp(self.parts.data_ptr()), self._st()), "sfs_pcg_init")

    def step1(self, cur, first, iN, iD, iB, out_idx):
        vp, cl = C.c_void_p, C.c_long
        W = self.W
        lo, hi = self.row0 - self.lay.top, self.row1 + self.lay.bot           # owned rows +- ghost rows: keeps p current there
        self._chk(self.L.thallo_hip_pcg_pupdate_ranges(vp(self.z.data_ptr()), vp(self.p[cur].data_ptr()), vp(self.p[cur ^ 1].data_ptr()), vp(self.delta.data_ptr()),
                                                       cl(W * lo), cl(W * (hi - lo)), cl(0), cl(0), 1 if first else 0, self._sum(iN), self._sum(iD), self._sum(iB),
                                                       self._st()), "pcg_pupdate_ranges")
        self.nb = self._chk(self.L.thallo_hip_sfs_apply_jtj(self.W, self.Hl, self.row0, self.row1, self.yoff, self.Hg, self.hp, vp(self.G.data_ptr()), vp(self.Wt.data_ptr()),
                                                            vp(self.fl.data_ptr()), vp(self.U.data_ptr()), vp(self.R.data_ptr()), vp(self.p[cur ^ 1].data_ptr()),
                                                            vp(self.Ap.data_ptr()), vp(self.parts.data_ptr()), self._st()), "sfs_apply_jtj")
        self._chk(self.L.thallo_hip_finish_sum(self._local(), vp(self.S.data_ptr() + 4 * out_idx), self._st()), "finish_sum")

    def step2(self, iN, iD):
        vp, cl = C.c_void_p, C.c_long
        W = self.W
        self.nb = self._chk(self.L.thallo_hip_pcg_step2_ranges(vp(self.r.data_ptr()), vp(self.Ap.data_ptr()), None, vp(self.z.data_ptr()),
                                                               cl(W * self.row0), cl(W * (self.row1 - self.row0)), cl(0), cl(0),
                                                               self._sum(iN), self._sum(iD), vp(self.parts.data_ptr()), self._st()), "pcg_step2_ranges")

    def pack(self):
        self._chk(self.L.thallo_hip_slab_pack(C.c_void_p(self.z.data_ptr()), self.seg_first_last, self._local(), C.c_void_p(self.send.data_ptr()), self._st()), "slab_pack")

    def unpack(self, out_idx, gathered):
        lay, msg, g, W = self.lay, self.msg, self.lay.ghost, self.W
        base = gathered.data_ptr()
        src_top = C.c_void_p(base + 4 * ((lay.rank - 1) * msg + 1 + g * W)) if lay.top else None      # rank-1's LAST g rows
        src_bot = C.c_void_p(base + 4 * ((lay.rank + 1) * msg + 1)) if lay.bot else None              # rank+1's FIRST g rows
        self._chk(self.L.thallo_hip_slab_unpack(C.c_void_p(self.z.data_ptr()), self.seg_top_ghost, src_top, self.seg_bot_ghost, src_bot,
                                                C.c_void_p(base), C.c_long(msg), lay.world, C.c_void_p(self.S.data_ptr() + 4 * out_idx), self._st()), "slab_unpack")

    def linear_update(self, cur, iN, iD, with_p):
        vp = C.c_void_p
        W = self.W
        off, ln = W * self.row0, W * (self.row1 - self.row0)
        p_ptr = vp(self.p[cur].data_ptr() + 4 * off) if with_p else None
        self._chk(self.L.thallo_hip_linear_update(vp(self.X.data_ptr() + 4 * off), vp(self.delta.data_ptr() + 4 * off), p_ptr, C.c_long(ln),
                                                  self._sum(iN), self._sum(iD), self._st()), "linear_update")

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
        return float(self.S[idx].item())


def make_hip_sfs_solver(params_global, W, H, rank, world, max_l_iters):
    lay = SlabLayout(H, rank, world, ghost=2)
    local = [lay.local(a) if isinstance(a, np.ndarray) else a for a in params_global]
    be = HipSfsSlabBackend(W, lay, local, H, max_l_iters)
    return SlabSolver(be, lay), lay
