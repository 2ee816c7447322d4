"""Row slabs of shape_from_shading across GPUs, one process per GPU (image-stencil row; BASELINE config 4), behind
Thallo_ProblemStep like image_warping's (csrc/solver_dist.cpp, flat form): TWO ghost rows per interior side -- the chain
B_I(c) -> shading row(q) -> J^T gather(i) has radius 2 -- pixel coordinates and the image-border guard are global (global_row0 /
global_rows of ThalloX_Distributed), Gauss-Newton in the single-reduction form (one exchange per PCG iteration: alphaD, N, S1, S2 + the
boundary rows of Ap) or, with lm=True, the Levenberg-Marquardt branch (alphaD; q; betaN + the ghost rows of z).  The exchange is ONE launch on the
device (thallo_hip_dist_xrows: rows into the neighbours' inboxes over xGMI, scalar granules to every rank; chosen after a self-check at Init,
distributed_info() says which) or, device_exchange=False, pack + all-gather + unpack.
This module is set-up only: the row split, the local buffers, the all-gather callback.
"""
import numpy as np
import torch

from . import api
from .distributed import SlabLayout, library_rccl, torch_allgather


class PlanSfsSlabSolver:
    def __init__(self, params_global, W, H, rank, world, l_iters, lm=False, group=None, force_rccl=False, device_exchange=True):
        self.lay = lay = SlabLayout(H, rank, world, ghost=2)
        self.W, self.H, self.rank, self.world = W, H, rank, world
        dev = torch.device("cuda", torch.cuda.current_device())
        local = [lay.local(a) if isinstance(a, np.ndarray) else a for a in params_global]
        self.images = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in local[16:21]]      # X (unknown), D, Im, edgeMaskR, edgeMaskC
        self.solver = api.ThalloSolver((W, lay.Hl), api.energy_file("shape_from_shading"), timing_level=0)
        if lm:
            self.solver.enable_lm()
        self.l_iters = l_iters
        self.solver.set_solver_parameters(nIterations=1 << 30, lIterations=l_iters)
        self.library_rccl = library_rccl(self.solver, rank, world, group, force=force_rccl)
        ag = torch_allgather(group, dev) if world > 1 and not self.library_rccl else None
        self.solver.set_distributed(rank, world, lay.row0, lay.row1, allgather=ag, device_exchange=device_exchange,
                                    global_row0=lay.g0 - lay.top, global_rows=H)
        self.params = self.solver.make_params([float(v) for v in local[:16]] + self.images)

    def solve(self, n_iters, **solver_params):
        """Init + up to n_iters Steps (LM may stop earlier, on every rank alike); returns the cost trajectory"""
        self.solver.set_solver_parameters(nIterations=n_iters, **solver_params)
        self.solver.init(self.params)
        if not self.solver.ready():
            raise RuntimeError("Thallo_ProblemInit failed: " + api.last_error())
        costs = [self.solver.current_cost()]
        while self.solver.step(self.params):
            costs.append(self.solver.current_cost())
        final = self.solver.current_cost()
        if len(costs) == 1 or final != costs[-1]:
            costs.append(final)
        return costs

    def owned(self):
        lay = self.lay
        return self.images[0].view(lay.Hl, self.W)[lay.row0:lay.row1].cpu().numpy()
