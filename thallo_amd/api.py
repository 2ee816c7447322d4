"""ctypes mirror of include/Thallo.h (the drop-in C API) and a ThalloSolver harness class.

The class mirrors the reference's application harness, examples/shared/ThalloSolver.h:40-112
(ctor = NewState -> ProblemDefine -> ProblemPlan; solve = SetSolverParameter* -> Solve, or
Init + while(Step) with a cost read after every step like launchProfiledSolve,
examples/shared/ThalloUtils.h:75-92), so parity tests read like the reference's own programs.

There is no CPU fallback: a missing libThallo.so or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libThallo.so")
ENERGY_DIR = os.path.join(_HERE, "energies")


class InitializationParameters(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("doublePrecision", "verbosityLevel", "timingLevel",
                                       "threadsPerBlock", "useAutoscheduler", "cpuOnly")]


class PerformanceEntry(C.Structure):
    _fields_ = [("count", C.c_uint), ("minMS", C.c_double), ("maxMS", C.c_double),
                ("meanMS", C.c_double), ("stddevMS", C.c_double)]


class PerformanceSummary(C.Structure):
    _fields_ = [(n, PerformanceEntry) for n in ("total", "nonlinearIteration", "nonlinearSetup",
                                                "linearSolve", "nonlinearResolve")]


class SumT(C.Structure):      # thallo_sum_t of include/thallo_hip.h
    _fields_ = [("partials", C.c_void_p), ("count", C.c_int)]


class DistT(C.Structure):
    """thallo_dist_t (include/thallo_hip.h): mailboxes + neighbour r vectors of the device-side multi-GPU exchange"""
    _fields_ = [("mail", C.c_void_p), ("peer_mail", C.c_void_p * 8), ("peer_r", C.c_void_p * 2), ("peer_off_o", C.c_long * 2),
                ("peer_off_a", C.c_long * 2), ("ctl", C.c_void_p), ("world", C.c_int), ("rank", C.c_int)]


class SegsT(C.Structure):     # thallo_segs_t of include/thallo_hip.h
    _fields_ = [("off", C.c_long * 8), ("len", C.c_long * 8), ("n", C.c_int)]


INT_PARAMS = ("nIterations", "lIterations", "residual_reset_period", "nIter")

_lib = None


# include/Thallo.h: ThalloX_AllGatherFn / ThalloX_Distributed
AllGatherFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p)
AllReduceFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p)


class DistributedT(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("row0", C.c_uint), ("row1", C.c_uint),
                ("allgather", AllGatherFn), ("user", C.c_void_p), ("device_exchange", C.c_int),
                ("global_row0", C.c_uint), ("global_rows", C.c_uint), ("allreduce", AllReduceFn)]


def lib():
    """Load libThallo.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("THALLO_LIB") or LIB_PATH          # THALLO_LIB: tools only (an A/B build from `make VARIANT=...`, tools/ab/)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C thallo_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    L.Thallo_NewState.argtypes = [InitializationParameters]; L.Thallo_NewState.restype = vp
    L.Thallo_ProblemDefine.argtypes = [vp, C.c_char_p, C.c_char_p]; L.Thallo_ProblemDefine.restype = vp
    L.Thallo_ProblemDelete.argtypes = [vp, vp]
    L.Thallo_ProblemPlan.argtypes = [vp, vp, C.POINTER(C.c_uint)]; L.Thallo_ProblemPlan.restype = vp
    L.Thallo_PlanFree.argtypes = [vp, vp]
    L.Thallo_SetSolverParameter.argtypes = [vp, vp, C.c_char_p, vp]
    L.Thallo_GetSolverParameter.argtypes = [vp, vp, C.c_char_p, vp]
    L.Thallo_ProblemSolve.argtypes = [vp, vp, vpp]
    L.Thallo_ProblemInit.argtypes = [vp, vp, vpp]
    L.Thallo_ProblemStep.argtypes = [vp, vp, vpp]; L.Thallo_ProblemStep.restype = C.c_int
    L.Thallo_ProblemCurrentCost.argtypes = [vp, vp]; L.Thallo_ProblemCurrentCost.restype = C.c_double
    L.Thallo_GetPerformanceSummary.argtypes = [vp, vp, C.POINTER(PerformanceSummary)]
    L.ThalloX_SetStream.argtypes = [vp, vp]
    L.ThalloX_SetKernelSampling.argtypes = [vp, C.c_int]
    L.ThalloX_GetKernelStat.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.ThalloX_GetKernelStat.restype = C.c_int
    L.ThalloX_GetKernelStatOwn.argtypes = [vp, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.ThalloX_GetKernelStatOwn.restype = C.c_int
    L.ThalloX_ResetKernelStats.argtypes = [vp]
    L.ThalloX_GetAlphaBetaTrace.argtypes = [vp, vp, C.c_int]; L.ThalloX_GetAlphaBetaTrace.restype = C.c_int
    L.ThalloX_EnableLM.argtypes = [vp, C.c_int]
    L.ThalloX_PlanEnergyName.argtypes = [vp]; L.ThalloX_PlanEnergyName.restype = C.c_char_p
    L.ThalloX_LastError.restype = C.c_char_p
    L.ThalloX_PlanReady.argtypes = [vp]; L.ThalloX_PlanReady.restype = C.c_int
    L.ThalloX_PlanScheduleName.argtypes = [vp]; L.ThalloX_PlanScheduleName.restype = C.c_char_p
    L.ThalloX_PlanSetDistributed.argtypes = [vp, C.POINTER(DistributedT)]; L.ThalloX_PlanSetDistributed.restype = C.c_int
    L.ThalloX_PlanDistributedInfo.argtypes = [vp]; L.ThalloX_PlanDistributedInfo.restype = C.c_char_p
    L.ThalloX_DistributedControl.argtypes = [vp, C.c_int, C.c_int]; L.ThalloX_DistributedControl.restype = C.c_int
    L.ThalloX_UnknownsChanged.argtypes = [vp, vp]
    L.ThalloX_RcclUniqueId.argtypes = [vp]; L.ThalloX_RcclUniqueId.restype = C.c_int
    L.ThalloX_PlanUseRccl.argtypes = [vp, vp, C.c_int, C.c_int]; L.ThalloX_PlanUseRccl.restype = C.c_int
    L.ThalloX_RcclSelfTest.argtypes = []; L.ThalloX_RcclSelfTest.restype = C.c_int
    L.ThalloX_RcclAvailable.argtypes = []; L.ThalloX_RcclAvailable.restype = C.c_int
    L.ThalloX_DistributedKernelOnly.argtypes = [vp, C.c_int]; L.ThalloX_DistributedKernelOnly.restype = C.c_int
    L.ThalloX_ProblemFileHash.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; L.ThalloX_ProblemFileHash.restype = C.c_ulonglong
    L.ThalloX_ProblemFileSchedule.argtypes = [C.c_char_p]; L.ThalloX_ProblemFileSchedule.restype = C.c_int
    # --- kernel shim entry points that python drives directly (distributed driver, bench, tests): typed, so a
    #     signature drift raises an ArgumentError instead of corrupting the call
    fl, ci, cl = C.c_float, C.c_int, C.c_long
    L.thallo_hip_vector_elems.argtypes = [cl]; L.thallo_hip_vector_elems.restype = cl
    L.thallo_hip_finish_sum.argtypes = [SumT, vp, vp]
    L.thallo_hip_iw_cost.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, fl, fl, vp, vp]
    L.thallo_hip_iw_pcg_init.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, fl, fl, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.thallo_hip_iw_pcg_step1.argtypes = [ci, ci, ci, ci, vp, vp, vp, fl, fl, vp, vp, vp, vp, vp, ci, SumT, SumT, SumT, SumT, SumT, vp, vp, vp, vp]
    L.thallo_hip_linear_update2.argtypes = [vp, vp, vp, SumT, SumT, vp, SumT, SumT, cl, vp]
    L.thallo_hip_ipc_alloc.argtypes = [C.c_long, C.POINTER(vp), vp]
    L.thallo_hip_ipc_alloc2.argtypes = [C.c_long, C.POINTER(vp), vp, C.POINTER(ci)]
    L.thallo_hip_ipc_open.argtypes = [vp, C.POINTER(vp)]
    L.thallo_hip_ipc_close.argtypes = [vp]; L.thallo_hip_ipc_free.argtypes = [vp]
    L.thallo_hip_dist_begin_step.argtypes = [DistT, vp]
    L.thallo_hip_dist_collect.argtypes = [DistT, ci, ci, vp, vp]
    L.thallo_hip_dist_error.argtypes = [DistT, ci, vp]
    L.thallo_hip_slab_pack_iter.argtypes = [vp, SegsT, vp, vp, ci, vp, vp]
    L.thallo_hip_slab_unpack_iter.argtypes = [vp, SegsT, vp, SegsT, vp, vp, cl, ci, SumT, vp, vp, vp]
    L.thallo_hip_dist_exchange.argtypes = [DistT, ci, SumT, vp, vp]
    L.thallo_hip_dist_exchange_iter.argtypes = [DistT, ci, vp, vp, ci, SumT, vp, vp, vp]
    _iter = [ci, ci, ci, ci, vp, vp, vp, vp, fl, fl, vp, vp, vp, vp, vp, vp, vp, ci, SumT, SumT, SumT, SumT, SumT, vp]
    L.thallo_hip_iw_pcg_iter.argtypes = _iter + [vp, vp, vp, vp, vp, vp]
    L.thallo_hip_iw_pcg_iter_dist.argtypes = _iter + [DistT, vp, vp, vp, ci, vp, vp, vp]
    _march = [ci, ci, ci, ci, vp, vp, fl, fl, vp, vp, vp, vp, vp, vp, vp, ci, SumT, SumT, SumT, SumT, SumT, vp]     # no urshape / pre
    L.thallo_hip_iw_pcg_iter_march.argtypes = _march + [vp, vp, vp, vp, vp, vp]
    L.thallo_hip_iw_pcg_iter_march_dist.argtypes = _march + [DistT, vp, vp, vp, ci, vp, vp, vp]
    L.thallo_hip_iw_urshape_irregular.argtypes = [ci, ci, vp, vp, vp]
    L.thallo_hip_march_debug_set.argtypes = [ci, ci]
    L.thallo_hip_iw_pcg_iter_finish.argtypes = [vp, vp, ci, SumT, vp, vp, vp]
    L.thallo_hip_iw_pcg_step2_dist.argtypes = [ci, ci, ci, ci, vp, fl, fl, vp, vp, SumT, SumT, DistT, vp, vp]
    L.thallo_hip_iw_pcg_step2.argtypes = [ci, ci, ci, ci, vp, fl, fl, vp, vp, vp, vp, SumT, SumT, vp, vp, vp]
    L.thallo_hip_iw_apply_jtj.argtypes = [ci, ci, ci, ci, vp, vp, vp, fl, fl, vp, vp, vp, vp, vp]
    L.thallo_hip_pcg_step2.argtypes = [vp, vp, vp, vp, cl, SumT, SumT, vp, vp]
    L.thallo_hip_pcg_step2_ranges.argtypes = [vp, vp, vp, vp, cl, cl, cl, cl, SumT, SumT, vp, vp]
    L.thallo_hip_linear_update.argtypes = [vp, vp, vp, cl, SumT, SumT, vp]
    L.thallo_hip_slab_pack.argtypes = [vp, SegsT, SumT, vp, vp]
    L.thallo_hip_slab_unpack.argtypes = [vp, SegsT, vp, SegsT, vp, vp, cl, ci, vp, vp]
    _lib = L
    return L


def energy_file(name):
    """Path of a bundled problem specification (thallo_amd/energies/<name>.t)."""
    p = os.path.join(ENERGY_DIR, name if name.endswith(".t") else name + ".t")
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def last_error():
    return lib().ThalloX_LastError().decode()


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0); raises when the library cannot bind RCCL"""
    buf = (C.c_ubyte * 128)()
    if lib().ThalloX_RcclUniqueId(C.addressof(buf)) != 0:
        raise RuntimeError("ThalloX_RcclUniqueId failed: " + last_error())
    return bytes(buf)


def _ptr_of(x):
    """Device pointer of a torch tensor, host pointer of a ctypes scalar, or a raw int."""
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, (C.c_float, C.c_int, C.c_double, C.c_uint)):
        return C.addressof(x)
    if isinstance(x, int):
        return x
    raise TypeError(f"cannot turn {type(x)} into a problem parameter")


class ThalloSolver:
    """NewState -> ProblemDefine -> ProblemPlan, like examples/shared/ThalloSolver.h:43-74."""

    def __init__(self, dims, thallofile, solverkind="gauss_newton", verbosity=0, timing_level=1,
                 autoschedule=1, double_precision=False, cpu_only=False):
        L = lib()
        ip = InitializationParameters(int(double_precision), verbosity, timing_level, 0, autoschedule, int(cpu_only))
        self._L = L
        self.state = L.Thallo_NewState(ip)
        if not self.state:
            raise RuntimeError("Thallo_NewState failed: " + last_error())
        self.problem = L.Thallo_ProblemDefine(self.state, os.fsencode(thallofile), solverkind.encode())
        if not self.problem:
            raise RuntimeError("Thallo_ProblemDefine failed: " + last_error())
        self._dims = (C.c_uint * 10)(*dims)          # retained by the plan (thallo.t:1419)
        self.plan = L.Thallo_ProblemPlan(self.state, self.problem, self._dims)
        if not self.plan:
            raise RuntimeError("Thallo_ProblemPlan failed: " + last_error())
        self._keep = None

    def close(self):
        if getattr(self, "plan", None):
            self._L.Thallo_PlanFree(self.state, self.plan); self.plan = None
        if getattr(self, "problem", None):
            self._L.Thallo_ProblemDelete(self.state, self.problem); self.problem = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def energy_name(self):
        return self._L.ThalloX_PlanEnergyName(self.plan).decode()

    @property
    def schedule_name(self):
        return self._L.ThalloX_PlanScheduleName(self.plan).decode()

    def set_solver_parameters(self, **kw):
        for k, v in kw.items():
            val = C.c_int(int(v)) if k in INT_PARAMS else C.c_float(float(v))
            self._L.Thallo_SetSolverParameter(self.state, self.plan, k.encode(), C.addressof(val))

    def get_solver_parameter(self, name):
        val = C.c_int() if name in INT_PARAMS else C.c_float()
        self._L.Thallo_GetSolverParameter(self.state, self.plan, name.encode(), C.addressof(val))
        return val.value

    def make_params(self, problem_params):
        """void** indexed like the .t Inputs{}: tensors -> device pointers, python floats -> host float*."""
        keep = []
        ptrs = []
        for p in problem_params:
            if isinstance(p, np.float64):       # Param(thallo_float, ...) under double_precision=True: a host double
                p = C.c_double(float(p))
            elif isinstance(p, (float, np.float32)):
                p = C.c_float(float(p))
            keep.append(p)
            ptrs.append(_ptr_of(p))
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        self._keep = (keep, arr)
        return arr

    def init(self, params):
        self._L.Thallo_ProblemInit(self.state, self.plan, params)

    def ready(self):
        return self._L.ThalloX_PlanReady(self.plan) == 1

    def step(self, params):
        return self._L.Thallo_ProblemStep(self.state, self.plan, params)

    def current_cost(self):
        return self._L.Thallo_ProblemCurrentCost(self.state, self.plan)

    def solve(self, problem_params, profiled=False, **solver_params):
        """Returns (final_cost, costs): costs[0] is the initial cost, costs[k] the cost after GN step k
        when profiled (Init + while(Step) + CurrentCost, ThalloUtils.h:75-92), else just [final]."""
        self.set_solver_parameters(**solver_params)
        params = problem_params if isinstance(problem_params, C.Array) else self.make_params(problem_params)
        costs = []
        if profiled:
            self.init(params)
            costs.append(self.current_cost())
            while self.step(params):
                costs.append(self.current_cost())
        else:
            self._L.Thallo_ProblemSolve(self.state, self.plan, params)
        final = self.current_cost()
        if not profiled:
            costs.append(final)
        return final, costs

    def performance_summary(self):
        s = PerformanceSummary()
        self._L.Thallo_GetPerformanceSummary(self.state, self.plan, C.byref(s))
        return {n: {"count": getattr(s, n).count, "minMS": getattr(s, n).minMS, "maxMS": getattr(s, n).maxMS,
                    "meanMS": getattr(s, n).meanMS, "stddevMS": getattr(s, n).stddevMS} for n, _ in s._fields_}

    # ---- extensions
    def set_stream(self, stream_ptr):
        self._L.ThalloX_SetStream(self.plan, C.c_void_p(stream_ptr))

    def set_distributed(self, rank, world, row0, row1, allgather=None, device_exchange=True, global_row0=0, global_rows=0, allreduce=None):
        """Collective (include/Thallo.h ThalloX_PlanSetDistributed): this plan is rank `rank`'s row slab; `allgather(send_ptr, recv_ptr,
        bytes_per_rank, stream_ptr) -> None` moves DEVICE bytes (thallo_amd.distributed.torch_allgather builds one over torch.distributed)."""
        def _cb(_user, send, recv, nbytes, stream):
            try:
                allgather(send, recv, nbytes, stream or 0)
                return 0
            except Exception:      # noqa: BLE001 - an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        def _ar(_user, buf, count, stream):
            try:
                allreduce(buf, count, stream or 0)
                return 0
            except Exception:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -1
        self._dist_cb = AllGatherFn(_cb) if allgather is not None else C.cast(None, AllGatherFn)      # kept alive with the plan
        self._dist_ar = AllReduceFn(_ar) if allreduce is not None else C.cast(None, AllReduceFn)
        cfg = DistributedT(rank, world, row0, row1, self._dist_cb, None, 1 if device_exchange else 0, global_row0, global_rows, self._dist_ar)
        if self._L.ThalloX_PlanSetDistributed(self.plan, C.byref(cfg)) != 0:
            raise RuntimeError("ThalloX_PlanSetDistributed failed: " + last_error())

    def set_ghost_exchange(self, boundary_units, ghost_units, ghost_src_rank, ghost_src_pos):
        """Before set_distributed (include/Thallo.h ThalloX_PlanSetGhostExchange): this plan is a rank's LOCAL sub-problem of a partitioned graph energy -- owned units
        first, then ghosts; the lists say which owned units travel and where each ghost's values come from (thallo_amd.distributed_graph.GhostPartition builds them)."""
        import numpy as np
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (boundary_units, ghost_units, ghost_src_rank, ghost_src_pos)]
        self._L.ThalloX_PlanSetGhostExchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._L.ThalloX_PlanSetGhostExchange.restype = C.c_int
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        if self._L.ThalloX_PlanSetGhostExchange(self.plan, int(arrs[0].size), ptr(arrs[0]), int(arrs[1].size), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3])) != 0:
            raise RuntimeError("ThalloX_PlanSetGhostExchange failed: " + last_error())

    def use_rccl(self, unique_id, rank, world):
        """Collective, before set_distributed: the plan makes its own RCCL communicator from the 128-byte id (rccl_unique_id() on rank 0, handed to every
        rank by the application); set_distributed without callbacks then means ncclAllGather / ncclAllReduce inside the library."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        if self._L.ThalloX_PlanUseRccl(self.plan, C.addressof(buf), rank, world) != 0:
            raise RuntimeError("ThalloX_PlanUseRccl failed: " + last_error())

    def rccl_info(self):
        """{world, device, rank} as the plan's own RCCL communicator reports them (ncclCommCount / CuDevice / UserRank); -1 = no communicator."""
        out = (C.c_int * 3)(-1, -1, -1)
        self._L.ThalloX_PlanRcclInfo.argtypes = [C.c_void_p, C.c_void_p]; self._L.ThalloX_PlanRcclInfo.restype = None
        self._L.ThalloX_PlanRcclInfo(self.plan, C.addressof(out))
        return {"world": out[0], "device": out[1], "rank": out[2]}

    def distributed_info(self):
        import json
        t = self._L.ThalloX_PlanDistributedInfo(self.plan).decode()
        return json.loads(t) if t else {}

    def distributed_error(self, clear=True):
        return self._L.ThalloX_DistributedControl(self.plan, 0, 1 if clear else 0)

    def distributed_use_allgather(self):
        return self._L.ThalloX_DistributedControl(self.plan, 1, 0)

    def distributed_kernel_only(self, reps):
        return self._L.ThalloX_DistributedKernelOnly(self.plan, reps)

    def unknowns_changed(self):
        """The caller rewrote unknowns / inputs in place between two LM steps (include/Thallo.h ThalloX_UnknownsChanged)."""
        self._L.ThalloX_UnknownsChanged(self.state, self.plan)

    def enable_lm(self, on=True):
        """Run the LM branch of gauss_newton.t (dead as shipped in the reference, see include/Thallo.h)."""
        self._L.ThalloX_EnableLM(self.plan, 1 if on else 0)

    def set_kernel_sampling(self, period):
        self._L.ThalloX_SetKernelSampling(self.plan, period)

    def reset_kernel_stats(self):
        self._L.ThalloX_ResetKernelStats(self.plan)

    def kernel_stats(self):
        out = {}
        i = 0
        while True:
            name = C.c_char_p(); launches = C.c_long(); samples = C.c_long(); total = C.c_double()
            if self._L.ThalloX_GetKernelStat(self.plan, i, C.byref(name), C.byref(launches), C.byref(samples), C.byref(total)):
                break
            ks = C.c_long(); kt = C.c_double()
            self._L.ThalloX_GetKernelStatOwn(self.plan, i, C.byref(ks), C.byref(kt))
            out[name.value.decode()] = {"launches": launches.value, "samples": samples.value, "total_ms": total.value,
                                        "mean_ms": total.value / samples.value if samples.value else None,
                                        "own_samples": ks.value, "own_mean_ms": kt.value / ks.value if ks.value else None}
            i += 1
        return out

    def alpha_beta_trace(self, cap=4096):
        buf = (C.c_float * (2 * cap))()
        n = self._L.ThalloX_GetAlphaBetaTrace(self.plan, C.addressof(buf), cap)
        n = min(n, cap)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


def iw_fused_bytes_per_iter(L, every_iteration_delta=False, ring=False):
    """Bytes per pixel and PCG iteration the image_warping one-kernel schedule has to move, each array once, averaged over the L iterations of
    a GN step.  Round 4 (energy_image_warping_march_rc.hip, no A p plane): read r 12, p 12, cs 8, flags 1; write r 12, p 12 = 57, + the delta update the launch carries:
    none on odd iterations and two (read delta 12 + p_{k-2} 12, write delta 12 = 36) on even ones -- or, on the multi-GPU device-side transport, one (24) every
    iteration -- or (round 5, ring=True: one GPU, the default) none at all, the update being a launch of its own per ring of p planes; the FIRST iteration of a GN
    step runs the stored-plane kernel without its A p read: 69.  THALLO_MARCH=0 / 3 / 4 (tile kernel / stored-plane marching
    kernel, the A/B forms): 99."""
    if os.environ.get("THALLO_MARCH", "1")[:1] in ("0", "3", "4"):
        return 99.0
    if L < 1:
        return 0.0
    if ring:        # round 5: p_k goes into a ring of planes, no launch touches delta (thallo_hip_linear_update_n does, once per ring: not this kernel's bytes)
        return (69.0 + 57.0 * (L - 1)) / L
    if every_iteration_delta:
        return (69.0 + 81.0 * (L - 1)) / L
    odd, even = L // 2, (L - 1) // 2          # k = 1, 3, ...: no delta update; k = 2, 4, ...: two
    return (69.0 + 57.0 * odd + 93.0 * even) / L

