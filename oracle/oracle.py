"""ctypes binding of the CPU oracle (oracle/thallo_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (thallo_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libthallo_oracle.so")

LAPLACIAN_IMAGE, LAPLACIAN_GRAPH, IMAGE_WARPING, ARAP_MESH, BUNDLE_ADJUST, SFS = 1, 2, 3, 4, 5, 6


class SolverParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "min_relative_decrease", "min_trust_region_radius", "max_trust_region_radius",
        "q_tolerance", "function_tolerance", "trust_region_radius", "radius_decrease_factor",
        "min_lm_diagonal", "max_lm_diagonal")] + [(n, C.c_int) for n in (
        "residual_reset_period", "nIterations", "lIterations", "use_lm", "float_sums")]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("thallo_oracle.c", "thallo_oracle.h", "cpu_port_image_warping.c")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vpp = C.POINTER(C.c_void_p)
        common = [C.c_int, C.POINTER(C.c_uint), vpp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.orc_default_params.argtypes = [C.POINTER(SolverParams)]
        L.orc_msvc_rand_fill.argtypes = [C.c_void_p, C.c_long, C.c_uint]
        L.orc_solve_kind.argtypes = common + [C.POINTER(SolverParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_solve_kind.restype = C.c_int
        L.orc_cost_kind.argtypes = common + [C.c_int]
        L.orc_cost_kind.restype = C.c_double
        L.orc_n_unknowns_kind.argtypes = common
        L.orc_n_unknowns_kind.restype = C.c_long
        L.orc_eval_jtf_kind.argtypes = common + [C.c_void_p, C.c_void_p]
        L.orc_apply_jtj_kind.argtypes = common + [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_apply_jtj_kind.restype = C.c_double
        L.orc_count_rows_kind.argtypes = common + [C.POINTER(C.c_long)]
        L.orc_count_rows_kind.restype = C.c_long
        L.orc_export_csr_kind.argtypes = common + [C.c_void_p] * 4
        L.orc_export_csr_kind.restype = C.c_long
        L.orc_excluded_mask_kind.argtypes = common + [C.c_void_p]
        _lib = L
    return _lib


def default_params(**kw):
    sp = SolverParams()
    lib().orc_default_params(C.byref(sp))
    for k, v in kw.items():
        setattr(sp, k, v)
    return sp


def msvc_rand(n, seed=1):
    out = np.empty(n, np.float32)
    lib().orc_msvc_rand_fill(out.ctypes.data, n, seed)
    return out


class Problem:
    """Host-side problem instance: `params` is a list indexed like the .t Inputs{};
    numpy arrays (float32 / int32, C-contiguous) for Array/Unknown/Sparse, python floats for Param."""

    def __init__(self, kind, dims, params, fconst=None, iconst=None):
        self.kind = kind
        self.dims = (C.c_uint * 4)(*list(dims) + [0] * (4 - len(dims)))
        self._keep = []
        self.params = list(params)
        ptrs = []
        for p in self.params:
            if isinstance(p, np.ndarray):
                assert p.flags["C_CONTIGUOUS"] and p.dtype in (np.float32, np.int32, np.uint8)
                ptrs.append(p.ctypes.data)
            else:
                f = C.c_float(float(p))
                self._keep.append(f)
                ptrs.append(C.addressof(f))
        self.ptrs = (C.c_void_p * len(ptrs))(*ptrs)
        fc = list(fconst or []) + [0.0] * 8
        ic = list(iconst or []) + [0] * 4
        self.fconst = (C.c_float * 8)(*fc[:8])
        self.iconst = (C.c_int * 4)(*ic[:4])

    def _common(self):
        return (self.kind, self.dims, self.ptrs, self.fconst, self.iconst)

    @property
    def n_unknowns(self):
        return lib().orc_n_unknowns_kind(*self._common())

    def cost(self, float_sums=0):
        return lib().orc_cost_kind(*self._common(), float_sums)

    def eval_jtf(self):
        n = self.n_unknowns
        r = np.zeros(n, np.float32)
        pre = np.zeros(n, np.float32)
        lib().orc_eval_jtf_kind(*self._common(), r.ctypes.data, pre.ctypes.data)
        return r, pre

    def apply_jtj(self, p, float_sums=0):
        p = np.ascontiguousarray(p, np.float32)
        Ap = np.zeros_like(p)
        d = lib().orc_apply_jtj_kind(*self._common(), p.ctypes.data, Ap.ctypes.data, float_sums)
        return Ap, d

    def excluded(self):
        m = np.zeros(self.n_unknowns, np.uint8)
        lib().orc_excluded_mask_kind(*self._common(), m.ctypes.data)
        return m.astype(bool)

    def csr(self):
        nnz = C.c_long()
        nr = lib().orc_count_rows_kind(*self._common(), C.byref(nnz))
        rowptr = np.zeros(nr + 1, np.int32)
        col = np.zeros(nnz.value, np.int32)
        val = np.zeros(nnz.value, np.float32)
        res = np.zeros(nr, np.float32)
        lib().orc_export_csr_kind(*self._common(), rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, res.ctypes.data)
        return rowptr, col, val, res

    def solve(self, sp=None, want_trace=False, **kw):
        """Runs Init + while(Step) in place on the unknown arrays. Returns (costs, trace)."""
        sp = sp or default_params(**kw)
        cap = sp.nIterations + 2
        costs = np.zeros(cap, np.float64)
        tcap = sp.nIterations * sp.lIterations if want_trace else 0
        trace = np.zeros((max(tcap, 1), 2), np.float32)
        n = lib().orc_solve_kind(*self._common(), C.byref(sp), costs.ctypes.data, cap,
                                 trace.ctypes.data if want_trace else None, tcap)
        assert n >= 0
        return costs[: n + 1], (trace[:tcap] if want_trace else None)


def last_pcg_counts():
    """PCG iterations each GN / LM step of the last solve() ran (LM's zeta test ends the loop early)."""
    out = (C.c_int * 1024)()
    n = lib().orc_last_pcg_counts(out, 1024)
    return list(out[: min(n, 1024)])


def last_trust_region():
    """(radius, radius_decrease_factor) as the last LM solve() left them: with the unknowns, the state from which the trajectory continues."""
    r, d = C.c_float(0), C.c_float(0)
    lib().orc_last_trust_region(C.byref(r), C.byref(d))
    return float(r.value), float(d.value)


def set_threads(n):
    """n > 1: the oracle's row loops run on n OpenMP threads (full-size configurations on the GPU box's host cores); 1 = the serial,
    bit-exact known-answer path.  Returns the previous setting."""
    L = lib()
    L.orc_get_threads.restype = C.c_int
    prev = L.orc_get_threads()
    L.orc_set_threads(int(n))
    return prev


def cpu_port_image_warping(W, H, params, nIterations, lIterations, want_costs=True, want_trace=False):
    """OpenMP port (oracle/cpu_port_image_warping.c) -- bench.py's cpu_baseline and the full-size checker. Updates params[0],
    params[1] in place. Returns dict(costs, seconds_pcg, seconds_total, threads[, trace = (alpha_k, beta_k) rows])."""
    L = lib()
    L.orc_cpu_port_image_warping2.restype = C.c_int
    L.orc_cpu_port_image_warping2.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_int, C.c_int,
                                                                                     C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                                                     C.c_void_p, C.c_int]
    costs = np.zeros(nIterations + 1, np.float64)
    ntr = nIterations * lIterations if want_trace else 0
    trace = np.zeros((max(ntr, 1), 2), np.float32)
    tp, tt = C.c_double(), C.c_double()
    th = L.orc_cpu_port_image_warping2(W, H, params[0].ctypes.data, params[1].ctypes.data, params[2].ctypes.data,
                                       params[3].ctypes.data, params[4].ctypes.data, float(params[5]), float(params[6]),
                                       nIterations, lIterations, costs.ctypes.data if want_costs else None, C.byref(tp), C.byref(tt),
                                       trace.ctypes.data if want_trace else None, ntr)
    L.orc_cpu_port_threads_pinned.restype = C.c_int
    out = {"costs": costs if want_costs else None, "seconds_pcg": tp.value, "seconds_total": tt.value, "threads": th,
           "pinned": bool(L.orc_cpu_port_threads_pinned())}
    if want_trace:
        out["trace"] = trace[:ntr]
    return out
