"""Thallo_InitializationParameters::doublePrecision = 1 (API/src/precision.t:3-6: thallo_float = double): the front-end generates the energy's kernels with
thallo_float = double and the reference-shaped double loop of csrc/solver_f64.cpp drives them.  Checked against float64 restatements (linear energies: to 1e-12)
and against the float path of the same files (nonlinear energies: to float accuracy, with the double run converging at least as far)."""
import numpy as np
import pytest

import thallo_amd
from thallo_amd import api, synthetic as syn
from helpers import to_device, to_host, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


WEIGHTED = '''local W,H = Dims("W","H")
Inputs {
    X = Unknown(thallo_float2,{W,H},0),
    A = Array(thallo_float2,{W,H},1),
    Wt = Array(float,{W,H},2),
    Mask = Array(float,{W,H},3),
    w_fit = Param(%s,4)
}
UsePreconditioner(true)
local x,y = W(),H()
X:Exclude(Not(eq(Mask(x,y),0)))
local function edge(dx,dy)
    return Select(InBounds(x+dx,y+dy), Wt(x,y)*(X(x,y) - X(x+dx,y+dy)), 0)
end
r = Residuals {
    fit = w_fit*(X(x,y) - A(x,y)),
    regx = edge(1,0),
    regy = edge(0,1)
}
'''


@pytest.mark.parametrize("param_type", ["float", "thallo_float"])
def test_double_precision_linear_energy_against_float64(torch, tmp_path, param_type):
    """A masked, weighted Laplacian: thallo_float2 unknowns and targets become doubles, the `float` weight and mask planes STAY floats (the reference's double
    mode switches thallo_float only), the weight Param is a host float or a host double by its declared type.  GN 4 x PCG 15 against the same recurrences in
    numpy float64: costs to 1e-12, unknowns to 1e-12, excluded unknowns bit-untouched."""
    import scipy.sparse as sps
    W, H = 40, 28
    f = tmp_path / "weighted_laplacian.t"
    f.write_text(WEIGHTED % param_type)
    rng = np.random.default_rng(3)
    A = rng.uniform(0, 1, (H, W, 2))
    Wt = rng.uniform(0.5, 1.5, (H, W)).astype(np.float32)
    Mask = (rng.uniform(0, 1, (H, W)) < 0.1).astype(np.float32)
    X0 = A + 0.3 * rng.standard_normal(A.shape)
    w_fit = np.float32(0.7) if param_type == "float" else np.float64(0.7)
    dev = [torch.from_numpy(X0.copy()).cuda(), torch.from_numpy(A).cuda(), torch.from_numpy(Wt).cuda(), torch.from_numpy(Mask).cuda(), w_fit]
    s = api.ThalloSolver((W, H), str(f), double_precision=True)
    assert s.energy_name == "generated:weighted_laplacian.t"
    final, costs = s.solve(dev, profiled=True, nIterations=4, lIterations=15)
    s.close()
    wf = float(w_fit)
    n = 2 * W * H; idx = lambda x, y, c: 2 * (y * W + x) + c
    rows, cols, vals, rhs = [], [], [], []
    def add_row(entries):
        k = len(rhs); rhs.append(0.0)
        for j, v in entries: rows.append(k); cols.append(j); vals.append(v)
    for y in range(H):
        for x in range(W):
            for c in range(2):
                add_row([(idx(x, y, c), wf)]); rhs[-1] = -wf * float(A[y, x, c])
    for dx, dy in ((1, 0), (0, 1)):
        for y in range(H):
            for x in range(W):
                for c in range(2):
                    if x + dx < W and y + dy < H: add_row([(idx(x, y, c), float(Wt[y, x])), (idx(x + dx, y + dy, c), -float(Wt[y, x]))])
                    else: add_row([])
    J = sps.csr_matrix((vals, (rows, cols)), shape=(len(rhs), n)); b0 = np.array(rhs)
    free = np.repeat(Mask.reshape(-1) == 0, 2)
    xk = X0.reshape(-1).copy()
    ref = [0.5 * np.sum((J @ xk + b0) ** 2)]
    Jf = J[:, free]
    for _ in range(4):
        F = J @ xk + b0
        g = Jf.T @ F; d = np.asarray(Jf.multiply(Jf).sum(axis=0)).ravel()
        M = 1.0 / (1.0 + np.sqrt(d)) ** 2
        r = -g; z = M * r; pvec = z.copy(); delta = np.zeros_like(r); aN = r @ z
        for _k in range(15):
            Ap = Jf.T @ (Jf @ pvec); aD = pvec @ Ap
            alpha = aN / aD if aD != 0 else 0.0
            delta += alpha * pvec; r -= alpha * Ap; z = M * r; bN = z @ r
            beta = bN / aN if aN != 0 else 0.0
            pvec = z + beta * pvec; aN = bN
        xk[free] += delta
        ref.append(0.5 * np.sum((J @ xk + b0) ** 2))
    assert rel_err(np.array(costs), np.array(ref)) < 1e-12, (costs, ref)
    got = to_host(dev[0]).reshape(-1)
    assert got.dtype == np.float64
    assert np.abs(got[~free] - X0.reshape(-1)[~free]).max() == 0.0
    assert np.abs(got - xk).max() < 1e-12


@pytest.mark.parametrize("q_tolerance", [0.0, 0.05])
def test_double_precision_levenberg_marquardt_against_float64(torch, tmp_path, q_tolerance):
    """doublePrecision = 1 with ThalloX_EnableLM (round 4: the mode ran Gauss-Newton only): the LM branch of gauss_newton.t on double vectors, reference-shaped -- SSq / CtC /
    the damped preconditioner from the raw diagonal, (J^T J + CtC) p, the unguarded divides, q = 0.5 delta . (r + b) with the zeta test after every PCG iteration, the residual
    reset every residual_reset_period iterations, the model cost, accept / revert and the trust region -- against the same recurrences in numpy float64 on the masked weighted
    Laplacian: costs and unknowns to 1e-11 over 6 LM steps of up to 25 PCG iterations (two resets per step; with q_tolerance = 0.05 the zeta test ends the loops early)."""
    import scipy.sparse as sps
    W, H = 40, 28
    f = tmp_path / "weighted_laplacian.t"
    f.write_text(WEIGHTED % "float")
    rng = np.random.default_rng(11)
    A = rng.uniform(0, 1, (H, W, 2))
    Wt = rng.uniform(0.5, 1.5, (H, W)).astype(np.float32)
    Mask = (rng.uniform(0, 1, (H, W)) < 0.1).astype(np.float32)
    X0 = A + 0.3 * rng.standard_normal(A.shape)
    w_fit = np.float32(0.7)
    dev = [torch.from_numpy(X0.copy()).cuda(), torch.from_numpy(A).cuda(), torch.from_numpy(Wt).cuda(), torch.from_numpy(Mask).cuda(), w_fit]
    s = api.ThalloSolver((W, H), str(f), double_precision=True)
    s.enable_lm()
    nit, L, period = 6, 25, 10
    sp = dict(nIterations=nit, lIterations=L, q_tolerance=q_tolerance, trust_region_radius=30.0)      # (a small radius: the damping is felt, steps are partial)
    final, costs = s.solve(dev, profiled=True, **sp)
    s.close()
    wf = float(w_fit)
    n = 2 * W * H; idx = lambda x, y, c: 2 * (y * W + x) + c
    rows, cols, vals, rhs = [], [], [], []
    def add_row(entries):
        k = len(rhs); rhs.append(0.0)
        for j, v in entries: rows.append(k); cols.append(j); vals.append(v)
    for y in range(H):
        for x in range(W):
            for c in range(2):
                add_row([(idx(x, y, c), wf)]); rhs[-1] = -wf * float(A[y, x, c])
    for dx, dy in ((1, 0), (0, 1)):
        for y in range(H):
            for x in range(W):
                for c in range(2):
                    if x + dx < W and y + dy < H: add_row([(idx(x, y, c), float(Wt[y, x])), (idx(x + dx, y + dy, c), -float(Wt[y, x]))])
                    else: add_row([])
    J = sps.csr_matrix((vals, (rows, cols)), shape=(len(rhs), n)); b0 = np.array(rhs)
    free = np.repeat(Mask.reshape(-1) == 0, 2)
    Jf = J[:, free]
    cost_of = lambda x: 0.5 * np.sum((J @ x + b0) ** 2)
    f32 = lambda v: float(np.float32(v))           # solver parameters are floats in both precisions (gauss_newton.t:200-216)
    min_lm, max_lm, min_rel, ftol, max_radius, min_radius = f32(1e-6), f32(1e32), f32(1e-3), f32(1e-6), f32(1e16), f32(1e-32)
    radius, dec = f32(30.0), f32(2.0)
    xk = X0.reshape(-1).copy()
    prev = cost_of(xk); ref = [prev]; SSq = None; early = False
    for it in range(nit):
        F = J @ xk + b0
        r = -(Jf.T @ F); d = np.asarray(Jf.multiply(Jf).sum(axis=0)).ravel()
        if it == 0: SSq = 1.0 / (1.0 + np.sqrt(d)) ** 2
        unclamped = d / radius; cm = (1.0 / SSq) / radius
        CtC = np.minimum(np.maximum(unclamped, min_lm * cm), max_lm * cm)
        M = 1.0 / (CtC + radius * unclamped)
        b = r.copy(); z = M * r; aN = r @ z; delta = np.zeros_like(r); Q0 = 0.0; pvec = None
        for k in range(L):
            pvec = z.copy() if k == 0 else z + (bN / aN_prev) * pvec
            if k: aN = bN
            Ap = Jf.T @ (Jf @ pvec) + CtC * pvec; aD = pvec @ Ap
            alpha = aN / aD
            delta = delta + alpha * pvec
            if (k + 1) % period == 0: r = b - (Jf.T @ (Jf @ delta) + CtC * delta)
            else: r = r - alpha * Ap
            z = M * r; bN = z @ r; aN_prev = aN
            Q1 = 0.5 * delta @ (r + b)
            zeta = (k + 1) * (Q1 - Q0) / Q1
            if not np.isfinite(Q1) or not np.isfinite(zeta) or zeta < f32(q_tolerance): early = early or k + 1 < L; break
            Q0 = Q1
        model = delta @ b - 0.5 * delta @ (Jf.T @ (Jf @ delta))
        xn = xk.copy(); xn[free] += delta
        new = cost_of(xn); change = prev - new; rho = change / model
        if change >= 0 and rho > min_rel:
            if change <= prev * ftol: xk = xn; break
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), max_radius); dec = 2.0; prev = new; xk = xn
        else:
            radius = radius / dec; dec = 2.0 * dec
            if radius < min_radius: break
        ref.append(cost_of(xk))
    costs = np.array(costs); ref = np.array(ref)
    assert len(costs) == len(ref) >= 4, (costs, ref)
    assert rel_err(costs, ref) < 1e-11, (costs, ref)
    assert (q_tolerance > 0) == early
    got = to_host(dev[0]).reshape(-1)
    assert np.abs(got[~free] - X0.reshape(-1)[~free]).max() == 0.0
    assert np.abs(got - xk).max() < 1e-10
    assert costs[-1] < 0.2 * costs[0]


def _solve_both(fname, dims, params32, monkeypatch, keep32=(), lm=False, **sp):
    """The same bundled .t through the front-end in float and in double (unknowns / thallo_float arrays as doubles; float Params and the arrays the file declares
    with a fixed float type -- keep32: their parameter indices -- as they are)."""
    out = {}
    for dbl in (False, True):
        monkeypatch.setenv("THALLO_FRONTEND", "generate")
        import torch
        dev = []
        for k, p in enumerate(params32):
            if isinstance(p, np.ndarray) and p.dtype == np.float32: dev.append(torch.from_numpy(p.astype(np.float64) if dbl and k not in keep32 else p.copy()).cuda())
            elif isinstance(p, np.ndarray): dev.append(torch.from_numpy(p.copy()).cuda())
            else: dev.append(np.float32(p))
        s = api.ThalloSolver(dims, thallo_amd.energy_file(fname), double_precision=dbl)
        assert s.energy_name.startswith("generated:")
        if lm: s.enable_lm()
        final, costs = s.solve(dev, profiled=True, **sp)
        s.close()
        out[dbl] = (dev, np.array(costs))
    return out


def test_double_precision_image_warping_follows_the_float_path(torch, monkeypatch):
    """image_warping (nonlinear: cos / sin of the Angle unknown, Exclude, two unknown images) generated in double: the trajectory agrees with the float kernels of
    the same file to float accuracy, and the double run never ends above it by more than that."""
    W, H = 96, 64
    p = syn.image_warping(W, H)
    res = _solve_both("image_warping.t", (W, H), p, monkeypatch, nIterations=5, lIterations=30)
    cf, cd = res[False][1], res[True][1]
    assert len(cf) == len(cd) == 6
    assert rel_err(cd[:1], cf[:1]) < 1e-6
    assert rel_err(cd, cf) < 2e-4, (cd, cf)
    assert cd[-1] <= cf[-1] * (1 + 1e-4)
    xf, xd = to_host(res[False][0][0]), to_host(res[True][0][0])
    assert xd.dtype == np.float64 and np.abs(xd - xf).max() < 5e-3 * max(1.0, np.abs(xf).max())


def test_double_precision_graph_energy_follows_the_float_path(torch, monkeypatch):
    """arap_mesh_deformation (graph domain through Sparse maps, float3 unknowns, rotations) in double against the float kernels of the same file."""
    p = syn.arap_mesh(24, 16)
    N = p[2].shape[0]; E = p[6].shape[0]
    res = _solve_both("arap_mesh_deformation.t", (N, E), p, monkeypatch, nIterations=4, lIterations=25)
    cf, cd = res[False][1], res[True][1]
    assert rel_err(cd[:1], cf[:1]) < 1e-6
    assert rel_err(cd, cf) < 2e-4, (cd, cf)
    assert cd[-1] <= cf[-1] * (1 + 1e-4)


@pytest.mark.parametrize("which", ["iw", "ba"])
def test_double_precision_levenberg_marquardt_follows_the_float_path(torch, monkeypatch, which):
    """The LM branch in double on nonlinear energies (image_warping; bundle_adjustment through Sparse maps) against the float LM loop of solver.cpp on the kernels generated from
    the same file: same accept / reject decisions, trajectories to float accuracy."""
    if which == "iw":
        res = _solve_both("image_warping.t", (64, 48), syn.image_warping(64, 48, n_markers=8), monkeypatch, lm=True, nIterations=5, lIterations=25)
    else:
        res = _solve_both("bundle_adjustment.t", (12, 60, 300), syn.bundle_adjustment(C=12, P=60, O=300, band=8), monkeypatch, keep32=(2,), lm=True, nIterations=5, lIterations=20)
    cf, cd = res[False][1], res[True][1]
    assert len(cf) == len(cd) >= 4, (cf, cd)
    assert rel_err(cd[:1], cf[:1]) < 2e-6
    assert rel_err(cd, cf) < 5e-4, (cd, cf)
    assert cd[-1] < cd[0] and all(cd[i + 1] <= cd[i] * (1 + 1e-9) for i in range(len(cd) - 1))


@pytest.mark.parametrize("which", ["ba", "sfs"])
def test_double_precision_on_the_other_bundled_energies(torch, monkeypatch, which):
    """bundle_adjustment (Sparse maps into two unknown arrays, AngleAxisRotatePoint, a division by depth) and shape_from_shading (computed arrays over neighbour
    pixels, uint8-free mask planes, sixteen Params) generated in double against the float kernels of the same files."""
    if which == "ba":
        p = syn.bundle_adjustment(C=12, P=60, O=300, band=8)
        res = _solve_both("bundle_adjustment.t", (12, 60, 300), p, monkeypatch, keep32=(2,), nIterations=3, lIterations=10)      # observations = Array(float2, ...): floats in both modes
    else:
        p = syn.shape_from_shading(64, 48)
        res = _solve_both("shape_from_shading.t", (64, 48), p, monkeypatch, nIterations=3, lIterations=10)
    cf, cd = res[False][1], res[True][1]
    assert len(cf) == len(cd) == 4
    assert rel_err(cd[:1], cf[:1]) < 2e-6
    assert rel_err(cd, cf) < 3e-4, (cd, cf)
    assert cd[-1] < cd[0]


def test_double_precision_refuses_what_it_cannot_run(torch, tmp_path):
    """An unknown declared with a fixed float type has no place in double solver vectors: the Plan fails with a message, it does not run in float silently."""
    f = tmp_path / "fixed.t"
    f.write_text('local N = Dims("N")\nInputs { X = Unknown(float,{N},0), A = Array(float,{N},1) }\nlocal i = N()\nr = Residuals { fit = X(i) - A(i) }\n')
    with pytest.raises(RuntimeError, match="doublePrecision"):
        api.ThalloSolver((16,), str(f), double_precision=True)


@pytest.mark.parametrize("form", ["graph", "dense"])
@pytest.mark.parametrize("dbl", [True, False])
def test_two_parameter_curve_fit_like_the_reference_dense_test(torch, form, dbl):
    """The scenario of the reference's tests/dense (main.cpp:9-66: doublePrecision = 1, 512 samples of y = a cos(b x) + b sin(a x), generator (100, 102), start
    (99.7, 101.6)): every residual adds into the same two unknowns (wave-aggregated atomics on doubles), through Sparse maps or over the product domain {N, U}.
    Against the same Gauss-Newton / PCG recurrences in numpy float64: the double run to 1e-9 of the initial cost at every step, the float run to float accuracy."""
    import os
    dim = 512
    a0, b0 = 100.0, 102.0
    x = np.arange(dim) * 2.0 * 3.141592653589 / dim
    y = a0 * np.cos(b0 * x) + b0 * np.sin(a0 * x)
    dt = np.float64 if dbl else np.float32
    samples = np.stack([x, y], 1).astype(dt)
    start = np.array([[99.7, 101.6]], dtype=dt)
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "energies")
    if form == "graph":
        dev = [torch.from_numpy(start.copy()).cuda(), torch.from_numpy(samples).cuda(), torch.arange(dim, dtype=torch.int32).cuda(), torch.zeros(dim, dtype=torch.int32).cuda()]
        s = api.ThalloSolver((dim, 1, dim), os.path.join(here, "curve_fit_graph.t"), double_precision=dbl)
    else:
        dev = [torch.from_numpy(start.copy()).cuda(), torch.from_numpy(samples).cuda()]
        s = api.ThalloSolver((dim, 1), os.path.join(here, "curve_fit_dense.t"), double_precision=dbl)
    final, costs = s.solve(dev, profiled=True, nIterations=6, lIterations=4)
    s.close()
    xs, ys = samples[:, 0].astype(np.float64), samples[:, 1].astype(np.float64)
    p = start[0].astype(np.float64).copy()
    F = lambda q: ys - (q[0] * np.cos(q[1] * xs) + q[1] * np.sin(q[0] * xs))
    ref = [0.5 * np.sum(F(p) ** 2)]
    for _ in range(6):
        a, b = p
        J = np.stack([-(np.cos(b * xs) + b * xs * np.cos(a * xs)), -(-a * xs * np.sin(b * xs) + np.sin(a * xs))], 1)
        g = J.T @ F(p); d = (J * J).sum(0)
        M = 1.0 / (1.0 + np.sqrt(d)) ** 2
        r = -g; z = M * r; pv = z.copy(); delta = np.zeros(2); aN = r @ z
        for _k in range(4):
            Ap = J.T @ (J @ pv); aD = pv @ Ap
            alpha = aN / aD if aD != 0 else 0.0
            delta += alpha * pv; r -= alpha * Ap; z = M * r; bN = z @ r
            beta = bN / aN if aN != 0 else 0.0
            pv = z + beta * pv; aN = bN
        p += delta
        ref.append(0.5 * np.sum(F(p) ** 2))
    costs, ref = np.array(costs), np.array(ref)
    tol = 1e-9 if dbl else 2e-3
    assert np.abs(costs - ref).max() <= tol * ref[0], (costs, ref)
    if dbl:
        assert np.abs(to_host(dev[0])[0] - p).max() < 1e-9 * 100.0
        assert costs[-1] < 1e-6 * costs[0]          # the fit is found (the generator's parameters reproduce the samples exactly)


@pytest.mark.parametrize("kind", ["handwritten", "generated", "generated_double", "lm"])
def test_plan_free_cycles_do_not_leak_device_memory(torch, monkeypatch, kind):
    """The reference's tests/create_delete_cycle (main.cpp:22-26: Plan / Solve / Free ten times and watch the memory): device memory after the tenth cycle is what it
    was after the second -- solver vectors, partial slots, exchange buffers, hipRTC modules and the plans' events all go with Thallo_PlanFree."""
    W, H = 512, 384             # ~25 MB of solver vectors per plan: eight leaked plans would be 200 MB
    p = syn.image_warping(W, H)
    dbl = kind == "generated_double"
    if kind in ("generated", "generated_double"): monkeypatch.setenv("THALLO_FRONTEND", "generate")
    else: monkeypatch.delenv("THALLO_FRONTEND", raising=False)
    dev = []
    for q in p:
        if isinstance(q, np.ndarray): dev.append(torch.from_numpy(q.astype(np.float64) if dbl else q.copy()).cuda())
        else: dev.append(np.float32(q))
    ncyc = 6 if kind in ("generated", "generated_double") else 10       # (a generated plan compiles its kernels with hipRTC at every Plan: 1.5 s per cycle; a leak per plan shows after every cycle)

    def cycles():
        free = []
        for cycle in range(ncyc):
            s = api.ThalloSolver((W, H), thallo_amd.energy_file("image_warping.t"), double_precision=dbl)
            assert s.energy_name.startswith("generated:") == (kind in ("generated", "generated_double"))
            if kind == "lm": s._L.ThalloX_EnableLM(s.plan, 1)
            s.solve(dev, nIterations=2, lIterations=5)
            s.close()
            del s
            torch.cuda.synchronize()
            free.append(torch.cuda.mem_get_info()[0])
        # A leak per plan shows as a drop after EVERY cycle (25 MB here).  The runtime's own pools (code objects, signals, kernel arguments) also grow while the whole suite
        # runs in one process -- in steps of 2 to 32 MB, at most once or twice over these ten cycles, and never per cycle: so the cycles whose free memory dropped are counted.
        drops = sum(1 for a, b in zip(free[1:], free[2:]) if b < a - (1 << 20))
        return drops <= 2 and free[1] - free[-1] <= (96 << 20), free
    # mem_get_info is the DEVICE's free memory: anything else that allocates on the card while the cycles run (seen once in round 4: five equal 52-MB drops in a run whose
    # neighbours -- the same tests in the same order on another box -- were flat) reads as a leak.  A leak of this library repeats; so a failed measurement is taken again.
    ok, free = cycles()
    if not ok: ok, free2 = cycles(); free = free + free2
    assert ok, free


def test_an_array_declared_double_needs_double_precision(torch, tmp_path):
    """`Array(double, ...)` is a double array whatever the state's precision: the single-precision kernels would read it as floats, so the Plan says so; under
    doublePrecision = 1 it is simply one more double array."""
    f = tmp_path / "dbl_array.t"
    f.write_text('local N = Dims("N")\nInputs { X = Unknown(thallo_float,{N},0), A = Array(double,{N},1) }\nlocal i = N()\nr = Residuals { fit = X(i) - A(i) }\n')
    with pytest.raises(RuntimeError, match="declared double"):
        api.ThalloSolver((16,), str(f))
    A = np.linspace(0, 1, 16)
    dev = [torch.zeros(16, dtype=torch.float64, device="cuda"), torch.from_numpy(A).cuda()]
    s = api.ThalloSolver((16,), str(f), double_precision=True)
    s.solve(dev, nIterations=2, lIterations=2)
    s.close()
    assert np.abs(to_host(dev[0]) - A).max() < 1e-14
