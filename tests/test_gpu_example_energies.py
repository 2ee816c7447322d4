"""Energies with the structure of further reference examples (gradient-domain pasting, cotangent mesh fairing, robust point-to-plane alignment), stated for this
repo in tests/energies/, through the front-end -- against a numpy float64 mirror of the solver that knows nothing about the front-end's derivatives: the Jacobian
comes from complex-step differentiation of the residual function (exact to rounding), then the same Gauss-Newton / PCG recurrences (gauss_newton.t:1545-1785).
doublePrecision = 1 makes the comparison sharp (1e-9 of the initial cost at every step); the float kernels of the same files are held to float accuracy."""
import os

import numpy as np
import pytest

from thallo_amd import api
from helpers import to_host

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "energies")


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


def mirror(F, x0, free, n_iter, l_iter, precond):
    """Gauss-Newton with l_iter PCG iterations per step on 0.5 |F(x)|^2; F maps a (complex-capable) vector of ALL unknowns to the residual vector."""
    x = x0.astype(np.float64).copy()
    cols = np.nonzero(free)[0]
    costs = [0.5 * np.sum(F(x).real ** 2)]
    for _ in range(n_iter):
        Fx = F(x).real
        J = np.empty((Fx.size, cols.size))
        for k, j in enumerate(cols):
            xp = x.astype(np.complex128); xp[j] += 1e-30j
            J[:, k] = F(xp).imag / 1e-30
        g = J.T @ Fx; d = (J * J).sum(0)
        M = 1.0 / (1.0 + np.sqrt(d)) ** 2 if precond else np.ones_like(d)
        r = -g; z = M * r; p = z.copy(); delta = np.zeros_like(r); aN = r @ z
        for _k in range(l_iter):
            Ap = J.T @ (J @ p); aD = p @ Ap
            alpha = aN / aD if aD != 0 else 0.0
            delta += alpha * p; r -= alpha * Ap; z = M * r; bN = z @ r
            beta = bN / aN if aN != 0 else 0.0
            p = z + beta * p; aN = bN
        x[cols] += delta
        costs.append(0.5 * np.sum(F(x).real ** 2))
    return x, np.array(costs)


def run(torch, file, dims, params, unknown_slots, dbl, here=None, **sp):
    dev = []
    for p in params:
        if isinstance(p, np.ndarray) and p.dtype == np.float64: dev.append(torch.from_numpy(p if dbl else p.astype(np.float32)).cuda())
        elif isinstance(p, np.ndarray): dev.append(torch.from_numpy(p.copy()).cuda())
        else: dev.append(np.float32(p))
    s = api.ThalloSolver(dims, os.path.join(here or HERE, file), double_precision=dbl)
    assert s.energy_name == "generated:" + file
    final, costs = s.solve(dev, profiled=True, **sp)
    s.close()
    return np.concatenate([to_host(dev[k]).astype(np.float64).reshape(-1) for k in unknown_slots]), np.array(costs)


def check(got_x, got_costs, ref_x, ref_costs, dbl):
    tol = 1e-9 if dbl else 3e-4
    assert np.abs(got_costs - ref_costs).max() <= tol * ref_costs[0], (got_costs, ref_costs)
    assert np.abs(got_x - ref_x).max() <= (1e-8 if dbl else 2e-3) * max(1.0, np.abs(ref_x).max())


@pytest.mark.parametrize("dbl", [True, False])
def test_gradient_domain_pasting(torch, dbl):
    W, H = 14, 11
    rng = np.random.default_rng(5)
    T = rng.uniform(0, 1, (H, W, 4))
    X0 = rng.uniform(0, 1, (H, W, 4))
    M = np.ones((H, W), np.float32); M[2:9, 3:12] = 0.0; M[5, 6] = 1.0          # a region with a pinned pixel inside
    free = np.repeat(M.reshape(-1) == 0, 4)

    def F(x):
        X = x.reshape(H, W, 4); out = []
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            r = np.zeros_like(X)
            ys, xs = slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx))
            yn, xn = slice(max(0, dy), H - max(0, -dy)), slice(max(0, dx), W - max(0, -dx))
            r[ys, xs] = (X[ys, xs] - X[yn, xn]) - (T[ys, xs] - T[yn, xn])
            out.append(r.reshape(-1))
        return np.concatenate(out)

    ref_x, ref_costs = mirror(F, X0.reshape(-1), free, 3, 20, precond=False)
    got_x, got_costs = run(torch, "gradient_paste.t", (W, H), [X0.copy(), T, M], [0], dbl, nIterations=3, lIterations=20)
    check(got_x, got_costs, ref_x, ref_costs, dbl)
    assert np.array_equal(got_x[~free], (X0 if dbl else X0.astype(np.float32).astype(np.float64)).reshape(-1)[~free])      # pixels outside the region untouched


@pytest.mark.parametrize("dbl", [True, False])
def test_cotangent_weighted_fairing(torch, dbl):
    nx, ny = 9, 7
    rng = np.random.default_rng(8)
    gx, gy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64))
    A = np.stack([gx, gy, 0.3 * np.sin(0.7 * gx) * np.cos(0.5 * gy)], -1).reshape(-1, 3)
    X0 = A + 0.05 * rng.standard_normal(A.shape)
    vid = lambda i, j: j * nx + i
    e = [(vid(i, j), vid(i + 1, j), vid(i, j + 1), vid(i + 1, j - 1)) for j in range(1, ny - 1) for i in range(nx - 1)]
    v = [np.array([q[k] for q in e], np.int32) for k in range(4)]
    N, E = A.shape[0], len(e)
    w_fit, w_reg = float(np.float32(0.8)), float(np.float32(0.6))          # Param(float): both precisions receive the floats nearest to 0.8 / 0.6

    def F(x):
        X = x.reshape(N, 3)
        def cot(p, q, apex):
            a, b = p - apex, q - apex
            c = np.cross(a, b)
            return (a * b).sum(-1) / np.sqrt((c * c).sum(-1))
        p0, p1, p2, p3 = X[v[0]], X[v[1]], X[v[2]], X[v[3]]
        wgt = 0.5 * (cot(p0, p1, p2) + cot(p0, p1, p3))
        return np.concatenate([(w_fit * (X - A)).reshape(-1), (w_reg * wgt[:, None] * (p1 - p0)).reshape(-1)])

    got_x, got_costs = run(torch, "cotan_smooth.t", (N, E), [w_fit, w_reg, X0.copy(), A, v[0], v[1], v[2], v[3]], [2], dbl, nIterations=4, lIterations=12)
    ref_x, ref_costs = mirror(F, X0.reshape(-1), np.ones(3 * N, bool), 4, 12, precond=True)
    check(got_x, got_costs, ref_x, ref_costs, dbl)


@pytest.mark.parametrize("dbl", [True, False])
def test_robust_point_to_plane(torch, dbl):
    N = 60
    rng = np.random.default_rng(12)
    T = rng.uniform(-1, 1, (N, 3))
    Nrm = rng.standard_normal((N, 3)); Nrm /= np.linalg.norm(Nrm, axis=1, keepdims=True)
    X0 = T + 0.2 * rng.standard_normal((N, 3))
    R0 = rng.uniform(0.6, 1.0, N)
    nb = rng.permutation(N).astype(np.int32)
    w_rob = float(np.float32(0.35))

    def F(x):
        X = x[:3 * N].reshape(N, 3); R = x[3 * N:]
        d = (Nrm * (X - T)).sum(-1)
        return np.concatenate([R * R * d, w_rob * (1.0 - R * R), (0.3 * (X - X[nb])).reshape(-1)])

    x0 = np.concatenate([X0.reshape(-1), R0])
    ref_x, ref_costs = mirror(F, x0, np.ones(4 * N, bool), 5, 10, precond=True)
    got_x, got_costs = run(torch, "robust_plane.t", (N,), [X0.copy(), R0.copy(), T, Nrm, w_rob, nb], [0, 1], dbl, nIterations=5, lIterations=10)
    check(got_x, got_costs, ref_x, ref_costs, dbl)


def _rotate3d(a, v):
    """lib.t:123-137 (Euler angles alpha, beta, gamma -> R v), on (..., 3) arrays; complex-capable"""
    ca, cb, cg, sa, sb, sg = np.cos(a[..., 0]), np.cos(a[..., 1]), np.cos(a[..., 2]), np.sin(a[..., 0]), np.sin(a[..., 1]), np.sin(a[..., 2])
    m = [cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca, sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca, -sb, cb * sa, cb * ca]
    return np.stack([m[0] * v[..., 0] + m[1] * v[..., 1] + m[2] * v[..., 2], m[3] * v[..., 0] + m[4] * v[..., 1] + m[5] * v[..., 2],
                     m[6] * v[..., 0] + m[7] * v[..., 1] + m[8] * v[..., 2]], -1)


@pytest.mark.parametrize("dbl", [True, False])
def test_lattice_arap_over_a_three_dimensional_domain(torch, dbl):
    W, H, D = 5, 4, 3
    rng = np.random.default_rng(21)
    gz, gy, gx = np.meshgrid(np.arange(D, dtype=np.float64), np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    Rest = np.stack([gx, gy, gz], -1)                                     # [D][H][W][3]: x fastest, like the solver's images
    Tgt = np.full_like(Rest, -1e6)
    Tgt[0, :, 0] = Rest[0, :, 0]                                          # one lattice line held in place ...
    Tgt[D - 1, :, W - 1] = Rest[D - 1, :, W - 1] + np.array([0.6, 0.2, -0.3])     # ... the opposite one pulled away
    Pos0 = Rest + 0.05 * rng.standard_normal(Rest.shape)
    Ang0 = 0.05 * rng.standard_normal(Rest.shape)
    w_fit, w_reg = float(np.float32(2.0)), float(np.float32(0.9))
    n = W * H * D * 3

    def F(x):
        P = x[:n].reshape(D, H, W, 3); A = x[n:].reshape(D, H, W, 3)
        held = Tgt[..., 0] >= -999999.9
        out = [np.where(held[..., None], w_fit * (P - Tgt), 0.0).reshape(-1)]
        for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
            r = np.zeros_like(P)
            s = (slice(max(0, -dz), D - max(0, dz)), slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx)))
            t = (slice(max(0, dz), D - max(0, -dz)), slice(max(0, dy), H - max(0, -dy)), slice(max(0, dx), W - max(0, -dx)))
            r[s] = w_reg * ((P[s] - P[t]) - _rotate3d(A[s], Rest[s] - Rest[t]))
            out.append(r.reshape(-1))
        return np.concatenate(out)

    x0 = np.concatenate([Pos0.reshape(-1), Ang0.reshape(-1)])
    got_x, got_costs = run(torch, "volume_arap.t", (W, H, D), [w_fit, w_reg, Pos0.copy(), Ang0.copy(), Rest, Tgt], [2, 3], dbl, nIterations=4, lIterations=15)
    ref_x, ref_costs = mirror(F, x0, np.ones(2 * n, bool), 4, 15, precond=True)
    check(got_x, got_costs, ref_x, ref_costs, dbl)


@pytest.mark.parametrize("dbl", [True, False])
def test_two_unknown_index_spaces(torch, dbl):
    """tests/energies/two_domains.t: unknowns over {U} and over {N}, a residual on the product domain {N, U} (every S(u) gathers N terms, every P(n) U terms) and one on
    {N}: linear, so the mirror is exact."""
    N, U = 40, 3
    rng = np.random.default_rng(4)
    T = rng.uniform(-1, 1, N)
    S0 = rng.uniform(-0.5, 0.5, U); P0 = rng.uniform(-1, 1, N)

    def F(x):
        S, P = x[:U], x[U:]
        fit = (S[None, :] + P[:, None] - T[:, None])          # element (n, u); the order of the residuals does not matter to J^T J
        return np.concatenate([fit.reshape(-1), 0.5 * P])

    got_x, got_costs = run(torch, "two_domains.t", (N, U), [S0.copy(), P0.copy(), T], [0, 1], dbl, nIterations=3, lIterations=8)
    ref_x, ref_costs = mirror(F, np.concatenate([S0, P0]), np.ones(N + U, bool), 3, 8, precond=True)
    check(got_x, got_costs, ref_x, ref_costs, dbl)


@pytest.mark.parametrize("dbl,KS", [(True, 11), (False, 11), (True, 17)])
def test_deconvolution_with_a_window_wider_than_the_dual_width(torch, tmp_path, dbl, KS):
    """tests/energies/conv2d_wide.t: an 11 x 11 Sum, 121 unknown accesses per residual -- more than the forward-mode lowering carries at once (48), so the front-end's WIDE
    lowering runs (32 partials per evaluation of the residual, chunk after chunk) -- and the same file with a 17 x 17 window (289 accesses, ten chunks: the size of the
    reference's spatially_varying_deconvolution as shipped)."""
    import time
    W, H = 40, 30
    rng = np.random.default_rng(31)
    Ker = rng.uniform(0.0, 1.0, (KS, KS)); Ker /= Ker.sum()
    truth = rng.uniform(0.0, 1.0, (H, W))
    c = KS // 2

    def blur(X):
        out = np.zeros((H, W), dtype=X.dtype)
        for ky in range(KS):
            for kx in range(KS):
                out[c:H - c, c:W - c] += Ker[ky, kx] * X[ky:H - 2 * c + ky, kx:W - 2 * c + kx]
        return out

    B = blur(truth) + 1e-3 * rng.standard_normal((H, W))
    X0 = B.copy()

    def F(x):
        X = x.reshape(H, W)
        data = np.zeros((H, W), dtype=X.dtype)
        data[c:H - c, c:W - c] = (blur(X) - B)[c:H - c, c:W - c]
        return np.concatenate([data.reshape(-1), 0.05 * x])

    text = open(os.path.join(HERE, "conv2d_wide.t")).read()
    if KS != 11: text = text.replace("kx - 5", f"kx - {c}").replace("ky - 5", f"ky - {c}").replace("InBoundsExpanded(x, y, 5)", f"InBoundsExpanded(x, y, {c})")
    (tmp_path / "conv2d_wide.t").write_text(text)
    t0 = time.time()
    got_x, got_costs = run(torch, "conv2d_wide.t", (W, H, KS, KS), [X0.copy(), B, np.ascontiguousarray(Ker)], [0], dbl, here=str(tmp_path), nIterations=3, lIterations=3)
    print(f"plan + solve of the wide unit ({KS} x {KS}):", round(time.time() - t0, 1), "s")
    # (three PCG iterations per step: deconvolution is ill-conditioned, and a long CG run near convergence amplifies the last bit -- the order of the atomics -- to 1e-5
    #  by the second step of 12; the first step of 12 agrees to 1e-17, three steps of 3 to 1e-15)
    ref_x, ref_costs = mirror(F, X0.reshape(-1), np.ones(W * H, bool), 3, 3, precond=True)
    check(got_x, got_costs, ref_x, ref_costs, dbl)


@pytest.mark.parametrize("dbl", [True, False])
def test_one_rigid_motion_for_a_point_set(torch, dbl):
    """tests/energies/rigid_fit.t: six unknowns shared by every residual over the product domain {N, U = 1}; in single precision the file's schedule lines select the dense
    [JtJ]p schedule (J^T J formed by v_mfma_f32_32x32x2_f32 tiles, spmv_kernels.hip), in double the inline one."""
    N = 200
    rng = np.random.default_rng(17)
    Rest = rng.uniform(-1, 1, (N, 3))
    ang = np.array([0.3, -0.2, 0.25]); sh = np.array([0.4, -0.1, 0.2])
    Goal = _rotate3d(np.broadcast_to(ang, (N, 3)), Rest) + sh + 0.01 * rng.standard_normal((N, 3))
    Goal[rng.uniform(size=N) < 0.2] = -1e6
    x0 = np.zeros(6)

    def F(x):
        Shift, Euler = x[:3], x[3:]
        r = _rotate3d(np.broadcast_to(Euler, (N, 3)).astype(x.dtype), Rest.astype(x.dtype)) + Shift - Goal
        return np.where((Goal[:, :1] >= -999999.9), r, 0.0).reshape(-1)

    dev_params = [np.zeros((1, 3)), np.zeros((1, 3)), Rest, Goal]
    dev = []
    for p in dev_params: dev.append(torch.from_numpy(p if dbl else p.astype(np.float32)).cuda())
    s = api.ThalloSolver((N, 1), os.path.join(HERE, "rigid_fit.t"), double_precision=dbl)
    if not dbl: assert s.schedule_name == "dense [JtJ]p", s.schedule_name
    final, costs = s.solve(dev, profiled=True, nIterations=5, lIterations=6)
    s.close()
    got_x = np.concatenate([to_host(dev[0]).astype(np.float64).reshape(-1), to_host(dev[1]).astype(np.float64).reshape(-1)])
    ref_x, ref_costs = mirror(F, x0, np.ones(6, bool), 5, 6, precond=True)
    check(got_x, np.array(costs), ref_x, ref_costs, dbl)
    assert np.abs(ref_x[3:] - ang).max() < 0.02 and np.abs(ref_x[:3] - sh).max() < 0.02          # (the motion the goals were made with comes back)


@pytest.mark.parametrize("dbl", [True, False])
def test_embedded_deformation_nodes(torch, dbl):
    """tests/energies/embedded_nodes.t: a nine-channel unknown (a free 3 x 3 matrix per node) next to a three-channel one, gemv through Sparse maps, a residual with six
    components."""
    nx, ny = 6, 5
    N = nx * ny
    rng = np.random.default_rng(23)
    gx, gy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64))
    Rest = np.stack([gx, gy, 0.2 * np.sin(gx + 0.5 * gy)], -1).reshape(N, 3)
    Goal = np.full((N, 3), -1e6); Goal[:nx] = Rest[:nx]; Goal[-nx:] = Rest[-nx:] + np.array([0.3, 0.1, 0.5])
    vid = lambda i, j: j * nx + i
    pairs = [(vid(i, j), vid(i + 1, j)) for j in range(ny) for i in range(nx - 1)] + [(vid(i, j), vid(i, j + 1)) for j in range(ny - 1) for i in range(nx)]
    pairs = pairs + [(q, p) for p, q in pairs]
    a = np.array([p for p, _ in pairs], np.int32); b = np.array([q for _, q in pairs], np.int32)
    E = len(pairs)
    Off0 = Rest + 0.02 * rng.standard_normal((N, 3))
    M0 = np.tile(np.eye(3).reshape(1, 9), (N, 1)) + 0.02 * rng.standard_normal((N, 9))
    w_fit, w_reg, w_rot = float(np.float32(3.0)), float(np.float32(1.0)), float(np.float32(2.0))

    def F(x):
        Off = x[:3 * N].reshape(N, 3); M = x[3 * N:].reshape(N, 3, 3)
        fit = np.where(Goal[:, :1] >= -999999.9, w_fit * (Off - Goal), 0.0)
        d = Rest[b] - Rest[a]
        edge = (Off[b] - Off[a]) - np.einsum("eij,ej->ei", M[a], d.astype(x.dtype))
        c0, c1, c2 = M[:, :, 0], M[:, :, 1], M[:, :, 2]
        dot = lambda u, v: (u * v).sum(-1)
        rot = w_rot * np.stack([dot(c0, c1), dot(c0, c2), dot(c1, c2), dot(c0, c0) - 1, dot(c1, c1) - 1, dot(c2, c2) - 1], -1)
        return np.concatenate([fit.reshape(-1), (w_reg * edge).reshape(-1), rot.reshape(-1)])

    x0 = np.concatenate([Off0.reshape(-1), M0.reshape(-1)])
    got_x, got_costs = run(torch, "embedded_nodes.t", (N, E), [w_fit, w_reg, w_rot, Off0.copy(), M0.copy(), Rest, Goal, a, b], [3, 4], dbl, nIterations=4, lIterations=10)
    ref_x, ref_costs = mirror(F, x0, np.ones(12 * N, bool), 4, 10, precond=True)
    check(got_x, got_costs, ref_x, ref_costs, dbl)
