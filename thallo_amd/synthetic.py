"""Seeded synthetic problem instances (host numpy arrays) for the bundled energies.

The shapes follow the reference's own example harnesses; there is no network for the real
data sets, and only decoded pixels of the two gold PNGs travel as fixtures.

  image_warping  examples/image_warping/src/CombinedSolver.h:158-204 (UrShape = pixel grid,
                 Offset0 = UrShape, Angle0 = 0, constraints = -1 except markers + pinned border,
                 weights sqrt(100)/sqrt(0.01) :123-127) -- perturbed so J is non-trivial at iter 0
                 (SURVEY.md 8d "C-main").
  arap_mesh      examples/arap_mesh_deformation/src/CombinedSolver.h:94-133 (directed edge pairs,
                 unconstrained vertices carry -inf-like constraints :80), weights 4/1 main.cpp:115-116.
"""
import numpy as np


def image_warping(W, H, seed=1234, perturb=0.5, angle_amp=0.1, n_markers=64, max_disp=32.0,
                  mask_disc=0.1, w_fit=10.0, w_reg=0.1):
    """Returns the params list indexed like image_warping.t Inputs{} (0..6)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    ur = np.stack([xs, ys], -1).astype(np.float32)                      # UrShape(x,y) = (x,y)
    off = (ur + perturb * rng.uniform(-1, 1, ur.shape)).astype(np.float32)
    ang = (angle_amp * rng.uniform(-1, 1, (H, W))).astype(np.float32)
    mask = np.zeros((H, W), np.float32)
    if mask_disc > 0:
        rr = (xs - W / 2.0) ** 2 + (ys - H / 2.0) ** 2
        mask[rr < (mask_disc * W) ** 2] = 255.0
    cons = np.full((H, W, 2), -1.0, np.float32)
    border = (xs == 0) | (ys == 0) | (xs == W - 1) | (ys == H - 1)
    cons[border] = ur[border]                                           # main.cpp:119-129
    scale = max_disp * min(W, H) / 2048.0 if min(W, H) < 2048 else max_disp
    placed = 0
    while placed < n_markers:
        x = int(rng.integers(1, W - 1)); y = int(rng.integers(1, H - 1))
        if mask[y, x] != 0 or cons[y, x, 0] >= 0:
            continue
        d = rng.uniform(-scale, scale, 2)
        t = np.clip(np.array([x, y]) + d, 0, [W - 1, H - 1])
        cons[y, x] = t
        placed += 1
    return [np.ascontiguousarray(off), np.ascontiguousarray(ang), np.ascontiguousarray(ur),
            np.ascontiguousarray(cons), np.ascontiguousarray(mask), float(w_fit), float(w_reg)]


def laplacian_image(W, H, seed=3):
    rng = np.random.default_rng(seed)
    A = rng.uniform(0, 1, (H, W)).astype(np.float32)
    return [A.copy(), A]


def laplacian_graph(N, seed=5, extra_edges=0):
    rng = np.random.default_rng(seed)
    A = rng.uniform(0, 1, N).astype(np.float32)
    v0 = np.arange(N - 1, dtype=np.int32)
    v1 = v0 + 1
    if extra_edges:
        a = rng.integers(0, N, extra_edges).astype(np.int32)
        b = rng.integers(0, N, extra_edges).astype(np.int32)
        keep = a != b
        v0 = np.concatenate([v0, a[keep]]); v1 = np.concatenate([v1, b[keep]])
    return [A.copy(), A, np.ascontiguousarray(v0), np.ascontiguousarray(v1)]


def torus_mesh(nu, nv, R=2.0, r=0.7):
    """Closed triangle mesh on a torus grid: nu*nv vertices, valence 6.
    Returns (positions [N,3] float32, directed edge arrays v0, v1 int32) -- both directions of every
    undirected edge, like ThalloGraph (examples/shared/ThalloGraph.h:67-79)."""
    u, v = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    tu = 2 * np.pi * u / nu; tv = 2 * np.pi * v / nv
    pos = np.stack([(R + r * np.cos(tv)) * np.cos(tu), (R + r * np.cos(tv)) * np.sin(tu), r * np.sin(tv)], -1)
    idx = (u * nv + v).astype(np.int32)

    def nb(du, dv):
        return (((u + du) % nu) * nv + (v + dv) % nv).astype(np.int32)
    und = [(idx, nb(1, 0)), (idx, nb(0, 1)), (idx, nb(1, 1))]          # 3 undirected edges per vertex
    a = np.concatenate([p[0].ravel() for p in und]); b = np.concatenate([p[1].ravel() for p in und])
    v0 = np.concatenate([a, b]); v1 = np.concatenate([b, a])
    return pos.reshape(-1, 3).astype(np.float32), np.ascontiguousarray(v0), np.ascontiguousarray(v1)


def arap_mesh(nu, nv, seed=11, n_handles=32, w_fit=4.0, w_reg=1.0, angle_amp=0.05, pos_noise=0.01):
    """Returns the params list indexed like arap_mesh_deformation.t Inputs{} (0..7)."""
    rng = np.random.default_rng(seed)
    orig, v0, v1 = torus_mesh(nu, nv)
    N = orig.shape[0]
    pos = (orig + pos_noise * rng.standard_normal(orig.shape)).astype(np.float32)
    ang = (angle_amp * rng.uniform(-1, 1, (N, 3))).astype(np.float32)
    cons = np.full((N, 3), -np.inf, np.float32)
    cons[:] = -1.0e30                                                  # "unconstrained" sentinel < -999999.9
    hs = rng.choice(N, n_handles, replace=False)
    cons[hs] = orig[hs] + 0.3 * rng.standard_normal((n_handles, 3)).astype(np.float32)
    return [float(w_fit), float(w_reg), pos, ang, orig.copy(), np.ascontiguousarray(cons), v0, v1]


# ------------------------------------------------------------------ bundle adjustment (BAL-shaped)
def _rodrigues_inverse(R):
    """rotation matrix -> angle-axis (float64)"""
    ang = np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))
    if ang < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * np.sin(ang))
    return ax * ang


def ba_project(cams, pts, ci, pi):
    """Snavely projection in float64 (bundle_adjustment.t:12-31): predicted 2-D positions of observations."""
    c = cams[ci].astype(np.float64); X = pts[pi].astype(np.float64)
    aa = c[:, :3]
    th2 = (aa * aa).sum(1)
    th = np.sqrt(np.where(th2 > 1e-8, th2, 1.0))
    w = aa / th[:, None]
    ct, st = np.cos(th), np.sin(th)
    big = X * ct[:, None] + np.cross(w, X) * st[:, None] + w * ((w * X).sum(1) * (1 - ct))[:, None]
    small = X + np.cross(aa, X)
    p = np.where((th2 > 1e-8)[:, None], big, small) + c[:, 3:6]
    cd = -p[:, :2] / p[:, 2:3]
    r2 = (cd * cd).sum(1)
    dist = 1.0 + r2 * (c[:, 7] + c[:, 8] * r2)
    return cd * (c[:, 6] * dist)[:, None]


def bundle_adjustment(C=1723, P=156502, O=678718, seed=7, band=32, noise_px=1.0, cam_noise=1e-3, pt_noise=2e-2):
    """BAL-shaped synthetic instance: cameras on a ring looking at a point blob, banded
    visibility, >= 2 observations per point.  Returns the params list of bundle_adjustment.t Inputs{} (0..4):
    cameras [C,9], points [P,3], observations [O,2], oToC [O] int32, oToP [O] int32 -- unknowns perturbed from
    the ground truth that generated the (noisy) observations."""
    rng = np.random.default_rng(seed)
    assert O >= 2 * P and C >= 2
    pts = rng.normal(0.0, 1.0, (P, 3)) * np.array([1.5, 0.6, 1.5])
    cams = np.zeros((C, 9))
    for i in range(C):
        phi = 2 * np.pi * i / C
        pos = np.array([12.0 * np.cos(phi), 1.0 + 0.5 * np.sin(3 * phi), 12.0 * np.sin(phi)])
        zc = pos / np.linalg.norm(pos)                       # camera looks down -z at the origin
        xc = np.cross([0.0, 1.0, 0.0], zc); xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc])                           # world -> camera
        cams[i, :3] = _rodrigues_inverse(R)
        cams[i, 3:6] = -R @ pos
        cams[i, 6] = 800.0 + 40.0 * rng.standard_normal()
        cams[i, 7] = -1e-2 * rng.random(); cams[i, 8] = 1e-3 * rng.random()
    # visibility: point j is seen by cnt_j cameras inside a band around its base camera
    cnt = np.full(P, O // P, np.int64); cnt[: O - cnt.sum()] += 1
    band = min(band, C)
    assert cnt.max() <= band
    oToP = np.repeat(np.arange(P, dtype=np.int64), cnt)
    base = (np.arange(P, dtype=np.int64) * C) // P
    offs = np.argsort(rng.random((P, band)), axis=1)         # a random subset of the band per point
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    within = np.arange(O) - start[oToP]
    oToC = (base[oToP] + offs[oToP, within] - band // 2) % C
    obs = ba_project(cams, pts, oToC, oToP) + noise_px * rng.standard_normal((O, 2))
    cams0 = cams + cam_noise * rng.standard_normal(cams.shape) * np.array([1, 1, 1, 10, 10, 10, 1000, 1, 0.1])
    pts0 = pts + pt_noise * rng.standard_normal(pts.shape)
    return [np.ascontiguousarray(cams0, np.float32), np.ascontiguousarray(pts0, np.float32), np.ascontiguousarray(obs, np.float32),
            np.ascontiguousarray(oToC, np.int32), np.ascontiguousarray(oToP, np.int32)]


# ------------------------------------------------------------------ shape from shading
SFS_LIGHTING = (0.6908, 0.0446, 0.0181, -0.1773, -0.0407, 0.1447, 0.0239, -0.2466, 0.0058)   # shipped default_* data set


def _sfs_shift(a, dx, dy):
    """b[y,x] = a[y+dy, x+dx], 0 outside (the reference's guarded loads)"""
    out = np.zeros_like(a)
    H, W = a.shape
    ys = slice(max(0, -dy), min(H, H - dy)); xs = slice(max(0, -dx), min(W, W - dx))
    yd = slice(max(0, dy), min(H, H + dy)); xd = slice(max(0, dx), min(W, W + dx))
    out[ys, xs] = a[yd, xd]
    return out


def sfs_BI(X, D, Im, fx, fy, ux, uy, L):
    """B_I image of shape_from_shading.t:44-80 in float64 (vectorised): shading minus smoothed intensity, 0 where invalid."""
    X = X.astype(np.float64); H, W = X.shape
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    l, u = _sfs_shift(X, -1, 0), _sfs_shift(X, 0, -1)
    nx = u * (X - l) / fy
    ny = l * (X - u) / fx
    nz = nx * (ux - xs) / fx + ny * (uy - ys) / fy - l * u / (fx * fy)
    sq = nx * nx + ny * ny + nz * nz
    inv = np.where(sq > 0, 1.0 / np.sqrt(np.where(sq > 0, sq, 1.0)), 1.0)
    nx, ny, nz = inv * nx, inv * ny, inv * nz
    B = (L[0] + L[1] * ny + L[2] * nz + L[3] * nx + L[4] * nx * ny + L[5] * ny * nz +
         L[6] * (-nx * nx - ny * ny + 2 * nz * nz) + L[7] * nz * nx + L[8] * (nx * nx - ny * ny))
    Imd = Im.astype(np.float64)
    I = Imd * 0.5 + 0.25 * (_sfs_shift(Imd, -1, 0) + _sfs_shift(Imd, 0, -1))
    Dd = D.astype(np.float64)
    valid = (_sfs_shift(Dd, -1, 0) > 0) & (Dd > 0) & (_sfs_shift(Dd, 0, -1) > 0)
    return np.where(valid, B - I, 0.0)


def sfs_residuals(params):
    """All residuals of shape_from_shading.t in float64, shape [H, W, 6] = fit, shading_h, shading_v, reg xyz."""
    wp, ws, wg = (np.sqrt(float(params[k])) for k in range(3))
    fx, fy, ux, uy = (float(params[k]) for k in range(3, 7))
    L = [float(params[k]) for k in range(7, 16)]
    X, D, Im, mR, mC = (np.asarray(params[k]) for k in range(16, 21))
    Xd, Dd = X.astype(np.float64), D.astype(np.float64)
    H, W = X.shape
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    out = np.zeros((H, W, 6))
    out[..., 0] = np.where(Dd > 0, wp * (Xd - Dd), 0.0)
    BI = sfs_BI(X, D, Im, fx, fy, ux, uy, L)
    inner = (xs >= 1) & (xs <= W - 2) & (ys >= 1) & (ys <= H - 2)
    out[..., 1] = np.where(inner, wg * (BI - _sfs_shift(BI, 1, 0)) * mR, 0.0)
    out[..., 2] = np.where(inner, wg * (BI - _sfs_shift(BI, 0, 1)) * mC, 0.0)
    valid = Dd > 0
    acc = [4 * ((xs - ux) / fx) * Xd, 4 * ((ys - uy) / fy) * Xd, 4 * Xd]
    for dx, dy in ((-1, 0), (0, -1), (1, 0), (0, 1)):
        Xn, Dn = _sfs_shift(Xd, dx, dy), _sfs_shift(Dd, dx, dy)
        valid &= (Dn > 0) & (np.abs(X.astype(np.float32) - _sfs_shift(X.astype(np.float32), dx, dy)) < np.float32(0.01))
        acc[0] -= ((xs + dx - ux) / fx) * Xn; acc[1] -= ((ys + dy - uy) / fy) * Xn; acc[2] -= Xn
    for c in range(3):
        out[..., 3 + c] = np.where(valid, ws * acc[c], 0.0)
    return out


def shape_from_shading(W, H, seed=21, w_p=100.0, w_s=100.0, w_g=1.0, noise=5e-4, hole=True):
    """Synthetic SFS instance shaped like the shipped 640x480 data set (examples/shape_from_shading/src/SFSSolverInput.h:20-44):
    a smooth depth map (metres) with a small invalid region, intensity rendered from it with the shipped SH lighting,
    random 0/1 edge masks, unknown initialised to the noisy depth.  Returns params indexed like the .t Inputs{} (0..20)."""
    rng = np.random.default_rng(seed)
    fx = fy = 574.05 * W / 640.0
    ux, uy = W / 2.0, H / 2.0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    a = min(0.12, 0.002 * min(W, H) / (2 * np.pi))            # keeps |dX| per pixel well under the 0.01 m continuity gate
    depth = 1.0 + a * np.sin(2 * np.pi * xs / W) * np.cos(2 * np.pi * ys / H) + 0.3 * a * np.cos(4 * np.pi * xs / W + 1.0)
    D = depth.copy()
    if hole and min(W, H) >= 12:
        rr = (xs - 0.7 * W) ** 2 + (ys - 0.3 * H) ** 2
        D[rr < (0.06 * min(W, H)) ** 2] = 0.0
    L = SFS_LIGHTING
    # intensity of the true surface: B with I = 0, smoothing undone approximately by using B itself
    BI0 = sfs_BI(depth, np.ones_like(depth), np.zeros_like(depth), fx, fy, ux, uy, L)
    Im = np.clip(BI0 + 0.01 * rng.standard_normal((H, W)), 0.0, 2.0)
    X0 = np.where(D > 0, depth + noise * rng.standard_normal((H, W)), depth)
    mR = (rng.random((H, W)) < 0.9).astype(np.uint8)
    mC = (rng.random((H, W)) < 0.9).astype(np.uint8)
    f32 = lambda v: np.ascontiguousarray(v, np.float32)
    return [float(w_p), float(w_s), float(w_g), float(fx), float(fy), float(ux), float(uy)] + [float(v) for v in L] + \
           [f32(X0), f32(D), f32(Im), np.ascontiguousarray(mR), np.ascontiguousarray(mC)]
