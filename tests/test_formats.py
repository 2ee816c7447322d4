"""Data formats on either side of the hot path (thallo_amd/formats.py): round trips on synthetic data and the two
small data fixtures of the reference's image_warping example (tests/golden/cat512_mask.png, cat512.constraints: data, not code)."""
import os

import numpy as np
import pytest

from thallo_amd import formats as F

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_constraints_fixture_and_roundtrip(tmp_path):
    c = F.read_constraints(os.path.join(GOLD, "cat512.constraints"))
    assert c.shape == (9, 4) and c[0].tolist() == [30, 132, 59, 44] and c[-1].tolist() == [92, 192, 84, 192]
    p = tmp_path / "c.constraints"
    F.write_constraints(p, c)
    assert (F.read_constraints(p) == c).all()
    full = F.add_border_constraints(c, 512, 512)
    assert len(full) == 9 + 4 * 512 - 4 and full[9].tolist() == [0, 0, 0, 0] and full[-1].tolist() == [511, 511, 511, 511]


def test_png_fixture_and_roundtrip(tmp_path):
    m = F.read_png(os.path.join(GOLD, "cat512_mask.png"))
    assert m.shape == (512, 512, 4) and m.dtype == np.uint8
    active = int((m[:, :, 0] == 0).sum())
    assert 0 < active < 512 * 512                 # the harness prints this as numActivePixels (main.cpp:104-113)
    rng = np.random.default_rng(0)
    for shape in [(7, 5), (9, 4, 3), (3, 11, 4), (5, 5, 2)]:
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = tmp_path / "a.png"
        F.write_png(p, a)
        b = F.read_png(p)
        assert (b.reshape(a.shape) == a).all()
    pil = pytest.importorskip("PIL.Image")        # cross-check every filter type against an independent decoder
    a = (np.add.outer(np.arange(64), np.arange(48)) % 256).astype(np.uint8)
    a = np.stack([a, a[::-1], a.T[:64, :48] if a.T.shape == a.shape else a], axis=2)
    p = tmp_path / "pil.png"
    pil.fromarray(a).save(p, optimize=True)
    assert (F.read_png(p) == a).all()


def test_constraint_image_interpolation():
    c = np.array([[2, 3, 6, 1], [1, 1, 5, 5]])
    mask = np.zeros((8, 8), dtype=np.float32); mask[1, 1] = 255
    img = F.constraint_image(c, mask, alpha=0.5)
    assert img.shape == (8, 8, 2) and img[3, 2].tolist() == [4.0, 2.0]
    assert img[1, 1].tolist() == [-1.0, -1.0]     # masked marker is dropped (CombinedSolver.h:192)
    assert (img == -1).sum() == 8 * 8 * 2 - 2


def test_imagedump_and_sfs_params(tmp_path):
    a = np.random.default_rng(1).standard_normal((6, 9)).astype(np.float32)
    a[0, 0], a[1, 1] = np.inf, -np.inf
    p = tmp_path / "d.imagedump"
    F.write_imagedump(p, a)
    b = F.read_imagedump(p)
    assert b.shape == (6, 9, 1) and b[0, 0, 0] == np.finfo(np.float32).max and b[1, 1, 0] == -10000.0
    assert (b[2:, :, 0] == a[2:]).all()
    assert np.isinf(F.read_imagedump(p, clamp_infinity=False)[0, 0, 0])
    u = np.arange(24, dtype=np.uint8).reshape(4, 6)
    F.write_imagedump(p, u)
    assert (F.read_imagedump(p)[:, :, 0] == u).all()
    d = {k: float(i + 1) for i, k in enumerate(F._SFS_FIELDS)}
    d["deltaTransform"] = np.eye(4, dtype=np.float32); d["lightingCoefficients"] = np.arange(9, dtype=np.float32) / 8; d["unused"] = (1, 10, 8)
    q = tmp_path / "x.SFSSolverParameters"
    F.write_sfs_params(q, d)
    assert os.path.getsize(q) == 160
    e = F.read_sfs_params(q)
    assert all(e[k] == d[k] for k in F._SFS_FIELDS) and (e["lightingCoefficients"] == d["lightingCoefficients"]).all() and e["unused"] == (1, 10, 8)


def test_bal_roundtrip_and_coherence_sort(tmp_path):
    rng = np.random.default_rng(2)
    C, P, O = 3, 7, 15
    cams, pts = rng.standard_normal((C, 9)), rng.standard_normal((P, 3))
    ci, pi = rng.integers(0, C, O), rng.integers(0, P, O)
    obs = rng.standard_normal((O, 2)) * 100
    p = tmp_path / "problem.txt"
    F.write_bal(p, cams, pts, obs, ci, pi)
    raw = F.read_bal(p, sort_for_coherency=False)
    assert np.allclose(raw["cameras"], cams) and np.allclose(raw["points"], pts) and (raw["cam_idx"] == ci).all() and (raw["pt_idx"] == pi).all()
    assert np.allclose(raw["observations"], obs, rtol=1e-6)
    s = F.read_bal(p)
    key = s["cam_idx"].astype(np.int64) * P + s["pt_idx"]
    assert (np.diff(key) >= 0).all() and sorted(key.tolist()) == sorted((ci.astype(np.int64) * P + pi).tolist())


def test_meshes_and_landmarks(tmp_path):
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    faces = [[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]]
    for writer, reader, name in ((F.write_off, F.read_off, "m.off"), (lambda p, v, f: F.write_ply(p, v, f, True), F.read_ply, "b.ply"),
                                 (lambda p, v, f: F.write_ply(p, v, f, False), F.read_ply, "a.ply")):
        p = tmp_path / name
        writer(p, V, faces)
        V2, f2 = reader(p)
        assert np.allclose(V2, V) and [list(x) for x in f2] == faces
    v0, v1 = F.mesh_directed_edges(faces, 4)
    assert len(v0) == 12 and v0.tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3] and v1[:3].tolist() == [1, 2, 3]
    p = tmp_path / "m.mrk"
    F.write_mrk(p, [3, 1], np.array([[0.5, 0.25, 1], [2, 3, 4]], dtype=np.float32))
    idx, tgt = F.read_mrk(p)
    assert idx.tolist() == [3, 1] and np.allclose(tgt, [[0.5, 0.25, 1], [2, 3, 4]])
