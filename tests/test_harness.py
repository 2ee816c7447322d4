"""The application-side harness (harness/image_warping): the reference's image_warping example over libThallo.so
with the reference's artefacts (finalCosts.json, perf.json, results/results_float.csv)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from thallo_amd import formats as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "harness", "image_warping")
BIN_SFS = os.path.join(ROOT, "harness", "shape_from_shading")
BIN_ARAP = os.path.join(ROOT, "harness", "arap_mesh_deformation")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "harness")], check=True, stdout=subprocess.DEVNULL)


def test_harness_reads_the_reference_inputs(tmp_path):
    """C++ PNG / .constraints readers == the python readers on the reference's cat512 mask and marker list (no GPU touched)."""
    _build()
    shutil.copy(os.path.join(GOLD, "cat512_mask.png"), tmp_path / "cat512_mask.png")
    shutil.copy(os.path.join(GOLD, "cat512.constraints"), tmp_path / "cat512.constraints")
    out = subprocess.run([BIN, str(tmp_path / "cat512.png"), "--io-only"], check=True, capture_output=True, text=True).stdout
    m = F.read_png(os.path.join(GOLD, "cat512_mask.png"))
    c = F.read_constraints(os.path.join(GOLD, "cat512.constraints"))
    assert "width 512, height 512" in out
    assert f"numActivePixels: {int((m[:, :, 0] == 0).sum())}" in out
    assert f"markers: 9 checksum {int((c * np.array([1, 3, 5, 7])).sum())}" in out
    out2 = subprocess.run([BIN, str(tmp_path / "cat512.png"), "--io-only", "-d", "4"], check=True, capture_output=True, text=True).stdout
    assert "width 128, height 128" in out2 and f"numActivePixels: {int((m[::4, ::4, 0] == 0).sum())}" in out2


@pytest.mark.gpu
def test_harness_continuation_matches_the_python_driver(tmp_path):
    """Outer continuation (marker targets interpolated over numIter solves) on a small instance written as PNG + .constraints:
    the harness' finalCosts.json / results csv equal the same procedure driven through the python mirror of the API."""
    import torch
    import thallo_amd
    from thallo_amd import api
    assert torch.cuda.is_available(), "this test needs the MI355X"
    _build()
    W, H, numIter, nIt, lIt = 96, 80, 3, 3, 25
    mask = np.zeros((H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    mask[(xx - 50) ** 2 + (yy - 40) ** 2 < 64] = 255
    cons = np.array([[20, 20, 26, 15], [70, 22, 64, 30], [30, 60, 35, 66], [75, 62, 70, 55], [50, 40, 10, 10]])   # the last one sits on the mask: dropped
    F.write_png(tmp_path / "toy_mask.png", np.stack([mask] * 3, axis=2))
    F.write_constraints(tmp_path / "toy.constraints", cons)
    energy = thallo_amd.energy_file("image_warping")
    r = subprocess.run([BIN, str(tmp_path / "toy.png"), "-o", energy, "-n", str(numIter), "-N", str(nIt), "-L", str(lIt), "--profile"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    fc = json.load(open(tmp_path / "finalCosts.json"))
    perf = json.load(open(tmp_path / "perf.json"))
    assert fc["name"] == "Image Warping" and list(fc["costs"]) == ["ThalloGN"]
    assert set(perf["performance"]["ThalloGN"]) == {"total", "nonlinearIteration", "nonlinearSetup", "linearSolve", "nonlinearResolve"}
    assert perf["performance"]["ThalloGN"]["linearSolve"]["count"] == nIt
    rows = [ln.split(",") for ln in open(tmp_path / "results" / "results_float.csv").read().strip().splitlines()]
    assert rows[0][0] == "Iter" and len(rows) == 1 + numIter * (nIt + 1)
    csv_costs = np.array([float(rw[2]) for rw in rows[1:]])
    # the same procedure through the python mirror
    full = F.add_border_constraints(cons, W, H)
    maskf = mask.astype(np.float32)
    ur = np.stack([xx, yy], axis=2).astype(np.float32)
    dev = [torch.from_numpy(ur.copy()).cuda(), torch.zeros(H, W, device="cuda"), torch.from_numpy(ur.copy()).cuda(), None,
           torch.from_numpy(maskf).cuda(), float(np.sqrt(np.float32(100.0))), float(np.sqrt(np.float32(0.01)))]
    s = api.ThalloSolver((W, H), energy)
    costs = []
    for i in range(numIter):
        alpha = np.float32(i + 1) / np.float32(numIter)
        dev[3] = torch.from_numpy(F.constraint_image(full, maskf, alpha)).cuda()
        final, cs = s.solve(dev, profiled=True, nIterations=nIt, lIterations=lIt)
        costs += cs
    s.close()
    assert np.allclose(csv_costs, np.array(costs), rtol=1e-6), (csv_costs, costs)
    assert abs(fc["costs"]["ThalloGN"] - final) <= 1e-6 * abs(final)
    off = np.fromfile(tmp_path / "warp_offset.f32", dtype=np.float32).reshape(H, W, 2)
    assert np.abs(off - dev[0].cpu().numpy()).max() <= 1e-4
    assert os.path.exists(tmp_path / "out_displacement.png")


@pytest.mark.gpu
def test_harness_shape_from_shading_on_the_reference_data(tmp_path):
    """The SFS application on the reference's default data set (quarter resolution fixture, written back as the four .imagedump files
    and the 160-byte parameter file it ships as): artefacts + final cost equal to the python driver's on the same inputs."""
    import torch
    import thallo_amd
    from thallo_amd import api
    assert torch.cuda.is_available(), "this test needs the MI355X"
    _build()
    d = np.load(os.path.join(GOLD, "sfs_default_q4.npz"))
    H, W = d["depth"].shape
    pre = str(tmp_path / "q4")
    depth_inf = d["depth"].copy(); depth_inf[depth_inf == -10000.0] = -np.inf          # as shipped: holes are -inf, clamped by the reader
    F.write_imagedump(pre + "_targetDepth.imagedump", depth_inf)
    F.write_imagedump(pre + "_targetIntensity.imagedump", d["intensity"])
    F.write_imagedump(pre + "_initialUnknown.imagedump", d["initial"])
    F.write_imagedump(pre + "_maskEdgeMap.imagedump", np.concatenate([d["edge_r"], d["edge_c"]], axis=0))
    sc = d["scalars"]
    F.write_sfs_params(pre + ".SFSSolverParameters", {"weightFitting": sc[0], "weightRegularizer": sc[1], "weightPrior": 0, "weightShading": sc[2],
                                                      "weightShadingStart": 1, "weightShadingIncrement": 0, "weightBoundary": 0, "fx": sc[3], "fy": sc[4],
                                                      "ux": sc[5], "uy": sc[6], "deltaTransform": np.eye(4), "lightingCoefficients": sc[7:16]})
    energy = thallo_amd.energy_file("shape_from_shading")
    r = subprocess.run([BIN_SFS, pre, "-o", energy, "-N", "4", "-L", "10", "--profile"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Num Active Unknowns: {int((d['depth'] > 0).sum())}" in r.stdout
    fc = json.load(open(tmp_path / "finalCosts.json"))
    assert fc["name"] == "Shape From Shading"
    p = [float(v) for v in sc] + [torch.from_numpy(d[k].copy()).cuda() for k in ("initial", "depth", "intensity", "edge_r", "edge_c")]
    s = api.ThalloSolver((W, H), energy)
    final, costs = s.solve(p, profiled=True, nIterations=4, lIterations=10)
    s.close()
    rows = open(tmp_path / "results" / "results_float.csv").read().strip().splitlines()
    assert len(rows) == 1 + 5 and np.allclose([float(rw.split(",")[2]) for rw in rows[1:]], costs, rtol=1e-6)
    assert abs(fc["costs"]["ThalloGN"] - final) <= 1e-6 * abs(final)
    out = F.read_imagedump(str(tmp_path / "sfsOutput.imagedump"), clamp_infinity=False)[:, :, 0]
    assert np.abs(out - p[16].cpu().numpy()).max() <= 1e-6


@pytest.mark.gpu
def test_harness_arap_on_the_reference_mesh(tmp_path):
    """The ARAP application on the reference's small_armadillo.ply + .mrk (faces split at their centroids so that the landmark indices
    exist), 2 continuation solves: final cost and deformed mesh equal to the python driver's on the same inputs (unconstrained
    vertices carry -infinity, as in the reference)."""
    import torch
    import thallo_amd
    from thallo_amd import api
    assert torch.cuda.is_available(), "this test needs the MI355X"
    _build()
    for n in ("small_armadillo.ply", "small_armadillo.mrk"):
        shutil.copy(os.path.join(GOLD, n), tmp_path / n)
    energy = thallo_amd.energy_file("arap_mesh_deformation")
    numIter, nIt, lIt = 2, 3, 20
    r = subprocess.run([BIN_ARAP, str(tmp_path / "small_armadillo.ply"), "-o", energy, "--split-faces", "-n", str(numIter), "-N", str(nIt), "-L", str(lIt), "--profile"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Faces: 768\nVertices: 386" in r.stdout
    fc = json.load(open(tmp_path / "finalCosts.json"))
    V, faces = F.read_ply(os.path.join(GOLD, "small_armadillo.ply"))
    idx, target = F.read_mrk(os.path.join(GOLD, "small_armadillo.mrk"))
    nv = len(V)
    cent = np.zeros((len(faces), 3), np.float32)
    for i, fc_ in enumerate(faces):                                   # the same float32 running sum as the C++ side
        c = np.zeros(3, np.float32)
        for v in fc_:
            c = (c + V[v] / np.float32(len(fc_))).astype(np.float32)
        cent[i] = c
    V2 = np.concatenate([V, cent]).astype(np.float32)
    faces2 = [[f_[k], f_[(k + 1) % 3], nv + i] for i, f_ in enumerate(faces) for k in range(3)]
    v0, v1 = F.mesh_directed_edges(faces2, len(V2))
    dev = [float(np.sqrt(np.float32(4.0))), float(np.sqrt(np.float32(1.0))), torch.from_numpy(V2.copy()).cuda(), torch.zeros(len(V2), 3, device="cuda"),
           torch.from_numpy(V2.copy()).cuda(), None, torch.from_numpy(v0).cuda(), torch.from_numpy(v1).cuda()]
    s = api.ThalloSolver((len(V2), len(v0)), energy)
    costs = []
    for i in range(numIter):
        a = np.float32(i + 1) / np.float32(numIter)
        cons = np.full((len(V2), 3), -np.inf, np.float32)
        cons[idx] = (np.float32(1) - a) * V2[idx] + a * target
        dev[5] = torch.from_numpy(cons).cuda()
        final, cs = s.solve(dev, profiled=True, nIterations=nIt, lIterations=lIt)
        costs += cs
    s.close()
    assert np.isfinite(costs).all()
    rows = open(tmp_path / "results" / "results_float.csv").read().strip().splitlines()
    assert len(rows) == 1 + numIter * (nIt + 1)
    assert np.allclose([float(rw.split(",")[2]) for rw in rows[1:]], costs, rtol=1e-5)
    assert abs(fc["costs"]["ThalloGN"] - final) <= 1e-5 * abs(final)
    Vout, fout = F.read_ply(str(tmp_path / "out.ply"))
    assert len(Vout) == 386 and np.abs(Vout - dev[2].cpu().numpy()).max() <= 1e-3 * np.abs(V2).max()
