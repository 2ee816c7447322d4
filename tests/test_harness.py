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
