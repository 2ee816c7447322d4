"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the .t front-end
recognises the bundled energies (no GPU compute here)."""
import ctypes as C
import os
import re

import pytest

import thallo_amd
from thallo_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:Thallo_|ThalloX_|thallo_hip_)\w+)\s*\(", txt)))


@pytest.fixture(scope="module")
def L():
    from thallo_amd.build import build_library
    build_library()
    return thallo_amd.lib()


@pytest.mark.parametrize("header", ["Thallo.h", "thallo_hip.h"])
def test_exports_every_declared_symbol(L, header):
    names = _declared(header)
    assert len(names) >= 13
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_reference_api_surface_is_complete(L):
    # the 13 entry points of API/release/include/Thallo.h:41-105
    for n in ["Thallo_NewState", "Thallo_ProblemDefine", "Thallo_ProblemDelete", "Thallo_ProblemPlan", "Thallo_PlanFree",
              "Thallo_SetSolverParameter", "Thallo_GetSolverParameter", "Thallo_ProblemSolve", "Thallo_ProblemInit",
              "Thallo_ProblemStep", "Thallo_ProblemCurrentCost", "Thallo_GetPerformanceSummary"]:
        assert hasattr(L, n)
    assert C.sizeof(api.InitializationParameters) == 24
    assert C.sizeof(api.PerformanceEntry) == 40 and C.sizeof(api.PerformanceSummary) == 200


@pytest.mark.parametrize("fname,energy", [
    ("image_warping", "image_warping"), ("laplacian_image", "laplacian_image"),
    ("laplacian_image_shipped_guard", "laplacian_image"), ("laplacian_graph", "laplacian_graph"),
    ("arap_mesh_deformation", "arap_mesh"), ("bundle_adjustment", "bundle_adjustment"),
    ("shape_from_shading", "shape_from_shading")])
def test_frontend_recognises_bundled_energies(L, fname, energy):
    buf = C.create_string_buffer(64)
    h = L.ThalloX_ProblemFileHash(thallo_amd.energy_file(fname).encode(), buf, 64)
    assert h != 0 and buf.value.decode() == energy
    inc = open(os.path.join(ROOT, "thallo_amd", "csrc", "known_energy_hashes.inc")).read()
    assert f"0x{h:016x}" in inc, "run tools/gen_energy_hashes.py after editing a bundled .t"


def test_frontend_rejects_unknown_energy(L, tmp_path):
    f = tmp_path / "other.t"
    f.write_text('local N = Dims("N")\nInputs { X = Unknown(float,{N},0) }\nr = Residuals { only = X(N()) }\n')
    buf = C.create_string_buffer(64)
    L.ThalloX_ProblemFileHash(str(f).encode(), buf, 64)
    assert buf.value == b""


def test_frontend_ignores_comments_and_whitespace(L, tmp_path):
    src = open(thallo_amd.energy_file("laplacian_graph")).read()
    f = tmp_path / "g.t"
    f.write_text("--[[ block\ncomment ]]\n" + src.replace("\n", "   \n\t") + "\n-- trailing")
    b1, b2 = C.create_string_buffer(64), C.create_string_buffer(64)
    assert L.ThalloX_ProblemFileHash(str(f).encode(), b1, 64) == \
        L.ThalloX_ProblemFileHash(thallo_amd.energy_file("laplacian_graph").encode(), b2, 64)


def test_frontend_reads_the_materialize_schedule_lines(L, tmp_path):
    """r.<residual>.J:set_materialize(true) / .JtJ:set_materialize(true) (thallo.t:5661-5690; tests/minimal/laplacian.t:16-20)"""
    src = open(thallo_amd.energy_file("laplacian_image")).read()
    lines = {"fitJ": "r.fit.J:set_materialize(true)\n", "regJ": "r.reg.J:set_materialize(true)\n",
             "fitJtJ": "r.fit.JtJ:set_materialize(true)\n", "regJtJ": "r.reg.JtJ:set_materialize(true)\n"}
    cases = [("", 0), (lines["fitJ"] + lines["regJ"], 1), (lines["fitJ"], 0), ("".join(lines.values()), 2),
             ("-- " + lines["fitJ"] + "-- " + lines["regJ"], 0)]
    for extra, want in cases:
        f = tmp_path / "l.t"
        f.write_text(src + "\n" + extra)
        assert L.ThalloX_ProblemFileSchedule(str(f).encode()) == want, (extra, want)
    assert L.ThalloX_ProblemFileSchedule(str(tmp_path / "missing.t").encode()) == -1


def test_vector_padding_rule(L):
    L.thallo_hip_vector_elems.restype = C.c_long
    L.thallo_hip_vector_elems.argtypes = [C.c_long]
    assert L.thallo_hip_vector_elems(1) == 256 and L.thallo_hip_vector_elems(256) == 256 and L.thallo_hip_vector_elems(257) == 512


def test_marching_geometry_search_is_bounded(L):
    """rows-per-segment search of the marching kernels must terminate for every shape: with more column strips than workgroup slots
    (W > ~31.7k on 256 CUs, ~3.9k on a 32-CU partition) it answers 0 = 'stay on the tile kernel' instead of spinning.  Host logic only: the
    forced workgroup budget (debug knob 6) stands in for the device's CU count, so nothing here touches a GPU."""
    L.thallo_hip_iw_march_rows.restype = C.c_int
    L.thallo_hip_iw_march_rows.argtypes = [C.c_int, C.c_int]
    try:
        L.thallo_hip_march_debug_set(6, 256)
        assert L.thallo_hip_iw_march_rows(2048, 2048) == 35          # 17 strips x ceil(59 segments / 4 waves) = 255 workgroups <= 256
        assert L.thallo_hip_iw_march_rows(2048, 256) == 5            # 17 x ceil(52 / 4) = 221
        assert L.thallo_hip_iw_march_rows(124 * 256, 64) > 0         # exactly 256 strips: one segment row
        assert L.thallo_hip_iw_march_rows(124 * 256 + 2, 64) == 0    # 257 strips: no R fits -> 0, not an endless loop
        assert L.thallo_hip_iw_march_rows(65536, 4096) == 0
        L.thallo_hip_march_debug_set(6, 32)                           # a 32-CU partition
        assert L.thallo_hip_iw_march_rows(3968, 512) > 0 and L.thallo_hip_iw_march_rows(4096, 512) == 0
        L.thallo_hip_march_debug_set(6, 3)                            # fewer than 8 workgroup slots: the budget is clamped to 8, not rounded down to 0
        assert L.thallo_hip_iw_march_rows(512, 512) > 0 and L.thallo_hip_iw_march_rows(2048, 512) == 0
        assert L.thallo_hip_iw_march_rows(511, 512) == 0 and L.thallo_hip_iw_march_rows(512, 0) == 0      # odd width / no rows: not a marching shape
    finally:
        L.thallo_hip_march_debug_set(6, 0)
