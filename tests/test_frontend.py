"""The mini front-end for .t problem specifications (thallo_amd/csrc/dsl*.{hpp,cpp}) without a GPU: it executes the
bundled energy files (a Lua subset + the DSL's constructors and lib.t helpers as builtins), reports their declarations, generates one HIP
translation unit per file -- and that unit compiles for gfx950."""
import ctypes as C
import os
import subprocess

import pytest

import thallo_amd
from thallo_amd import api

ENERGIES = ["laplacian_image", "laplacian_graph", "image_warping", "arap_mesh_deformation", "bundle_adjustment", "shape_from_shading"]


def _text(path, what, dims=None, expect_error=False):
    L = api.lib()
    L.ThalloX_FrontendTextDims.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]; L.ThalloX_FrontendTextDims.restype = C.c_int
    buf = C.create_string_buffer(1 << 21)
    d = (C.c_uint * len(dims))(*dims) if dims else None
    n = L.ThalloX_FrontendTextDims(path.encode(), what, d, buf, len(buf))
    if expect_error:
        assert n < 0
        return api.last_error()
    assert n >= 0, api.last_error()
    assert n < len(buf)
    return buf.value.decode()


def test_declarations_of_the_bundled_energies():
    d = _text(thallo_amd.energy_file("image_warping"), 0)
    assert "dims: W H" in d and "unknown Offset slot 0 channels 2 over W H (Exclude)" in d and "unknown Angle slot 1 channels 1 over W H (Exclude)" in d
    assert "param w_fitSqrt slot 5" in d and "preconditioner 1" in d
    for r in ("reg_px", "reg_nx", "reg_py", "reg_ny", "fit"):
        assert f"residual {r} x2 over W H" in d
    d = _text(thallo_amd.energy_file("arap_mesh_deformation"), 0)
    assert "sparse V0 slot 6 E -> N" in d and "residual fit x3 over N" in d and "residual reg x3 over E" in d
    d = _text(thallo_amd.energy_file("bundle_adjustment"), 0)
    assert "unknown cameras slot 0 channels 9 over C" in d and "residual snavely_reprojection_error x2 over O" in d
    d = _text(thallo_amd.energy_file("shape_from_shading"), 0)
    assert "array edgeMaskR slot 19 channels 1 uint8 over W H" in d and "residual reg x3 over W H" in d and "preconditioner 0" in d
    d = _text(thallo_amd.energy_file("laplacian_image"), 0)
    assert "residual fit x1 over W H" in d and "residual reg x2 over W H" in d


def test_schedule_lines_are_recorded(tmp_path):
    """r.<name>.J / JtJ / Jp :set_materialize(true) (the surface tests/minimal/laplacian.t:16-20 and tests/minimal_graph/laplacian.t:19-20 use)"""
    src = open(thallo_amd.energy_file("laplacian_graph")).read()
    f = tmp_path / "sched.t"
    f.write_text(src + "\nr.fit.J:set_materialize(true)\nr.fit.JtJ:set_materialize(true)\nr.reg.Jp:set_materialize(true)\n")
    d = _text(str(f), 0)
    assert "residual fit x1 over N J JtJ" in d and "residual reg x1 over E Jp" in d


def test_unsupported_constructs_are_errors_not_guesses(tmp_path):
    L = api.lib()
    L.ThalloX_FrontendText.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]; L.ThalloX_FrontendText.restype = C.c_int
    buf = C.create_string_buffer(4096)
    for name, body, needle in (
            ("undefined.t", 'local W = Dims("W")\nInputs { X = Unknown(float,{W},0) }\nlocal x = W()\nr = Residuals { a = Frobnicate(X(x)) }\n', "Frobnicate"),
            ("syntax.t", 'local W = Dims("W"\n', "expected"),
            ("nounknown.t", 'local W = Dims("W")\nInputs { A = Array(float,{W},0) }\nlocal x = W()\nr = Residuals { a = A(x) }\n', "no Unknown"),
            ("fourd.t", 'local W,H,D,T = Dims("W","H","D","T")\nInputs { X = Unknown(float,{W,H,D,T},0) }\n', "3-dimensional")):
        f = tmp_path / name
        f.write_text(body)
        assert L.ThalloX_FrontendText(str(f).encode(), 0, buf, len(buf)) == -1
        assert needle in api.last_error(), api.last_error()


@pytest.mark.parametrize("energy", ENERGIES)
def test_generated_kernels_compile_for_gfx950(energy, tmp_path):
    src = _text(thallo_amd.energy_file(energy), 1)
    assert 'extern "C" __global__' in src and "jtj_0" in src and "applyjt_0" in src
    f = tmp_path / (energy + ".hip")
    f.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(f), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_wave_aggregated_scatter_is_chosen_per_access(tmp_path):
    """An unknown access whose index does not vary with the innermost iteration dimension scatters through the wave-aggregated add (the reference's
    get_peers / reduce_peers case, thallo.t:3349-3399); every other access keeps the plain atomic.  The unit compiles for gfx950."""
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "energies", "row_gain.t")
    src = _text(f, 1)
    fit = src[src.index("void jtj_0"):src.index("void applyj_0")]
    assert "scatter_add<true>(Ap" in fit and "scatter_add<false>" not in fit
    reg = src[src.index("void jtj_1"):src.index("void applyj_1")]          # residual over H alone: nothing to aggregate
    assert "scatter_add<true>" not in reg and "scatter_add<false>(Ap" in reg
    iw = _text(thallo_amd.energy_file("image_warping"), 1)                 # every access involves x
    assert "scatter_add<true>" not in iw[iw.index("void jtj_0"):]
    out = tmp_path / "row_gain.hip"
    out.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(out), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_unit_fingerprint_separates_look_alikes(tmp_path):
    """ThalloX_ProblemFileUnitHash: schedule lines, comments and residual names do not change it; a different weight or a different expression does.
    (A file with a bundled energy's declarations runs on that energy's hand-written plugin only if its fingerprint is the bundled file's.)"""
    L = thallo_amd.lib()
    L.ThalloX_ProblemFileUnitHash.restype = C.c_ulonglong
    L.ThalloX_ProblemFileUnitHash.argtypes = [C.c_char_p]
    fp = lambda path: L.ThalloX_ProblemFileUnitHash(str(path).encode())
    src = open(thallo_amd.energy_file("laplacian_graph")).read()
    base = fp(thallo_amd.energy_file("laplacian_graph"))
    assert base != 0
    a = tmp_path / "a.t"; a.write_text("-- a comment\n" + src + "\nr.fit.J:set_materialize(true)\nr.reg.J:set_materialize(true)\n")
    assert fp(a) == base
    b = tmp_path / "b.t"; b.write_text(src.replace("w_fit = 0.5", "w_fit = 0.25"))
    assert fp(b) not in (0, base)
    c = tmp_path / "c.t"; c.write_text(src.replace("X(v0(e)) - X(v1(e))", "sin(X(v0(e))) - sin(X(v1(e)))"))
    assert fp(c) not in (0, base, fp(b))
    d = tmp_path / "d.t"; d.write_text(src.replace("reg =", "smooth =").replace("r.reg", "r.smooth"))
    assert fp(d) == base
    assert fp(os.path.join(os.path.dirname(os.path.abspath(__file__)), "energies", "graph_get.t")) not in (0, base)
    assert fp(tmp_path / "missing.t") == 0


def test_the_compiled_in_unit_fingerprints_are_current():
    """known_energy_hashes.inc (tools/gen_energy_hashes.py) holds the fingerprint of every bundled energy as THIS build of the front-end computes it --
    a stale table silently sends a known energy with extra schedule lines to the generated kernels instead of its hand-written plugin."""
    import glob
    import re
    L = thallo_amd.lib()
    L.ThalloX_ProblemFileUnitHash.restype = C.c_ulonglong
    L.ThalloX_ProblemFileUnitHash.argtypes = [C.c_char_p]
    inc = open(os.path.join(os.path.dirname(os.path.abspath(thallo_amd.__file__)), "csrc", "known_energy_hashes.inc")).read()
    table = inc[inc.index("KNOWN_UNIT_HASHES"):]
    known = {int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ULL", table)}
    files = sorted(glob.glob(os.path.join(os.path.dirname(thallo_amd.energy_file("image_warping")), "*.t")))
    assert len(files) >= 6
    for f in files:
        assert L.ThalloX_ProblemFileUnitHash(f.encode()) in known, f


REF_PAIRS = [("examples/image_warping/image_warping.t", "image_warping"), ("examples/arap_mesh_deformation/arap_mesh_deformation.t", "arap_mesh_deformation"),
             ("examples/shape_from_shading/shape_from_shading.t", "shape_from_shading"), ("examples/bundle_adjustment/bundle_adjustment.t", "bundle_adjustment"),
             ("tests/minimal_graph/laplacian.t", "laplacian_graph"), ("tests/minimal/laplacian.t", "laplacian_image_shipped_guard")]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("rel,mine", REF_PAIRS)
def test_bundled_energies_are_the_references_energies(rel, mine):
    """The bundled .t files are re-written specifications (different text).  Run through the front-end, each of them and the reference's own file (read
    in place, never copied) produce the SAME translation unit -- the same expression DAG per residual component, the same unknown accesses, the same
    guards -- up to the residuals' names.  So everything the GPU tests establish about a bundled energy (oracle trajectories, gold PNGs, generated
    vs hand-written plugins) is established about the energy the reference's file states, not about a restatement of it."""
    import re
    norm = lambda t: re.sub(r"// ---- residual \w+", "// ---- residual R", t)
    a, b = norm(_text(os.path.join("/root/reference", rel), 1)), norm(_text(thallo_amd.energy_file(mine), 1))
    assert a == b
    da, db = _text(os.path.join("/root/reference", rel), 0), _text(thallo_amd.energy_file(mine), 0)
    strip = lambda d: [re.sub("( J| JtJ| Jp)+$", "", re.sub("^residual [^ ]+", "residual", ln)) for ln in d.splitlines()]
    assert strip(da) == strip(db)


REF = "/root/reference"
REF_EXAMPLES = ["examples/image_warping/image_warping.t", "examples/arap_mesh_deformation/arap_mesh_deformation.t",
                "examples/shape_from_shading/shape_from_shading.t", "examples/bundle_adjustment/bundle_adjustment.t",
                "examples/cotangent_mesh_smoothing/cotangent_mesh_smoothing.t", "examples/poisson_image_editing/poisson_image_editing.t",
                "examples/procrustes_alignment/procrustes_alignment.t", "examples/shape_and_shading/shape_and_shading.t",
                "examples/volumetric_mesh_deformation/volumetric_mesh_deformation.t", "examples/embedded_mesh_deformation/embedded_mesh_deformation.t",
                "examples/intrinsic_image_decomposition/intrinsic_image_decomposition.t", "examples/robust_nonrigid_alignment/robust_nonrigid_alignment.t",
                "examples/sparse_bundle_fusion/bundle_fusion_solve.t", "examples/bundle_fusion_solve/bundle_fusion_solve.t", "examples/optical_flow/optical_flow.t",
                "tests/minimal_sparse_materialize/minimal_sparse_materialize.t", "tests/expansive_sparse_materialize/expansive_sparse_materialize.t",
                "tests/minimal/laplacian.t", "tests/minimal_graph/laplacian.t", "tests/minimal_exclude/minimal_exclude.t",
                "tests/minimal_materialize/minimal_materialize.t", "tests/multidomain/multidomain.t", "tests/dense/curveFitting.t",
                "tests/energy_unit_tests/laplacian.t", "tests/create_delete_cycle/laplacian.t", "tests/minimal_2d_graph/laplacian.t", "tests/dense/curveFitting_dense.t"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("rel", REF_EXAMPLES)
def test_the_references_own_files_go_through_the_front_end(rel):
    """14 of the reference's 17 example energies and 12 of its test energies, as shipped (read in place, never copied): the front-end executes
    them and emits their kernels.  (Sum and index arithmetic between two iteration variables: test_sum_is_expanded_at_plan_time.  Not yet: SampledImageArray.)"""
    src = _text(os.path.join(REF, rel), 1)
    assert "cost_0" in src and "jtf_0" in src and "jtj_0" in src


def test_gather_lowering_exists_exactly_where_it_can(tmp_path):
    """compute_at_output (thallo.t:5661-5674, bodies createjtjcentered / createjtfcentered :3603-3712): residuals whose dims are the dims of every unknown they read
    through constant-offset accesses get unknown-wise kernels jtjg_ / jtfg_ (one thread per unknown pixel, no atomics); graph residuals (Sparse maps) and residuals
    over other dims than their unknowns' do not.  The schedule call is recorded per residual, and the unit with the gather kernels compiles for gfx950."""
    iw = _text(thallo_amd.energy_file("image_warping"), 1)
    assert iw.count("void jtjg_") == iw.count("void jtj_") and "atomicAdd" not in iw[iw.index("void jtjg_0"):iw.index("void jtfg_0")]
    sfs = _text(thallo_amd.energy_file("shape_from_shading"), 1)
    assert "void jtjg_0" in sfs
    for graph in ("laplacian_graph", "arap_mesh_deformation", "bundle_adjustment"):
        src = _text(thallo_amd.energy_file(graph), 1)
        # (ARAP's fit residual is over the vertices alone: gather; its edge residual reads through V0 / V1: scatter)
        assert "void jtjg_" + str(src.count("void jtj_") - 1) not in src, graph
    rg = _text(os.path.join(os.path.dirname(os.path.abspath(__file__)), "energies", "row_gain.t"), 1)
    assert "void jtjg_0" not in rg                     # residual over (x, y) reads the unknown G(y): other dims than its own -> no gather form
    f = tmp_path / "lap_at_output.t"
    f.write_text(open(thallo_amd.energy_file("laplacian_image")).read() + "\nr.reg:compute_at_output(true)\nr.fit:compute_at_output(false)\n")
    assert "void jtjg_0" in _text(str(f), 1)           # (the schedule call parses; which kernels RUN is decided per residual at Plan time: test_gpu_frontend.py)
    out = tmp_path / "iw_gather.hip"
    out.write_text(iw)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(out), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


HERE = os.path.dirname(os.path.abspath(__file__))


def test_sum_is_expanded_at_plan_time(tmp_path):
    """Sum({k, ...}, e) (lib.t:146 = P:TensorContraction, thallo.t:5887-5922) and index arithmetic between two iteration variables (R(n - k + 2)).  The front-end
    expands the sum when the problem is planned -- the dimensions are known then -- into one term per value of the summed variables: W(m) becomes the constant
    accesses W(0) .. W(M-1), R(n - k + 2) becomes R(n + 2), R(n + 1), ...; the residual's domain is what is left (N).  Without dimensions the file is refused with
    a message that says why.  The unit compiles for gfx950."""
    f = os.path.join(HERE, "energies", "series_fit.t")
    d = _text(f, 0, dims=(512, 16))
    assert "unknown Weights slot 0 channels 1 over M" in d and "array Basis slot 1 channels 1 over N M" in d and "residual fit x1 over N Jp" in d
    src = _text(f, 1, dims=(512, 16))
    assert "16 unknown access(es)" in src and "dd[15] = " in src and "dd[16] = " not in src          # 16 partials, every one structurally non-zero
    assert "16 unknown access(es)" not in _text(f, 1, dims=(512, 5)) and "5 unknown access(es)" in _text(f, 1, dims=(512, 5))
    assert "Sum needs the sizes" in _text(f, 1, expect_error=True)
    c = _text(os.path.join(HERE, "energies", "conv1d.t"), 1, dims=(512, 5))
    assert "5 unknown access(es)" in c and "residual conv" in c
    for off in ("i0 + (2)", "i0 + (1)", "i0 + (0)", "i0 + (-1)", "i0 + (-2)"):      # Signal(n - k + 2), k = 0 .. 4
        assert off in c, off
    assert "Sum over 262144 terms" in _text(f, 1, dims=(512, 262144), expect_error=True)
    out = tmp_path / "series_fit.hip"
    out.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(out), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("rel,dims,accesses", [("tests/minimal_fitting/minimal_fitting.t", (512, 16), 16), ("tests/convolution/convolution.t", (512, 5), 5),
                                               ("examples/face_fitting/face_fitting.t", (1000, 20, 1), 20),
                                               ("examples/spatially_varying_deconvolution/spatially_varying_deconvolution.t", (64, 64, 5, 4), 25)])
def test_the_references_sum_files_go_through_the_front_end(rel, dims, accesses):
    """the reference's files that use Sum, as shipped (read in place): its two tests at the sizes their main.cpp uses, face_fitting (Sum over the blendshape weights,
    image(n, channel) accesses) and spatially_varying_deconvolution with a 5 x 5 kernel (two iteration variables over ONE dimension -- k_0 = Kd(); k_1 = Kd() -- and a 3-D
    array indexed through a Sparse map)."""
    src = _text(os.path.join(REF, rel), 1, dims=dims)
    assert "cost_0" in src and f"{accesses} unknown access(es)" in src


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_a_sum_too_wide_for_forward_mode_duals_takes_the_wide_lowering(tmp_path):
    """spatially_varying_deconvolution at its shipped 17 x 17 kernel reads 289 unknown elements per instance of its convolution residual: more than forward-mode duals carry at
    once, so the residual gets the wide lowering (16 partials per evaluation at this size, chunk after chunk; round 2 refused the file) -- and the unit compiles for gfx950."""
    src = _text(os.path.join(REF, "examples/spatially_varying_deconvolution/spatially_varying_deconvolution.t"), 1, dims=(64, 64, 17, 4))
    assert "289 unknown access(es)" in src and "Dual<16> rr[" in src and "for (int cb = 0; cb < 304; cb += 16)" in src
    out = tmp_path / "wide.hip"
    out.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(out), "-o", str(tmp_path / "wide.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_bundle_fusion_solve_goes_through_the_front_end(tmp_path):
    """examples/bundle_fusion_solve (the last of the reference's 17 example energies; refused until round 4), as shipped and read in place: its dense residual reads the six
    pose unknowns of the SOURCE frame only (the target frame's inverse comes from the Const arrays) -- M(t0, t1):get(t_target(p), t_source(p)) binds the two variables over T
    positionally --, its sparse residual the twelve of both frames; the frame index goes to the SampledImageArrays as a map entry's value; Sparse maps declared over CorrDim
    are read with the PairDim variable, as the reference reads them.  The unit compiles for gfx950.  (Numerics of these constructs: tests/energies/pair_reprojection.t in
    tests/test_gpu_frontend.py.)"""
    rel = os.path.join(REF, "examples/bundle_fusion_solve/bundle_fusion_solve.t")
    d = _text(rel, 0)
    assert "residual dense x1 over PairDim W H JtJ" in d and "residual sparse x3 over CorrDim JtJ" in d, d
    src = _text(rel, 1)
    assert "residual dense: 1 component(s), 6 unknown access(es)" in src and "residual sparse: 3 component(s), 12 unknown access(es)" in src
    body = src[src.index("// ---- residual dense"):src.index("// ---- residual sparse")]
    assert "(float)((const int*)c.in[" in body and "__builtin_huge_valf()" in body        # t_t:asvalue() through the map; neq(nrm(0), -inf)
    out = tmp_path / "bfs.hip"
    out.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-munsafe-fp-atomics", "-c", str(out), "-o", str(tmp_path / "bfs.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sampled_image_array_generates():
    """SampledImageArray (lib.t:145, thallo.t:5887-5922): a 3-D Array sampled at (x, y) in layer z, value only (the reference's partials are 0.0 too)"""
    src = _text(os.path.join(HERE, "energies", "layer_sample.t"), 1)
    assert "f_sample((const float*)c.in[1] + (long)c.dim[0] * c.dim[1] * 2 * min(max((int)val(" in src and "1 unknown access(es)" in src


def test_sampled_image_array_needs_three_dimensions(tmp_path):
    f = tmp_path / "bad.t"
    f.write_text('local W, H = Dims("W", "H")\nInputs { U = Unknown(float, {W, H}, 0), A = Array(float, {W, H}, 1) }\nlocal S = SampledImageArray(A)\nResiduals { r = U(W(), H()) }\n')
    assert "sampled image arrays must be 3D" in _text(str(f), 1, expect_error=True)
