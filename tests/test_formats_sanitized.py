"""The harness's file readers (harness/data_formats.hpp, harness/image_io.hpp) under AddressSanitizer + UBSan on the CPU (tools/formats_fuzz.cpp): the committed
fixtures and small files of the other formats as they are, then a few hundred seeded mutants of each.  A reader returns or throws std::runtime_error; a sanitizer
report, std::bad_alloc / std::length_error from a count the file cannot back, or a run-away loop fails the test.  (Before round 3 every one of the seven readers
failed this: chunk lengths past the end of a PNG, PLY properties without an element, face indices outside the vertex list, counts sized before they were checked.)"""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_harness_readers_survive_mutated_files_under_sanitizers(tmp_path):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "formats_fuzz")
    cc = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", os.path.join(ROOT, "tools", "formats_fuzz.cpp"), "-lz", "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and ("sanitize" in cc.stderr or "asan" in cc.stderr or "ubsan" in cc.stderr or "-lz" in cc.stderr or "zlib" in cc.stderr):
        pytest.skip("no sanitizer runtime / zlib for this g++: " + cc.stderr[-300:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    w, h = 7, 5
    (tmp_path / "a.imagedump").write_bytes(struct.pack("<4i", w, h, 1, 0) + np.arange(w * h, dtype=np.float32).tobytes())
    (tmp_path / "b.imagedump").write_bytes(struct.pack("<4i", w, h, 3, 1) + bytes(range(w * h * 3)))
    (tmp_path / "p.sfs").write_bytes(np.arange(40, dtype=np.float32).tobytes())
    (tmp_path / "m.off").write_text("OFF\n4 2 0\n0 0 0\n1 0 0\n0 1 0\n1 1 0\n3 0 1 2\n3 1 3 2\n")
    seeds = ["png:" + os.path.join(GOLD, "cat512_mask.png"), "constraints:" + os.path.join(GOLD, "cat512.constraints"), "ply:" + os.path.join(GOLD, "small_armadillo.ply"),
             "mrk:" + os.path.join(GOLD, "small_armadillo.mrk"), "imagedump:" + str(tmp_path / "a.imagedump"), "imagedump:" + str(tmp_path / "b.imagedump"),
             "sfsparams:" + str(tmp_path / "p.sfs"), "off:" + str(tmp_path / "m.off")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, "400", *seeds], capture_output=True, text=True, env=env, timeout=600)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail and "LeakSanitizer" not in tail, tail
    assert run.stdout.strip().splitlines()[-1].startswith("files 8 (8 read as they are)"), tail
