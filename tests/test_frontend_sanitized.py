"""The mini front-end under AddressSanitizer + UBSan on the CPU (tools/frontend_fuzz.cpp): every bundled and test energy file as it is and a few hundred seeded
mutants of each (truncations, flipped bytes, spliced text, huge numbers, deep nesting) -- each must lower or be refused with a message; any sanitizer report
(out-of-bounds, use-after-free, signed overflow, a leak -- a Plan of a file with a `local function` used to leak its closure / environment cycle) fails the test.
An energy file is input from outside the library, so the interpreter has to survive anything."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_front_end_survives_mutated_energy_files_under_sanitizers(tmp_path):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "frontend_fuzz")
    src = [os.path.join(ROOT, "tools", "frontend_fuzz.cpp"), os.path.join(ROOT, "thallo_amd", "csrc", "dsl_lua.cpp"), os.path.join(ROOT, "thallo_amd", "csrc", "dsl_codegen.cpp")]
    flags = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    if os.path.isdir("/opt/rocm/include"):      # + the recogniser of the bundled energies (csrc/frontend.cpp: its own lexer, declaration parser, hashes); host-only, no device
        flags += ["-DWITH_RECOGNISER", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include")]
        src.append(os.path.join(ROOT, "thallo_amd", "csrc", "frontend.cpp"))
    cc = subprocess.run([*flags, *src, "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and ("sanitize" in cc.stderr or "asan" in cc.stderr or "ubsan" in cc.stderr):
        pytest.skip("this g++ has no sanitizer runtime: " + cc.stderr[-300:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    files = sorted(glob.glob(os.path.join(ROOT, "thallo_amd", "energies", "*.t")) + glob.glob(os.path.join(ROOT, "tests", "energies", "*.t")))
    assert len(files) >= 20
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, "150", *files], capture_output=True, text=True, env=env, timeout=600)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail and "LeakSanitizer" not in tail, tail
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith(f"files {len(files)} ({len(files)} lower as they are)"), last
