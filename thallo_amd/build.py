"""Build helpers: compile the gfx950 kernels + C-ABI library in-tree (thallo_amd/libThallo.so)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libThallo.so")


def build_library(force=False, jobs=8, quiet=True):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the library path."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libThallo.so failed:\n" + res.stdout)
    if not quiet:
        print(res.stdout)
    if not os.path.exists(LIB):
        raise RuntimeError("make succeeded but %s is missing" % LIB)
    return LIB
