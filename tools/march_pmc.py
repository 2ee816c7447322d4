"""A few launches of the marching iteration kernel (mode 2 only / mode 4 only) and of the streaming reference, for rocprofv3 --pmc passes
(tools/march_pmc.sh).  Sweep build only."""
import os, sys
sys.argv = [sys.argv[0]]
os.environ.setdefault("MB_REPS", "3")
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "march_probe.py"), run_name="pmc")
