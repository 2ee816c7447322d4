"""bench.py --gpus N starts its N ranks itself when no launcher did; checked without a GPU through the dry-run
switch: the ranks rendezvous over gloo on 127.0.0.1, agree on the world size, rank 0 prints the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_spawns_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"THALLO_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_launcher_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "4"], {"THALLO_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4" in r.stderr and "WORLD_SIZE=2" in r.stderr
