"""Shared helpers for the parity tests (GPU side)."""
import numpy as np


def to_device(params):
    """numpy arrays -> cuda tensors (device pointers); floats stay host scalars (Param)."""
    import torch
    out = []
    for p in params:
        if isinstance(p, np.ndarray):
            out.append(torch.from_numpy(p.copy()).cuda())
        else:
            out.append(float(p))
    return out


def to_host(t):
    return t.detach().cpu().numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def copy_params(params):
    return [p.copy() if isinstance(p, np.ndarray) else p for p in params]


def oracle_fixture(name, params):
    """The committed oracle trajectory `name` (tests/golden/oracle_trajectories.json, written by tests/golden/make_oracle_trajectories.py) if THIS machine's
    instance reproduces the inputs it was computed from (sha256 over the input arrays), else None: the caller then runs the oracle live, as the suite did
    before round 4 (the full-size oracle solves took 40 % of the GPU suite's time on the GPU box's host cores)."""
    import hashlib
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_trajectories.json")
    if not os.path.exists(path):
        return None
    fx = json.load(open(path)).get(name)
    if fx is None:
        return None
    h = hashlib.sha256()
    for a in params:
        if isinstance(a, np.ndarray):
            h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(np.ascontiguousarray(a).tobytes())
        else:
            h.update(np.float64(a).tobytes())
    return fx if h.hexdigest()[:32] == fx["input_checksum"] else None


def set_ab(monkeypatch, **kw):
    """THALLO_AB=key=value,...: the library's A/B alternatives (csrc/solver.cpp env_switch).  set_ab(monkeypatch, one_kernel="0", fin_in_kernel=None) sets / removes keys
    and leaves the others as they are."""
    import os
    cur = dict(tok.split("=", 1) for tok in os.environ.get("THALLO_AB", "").split(",") if "=" in tok)
    for k, v in kw.items():
        if v is None: cur.pop(k, None)
        else: cur[k] = str(v)
    if cur: monkeypatch.setenv("THALLO_AB", ",".join(f"{k}={v}" for k, v in cur.items()))
    else: monkeypatch.delenv("THALLO_AB", raising=False)
