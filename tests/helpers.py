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
