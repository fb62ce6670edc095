"""One cached Engine per (device, H, W) for the module-level drop-ins (stn.inverse_warp2, helpers.compute_photometric_error,
train_mono.solve_pose_iteratively, ...), which -- like the reference functions they mirror -- take no handle argument."""
from __future__ import annotations

import torch

from .engine import Engine

_engines = {}


def get_engine(H: int, W: int, n_pairs: int) -> Engine:
    key = (torch.cuda.current_device(), int(H), int(W))
    e = _engines.get(key)
    if e is None or e.max_pairs < n_pairs:
        e = Engine(H, W, max(int(n_pairs), 2 * (e.max_pairs if e else 0)))
        _engines[key] = e
    return e
