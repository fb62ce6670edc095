"""Drop-in for optimization_experiments/plot_loss_surface.py:11-87 (`generate_loss_surface`): the reconstruction error of
one frame pair as its forward translation (tz) and yaw are swept around a pose -- 50 samples each, which the reference
evaluates with 100 sequential `compute_photometric_error` calls and this module with one `tcsfm_loss_surface` launch per sweep.

Same arguments, same result keys (`delta_list`, `reconstruction_errors`, `best_trans_delta`, `best_pose_vec`, `best_error`,
`original_error` and the `_yaw` twins).  `plotting` / `results_dir` are accepted and ignored: drawing is the caller's business.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine

_engines = {}


def _engine(H, W, P):
    key = (torch.cuda.current_device(), H, W)
    if key not in _engines or _engines[key].max_pairs < P:
        _engines[key] = Engine(H, W, P)
    return _engines[key]


def generate_loss_surface(data, depths, pose_vec, sample_trans=False, sample_yaw=False, plotting=False, results_dir=None):
    target_img, source_img_list = data[0], data[1]
    intrinsics = data[5]
    tgt, src = target_img[0:1].float().contiguous(), source_img_list[0][0:1].float().contiguous()
    d_t, d_s = depths[0][0:1].float().contiguous(), depths[1][0:1].float().contiguous()
    K = intrinsics[0:1].float().contiguous()
    H, W = tgt.shape[2:]
    pose_vec = pose_vec.float()
    base = pose_vec[0:1, :6].contiguous()
    results = {}
    sweeps = []
    if sample_trans:
        tz = pose_vec[0, 2].abs().cpu()
        delta_list = torch.arange(-3 * tz, 3 * tz, 6 * tz / 50.)                 # plot_loss_surface.py:21
        results["delta_list"] = delta_list.numpy()
        sweeps.append((delta_list, 2, ""))
    if sample_yaw:
        delta_list_yaw = torch.arange(-0.02, 0.02, 0.04 / 50)                    # :26
        results["delta_list_yaw"] = delta_list_yaw.numpy()
        sweeps.append((delta_list_yaw, 4, "_yaw"))
    eng = _engine(H, W, 64)
    original_error = float(eng.loss_surface(tgt, src, d_t, d_s, K, base)[0])      # :30-33
    results["original_error"] = original_error
    for deltas, idx, suffix in sweeps:
        poses = base.repeat(len(deltas), 1)
        poses[:, idx] += deltas.to(poses.device, poses.dtype)
        errs = eng.loss_surface(tgt, src, d_t, d_s, K, poses.contiguous())
        # the running "best" of the reference loop (:49-54): strictly smaller than everything before it, original included
        best_error, best_delta, best_pose = original_error, torch.FloatTensor([0]), pose_vec.clone()
        for k in range(len(deltas)):
            if errs[k] < best_error:
                best_error, best_delta = float(errs[k]), float(deltas[k])
                best_pose = pose_vec.clone(); best_pose[0, idx] = poses[k, idx]
        results["reconstruction_errors" + suffix] = errs.astype(np.float32)
        results["best_trans_delta" if suffix == "" else "best_yaw_delta"] = best_delta
        results["best_pose_vec" + suffix] = best_pose.cpu().numpy()
        results["best_error" + suffix] = best_error
    return results
