"""Drop-ins for optimization_experiments/helpers.py (same names, arguments and result keys)."""
from __future__ import annotations

import numpy as np
import torch

from ._shared import get_engine
from .learning_helpers import disp_to_depth
from .optimizer import avg_final_predictions, batch_post_process_disparity  # noqa: F401   (helpers.py:25-33)


def compute_photometric_error(target_img, source_img, target_depth, source_depth, pose, intrinsics):
    """optimization_experiments/helpers.py:8-23 -> {'diff_img','img_rec','valid_mask' (validity x auto-mask),'weight_mask','poses'}"""
    N, _, H, W = target_img.shape
    r = get_engine(H, W, N).compute_photometric_error(target_img.float(), source_img.float(), target_depth.float(), source_depth.float(),
                                                      pose.float(), intrinsics.float())
    return {k: r[k] for k in ("diff_img", "img_rec", "valid_mask", "weight_mask", "poses")}


def get_disp_for_eigen(depth_model, target_img, config):
    """optimization_experiments/helpers.py:35-49: flip-averaged, post-processed disparity of the target frames"""
    with torch.no_grad():
        both = torch.cat((target_img, torch.flip(target_img, [3])), 0)
        disparities, _ = depth_model(both, epoch=50)
        disps, _ = disp_to_depth(disparities[0].float().contiguous(), config["min_depth"], config["max_depth"])
        pred = disps.cpu().detach()[:, 0].numpy()
        n = pred.shape[0] // 2
        return batch_post_process_disparity(pred[:n], pred[n:, :, ::-1])
