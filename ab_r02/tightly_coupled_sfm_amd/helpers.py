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


@torch.no_grad()
def get_disp_for_eigen(depth_model, target_img, config):
    """optimization_experiments/helpers.py:35-49: disparity of the target frames for depth evaluation -- the network is run on
    the frames and on their mirror images, both go through disp_to_depth (HIP) and the two halves are blended with the
    Monodepth border ramp (batch_post_process_disparity)."""
    n_frames = target_img.shape[0]
    mirrored = torch.flip(target_img, dims=[3])
    net_out, _ = depth_model(torch.cat([target_img, mirrored], dim=0), epoch=50)
    scaled, _ = disp_to_depth(net_out[0].float().contiguous(), config["min_depth"], config["max_depth"])
    scaled = scaled[:, 0].cpu().numpy()
    straight, flipped_back = scaled[:n_frames], scaled[n_frames:, :, ::-1]
    return batch_post_process_disparity(straight, flipped_back)
