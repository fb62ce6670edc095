"""Host-side mirrors of the reference's scalar loss assembly (global reductions over maps the HIP library produced;
SURVEY.md section 8a rows a9, a10).  The Gauss-Newton engine does not call these -- it minimises the per-pair cost
directly -- they exist so that callers who log / compare the reference's `compute_optimization_loss` can keep doing so,
and they are checked against the reference's own numbers (tests/golden/golden_batch24x40.npz, G5)."""
from __future__ import annotations

import torch


class SSIM_Loss:
    """losses.py:16-41: callable like the reference's nn.Module -- SSIM_Loss()(x, y) -> clamp((1 - SSIM) / 2, 0, 1) per pixel
    and channel, 3x3 windows over a reflect-padded image; one fused HIP kernel (tcsfm_ssim) instead of 2 pads + 5 pools."""

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        from ._shared import get_engine
        H, W = x.shape[-2:]
        return get_engine(H, W, max(1, x.shape[0])).ssim_loss(x.float().contiguous(), y.float().contiguous())

    forward = __call__


def get_smooth_loss(disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """losses.py:43-61: edge-aware smoothness of the mean-normalised disparity.  GPU tensors of the engine's image size go
    through the HIP kernels (tcsfm_smooth_loss); anything else (CPU tensors: the golden tests of the scalar-loss mirror) through
    the same expression in torch."""
    if disp.is_cuda and disp.dim() == 4 and disp.shape[1] == 1 and img.shape[1] == 3:
        from ._shared import get_engine
        H, W = disp.shape[-2:]
        v = get_engine(H, W, max(1, disp.shape[0])).smooth_loss(disp.float().contiguous(), img.float().contiguous())
        return torch.tensor(v, device=disp.device, dtype=disp.dtype)
    mean_disp = disp.mean(2, True).mean(3, True)
    disp = disp / (mean_disp + 1e-7)
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def compute_optimization_loss(options: dict, target_img, target_disparity, target_disparity_init, fwd_data: dict, inv_data: dict,
                              ssim_loss) -> torch.Tensor:
    """optimization_experiments/optimizer.py:29-134 without the plotting branches.

    fwd_data / inv_data: dicts with the keys solve_pose_iteratively emits (train_mono.py:94-100): diff_img, valid_mask,
    weight_mask, auto_mask_error, auto_mask, poses -- e.g. slices of Engine.compute_photometric_error output, with
    'valid_mask' the WARP validity (its key 'warp_valid').  ssim_loss: callable(x, y) e.g. Engine.ssim_loss."""
    B = target_img.shape[0]
    S = options["num_source_imgs"]
    loss = 0
    if options["diff_img_argmin"]:
        diff = torch.cat([fwd_data["diff_img"][i * B:(i + 1) * B] for i in range(S)], 1).unsqueeze(2)
        diff_min, _ = torch.min(diff, 1)                                                        # optimizer.py:47-51
        valid_min = torch.cat([fwd_data["valid_mask"][i * B:(i + 1) * B] for i in range(S)], 1).sum(1, keepdim=True).clamp(0, 1)
        if options["automasking"]:
            am = torch.cat([fwd_data["auto_mask_error"][i * B:(i + 1) * B] for i in range(S)], 1).unsqueeze(2)
            am_min, _ = torch.min(am, 1)
            valid_min = (diff_min < am_min).float() * valid_min                                 # optimizer.py:63-68
        loss = loss + (diff_min * valid_min * fwd_data["weight_mask"][0:B]).sum(3).sum(2).sum(0) / valid_min.sum(3).sum(2).sum(0)
    else:
        loss = loss + 0.25 * (fwd_data["diff_img"] * fwd_data["valid_mask"] * fwd_data["weight_mask"]).sum() / fwd_data["valid_mask"].sum()
    inv_masked = inv_data["diff_img"] * inv_data["valid_mask"] * inv_data["weight_mask"]
    if options["l_inverse_reconstruction"]:
        if options["automasking"]:
            loss = loss + 0.25 * (inv_masked * inv_data["auto_mask"]).sum() / (inv_data["valid_mask"] * inv_data["auto_mask"]).sum()
        else:
            loss = loss + 0.25 * inv_masked.sum() / inv_data["valid_mask"].sum()
    if options["l_depth_consist"]:
        loss = loss + options["l_depth_consist_weight"] * (1 - fwd_data["weight_mask"]).mean()
        if options["l_inverse_reconstruction"]:
            loss = loss + options["l_depth_consist_weight"] * (1 - inv_data["weight_mask"]).mean()
    if options["l_depth_init"]:
        loss = loss + options["l_depth_init_weight"] * ssim_loss(target_disparity, target_disparity_init).mean()
    if options["l_smooth"]:
        loss = loss + options["l_smooth_weight"] * get_smooth_loss(target_disparity, target_img)
    if options["l_pose_consist"]:
        loss = loss + 0.1 * (fwd_data["poses"] + inv_data["poses"]).abs().mean()
    return loss
