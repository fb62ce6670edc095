"""PyTorch-CPU restatement of the REFERENCE-STYLE optimisation step (test infrastructure, NOT product code).

The reference refines by Adam + autograd through its PyTorch residual (optimization_experiments/optimizer.py:217-274),
not by Gauss-Newton.  BASELINE.md section 4 asks for that style of step to be timed on the GPU box's host cores next to the
HIP engine.  The reference's files cannot travel there, so this module restates the path with stock torch ops --
written from the formulas in SURVEY.md section 8a, checked against the golden vectors the reference itself produced
(tests/test_oracle_vs_golden.py::test_torch_twin_vs_reference_golden) -- and bench.py times it.

Only tests/ and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F


def pose_matrix(pose):
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4] with R = Rx Ry Rz  """
    x, y, z = pose[:, 3], pose[:, 4], pose[:, 5]
    cx, sx, cy, sy, cz, sz = x.cos(), x.sin(), y.cos(), y.sin(), z.cos(), z.sin()
    o, l = torch.zeros_like(x), torch.ones_like(x)
    Rx = torch.stack([l, o, o, o, cx, -sx, o, sx, cx], 1).view(-1, 3, 3)
    Ry = torch.stack([cy, o, sy, o, l, o, -sy, o, cy], 1).view(-1, 3, 3)
    Rz = torch.stack([cz, -sz, o, sz, cz, o, o, o, l], 1).view(-1, 3, 3)
    return torch.cat([Rx @ Ry @ Rz, pose[:, :3, None]], 2)


def warp(img, depth, ref_depth, pose, K):
    """back-project, transform, project, sample -> (reconstruction, valid, projected depth, computed depth)"""
    B, _, H, W = img.shape
    dt, dev = img.dtype, img.device
    v, u = torch.meshgrid(torch.arange(H, dtype=dt, device=dev), torch.arange(W, dtype=dt, device=dev), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], 0).view(1, 3, -1)
    cam = (torch.inverse(K) @ pix) * depth.view(B, 1, -1)
    P = K @ pose_matrix(pose)
    pc = P[:, :, :3] @ cam + P[:, :, 3:]
    Z = pc[:, 2].clamp(min=1e-3)
    xn = 2 * (pc[:, 0] / Z) / (W - 1) - 1
    yn = 2 * (pc[:, 1] / Z) / (H - 1) - 1
    xn = torch.where((xn.detach().abs() > 1), torch.full_like(xn, 2.0), xn)      # out-of-range sentinel, per axis
    yn = torch.where((yn.detach().abs() > 1), torch.full_like(yn, 2.0), yn)
    grid = torch.stack([xn, yn], 2).view(B, H, W, 2)
    rec = F.grid_sample(img, grid, padding_mode="zeros", align_corners=False)
    valid = (grid.abs().max(dim=-1)[0] <= 1).to(dt).unsqueeze(1)
    proj = F.grid_sample(ref_depth, grid, padding_mode="zeros", align_corners=False)
    return rec, valid, proj, Z.view(B, 1, H, W)


def ssim(x, y):
    """3x3 box statistics on a reflect-padded image -> clamp((1 - SSIM) / 2, 0, 1)"""
    x, y = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
    mx, my = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x * x, 3, 1) - mx * mx
    sy = F.avg_pool2d(y * y, 3, 1) - my * my
    sxy = F.avg_pool2d(x * y, 3, 1) - mx * my
    n = (2 * mx * my + 1e-4) * (2 * sxy + 9e-4)
    d = (mx * mx + my * my + 1e-4) * (sx + sy + 9e-4)
    return ((1 - n / d) / 2).clamp(0, 1)


def photometric(tgt, src, depth_t, depth_s, pose, K):
    """SURVEY 8a row a7 (single-pair twin of the residual assembly): call sites pass -pose to the warp"""
    rec, valid, pd, cd = warp(src, depth_t, depth_s, -pose, K)
    diff = (0.15 * (rec - tgt).abs().clamp(0, 1) + 0.85 * ssim(tgt, rec)).mean(1, True)
    auto_err = (0.15 * (tgt - src).abs().clamp(0, 1) + 0.85 * ssim(tgt, src)).mean(1, True)
    weight = 1 - ((cd - pd).abs() / (cd + pd)).clamp(0, 1)
    mask = valid * (diff < auto_err).to(diff.dtype)
    return dict(diff=diff, rec=rec, valid=valid, mask=mask, weight=weight, auto_err=auto_err, proj_depth=pd, comp_depth=cd)


def masked_cost(r):
    return (r["diff"] * r["mask"] * r["weight"]).sum() / r["mask"].sum()


def time_adam_steps(tgt, src, sig_t, sig_s, K, pose, seconds, threads, min_depth=0.06, max_depth=2.67):
    """Reference-style optimisation steps on ONE window (forward + inverse directed pair): parameters = the pose 6-vector of
    each direction and the two disparity maps at quarter resolution (optimize_depth_pred, optimizer.py:194-198,235-239);
    loss = forward + 0.25 inverse masked means + 0.15 depth-consistency means (optimizer.py:69-86); backward; Adam (lr 2e-4).
    Runs for about `seconds`; returns (steps per second, steps done)."""
    torch.set_num_threads(threads)
    H, W = tgt.shape[-2:]
    disp = torch.cat([sig_t, sig_s], 1)
    disp = F.interpolate(disp, (H // 4, W // 4), mode="bilinear").clone().requires_grad_()
    poses = torch.cat([pose, -pose], 0).clone().requires_grad_()
    opt = torch.optim.Adam([{"params": [disp, poses], "lr": 2e-4}])
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    K2 = K.repeat(2, 1, 1)
    steps, t0 = 0, time.perf_counter()
    while True:
        opt.zero_grad()
        up = F.interpolate(disp, (H, W), mode="bilinear")
        depth = 1.0 / (lo + (hi - lo) * up)
        d_t, d_s = depth[:, 0:1], depth[:, 1:2]
        r = photometric(torch.cat([tgt, src]), torch.cat([src, tgt]), torch.cat([d_t, d_s]), torch.cat([d_s, d_t]), poses, K2)
        m = r["mask"]
        fwd = (r["diff"][:1] * m[:1] * r["weight"][:1]).sum() / m[:1].sum()
        inv = (r["diff"][1:] * m[1:] * r["weight"][1:]).sum() / m[1:].sum()
        loss = fwd + 0.25 * inv + 0.15 * ((1 - r["weight"][:1]).mean() + (1 - r["weight"][1:]).mean())
        loss.backward()
        opt.step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return steps / dt, steps
