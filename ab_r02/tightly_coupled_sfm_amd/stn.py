"""Drop-ins for the reference's models/stn.py functions on the hot path (same names, arguments and outputs)."""
from __future__ import annotations

import torch

from ._shared import get_engine


def inverse_warp2(img, depth, ref_depth, pose, intrinsics, padding_mode="zeros"):
    """models/stn.py:234-273: img [B,3,H,W], depth / ref_depth [B,1,H,W], pose [B,6] (call sites pass -pose), intrinsics
    [B,3,3] -> (projected_img, valid_mask, projected_depth, computed_depth).  One fused HIP kernel instead of ~60 torch ops."""
    if padding_mode != "zeros":
        raise NotImplementedError("the reference's hot path only uses padding_mode='zeros' (train_mono.py:69)")
    B, _, H, W = img.shape
    return get_engine(H, W, B).inverse_warp2(img.float(), depth.float(), ref_depth.float(), pose.float(), intrinsics.float())


def euler2mat(angle):
    """models/stn.py:81-116: [B,3] (rx,ry,rz) -> R = Rx Ry Rz [B,3,3]"""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    cx, sx, cy, sy, cz, sz = torch.cos(x), torch.sin(x), torch.cos(y), torch.sin(y), torch.cos(z), torch.sin(z)
    o, l = torch.zeros_like(x), torch.ones_like(x)
    Rx = torch.stack([l, o, o, o, cx, -sx, o, sx, cx], 1).reshape(-1, 3, 3)
    Ry = torch.stack([cy, o, sy, o, l, o, -sy, o, cy], 1).reshape(-1, 3, 3)
    Rz = torch.stack([cz, -sz, o, sz, cz, o, o, o, l], 1).reshape(-1, 3, 3)
    return Rx @ Ry @ Rz


def pose_vec2mat(vec, rotation_mode="euler"):
    """models/stn.py:143-158: [B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4]"""
    if rotation_mode != "euler":
        raise NotImplementedError("only the Euler mode is on the reference's hot path")
    return torch.cat([euler2mat(vec[:, 3:6]), vec[:, :3].unsqueeze(-1)], 2)
