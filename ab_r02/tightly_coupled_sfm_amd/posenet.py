"""PoseNet (models/pose_models.py:88-147) on the HIP library: the hand-written gfx950 convolution stack of
csrc/posenet_kernel.h behind the reference module's call convention, and the coupled loop of train_mono.py:41-81 kept inside
the library (tcsfm_solve_pose_iteratively)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import Engine, _chk

_CONV = ["conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7"]


def read_pose_state_dict(path: str, load_best: bool = True) -> dict:
    """'pose_state_dict' of a reference checkpoint (utils/learning_helpers.py:29-37 resolves the file the same way)"""
    import os
    if os.path.isdir(path):
        path = os.path.join(path, "best_model", "best_model.pt") if load_best else os.path.join(path, "checkpoint.pt")
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "pose_state_dict" not in ck:
        raise KeyError(f"{path}: no 'pose_state_dict' entry (keys: {sorted(ck)})")
    return ck["pose_state_dict"]


def is_reference_posenet(module) -> bool:
    """does `module` carry the parameters of the reference's pose_model (conv1..conv7 = (conv2d_wn, GroupNorm, ReLU), pose_pred)?"""
    try:
        sd = module.state_dict()
    except Exception:
        return False
    need = [f"{c}.0.weight" for c in _CONV] + [f"{c}.1.weight" for c in _CONV] + ["pose_pred.weight", "pose_pred.bias"]
    return all(k in sd for k in need) and tuple(sd["conv1.0.weight"].shape) == (16, 6, 7, 7) and tuple(sd["pose_pred.weight"].shape[:2]) == (6, 256)


class PoseNetHIP:
    """pose_model.forward on the engine's GPU.  `params`: a reference pose_model (nn.Module) or its state_dict / a dict of
    numpy arrays under the same names."""

    def __init__(self, engine: Engine, max_images: int, params=None):
        self.eng, self.lib = engine, engine.lib
        self.max_images = int(max_images)
        pn = C.c_void_p()
        engine._call(self.lib.tcsfm_posenet_create(engine._h, self.max_images, C.byref(pn)))
        self._pn = pn
        if params is not None:
            self.load(params)

    def close(self):
        if getattr(self, "_pn", None):
            self.lib.tcsfm_posenet_destroy(self._pn)
            self._pn = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, params):
        sd = params.state_dict() if hasattr(params, "state_dict") else params
        arr = lambda k: None if k not in sd else np.ascontiguousarray(
            (sd[k].detach().cpu().numpy() if isinstance(sd[k], torch.Tensor) else np.asarray(sd[k])), dtype=np.float32)
        keep = []
        def table(suffix):
            t = (C.c_void_p * 7)()
            for i, c in enumerate(_CONV):
                a = arr(f"{c}.{suffix}")
                keep.append(a)
                t[i] = None if a is None else a.ctypes.data
            return t
        cw, cb, gw, gb = table("0.weight"), table("0.bias"), table("1.weight"), table("1.bias")
        hw, hb = arr("pose_pred.weight").reshape(6, 256).copy(), arr("pose_pred.bias")
        self.eng._bind()
        self.eng._call(self.lib.tcsfm_posenet_load(self._pn, cw, cb, gw, gb, hw.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p)))
        return self

    def load_checkpoint(self, path: str, load_best: bool = True):
        """The reference's checkpoint files (utils/learning_helpers.py:20-48: `torch.save` of a dict whose 'pose_state_dict' entry is
        the PoseNet's state_dict; `<dir>/best_model/best_model.pt` or `<dir>/checkpoint.pt`).  `path` is such a file or the directory."""
        self.load(read_pose_state_dict(path, load_best))

    def __call__(self, imgs: torch.Tensor) -> torch.Tensor:
        """pose_model(imgs): imgs [N,6,H,W] -> [N,6]"""
        e = self.eng
        e._bind()
        N = imgs.shape[0]
        imgs = _chk(imgs, (N, 6, e.H, e.W), "imgs")
        out = torch.empty((N, 6), device=imgs.device, dtype=torch.float32)
        e._call(self.lib.tcsfm_posenet_forward(self._pn, N, e._p(imgs), e._p(out)))
        return out

    def odometry_sequence(self, frames, depths, K, opts=None, sources: int = 1, iterations: int = 4, ring: int = 0, windows_per_call: int = 0,
                          target_pos: int = 0):
        """tcsfm_odometry_sequence: for every window of a sequence (frames [T,3,H,W] / depths [T,1,H,W] CPU tensors, pinned for
        asynchronous copies; K [3,3]) the coupled PoseNet loop gives the initial poses and the engine refines them, windows
        running on the engine's lanes -> (initial poses, refined poses), each [T-S, 2S, 6] CPU tensors"""
        e = self.eng
        e._bind()
        from .engine import default_opts
        o = opts or default_opts()
        T, S = int(frames.shape[0]), int(sources)
        frames = e._cpu(frames, (T, 3, e.H, e.W), "frames"); depths = e._cpu(depths, (T, 1, e.H, e.W), "depths")
        Kc = e._cpu(torch.as_tensor(np.asarray(K, dtype=np.float32)), (3, 3), "K")
        init = torch.empty((T - S, 2 * S, 6), dtype=torch.float32); out = torch.empty_like(init)
        hp = lambda t: C.c_void_p(t.data_ptr())
        e._call(self.lib.tcsfm_odometry_sequence(e._h, self._pn, int(iterations), C.byref(o), T, S, hp(frames), hp(depths), hp(Kc), hp(init), hp(out),
                                                 None, int(ring), int(windows_per_call), int(target_pos)))
        return init, out

    def solve_pose_iteratively(self, num_iter: int, tgt, srcs, depth_t, depth_s, K):
        """the coupled loop of train_mono.py:41-81 for a window (layouts of Engine.refine_window) -> (poses [2SB,6], stacked [2SB,num_iter,6])"""
        e = self.eng
        e._bind()
        if isinstance(srcs, (list, tuple)):
            srcs = torch.stack(list(srcs), 0)
        if isinstance(depth_s, (list, tuple)):
            depth_s = torch.stack(list(depth_s), 0)
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        tgt = _chk(tgt, (B, 3, e.H, e.W), "tgt"); srcs = _chk(srcs, (S, B, 3, e.H, e.W), "srcs")
        depth_t = _chk(depth_t, (B, 1, e.H, e.W), "depth_t"); depth_s = _chk(depth_s, (S, B, 1, e.H, e.W), "depth_s")
        K = _chk(K, (B, 3, 3), "K")
        N = 2 * S * B
        poses = torch.empty((N, 6), device=tgt.device, dtype=torch.float32)
        stacked = torch.empty((N, int(num_iter), 6), device=tgt.device, dtype=torch.float32)
        e._call(self.lib.tcsfm_solve_pose_iteratively(e._h, self._pn, int(num_iter), B, S, e._p(tgt), e._p(srcs), e._p(depth_t), e._p(depth_s),
                                                      e._p(K), e._p(poses), e._p(stacked)))
        return poses, stacked
