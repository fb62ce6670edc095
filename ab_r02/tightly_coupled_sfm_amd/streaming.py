"""Streaming a frame sequence through the engine (the loop of run_sequential_optimization.py:186-247 -- DataLoader batch ->
H2D in process_sample_batch, data/kitti_loader.py:60-98 -> optimize_window -- re-hosted for the HIP engine).

What the reference does per window: copies the window's 3 frames host -> device again (every frame crosses PCIe up to three
times over a sequence), then optimises it, strictly one window after the other.  Here
  * every frame crosses PCIe ONCE: it is copied asynchronously (pinned host memory, a dedicated copy stream) into a ring of
    device-resident frames; a window is refined from the ring by pointer (window form: no repeated / concatenated tensors);
  * consecutive windows are independent least-squares problems, so they are refined round-robin on the engine's LANES
    (include/tcsfm.h): the next window's kernels fill the gaps the B=1 kernel chain of the current one leaves, and the copies
    of later frames overlap both.
Nothing in here computes: PyTorch provides pinned memory, streams and events.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .engine import Engine, Opts, default_opts


class SequenceRefiner:
    """Refines the windows (frame t = target, frames t+1 .. t+S = sources; forward + inverse directed pairs) of a sequence.
    (Python-level loop, kept for callers that interleave their own torch work; `Engine.refine_sequence` runs the same loop inside
    the library -- tcsfm_refine_sequence -- at about twice the rate.)

    frames / depths: pinned CPU tensors [T,3,H,W] / [T,1,H,W] (or anything torch.as_tensor takes; pinned on the fly),
    K [3,3], init_poses [T-S, 2*S, 6] in the stacked order of train_mono.py:54-62 (forward pairs, then inverse pairs).
    """

    def __init__(self, H: int, W: int, sources: int = 1, lanes: int = 2, ring: Optional[int] = None, opts: Optional[Opts] = None,
                 device: Optional[int] = None):
        self.H, self.W, self.S, self.lanes = H, W, int(sources), int(lanes)
        self.eng = Engine(H, W, 2 * self.S, device=device, lanes=self.lanes)
        self.opts = opts or default_opts(n_iters=4)
        self.R = ring or (self.S + 1 + 2 * self.lanes + 2)
        assert self.R < 64 * self.lanes, "ring slots must be recycled before the lanes' event rings wrap (tcsfm_lane_event)"
        dev = self.eng.dev
        self.ring_img = torch.empty((self.R, 3, H, W), device=dev, dtype=torch.float32)
        self.ring_depth = torch.empty((self.R, 1, H, W), device=dev, dtype=torch.float32)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.copied = [torch.cuda.Event() for _ in range(self.R)]   # frame data has landed in this slot (re-recorded per use)
        self.slot_mark = [[] for _ in range(self.R)]   # lane_events of the windows that read this slot's current frame
        # per-slot views in the shapes the window form takes (target [1,3,H,W] / [1,1,H,W], single source [1,1,3,H,W] / [1,1,1,H,W])
        self.v_tgt = [self.ring_img[k][None] for k in range(self.R)]
        self.v_dt = [self.ring_depth[k][None] for k in range(self.R)]
        self.v_src = [self.ring_img[k][None, None] for k in range(self.R)]
        self.v_ds = [self.ring_depth[k][None, None] for k in range(self.R)]

    def _upload(self, t: int, img: torch.Tensor, depth: torch.Tensor):
        slot = t % self.R
        for mark in self.slot_mark[slot]:      # the slot's previous frame may still be read by the lanes: the copy waits for its readers
            self.eng.stream_wait_event(self.copy_stream, mark)
        self.slot_mark[slot] = []
        with torch.cuda.stream(self.copy_stream):
            self.ring_img[slot].copy_(img, non_blocking=True)
            self.ring_depth[slot].copy_(depth, non_blocking=True)
            self.copied[slot].record(self.copy_stream)

    def run_native(self, frames, depths, K, init_poses, ring: int = 0) -> torch.Tensor:
        """The same loop inside the library (tcsfm_refine_sequence: four-frame copies, events and lanes driven from C++) -> refined
        poses [T-S, 2*S, 6] as a CPU tensor; bit-identical to run()"""
        pin = lambda a: a if (isinstance(a, torch.Tensor) and not a.is_cuda) else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
        return self.eng.refine_sequence(pin(frames), pin(depths), K, init_poses, self.opts, sources=self.S, ring=ring)

    def run(self, frames, depths, K, init_poses) -> torch.Tensor:
        """-> refined poses [T-S, 2*S, 6] (GPU tensor, complete on return)"""
        pin = lambda a: a if (isinstance(a, torch.Tensor) and a.is_pinned()) else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).pin_memory()
        frames, depths = pin(frames), pin(depths)
        T, S, eng = frames.shape[0], self.S, self.eng
        nwin = T - S
        dev = eng.dev
        Kd = torch.as_tensor(np.asarray(K, dtype=np.float32)).reshape(1, 3, 3).to(dev).contiguous()
        p0 = torch.as_tensor(np.asarray(init_poses, dtype=np.float32)).to(dev).contiguous()
        out = torch.empty_like(p0)
        main = torch.cuda.current_stream(eng.device)
        eng._bind()
        p0w, outw = list(p0.unbind(0)), list(out.unbind(0))      # per-window views, made once
        ahead = S + self.lanes                 # frames kept in flight ahead of the window being issued
        nxt = 0
        for w in range(nwin):
            while nxt < T and nxt <= w + ahead and nxt - w < self.R - 1:
                self._upload(nxt, frames[nxt], depths[nxt]); nxt += 1
            slots = [(w + k) % self.R for k in range(S + 1)]
            for s_ in slots:
                main.wait_event(self.copied[s_])
            lane = w % self.lanes
            tgt, dt = self.v_tgt[slots[0]], self.v_dt[slots[0]]
            if S == 1:
                srcs, ds = self.v_src[slots[1]], self.v_ds[slots[1]]
            else:   # ring slots of one window are not contiguous in general: gather the S source frames (device-side, tiny)
                srcs = torch.stack([self.ring_img[s_] for s_ in slots[1:]])[:, None].contiguous()
                ds = torch.stack([self.ring_depth[s_] for s_ in slots[1:]])[:, None].contiguous()
            eng.refine_window_async(lane, tgt, srcs, dt, ds, Kd, p0w[w], outw[w], self.opts)
            mark = eng.lane_event(lane)
            for s_ in slots:
                self.slot_mark[s_].append(mark)
        for lane in range(self.lanes):
            eng.lane_synchronize(lane)
        return out
