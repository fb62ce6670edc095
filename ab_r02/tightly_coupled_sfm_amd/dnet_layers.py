"""Drop-in for the reference's models/dnet_layers.py ScaleRecovery (DNet ground-plane scale, :249-327)."""
from __future__ import annotations

from ._shared import get_engine


class ScaleRecovery:
    """ScaleRecovery(batch_size, height, width)(depth [B,1,H,W], K [B,3,3] or [B,4,4], real_cam_height) -> scale [1].
    A short batch is padded with copies of image 0 up to `batch_size`, as the reference does (:307-311)."""

    def __init__(self, batch_size, height, width):
        self.batch_size, self.height, self.width = int(batch_size), int(height), int(width)

    def to(self, *a, **k):      # nn.Module-style placement calls are accepted: the library follows the tensors' device
        return self

    cuda = eval = train = to

    def forward(self, depth, K, real_cam_height):
        H, W = depth.shape[-2:]
        K3 = K[:, :3, :3].float().contiguous()
        return get_engine(H, W, depth.shape[0]).scale_recovery(depth.float().contiguous(), K3, float(real_cam_height),
                                                               pad_to_batch=self.batch_size)

    __call__ = forward
