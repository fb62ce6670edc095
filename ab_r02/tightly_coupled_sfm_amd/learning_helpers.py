"""Drop-ins for the reference's utils/learning_helpers.py functions on the return path of the optimiser."""
from __future__ import annotations

from ._shared import get_engine
from .optimizer import batch_post_process_disparity  # noqa: F401   (utils/learning_helpers.py:115-123)


def disp_to_depth(disp, min_depth, max_depth):
    """utils/learning_helpers.py:77-86: sigmoid disparity -> (scaled_disp, depth); one HIP kernel, one HBM round trip"""
    H, W = disp.shape[-2:]
    return get_engine(H, W, 1).disp_to_depth(disp, min_depth, max_depth)
