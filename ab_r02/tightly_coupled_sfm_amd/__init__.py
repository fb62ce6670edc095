"""tightly_coupled_sfm_amd -- MI355X-native photometric pose/depth refinement (hot path of
utiasSTARS/tightly-coupled-SfM).  The compute path is libtcsfm_hip.so
(hand-written HIP for gfx950 behind the C ABI of include/tcsfm.h); this package is the thin
Python host side mirroring the reference's call surface."""
from . import synth  # noqa: F401

__all__ = ["synth"]
