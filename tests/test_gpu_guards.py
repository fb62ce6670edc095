"""Guard bands around the library's device allocations (include/tcsfm.h: tcsfm_debug_check_guards; tests/conftest.py switches them on and checks
them after every GPU test -- GPU AddressSanitizer is not available on the pool): the bands are live in this process, they catch a write four
bytes past the end of an allocation, and a refinement in every mode family leaves them intact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_guard_bands_are_on_and_catch_an_overrun():
    from tightly_coupled_sfm_amd import _lib
    assert os.environ.get("TCSFM_DEBUG_GUARDS") == "1"
    det = C.c_int(-2)
    assert _lib.load().tcsfm_debug_guard_selftest(C.byref(det)) == 0 and det.value == 1
    n, bad = _lib.check_guards()          # (the self-test's own allocation is gone; nothing else is damaged)
    assert n >= 0 and bad == 0


def test_every_mode_family_leaves_the_bands_intact():
    from tightly_coupled_sfm_amd import _lib, synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, B, S = 48, 160, 2, 2
    N = 2 * S * B
    e = Engine(H, W, N)            # exactly sized: every per-pair / per-target array is used to its last element
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    tg, sr, dt, ds, K, p0 = [], [[] for _ in range(S)], [], [[] for _ in range(S)], [], [[] for _ in range(S)]
    for b in range(B):
        for s in range(S):
            p = synth.make_pair(H, W, seed=11 + 5 * b, pose_gt=np.array([0.003, -0.002, 0.03, 0.002, -0.003, 0.001]) * (1 if s == 0 else -1), dtype=np.float64)
            if s == 0:
                tg.append(p["tgt"]); dt.append(p["depth_t"]); K.append(p["K"])
            sr[s].append(p["src"]); ds[s].append(p["depth_s"]); p0[s].append(synth.perturb_pose(p["pose_gt"], 3 + s))
    fwd = np.concatenate([np.stack(x) for x in p0])
    tgt, srcs, K = t(np.stack(tg)), t(np.stack([np.stack(x) for x in sr])), t(np.stack(K))
    dt4, ds5 = t(np.stack(dt))[:, None].contiguous(), t(np.stack([np.stack(x) for x in ds]))[:, :, None].contiguous()
    pose = t(np.concatenate([fwd, -fwd]))
    kw = dict(n_iters=2, min_depth=0.06, max_depth=2.67)
    e.refine_window(tgt, srcs, dt4, ds5, K, pose, default_opts(**kw), argmin=True)
    e.refine_window(tgt, srcs, dt4, ds5, K, pose, default_opts(window_rule=_lib.WINDOW_REFERENCE, w_dc=0.15, w_pose_consist=0.1, **kw), argmin=True)
    e.refine_dense_window(tgt, srcs, dt4, ds5, K, pose, default_opts(**kw), argmin=True)
    ref = dict(window_rule=_lib.WINDOW_REFERENCE, w_dc=0.15, prior_init=0.1, lambda_depth=1.0, **kw)
    for extra in (dict(), dict(depth_param=_lib.DEPTH_QUARTER), dict(free_source_depths=1), dict(depth_param=_lib.DEPTH_QUARTER, free_source_depths=1),
                  dict(w_smooth=2.0), dict(w_pose_consist=0.1)):
        e.refine_dense_window(tgt, srcs, dt4, ds5, K, pose, default_opts(**ref, **extra), argmin=True)
    torch.cuda.synchronize()
    n, bad = _lib.check_guards()
    assert n > 20 and bad == 0, (n, bad)
    e.close()
