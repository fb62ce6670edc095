"""Window forms of the pose modes pack every image ONCE (LinParams::tshare, kernels.h: the target of a pair is read from its partner's bordered
source pack, the auto-mask error from one plane per couple) -- same values, so the refined poses must equal, bit for bit, those of the
round-4 layout (one target pack + one source pack per pair), which TCSFM_TSHARE=0 keeps.  The switch is read once per process: two child
processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T
out = {}
for tag, B, S, H, W, kw in (("kitti_S2_argmin", 1, 2, 48, 160, dict(w_dc=0.15)), ("S1_B3", 3, 1, 48, 160, dict()),
                            ("S3_reference_rule", 2, 3, 24, 40, dict(window_rule=_lib.WINDOW_REFERENCE, w_dc=0.15, w_pose_consist=0.1)),
                            ("pose_scale_lm", 2, 2, 48, 160, dict(refine=_lib.REFINE_POSE_SCALE, solver=_lib.SOLVER_LM, n_iters=5))):
    w = T._window(B, S, H, W, seed=17)
    t = {k: T._dev(v) for k, v in w.items()}
    e = Engine(H, W, 4 * S * B)
    o = default_opts(min_depth=0.06, max_depth=2.67, argmin=1, **kw)
    ls = torch.zeros(2 * S * B, device="cuda") if "refine" in kw else None
    pose, ls_out, st = e.refine_window(t["tgt"], t["srcs"], t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous(), t["K"], t["pose"], o,
                                       argmin=True, stats=True, log_scale=ls)
    out[tag] = pose.cpu().numpy(); out[tag + "_stats"] = st.cpu().numpy()
    e.set_coalesce(2)          # ... and as queued calls merged by the library
    po = [torch.zeros(2 * S * B, 6, device="cuda") for _ in range(2)]
    if "refine" not in kw:
        for p in po:
            e.refine_window_queued(t["tgt"], t["srcs"], t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous(), t["K"], t["pose"], p, o)
        e.flush(); e.synchronize()
        out[tag + "_queued"] = po[1].cpu().numpy()
    e.close()
np.savez(sys.argv[1], **out)
"""


def test_shared_pack_is_bit_identical_to_the_two_pack_layout(tmp_path):
    res = {}
    for v in ("0", "1"):
        f = str(tmp_path / f"poses_{v}.npz")
        env = dict(os.environ, TCSFM_TSHARE=v)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests")), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[v] = dict(np.load(f))
    assert set(res["0"]) == set(res["1"]) and len(res["0"]) >= 10
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res["1"][k]), k
        assert np.isfinite(res["1"][k]).all()
    assert np.array_equal(res["1"]["kitti_S2_argmin"], res["1"]["kitti_S2_argmin_queued"])
