"""End to end on a synthetic sequence, the way run_sequential_optimization.py uses the optimiser:
per-window refinement of PoseNet-quality initial poses, then trajectory composition (validate.py:61-68) and the odometry
error metrics.

The yardstick is the photometric optimum, not the scene's ground-truth pose: the reference's warp samples at
u_proj W/(W-1) - 1/2 ("identity pose is not an identity warp"), while the synthetic views are rendered
from the true geometry, so the minimiser of the reference's residual sits a few per cent of the motion away from the
true pose (bench.py reports that distance as `check`).  In the reference's own pipeline the networks are trained through
the same warp and absorb the convention."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_refined_trajectory_beats_initial_trajectory():
    from tightly_coupled_sfm_amd import synth, trajectory
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, NF = 96, 320, 12
    rng = np.random.default_rng(5)
    gt, init, pairs = [], [], []
    for i in range(NF):
        pose = np.array([0.002, -0.001, 0.033, 0.001, -0.003, 0.001]) + rng.normal(scale=[3e-4, 3e-4, 2e-3, 5e-4, 1e-3, 5e-4])
        p = synth.make_pair(H, W, seed=200 + i, pose_gt=pose, dtype=np.float32)
        pairs.append(p); gt.append(p["pose_gt"].astype(np.float64))
        init.append(synth.perturb_pose(p["pose_gt"], 300 + i).astype(np.float64))          # PoseNet-quality start
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    stack = lambda k: t(np.stack([p[k] for p in pairs]))
    e = Engine(H, W, NF)
    pose, _, st = e.refine(stack("tgt"), stack("src"), stack("depth_t")[:, None], stack("depth_s")[:, None], stack("K"), t(np.stack(init)),
                           default_opts(n_iters=8), stats=True)
    refined = pose.cpu().numpy().astype(np.float64)
    # the optimum: a long refinement started at the true poses
    opt, _, _ = e.refine(stack("tgt"), stack("src"), stack("depth_t")[:, None], stack("depth_s")[:, None], stack("K"), t(np.stack(gt)),
                         default_opts(n_iters=40))
    optimum = opt.cpu().numpy().astype(np.float64)
    traj_opt, _ = trajectory.compose_trajectory(optimum)
    traj_init, _ = trajectory.compose_trajectory(np.stack(init))
    traj_ref, _ = trajectory.compose_trajectory(refined)
    et_i, er_i = trajectory.mean_err(traj_opt, traj_init)
    et_r, er_r = trajectory.mean_err(traj_opt, traj_ref)
    assert et_r < 0.15 * et_i and er_r < 0.15 * er_i, (et_i, et_r, er_i, er_r)      # 8 iterations from the perturbed start reach it
    path = np.sum(np.linalg.norm(optimum[:, :3], axis=1))
    assert np.linalg.norm(traj_ref[-1][:3, 3] - traj_opt[-1][:3, 3]) < 2e-3 * path  # endpoint agreement over the sequence
    seg = trajectory.segment_errors(traj_opt, traj_ref, [0.1, 0.2])
    assert np.all(seg[:, 1] < 0.01)                                                 # < 1 % translation error on every segment length
    assert np.all(st.cpu().numpy()[:, 7, 0] < st.cpu().numpy()[:, 0, 0])
