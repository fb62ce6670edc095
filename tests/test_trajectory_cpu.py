"""Trajectory composition (validate.py:61-68, pinned by the reference's formula) and the KITTI-style metrics
(pyslam is absent and unpinned: self-consistency properties only)."""
import numpy as np


def _gt_traj(poses):
    from tightly_coupled_sfm_amd.trajectory import compose_trajectory
    return compose_trajectory(poses)[0]


def test_composition_formula():
    from tightly_coupled_sfm_amd.engine import se3_exp
    from tightly_coupled_sfm_amd.trajectory import compose_trajectory
    rng = np.random.default_rng(0)
    poses = rng.normal(scale=[0.02, 0.01, 1.0, 0.002, 0.01, 0.002], size=(30, 6))
    est, cum = compose_trajectory(poses)
    T = np.eye(4)
    for i, p in enumerate(poses):          # est[i+1] = est[i] @ inv(exp(p))
        E = np.vstack([se3_exp(p), [0, 0, 0, 1]])
        T = T @ np.linalg.inv(E)
        assert np.allclose(est[i + 1], T, atol=1e-12)
    assert np.allclose(cum[1:], np.cumsum([np.linalg.norm(se3_exp(p)[:, 3]) for p in poses]))


def test_metrics_properties():
    from tightly_coupled_sfm_amd.trajectory import compute_trajectory
    rng = np.random.default_rng(1)
    poses = np.tile([0.0, 0.0, -1.0, 0.0, 0.004, 0.0], (900, 1)) + rng.normal(scale=1e-4, size=(900, 6))
    gt = _gt_traj(poses)
    est, gt2, errs, cum = compute_trajectory(poses, gt, compute_seg_err=True)
    assert errs == (0.0, 0.0, 0.0, 0.0) and np.allclose(est, gt)
    scaled = poses.copy(); scaled[:, :3] *= 1.05          # 5 % translation scale error -> ~5 % segment error, no rotation error
    _, _, errs, _ = compute_trajectory(scaled, gt, compute_seg_err=True)
    assert 3.5 < errs[2] < 5.5 and errs[3] < 0.02 and errs[0] > 1.0      # (curved path: chord error < arc-length scale error)
