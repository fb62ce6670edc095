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


def test_metric_definitions_on_a_hand_computed_case():
    """mean_err is the MEAN of the per-frame error norms, rms_err their RMSE; numbers worked out by hand: ground truth at rest,
    estimates displaced by 0, 3, 4 units along x and rotated by 0, 0.1, 0.2 rad about z"""
    from tightly_coupled_sfm_amd.trajectory import TrajectoryMetrics, error_norms, mean_err, rms_err
    def T(tx, rz):
        M = np.eye(4); c, s = np.cos(rz), np.sin(rz)
        M[:2, :2] = [[c, -s], [s, c]]; M[0, 3] = tx
        return M
    gt = [np.eye(4)] * 3
    est = [T(0.0, 0.0), T(3.0, 0.1), T(4.0, 0.2)]
    e = error_norms(gt, est)
    assert np.allclose(e[:, 0], [0, 3, 4]) and np.allclose(e[:, 1], [0, 0.1, 0.2], atol=1e-12)
    mt, mr = mean_err(gt, est)
    assert abs(mt - 7.0 / 3.0) < 1e-12 and abs(mr - 0.1) < 1e-12
    rt, rr = rms_err(gt, est)
    assert abs(rt - np.sqrt(25.0 / 3.0)) < 1e-12 and abs(rr - np.sqrt(0.05 / 3.0)) < 1e-12
    tm = TrajectoryMetrics(gt, est)
    assert np.allclose(tm.mean_err(), (mt, mr)) and np.allclose(tm.rms_err(), (rt, rr)) and rt > mt


def test_reference_style_trajectory_code_runs_on_the_stand_ins():
    """the body of the reference's compute_trajectory (validate.py:61-91) written against liegroups.SE3 and
    pyslam TrajectoryMetrics, executed with this package's stand-ins for both absent dependencies"""
    from tightly_coupled_sfm_amd.liegroups import SE3
    from tightly_coupled_sfm_amd.trajectory import TrajectoryMetrics, compute_trajectory
    rng = np.random.default_rng(1)
    rel = np.array([0.0, 0.0, 1.0, 0.0, 0.01, 0.0]) + 0.01 * rng.normal(size=(60, 6))
    gt = [np.eye(4)]
    for p in rel:
        gt.append(SE3.as_matrix((SE3.exp(p).dot(SE3.from_matrix(gt[-1], normalize=True).inv())).inv()))
    noisy = rel + 0.002 * rng.normal(size=rel.shape)
    est = [gt[0]]
    for p in noisy:                                                        # validate.py:64-68, verbatim structure
        dT = SE3.exp(p)
        est.append(SE3.as_matrix((dT.dot(SE3.from_matrix(est[-1], normalize=True).inv())).inv()))
    tm = TrajectoryMetrics([SE3.from_matrix(T, normalize=True) for T in gt], [SE3.from_matrix(T, normalize=True) for T in est], convention="Twv")
    mt, mr = tm.mean_err()
    _, seg = tm.segment_errors([10, 20, 30], rot_unit="rad")
    assert mt > 0 and mr > 0 and seg.shape == (3, 3) and np.all(seg[:, 1] < 0.05)
    e2, g2, errors, cum = compute_trajectory(noisy, np.array(gt))
    assert np.allclose(e2, np.array(est), atol=1e-12) and abs(errors[0] - round(float(mt), 3)) < 1e-9


def test_compute_trajectory_vs_reference_code_G11():
    """golden G11: the REFERENCE's validate.compute_trajectory (validate.py:61-103), executed with this package's SE3 /
    TrajectoryMetrics stand-ins in place of the absent liegroups / pyslam: same composition, errors and rounding"""
    from conftest import load_golden
    from tightly_coupled_sfm_amd.trajectory import compute_trajectory
    g = load_golden("traj400")
    est, gt, errors, cum = compute_trajectory(g["pose_vec"], g["gt_traj"], compute_seg_err=True)
    assert np.allclose(est, g["est_traj"], atol=1e-12) and np.allclose(cum, g["cum_dist"], atol=1e-12)
    assert np.allclose(np.array(errors, dtype=np.float64), g["errors"], atol=1e-9)
    assert g["errors"][2] > 0 and g["errors"][3] > 0          # the 100..800-unit segment errors were actually evaluated
