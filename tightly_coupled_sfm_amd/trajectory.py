"""Trajectory composition and KITTI-style odometry metrics.

`compute_trajectory` mirrors validate.py:61-103: the estimated trajectory is composed from the per-frame 6-vectors
exactly as the reference does -- est[i+1] = (exp(pose_i) est[i]^-1)^-1 -- using this library's own SE(3) routines
(liegroups is an absent third-party dependency of the reference).

The error metrics in the reference come from pyslam.metrics.TrajectoryMetrics (also absent, version unpinned, no
fixtures in the reference): PARITY UNPINNED.  What is implemented here is the standard definition:
  mean_err        MEAN over frames of the per-frame error norms |trans(T_gt^-1 T_est)| and |log(rot(T_gt^-1 T_est))| (what
                  validate.compute_trajectory prints as "mean trans. / rot. error"); rms_err is the RMSE of the same norms
  segment_errors  KITTI devkit: for every start frame and every segment length L, the relative-motion error between
                  estimate and ground truth over the first sub-trajectory of (ground-truth) length >= L, divided by L.
"""
from __future__ import annotations

import numpy as np

from .engine import se3_exp, se3_inv, se3_log, se3_mul


def _T4(T34):
    return np.vstack([np.asarray(T34, dtype=np.float64).reshape(3, 4), [0, 0, 0, 1]])


def compose_trajectory(pose_vec, T0=None):
    """est[0] = T0 (4x4, default identity); est[i+1] = est[i] exp(pose_i)^-1   (validate.py:64-68)."""
    est = [np.eye(4) if T0 is None else np.asarray(T0, dtype=np.float64)]
    cum = [0.0]
    for p in np.asarray(pose_vec, dtype=np.float64):
        dT = se3_exp(p)
        est.append(_T4(se3_inv(se3_mul(dT, se3_inv(est[-1][:3])))))
        cum.append(cum[-1] + float(np.linalg.norm(dT[:, 3])))
    return np.array(est), np.array(cum)


def _rel_err(Tg, Te):
    E = se3_mul(se3_inv(Tg[:3]), Te[:3])
    return float(np.linalg.norm(E[:, 3])), float(np.linalg.norm(se3_log(E)[3:]))


def error_norms(gt_traj, est_traj):
    """per-frame (translational, rotational [rad]) error norms of T_gt^-1 T_est -> array [n, 2]"""
    return np.array([_rel_err(g, t) for g, t in zip(gt_traj, est_traj)])


def mean_err(gt_traj, est_traj):
    """mean of the per-frame error norms (NOT their RMSE: that is rms_err)"""
    e = error_norms(gt_traj, est_traj)
    return float(np.mean(e[:, 0])), float(np.mean(e[:, 1]))


def rms_err(gt_traj, est_traj):
    e = error_norms(gt_traj, est_traj)
    return float(np.sqrt(np.mean(e[:, 0] ** 2))), float(np.sqrt(np.mean(e[:, 1] ** 2)))


def segment_errors(gt_traj, est_traj, seg_lengths, step=1, return_all=False):
    """-> rows [n_lengths, 3] = (length, mean translational error / length, mean rotational error / length); with return_all also the
    individual segments [n_segments, 3] = (length, translational error / length, rotational error / length), one per start frame"""
    gt = np.asarray(gt_traj); est = np.asarray(est_traj)
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1))])
    rows, every = [], []
    for L in seg_lengths:
        errs = []
        for i in range(0, len(gt), step):
            j = np.searchsorted(dist, dist[i] + L)
            if j >= len(gt):
                break
            dg = se3_mul(se3_inv(gt[i][:3]), gt[j][:3]); de = se3_mul(se3_inv(est[i][:3]), est[j][:3])
            t, r = _rel_err(_T4(dg), _T4(de))
            errs.append((t / L, r / L))
        if errs:
            errs = np.array(errs)
            rows.append((L, errs[:, 0].mean(), errs[:, 1].mean()))
            every += [(L, t, r) for t, r in errs]
        else:
            rows.append((L, np.nan, np.nan))
    rows = np.array(rows)
    return (rows, np.array(every).reshape(-1, 3)) if return_all else rows


def compute_trajectory(pose_vec, gt_traj, method="odom", compute_seg_err=False, verbose=False):
    """validate.py:61-103 -> (est_traj [n+1,4,4], gt_traj, (mean_trans, mean_rot_deg, seg_trans_%, seg_rot_deg_per_100m), cum_dist)"""
    est, cum = compose_trajectory(pose_vec, gt_traj[0])
    mt, mr = mean_err(gt_traj, est)
    mt, mr = round(mt, 3), round(mr * 180 / np.pi, 3)
    ts, rs = 0, 0
    if compute_seg_err:
        seg = segment_errors(gt_traj, est, list(range(100, 801, 100)))
        if np.isnan(np.mean(seg[:, 1])):
            max_dist = cum[-1] - cum[-1] % 100 + 1 - 100
            seg = segment_errors(gt_traj, est, list(range(100, int(max_dist), 100)))
        rs = round(100 * float(np.mean(seg[:, 2])) * 180 / np.pi, 3)
        ts = round(float(np.mean(seg[:, 1])) * 100, 3)
    if verbose:
        print(f"{method} mean trans. error: {mt} | mean rot. error: {mr}")
    return est, np.array(gt_traj), (mt, mr, ts, rs), cum


class TrajectoryMetrics:
    """Stand-in for pyslam.metrics.TrajectoryMetrics as validate.py:73-91 uses it (absent, unpinned third-party dependency:
    PARITY UNPINNED, standard definitions -- see the module docstring).  Trajectories: lists of SE3 objects (anything with
    `.as_matrix()`) or of 4x4 / 3x4 matrices, camera-to-world ('Twv')."""

    def __init__(self, poses_gt, poses_est, convention="Twv"):
        if convention != "Twv":
            raise NotImplementedError("the reference only uses convention='Twv' (validate.py:73)")
        mat = lambda T: _T4(np.asarray(T.as_matrix() if hasattr(T, "as_matrix") else T, dtype=np.float64)[:3])
        self.gt, self.est = np.array([mat(T) for T in poses_gt]), np.array([mat(T) for T in poses_est])

    def mean_err(self):
        """-> (mean translational error, mean rotational error [rad])"""
        t, r = mean_err(self.gt, self.est)
        return np.float64(t), np.float64(r)

    def rms_err(self):
        """-> (RMS translational error, RMS rotational error [rad])"""
        t, r = rms_err(self.gt, self.est)
        return np.float64(t), np.float64(r)

    def segment_errors(self, segment_lengths, rot_unit="rad"):
        """-> (every segment [n_segments, 3], per-length averages [n_lengths, 3]); columns: length, translational error / length,
        rotational error / length.  validate.py:82 reads the second element."""
        rows, every = segment_errors(self.gt, self.est, list(segment_lengths), return_all=True)
        if rot_unit == "deg":
            rows = rows.copy(); rows[:, 2] *= 180.0 / np.pi
            every = every.copy(); every[:, 2] *= 180.0 / np.pi
        return every, rows
