"""An INDEPENDENT measure of what the solver achieves (VERDICT r03 next #6a): synthetic pairs whose SOURCE is rendered through the
reference's own sampling model (synth.make_pair(sampler_consistent=True): source pixel x stands for the camera coordinate
(x + 1/2)(W-1)/W, models/stn.py:198-231,266), so that the minimiser of the reference's residual is the scene's TRUE pose -- the
truth comes from the renderer, not from the oracle.  At 640x192, from a PoseNet-quality start (synth.perturb_pose: ~3-8 % of the
motion, 0.01-0.04 deg), the refined pose must be within 1 % / 0.02 deg of the truth for the 6-DoF and the pose + depth-scale modes
(16 LM iterations; 8 plain GN iterations converge linearly -- IRLS on an L1 + SSIM residual -- and are held to 2.5 % / 0.06 deg), and
the dense mode must move pose AND depth towards the truth.
(SURVEY 8d's looser perturbation, N(0, 0.01) / N(0, 0.002 rad), is 7+ px of flow on the near ground plane: outside the single-scale
basin for the float64 oracle as well -- measured again in round 4, DESIGN section 5 -- so it is not what is asserted here.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H, W = 192, 640


def _errs(pose, gt):
    return float(np.linalg.norm(pose[:3] - gt[:3]) / np.linalg.norm(gt[:3])), float(np.degrees(np.linalg.norm(pose[3:] - gt[3:])))


def _batch(seeds):
    from tightly_coupled_sfm_amd import synth
    ps = [synth.make_pair(H, W, seed=s, noise=0.0, dtype=np.float32, sampler_consistent=True) for s in seeds]
    init = np.stack([synth.perturb_pose(p["pose_gt"], s) for p, s in zip(ps, seeds)])
    st = lambda k, extra=False: torch.as_tensor(np.stack([p[k][None] if extra else p[k] for p in ps])).cuda()
    return ps, init, dict(tgt=st("tgt"), src=st("src"), depth_t=st("depth_t", True), depth_s=st("depth_s", True), K=st("K"), pose=torch.as_tensor(init).cuda())


def test_refined_poses_reach_the_scene_truth():
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    seeds = (0, 1, 2, 3)
    ps, init, t = _batch(seeds)
    e = Engine(H, W, len(seeds))
    a = (t["tgt"], t["src"], t["depth_t"], t["depth_s"], t["K"], t["pose"])
    e0 = [_errs(init[i].astype(np.float64), ps[i]["pose_gt"].astype(np.float64)) for i in range(len(seeds))]
    assert min(x[0] for x in e0) > 0.015                                                        # the starts are off by 2-8 % of the motion
    # 6-DoF, LM
    pose, _, _ = e.refine(*a, default_opts(n_iters=16, solver=_lib.SOLVER_LM, lambda0=1e-3))
    for i, q in enumerate(pose.cpu().numpy().astype(np.float64)):
        et, er = _errs(q, ps[i]["pose_gt"].astype(np.float64))
        assert et < 0.01 and er < 0.02, ("lm", seeds[i], et, er, e0[i])
    # 6-DoF, 8 plain Gauss-Newton iterations (the BASELINE configs run 4-8 of them)
    pose, _, st = e.refine(*a, default_opts(n_iters=8), stats=True)
    for i, q in enumerate(pose.cpu().numpy().astype(np.float64)):
        et, er = _errs(q, ps[i]["pose_gt"].astype(np.float64))
        assert et < 0.025 and er < 0.06 and et < 0.6 * e0[i][0], ("gn", seeds[i], et, er, e0[i])
    assert torch.all(st[:, 7, 0] < 0.6 * st[:, 0, 0])
    # pose + depth scale: the true depth maps are given, the scale must stay at 1 and the pose reach the truth
    pose, ls, _ = e.refine(*a, default_opts(n_iters=16, solver=_lib.SOLVER_LM, lambda0=1e-3, refine=_lib.REFINE_POSE_SCALE))
    assert float(ls.abs().max()) < 2e-3
    for i, q in enumerate(pose.cpu().numpy().astype(np.float64)):
        et, er = _errs(q, ps[i]["pose_gt"].astype(np.float64))
        assert et < 0.01 and er < 0.02, ("lm7", seeds[i], et, er)


def test_dense_mode_moves_pose_and_depth_towards_the_truth():
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    seeds = (0, 1, 2, 3)
    ps, init, t = _batch(seeds)
    e = Engine(H, W, len(seeds))
    bias = torch.as_tensor((1 + 0.02 * np.sin(np.arange(W) / 17.0)).astype(np.float32)).cuda()
    dep0 = (t["depth_t"] * bias[None, None, None, :]).contiguous()
    pose, depth, _ = e.refine_dense(t["tgt"], t["src"], dep0, t["depth_s"], t["K"], t["pose"], default_opts(n_iters=8, min_depth=0.03, max_depth=3.0))
    truth = t["depth_t"]
    err0 = (dep0 / truth - 1).abs().mean(dim=(1, 2, 3)); err1 = (depth / truth - 1).abs().mean(dim=(1, 2, 3))
    assert torch.all(err1 < 0.97 * err0), (err0, err1)
    for i, q in enumerate(pose.cpu().numpy().astype(np.float64)):
        et, er = _errs(q, ps[i]["pose_gt"].astype(np.float64))
        e0 = _errs(init[i].astype(np.float64), ps[i]["pose_gt"].astype(np.float64))
        assert et < 0.025 and er < 0.06 and et < 0.7 * e0[0], ("dense", seeds[i], et, er, e0)
