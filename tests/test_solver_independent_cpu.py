"""Independent checks of the solver half (VERDICT r03 "weak" #2 / next #6b): the Gauss-Newton model of the float64 oracle -- which
the HIP kernels follow to 1e-4 -- is compared with FINITE DIFFERENCES of the cost itself, so a design error shared by the oracle
and the kernels (a wrong sign in a Jacobian column, a missing factor in a curvature block) cannot hide behind their agreement:
  * the gradient is the directional derivative of the cost along random pose / scale / depth directions (6-DoF, 7-DoF, dense, and the
    dense mode on the reference's loss);
  * the quadratic model g'd + 1/2 d'Hd of the damped Gauss-Newton step predicts the realised decrease of the cost in sign and size."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.oracle import default_opts


def _crop(g):
    return g["tgt"], g["src"], g["depth_t"], g["depth_s"], g["K"]


def _pair(oracle64, H=24, W=40, seed=3):
    from tightly_coupled_sfm_amd import synth
    p = synth.make_pair(H, W, seed=seed, noise=0.0, dtype=np.float64)
    return p, synth.perturb_pose(p["pose_gt"], seed, sigma_t=3e-4, sigma_r=1e-4)


@pytest.mark.parametrize("nparam,w_dc", [(6, 0.0), (6, 0.15), (7, 0.0)])
def test_gradient_is_the_derivative_of_the_cost_and_the_model_predicts_the_step(nparam, w_dc, oracle64):
    p, pose = _pair(oracle64)
    # automask off: the auto-mask is a discontinuity of the COST (the reference detaches it); validity changes only at the border
    o = default_opts(nparam=nparam, w_dc=w_dc, automask=0, irls_eps=1e-9)
    a = (p["tgt"], p["src"], p["depth_t"], p["depth_s"])
    T0 = oracle64.pose_to_T(pose)
    cost = lambda xi, ls=0.0: oracle64.linearize(*a, pose, p["K"], o, log_scale=ls, T=_left(oracle64, xi, T0))["cost"]
    L = oracle64.linearize(*a, pose, p["K"], o, T=T0)
    rng = np.random.default_rng(0)
    for _ in range(6):
        d = rng.normal(size=nparam) * np.array([1, 1, 1, 0.3, 0.3, 0.3, 1.0][:nparam])
        d /= np.linalg.norm(d)
        eps = 2e-6
        fd = (cost(eps * d[:6], eps * d[6] if nparam == 7 else 0.0) - cost(-eps * d[:6], -eps * d[6] if nparam == 7 else 0.0)) / (2 * eps)
        an = float(L["g"] @ d)
        assert abs(fd - an) < 2e-3 * max(abs(an), np.linalg.norm(L["g"]) * 0.05), (nparam, w_dc, fd, an)
    # the damped Gauss-Newton step and its predicted decrease (7-DoF: the scale prior is part of the solver, not of `cost`: skip)
    if nparam == 6:
        H = L["H"] + 1e-4 * np.diag(np.diag(L["H"]))
        step = -np.linalg.solve(H, L["g"])
        pred = float(L["g"] @ step + 0.5 * step @ L["H"] @ step)
        real = cost(step) - L["cost"]
        assert pred < 0 and real < 0 and 0.3 < real / pred < 1.7, (pred, real)


def _left(orc, xi, T0):
    E = orc.se3_exp(np.asarray(xi, dtype=np.float64))
    R = E[:, :3] @ T0[:, :3]
    t = E[:, :3] @ T0[:, 3] + E[:, 3]
    return np.concatenate([R, t[:, None]], 1)


def test_dense_gradient_is_the_derivative_of_the_cost(oracle64):
    """per-pixel inverse depth: d cost / d rho along smooth and along random depth directions, and the Schur-reduced system predicts
    the decrease of the cost under the joint (pose, depth) step"""
    p, pose = _pair(oracle64)
    o = default_opts(automask=0, irls_eps=1e-9)
    a = lambda dep: (p["tgt"], p["src"], dep, p["depth_s"], pose, p["K"], o)
    L = oracle64.linearize_dense(*a(p["depth_t"]), lambda_depth=0.0, w_prior=0.0)
    rho = 1.0 / p["depth_t"]
    rng = np.random.default_rng(1)
    H_, W_ = rho.shape
    yy, xx = np.mgrid[0:H_, 0:W_]
    for d in (np.sin(xx / 5.0) * np.cos(yy / 3.0), rng.normal(size=rho.shape)):
        d = d / np.abs(d).max() * rho.mean()
        eps = 1e-5
        cp = oracle64.linearize_dense(*a(1.0 / (rho + eps * d)))["cost"]; cm = oracle64.linearize_dense(*a(1.0 / (rho - eps * d)))["cost"]
        fd, an = (cp - cm) / (2 * eps), float((L["g_rho"] * d).sum())
        assert abs(fd - an) < 5e-3 * max(abs(an), 1e-3 * np.abs(L["g_rho"]).sum() * rho.mean()), (fd, an)


def test_dense_reference_gradients_are_the_derivative_of_the_loss(oracle64):
    """the dense mode on the reference's COMPLETE loss (forward + inverse + depth consistency + SSIM prior): finite differences of the
    loss along a depth direction and along every pair's pose directions -- this covers the scattered sampled-depth terms, which no
    per-pixel curvature models"""
    g = load_golden("winloss24x40")
    S, B = g["sources"].shape[:2]
    mind, maxd = (float(x) for x in g["min_max_depth"])
    rd = 1.0 / mind - 1.0 / maxd
    d0 = 1.0 / (1.0 / maxd + rd * g["sig_t0"])
    o = default_opts(n_iters=1, irls_eps=1e-12, w_dc=0.15, automask=0)
    base = (g["target"], g["sources"])
    f = lambda dep, poses: oracle64.linearize_dense_ref(*base, dep, g["depth_s"][:, :, 0], g["K"], poses, o, argmin=False, w_init=0.1, depth0=d0,
                                                        min_depth=mind, max_depth=maxd)
    L = f(g["depth_t"][:, 0], g["first"])
    rho = 1.0 / g["depth_t"][:, 0]
    yy, xx = np.mgrid[0:rho.shape[1], 0:rho.shape[2]]
    d = (np.sin(xx / 4.0 + 1.0) * np.cos(yy / 5.0))[None] * rho.mean() * np.ones((B, 1, 1))
    eps = 2e-6
    fd = (f(1.0 / (rho + eps * d), g["first"])["loss"] - f(1.0 / (rho - eps * d), g["first"])["loss"]) / (2 * eps)
    an = float((L["g_rho"] * d).sum())
    assert abs(fd - an) < 5e-3 * abs(an), (fd, an)
    # pose directions of one forward and one inverse pair (left perturbation -> reference pose vector through the chart Jacobian)
    rng = np.random.default_rng(5)
    for m in (0, S * B + 1):
        dv = rng.normal(size=6) * np.array([1, 1, 1, 0.3, 0.3, 0.3]); dv /= np.linalg.norm(dv)
        A = oracle64.euler_left_jacobian(g["first"][m])           # d xi = A d pose
        gp = A.T @ L["g_xi"][m]
        pp, pm = g["first"].copy(), g["first"].copy()
        pp[m] += 1e-6 * dv; pm[m] -= 1e-6 * dv
        fd = (f(g["depth_t"][:, 0], pp)["loss"] - f(g["depth_t"][:, 0], pm)["loss"]) / 2e-6
        assert abs(fd - float(gp @ dv)) < 5e-3 * max(abs(float(gp @ dv)), 0.05 * np.linalg.norm(gp)), (m, fd, float(gp @ dv))
