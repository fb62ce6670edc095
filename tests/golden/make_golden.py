#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference, pure PyTorch) is imported with its absent third-party
dependencies stubbed (liegroups, cv2, pykitti, pyslam, torchvision, tensorboardX, imageio --
none of them is touched by the hot path).  It is executed on seeded
synthetic inputs from tightly_coupled_sfm_amd.synth in float64 and float32, and inputs +
outputs are stored as small .npz fixtures.  Nothing from the reference's source text is
copied: fixtures are data only.  /root/reference does not exist on the GPU box; tests read
the committed .npz files.

Goldens (SURVEY.md section 8c naming):
  G1 inverse_warp2 4 outputs incl. OOB / border / Z<1e-3            models/stn.py:234-273
  G2 SSIM_Loss map                                                   losses.py:27-41
  G3 compute_photometric_error dict, several poses                   optimization_experiments/helpers.py:8-23
  G4 solve_pose_iteratively error images, constant-pose PoseNet      train_mono.py:41-120
  G5 compute_optimization_loss scalar, default options + toggles     optimization_experiments/optimizer.py:29-134
  G6 autograd d(cost)/d(pose), d(cost)/d(depth); per-pixel Jacobian rows of the three residual maps
  G7 generate_loss_surface tz / yaw sweeps                           optimization_experiments/plot_loss_surface.py:11-87
  G8 disp_to_depth, batch_post_process_disparity, avg_final_predictions
  G9 one DepthOptimizer.optimize_window (optimize_depth_pred, 5 epochs, stand-in nets): result-dict schema + the values
     that do not depend on the optimiser (initial poses, depths, flip-averaged disparity)   optimizer.py:136-297
  G10 ScaleRecovery                                                   models/dnet_layers.py:249-327
  G13 compute_optimization_loss with S = 2 sources and its autograd gradients w.r.t. the poses of all directed pairs, the SHARED
      target depth and the source depths; term by term (forward / + inverse / + depth consistency / without argmin)   optimizer.py:29-134
  G11 validate.compute_trajectory run on this package's stand-ins for the absent liegroups / pyslam (pins the
      composition order, the error bookkeeping and the rounding of the reference code; the SE(3) and metric arithmetic
      itself stays unpinned)                                          validate.py:61-103
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass


def import_reference():
    _stub("liegroups", SE3=_Dummy, SO3=_Dummy); _stub("liegroups.torch", SE3=_Dummy, SO3=_Dummy)
    _stub("cv2"); _stub("pykitti"); _stub("pyslam"); _stub("pyslam.metrics", TrajectoryMetrics=_Dummy)
    tv = _stub("torchvision"); tv.models = _stub("torchvision.models", ResNet=nn.Module)
    _stub("torchvision.models.resnet"); tv.transforms = _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional"); tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    _stub("tensorboardX", SummaryWriter=_Dummy); _stub("imageio")
    sys.path.insert(0, REF); sys.path.insert(0, os.path.join(REF, "optimization_experiments"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "optimization_experiments"))
    import warnings
    warnings.simplefilter("ignore")
    import helpers, losses, train_mono, optimizer, plot_loss_surface  # noqa
    sys.path.insert(0, REF)
    import validate  # noqa
    from models import stn
    from utils import learning_helpers
    os.chdir(cwd)
    return dict(helpers=helpers, losses=losses, train_mono=train_mono, optimizer=optimizer,
                plot_loss_surface=plot_loss_surface, stn=stn, learning_helpers=learning_helpers, validate=validate)


def main():
    from tightly_coupled_sfm_amd import synth
    ref = import_reference()
    stn, helpers, losses, lh = ref["stn"], ref["helpers"], ref["losses"], ref["learning_helpers"]
    out = {}

    def T(a, dt):
        return torch.tensor(np.asarray(a), dtype=dt)

    def N(t):
        return t.detach().cpu().numpy()

    def reset_grid():
        # the reference caches its pixel grid in a module global keyed only on height (stn.py:10-21,43-44)
        stn.pixel_coords = None

    # ------------------------------------------------------------------ G1/G2/G3/G6 per size & dtype
    cases = [("s8x16", 8, 16, 11), ("s24x40", 24, 40, 3), ("s48x160", 48, 160, 5)]
    for name, H, W, seed in cases:
        p = synth.make_pair(H, W, seed=seed, dtype=np.float64)
        poses = [synth.perturb_pose(p["pose_gt"], seed + k) for k in range(3)]
        # stress poses: large yaw (OOB columns), strong backwards translation (Z clamp), identity
        poses += [np.array([0.0, 0, 0, 0, 0.35, 0]), np.array([0.0, 0.0, 2.0, 0.0, 0.0, 0.0]),
                  np.zeros(6), np.array([0.05, -0.02, -0.1, 0.02, -0.03, 0.04])]
        if H * W > 2000:      # keep the larger fixture small: one nominal, one OOB-heavy, one mixed pose
            poses = [poses[0], poses[3], poses[6]]
        poses = np.stack(poses)
        g = dict(tgt=p["tgt"], src=p["src"], depth_t=p["depth_t"], depth_s=p["depth_s"], K=p["K"], poses=poses)
        for dtn, dt in (("f64", torch.float64), ("f32", torch.float32)):
            reset_grid()
            t, s = T(p["tgt"], dt)[None], T(p["src"], dt)[None]
            dpt, dps = T(p["depth_t"], dt)[None, None], T(p["depth_s"], dt)[None, None]
            K = T(p["K"], dt)[None]
            recs, vals, pds, cds, diffs, wts, masks = [], [], [], [], [], [], []
            costs, gposes = [], []
            for k in range(len(poses)):
                po = T(poses[k], dt)[None].clone().requires_grad_()
                rec, valid, pd, cd = stn.inverse_warp2(s, dpt, dps, -po, K, "zeros")
                recs.append(N(rec[0])); vals.append(N(valid[0, 0])); pds.append(N(pd[0, 0])); cds.append(N(cd[0, 0]))
                r = helpers.compute_photometric_error(t, s, dpt, dps, po, K)
                diffs.append(N(r["diff_img"][0, 0])); wts.append(N(r["weight_mask"][0, 0])); masks.append(N(r["valid_mask"][0, 0]))
                den = r["valid_mask"].sum()
                L = (r["diff_img"] * r["valid_mask"] * r["weight_mask"]).sum() / den
                costs.append(L.item() if den.item() > 0 else 0.0)
                if den.item() > 0:
                    L.backward()
                    gposes.append(N(po.grad[0]))
                else:
                    gposes.append(np.zeros(6))
            g[f"{dtn}_rec"] = np.stack(recs); g[f"{dtn}_valid"] = np.stack(vals)
            g[f"{dtn}_proj_depth"] = np.stack(pds); g[f"{dtn}_comp_depth"] = np.stack(cds)
            g[f"{dtn}_diff"] = np.stack(diffs); g[f"{dtn}_weight"] = np.stack(wts); g[f"{dtn}_mask"] = np.stack(masks)
            g[f"{dtn}_cost"] = np.array(costs); g[f"{dtn}_grad_pose"] = np.stack(gposes)
            g[f"{dtn}_ssim_ts"] = N(losses.SSIM_Loss()(t, s)[0])
        out[name] = g

    # ------------------------------------------------------------------ G6: depth gradients + Jacobian rows (f64, 24x40)
    H, W, seed = 24, 40, 3
    p = synth.make_pair(H, W, seed=seed, dtype=np.float64)
    pose = synth.perturb_pose(p["pose_gt"], seed)
    dt = torch.float64
    reset_grid()
    t, s = T(p["tgt"], dt)[None], T(p["src"], dt)[None]
    K = T(p["K"], dt)[None]
    ssim = losses.SSIM_Loss()

    def rows(po, dpt, dps):
        """the three per-pixel residual maps the GN engine uses, composed from reference functions only
        (same composition as helpers.py:11-14): E1 = W*mean_c .15|.|, E2 = W*mean_c .85 SSIM, E3 = 1-W"""
        rec, valid, pd, cd = stn.inverse_warp2(s, dpt, dps, -po, K, "zeros")
        e1 = (0.15 * (rec - t).abs().clamp(0, 1)).mean(1, True)
        e2 = (0.85 * ssim(t, rec)).mean(1, True)
        dd = ((cd - pd).abs() / (cd + pd)).clamp(0, 1)
        w = 1 - dd
        return torch.stack([w * e1, w * e2, dd], 0).reshape(3, H, W)

    po = T(pose, dt)[None]
    dpt, dps = T(p["depth_t"], dt)[None, None], T(p["depth_s"], dt)[None, None]
    Jp = torch.autograd.functional.jacobian(lambda q: rows(q, dpt, dps), po, vectorize=True)  # [3,H,W,1,6]
    # depth-scale column: d/d(log s) with both depth maps scaled by s
    ls = torch.zeros((), dtype=dt)
    Js = torch.autograd.functional.jacobian(lambda q: rows(po, dpt * torch.exp(q), dps * torch.exp(q)), ls, vectorize=True)
    po2 = po.clone().requires_grad_(); d1 = dpt.clone().requires_grad_(); d2 = dps.clone().requires_grad_()
    r = helpers.compute_photometric_error(t, s, d1, d2, po2, K)
    L = (r["diff_img"] * r["valid_mask"] * r["weight_mask"]).sum() / r["valid_mask"].sum()
    L.backward()
    out["jac24x40"] = dict(tgt=p["tgt"], src=p["src"], depth_t=p["depth_t"], depth_s=p["depth_s"], K=p["K"], pose=pose,
                           E=N(rows(po, dpt, dps)), J_pose=N(Jp.reshape(3, H, W, 6)), J_logscale=N(Js.reshape(3, H, W)),
                           cost=L.item(), grad_pose=N(po2.grad[0]), grad_depth_t=N(d1.grad[0, 0]), grad_depth_s=N(d2.grad[0, 0]),
                           mask=N(r["valid_mask"][0, 0]))

    # ------------------------------------------------------------------ G4/G5: batched solve_pose_iteratively + loss
    class ConstPose(nn.Module):
        """stand-in PoseNet: returns fixed per-sample poses on the first call and fixed small
        corrections on later calls (bypasses the conv net, keeps the reference control flow)."""
        def __init__(self, first, corr):
            super().__init__(); self.first, self.corr, self.calls = first, corr, 0
        def forward(self, x):
            self.calls += 1
            return self.first.clone() if self.calls == 1 else self.corr.clone()

    B, S, H, W = 2, 2, 24, 40
    dt = torch.float64
    reset_grid()
    tg, srcs, dts, dss, Ks, pgt = [], [[] for _ in range(S)], [], [[] for _ in range(S)], [], [[] for _ in range(S)]
    for b in range(B):
        for si in range(S):
            sign = 1.0 if si == 0 else -1.0
            base = np.array([0.003, -0.002, 0.033, 0.002, -0.004, 0.0015]) * sign
            p = synth.make_pair(H, W, seed=20 + b, pose_gt=base, dtype=np.float64)
            if si == 0:
                tg.append(p["tgt"]); dts.append(p["depth_t"]); Ks.append(p["K"])
            srcs[si].append(p["src"]); dss[si].append(p["depth_s"]); pgt[si].append(p["pose_gt"])
    target = T(np.stack(tg), dt); source_list = [T(np.stack(x), dt) for x in srcs]
    depths = [T(np.stack(dts), dt)[:, None]] + [T(np.stack(x), dt)[:, None] for x in dss]
    Kb = T(np.stack(Ks), dt)
    rng = np.random.default_rng(7)
    # stacked order of solve_pose_iteratively: [fwd s0 b0..bB-1, fwd s1 ..., inv s0 ..., inv s1 ...]
    gt_f = np.concatenate([np.stack(x) for x in pgt]); first = np.concatenate([gt_f, -gt_f]) + rng.normal(scale=2e-3, size=(2 * S * B, 6))
    corr = rng.normal(scale=3e-4, size=(2 * S * B, 6))
    g4 = dict(target=N(target), sources=np.stack([N(x) for x in source_list]), depths=np.stack([N(d) for d in depths]),
              K=N(Kb), first=first, corr=corr)
    for iters in (1, 4):
        pm = ConstPose(T(first, dt), T(corr, dt))
        poses, poses_inv, outputs = ref["train_mono"].solve_pose_iteratively(iters, depths, pm, target, source_list, Kb, return_errors=True)
        for d in ("fwd", "inv"):
            for k in ("diff_img", "valid_mask", "weight_mask", "auto_mask_error", "auto_mask", "poses", "img_rec"):
                g4[f"it{iters}_{d}_{k}"] = N(outputs[d][k])
        g4[f"it{iters}_poses"] = np.stack([N(x) for x in poses]); g4[f"it{iters}_poses_inv"] = np.stack([N(x) for x in poses_inv])
        # G5: scalar optimisation loss under the default options and toggles
        base_opts = {'epochs': 20, 'diff_img_argmin': True, 'automasking': True, 'l_depth_consist': True,
                     'l_depth_consist_weight': 0.15, 'l_depth_init': True, 'l_depth_init_weight': 0.1,
                     'l_inverse_reconstruction': True, 'l_smooth': False, 'l_smooth_weight': 2,
                     'l_pose_consist': False, 'num_source_imgs': S, 'plotting': False}
        DO = ref["optimizer"].DepthOptimizer
        for tag, upd in (("default", {}), ("noargmin", {'diff_img_argmin': False}), ("noauto", {'automasking': False}),
                         ("noinv", {'l_inverse_reconstruction': False}), ("nodc", {'l_depth_consist': False}),
                         ("smooth", {'l_smooth': True}), ("posec", {'l_pose_consist': True}), ("noinit", {'l_depth_init': False})):
            o = object.__new__(DO)
            o.options = dict(base_opts, **upd); o.ssim_loss = losses.SSIM_Loss()
            disp = T(synth.depth_to_sigmoid_disp(np.stack(dts)), dt)[:, None]
            o.target_disparity = (disp * 0.97 + 0.004).clone()
            loss = DO.compute_optimization_loss(o, 0, 0, target, disp, outputs['fwd'], outputs['inv'])
            g4[f"it{iters}_loss_{tag}"] = np.array(loss.reshape(-1)[0].item())
        g4["loss_disp"] = N(disp); g4["loss_disp0"] = N(o.target_disparity)
    out["batch24x40"] = g4

    # ------------------------------------------------------------------ G13: the window loss and its autograd gradients (S = 2)
    # compute_optimization_loss (optimizer.py:29-134) on the outputs of solve_pose_iteratively (train_mono.py:41-120) with a stand-in
    # PoseNet whose output IS a leaf tensor: loss.backward() gives d loss / d pose of all 2 S B directed pairs, and -- the depth
    # maps being leaves too -- d loss / d depth of the SHARED target depth (it feeds every forward pair, train_mono.py:56) and of
    # the source depths.  Variants isolate the terms: forward term alone (:47-69), + inverse (:75-81), + depth consistency (:83-86),
    # and the same without the min over the sources (:71-73).
    class LeafPose(nn.Module):
        def __init__(self, first):
            super().__init__(); self.first = first
        def forward(self, x):
            return self.first

    for tagsz, (B13, S13, H13, W13) in (("24x40", (2, 2, 24, 40)), ("48x160", (1, 2, 48, 160))):
        reset_grid()
        tg, srcs, dts, dss, Ks, pgt = [], [[] for _ in range(S13)], [], [[] for _ in range(S13)], [], [[] for _ in range(S13)]
        for b in range(B13):
            for si in range(S13):
                sign = 1.0 if si == 0 else -1.0
                base = np.array([0.003, -0.002, 0.033, 0.002, -0.004, 0.0015]) * sign
                p = synth.make_pair(H13, W13, seed=130 + b, pose_gt=base, dtype=np.float64)
                if si == 0:
                    tg.append(p["tgt"]); dts.append(p["depth_t"]); Ks.append(p["K"])
                # the source depth maps are made mildly inconsistent with the geometry so that the depth-consistency weights
                # (train_mono.py:91-92) are non-trivial and differ between the sources
                srcs[si].append(p["src"]); dss[si].append(p["depth_s"] * (1.0 + 0.06 * (si + 1) * np.sin(np.arange(W13) / (5.0 + 2 * si))[None, :])); pgt[si].append(p["pose_gt"])
        rng13 = np.random.default_rng(13)
        gt_f = np.concatenate([np.stack(x) for x in pgt])
        first = np.concatenate([gt_f, -gt_f]) + rng13.normal(scale=1.5e-3, size=(2 * S13 * B13, 6))
        g13 = dict(target=np.stack(tg), sources=np.stack([np.stack(x) for x in srcs]), depth_t=np.stack(dts)[:, None],
                   depth_s=np.stack([np.stack(x) for x in dss])[:, :, None], K=np.stack(Ks), first=first)
        dt = torch.float64
        base_opts = {'epochs': 20, 'diff_img_argmin': True, 'automasking': True, 'l_depth_consist': False,
                     'l_depth_consist_weight': 0.15, 'l_depth_init': False, 'l_depth_init_weight': 0.1,
                     'l_inverse_reconstruction': False, 'l_smooth': False, 'l_smooth_weight': 2,
                     'l_pose_consist': False, 'num_source_imgs': S13, 'plotting': False}
        DO = ref["optimizer"].DepthOptimizer
        for tag, upd in (("fwd", {}), ("fwd_inv", {'l_inverse_reconstruction': True}),
                         ("full", {'l_inverse_reconstruction': True, 'l_depth_consist': True}),
                         ("noargmin_full", {'diff_img_argmin': False, 'l_inverse_reconstruction': True, 'l_depth_consist': True}),
                         ("noauto_fwd", {'automasking': False}), ("noargmin_fwd", {'diff_img_argmin': False}),
                         # round 4: + l_pose_consist, 0.1 (poses + poses_inv).abs().mean() (optimizer.py:95-96; off by default in the reference)
                         ("full_pc", {'l_inverse_reconstruction': True, 'l_depth_consist': True, 'l_pose_consist': True})):
            fp = T(first, dt).clone().requires_grad_()
            d_t = T(g13["depth_t"], dt).clone().requires_grad_()
            d_s = [T(g13["depth_s"][i], dt).clone().requires_grad_() for i in range(S13)]
            poses, poses_inv, outputs = ref["train_mono"].solve_pose_iteratively(1, [d_t] + d_s, LeafPose(fp), T(g13["target"], dt),
                                                                                [T(g13["sources"][i], dt) for i in range(S13)], T(g13["K"], dt), return_errors=True)
            o = object.__new__(DO)
            o.options = dict(base_opts, **upd); o.ssim_loss = losses.SSIM_Loss()
            loss = DO.compute_optimization_loss(o, 0, 0, T(g13["target"], dt), None, outputs['fwd'], outputs['inv']).reshape(-1)[0]
            loss.backward()
            g13[f"{tag}_loss"] = np.array(loss.item())
            g13[f"{tag}_grad_pose"] = N(fp.grad)
            if tag in ("fwd", "full", "noargmin_fwd"):
                g13[f"{tag}_grad_depth_t"] = N(d_t.grad[:, 0])
                g13[f"{tag}_grad_depth_s"] = np.stack([N(x.grad[:, 0]) for x in d_s])
            if tag == "fwd":
                # the same forward term with every pixel weighted by the map of the source that WON it (instead of source 0's,
                # optimizer.py:69) -- composed from the reference's own maps (diff_img, valid_mask, weight_mask, auto_mask_error of
                # solve_pose_iteratively, train_mono.py:82-100) exactly as optimizer.py:47-69 composes its version: the cost of the
                # joint dense mode, whose weight terms must not reach across sources
                fp2 = T(first, dt).clone().requires_grad_()
                d_t2 = T(g13["depth_t"], dt).clone().requires_grad_()
                d_s2 = [T(g13["depth_s"][i], dt).clone() for i in range(S13)]
                _, _, out2 = ref["train_mono"].solve_pose_iteratively(1, [d_t2] + d_s2, LeafPose(fp2), T(g13["target"], dt),
                                                                      [T(g13["sources"][i], dt) for i in range(S13)], T(g13["K"], dt), return_errors=True)
                f = out2['fwd']
                stack = lambda k: torch.cat([f[k][i * B13:(i + 1) * B13] for i in range(S13)], 1)          # [B, S, H, W]
                dmin, idx = torch.min(stack('diff_img'), 1, keepdim=True)
                keep = stack('valid_mask').sum(1, keepdim=True).clamp(0, 1) * (dmin < torch.min(stack('auto_mask_error'), 1, keepdim=True)[0]).float()
                w_own = torch.gather(stack('weight_mask'), 1, idx)
                loss2 = (dmin * keep * w_own).sum() / keep.sum()
                loss2.backward()
                g13["fwd_ownw_loss"] = np.array(loss2.item()); g13["fwd_ownw_grad_pose"] = N(fp2.grad); g13["fwd_ownw_grad_depth_t"] = N(d_t2.grad[:, 0])
            if tag == "full":
                for d in ("fwd", "inv"):
                    for k in ("diff_img", "valid_mask", "weight_mask", "auto_mask_error"):
                        g13[f"{d}_{k}"] = N(outputs[d][k][:, 0])
        # round 4: the COMPLETE default loss of optimize_depth_pred -- forward + inverse + depth consistency + l_depth_init, the SSIM
        # prior between the target's current sigmoid disparity and its initial one (optimizer.py:89-90) -- with the target's sigmoid
        # disparity as the leaf (depth = the reference's disp_to_depth of it, utils/learning_helpers.py:77-86; min / max depth as
        # run_mono_exps_kitti.sh:5): d loss / d pose of every directed pair and d loss / d sigma_target through every term
        MIN_D, MAX_D = 0.06, 2.67
        r_d = 1.0 / MIN_D - 1.0 / MAX_D
        sig_np = (1.0 / g13["depth_t"] - 1.0 / MAX_D) / r_d                                    # [B,1,H,W], inside (0, 1) for these scenes
        assert sig_np.min() > 0.0 and sig_np.max() < 1.0
        yy, xx = np.mgrid[0:H13, 0:W13]
        sig0_np = sig_np * (1.0 + 0.04 * np.cos(xx / 3.0 + 0.7 * yy)[None, None] + 0.02 * np.sin(yy / 2.0)[None, None])   # the "initial" map
        g13["sig_t"] = sig_np[:, 0]; g13["sig_t0"] = sig0_np[:, 0]; g13["min_max_depth"] = np.array([MIN_D, MAX_D])
        for tag, upd in (("fullinit", {'l_inverse_reconstruction': True, 'l_depth_consist': True, 'l_depth_init': True}),
                         ("fwdinit", {'l_depth_init': True}),
                         # round 4: + l_smooth, l_smooth_weight (2) x get_smooth_loss(target disparity, target image) (optimizer.py:92-93; off by default)
                         ("fullinit_smooth", {'l_inverse_reconstruction': True, 'l_depth_consist': True, 'l_depth_init': True, 'l_smooth': True})):
            fp = T(first, dt).clone().requires_grad_()
            sig = T(sig_np, dt).clone().requires_grad_()
            d_t = ref["learning_helpers"].disp_to_depth(sig, MIN_D, MAX_D)[1]
            d_s = [T(g13["depth_s"][i], dt).clone() for i in range(S13)]
            _, _, outputs = ref["train_mono"].solve_pose_iteratively(1, [d_t] + d_s, LeafPose(fp), T(g13["target"], dt),
                                                                    [T(g13["sources"][i], dt) for i in range(S13)], T(g13["K"], dt), return_errors=True)
            o = object.__new__(DO)
            o.options = dict(base_opts, **upd); o.ssim_loss = losses.SSIM_Loss()
            o.target_disparity = T(sig0_np, dt)
            loss = DO.compute_optimization_loss(o, 0, 0, T(g13["target"], dt), sig, outputs['fwd'], outputs['inv']).reshape(-1)[0]
            loss.backward()
            g13[f"{tag}_loss"] = np.array(loss.item()); g13[f"{tag}_grad_pose"] = N(fp.grad); g13[f"{tag}_grad_sig_t"] = N(sig.grad[:, 0])
            g13[f"{tag}_init_term"] = np.array((0.1 * losses.SSIM_Loss()(T(sig_np, dt), T(sig0_np, dt)).mean()).item())
        # round 4 (b): the reference's own PARAMETRISATION of optimize_depth_pred (optimizer.py:194-198, 235-239): the leaf is the
        # QUARTER-resolution sigmoid disparity of every frame (target + sources: F.interpolate of the concatenated maps to (H/4, W/4),
        # bilinear), every epoch upsamples it x4 (bilinear) and converts with disp_to_depth.  Recorded: the quarter-resolution maps, their
        # upsampled versions (pins the engine's / the oracle's interpolation weights on torch's), the complete default loss at them and its
        # autograd gradients w.r.t. the poses and w.r.t. the quarter-resolution maps.
        F = torch.nn.functional
        sig_s_np = (1.0 / g13["depth_s"] - 1.0 / MAX_D) / r_d                                  # [S,B,1,H,W]
        assert sig_s_np.min() > 0.0 and sig_s_np.max() < 1.0
        full = torch.cat([T(sig_np, dt)] + [T(sig_s_np[i], dt) for i in range(S13)], 1)        # [B, S+1, H, W]
        quarter = F.interpolate(full, (int(full.shape[2] / 4), int(full.shape[3] / 4)), mode='bilinear').clone().detach()
        quarter = quarter.requires_grad_()
        up = F.interpolate(quarter, (int(quarter.shape[2] * 4), int(quarter.shape[3] * 4)), mode='bilinear')
        dl = [up[:, i:i + 1] for i in range(0, quarter.shape[1])]
        depths_q = [ref["learning_helpers"].disp_to_depth(d, MIN_D, MAX_D)[1] for d in dl]
        fp = T(first, dt).clone().requires_grad_()
        _, _, outputs = ref["train_mono"].solve_pose_iteratively(1, depths_q, LeafPose(fp), T(g13["target"], dt),
                                                                [T(g13["sources"][i], dt) for i in range(S13)], T(g13["K"], dt), return_errors=True)
        o = object.__new__(DO)
        o.options = dict(base_opts, **{'l_inverse_reconstruction': True, 'l_depth_consist': True, 'l_depth_init': True}); o.ssim_loss = losses.SSIM_Loss()
        o.target_disparity = T(sig0_np, dt)
        loss = DO.compute_optimization_loss(o, 0, 0, T(g13["target"], dt), dl[0], outputs['fwd'], outputs['inv']).reshape(-1)[0]
        loss.backward()
        g13["q_sig"] = N(quarter.detach()); g13["q_up"] = N(up.detach())
        g13["qinit_loss"] = np.array(loss.item()); g13["qinit_grad_pose"] = N(fp.grad); g13["qinit_grad_q"] = N(quarter.grad)
        out[f"winloss{tagsz}"] = g13

    # ------------------------------------------------------------------ G7: loss-surface sweeps (f32, as the reference runs it)
    H, W, seed = 48, 160, 5
    p = synth.make_pair(H, W, seed=seed, dtype=np.float32)
    pose = synth.perturb_pose(p["pose_gt"], seed)
    dt = torch.float32
    reset_grid()
    data = (T(p["tgt"], dt)[None], [T(p["src"], dt)[None]], None, None, None, T(p["K"], dt)[None], None, None, None, None, None)
    depths = [T(p["depth_t"], dt)[None, None], T(p["depth_s"], dt)[None, None]]
    rs = ref["plot_loss_surface"].generate_loss_surface(data, depths, T(pose, dt)[None].clone(), sample_trans=True, sample_yaw=True)
    out["sweep48x160"] = dict(tgt=p["tgt"], src=p["src"], depth_t=p["depth_t"], depth_s=p["depth_s"], K=p["K"], pose=pose,
                              delta_list=rs["delta_list"], delta_list_yaw=rs["delta_list_yaw"],
                              errors=rs["reconstruction_errors"], errors_yaw=rs["reconstruction_errors_yaw"],
                              original_error=np.array(rs["original_error"]), best_trans_delta=np.array(rs["best_trans_delta"]),
                              best_yaw_delta=np.array(rs["best_yaw_delta"]))

    # ------------------------------------------------------------------ G8: small helpers
    rng = np.random.default_rng(9)
    disp = rng.uniform(0.02, 0.95, size=(2, 1, 6, 10))
    sd, dep = lh.disp_to_depth(T(disp, torch.float64), 0.06, 2.67)
    l, r_ = rng.uniform(0.1, 1, size=(2, 6, 10)), rng.uniform(0.1, 1, size=(2, 6, 10))
    lst = [T(rng.normal(size=(4, 6)), torch.float32) for _ in range(7)]
    # a3: pose_vec2mat / euler2mat on random 6-vectors, incl. large angles (stn.py:81-116,143-158)
    vec = rng.normal(size=(12, 6)) * np.array([0.1, 0.1, 0.3, 0.02, 0.05, 0.02]) * np.where(np.arange(12)[:, None] < 8, 1.0, 40.0)
    p2m = N(stn.pose_vec2mat(T(vec, torch.float64)))
    out["helpers"] = dict(pose_vec=vec, pose_mat=p2m, disp=disp, scaled_disp=N(sd), depth=N(dep), l_disp=l, r_disp=r_,
                          post=lh.batch_post_process_disparity(l, r_), avg_list=np.stack([N(x) for x in lst]),
                          avg5=N(helpers.avg_final_predictions(lst, 5)))

    # ------------------------------------------------------------------ G10: DNet ScaleRecovery
    # get_ground_mask hard-codes .cuda() (dnet_layers.py:298); on this GPU-less box the call is made a no-op for the
    # duration of the run -- the arithmetic is untouched.
    from models import dnet_layers
    B, H, W = 2, 48, 160
    ps = [synth.make_pair(H, W, seed=60 + b, dtype=np.float32) for b in range(B)]
    depth = np.stack([p["depth_t"] for p in ps]) * np.array([1.0, 1.3], dtype=np.float32)[:, None, None]
    Kb = np.stack([p["K"] for p in ps])
    K4 = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1)); K4[:, :3, :3] = Kb
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        g10 = dict(depth=depth, K=Kb, cam_height=np.array(1.65 / 30.0))
        for dtn, dt in (("f32", torch.float32), ("f64", torch.float64)):
            sr = dnet_layers.ScaleRecovery(B, H, W)
            if dt == torch.float64:
                sr = sr.double()
            d_t, K_t = T(depth, dt)[:, None], T(K4, dt)
            inv_K = torch.inverse(K_t)
            cam = sr.backproject_depth(d_t, inv_K)
            nrm = sr.get_surface_normal(cam)
            gm = sr.get_ground_mask(cam, nrm)
            hts = (cam[:, :-1] * nrm).sum(1).abs()
            g10[f"{dtn}_height"] = N(hts); g10[f"{dtn}_mask"] = N(gm[:, 0].float())
            g10[f"{dtn}_scale"] = N(sr(d_t, K_t, 1.65 / 30.0))
            g10[f"{dtn}_median"] = N(torch.median(torch.masked_select(hts.unsqueeze(1), gm)))
            # a short batch is padded with copies of image 0 up to the constructor's batch size (dnet_layers.py:307-311)
            sr5 = dnet_layers.ScaleRecovery(5, H, W)
            if dt == torch.float64:
                sr5 = sr5.double()
            g10[f"{dtn}_scale_pad5"] = N(sr5(d_t, K_t, 1.65 / 30.0))
    finally:
        torch.Tensor.cuda = _cuda
    out["scale48x160"] = g10

    # ------------------------------------------------------------------ G9: optimize_window result dict (schema for the shim)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import standins
    B, S, H, W, ITER = 2, 2, 48, 160, 3
    reset_grid()
    w = standins.make_window(B, S, H, W)
    pose_model, depth_model = standins.window_models(w, ITER)
    options = {'epochs': 5, 'lr': 4e-3, 'optimizer': 'adam', 'optimize_depth_weights_bottleneck_beyond': False,
               'optimize_depth_weights_all': False, 'optimize_depth_encoder': False, 'optimize_pose_weights_all': False,
               'optimize_depth_pred': True, 'optimize_depth_bottleneck_values': False, 'diff_img_argmin': True,
               'automasking': True, 'mode': 'scaled', 'l_depth_consist': True, 'l_depth_consist_weight': 0.15,
               'l_depth_init': True, 'l_depth_init_weight': 0.1, 'l_inverse_reconstruction': True, 'l_smooth': False,
               'l_smooth_weight': 2, 'l_pose_consist': False, 'avg_final_epochs': 5, 'num_source_imgs': S, 'plotting': False}
    config = {'minibatch': B, 'device': 'cpu', 'min_depth': 0.06, 'max_depth': 2.67, 'iterations': ITER,
              'camera_height': 1.65, 'flow_type': 'none'}
    do = ref["optimizer"].DepthOptimizer(options, config, pose_model, depth_model, "09_02")
    res = do.optimize_window(0, standins.loader_batch(w))
    g9 = {f"in_{k}": v for k, v in w.items()}
    g9["in_iterations"] = np.array(ITER)
    schema = []
    for k in sorted(res):
        v = res[k]
        if isinstance(v, (list, tuple)):
            schema.append(f"{k}|list{len(v)}|{tuple(v[0].shape)}|{str(v[0].dtype)}|{v[0].device.type}")
            g9[f"out_{k}"] = np.stack([N(x) for x in v])
        elif isinstance(v, np.ndarray):
            schema.append(f"{k}|ndarray|{v.shape}|{v.dtype}|host"); g9[f"out_{k}"] = v
        else:
            schema.append(f"{k}|tensor|{tuple(v.shape)}|{str(v.dtype)}|{v.device.type}"); g9[f"out_{k}"] = N(v)
    g9["schema"] = np.array(schema)
    out["window48x160"] = g9

    # ------------------------------------------------------------------ G11: the reference's compute_trajectory on the stand-ins
    from tightly_coupled_sfm_amd.liegroups import SE3 as MySE3
    from tightly_coupled_sfm_amd.trajectory import TrajectoryMetrics as MyTM
    val = ref["validate"]
    val.SE3, val.TrajectoryMetrics = MySE3, MyTM
    rng = np.random.default_rng(11)
    rel = np.array([0.0, 0.0, 1.0, 0.0, 0.01, 0.0]) + 0.02 * rng.normal(size=(400, 6)) * np.array([1, 1, 1, 0.3, 0.3, 0.3])
    gt = [np.eye(4)]
    for q in rel:
        gt.append(MySE3.exp(q).dot(MySE3.from_matrix(gt[-1]).inv()).inv().as_matrix())
    noisy = rel + 0.004 * rng.normal(size=rel.shape) * np.array([1, 1, 1, 0.2, 0.2, 0.2])
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        est, gt_out, errors, cum = val.compute_trajectory(noisy, np.array(gt), method="odom", compute_seg_err=True)
    out["traj400"] = dict(pose_vec=noisy, gt_traj=np.array(gt), est_traj=est, errors=np.array(errors, dtype=np.float64), cum_dist=cum)

    # ------------------------------------------------------------------ G12: the reference's PoseNet and coupled pose loop
    # models/pose_models.py:88-147 with seeded parameters (tests/standins.posenet_params: the fixture carries the seed, not 6 MB of
    # weights), float32 as the reference runs it, on (a) 4 six-channel inputs of 48x160 and (b) the fwd + inv pair of a 192x640
    # window; then solve_pose_iteratively (train_mono.py:41-81) with that network, B=2 targets x S=2 sources, 4 iterations
    import importlib
    pm_mod = importlib.import_module("models.pose_models")
    sdp = standins.posenet_params(0)
    net = pm_mod.pose_model({'flow_type': 'none'})
    net.load_state_dict({k: torch.tensor(v) for k, v in sdp.items()})
    net.eval()
    g12 = {"seed": np.array(0)}
    with torch.no_grad():
        for tag, (hh, ww, nimg) in (("a", (48, 160, 4)), ("b", (192, 640, 2))):
            bb = synth.make_batch(nimg, hh, ww, seed0=40, both_directions=True)
            x = torch.tensor(np.concatenate([bb["tgt"], bb["src"]], 1))
            pose, feats = net(x, return_features=True)
            g12[f"{tag}_pose"] = N(pose)
            g12[f"{tag}_in_checksum"] = np.array([float(x.double().sum()), float(x.double().abs().max())])
            for i, f in enumerate(feats):      # per layer: mean, mean |.|, and a strided sample of the activated features
                g12[f"{tag}_feat{i + 1}_stats"] = np.array([float(f.double().mean()), float(f.double().abs().mean())])
            g12[f"{tag}_feat1_sub"] = N(feats[0][:, :, ::9, ::13]); g12[f"{tag}_feat4_sub"] = N(feats[3][:, ::8])
            g12[f"{tag}_feat7"] = N(feats[6])
        w12 = standins.make_window(2, 2, 48, 160, seed0=90)
        # input depths from the REFERENCE's own disp_to_depth (utils/learning_helpers.py:77-86; float64, then rounded to float32 as the
        # tests form them) -- a golden generator does not import the thing it pins (the oracle stood here)
        _d2d = lambda a: ref["learning_helpers"].disp_to_depth(torch.tensor(np.asarray(a, dtype=np.float64)), 0.06, 2.67)[1].to(torch.float32)
        dts = [_d2d(w12["disp_t"])] + [_d2d(w12["disp_s"][i]) for i in range(2)]
        reset_grid()
        poses, poses_inv, outs = ref["train_mono"].solve_pose_iteratively(4, dts, net, torch.tensor(w12["target"]), [torch.tensor(w12["sources"][i]) for i in range(2)],
                                                                        torch.tensor(w12["K"]), return_errors=True)
        g12["loop_stacked"] = np.concatenate([N(outs["fwd"]["poses"]), N(outs["inv"]["poses"])])      # [2SB, 4, 6]
        g12["loop_poses"] = np.concatenate([np.concatenate([N(p) for p in poses]), np.concatenate([N(p) for p in poses_inv])])
    out["posenet"] = g12

    # ------------------------------------------------------------------ full-size summary (192x640, f32 as run by the reference)
    H, W, seed = 192, 640, 0
    p = synth.make_pair(H, W, seed=seed, dtype=np.float32)
    pose = synth.perturb_pose(p["pose_gt"], seed)
    full = dict(pose=pose, pose_gt=p["pose_gt"], K=p["K"],
                in_checksum=np.array([p["tgt"].astype(np.float64).sum(), p["src"].astype(np.float64).sum(),
                                      p["depth_t"].astype(np.float64).sum(), p["depth_s"].astype(np.float64).sum()]))
    for dtn, dt in (("f64", torch.float64), ("f32", torch.float32)):
        reset_grid()
        t, s = T(p["tgt"], dt)[None], T(p["src"], dt)[None]
        dpt, dps = T(p["depth_t"], dt)[None, None], T(p["depth_s"], dt)[None, None]
        K = T(p["K"], dt)[None]
        po = T(pose, dt)[None].clone().requires_grad_()
        r = helpers.compute_photometric_error(t, s, dpt, dps, po, K)
        L = (r["diff_img"] * r["valid_mask"] * r["weight_mask"]).sum() / r["valid_mask"].sum()
        L.backward()
        full[f"{dtn}_cost"] = np.array(L.item()); full[f"{dtn}_grad_pose"] = N(po.grad[0])
        full[f"{dtn}_n_mask"] = np.array(r["valid_mask"].sum().item())
        full[f"{dtn}_sum_diff"] = np.array(r["diff_img"].double().sum().item())
        full[f"{dtn}_sum_weight"] = np.array(r["weight_mask"].double().sum().item())
        full[f"{dtn}_diff_sub"] = N(r["diff_img"][0, 0, ::7, ::7]); full[f"{dtn}_weight_sub"] = N(r["weight_mask"][0, 0, ::7, ::7])
        full[f"{dtn}_mask_sub"] = N(r["valid_mask"][0, 0, ::7, ::7]); full[f"{dtn}_rec_sub"] = N(r["img_rec"][0, :, ::7, ::7])
    out["full192x640"] = full

    only = [a for a in sys.argv[1:] if not a.startswith("-")]      # `make_golden.py winloss24x40 ...`: write only these fixtures
    for name, d in out.items():
        if only and name not in only:
            continue
        path = os.path.join(HERE, f"golden_{name}.npz")
        np.savez_compressed(path, **d)
        print(f"{name:14s} {os.path.getsize(path) / 1024:8.1f} KiB  keys={len(d)}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
