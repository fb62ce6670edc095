"""Drop-in for the reference's test-time optimiser call surface.

    from tightly_coupled_sfm_amd.optimizer import DepthOptimizer      # instead of `from optimizer import DepthOptimizer`
    opt = DepthOptimizer(options, config, pose_model, depth_model, seq)   # optimizer.py:15-27
    results = opt.optimize_window(img_idx, data)                          # optimizer.py:136-297

Same constructor, same `optimize_window(img_idx, data)` signature, same result-dict keys / shapes / CPU placement
(run_sequential_optimization.py:195-216, run_sample_optimization_demo.py:178-186).  What differs is WHAT is
optimised: the reference runs Adam over depth-network weights with poses re-predicted by PoseNet; this engine keeps
the networks frozen, takes PoseNet's coupled estimate (solve_pose_iteratively, train_mono.py:41-120, with the warp
done by the HIP library) as the initial pose of every directed pair and refines it with `gn_iters` Gauss-Newton /
LM iterations on the reference's own residual (libtcsfm_hip.so).  New option keys (all optional):
    solver 'gn'|'lm', gn_iters (4), refine 'pose'|'pose+scale'|'pose+depth', lambda0, param 'se3'|'euler',
    (the reference's own `diff_img_argmin`, `automasking`, `l_depth_consist(+_weight)`, `mode` keys are honoured),
    prior_depth, lambda_depth (dense mode).
The reference's `optimize_depth_pred` switch (Adam on the disparity maps themselves, optimizer.py:194-198) selects
refine='pose+depth' unless `refine` is given: pose + per-pixel inverse depth of the target by Gauss-Newton with a Schur complement,
by default ON THE REFERENCE'S OWN LOSS (optimizer.py:47-90: forward term with source 0's weight map, 0.25 x inverse term, depth
consistency, l_depth_init SSIM prior; `diff_img_argmin`, `automasking`, `l_depth_consist(+_weight)`, `l_depth_init(+_weight)` honoured;
include/tcsfm.h "REFERENCE LOSS"); options['window_rule'] = 'pair' or solver 'lm' select the library's own joint / per-pair dense modes.
Its weight-tuning switches (optimize_depth_encoder, ...) need autograd through the networks, which is out of scope:
they are ignored with a warning, or refused when options['strict_legacy'] is set.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch

from . import _lib
from .engine import Engine, default_opts

_LEGACY = ("optimize_depth_weights_bottleneck_beyond", "optimize_depth_weights_all", "optimize_depth_encoder",
           "optimize_pose_weights_all", "optimize_depth_bottleneck_values")


def process_sample_batch(data, config):
    """data/kitti_loader.py:60-98: DataLoader batch -> the 11-tuple on config['device']."""
    device = config["device"]
    target_img, source_imgs, lie_alg, intrinsics, flow_imgs = data
    target_img_aug = target_img["color_aug_left"].to(device)
    target = target_img["color_left"].to(device)
    lie_aug, lie = lie_alg["color_aug"], lie_alg["color"]
    src, src_aug, gt, vo, gt_aug, vo_aug = [], [], [], [], [], []
    for i, im in enumerate(source_imgs["color_aug_left"]):
        src_aug.append(im.to(device)); src.append(source_imgs["color_left"][i].to(device))
        gt.append(lie[i][0].float().to(device)); vo.append(lie[i][1].float().to(device))
        gt_aug.append(lie_aug[i][0].float().to(device)); vo_aug.append(lie_aug[i][1].float().to(device))
    flows = [[None for _ in src] for _ in range(2)]
    K_aug = intrinsics["color_aug_left"].float().to(device)[:, 0, :, :]
    K = intrinsics["color_left"].float().to(device)[:, 0, :, :]
    return target, src, gt, vo, flows, K, target_img_aug, src_aug, gt_aug, vo_aug, K_aug


_RAMPS = {}


def _flip_ramp(w):
    """blend weight of the flipped prediction along the image width: 1 on the left 5 %, linear down to 0 at 10 %, 0 beyond"""
    if w not in _RAMPS:
        x = np.arange(w, dtype=np.float64) / max(w - 1, 1)
        _RAMPS[w] = np.clip(1.0 - 20.0 * (x - 0.05), 0.0, 1.0)
    return _RAMPS[w]


def batch_post_process_disparity(l_disp, r_disp):
    """Flip post-processing of Monodepth as the reference applies it (utils/learning_helpers.py:115-123): a prediction and the
    un-flipped prediction of the mirrored image are averaged, except near the left / right border where only the one that saw
    the scene content beyond that border is kept.  [B,H,W] arrays in, [B,H,W] out."""
    left = _flip_ramp(l_disp.shape[-1])           # weight of r_disp
    right = left[::-1]                            # weight of l_disp
    out = 0.5 * (1.0 - left - right) * (l_disp + r_disp)
    out += right * l_disp
    out += left * r_disp
    return out


def avg_final_predictions(pred_list, num):
    """optimization_experiments/helpers.py:25-33."""
    pred_list = pred_list[-num:]
    acc = pred_list[0] * 0
    for p in pred_list:
        acc = acc + p
    return acc / len(pred_list)


class DepthOptimizer:
    def __init__(self, options, config, pose_model, depth_model, seq):
        self.options, self.config, self.seq = options, config, seq
        self.pose_model = pose_model.train(False).eval()
        self.depth_model = depth_model.train(False).eval()
        legacy = [k for k in _LEGACY if options.get(k, False)]
        if legacy:
            msg = (f"options {legacy} tune network weights/activations through autograd; the HIP engine refines the "
                   "pose (and depth scale) of each frame pair by Gauss-Newton instead")
            if options.get("strict_legacy", False):
                raise NotImplementedError(msg)
            warnings.warn(msg)
        # l_pose_consist IS a term of the pose mode (opts.w_pose_consist, round 4) and of the pose + depth mode (round 5) under the window rule
        # REFERENCE; elsewhere it is ignored
        pc_ok = (options.get("window_rule", "reference") != "pair" and options.get("solver", "gn") != "lm" and
                 options.get("refine", "pose+depth" if options.get("optimize_depth_pred", False) else "pose") in ("pose", "pose+depth") and options.get("param", "se3") == "se3")
        # l_smooth IS a term of the pose + depth mode on the reference's loss (opts.w_smooth, round 4)
        sm_ok = (options.get("refine", "pose+depth" if options.get("optimize_depth_pred", False) else "pose") == "pose+depth" and
                 options.get("window_rule", "reference") != "pair" and options.get("solver", "gn") != "lm")
        ignored = [k for k in ("l_smooth", "l_pose_consist") if options.get(k, False) and not ((k == "l_pose_consist" and pc_ok) or (k == "l_smooth" and sm_ok))]
        if ignored:   # off by default in the reference (run_sequential_optimization.py:87,89); not part of the per-pair GN cost
            warnings.warn(f"options {ignored} are not terms of the Gauss-Newton cost and are ignored "
                          "(losses.get_smooth_loss / compute_optimization_loss still evaluate them for logging)")
        self._engine = None
        self.full_results = []
        # measurement hook (bench.py `shim`): with time_engine the refine call inside optimize_window is bracketed by device synchronisations
        # and its wall time left in last_engine_call_us (off by default: the synchronisations are not free)
        self.time_engine = False
        self.last_engine_call_us = None

    # -- engine / options ------------------------------------------------------------------------------------
    def _eng(self, H, W, npairs):
        if self._engine is None or (self._engine.H, self._engine.W) != (H, W) or self._engine.max_pairs < npairs:
            self._engine = Engine(H, W, npairs)
        return self._engine

    def _refine_mode(self):
        o = self.options
        return o.get("refine", "pose+depth" if o.get("optimize_depth_pred", False) else "pose")

    def _dense_reference(self):
        """pose + depth by default minimises the reference's own loss (window rule REFERENCE, Gauss-Newton; the kernels are instantiated for
        S <= 3 sources per target: optimize_window checks the ACTUAL number of source images before it calls the engine);
        options['window_rule'] = 'pair' or solver 'lm' select the library's own dense modes"""
        o = self.options
        return o.get("window_rule", "reference") != "pair" and o.get("solver", "gn") != "lm"

    def _opts(self):
        o = self.options
        if self._refine_mode() == "pose+depth":
            kw = {k: float(o[k]) for k in ("prior_depth", "lambda_depth") if k in o}
            if self._dense_reference():
                # the reference's OWN loss (optimizer.py:47-90; round 4, golden G13 `full` / `fullinit`): forward term with source 0's
                # weights, 0.25 x inverse term, depth consistency, l_depth_init -- each switched by the reference's option keys
                return default_opts(n_iters=int(o.get("gn_iters", 4)), automask=1 if o.get("automasking", True) else 0,
                                    w_dc=float(o.get("l_depth_consist_weight", 0.15)) if o.get("l_depth_consist", False) else 0.0,
                                    prior_init=float(o.get("l_depth_init_weight", 0.1)) if o.get("l_depth_init", True) else 0.0,
                                    w_smooth=float(o.get("l_smooth_weight", 2.0)) if o.get("l_smooth", False) else 0.0,      # optimizer.py:92-93
                                    w_pose_consist=0.1 if (o.get("l_pose_consist", False) and o.get("param", "se3") == "se3") else 0.0,      # optimizer.py:95-96
                                    solver=_lib.SOLVER_GN, lambda0=float(o.get("lambda0", 1e-4)), min_depth=float(self.config["min_depth"]),
                                    max_depth=float(self.config["max_depth"]), window_rule=_lib.WINDOW_REFERENCE,
                                    # optimizer.py:194-198: the reference's leaf is ONE tensor with the disparities of the target AND of every
                                    # source; options['optimize_source_depths'] = True makes the source maps unknowns here as well (default:
                                    # only the target's map moves -- the one `l_depth_init` and `disp_opt` are about)
                                    free_source_depths=1 if o.get("optimize_source_depths", False) else 0,
                                    **{k: v for k, v in kw.items() if k == "lambda_depth"})
            # the library's per-pair / joint dense modes: GN on the SE(3) chart, Tikhonov depth prior instead of the DC term
            return default_opts(n_iters=int(o.get("gn_iters", 4)), automask=1 if o.get("automasking", True) else 0, w_dc=0.0,
                                solver=_lib.SOLVER_LM if o.get("solver", "gn") == "lm" else _lib.SOLVER_GN,
                                lambda0=float(o.get("lambda0", 1e-4)), min_depth=float(self.config["min_depth"]),
                                max_depth=float(self.config["max_depth"]), dense_joint=1 if o.get("dense_joint", True) else 0, **kw)
        return default_opts(
            n_iters=int(o.get("gn_iters", 4)),
            solver=_lib.SOLVER_LM if o.get("solver", "gn") == "lm" else _lib.SOLVER_GN,
            param=_lib.PARAM_EULER if o.get("param", "se3") == "euler" else _lib.PARAM_SE3,
            refine=_lib.REFINE_POSE_SCALE if self._refine_mode() == "pose+scale" else _lib.REFINE_POSE,
            automask=1 if o.get("automasking", True) else 0,
            w_dc=float(o.get("l_depth_consist_weight", 0.15)) if o.get("l_depth_consist", False) else 0.0,
            lambda0=float(o.get("lambda0", 1e-4)),
            # this class stands in for the reference's optimiser: by default the scalar it minimises is the reference's
            # compute_optimization_loss itself (optimizer.py:47-86; golden G13); 'pair' = the library's batch-independent default
            window_rule=_lib.WINDOW_PAIR if o.get("window_rule", "reference") == "pair" else _lib.WINDOW_REFERENCE,
            # optimizer.py:95-96: 0.1 (poses + poses_inv).abs().mean(), a term of the 6-DoF Gauss-Newton pose mode under that rule
            w_pose_consist=0.1 if (o.get("l_pose_consist", False) and o.get("window_rule", "reference") != "pair" and o.get("solver", "gn") != "lm"
                                   and self._refine_mode() == "pose" and o.get("param", "se3") == "se3") else 0.0)

    def _disparities(self, imgs):
        """depth net forward in the reference's two call forms (optimizer.py:146-147) or a plain callable"""
        try:
            _, skips = self.depth_model(x=imgs, return_disp=False, epoch=50)
            disparities, _ = self.depth_model(x=None, skips=skips, epoch=50)
        except TypeError:
            disparities, _ = self.depth_model(imgs)
        return disparities[0]

    # -- the reference's coupled pose initialisation, warp done by the HIP library -------------------------------
    def _solve_pose_iteratively(self, eng, num_iter, depths, target_img, source_img_list, intrinsics):
        """train_mono.py:41-81: PoseNet -> warp -> PoseNet correction, (iters-1) times, fwd and inv pairs stacked."""
        S, B = len(source_img_list), target_img.shape[0]
        target_depths = depths[0].repeat(S, 1, 1, 1)
        source_depths = torch.cat(depths[1:], 0)
        source_imgs = torch.cat(source_img_list, 0)
        K = intrinsics.repeat(2 * S, 1, 1)
        target_imgs = target_img.repeat(S, 1, 1, 1)
        imgs = torch.cat([torch.cat([target_imgs, source_imgs], 1), torch.cat([source_imgs, target_imgs], 1)], 0)
        d_t = torch.cat([target_depths, source_depths], 0).contiguous()
        d_s = torch.cat([source_depths, target_depths], 0).contiguous()
        from .train_mono import _library_posenet
        net = _library_posenet(self.pose_model, eng, 2 * S * B)
        if net is not None:     # a PoseNet with the reference's parameters: network, warps and corrections all inside the library
            full, stacked = net.solve_pose_iteratively(num_iter, target_img, list(source_img_list), depths[0], list(depths[1:]), intrinsics)
            return full, stacked, imgs, d_t, d_s, K
        full = self.pose_model(imgs)
        stacked = [full.clone()]
        tgt, src = imgs[:, 0:3].contiguous(), imgs[:, 3:6].contiguous()
        for _ in range(num_iter - 1):
            # (tgt * valid | img_rec) written by the warp kernel itself: no clone / mask / copy round trips (8f row 4)
            new = eng.posenet_input(tgt, src, d_t, d_s, full[:, :6].contiguous(), K.contiguous())
            full = full + self.pose_model(new)
            stacked.append(full.clone())
        return full[:, :6].contiguous(), torch.stack(stacked, 1), imgs, d_t, d_s, K

    # -- the call the drivers make ------------------------------------------------------------------------------
    def _to_host(self, named):
        """ONE device-to-host copy for all the small outputs of a window (round 5: the per-tensor `.cpu()` calls were 17 synchronisations per
        window): the float32 tensors are flattened into one device buffer, copied once into a pinned staging buffer, and handed back as CPU
        tensors that own their storage (the staging buffer is reused by the next window)."""
        keys = [k for k, v in named if v is not None]
        flat = torch.cat([v.reshape(-1).float() for k, v in named if v is not None])
        n = flat.numel()
        if getattr(self, "_stage", None) is None or self._stage.numel() < n:
            self._stage = torch.empty(max(n, 4096), dtype=torch.float32).pin_memory()
        self._stage[:n].copy_(flat, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = self._stage[:n].clone()
        out, o = {k: None for k, _ in named}, 0
        for k, v in named:
            if v is None:
                continue
            m = v.numel()
            out[k] = host[o:o + m].reshape(v.shape)
            o += m
        return out

    @torch.no_grad()
    def optimize_window(self, img_idx, data):
        res = {}
        tup = data if len(data) == 11 else process_sample_batch(data, self.config)
        target_img, source_img_list, gt_lie_alg_list, _, _, intrinsics = tup[:6]
        B, _, H, W = target_img.shape
        S = len(source_img_list)
        split = S * B
        dense = self._refine_mode() == "pose+depth"
        if dense and self._dense_reference() and S > 3:      # (the real S, before anything is launched: options['num_source_imgs'] may not match the batch)
            raise ValueError("pose + depth on the reference's loss handles up to 3 source images per target (options['window_rule'] = 'pair' lifts it)")
        eng = self._eng(H, W, 2 * split)
        cfg = self.config

        # ONE depth-network pass for the window's S + 1 frames AND the mirrored targets of the flip-averaged disparity (helpers.py:35-49; the
        # reference runs that second pass after its optimisation because its network weights have moved -- here they are frozen, so the
        # prediction is the same and the pass rides in the first batch); options['batch_flip_pass'] = False keeps two passes
        batch_flip = bool(self.options.get("batch_flip_pass", True))
        frames = [target_img] + list(source_img_list)
        imgs = torch.cat(frames + ([torch.flip(target_img, [3])] if batch_flip else []), 0)
        disp_all = self._disparities(imgs).float().contiguous()
        sd_all, depth_all = eng.disp_to_depth(disp_all, cfg["min_depth"], cfg["max_depth"])     # scaled disparity and depth of every frame: one launch
        depths = [depth_all[i * B:(i + 1) * B] for i in range(S + 1)]

        pose0, stacked0, stack_imgs, d_t, d_s, K = self._solve_pose_iteratively(
            eng, int(cfg.get("iterations", 1)), depths, target_img.float(), [s.float() for s in source_img_list], intrinsics.float())
        res["depths_init"] = [d.clone() for d in depths]
        unscaled = self.options.get("mode", "scaled") == "unscaled"
        # DNet ground-plane rescaling of the INITIAL depths (the reference evaluates it at epoch 0, optimizer.py:254-261)
        sf_init = (eng.scale_recovery(depths[0].contiguous(), intrinsics.float().contiguous(), cfg["camera_height"] / 30.0,
                                      pad_to_batch=int(cfg.get("minibatch", B))) if unscaled else None)

        opts = self._opts()
        if dense and self._dense_reference() and self.options.get("depth_param", "quarter") == "quarter":
            # the reference's own unknown (optimizer.py:194-198, 235-239): the QUARTER-resolution map, upsampled x4 for every evaluation of
            # the loss (golden G13 `qinit`); options['depth_param'] = 'full': one inverse depth per pixel
            if H % 4 == 0 and W % 4 == 0:
                opts.depth_param = _lib.DEPTH_QUARTER
            else:
                warnings.warn("depth_param 'quarter' needs H and W to be multiples of 4: refining the full-resolution map instead")
        if self.time_engine:
            import time as _time
            torch.cuda.synchronize(); _t0 = _time.perf_counter()
        if dense:
            pose, depth_ref, stats = eng.refine_dense_window(
                target_img.float(), [s.float() for s in source_img_list], depths[0].contiguous(), [d.contiguous() for d in depths[1:]],
                intrinsics.float(), pose0, opts, stats=True, argmin=bool(self.options.get("diff_img_argmin", True)))
            log_scale = None
        else:
            # window form: the library forms the fwd / inv pairs itself; per-pixel min over the sources as the reference's loss
            # does when options['diff_img_argmin'] is set (optimizer.py:47-69)
            pose, log_scale, stats = eng.refine_window(
                target_img.float(), [s.float() for s in source_img_list], depths[0].contiguous(), [d.contiguous() for d in depths[1:]],
                intrinsics.float(), pose0, opts, stats=True, argmin=bool(self.options.get("diff_img_argmin", True)))
        if self.time_engine:
            torch.cuda.synchronize(); self.last_engine_call_us = (_time.perf_counter() - _t0) * 1e6
        if not self.options.get("l_inverse_reconstruction", True):
            # the reference then leaves the inverse direction out of its objective (optimizer.py:74-79): the inverse poses stay
            # what the pose network predicted
            pose = torch.cat([pose[:split], pose0[split:]], 0)
            stats = stats.clone(); stats[split:, :, _lib.STAT_POSE:_lib.STAT_POSE + 6] = pose0[split:, None, :]
        traj = stats[:, :, _lib.STAT_POSE:_lib.STAT_POSE + 6].clone()   # [2SB, gn_iters+1, 6]: the iterates (cf. train_mono.py:71-79)
        traj[:, -1] = pose            # LM: the last row is the TRIAL pose even when that step was rejected; report what was returned
        res["stacked_poses_opt"] = traj[:split]
        res["stacked_poses_inv_opt"] = traj[split:]
        if log_scale is not None:
            s = torch.exp(log_scale[:split].reshape(S, B).mean(0)).reshape(B, 1, 1, 1)
            depths = [d * s for d in depths]
        if dense:
            # joint mode (default): the S forward slots hold ONE refined map of the target frame, shared by its S forward pairs
            # (the mean below is then the identity); options['dense_joint'] = False: every forward pair refined its own copy and
            # the copies are fused by averaging inverse depths.  Source frame s was refined by its inverse pair
            # reference-loss mode (default): the inverse slots are the source depths as given (not unknowns there) -- or their refined maps
            # under options['optimize_source_depths']
            inv_t = (1.0 / depth_ref[:split]).reshape(S, B, 1, H, W).mean(0)
            depths = [1.0 / inv_t] + [depth_ref[split + i * B: split + (i + 1) * B] for i in range(S)]
        res["depths_opt"] = depths

        # DNet ground-plane rescaling of the refined depths, optimizer.py:254-256 (self.dgc = ScaleRecovery(minibatch, 192, 640))
        sf = (eng.scale_recovery(depths[0].contiguous(), intrinsics.float().contiguous(), cfg["camera_height"] / 30.0,
                                 pad_to_batch=int(cfg.get("minibatch", B))) if unscaled else None)

        # disparity for depth evaluation: flip-averaged prediction (helpers.py:35-49), blended ON THE DEVICE in float64 with the operations of
        # batch_post_process_disparity in their order (same bits as the NumPy form): one [B,H,W] float64 copy instead of two float32 maps
        if batch_flip:
            sd_t, sd_f = sd_all[:B, 0], sd_all[(S + 1) * B:, 0]
        else:
            flipped = self._disparities(torch.cat((target_img, torch.flip(target_img, [3])), 0)).float().contiguous()
            sd2, _ = eng.disp_to_depth(flipped, cfg["min_depth"], cfg["max_depth"])
            sd_t, sd_f = sd2[:B, 0], sd2[B:, 0]
        key = (W, str(sd_t.device))
        if key not in _RAMPS:
            _RAMPS[key] = torch.as_tensor(_flip_ramp(W), dtype=torch.float64, device=sd_t.device)
        left = _RAMPS[key]
        right = torch.flip(left, [0])
        r32 = torch.flip(sd_f, [2])
        disp_dev = 0.5 * (1.0 - left - right) * (sd_t + r32).double()      # (the float32 sum first, as the NumPy form adds two float32 arrays)
        disp_dev += right * sd_t.double()
        disp_dev += left * r32.double()

        # everything small the drivers read on the CPU: ONE staged copy
        gt = torch.cat(gt_lie_alg_list, 0) if gt_lie_alg_list[0] is not None else None
        hst = self._to_host([("pose0", pose0), ("stacked0", stacked0), ("pose", pose), ("traj", traj), ("cost", stats[:, :, 0]), ("gt", gt),
                             ("log_scale", log_scale), ("sf", sf), ("sf_init", sf_init)])
        res["poses_init"], res["poses_inv_init"] = hst["pose0"][:split], hst["pose0"][split:]
        res["gt_poses"] = hst["gt"]
        res["gt_poses_inv"] = -res["gt_poses"] if res["gt_poses"] is not None else None
        res["stacked_poses_init"], res["stacked_poses_inv_init"] = hst["stacked0"][:split], hst["stacked0"][split:]
        res["poses_opt"], res["poses_inv_opt"] = hst["pose"][:split], hst["pose"][split:]
        res["gn_cost"] = hst["cost"]                   # per pair, per linearisation (extra key)
        if log_scale is not None:
            res["log_depth_scale"] = hst["log_scale"]
        res["scale_factor"] = hst["sf"] if unscaled else torch.FloatTensor([1])
        res["scale_factor_init"] = hst["sf_init"] if unscaled else torch.FloatTensor([1])
        # the reference hands back a float64 CPU tensor here (avg_final_predictions adds numpy arrays into a tensor, G9)
        res["disp_opt"] = disp_dev.cpu()
        # the demo variant of the reference optimiser (optimizer_for_cont_plot.py:27,116,270) keeps one result dict per
        # optimisation step in `full_results`; here: one per Gauss-Newton iterate
        th = hst["traj"]
        self.full_results = [dict(res, poses_opt=th[:split, k], poses_inv_opt=th[split:, k]) for k in range(1, th.shape[1])]
        return res
