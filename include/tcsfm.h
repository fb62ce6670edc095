/*
 * tcsfm.h -- C ABI of libtcsfm_hip.so, the MI355X (gfx950) photometric pose/depth refinement engine.
 *
 * This is the drop-in boundary for the hot path of utiasSTARS/tightly-coupled-SfM
 * (SURVEY.md section 8a/8b).  The reference is pure Python/PyTorch and has no FFI; every entry
 * point below names the reference function (file:line under /root/reference) it replaces.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no torch types.  All array arguments are fp32, contiguous, NCHW like the reference:
 *       images [N,3,H,W], depth maps [N,1,H,W] (== [N,H,W]), intrinsics [N,3,3] row-major, poses [N,6].
 *   - a "pair" is one DIRECTED frame pair (target, source); the reference stacks forward and inverse
 *     pairs along N the same way (train_mono.py:54-62).
 *   - pose 6-vector = the reference's [tx,ty,tz,rx,ry,rz] (models/stn.py:143-158).  The warp uses
 *     pose_vec2mat(-pose) exactly like the reference call sites (train_mono.py:69, helpers.py:11).
 *   - intrinsics must be pinhole [fx 0 cx; 0 fy cy; 0 0 1] (all of the reference's loaders produce
 *     this form); anything else returns TCSFM_E_INTRINSICS: at once for host pointers and for the first use
 *     of a device buffer (one small blocking copy), and -- when the CONTENTS of an already validated device
 *     buffer change behind the library's back -- from a device-side guard: that call's results are NaN and
 *     the error is returned by the next call on the handle or by tcsfm_synchronize().
 *   - every entry point runs on the handle's device and restores the caller's current device on return.
 *   - pointers are DEVICE pointers unless tcsfm_opts.host_ptrs != 0, in which case the library stages
 *     them through its own device buffers (PCIe-inclusive path).
 *   - every call is asynchronous on the handle's HIP stream when given device pointers, except that
 *     host-pointer calls and calls returning host scalars synchronise that stream before returning.
 *   - return value: 0 = ok, negative = error; tcsfm_last_error() gives the message.  Nothing throws.
 *   - one handle = one device + one stream; calls on a handle are not re-entrant.
 */
#ifndef TCSFM_H
#define TCSFM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tcsfm_ctx *tcsfm_handle;

enum {
    TCSFM_OK = 0,
    TCSFM_E_ARG = -1,        /* bad argument (null pointer, size out of range, ...)   */
    TCSFM_E_HIP = -2,        /* a HIP runtime call failed                              */
    TCSFM_E_INTRINSICS = -3, /* non-pinhole intrinsics                                 */
    TCSFM_E_NOMEM = -4
};

enum { TCSFM_SOLVER_GN = 0, TCSFM_SOLVER_LM = 1 };
enum { TCSFM_PARAM_SE3 = 0,    /* T <- exp(delta^) T, delta = [rho, phi] (liegroups ordering, validate.py:65) */
       TCSFM_PARAM_EULER = 1   /* pose <- pose + delta on the reference's [t, euler] vector                    */ };
enum { TCSFM_REFINE_POSE = 0,        /* 6 DoF                                                  */
       TCSFM_REFINE_POSE_SCALE = 1   /* 6 DoF + log depth-scale shared by both depth maps (7x7) */ };

/* How tcsfm_refine_window / tcsfm_linearize_window combine the 2*S*B directed pairs of a call:
 *   PAIR       every directed pair is its own least-squares problem: normalised by ITS OWN mask count, weighted by ITS OWN
 *              depth-consistency weight map, depth-consistency term w_dc * mean over its own pixels.
 *   REFERENCE  the scalar that is minimised is the reference's compute_optimization_loss itself (optimizer.py:47-86, default
 *              options; pinned by golden G13 = the reference's loss and autograd gradients):
 *                forward term   sum_b sum_p keep dmin W_{source 0} / sum_b sum_p keep   (batch-summed normaliser; the weight map
 *                               of SOURCE 0 multiplies every forward pixel whichever source won it, optimizer.py:69);
 *                               without argmin: 0.25 sum valid W diff / sum valid over all forward pairs, no auto-mask (:71-73)
 *                inverse term   0.25 sum M W diff / sum M over all inverse pairs of the call (:75-79)
 *                depth consist. w_dc * (mean over all forward pairs and pixels + mean over all inverse pairs and pixels) (:83-86)
 *              Every pair still takes its own 6x6 / 7x7 step, with the exact gradient of that scalar w.r.t. its pose (the pose of
 *              source 0 also moves the weight of the pixels the other sources won) and its own block of the curvature model.
 *              The batch is the call's B targets: results depend on how windows are grouped into calls, exactly as the
 *              reference's loss depends on config['minibatch']. */
enum { TCSFM_WINDOW_PAIR = 0, TCSFM_WINDOW_REFERENCE = 1 };

/* the unknown of a target's depth in the dense window mode under TCSFM_WINDOW_REFERENCE:
 * FULL     one inverse depth per pixel (default).
 * QUARTER  the reference's own parametrisation of optimize_depth_pred (optimizer.py:194-198, 235-239): the QUARTER-resolution map
 *          (F.interpolate of the input to (H/4, W/4), bilinear), which every linearisation sees through its x4 bilinear upsampling
 *          (align_corners = False; sigmoid disparity and inverse depth are affine in each other, so it is the same unknown).  H and W
 *          must be multiples of 4.  The call starts from the quarter-resolution projection of the input map -- as the reference does --
 *          and returns the upsampled map; the SSIM prior stays centred on the full-resolution input (`self.target_disparity`, :89-90).
 *          Gradient: the full-resolution one through the transpose of the upsampling (exact; pinned on reference autograd w.r.t. the
 *          quarter-resolution leaf, golden G13 `qinit`).  Curvature: every cell gets the row-sum lumping sum_p U_pc D_p of the pixels'
 *          diagonal model, which majorises U' diag(D) U, so the depth is still eliminated per cell (a 6S x 6S system per target). */
enum { TCSFM_DEPTH_FULL = 0, TCSFM_DEPTH_QUARTER = 1 };

typedef struct tcsfm_opts {
    int32_t n_iters;       /* linearisations per refine call (BASELINE.json: 4)                         */
    int32_t solver;        /* TCSFM_SOLVER_*                                                             */
    int32_t param;         /* TCSFM_PARAM_*                                                              */
    int32_t refine;        /* TCSFM_REFINE_*                                                             */
    int32_t automask;      /* mask = valid * (diff < auto_err), helpers.py:17-19; options['automasking'] */
    int32_t depth_is_disp; /* depth inputs are sigmoid disparities: disp_to_depth is fused (learning_helpers.py:77-86) */
    int32_t host_ptrs;     /* 1: array arguments are host pointers (the call synchronises); 2: PINNED host pointers,
                              asynchronous (tcsfm_refine_window_async only)                                 */
    int32_t argmin;        /* tcsfm_refine_window with S > 1: per-pixel min over the sources, options['diff_img_argmin'] */
    float w_l1, w_ssim;    /* 0.15 / 0.85, train_mono.py:87                                              */
    float w_dc;            /* options['l_depth_consist_weight'] if options['l_depth_consist'] else 0, optimizer.py:83-86 */
    float irls_eps;        /* floor of the IRLS denominators                                             */
    float lambda0, lambda_up, lambda_down, lambda_min; /* Marquardt damping (relative to diag H)        */
    float min_depth, max_depth; /* config['min_depth'], config['max_depth'] (depth_is_disp only)       */
    float prior_scale;     /* POSE_SCALE only: weight of (log_scale - initial)^2.  The photometric cost cannot separate
                              depth scale from |t| (exact gauge); this prior makes the 7-DoF problem well posed.   */
    float lambda_depth;    /* dense mode: Marquardt damping of the per-pixel depth block (default 1.0)                */
    float prior_depth;     /* dense mode: weight of the masked prior sum M ((rho-rho0)/rho0)^2 / sum M (default 10)   */
    int32_t window_rule;   /* TCSFM_WINDOW_*: how the window forms put the directed pairs' costs together (default PAIR) */
    int32_t dense_joint;   /* tcsfm_refine_dense_window with S > 1 (default 1): the S forward pairs of a target share ONE inverse-
                              depth map and are solved JOINTLY (6S x 6S reduced camera system); 0: every forward pair refines its own
                              copy of the target depth (the round-2 behaviour)                                                  */
    float prior_init;      /* dense window mode under TCSFM_WINDOW_REFERENCE: options['l_depth_init_weight'] if options['l_depth_init'] else 0
                              (optimizer.py:89-90): weight of mean SSIM(current, initial sigmoid disparity of the target); default 0.1  */
    int32_t depth_param;   /* TCSFM_DEPTH_*: dense window mode under TCSFM_WINDOW_REFERENCE (default FULL)                             */
    float w_pose_consist;  /* pose window modes AND the dense window mode under TCSFM_WINDOW_REFERENCE, 6-DoF Gauss-Newton on the SE(3) chart:
                              0.1 if options['l_pose_consist'] else 0
                              (default 0, as the reference's drivers) -- w * mean |p_fwd + p_inv| over the S*B x 6 entries of the
                              reference's 6-vectors (optimizer.py:95-96).  Every pair gets the exact gradient w.r.t. its own pose (IRLS on
                              the L1 term, floor irls_eps) and the block-Jacobi majoriser of the curvature, its partner held at the
                              linearisation point; value and gradient pinned on reference autograd (golden G13 `full_pc`).  In the dense
                              mode the term enters the REDUCED pose systems (it does not depend on the maps): the forward pairs' diagonal
                              blocks of their target's joint system, the inverse pairs' own systems; tcsfm_linearize_dense_window exports
                              its value and gradient with the loss; queued calls that carry it are not merged                            */
    float w_smooth;        /* dense window mode under TCSFM_WINDOW_REFERENCE: options['l_smooth_weight'] if options['l_smooth'] else 0 (default
                              0, as the reference's drivers) -- w * get_smooth_loss(target disparity, target image) (optimizer.py:92-93,
                              losses.py:43-61: edge-aware L1 smoothness of the mean-normalised sigmoid disparity).  Exact gradient incl. the
                              normalisation's per-image constant (one more launch per linearisation: k_dref_smooth); curvature = the diagonal
                              majoriser of the edges' IRLS weights; pinned on reference autograd (golden G13 `fullinit_smooth`)           */
    int32_t free_source_depths; /* dense window mode under TCSFM_WINDOW_REFERENCE, either depth_param (default 0): 1 = the SOURCE depth maps
                              are unknowns as well, as in the reference's optimize_depth_pred (optimizer.py:194-198 optimises the disparities of
                              target AND sources; no prior on the sources).  Every inverse pair then is a group of its own -- its pose and the
                              source map it back-projects, per-pixel Schur elimination as in a forward group of one source -- with the adjoint
                              of the forward pair's samples of that map in its gradient (tcsfm_linearize_dense_window_sources); after every
                              step the forward pairs sample the new source maps.  depth_out's inverse slots return the refined source maps.  With
                              depth_param = TCSFM_DEPTH_QUARTER the unknowns are the reference's own leaves: quarter-resolution maps of target and sources */
} tcsfm_opts;

/* per-pair, per-linearisation statistics written by tcsfm_refine: [N][n_iters+1][TCSFM_NSTAT] fp32.
 * Row i < n_iters: the i-th linearisation (cost, photometric part, mask count, damping) and the pose it was evaluated at,
 * i.e. the trajectory of iterates (the analogue of the reference's stacked poses, train_mono.py:71-79).  Row n_iters:
 * Gauss-Newton -- the final pose, cost fields 0 (not evaluated); LM -- the cost check of the last trial step. */
enum { TCSFM_STAT_COST = 0, TCSFM_STAT_COST_PHOTO = 1, TCSFM_STAT_NMASK = 2, TCSFM_STAT_LAMBDA = 3,
       TCSFM_STAT_POSE = 4 /* ..9: the 6-vector the row was evaluated at */, TCSFM_NSTAT = 10 };

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Allocates all device scratch for up to max_pairs directed pairs of H x W images on `device`.
 * Nothing in the reference corresponds to this: PyTorch owns memory there (optimizer.py:15-27). */
int tcsfm_create(tcsfm_handle *out, int device, int H, int W, int max_pairs);
void tcsfm_destroy(tcsfm_handle h);
const char *tcsfm_last_error(tcsfm_handle h); /* h may be NULL: last create() error */
/* Run on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL is HIP's legacy default
 * stream (stream 0), which is what PyTorch's default stream is.  A fresh handle runs on its own non-blocking stream;
 * tcsfm_use_own_stream() switches back to it. */
int tcsfm_set_stream(tcsfm_handle h, void *hip_stream);
int tcsfm_use_own_stream(tcsfm_handle h);
int tcsfm_synchronize(tcsfm_handle h);
void tcsfm_default_opts(tcsfm_opts *o);
/* bytes of HBM traffic the algorithm must move per pixel per pair per linearisation: 32 */
int tcsfm_algorithmic_bytes_per_pixel(const tcsfm_opts *o);

/* ---- reference-function drop-ins (a1, a5, a7, a12) -------------------------------------------- */

/* disp_to_depth, utils/learning_helpers.py:77-86.  n elements; scaled/depth may be NULL. */
int tcsfm_disp_to_depth(tcsfm_handle h, const tcsfm_opts *o, int64_t n, const float *disp, float *scaled_disp, float *depth);

/* SSIM_Loss.forward(x, y), losses.py:27-41: `planes` = N*C images of H x W each (reflect pad 1, 3x3 means, clamp). */
int tcsfm_ssim(tcsfm_handle h, const tcsfm_opts *o, int planes, const float *x, const float *y, float *out);

/* get_smooth_loss(disp, img), losses.py:43-61: edge-aware smoothness of the mean-normalised disparity, disp [N,1,H,W],
 * img [N,3,H,W] -> one scalar (host double; the call synchronises the stream).  Off by default in the reference
 * (options['l_smooth'], run_sequential_optimization.py:87); a logging quantity here, not a term of the Gauss-Newton cost. */
int tcsfm_smooth_loss(tcsfm_handle h, const tcsfm_opts *o, int N, const float *disp, const float *img, double *loss_out);

/* inverse_warp2(src, depth_t, depth_s, -pose, K), models/stn.py:234-273.
 * Outputs (any may be NULL): img_rec [N,3,H,W], valid [N,1,H,W], proj_depth, comp_depth [N,1,H,W]. */
int tcsfm_warp(tcsfm_handle h, const tcsfm_opts *o, int N, const float *src, const float *depth_t, const float *depth_s,
               const float *pose, const float *K, float *img_rec, float *valid, float *proj_depth, float *comp_depth);

/* The coupled-iteration input assembly of solve_pose_iteratively, train_mono.py:73-77, fused into the warp:
 * posenet_in [N,6,H,W] = (tgt * valid_mask, img_rec) for the next PoseNet call; valid [N,1,H,W] optional. */
int tcsfm_warp_posenet_input(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                             const float *depth_s, const float *pose, const float *K, float *posenet_in, float *valid);

/* compute_photometric_error, optimization_experiments/helpers.py:8-23 == per-pair residual assembly of
 * solve_pose_iteratively, train_mono.py:82-100.  Outputs (any may be NULL), all [N,1,H,W] except img_rec:
 * diff (diff_img), valid (warp validity, stn.py:268-269), weight (weight_mask), auto_err (auto_mask_error),
 * auto_mask (diff < auto_err), img_rec [N,3,H,W]. */
int tcsfm_photometric(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                      const float *depth_s, const float *pose, const float *K, float *diff, float *valid, float *weight,
                      float *auto_err, float *auto_mask, float *img_rec);

/* The scalar generate_loss_surface sweeps, optimization_experiments/plot_loss_surface.py:31-33,45-47:
 * cost[i] = sum(diff*mask*weight)/sum(mask) (+ w_dc*mean(1-weight)) of ONE pair (first pair of the arrays)
 * under P candidate poses [P,6].  cost_out: P doubles (host pointer always). */
int tcsfm_loss_surface(tcsfm_handle h, const tcsfm_opts *o, const float *tgt, const float *src, const float *depth_t,
                       const float *depth_s, const float *K, int P, const float *poses, double *cost_out);

/* ---- the Gauss-Newton / LM engine (new functionality; north star) ------------------------------ */

/* One linearisation of N pairs at the given poses (and log depth-scales, may be NULL): normal equations
 * for parity tests.  np = 6 or 7 per o->refine.  Host outputs (doubles): Hmat [N,np,np], g [N,np],
 * stats [N,4] = cost, cost_photo, cost_dc, n_mask. */
int tcsfm_linearize(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                    const float *depth_s, const float *pose, const float *log_scale, const float *K,
                    double *Hmat, double *g, double *stats);

/* Window form of tcsfm_linearize: ONE linearisation of the 2*S*B directed pairs of a window (layouts of tcsfm_refine_window)
 * at the given poses, under o->argmin / o->window_rule.  Host outputs (doubles): Hmat [2SB,np,np], g [2SB,np],
 * stats [2SB,4] = cost, cost_photo, cost_dc, n_mask -- under TCSFM_WINDOW_REFERENCE the pairs' costs add up to
 * compute_optimization_loss (optimizer.py:47-86) and g is its gradient w.r.t. each pair's left SE(3) perturbation. */
int tcsfm_linearize_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                           const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *log_scale,
                           double *Hmat, double *g, double *stats);

/* Refine N directed pairs: replaces the epoch loop of DepthOptimizer.optimize_window
 * (optimization_experiments/optimizer.py:217-274) for the pose / pose+scale unknowns with
 * o->n_iters Gauss-Newton (or LM) iterations on the reference's residual.
 *   pose_in       [N,6] initial pose (e.g. PoseNet output);  pose_out [N,6] refined pose (may alias pose_in)
 *   log_scale_in  [N] or NULL (start at 0), log_scale_out [N] or NULL: only read/written when refine == POSE_SCALE
 *   stats_out     [N,n_iters+1,TCSFM_NSTAT] or NULL */
int tcsfm_refine(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                 const float *depth_s, const float *K, const float *pose_in, const float *log_scale_in, float *pose_out,
                 float *log_scale_out, float *stats_out);

/* Window form of tcsfm_refine: the call surface of solve_pose_iteratively / optimize_window (train_mono.py:41-62,
 * optimizer.py:136-160): B target frames with S source frames each, given ONCE --
 *   tgt [B,3,H,W], srcs [S,B,3,H,W], depth_t [B,1,H,W], depth_s [S,B,1,H,W], K [B,3,3] --
 * and refined as 2*S*B directed pairs in the reference's stacked order (train_mono.py:54-62): pair s*B+b reconstructs
 * target b from source s (forward), pair S*B+s*B+b the reverse (inverse).  pose_in / pose_out [2*S*B,6],
 * log_scale_* [2*S*B] or NULL, stats_out [2*S*B,n_iters+1,TCSFM_NSTAT] or NULL, as in tcsfm_refine.
 * With o->argmin and S > 1 the forward pairs of a target use the reference's per-pixel min over the sources
 * (compute_optimization_loss, optimizer.py:47-69): at every linearisation a pixel counts only for the source with the
 * smallest photometric error there, under the union validity mask and the auto-mask of the minima.
 * With the default o->window_rule = TCSFM_WINDOW_PAIR each directed pair is its own least-squares problem: normalised by ITS OWN
 * count of selected pixels, with ITS OWN depth-consistency weight map; TCSFM_WINDOW_REFERENCE minimises the reference's scalar
 * loss itself -- batch-summed normalisers, the weight map of source 0 on every forward pixel (optimizer.py:69), 0.25 x the
 * inverse term -- see the enum above (golden G13).  The selection (which pixels count, for which source) is the reference's
 * under both rules and is pinned on its own maps (golden G4).
 * The handle must have been created with max_pairs >= 2*S*B. */
int tcsfm_refine_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                        const float *depth_t, const float *depth_s, const float *K, const float *pose_in,
                        const float *log_scale_in, float *pose_out, float *log_scale_out, float *stats_out);

/* Dense mode (BASELINE config 5): refine the 6-DoF pose AND the per-pixel inverse depth of the target of N directed
 * pairs: o->n_iters Gauss-Newton iterations, exact depth gradient (equal to reference autograd d loss / d depth), per-pixel
 * Schur elimination of the depth block, 6x6 reduced pose system, back-substitution.  depth_t is the initial target depth
 * (or sigmoid disparity with depth_is_disp); depth_out [N,1,H,W] receives the refined DEPTH.  w_dc must be 0 (the depth prior of
 * opts.prior_depth regularises instead).  With TCSFM_SOLVER_LM the pose block is Marquardt-damped and a trial that does not
 * lower the cost is rolled back -- pose AND depth map -- before the step is recomputed from the accepted linearisation. */
int tcsfm_refine_dense(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                       const float *depth_s, const float *K, const float *pose_in, float *pose_out, float *depth_out,
                       float *stats_out);

/* Window form of tcsfm_refine_dense (the `optimize_depth_pred` mode of optimize_window with its default options,
 * optimizer.py:194-198 + 47-69): B targets x S sources given once as in tcsfm_refine_window; depth_out [2*S*B,1,H,W] in the
 * stacked pair order.
 *   JOINT (o->dense_joint, the default, S = 2 or 3): the reference optimises ONE disparity per frame that every term of the loss
 *     sees (optimizer.py:235-247).  The S forward pairs of target b share one inverse-depth map; unknowns per target = S poses +
 *     the map; cost = the forward term of the reference's loss (min over the sources under o->argmin, or every valid source
 *     without it, :71-73) with every pixel weighted by the depth-consistency map of the source it counts for, + the depth prior.
 *     (optimizer.py:69 multiplies every pixel by the map of source 0; with the depth as an unknown that makes the depth of a pixel
 *     whose SOURCE-0 sample touches the zero padding depend chaotically on the fifth digit of source 0's pose although another
 *     source won it -- measured, DESIGN.md section 2 -- so o->window_rule does not apply here; the pose modes offer it.)
 *     Exact gradient (= reference autograd of that cost w.r.t. the poses AND the shared depth, golden G13 `fwd_ownw`);
 *     per-pixel Schur elimination of the depth -> ONE reduced camera system of 6S x 6S per target (12 x 12 for the KITTI
 *     window; its off-diagonal blocks vanish identically under argmin, where every pixel counts for exactly one source),
 *     back-substitution once per target pixel.  LM accepts / rejects all S poses and the map together.  The S forward slots of
 *     depth_out hold the same refined map; stats rows of the forward pairs: [joint cost of the target, own share, own mask
 *     count, lambda, iterate].  The inverse pairs refine their pose and the depth of THEIR target (source frame (s,b)) as before.
 *   PER-PAIR COPIES (o->dense_joint = 0, or S = 1 where the two coincide): every directed pair refines ITS OWN copy of its
 *     target's depth; with o->argmin the forward pairs use the min over the sources at the current poses and depth copies.
 *   REFERENCE LOSS (o->window_rule = TCSFM_WINDOW_REFERENCE; S = 1 .. 3, Gauss-Newton; round 4): the scalar that is minimised is
 *     the reference's compute_optimization_loss as optimize_depth_pred sees it (optimizer.py:47-90):
 *       c_f / K_f sum M_s W_x diff_s  (forward term: K_f summed over the call's B targets; W_x = the weight map of SOURCE 0 on every
 *                                      selected pixel under o->argmin, c_f = 1; without argmin W_x = W_s, c_f = 0.25, no auto-mask)
 *       + 0.25 / K_i sum M_i W_i diff_i  (inverse pairs, K_i summed over all of them)
 *       + o->w_dc / (S B HW) sum (dd_fwd + dd_inv)  (depth consistency of both directions; w_dc > 0 is allowed in this mode)
 *       + o->prior_init / (B HW) sum SSIM(sigma, sigma_0)  (l_depth_init: SSIM between the target's current and initial sigmoid
 *                                      disparity, sigma = (1/depth - 1/max_depth) / (1/min_depth - 1/max_depth))
 *     Unknowns: the poses of all 2 S B directed pairs and ONE inverse-depth map per target; the source depth maps stay at their
 *     input unless o->free_source_depths is set (the reference lets them drift too, with no prior on them: see that field).  The target depth enters the forward pairs as the
 *     back-projected depth and the inverse pairs as the depth they SAMPLE (stn.py:271): the gradient contains both -- the second as
 *     the adjoint of the bilinear sample, scattered with 64-bit fixed-point atomics (order-independent: results stay bit-
 *     reproducible) -- and equals reference autograd w.r.t. every pose and the shared depth (golden G13 `full`, `fullinit`; see
 *     tcsfm_linearize_dense_window).  Curvature: the joint dense model + the IRLS curvature of the depth-consistency terms + a
 *     diagonal model of the 3x3-coupled prior, w / r^2 (1/d2 + 1/(9 d1)) with the pixel's own SSIM denominators; the sampled-depth
 *     terms are gradient-only.  The inverse pairs take 6 x 6 pose steps under the window rule.  After every step the new map also
 *     replaces the depth the inverse pairs sample.  depth_out: the S forward slots hold the refined map, the inverse slots the
 *     source depths (unchanged; refined under o->free_source_depths).  stats rows of the forward pairs: [forward group's loss (forward + its depth consistency + prior),
 *     own share, own mask count, lambda, iterate]; of the inverse pairs: as tcsfm_refine_window under the rule.
 *     LAUNCHES (round 5): with the source maps fixed an iteration is FOUR dependent launches -- one over all 2 S B directed pairs
 *     (the forward pairs' mask / min-over-sources selection and its count K_f; the inverse pairs' 6 x 6 systems, K_i and the adjoint
 *     scatter, whose two parts are summed without their factors so that no count has to precede it), the joint kernel, one launch
 *     that solves the targets' 6S x 6S systems and the inverse pairs' 6 x 6 systems, the back-substitution (five with the quarter-
 *     resolution unknown) -- and a call issues no memset or copy: csrc/dense_ref_kernel.h. */
int tcsfm_refine_dense_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                              const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                              float *depth_out, float *stats_out);

/* ONE linearisation of tcsfm_refine_dense_window's REFERENCE-LOSS mode at `pose` and `depth_t` (nothing is updated): the loss and its
 * exact gradients, for pinning against the reference's loss and autograd (golden G13) -- the dense counterpart of
 * tcsfm_linearize_window.  Outputs (HOST pointers, float64): scal_out [8] = loss, forward group (forward term + its depth
 * consistency + prior), inverse photometric term, inverse depth-consistency term, K_f, K_i, a_f = c_f / K_f, the l_pose_consist term
 * (o->w_pose_consist; part of the loss and of g_pose_out, golden G13 `full_pc`);
 * g_pose_out [2*S*B][6] = d loss / d (left SE(3) perturbation of every directed pair's warp transform) in the stacked pair order;
 * g_rho_out [B][H*W] float32 (device or host pointer as o->host_ptrs says) = d loss / d (inverse depth of target b).
 * depth0 [B,1,H,W] (same kind of pointer as depth_t; DEPTH, not disparity) or NULL: the centre of the l_depth_init prior -- the
 * reference's self.target_disparity, the map the optimisation started from (optimizer.py:156,90); NULL: depth_t itself (the state at
 * the first epoch, where the prior and its gradient vanish). */
int tcsfm_linearize_dense_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                 const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *depth0,
                                 double *scal_out, double *g_pose_out, float *g_rho_out);

/* The same linearisation with one more output: g_rho_src_out [S][B][H*W] float32 (kind of pointer as o->host_ptrs says) = d loss /
 * d (inverse depth of SOURCE map (s, b)) -- the depth maps tcsfm_refine_dense_window holds fixed, which the reference's optimize_depth_pred
 * optimises as well (optimizer.py:194-198: the quarter-resolution disparities of the target AND of every source are leaves).  A source
 * map is the back-projected depth of its inverse pair (local: photometric term through the SSIM window, the pair's own weight, its depth-
 * consistency term) and the depth its forward pair SAMPLES (stn.py:271: through that pair's depth-consistency term and the weight map it
 * provides -- source 0's multiplies every selected pixel under o->argmin, optimizer.py:69): both parts, the second as the adjoint of the
 * bilinear sample.  Equals reference autograd (golden G13 `full_grad_depth_s`; oracle: dref_source_depth_gradient).  With this every leaf
 * of the reference's loss has its exact gradient on the device; making the source maps unknowns of the Gauss-Newton step is not done. */
int tcsfm_linearize_dense_window_sources(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                         const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *depth0,
                                         double *scal_out, double *g_pose_out, float *g_rho_out, float *g_rho_src_out);

/* ScaleRecovery.forward, models/dnet_layers.py:249-327 (the step right after the path in optimize_window,
 * optimizer.py:254-258): camera-height map |P.n| from 8-neighbour surface normals, ground mask, exact lower median of the
 * masked heights over the batch, scale = real_cam_height / median.  pad_to_batch mirrors the reference's padding of a short
 * batch with copies of image 0 (dnet_layers.py:307-311; pass config['minibatch'], or 0 for none).
 * scale_out [1]; optional: median_out [1], height_out / mask_out [N,1,H,W]. */
int tcsfm_scale_recovery(tcsfm_handle h, const tcsfm_opts *o, int N, const float *depth, const float *K, float real_cam_height,
                         int pad_to_batch, float *scale_out, float *median_out, float *height_out, float *mask_out);

/* ---- PoseNet and the coupled pose loop ----------------------------------------
 * The reference's PoseNet (models/pose_models.py:88-147: seven weight-standardised stride-2 convolutions + GroupNorm(16) + ReLU,
 * 1x1 head, spatial mean, x 0.01) is evaluated `iterations` times per window inside solve_pose_iteratively (train_mono.py:64,77).
 * Here it runs as hand-written gfx950 kernels (fp32 matrix instructions, weight standardisation folded into the loaded weights,
 * GroupNorm + ReLU applied by the consuming layer), so that the whole coupled loop stays inside the library.
 *   tcsfm_posenet_create   activations for up to max_images samples of the handle's H x W
 *   tcsfm_posenet_load     HOST pointers to the parameters of the reference module, in its own layouts: conv_w[l] = conv{l+1}.0.weight
 *                          [cout,cin,k,k], conv_b[l] = conv{l+1}.0.bias [cout] (NULL: 0), gn_w / gn_b[l] = conv{l+1}.1.weight / .bias
 *                          [cout] (NULL: 1 / 0), head_w = pose_pred.weight [6,256(,1,1)], head_b = pose_pred.bias [6]
 *   tcsfm_posenet_forward  pose_model(imgs): imgs [N,6,H,W] (device) -> pose [N,6] (device).  The work split of the layers (K split,
 *                          channel blocks per wave) is chosen from the NUMBER OF IMAGES of the call -- two fixed regimes, N <= 4 and N > 4
 *                          -- and the K split fixes the summation order: results are bit-identical for every N within a regime and
 *                          agree to ~1e-6 relative across the two.  Consequence for the sequence calls below: the PoseNet poses of a
 *                          window depend, at the 1e-6 level, on whether its call holds up to 4 or more images (windows_per_call), and the
 *                          coupled loop's discrete warp-validity decisions can amplify that on individual windows (measured with random
 *                          weights: 3 of 199 windows at 1e-5 .. 1e-2, all others ~1e-6).
 *   tcsfm_solve_pose_iteratively   train_mono.py:41-81 for a window (layouts of tcsfm_refine_window): PoseNet on (tgt | src) /
 *                          (src | tgt), then num_iter-1 rounds of { inverse_warp2 with -pose; PoseNet on (tgt * valid | img_rec);
 *                          pose += correction }.  poses_out [2*S*B,6] = the last iterate; stacked_out [2*S*B,num_iter,6] optional
 *                          (the reference's stacked_poses).  Asynchronous on the handle's stream. */
typedef struct tcsfm_posenet tcsfm_posenet;
int tcsfm_posenet_create(tcsfm_handle h, int max_images, tcsfm_posenet **out);
void tcsfm_posenet_destroy(tcsfm_posenet *pn);
int tcsfm_posenet_load(tcsfm_posenet *pn, const float *const conv_w[7], const float *const conv_b[7], const float *const gn_w[7],
                       const float *const gn_b[7], const float *head_w, const float *head_b);
int tcsfm_posenet_forward(tcsfm_posenet *pn, int N, const float *imgs, float *pose_out);
int tcsfm_solve_pose_iteratively(tcsfm_handle h, tcsfm_posenet *pn, int num_iter, int B, int S, const float *tgt, const float *srcs,
                                 const float *depth_t, const float *depth_s, const float *K, float *poses_out, float *stacked_out);

/* ---- lanes: several refinements in flight (streaming a sequence) ----------------------------------
 * The reference's driver refines one window after another (run_sequential_optimization.py:186-247: DataLoader batch -> H2D ->
 * optimize_window); consecutive windows do not depend on each other.  A B=1 refinement leaves the GPU idle between its short
 * kernels, and a host-pointer call spends more time on PCIe than on the refinement, so the library can keep several calls in
 * flight: lane k >= 1 owns a HIP stream and a full set of scratch buffers (lane 0 is the handle itself).
 *   tcsfm_set_lanes          1..8 lanes (allocates / frees the lanes' scratch; default 1).  MEASURED HAZARD (round 4, ROCm 7.2 / MI355X,
 *                            scripts/lane_order_probe.py): how well lanes overlap depends on the order in which the PROCESS created
 *                            its HIP streams.  A handle (and its lanes) created before the process's first device work -- an upload, a
 *                            kernel -- gives four lanes that are slower than one (10 000 against 13 500 frame-pairs/s; created after
 *                            the inputs are on the card: 22 000-26 000), and so do five or more lanes in any order: create the handle
 *                            after the first upload, and use at most four lanes.
 *   tcsfm_refine_window_async   tcsfm_refine_window on `lane`, asynchronously.  Device pointers (host_ptrs = 0): the lane first
 *                            waits for the work queued on the handle's stream at call time (the producers of the inputs).
 *                            Pinned host pointers (host_ptrs = 2): the copies run on the lane's stream -- they overlap the other
 *                            lanes' kernels -- and the call does NOT synchronise; read the outputs after tcsfm_lane_synchronize.
 *   tcsfm_lane_wait          the handle's stream waits (on the device, not the host) for the lane's last call
 *   tcsfm_lane_synchronize   the host waits for the lane's last call; reports a pending TCSFM_E_INTRINSICS of that lane
 *   tcsfm_lane_event         marks the current end of the lane's work with an event (owned by the library; one of a ring of 64
 *                            per lane, so a mark stays valid until 64 later marks of that lane) -- for callers that recycle
 *                            input buffers: tcsfm_stream_wait_event(stream, mark) makes e.g. their copy stream wait, on the
 *                            device, until the lane has consumed the buffer */
int tcsfm_set_lanes(tcsfm_handle h, int n_lanes);
/* Round 5: tcsfm_set_lanes MEASURES whether lanes pay in this process (24 B=1 refinements of the handle's own kernels on stand-in images, one
 * after the other on the handle's stream, then round-robin over all lanes; ~5 ms).  If they do not -- the hazard above -- the handle falls back: lane calls (tcsfm_refine_window_async, the
 * sequence calls, the merged sequences' second stream) run on the handle's own stream, one after the other, results unchanged, one line on
 * stderr.  tcsfm_lane_probe reports the outcome: *serial = 1 when the fallback is active, and the two probe times in ms.  The queued calls
 * (tcsfm_set_coalesce) keep the chip busy either way.  TCSFM_LANE_PROBE=0 in the environment skips the probe. */
int tcsfm_lane_probe(tcsfm_handle h, int *serial, float *one_stream_ms, float *two_streams_ms);
/* Debug aid (no reference counterpart; GPU AddressSanitizer is not available on the target pool): with TCSFM_DEBUG_GUARDS=1 in the environment
 * when the library is loaded, every device allocation of the library carries a 4 KB pattern band in front of it and behind it.  This call
 * synchronises the device, reads all bands back and reports (stderr, once per allocation) those a kernel wrote into: *n_allocations = live
 * allocations checked (-1: guards are off), *n_damaged = how many have a damaged band.  The GPU test suite runs with the guards on. */
int tcsfm_debug_check_guards(int *n_allocations, int *n_damaged);
/* ... and its self-test: writes four bytes past the end of a scratch allocation of its own and says whether the bands caught it
 * (*detected = 1 / 0; -1: guards are off) */
int tcsfm_debug_guard_selftest(int *detected);
int tcsfm_refine_window_async(tcsfm_handle h, int lane, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                              const float *depth_t, const float *depth_s, const float *K, const float *pose_in,
                              const float *log_scale_in, float *pose_out, float *log_scale_out, float *stats_out);
/* the dense mode (tcsfm_refine_dense_window) on a lane: same ordering rules */
int tcsfm_refine_dense_window_async(tcsfm_handle h, int lane, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                    const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                                    float *depth_out, float *stats_out);
/* ---- coalesced calls (round 4): queued B-window calls of one shape run as ONE launch sequence ------------------------
 * The robust way of keeping the chip busy with small calls: lanes depend on how the runtime places their hardware queues (see
 * tcsfm_set_lanes), a merged launch sequence does not.  tcsfm_refine_window_queued takes the arguments of tcsfm_refine_window
 * (device pointers, no statistics outputs; tcsfm_refine_window_scale_queued carries the log depth-scales of TCSFM_REFINE_POSE_SCALE) and only NOTES the call; when `max_calls` calls of the same
 * shape (B, S) and options are waiting -- or at tcsfm_flush / tcsfm_synchronize, or when a call of another shape arrives -- they run as
 * one pack / (linearise, solve) x n_iters sequence over all their directed pairs on the handle's stream: the kernels reach every
 * call's own buffers through a pointer table (k_pack_coal) and every call's refined poses go to its own pose_out.  Per window the
 * results are the bits of tcsfm_refine_window run on its own (the kernels are batch-independent under TCSFM_WINDOW_PAIR; a call under
 * TCSFM_WINDOW_REFERENCE, whose loss couples the windows of a call, is never merged with others and runs at once).  The caller's
 * buffers must stay valid and unchanged until the flush that runs them has been issued AND has completed on the handle's stream.
 * ORDER (round 5): a queued call is never overtaken.  Every other entry point that puts work on the handle's stream (tcsfm_refine*,
 * tcsfm_refine_dense*, the *_async and sequence calls, tcsfm_linearize*, the drop-ins, the PoseNet calls) first launches what is
 * waiting and orders the handle's stream behind the merged sequences that ran on lanes; tcsfm_set_stream does so on the OLD stream
 * before it switches (the queued calls were noted behind that stream's producers), tcsfm_use_own_stream likewise, and tcsfm_destroy
 * runs what is waiting before it tears the handle down (queued calls are never dropped).
 *   tcsfm_set_coalesce(h, max_calls)   0 / 1: off (every queued call runs at once); up to 16; the handle's max_pairs bounds the merged
 *                                      sequence as well (2 S B x calls <= max_pairs)
 *   tcsfm_set_coalesce_lanes(h, n)     merged sequences alternate over n of the handle's streams (1: the handle's own, the default;
 *                                      up to the lanes of tcsfm_set_lanes): sequence k runs on lane k mod n, behind everything queued
 *                                      on the handle's stream when it is issued.  The 20-workgroup solve kernels of one sequence then
 *                                      overlap the chip-filling launches of the other (same bits).  A sequence issued on a lane is
 *                                      ordered before later work of the handle's stream only by tcsfm_flush / tcsfm_synchronize.
 *   tcsfm_flush(h)                     launch what is waiting (asynchronous) and order the handle's stream behind every merged sequence
 *                                      issued so far, whichever lane ran it
 *   tcsfm_coalesce_counts              launch sequences issued / calls they carried, since the handle was created */
int tcsfm_set_coalesce(tcsfm_handle h, int max_calls);
int tcsfm_set_coalesce_lanes(tcsfm_handle h, int n_streams);
int tcsfm_refine_window_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                               const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out);
/* the same with the depth-scale unknown (TCSFM_REFINE_POSE_SCALE): log_scale_in [2*S*B] or NULL (0), log_scale_out [2*S*B] or NULL */
int tcsfm_refine_window_scale_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                     const float *depth_t, const float *depth_s, const float *K, const float *pose_in,
                                     const float *log_scale_in, float *pose_out, float *log_scale_out);
int tcsfm_flush(tcsfm_handle h);
int tcsfm_coalesce_counts(tcsfm_handle h, int *batches, int *calls);
/* The dense counterpart (arguments of tcsfm_refine_dense_window, device pointers, no statistics): queued per-pair Gauss-Newton dense calls
 * with ONE source per target (S = 1, TCSFM_WINDOW_PAIR) of the same shape and options are merged like the pose calls -- every call's
 * refined poses and depth maps go to its own outputs, bit-identical to the call on its own.  Round 5: calls under TCSFM_WINDOW_REFERENCE (the
 * reference's own loss, optimizer.py:47-90; S <= 3, Gauss-Newton, fixed source maps, per-pixel or quarter-resolution unknown) are merged as
 * well: that loss couples the windows of ONE call through its batch normalisers, so inside the merged sequence every call is a normaliser
 * group of its own (mask counts, per-map weights) and its results are the bits of the call run alone.  Any other dense call (the joint mode
 * for S > 1 under TCSFM_WINDOW_PAIR, LM, free_source_depths) flushes what is waiting and runs at once.  Pose calls and dense calls are never
 * merged with each other. */
int tcsfm_refine_dense_window_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                     const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                                     float *depth_out);

/* The reference's sequential driver loop as ONE call (run_sequential_optimization.py:186-247: for every window DataLoader batch ->
 * H2D in process_sample_batch, data/kitti_loader.py:60-98 -> optimize_window, strictly one window after the other).
 * Window w = the S + 1 consecutive frames w .. w+S, w = 0 .. T-S-1: its target is frame w + target_pos (target_pos = -1: the middle
 * one, (S+1)/2, as the reference's loaders choose it -- data/kitti_loader.py:271-273 -- i.e. the LATER frame of a two-frame window;
 * 0: the first), its sources are the other frames in order -> its 2*S directed pairs in the stacked order of train_mono.py:54-62
 * (forward pairs, then inverse pairs).  All array arguments are HOST pointers here (pinned memory makes the
 * copies asynchronous; pageable memory works, slower), whatever opts.host_ptrs says:
 *   frames [T,3,H,W], depths [T,1,H,W] (or disparities, opts.depth_is_disp), K [3,3] (one camera for the sequence),
 *   pose_init / pose_out [T-S, 2*S, 6], log_scale_out [T-S, 2*S] or NULL (TCSFM_REFINE_POSE_SCALE; the scale starts at 0).
 * Every frame crosses PCIe once, on a high-priority copy stream of the handle, four frames per copy, into a device ring of `ring`
 * frames (0 = default; at least windows_per_call + S + 4, or S + 2 with one window per call and single-frame copies); the calls are
 * issued round robin on the handle's lanes (tcsfm_set_lanes; 2-3 lanes: HIP maps a process's streams onto four hardware queues
 * unless GPU_MAX_HW_QUEUES is raised before the first HIP call, and the call's copy and pack streams take queues too; create the
 * handle AFTER the process's first device work -- see tcsfm_set_lanes) from the ring by pointer; a slot is recycled -- on the device, by events -- once every call reading it has
 * finished.  windows_per_call (0 = default 8, capped by max_pairs / (2 S)): the targets and every source of consecutive windows
 * are runs of the ring, so one call refines that many windows at once (the kernels fill the chip).  The call returns when all
 * windows are done (it synchronises).  Under o->window_rule = TCSFM_WINDOW_PAIR (the default) results are bit-identical to one
 * tcsfm_refine_window call per window, whatever the lanes, the ring and windows_per_call.  Under TCSFM_WINDOW_REFERENCE the
 * reference's batch-summed normalisers (optimizer.py:69,79) are taken over the windows of ONE CALL: windows_per_call plays the
 * role of the reference's config['minibatch'] (run_sequential_optimization.py:186 hands optimize_window a DataLoader batch of that
 * many windows), results depend on it exactly as the reference's loss depends on the minibatch, and are bit-identical to
 * tcsfm_refine_window calls over the same groups of windows; windows_per_call = 1 gives the per-window loss. */
int tcsfm_refine_sequence(tcsfm_handle h, const tcsfm_opts *o, int T, int S, const float *frames, const float *depths, const float *K,
                          const float *pose_init, float *pose_out, float *log_scale_out, int ring, int windows_per_call, int target_pos);
/* The same loop with the reference's pose initialisation inside it: for every window the coupled PoseNet loop of
 * train_mono.py:64-80 (tcsfm_solve_pose_iteratively, `num_iter` network evaluations: config['iterations'], 4 in the reference's
 * scripts) produces the initial poses on the window's lane, then the window is refined -- what optimize_window does per window
 * (optimizer.py:136-297) minus the depth network, whose per-frame output is the `depths` argument (depths, not disparities).
 * `pn`: a loaded PoseNet of this handle with max_images >= 2*S; the lanes run copies of it that share its weights.
 * pose_init_out [T-S, 2*S, 6] (or NULL) receives the PoseNet poses, pose_out the refined ones.  Bit-identical to one
 * tcsfm_solve_pose_iteratively + tcsfm_refine_window per CALL's windows (windows_per_call of them as one batch; the PoseNet's
 * work split -- hence its rounding -- depends on the number of images, see tcsfm_posenet_forward: per-window calls agree to 1e-5). */
int tcsfm_odometry_sequence(tcsfm_handle h, tcsfm_posenet *pn, int num_iter, const tcsfm_opts *o, int T, int S, const float *frames,
                            const float *depths, const float *K, float *pose_init_out, float *pose_out, float *log_scale_out, int ring,
                            int windows_per_call, int target_pos);
/* The dense mode (tcsfm_refine_dense_window: pose + per-pixel inverse depth with the Schur complement, BASELINE config 5) over a
 * sequence, same streaming and batching: depth_out [T-S, 2*S, H, W] (host) receives every directed pair's refined depth map.
 * Bit-identical to one tcsfm_refine_dense_window call per window. */
int tcsfm_refine_dense_sequence(tcsfm_handle h, const tcsfm_opts *o, int T, int S, const float *frames, const float *depths, const float *K,
                                const float *pose_init, float *pose_out, float *depth_out, int ring, int windows_per_call, int target_pos);
int tcsfm_lane_wait(tcsfm_handle h, int lane);
int tcsfm_lane_synchronize(tcsfm_handle h, int lane);
int tcsfm_lane_event(tcsfm_handle h, int lane, void **event_out);
int tcsfm_stream_wait_event(tcsfm_handle h, void *hip_stream, void *event);

/* ---- graph replay of repeated calls -------------------------------------------------------------------
 * A B = 1 refinement is nine short launches; with several calls in flight the HOST's launch rate (about 42 us per call), not the
 * kernels (about 40 us per call), sets the throughput, and on a slow host it alone does.  tcsfm_set_graph_replay(h, n > 0) lets the
 * handle and its lanes keep up to n captured calls each: a tcsfm_refine / tcsfm_refine_window / tcsfm_refine_dense / tcsfm_refine_dense_window
 * (also through the _async forms) with DEVICE pointers whose arguments -- options, sizes and every pointer -- equal those of an earlier
 * call is captured as one HIP graph the second time it is seen and replayed with a single hipGraphLaunch from the third time on
 * (least recently used entry evicted).  Same kernels, same order, same arguments: results are bit-identical to plain launches.
 * Calls with host pointers, under tcsfm_debug_trace / tcsfm_profile_begin, or on HIP's NULL stream are launched plainly.
 * The buffers named by the pointers must stay valid while the handle may replay the call (n = 0 drops all captures).  Default: off.
 *   tcsfm_graph_replay_counts   captures made / replays launched so far (handle + lanes), for tests and measurement.            */
int tcsfm_set_graph_replay(tcsfm_handle h, int max_graphs);
int tcsfm_graph_replay_counts(tcsfm_handle h, int *captures, int *replays);


/* ---- measurement hooks (bench.py) -------------------------------------------------------------- */

/* While profiling is on, every launch of the three kernel classes is bracketed by a pair of HIP events recorded on the
 * handle's stream.  tcsfm_profile_end() synchronises the stream and returns summed elapsed milliseconds and launch
 * counts per class: index 0 = k_linearize (the hot kernel), 1 = k_solve, 2 = k_pack. */
int tcsfm_profile_begin(tcsfm_handle h);
int tcsfm_profile_end(tcsfm_handle h, double ms_sum[3], int64_t launches[3]);
/* The same session's linearisation launches (k_linearize / k_dense_linearize) as the GPU itself timed them: every workgroup
 * stamps s_memrealtime (100 MHz) at its start and end, duration of a launch = latest end - earliest start.  An event pair
 * around a ~10 us kernel reads 2-4 us high (dispatch latency + the event packets); this figure is the one that agrees with
 * rocprofv3's kernel duration.  Call after tcsfm_profile_end; launches beyond the stamp buffer (4 M workgroups per session)
 * are not counted. */
int tcsfm_profile_kernel_time(tcsfm_handle h, double *ms_sum, int64_t *launches);
/* ... and the time during which AT LEAST ONE of those launches was running (the union of their [start, end] intervals over the handle and
 * its lanes, same counter): with launches of several streams sharing the chip, bytes / this = the chip's rate on the kernel. */
int tcsfm_profile_kernel_busy(tcsfm_handle h, double *ms_busy, int64_t *launches);
/* Parity-test hook.  The reference's masks are discontinuous (valid x [diff < auto_err], min over sources; helpers.py:17-19,
 * optimizer.py:47-69) and LM accepts / rejects on a cost comparison: near a tie the fp32 engine and a float64 checker may
 * decide differently.  While a trace is set, every linearisation `lin` of tcsfm_refine* / tcsfm_refine_dense* over N pairs
 * records the engine's decisions into caller-owned DEVICE buffers:
 *   bits   [lin][N][H*W] uint16  bit 0 = the pixel counts (final mask, incl. the min-over-sources selection), bit 1 = warp valid,
 *                                bits 2 / 3 = parity of the bilinear cell floor(ix) / floor(iy) (grid_sample's backward is
 *                                discontinuous across texel boundaries), bits 4-5 = sign of computed - projected depth,
 *                                bits 6-7 / 8-9 / 10-11 = sign of rec_c - tgt_c per colour channel (the signs in the derivatives
 *                                of |cd - pd| and |rec - tgt|; 2-bit codes: 0 exactly zero, 1 positive, 2 negative)
 *   decide [lin][N]      int32   LM: 1 = trial accepted (lin < n_iters) / last step kept (lin == n_iters); GN: 1
 * so that a checker can replay them and compare the continuous arithmetic at full tolerance (tests/test_gpu_parity.py).
 * Capacities in elements; a call that would overflow them returns TCSFM_E_ARG.  Either pointer may be NULL; (NULL, 0, NULL, 0)
 * switches the trace off.  Costs one wave-uniform branch per pixel when off. */
int tcsfm_debug_trace(tcsfm_handle h, uint16_t *bits, int64_t bits_capacity, int32_t *decide, int64_t decide_capacity);
/* Diagnostic builds: 100 MHz wall-clock stamps of the phases of the last k_solve launch of pair 0 (zeros unless the
 * handle was created with TCSFM_DEBUG_STAMPS set in the environment). */
int tcsfm_debug_stamps(tcsfm_handle h, long long out[8]);

/* ---- SE(3) utilities, host, double (replace liegroups.SE3 at data/kitti_loader_stereo.py:129-147,
 *      validate.py:65-71; liegroups is an absent third-party dependency, version unpinned) ---------- */
void tcsfm_pose_to_matrix(const double pose[6], double T[12]);   /* pose_vec2mat(-pose), 3x4 row-major  */
void tcsfm_matrix_to_pose(const double T[12], double pose[6]);
void tcsfm_se3_exp(const double xi[6], double T[12]);            /* xi = [rho, phi]                     */
void tcsfm_se3_log(const double T[12], double xi[6]);
void tcsfm_se3_mul(const double A[12], const double B[12], double C[12]);
void tcsfm_se3_inv(const double A[12], double B[12]);

#ifdef __cplusplus
}
#endif
#endif /* TCSFM_H */
