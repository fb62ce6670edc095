/*
 * tcsfm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's photometric pose/depth residual
 * (utiasSTARS/tightly-coupled-SfM, pure PyTorch) and the float64 Gauss-Newton /
 * Levenberg-Marquardt twin that the HIP engine is checked against.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (tightly_coupled_sfm_amd/) never does.
 *
 * Parity pinning (see tests/golden/make_golden.py, tests/test_oracle_vs_golden.py):
 *   PINNED by outputs of the reference itself, generated in the build container by
 *   importing /root/reference (float64 and float32):
 *     warp (img_rec, valid, projected_depth, computed_depth)     models/stn.py:234-273
 *     SSIM map                                                   losses.py:27-41
 *     residual maps diff/valid/weight/auto-mask                  optimization_experiments/helpers.py:8-23
 *     scalar cost                                                optimization_experiments/plot_loss_surface.py:31-33
 *     d(cost)/d(pose), d(cost)/d(depth) via reference autograd, per-pixel Jacobian rows
 *   UNPINNED by the reference (it has no GN/LM solver):
 *     GN/LM iterates, damping, SE(3) retraction -- this file IS the definition.
 *
 * Compiled twice: -DREAL=double (liboracle_f64.so, the oracle proper) and
 * -DREAL=float (liboracle_f32.so, diagnostic twin with fp32 per-pixel arithmetic).
 * All reductions and the K x K solve are in double in both builds.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define MAXP 7 /* 6 pose + 1 log depth-scale */

/* ------------------------------------------------------------------------- */
/* options (mirrors tcsfm_opts in include/tcsfm.h; kept as plain scalars)      */
typedef struct {
    int nparam;      /* 6: pose; 7: pose + log depth-scale                               */
    int automask;    /* M = valid * (diff < auto_err), helpers.py:18-20                  */
    int param;       /* 0: SE(3) left retraction T <- exp(d) T ; 1: additive on the       */
                     /*    reference's [t, euler] 6-vector (stn.py:143-158)               */
    int solver;      /* 0: Gauss-Newton (fixed damping lambda0) ; 1: Levenberg-Marquardt  */
    int n_iters;
    double w_l1, w_ssim; /* 0.15 / 0.85, train_mono.py:87                                */
    double w_dc;         /* depth-consistency weight, optimizer.py:83-86 (0 = off)       */
    double irls_eps;
    double lambda0, lambda_up, lambda_down, lambda_min;
    double prior_scale; /* Tikhonov weight on (log_scale - initial log_scale)^2: fixes the scale/translation gauge of nparam 7 */
    double w_pose_consist; /* window REFERENCE rule: options['l_pose_consist'] ? 0.1 : 0 -- 0.1 (poses + poses_inv).abs().mean(), optimizer.py:95-96 */
    double w_smooth;       /* dense mode on the reference's loss: options['l_smooth'] ? options['l_smooth_weight'] : 0 -- get_smooth_loss(target disparity, target image), optimizer.py:92-93 */
} orc_opts;

/* ------------------------------------------------------------------------- */
/* small dense helpers (double)                                                */

static void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

/* general 3x3 inverse (torch.inverse at stn.py:257) */
static void mat3_inv(const double *A, double *B) {
    double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    double id = 1.0 / det;
    B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* R = Rx(x) Ry(y) Rz(z), euler2mat stn.py:81-116 */
static void euler_R(const double r[3], double R[9]) {
    double cx = cos(r[0]), sx = sin(r[0]), cy = cos(r[1]), sy = sin(r[1]), cz = cos(r[2]), sz = sin(r[2]);
    double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double t[9];
    mat3_mul(Rx, Ry, t);
    mat3_mul(t, Rz, R);
}

/* T(3x4,row-major) = pose_vec2mat(-pose)  (stn.py:143-158 called with -pose at helpers.py:11, train_mono.py:69) */
void orc_pose_to_T(const double pose[6], double T[12]) {
    double r[3] = {-pose[3], -pose[4], -pose[5]}, R[9];
    euler_R(r, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = -pose[i];
    }
}

void orc_T_to_pose(const double T[12], double pose[6]) {
    double s = T[2];
    if (s > 1) s = 1;
    if (s < -1) s = -1;
    double b = asin(s), a = atan2(-T[6], T[10]), c = atan2(-T[1], T[0]);
    pose[0] = -T[3]; pose[1] = -T[7]; pose[2] = -T[11];
    pose[3] = -a; pose[4] = -b; pose[5] = -c;
}

/* SE(3) exp, xi = [rho(3), phi(3)] translation first (liegroups convention, validate.py:65).
 * T = [exp(phi^) | J_l(phi) rho]  with the SO(3) left Jacobian J_l. */
void orc_se3_exp(const double xi[6], double T[12]) {
    const double *rho = xi, *phi = xi + 3;
    double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2], th = sqrt(th2);
    double A, B, C; /* sin/th, (1-cos)/th^2, (th-sin)/th^3 */
    if (th < 1e-4) {
        A = 1 - th2 / 6 + th2 * th2 / 120; B = 0.5 - th2 / 24 + th2 * th2 / 720; C = 1.0 / 6 - th2 / 120 + th2 * th2 / 5040;
    } else {
        A = sin(th) / th; B = (1 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th);
    }
    double K[9] = {0, -phi[2], phi[1], phi[2], 0, -phi[0], -phi[1], phi[0], 0}, K2[9];
    mat3_mul(K, K, K2);
    for (int i = 0; i < 3; i++) {
        double v = 0;
        for (int j = 0; j < 3; j++) {
            double I = (i == j) ? 1.0 : 0.0;
            T[4 * i + j] = I + A * K[3 * i + j] + B * K2[3 * i + j];
            v += (I + B * K[3 * i + j] + C * K2[3 * i + j]) * rho[j];
        }
        T[4 * i + 3] = v;
    }
}

/* SE(3) log, inverse of orc_se3_exp (|phi| < pi) */
void orc_se3_log(const double T[12], double xi[6]) {
    double tr = T[0] + T[5] + T[10], c = 0.5 * (tr - 1);
    if (c > 1) c = 1;
    if (c < -1) c = -1;
    double th = acos(c);
    double w[3] = {T[9] - T[6], T[2] - T[8], T[4] - T[1]};
    double f = (th < 1e-6) ? 0.5 + th * th / 12 : th / (2 * sin(th));
    double phi[3] = {f * w[0], f * w[1], f * w[2]};
    double th2 = th * th;
    /* J_l^{-1} = I - 1/2 phi^ + D phi^2,  D = 1/th^2 - (1+cos)/(2 th sin) */
    double D = (th < 1e-4) ? 1.0 / 12 + th2 / 720 : 1.0 / th2 - (1 + cos(th)) / (2 * th * sin(th));
    double K[9] = {0, -phi[2], phi[1], phi[2], 0, -phi[0], -phi[1], phi[0], 0}, K2[9];
    mat3_mul(K, K, K2);
    for (int i = 0; i < 3; i++) {
        double v = 0;
        for (int j = 0; j < 3; j++) v += (((i == j) ? 1.0 : 0.0) - 0.5 * K[3 * i + j] + D * K2[3 * i + j]) * T[4 * j + 3];
        xi[i] = v;
        xi[3 + i] = phi[i];
    }
}

/* C = A * B for 3x4 rigid transforms */
void orc_se3_mul(const double A[12], const double B[12], double C[12]) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[4 * i + k] * B[4 * k + j];
            C[4 * i + j] = s;
        }
        double s = A[4 * i + 3];
        for (int k = 0; k < 3; k++) s += A[4 * i + k] * B[4 * k + 3];
        C[4 * i + 3] = s;
    }
}

void orc_se3_inv(const double A[12], double B[12]) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) B[4 * i + j] = A[4 * j + i];
        B[4 * i + 3] = -(A[i] * A[3] + A[4 + i] * A[7] + A[8 + i] * A[11]);
    }
}

/* d(xi_left)/d(pose): T(pose + dp) = exp((A dp)^) T(pose) to first order.
 * With t' = -t, theta = -r, J_e(theta) = [e_x | Rx e_y | Rx Ry e_z]:
 *   dphi = -J_e dr ,  drho = -dt - [t']x J_e dr                                  */
void orc_euler_left_jacobian(const double pose[6], double A[36]) {
    double th[3] = {-pose[3], -pose[4], -pose[5]};
    double cx = cos(th[0]), sx = sin(th[0]), cy = cos(th[1]), sy = sin(th[1]);
    double Je[9] = {1, 0, sy, 0, cx, -sx * cy, 0, sx, cx * cy}; /* columns e_x, Rx e_y, Rx Ry e_z */
    double tp[3] = {-pose[0], -pose[1], -pose[2]};
    double Tx[9] = {0, -tp[2], tp[1], tp[2], 0, -tp[0], -tp[1], tp[0], 0}, TJ[9];
    mat3_mul(Tx, Je, TJ);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++) {
        A[6 * i + i] = -1;
        for (int j = 0; j < 3; j++) {
            A[6 * i + 3 + j] = -TJ[3 * i + j];
            A[6 * (3 + i) + 3 + j] = -Je[3 * i + j];
        }
    }
}

/* in-place Cholesky solve of the n x n SPD system A x = b (row-major, n <= MAXP). returns 0 ok */
static int chol_solve(int n, double *A, double *b) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a1: disp_to_depth, utils/learning_helpers.py:77-86                          */
void orc_disp_to_depth(int n, const real *disp, double min_depth, double max_depth, real *scaled, real *depth) {
    real min_disp = (real)(1.0 / max_depth), max_disp = (real)(1.0 / min_depth);
    for (int i = 0; i < n; i++) {
        real s = min_disp + (max_disp - min_disp) * disp[i];
        if (scaled) scaled[i] = s;
        if (depth) depth[i] = (real)1 / s;
    }
}

/* ------------------------------------------------------------------------- */
/* per-pixel warp geometry, a2-a5                                              */

typedef struct {
    real M[9], m[3];    /* proj_cam_to_src_pixel = K @ [R|t], stn.py:262-264 */
    real K[9], Kinv[9], R[9], t[3];
    real es;            /* exp(log depth-scale) */
    int H, W;
} cam_t;

typedef struct {
    real ix, iy;        /* grid_sample un-normalised sample location                     */
    real Z;             /* computed_depth (clamped), stn.py:215                           */
    real Xp[3];         /* point in the source camera frame                               */
    real p[3];          /* K Xp                                                           */
    int oobx, ooby, zclamp;
    int adjx, adjy;     /* forced replay only: shift of the bilinear cell (-1, 0, +1) to the one the engine sampled */
    int nat_valid;      /* validity as decided HERE, before a forced replay overrides it (flip statistics) */
} geo_t;

static void cam_setup(cam_t *c, int H, int W, const real *K, const double T[12], double log_scale) {
    double Kd[9], Ki[9];
    for (int i = 0; i < 9; i++) Kd[i] = K[i];
    mat3_inv(Kd, Ki);
    for (int i = 0; i < 9; i++) { c->K[i] = K[i]; c->Kinv[i] = (real)Ki[i]; }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) c->R[3 * i + j] = (real)T[4 * i + j];
        c->t[i] = (real)T[4 * i + 3];
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += Kd[3 * i + k] * T[4 * k + j];
            c->M[3 * i + j] = (real)a;
        }
        double a = 0;
        for (int k = 0; k < 3; k++) a += Kd[3 * i + k] * T[4 * k + 3];
        c->m[i] = (real)a;
    }
    c->es = (real)exp(log_scale);
    c->H = H; c->W = W;
}

/* Forced decisions (parity tests only).  The reference's masks are discontinuous in the pose: a pixel whose error ties with its
 * auto-mask threshold, or whose projection lands on the image border, to fp32 rounding can be decided differently by the fp32
 * HIP engine and by this float64 restatement.  The *_forced entry points below replay the ENGINE's decisions -- per pixel and
 * linearisation: bit 0 = the pixel counts (final mask M, including the min-over-sources selection), bit 1 = the warp is valid
 * (stn.py:268-269), bits 2 / 3 = parity of the bilinear cell (floor of the sample coordinate ix / iy: the sample VALUE is
 * continuous across a texel boundary, its derivative -- grid_sample's backward -- is not), bits 4-5 = sign code of cd - pd (the
 * sign in the derivative of the depth-consistency weight, train_mono.py:91), bits 6-7 / 8-9 / 10-11 = sign codes of
 * rec_c - tgt_c of the three colour channels (the sign in the derivative of the L1 term, train_mono.py:87), codes: 0 zero,
 * 1 positive, 2 negative; per linearisation: the LM accept /
 * keep decision -- so that the continuous arithmetic can be compared at the north-star tolerance in every case, while the
 * number of flipped decisions is bounded by a separate assertion. */
static __thread const unsigned short *g_force_bits = NULL; /* [H*W] of the linearisation being evaluated, or NULL: decide here */
static __thread unsigned short *g_record_bits = NULL;      /* [H*W]: record the decisions taken here in the same format (CPU self-test) */

/* Flip statistics of a forced replay (parity tests): while the engine's decisions are replayed, the mask this restatement
 * would have chosen ITSELF at the same (replayed) iterate is compared with the engine's bit 0, per linearisation:
 *   n    = pixels decided differently,
 *   hard = those of them that are NOT explainable as a tie: a different warp validity (border of the valid region), an error
 *          within ORC_TIE of its auto-mask threshold, or -- with the min over the sources -- two sources' errors / the minimum
 *          and the smallest threshold within ORC_TIE of each other.
 * A kernel bug that corrupts masks only after the first pose update shows up here at linearisations 1..n. */
#define ORC_MAX_LIN 64
#define ORC_TIE ((real)5e-5)
static __thread int g_lin_idx = -1;                 /* linearisation being replayed, or -1: no statistics */
static __thread long g_flip_n[ORC_MAX_LIN], g_flip_hard[ORC_MAX_LIN];
static __thread const real *g_sel_margin = NULL;    /* [H*W] window mode: smallest gap among the selection's comparisons at the pixel */
void orc_flip_stats_reset(void) { memset(g_flip_n, 0, sizeof(g_flip_n)); memset(g_flip_hard, 0, sizeof(g_flip_hard)); }
void orc_flip_stats(long *n, long *hard, int count) {
    for (int i = 0; i < count && i < ORC_MAX_LIN; i++) { n[i] = g_flip_n[i]; hard[i] = g_flip_hard[i]; }
}
static inline void flip_note(real m_nat, int forced, int is_tie) {
    if (g_lin_idx < 0 || g_lin_idx >= ORC_MAX_LIN || (m_nat != 0) == (forced != 0)) return;
    g_flip_n[g_lin_idx]++;
    if (!is_tie) g_flip_hard[g_lin_idx]++;
}

/* sign of a quantity whose sign is a discrete decision of the residual's derivative (L1 term: rec - tgt of a channel; depth-
 * consistency term: cd - pd), kept as a 2-bit code at `shift`: 0 = exactly zero, 1 = positive, 2 = negative (an exact zero is not
 * rare in fp32: consistent depth maps give cd == pd bit for bit).  Replay: within `tie` of zero the engine's recorded sign wins. */
static inline real forced_sign(real x, real tie, int i, int shift) {
    real sgn = x > 0 ? (real)1 : (x < 0 ? (real)-1 : (real)0);
    if (g_force_bits && fabs(x) < tie) { const int c = (g_force_bits[i] >> shift) & 3; sgn = c == 1 ? (real)1 : (c == 2 ? (real)-1 : (real)0); }
    if (g_record_bits) g_record_bits[i] = (unsigned short)((g_record_bits[i] & ~(3 << shift)) | ((sgn > 0 ? 1 : (sgn < 0 ? 2 : 0)) << shift));
    return sgn;
}

/* pixel2cam stn.py:33-48, pose_vec2mat stn.py:143-158, cam2pixel2 stn.py:198-231,
 * grid un-normalisation of F.grid_sample(align_corners=False) stn.py:266 */
static void warp_geo(const cam_t *c, int u, int v, real depth, geo_t *g) {
    const real *Ki = c->Kinv, *R = c->R, *M = c->M;
    real D = c->es * depth;
    real ray[3], X[3];
    /* same operation order as the reference: (Kinv @ pix) * depth, then (K@R) @ X + K@t */
    for (int i = 0; i < 3; i++) ray[i] = Ki[3 * i] * (real)u + Ki[3 * i + 1] * (real)v + Ki[3 * i + 2];
    for (int i = 0; i < 3; i++) X[i] = ray[i] * D;
    for (int i = 0; i < 3; i++) g->p[i] = M[3 * i] * X[0] + M[3 * i + 1] * X[1] + M[3 * i + 2] * X[2] + c->m[i];
    /* source-frame point, needed by the Jacobian only */
    for (int i = 0; i < 3; i++) g->Xp[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + c->t[i];
    g->zclamp = g->p[2] < (real)1e-3;
    g->Z = g->zclamp ? (real)1e-3 : g->p[2];
    real xn = 2 * (g->p[0] / g->Z) / (real)(c->W - 1) - 1;
    real yn = 2 * (g->p[1] / g->Z) / (real)(c->H - 1) - 1;
    g->oobx = (xn > 1) || (xn < -1);
    g->ooby = (yn > 1) || (yn < -1);
    g->nat_valid = !(g->oobx || g->ooby);
    if (g_force_bits) { /* replay the engine's validity decision (the sample of an invalid pixel is zero as a whole) */
        const int valid = (g_force_bits[v * c->W + u] >> 1) & 1;
        if (valid) g->oobx = g->ooby = 0;
        else if (!(g->oobx || g->ooby)) g->oobx = g->ooby = 1;
    }
    if (g_record_bits) g_record_bits[v * c->W + u] = (unsigned short)((g_record_bits[v * c->W + u] & ~2) | ((g->oobx || g->ooby) ? 0 : 2));
    if (g->oobx) xn = 2; /* stn.py:223-227: OOB sentinel, detached */
    if (g->ooby) yn = 2;
    g->ix = ((xn + 1) * (real)c->W - 1) / 2;
    g->iy = ((yn + 1) * (real)c->H - 1) / 2;
    g->adjx = g->adjy = 0;
    if (!(g->oobx || g->ooby)) {
        const int cx = (int)floor(g->ix), cy = (int)floor(g->iy);
        if (g_record_bits) g_record_bits[v * c->W + u] = (unsigned short)((g_record_bits[v * c->W + u] & ~12) | ((cx & 1) << 2) | ((cy & 1) << 3));
        if (g_force_bits) { /* a sample within 1e-4 px of a texel boundary follows the engine's side of it */
            const int b = g_force_bits[v * c->W + u];
            const real rx = g->ix - floor(g->ix + (real)0.5), ry = g->iy - floor(g->iy + (real)0.5);
            if (((cx & 1) != ((b >> 2) & 1)) && fabs(rx) < (real)1e-4) g->adjx = rx >= 0 ? -1 : 1;
            if (((cy & 1) != ((b >> 3) & 1)) && fabs(ry) < (real)1e-4) g->adjy = ry >= 0 ? -1 : 1;
        }
    }
}

/* bilinear tap with zero padding: value and d/dix, d/diy (grid_sampler_2d fwd/bwd semantics) */
static void bilinear_cell(const real *img, int H, int W, real ix, real iy, int adjx, int adjy, real *val, real *gx, real *gy) {
    real fx = floor(ix) + (real)adjx, fy = floor(iy) + (real)adjy;   /* adj != 0: the neighbouring cell, weights extrapolated by <= 1e-4 */
    real wx = ix - fx, wy = iy - fy;
    /* far-out sentinel coordinates: avoid int overflow */
    if (!(fx > -4 && fx < W + 4 && fy > -4 && fy < H + 4)) { *val = 0; if (gx) { *gx = 0; *gy = 0; } return; }
    int x0 = (int)fx, y0 = (int)fy;
    real v00 = 0, v01 = 0, v10 = 0, v11 = 0;
    if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v00 = img[y0 * W + x0];
        if (x0 + 1 >= 0 && x0 + 1 < W) v01 = img[y0 * W + x0 + 1];
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v10 = img[(y0 + 1) * W + x0];
        if (x0 + 1 >= 0 && x0 + 1 < W) v11 = img[(y0 + 1) * W + x0 + 1];
    }
    *val = (1 - wx) * (1 - wy) * v00 + wx * (1 - wy) * v01 + (1 - wx) * wy * v10 + wx * wy * v11;
    if (gx) {
        *gx = (1 - wy) * (v01 - v00) + wy * (v11 - v10);
        *gy = (1 - wx) * (v10 - v00) + wx * (v11 - v01);
    }
}

static void bilinear(const real *img, int H, int W, real ix, real iy, real *val, real *gx, real *gy) {
    bilinear_cell(img, H, W, ix, iy, 0, 0, val, gx, gy);
}

/* inverse_warp2, stn.py:234-273.  src [3,H,W]; depth_t, depth_s [H,W]; T 3x4; K 3x3.
 * Outputs (any may be NULL): rec [3,H,W], valid [H,W], proj_depth [H,W], comp_depth [H,W]. */
void orc_warp(int H, int W, const real *src, const real *depth_t, const real *depth_s, const double T[12],
              const real *K, double log_scale, real *rec, real *valid, real *proj_depth, real *comp_depth) {
    cam_t c;
    cam_setup(&c, H, W, K, T, log_scale);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            geo_t g;
            int i = v * W + u;
            warp_geo(&c, u, v, depth_t[i], &g);
            for (int ch = 0; ch < 3; ch++) {
                real val;
                bilinear(src + ch * H * W, H, W, g.ix, g.iy, &val, NULL, NULL);
                if (rec) rec[ch * H * W + i] = val;
            }
            if (valid) valid[i] = (g.oobx || g.ooby) ? 0 : 1;
            if (proj_depth) {
                real val;
                bilinear(depth_s, H, W, g.ix, g.iy, &val, NULL, NULL);
                proj_depth[i] = c.es * val;
            }
            if (comp_depth) comp_depth[i] = g.Z;
        }
}

/* diagnostic: the bilinear sample positions (grid_sample's un-normalised ix, iy) of every target pixel */
void orc_sample_positions(int H, int W, const real *depth_t, const double T[12], const real *K, double log_scale, real *ix, real *iy) {
    cam_t c;
    cam_setup(&c, H, W, K, T, log_scale);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            geo_t g;
            warp_geo(&c, u, v, depth_t[v * W + u], &g);
            ix[v * W + u] = g.ix; iy[v * W + u] = g.iy;
        }
}

/* ReflectionPad2d(1) index, losses.py:22 */
static inline int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

#define SSIM_C1 ((real)(0.01 * 0.01))
#define SSIM_C2 ((real)(0.03 * 0.03))

typedef struct { real mux, muy, n1, n2, d1, d2, s; int clamped; } ssim_t;

/* SSIM_Loss.forward at one pixel/channel, losses.py:27-41 */
static void ssim_at(const real *x, const real *y, int H, int W, int u, int v, ssim_t *o) {
    real sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int dv = -1; dv <= 1; dv++)
        for (int du = -1; du <= 1; du++) {
            int j = refl(v + dv, H) * W + refl(u + du, W);
            real a = x[j], b = y[j];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    const real ninth = (real)1 / 9;
    real mux = sx * ninth, muy = sy * ninth;
    real sigx = sxx * ninth - mux * mux, sigy = syy * ninth - muy * muy, sigxy = sxy * ninth - mux * muy;
    o->mux = mux; o->muy = muy;
    o->n1 = 2 * mux * muy + SSIM_C1; o->n2 = 2 * sigxy + SSIM_C2;
    o->d1 = mux * mux + muy * muy + SSIM_C1; o->d2 = sigx + sigy + SSIM_C2;
    real raw = (1 - (o->n1 * o->n2) / (o->d1 * o->d2)) / 2;
    o->clamped = (raw < 0) || (raw > 1);
    o->s = raw < 0 ? 0 : (raw > 1 ? 1 : raw);
}

/* SSIM map over C planes, losses.py:27-41 */
void orc_ssim(int C, int H, int W, const real *x, const real *y, real *out) {
    for (int c = 0; c < C; c++)
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                ssim_t s;
                ssim_at(x + c * H * W, y + c * H * W, H, W, u, v, &s);
                out[c * H * W + v * W + u] = s.s;
            }
}

static inline real clamp01(real a) { return a < 0 ? 0 : (a > 1 ? 1 : a); }

/* (0.15 |a-b|.clamp(0,1) + 0.85 SSIM(tgt=b... ) ).mean(1): train_mono.py:84,87 / helpers.py:12,17
 * x = reconstruction target, y = other image; returns mean over the 3 channels. */
static void photo_err_map(int H, int W, const real *x, const real *y, double w_l1, double w_ssim, real *out) {
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            real acc = 0;
            for (int c = 0; c < 3; c++) {
                ssim_t s;
                ssim_at(x + c * H * W, y + c * H * W, H, W, u, v, &s);
                real l1 = clamp01(fabs(y[c * H * W + v * W + u] - x[c * H * W + v * W + u]));
                acc += (real)w_l1 * l1 + (real)w_ssim * s.s;
            }
            out[v * W + u] = acc / 3;
        }
}

/* compute_photometric_error, helpers.py:8-23 (single directed pair) == the per-pair slice of
 * solve_pose_iteratively's residual assembly, train_mono.py:82-100.
 * Outputs (any may be NULL): diff, valid (warp validity, stn.py:268-269), weight, auto_err, auto_mask, rec[3HW]. */
void orc_photometric(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                     const double T[12], const real *K, double log_scale, double w_l1, double w_ssim,
                     real *diff, real *valid, real *weight, real *auto_err, real *auto_mask, real *rec_out) {
    int n = H * W;
    real *rec = (real *)malloc(sizeof(real) * 3 * n), *pd = (real *)malloc(sizeof(real) * n),
         *cd = (real *)malloc(sizeof(real) * n), *d = (real *)malloc(sizeof(real) * n), *ae = (real *)malloc(sizeof(real) * n);
    orc_warp(H, W, src, depth_t, depth_s, T, K, log_scale, rec, valid, pd, cd);
    photo_err_map(H, W, tgt, rec, w_l1, w_ssim, d);
    photo_err_map(H, W, tgt, src, w_l1, w_ssim, ae);
    for (int i = 0; i < n; i++) {
        if (diff) diff[i] = d[i];
        if (weight) weight[i] = 1 - clamp01(fabs(cd[i] - pd[i]) / (cd[i] + pd[i])); /* helpers.py:13-14 */
        if (auto_err) auto_err[i] = ae[i];
        if (auto_mask) auto_mask[i] = d[i] < ae[i] ? 1 : 0;                             /* helpers.py:18 */
    }
    if (rec_out) memcpy(rec_out, rec, sizeof(real) * 3 * n);
    free(rec); free(pd); free(cd); free(d); free(ae);
}

/* ------------------------------------------------------------------------- */
/* linearisation: cost, gradient, Gauss-Newton matrix                          */

typedef struct {
    real rec[3], gx[3], gy[3]; /* warped source + d/d(ix,iy)                               */
    real pd, dgx, dgy, cd;     /* projected (sampled, scaled) / computed depth              */
    real a[MAXP], b[MAXP], zc[MAXP], dpd[MAXP]; /* d ix, d iy, d cd, d pd  w.r.t. parameters */
    int valid, nat_valid;
    int dc_in;                 /* the four taps of the projected-depth sample are real source pixels (none is zero padding) */
} px_t;

/* Parameters: xi = [rho, phi] left perturbation of T (T <- exp(xi^) T), optional 7th = log depth-scale
 * applied to BOTH depth maps.  dXp/drho_j = e_j ; dXp/dphi_j = e_j x Xp ; dXp/dsigma = Xp - t. */
static void px_jac(const cam_t *c, const geo_t *g, int np, px_t *o) {
    real cw = (real)c->W / (real)(c->W - 1), chh = (real)c->H / (real)(c->H - 1);
    const real *Xp = g->Xp;
    real dX[MAXP][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, -Xp[2], Xp[1]}, {Xp[2], 0, -Xp[0]}, {-Xp[1], Xp[0], 0},
                        {Xp[0] - c->t[0], Xp[1] - c->t[1], Xp[2] - c->t[2]}};
    real uz = g->p[0] / g->Z, vz = g->p[1] / g->Z;
    for (int j = 0; j < np; j++) {
        real dp[3];
        for (int i = 0; i < 3; i++) dp[i] = c->K[3 * i] * dX[j][0] + c->K[3 * i + 1] * dX[j][1] + c->K[3 * i + 2] * dX[j][2];
        real dZ = g->zclamp ? 0 : dp[2];
        o->zc[j] = dZ;
        o->a[j] = g->oobx ? 0 : cw * (dp[0] - uz * dZ) / g->Z;
        o->b[j] = g->ooby ? 0 : chh * (dp[1] - vz * dZ) / g->Z;
    }
}

static void px_eval(const cam_t *c, const real *src, const real *depth_t, const real *depth_s, int u, int v, int np, px_t *o) {
    geo_t g;
    int H = c->H, W = c->W;
    warp_geo(c, u, v, depth_t[v * W + u], &g);
    for (int ch = 0; ch < 3; ch++) bilinear_cell(src + ch * H * W, H, W, g.ix, g.iy, g.adjx, g.adjy, &o->rec[ch], &o->gx[ch], &o->gy[ch]);
    real dval;
    bilinear_cell(depth_s, H, W, g.ix, g.iy, g.adjx, g.adjy, &dval, &o->dgx, &o->dgy);
    o->pd = c->es * dval; o->dgx *= c->es; o->dgy *= c->es;
    {   /* (the cell the sample was actually taken from: a forced replay may have moved it across a texel boundary) */
        const real fx = floor(g.ix) + (real)g.adjx, fy = floor(g.iy) + (real)g.adjy;
        o->dc_in = !(g.oobx || g.ooby) && fx >= 0 && fx + 1 <= (real)(W - 1) && fy >= 0 && fy + 1 <= (real)(H - 1);
    }
    o->cd = g.Z;
    o->valid = !(g.oobx || g.ooby);
    o->nat_valid = g.nat_valid;
    px_jac(c, &g, np, o);
    for (int j = 0; j < np; j++) o->dpd[j] = o->dgx * o->a[j] + o->dgy * o->b[j] + ((j == 6) ? o->pd : 0);
}

/* diagnostic: quantities at ONE pixel that sit next to a discontinuity of the residual's derivative:
 * out = [ix, iy, cd - pd, (cd-pd)/(cd+pd), valid, then per channel c: rec_c - tgt_c, raw SSIM value before the clamp] */
void orc_pixel_debug(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s, const double T[12],
                     const real *K, int u, int v, double *out) {
    int n = H * W;
    cam_t c;
    cam_setup(&c, H, W, K, T, 0.0);
    real *rec = (real *)malloc(sizeof(real) * 3 * n);
    px_t P0;
    memset(&P0, 0, sizeof(P0));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            px_t P;
            px_eval(&c, src, depth_t, depth_s, x, y, 6, &P);
            for (int ch = 0; ch < 3; ch++) rec[ch * n + y * W + x] = P.rec[ch];
            if (x == u && y == v) P0 = P;
        }
    geo_t g;
    warp_geo(&c, u, v, depth_t[v * W + u], &g);
    out[0] = g.ix; out[1] = g.iy; out[2] = P0.cd - P0.pd; out[3] = (P0.cd - P0.pd) / (P0.cd + P0.pd); out[4] = P0.valid;
    for (int ch = 0; ch < 3; ch++) {
        ssim_t s;
        ssim_at(tgt + ch * n, rec + ch * n, H, W, u, v, &s);
        out[5 + 2 * ch] = rec[ch * n + v * W + u] - tgt[ch * n + v * W + u];
        out[6 + 2 * ch] = (1 - (s.n1 * s.n2) / (s.d1 * s.d2)) / 2;
    }
    free(rec);
}

/* outputs of one linearisation */
typedef struct {
    double H[MAXP * MAXP], g[MAXP];
    double cost, cost_photo, cost_dc, n_mask;
} lin_t;

/*
 * Cost (per directed pair), the reference's masked mean (plot_loss_surface.py:31-33, optimizer.py:69,79)
 * plus the optional depth-consistency mean (optimizer.py:83-86):
 *     C = sum_p M_p W_p diff_p / sum_p M_p  +  w_dc * mean_p dd_p
 *   diff = e1 + e2,  e1 = w_l1/3 sum_c clamp|rec-tgt|,  e2 = w_ssim/3 sum_c SSIM_c      (train_mono.py:87)
 *   dd   = clamp(|cd-pd|/(cd+pd),0,1), W = 1-dd                                          (train_mono.py:91-92)
 *   M    = valid * [diff < auto_err]  (non-differentiable, as in the reference)           (helpers.py:18-20)
 * Per-pixel error maps (E_k >= 0):  E1 = W e1, E2 = W e2, E3 = dd, with rows J_k = dE_k/dtheta.
 *     g = sum a (J1+J2) + b h J3         == EXACT gradient of C (masks detached, as reference autograd) for the
 *                                           photometric part; h = min(1, E3/eps) Huberises the depth-consistency part
 *   with a = M/sum(M), b = w_dc/(H W).
 * Gauss-Newton matrix (a generalised GN: exact gradient, PSD curvature model).  With the 2 x np
 * geometric Jacobian  Jg_p = [d ix/dtheta ; d iy/dtheta]  of the sample position and the bilinear image
 * gradient  gr_c = [d rec_c/d ix, d rec_c/d iy]:
 *     H = sum_p a_p Jg_p' Lam_p Jg_p  +  b J3 J3'/max(E3,eps)
 *     Lam_p = W_p sum_c {  w_l1/3 * gr_c gr_c' / max(|r_c|, eps)                 (IRLS for the L1 term)
 *                        + w_ssim/3 * ( Cov_3x3(gr_c)/d2_c + mean_3x3(gr_c) mean_3x3(gr_c)'/d1_c ) }
 *   where Cov_3x3 is taken as its centre-sample estimate 9/8 (gr_c - mean)(gr_c - mean)' (always PSD; summed over
 *   the pixels it equals the window covariance up to boundary effects).
 *   The SSIM part is the Gauss-Newton matrix of the exact decomposition
 *     1 - l  = (mu_x-mu_y)^2/d1 ,  1 - cs = Var_3x3(x-y)/d2 ,  SSIM loss = (1 - l cs)/2
 *   (d1, d2 = the two SSIM denominators, losses.py:38) with d1, d2 and the geometry frozen over the 3x3 window.
 *   Curvature of the W factor (e dW/dtheta) is neglected in H; it is present in g.
 * J1/J2/J3/E (optional, [H*W*np] / [H*W*3]) return the per-pixel rows for Jacobian pinning.
 */
/* Coupling of one directed pair to the rest of its window under the REFERENCE window rule (compute_optimization_loss,
 * optimizer.py:47-86 -- see orc_refine_window_rule): all fields optional. */
typedef struct {
    const real *w_map;  /* [H*W] weight map replacing the pair's OWN depth-consistency weight W in the photometric term, a constant
                           of this pair's pose (optimizer.py:69: every source's minimum is multiplied by the weight map of source 0) */
    const real *cross;  /* [H*W] extra gradient  -a cross(p) d dd_p/d theta : source 0's weight map also multiplies the pixels the
                           OTHER sources won, cross(p) = sum_{s != 0} M_s(p) diff_s(p) */
    double norm;        /* > 0: normaliser of the photometric sum (the batch-summed mask count) instead of the pair's own sum M */
    double scale;       /* > 0: factor on the photometric term (0.25 for the inverse pairs, optimizer.py:79) */
    double b_dc;        /* >= 0: per-pixel weight of the depth-consistency term instead of w_dc / (H W) */
    int no_automask;    /* the pair's own mask is the warp validity alone (forward pairs without argmin, optimizer.py:71-73) */
} lin_ext;

static void linearize_masked(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                   const double T[12], const real *K, double log_scale, const orc_opts *op, const real *auto_err_in,
                   const real *mask_in, const lin_ext *x, lin_t *out, real *J1o, real *J2o, real *J3o, real *Eo, real *Mo) {
    int n = H * W, np = op->nparam;
    cam_t c;
    cam_setup(&c, H, W, K, T, log_scale);
    px_t *px = (px_t *)malloc(sizeof(px_t) * n);
    real *ae = (real *)malloc(sizeof(real) * n);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) px_eval(&c, src, depth_t, depth_s, u, v, np, &px[v * W + u]);
    if (auto_err_in) memcpy(ae, auto_err_in, sizeof(real) * n);
    else photo_err_map(H, W, tgt, src, op->w_l1, op->w_ssim, ae);

    real *rec = (real *)malloc(sizeof(real) * 3 * n);
    for (int i = 0; i < n; i++)
        for (int ch = 0; ch < 3; ch++) rec[ch * n + i] = px[i].rec[ch];

    real *J1 = (real *)calloc((size_t)n * np, sizeof(real)), *J2 = (real *)calloc((size_t)n * np, sizeof(real)),
         *J3 = (real *)calloc((size_t)n * np, sizeof(real));
    real *E = (real *)calloc((size_t)n * 3, sizeof(real)), *M = (real *)calloc(n, sizeof(real));
    real *Lam = (real *)calloc((size_t)n * 3, sizeof(real)); /* lxx, lxy, lyy */
    const real reps = (real)op->irls_eps;
    const real wl = (real)(op->w_l1 / 3), ws = (real)(op->w_ssim / 3);
    double nmask = 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            int i = v * W + u;
            const px_t *P = &px[i];
            /* depth consistency, train_mono.py:91-92 */
            real sum = P->cd + P->pd, dif = P->cd - P->pd;
            real raw = fabs(dif) / sum;
            real dd = clamp01(raw), Wown = 1 - dd;
            const int wext = x && x->w_map;
            const real Wt = wext ? x->w_map[i] : Wown;   /* weight of the photometric term */
            real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
            real ddJ[MAXP];
            for (int j = 0; j < np; j++) ddJ[j] = sg * 2 * (P->pd * P->zc[j] - P->cd * P->dpd[j]) / (sum * sum);
            /* photometric rows */
            real e1 = 0, e2 = 0, de1[MAXP] = {0}, de2[MAXP] = {0};
            for (int ch = 0; ch < 3; ch++) {
                const real *x = tgt + ch * n, *y = rec + ch * n;
                real r = y[i] - x[i], ar = fabs(r);
                e1 += wl * clamp01(ar);
                real sgn = (ar <= 1) ? forced_sign(r, (real)1e-6, i, 6 + 2 * ch) : (real)0;
                for (int j = 0; j < np; j++) de1[j] += wl * sgn * (P->gx[ch] * P->a[j] + P->gy[ch] * P->b[j]);
                if (ar <= 1) { /* IRLS curvature of the L1 term */
                    real w1 = wl * Wt / (ar > reps ? ar : reps);
                    Lam[3 * i] += w1 * P->gx[ch] * P->gx[ch]; Lam[3 * i + 1] += w1 * P->gx[ch] * P->gy[ch];
                    Lam[3 * i + 2] += w1 * P->gy[ch] * P->gy[ch];
                }
                ssim_t s;
                ssim_at(x, y, H, W, u, v, &s);
                e2 += ws * s.s;
                if (!s.clamped) {
                    /* d s / d y_q = cA + cB y_q + cC x_q  (q in the reflect-padded 3x3 window) */
                    real nn = s.n1 * s.n2, dn = s.d1 * s.d2, ratio = nn / dn;
                    real pre = -(real)0.5 / dn / 9;
                    real cA = pre * (2 * s.mux * s.n2 - 2 * s.n1 * s.mux - ratio * (2 * s.muy * s.d2 - 2 * s.d1 * s.muy));
                    real cB = pre * (-ratio * 2 * s.d1);
                    real cC = pre * (2 * s.n1);
                    real Sx = 0, Sy = 0;
                    for (int dv = -1; dv <= 1; dv++)
                        for (int du = -1; du <= 1; du++) {
                            int q = refl(v + dv, H) * W + refl(u + du, W);
                            const px_t *Q = &px[q];
                            real cf = ws * (cA + cB * y[q] + cC * x[q]);
                            for (int j = 0; j < np; j++) de2[j] += cf * (Q->gx[ch] * Q->a[j] + Q->gy[ch] * Q->b[j]);
                            Sx += Q->gx[ch]; Sy += Q->gy[ch];
                        }
                    /* GN curvature of the SSIM term: Cov_3x3(gr)/d2 + mean mean'/d1, with the window covariance
                     * replaced by its centre-sample estimate 9/8 (gr_p - mean)(gr_p - mean)' (PSD, one outer product) */
                    const real ninth = (real)1 / 9;
                    real mx = Sx * ninth, my = Sy * ninth;
                    real ex = P->gx[ch] - mx, ey = P->gy[ch] - my;
                    real w2 = ws * Wt / s.d2 * (real)1.125, w3 = ws * Wt / s.d1;
                    Lam[3 * i] += w2 * ex * ex + w3 * mx * mx;
                    Lam[3 * i + 1] += w2 * ex * ey + w3 * mx * my;
                    Lam[3 * i + 2] += w2 * ey * ey + w3 * my * my;
                }
            }
            real diff = e1 + e2;
            const int am_on = op->automask && !(x && x->no_automask);
            real m = (real)P->nat_valid;
            if (am_on) m *= (diff < ae[i]) ? (real)1 : (real)0;
            if (mask_in) m = mask_in[i];   /* window mode: the per-pixel min-over-sources selection replaces the pair's own mask */
            if (g_force_bits) {
                const int fm = g_force_bits[i] & 1;
                flip_note(m, fm, mask_in ? (g_sel_margin ? g_sel_margin[i] < ORC_TIE : 1)
                                         : (P->nat_valid != P->valid) || (am_on && fabs(diff - ae[i]) < ORC_TIE));
                m = (real)fm;
            } else if (!mask_in) {
                m = (real)P->valid * (am_on ? ((diff < ae[i]) ? (real)1 : (real)0) : (real)1);
            }
            if (g_record_bits) g_record_bits[i] = (unsigned short)((g_record_bits[i] & ~1) | (m != 0 ? 1 : 0));
            M[i] = m; nmask += m;
            E[3 * i] = Wt * e1; E[3 * i + 1] = Wt * e2; E[3 * i + 2] = dd;
            for (int j = 0; j < np; j++) {   /* a foreign weight map does not depend on this pair's pose: no e dW term */
                J1[i * np + j] = Wt * de1[j] - (wext ? 0 : e1 * ddJ[j]);
                J2[i * np + j] = Wt * de2[j] - (wext ? 0 : e2 * ddJ[j]);
                J3[i * np + j] = ddJ[j];
            }
        }
    memset(out, 0, sizeof(*out));
    out->n_mask = nmask;
    const double den = (x && x->norm > 0) ? x->norm : nmask, sc = (x && x->scale > 0) ? x->scale : 1.0;
    double a = den > 0 ? sc / den : 0.0, b = (x && x->b_dc >= 0) ? x->b_dc : op->w_dc / (double)n, eps = op->irls_eps;
    for (int i = 0; i < n; i++) {
        double am = a * M[i];
        if (x && x->cross)
            for (int j = 0; j < np; j++) out->g[j] -= a * (double)x->cross[i] * J3[i * np + j];
        double E1 = E[3 * i], E2 = E[3 * i + 1], E3 = E[3 * i + 2];
        out->cost_photo += am * (E1 + E2);
        out->cost_dc += b * E3;
        /* IRLS curvature of the depth-consistency term -- over the pixels whose projected depth is a REAL depth sample only:
         * where the bilinear footprint touches grid_sample's zero padding (stn.py:271, valid stays 1 up to half a pixel outside)
         * the "projected depth" is a blend with 0, dd jumps from ~0 to 1 within one pixel of sample position and 1/max(dd,eps) x
         * (depth / 1 px)^2 from a handful of such border pixels would make up 90 % of this term's curvature (measured), at the
         * mercy of the fifth digit of their sample positions.  The GRADIENT keeps every pixel (it is the reference's autograd). */
        double k3 = px[i].dc_in ? b / fmax(E3, eps) : 0.0;
        double lxx = am * Lam[3 * i], lxy = am * Lam[3 * i + 1], lyy = am * Lam[3 * i + 2];
        const px_t *P = &px[i];
        for (int j = 0; j < np; j++) {
            double j1 = J1[i * np + j], j2 = J2[i * np + j], j3 = J3[i * np + j];
            /* depth-consistency gradient: Huber inside |dd| < eps (dd/eps instead of 1).  Where the two depth maps agree
             * to rounding, sign(cd - pd) is numerical noise -- in the reference's autograd too -- and an exact-L1 gradient is
             * not reproducible across precisions.  Same region where the IRLS curvature 1/max(dd,eps) is already quadratic. */
            out->g[j] += am * (j1 + j2) + b * fmin(1.0, E3 / eps) * j3;
            double la = lxx * P->a[j] + lxy * P->b[j], lb = lxy * P->a[j] + lyy * P->b[j];
            for (int k = 0; k <= j; k++)
                out->H[j * np + k] += la * P->a[k] + lb * P->b[k] + k3 * j3 * J3[i * np + k];
        }
    }
    for (int j = 0; j < np; j++)
        for (int k = j + 1; k < np; k++) out->H[j * np + k] = out->H[k * np + j];
    out->cost = out->cost_photo + out->cost_dc;
    if (J1o) memcpy(J1o, J1, sizeof(real) * n * np);
    if (J2o) memcpy(J2o, J2, sizeof(real) * n * np);
    if (J3o) memcpy(J3o, J3, sizeof(real) * n * np);
    if (Eo) memcpy(Eo, E, sizeof(real) * n * 3);
    if (Mo) memcpy(Mo, M, sizeof(real) * n);
    free(px); free(ae); free(rec); free(J1); free(J2); free(J3); free(E); free(M); free(Lam);
}

void orc_linearize(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                   const double T[12], const real *K, double log_scale, const orc_opts *op, const real *auto_err_in,
                   lin_t *out, real *J1o, real *J2o, real *J3o, real *Eo, real *Mo) {
    linearize_masked(H, W, tgt, src, depth_t, depth_s, T, K, log_scale, op, auto_err_in, NULL, NULL, out, J1o, J2o, J3o, Eo, Mo);
}

/* scalar cost only: the quantity generate_loss_surface sweeps, plot_loss_surface.py:31-33,45-47 */
double orc_cost(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                const double T[12], const real *K, double log_scale, const orc_opts *op) {
    int n = H * W;
    real *d = (real *)malloc(sizeof(real) * n), *va = (real *)malloc(sizeof(real) * n), *w = (real *)malloc(sizeof(real) * n),
         *am = (real *)malloc(sizeof(real) * n);
    orc_photometric(H, W, tgt, src, depth_t, depth_s, T, K, log_scale, op->w_l1, op->w_ssim, d, va, w, NULL, am, NULL);
    double num = 0, den = 0, dc = 0;
    for (int i = 0; i < n; i++) {
        double m = va[i] * (op->automask ? am[i] : 1);
        if (g_record_bits) g_record_bits[i] = (unsigned short)((g_record_bits[i] & ~1) | (m != 0 ? 1 : 0));
        num += m * w[i] * d[i]; den += m; dc += 1 - w[i];
    }
    free(d); free(va); free(w); free(am);
    return (den > 0 ? num / den : 0.0) + op->w_dc * dc / n;
}

/* ------------------------------------------------------------------------- */
/* GN / LM step and loop (float64; this is the DEFINITION the HIP solve kernel is checked against) */

/* Solve (H + lambda diag(H) + 1e-12 I) d = -g ; returns 0 ok. Marquardt scaling keeps the step
 * invariant to the translation/rotation unit mismatch. */
int orc_solve_step(int np, const double *Hm, const double *g, double lambda, double *delta) {
    double A[MAXP * MAXP], b[MAXP];
    for (int i = 0; i < np; i++) {
        for (int j = 0; j < np; j++) A[i * np + j] = Hm[i * np + j];
        A[i * np + i] += lambda * Hm[i * np + i] + 1e-12;
        b[i] = -g[i];
    }
    if (chol_solve(np, A, b)) { memset(delta, 0, sizeof(double) * np); return -1; }
    memcpy(delta, b, sizeof(double) * np);
    return 0;
}

/* apply a solved step to the state (T, pose6, log_scale) under the chosen parameterisation */
static void apply_step(const orc_opts *op, const double *Hm, const double *g, double lambda,
                       const double Tin[12], double s_in, double Tout[12], double *s_out) {
    int np = op->nparam;
    double delta[MAXP];
    if (op->param == 0) {
        orc_solve_step(np, Hm, g, lambda, delta);
        double E[12];
        orc_se3_exp(delta, E);
        orc_se3_mul(E, Tin, Tout);
    } else {
        /* additive on the reference's 6-vector: J_pose = J_xi A  =>  H_p = A' H A, g_p = A' g */
        double pose[6], A[36], Hp[MAXP * MAXP], gp[MAXP], Af[MAXP * MAXP];
        orc_T_to_pose(Tin, pose);
        orc_euler_left_jacobian(pose, A);
        memset(Af, 0, sizeof(Af));
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) Af[i * np + j] = A[6 * i + j];
        if (np == 7) Af[6 * np + 6] = 1;
        for (int i = 0; i < np; i++) {
            gp[i] = 0;
            for (int k = 0; k < np; k++) gp[i] += Af[k * np + i] * g[k];
            for (int j = 0; j < np; j++) {
                double s = 0;
                for (int k = 0; k < np; k++)
                    for (int l = 0; l < np; l++) s += Af[k * np + i] * Hm[k * np + l] * Af[l * np + j];
                Hp[i * np + j] = s;
            }
        }
        orc_solve_step(np, Hp, gp, lambda, delta);
        for (int i = 0; i < 6; i++) pose[i] += delta[i];
        orc_pose_to_T(pose, Tout);
    }
    *s_out = s_in + (np == 7 ? delta[6] : 0.0);
}

/*
 * Refine one directed pair.  pose_io: reference 6-vector in/out.  log_scale_io in/out (ignored for nparam 6).
 * stats (optional, [n_iters+1][4]): cost, cost_photo, n_mask, lambda at each linearisation point.
 *
 * GN  (solver 0): n_iters x { linearise at T ; T <- step(T) } with fixed damping lambda0.
 * LM  (solver 1): n_iters linearisations with accept/reject on the cost, then one cost-only
 *                 evaluation deciding whether the last step is kept.
 */
static double cost_masked(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                          const double T[12], const real *K, double log_scale, const orc_opts *op, const real *mask);

/* cost of one pair under the engine's decisions `bits` (uint16 per pixel, see g_force_bits) */
static double cost_forced(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                          const double T[12], const real *K, double log_scale, const orc_opts *op, const unsigned short *bits) {
    int n = H * W;
    real *mk = (real *)malloc(sizeof(real) * n);
    for (int i = 0; i < n; i++) mk[i] = (real)(bits[i] & 1);
    g_force_bits = bits;
    double c = cost_masked(H, W, tgt, src, depth_t, depth_s, T, K, log_scale, op, mk);
    g_force_bits = NULL;
    free(mk);
    return c;
}

static void refine_impl(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, double pose_io[6], double *log_scale_io, double *stats,
                       const unsigned short *bits, const int *decide, unsigned short *bits_out, int *decide_out) {
    int n = H * W, np = op->nparam;
    real *ae = (real *)malloc(sizeof(real) * n);
    photo_err_map(H, W, tgt, src, op->w_l1, op->w_ssim, ae);
    double Tcur[12], Ttry[12], scur = (np == 7 && log_scale_io) ? *log_scale_io : 0.0, stry;
    orc_pose_to_T(pose_io, Tcur);
    memcpy(Ttry, Tcur, sizeof(Tcur));
    stry = scur;
    lin_t cur, tr;
    double lambda = op->lambda0;
    int have_cur = 0;
    const double s0 = scur, ps = (np == 7) ? op->prior_scale : 0.0;
    for (int it = 0; it < op->n_iters; it++) {
        g_force_bits = bits ? bits + (size_t)it * n : NULL;
        g_record_bits = bits_out ? bits_out + (size_t)it * n : NULL;
        g_lin_idx = bits ? it : -1;
        orc_linearize(H, W, tgt, src, depth_t, depth_s, Ttry, K, stry, op, ae, &tr, NULL, NULL, NULL, NULL, NULL);
        g_force_bits = NULL; g_record_bits = NULL; g_lin_idx = -1;
        if (np == 7) { /* scale prior: the photometric cost alone cannot separate depth scale from |t| */
            tr.cost += ps * (stry - s0) * (stry - s0);
            tr.g[6] += 2 * ps * (stry - s0);
            tr.H[6 * np + 6] += 2 * ps;
        }
        if (stats) { stats[4 * it] = tr.cost; stats[4 * it + 1] = tr.cost_photo; stats[4 * it + 2] = tr.n_mask; stats[4 * it + 3] = lambda; }
        const int acc = op->solver == 0 || !have_cur || (decide ? decide[it] != 0 : tr.cost < cur.cost);
        if (decide_out) decide_out[it] = acc;
        if (acc) {
            if (op->solver == 1 && have_cur) lambda = fmax(lambda * op->lambda_down, op->lambda_min);
            cur = tr; memcpy(Tcur, Ttry, sizeof(Tcur)); scur = stry; have_cur = 1;
        } else {
            lambda *= op->lambda_up;
        }
        apply_step(op, cur.H, cur.g, lambda, Tcur, scur, Ttry, &stry);
    }
    if (op->solver == 1 && op->n_iters > 0) {
        g_record_bits = bits_out ? bits_out + (size_t)op->n_iters * n : NULL;
        double c = (bits ? cost_forced(H, W, tgt, src, depth_t, depth_s, Ttry, K, stry, op, bits + (size_t)op->n_iters * n)
                         : orc_cost(H, W, tgt, src, depth_t, depth_s, Ttry, K, stry, op)) + ps * (stry - s0) * (stry - s0);
        g_record_bits = NULL;
        if (stats) { int it = op->n_iters; stats[4 * it] = c; stats[4 * it + 1] = c; stats[4 * it + 2] = 0; stats[4 * it + 3] = lambda; }
        const int keep = decide ? decide[op->n_iters] != 0 : c < cur.cost;
        if (decide_out) decide_out[op->n_iters] = keep;
        if (keep) { memcpy(Tcur, Ttry, sizeof(Tcur)); scur = stry; }
    } else {
        memcpy(Tcur, Ttry, sizeof(Tcur)); scur = stry;
    }
    orc_T_to_pose(Tcur, pose_io);
    if (np == 7 && log_scale_io) *log_scale_io = scur;
    free(ae);
}

/* orc_refine with the engine's decisions replayed: bits [n_lin][H*W] (bit 0 mask, bit 1 warp validity) and decide [n_lin]
 * (LM: 1 = trial accepted / last step kept), n_lin = n_iters (+1 for LM's final cost check); either may be NULL. */
void orc_refine_forced(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, double pose_io[6], double *log_scale_io, double *stats,
                       const unsigned short *bits, const int *decide) {
    refine_impl(H, W, tgt, src, depth_t, depth_s, K, op, pose_io, log_scale_io, stats, bits, decide, NULL, NULL);
}

/* free-running refinement that RECORDS its own decisions in the trace format (CPU self-test of the replay mechanism) */
void orc_refine_record(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, double pose_io[6], double *log_scale_io, double *stats,
                       unsigned short *bits_out, int *decide_out) {
    refine_impl(H, W, tgt, src, depth_t, depth_s, K, op, pose_io, log_scale_io, stats, NULL, NULL, bits_out, decide_out);
}

void orc_refine(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                const real *K, const orc_opts *op, double pose_io[6], double *log_scale_io, double *stats) {
    refine_impl(H, W, tgt, src, depth_t, depth_s, K, op, pose_io, log_scale_io, stats, NULL, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------- */
/* window mode: B target frames x S sources, forward + inverse directed pairs, per-pixel min over the sources        */
/*
 * Pair order = the stacked order of solve_pose_iteratively (train_mono.py:54-62): n = s*B + b are the forward pairs
 * (target b reconstructed from source s), SB + s*B + b the inverse pairs.  With argmin (optimizer.py:47-69,
 * options['diff_img_argmin']) the forward pairs of one target share a per-pixel selection evaluated at the CURRENT poses:
 *     s*(p)   = first argmin_s diff_s(p)                                          (torch.min over the source axis)
 *     keep(p) = max_s valid_s(p) > 0  and  (automask ? min_s diff_s(p) < min_s auto_err_s(p) : 1)
 *     M_s(p)  = keep(p) [s == s*(p)]
 * and pair s is linearised with M_s in place of its own valid x auto-mask.
 *
 * Two rules for how the pairs' costs are put together (tcsfm_opts.window_rule):
 *   rule 0 (PAIR)       every directed pair is its own least-squares problem: own normaliser sum M_s, own weight map W_s.
 *   rule 1 (REFERENCE)  the scalar that is minimised is compute_optimization_loss itself (optimizer.py:47-86, default options):
 *         L = sum_b sum_p keep_b dmin_b W_{0,b} / sum_b sum_p keep_b                              forward term  (:69)
 *           + 0.25 sum_{inv m} sum_p Minv_m W_m diff_m / sum_{inv m} sum_p Minv_m               inverse term  (:75-79)
 *           + w_dc ( mean_{fwd m, p} dd_m + mean_{inv m, p} dd_m )                              depth consistency (:83-86)
 *       i.e. batch-summed normalisers, and the weight map of SOURCE 0 on every forward pixel whichever source won it.  Without
 *       argmin the forward term is 0.25 sum valid W diff / sum valid over all forward pairs, no auto-mask (:71-73).
 *       Every pair takes a Gauss-Newton step on ITS pose with the gradient of L (exact, incl. the cross term: the pose of
 *       source 0 moves the weight of the pixels the other sources won) and its own block of the curvature model.
 * Everything else (GN/LM, damping, retraction) is per pair and identical to orc_refine.
 */
static void window_select_maps(int H, int W, int S, const real *diff /* [S][n] */, const real *valid, const real *ae, int automask,
                               real *mask /* [S] maps, stride */, size_t mstride, real *margin /* [n] or NULL */) {
    const int n = H * W;
    for (int i = 0; i < n; i++) {
        int smin = 0;
        real dmin = diff[i], amin = ae[i], vany = valid[i], gap = (real)1e30;
        for (int s = 1; s < S; s++) {
            const real d = diff[(size_t)s * n + i];
            if (fabs(d - dmin) < gap) gap = fabs(d - dmin);
            if (d < dmin) { dmin = d; smin = s; }
            if (ae[(size_t)s * n + i] < amin) amin = ae[(size_t)s * n + i];
            if (valid[(size_t)s * n + i] > vany) vany = valid[(size_t)s * n + i];
        }
        if (S > 2) { /* gap to the runner-up, whichever order they came in */
            gap = (real)1e30;
            for (int s = 0; s < S; s++) if (s != smin && fabs(diff[(size_t)s * n + i] - dmin) < gap) gap = fabs(diff[(size_t)s * n + i] - dmin);
        }
        int keep = vany > 0 && (!automask || dmin < amin);
        if (margin) margin[i] = automask && fabs(dmin - amin) < gap ? fabs(dmin - amin) : gap;
        for (int s = 0; s < S; s++) mask[(size_t)s * mstride + i] = (keep && s == smin) ? 1 : 0;
    }
}

void orc_window_select(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, const double *T /* [S*B][12] */, const double *log_scale /* [S*B] or NULL */,
                       real *mask /* [S*B][H*W] */) {
    int n = H * W;
    real *diff = (real *)malloc(sizeof(real) * n * S), *valid = (real *)malloc(sizeof(real) * n * S), *ae = (real *)malloc(sizeof(real) * n * S);
    for (int b = 0; b < B; b++) {
        for (int s = 0; s < S; s++) {
            int m = s * B + b;
            orc_photometric(H, W, tgt + (size_t)b * 3 * n, srcs + (size_t)m * 3 * n, depth_t + (size_t)b * n, depth_s + (size_t)m * n,
                            T + 12 * m, K + 9 * b, log_scale ? log_scale[m] : 0.0, op->w_l1, op->w_ssim, diff + (size_t)s * n,
                            valid + (size_t)s * n, NULL, ae + (size_t)s * n, NULL, NULL);
        }
        window_select_maps(H, W, S, diff, valid, ae, op->automask, mask + (size_t)b * n, (size_t)B * n, NULL);
    }
    free(diff); free(valid); free(ae);
}

static double cost_masked_x(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                            const double T[12], const real *K, double log_scale, const orc_opts *op, const real *mask, const lin_ext *x) {
    int n = H * W;
    real *d = (real *)malloc(sizeof(real) * n), *va = (real *)malloc(sizeof(real) * n), *w = (real *)malloc(sizeof(real) * n);
    orc_photometric(H, W, tgt, src, depth_t, depth_s, T, K, log_scale, op->w_l1, op->w_ssim, d, va, w, NULL, NULL, NULL);
    double num = 0, den = 0, dc = 0;
    for (int i = 0; i < n; i++) { num += mask[i] * (x && x->w_map ? x->w_map[i] : w[i]) * d[i]; den += mask[i]; dc += 1 - w[i]; }
    free(d); free(va); free(w);
    if (x && x->norm > 0) den = x->norm;
    const double sc = (x && x->scale > 0) ? x->scale : 1.0, b = (x && x->b_dc >= 0) ? x->b_dc : op->w_dc / n;
    return (den > 0 ? sc * num / den : 0.0) + b * dc;
}
static double cost_masked(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                          const double T[12], const real *K, double log_scale, const orc_opts *op, const real *mask) {
    return cost_masked_x(H, W, tgt, src, depth_t, depth_s, T, K, log_scale, op, mask, NULL);
}

/* One evaluation of a whole window at the pairs' current transforms: the shared state the pairs' linearisations need
 * (selection masks, and under the REFERENCE rule the batch normalisers, source 0's weight maps and the cross terms). */
typedef struct {
    int H, W, B, S, argmin, rule;
    const orc_opts *op;
    const real **pt;        /* [2SB][4] tgt, src, depth_t, depth_s of every directed pair */
    const real *K;          /* [B][9] */
    const real *ae;         /* [2SB][n] */
    real *mask;             /* [2SB][n] the masks the pairs are linearised with (forward pairs: selection) -- or NULL entries unused */
    int use_mask;           /* forward pairs are given `mask` (selection / forced) */
    real *margin;           /* [B][n] gaps of the selection's comparisons (flip statistics) */
    real *w0, *cross;       /* [B][n] REFERENCE rule: weight map of source 0, sum_{s != 0} M_s diff_s */
    double Kfwd, Kinv;      /* REFERENCE rule: batch-summed mask counts of the forward / inverse pairs */
} win_ctx;

static const real *win_K(const win_ctx *c, int m) { const int SB = c->S * c->B; return c->K + 9 * ((m >= SB ? m - SB : m) % c->B); }

/* masks and couplings at transforms T [2SB][12] (log scales ls [2SB]); bits: the engine's decisions of this linearisation
 * ([2SB][n]) or NULL */
static void window_prepare(win_ctx *c, const double *T, const double *ls, const unsigned short *bits) {
    const int n = c->H * c->W, B = c->B, S = c->S, SB = S * B, N = 2 * SB;
    const orc_opts *op = c->op;
    const int sel = c->argmin && S > 1;
    c->use_mask = sel || (bits != NULL);
    c->Kfwd = c->Kinv = 0;
    if (!sel && !c->rule && !bits) return;
    real *diff = (real *)malloc(sizeof(real) * (size_t)n * S), *valid = (real *)malloc(sizeof(real) * (size_t)n * S), *wt = (real *)malloc(sizeof(real) * (size_t)n * S);
    for (int b = 0; b < B; b++) {
        for (int s = 0; s < S; s++) {
            const int m = s * B + b;
            orc_photometric(c->H, c->W, c->pt[4 * m], c->pt[4 * m + 1], c->pt[4 * m + 2], c->pt[4 * m + 3], T + 12 * m, win_K(c, m), ls[m],
                            op->w_l1, op->w_ssim, diff + (size_t)s * n, valid + (size_t)s * n, wt + (size_t)s * n, NULL, NULL, NULL);
        }
        if (sel) {   /* the selection as this restatement takes it (also during a forced replay: flip statistics) */
            real *ae_b = (real *)malloc(sizeof(real) * (size_t)n * S);
            for (int s = 0; s < S; s++) memcpy(ae_b + (size_t)s * n, c->ae + (size_t)(s * B + b) * n, sizeof(real) * n);
            window_select_maps(c->H, c->W, S, diff, valid, ae_b, op->automask, c->mask + (size_t)b * n, (size_t)B * n, c->margin + (size_t)b * n);
            free(ae_b);
        } else if (c->rule || bits) {   /* own masks of the forward pairs: validity (x auto-mask unless REFERENCE without argmin) */
            for (int s = 0; s < S; s++)
                for (int i = 0; i < n; i++) {
                    const int am = op->automask && !(c->rule && !c->argmin);
                    c->mask[(size_t)(s * B + b) * n + i] = (valid[(size_t)s * n + i] > 0 && (!am || diff[(size_t)s * n + i] < c->ae[(size_t)(s * B + b) * n + i])) ? 1 : 0;
                    if (c->margin) c->margin[(size_t)b * n + i] = (real)1e30;
                }
        }
        if (c->rule) {
            memcpy(c->w0 + (size_t)b * n, wt, sizeof(real) * n);
            for (int i = 0; i < n; i++) {
                double cr = 0;
                for (int s = 0; s < S; s++) {
                    const int mk = bits ? (bits[(size_t)(s * B + b) * n + i] & 1) : (c->mask[(size_t)(s * B + b) * n + i] != 0);
                    c->Kfwd += mk;
                    if (s > 0 && mk) cr += diff[(size_t)s * n + i];
                }
                c->cross[(size_t)b * n + i] = (real)cr;
            }
        }
    }
    if (c->rule) {   /* inverse pairs: own masks, batch-summed */
        for (int m = SB; m < N; m++) {
            if (bits) { for (int i = 0; i < n; i++) c->Kinv += bits[(size_t)m * n + i] & 1; continue; }
            orc_photometric(c->H, c->W, c->pt[4 * m], c->pt[4 * m + 1], c->pt[4 * m + 2], c->pt[4 * m + 3], T + 12 * m, win_K(c, m), ls[m],
                            op->w_l1, op->w_ssim, diff, valid, NULL, NULL, NULL, NULL);
            for (int i = 0; i < n; i++) c->Kinv += (valid[i] > 0 && (!op->automask || diff[i] < c->ae[(size_t)m * n + i])) ? 1 : 0;
        }
    }
    free(diff); free(valid); free(wt);
}

/* the coupling of pair m to its window (REFERENCE rule), or NULL */
static const lin_ext *window_ext(const win_ctx *c, int m, lin_ext *x) {
    if (!c->rule) return NULL;
    const int n = c->H * c->W, B = c->B, SB = c->S * B, fwd = m < SB, b = (fwd ? m : m - SB) % B, s = (fwd ? m : m - SB) / B;
    memset(x, 0, sizeof(*x));
    x->b_dc = c->op->w_dc / ((double)SB * n);
    if (fwd) {
        x->norm = c->Kfwd;
        if (c->argmin) {   /* optimizer.py:69 (with S == 1 the minimum is the pair itself and source 0's weight its own) */
            x->scale = 1.0;
            if (s > 0) x->w_map = c->w0 + (size_t)b * n; else if (c->S > 1) x->cross = c->cross + (size_t)b * n;
        } else { x->scale = 0.25; x->no_automask = 1; }
    } else { x->norm = c->Kinv; x->scale = 0.25; }
    if (!(x->norm > 0)) x->norm = 1;   /* nothing selected anywhere: the sums are zero too */
    return x;
}

/* l_pose_consist (optimizer.py:95-96): 0.1 mean |p_fwd + p_inv| over the S B x 6 entries of the reference's 6-vectors.  Pair m sees
 * c sum_j |r_j| with r = p_m + p_partner, c = w / (6 S B) (half of it is booked as ITS cost, so that the pairs' costs add up to the loss).
 * Gradient in pose coordinates c r_j / max(|r_j|, eps) (IRLS; = c sign(r_j) away from zero, as autograd); curvature 2 c / max(|r_j|, eps)
 * on the diagonal -- the factor 2 is the block-Jacobi majoriser (r moves with BOTH poses: (da + db)^2 <= 2 da^2 + 2 db^2; every pair
 * steps with the partner held at the linearisation point).  Carried to the left perturbation by dp = A^-1 dxi, A = d xi / d pose
 * (orc_euler_left_jacobian): A = [[-I, -Tx Je], [0, -Je]]  =>  A^-1 = [[-I, Tx], [0, -Je^-1]]. */
static void pose_consist_term(const double Tm[12], const double Tp[12], double c, double eps, double *cost, double g[6], double H[36]) {
    double pm[6], pp[6];
    orc_T_to_pose(Tm, pm); orc_T_to_pose(Tp, pp);
    const double th[2] = {-pm[3], -pm[4]};
    const double cx = cos(th[0]), sx = sin(th[0]), cy = cos(th[1]), sy = sin(th[1]);
    const double Je[9] = {1, 0, sy, 0, cx, -sx * cy, 0, sx, cx * cy};
    /* inverse of Je (det = cy) by cofactors */
    const double id = 1.0 / cy;
    const double Ji[9] = {(Je[4] * Je[8] - Je[5] * Je[7]) * id, -(Je[1] * Je[8] - Je[2] * Je[7]) * id, (Je[1] * Je[5] - Je[2] * Je[4]) * id,
                          -(Je[3] * Je[8] - Je[5] * Je[6]) * id, (Je[0] * Je[8] - Je[2] * Je[6]) * id, -(Je[0] * Je[5] - Je[2] * Je[3]) * id,
                          (Je[3] * Je[7] - Je[4] * Je[6]) * id, -(Je[0] * Je[7] - Je[1] * Je[6]) * id, (Je[0] * Je[4] - Je[1] * Je[3]) * id};
    const double tp[3] = {-pm[0], -pm[1], -pm[2]};
    const double Tx[9] = {0, -tp[2], tp[1], tp[2], 0, -tp[0], -tp[1], tp[0], 0};
    double Ai[36];
    memset(Ai, 0, sizeof(Ai));
    for (int i = 0; i < 3; i++) {
        Ai[6 * i + i] = -1;
        for (int j = 0; j < 3; j++) { Ai[6 * i + 3 + j] = Tx[3 * i + j]; Ai[6 * (3 + i) + 3 + j] = -Ji[3 * i + j]; }
    }
    double gp[6], D[6];
    *cost = 0;
    for (int j = 0; j < 6; j++) {
        const double r = pm[j] + pp[j], a = fabs(r), den = a > eps ? a : eps;
        *cost += 0.5 * c * a;
        gp[j] = c * r / den; D[j] = 2.0 * c / den;
    }
    for (int i = 0; i < 6; i++) {
        g[i] = 0;
        for (int k = 0; k < 6; k++) g[i] += Ai[6 * k + i] * gp[k];
        for (int j = 0; j < 6; j++) {
            double v = 0;
            for (int k = 0; k < 6; k++) v += Ai[6 * k + i] * D[k] * Ai[6 * k + j];
            H[6 * i + j] = v;
        }
    }
}

/* bits [n_lin][2SB][H*W], decide [n_lin][2SB] (layout of the engine's trace): see orc_refine_forced.  With bits bit 0 IS the
 * mask every pair is linearised with; the selection is still evaluated here, for the flip statistics only. */
void orc_refine_window_rule(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, int argmin, int rule, double *pose_io /* [2SB][6] */,
                       double *log_scale_io /* [2SB] or NULL */, double *stats /* [2SB][n_iters+1][4] or NULL */,
                       const unsigned short *bits, const int *decide, lin_t *lin_out /* [2SB] or NULL: export the FIRST linearisation and return */) {
    const int n = H * W, np = op->nparam, SB = S * B, N = 2 * SB;
    typedef struct { double Tcur[12], Ttry[12], scur, stry, s0, lambda; lin_t cur; int have_cur; } pstate;
    pstate *ps = (pstate *)calloc(N, sizeof(pstate));
    real *ae = (real *)malloc(sizeof(real) * (size_t)n * N), *mask = (real *)malloc(sizeof(real) * (size_t)n * SB);
    double *Tf = (double *)malloc(sizeof(double) * 12 * N), *lsf = (double *)malloc(sizeof(double) * N);
    const real **pt = (const real **)malloc(sizeof(real *) * N * 4);
    win_ctx c;
    memset(&c, 0, sizeof(c));
    c.H = H; c.W = W; c.B = B; c.S = S; c.argmin = argmin; c.rule = rule; c.op = op; c.pt = pt; c.K = K; c.ae = ae; c.mask = mask;
    c.margin = (real *)malloc(sizeof(real) * (size_t)n * B);
    c.w0 = (real *)malloc(sizeof(real) * (size_t)n * B); c.cross = (real *)malloc(sizeof(real) * (size_t)n * B);
    for (int m = 0; m < N; m++) {
        int inv = m >= SB, q = inv ? m - SB : m, b = q % B;
        const real *ti = tgt + (size_t)b * 3 * n, *si = srcs + (size_t)q * 3 * n, *td = depth_t + (size_t)b * n, *sd = depth_s + (size_t)q * n;
        pt[4 * m] = inv ? si : ti; pt[4 * m + 1] = inv ? ti : si; pt[4 * m + 2] = inv ? sd : td; pt[4 * m + 3] = inv ? td : sd;
        photo_err_map(H, W, pt[4 * m], pt[4 * m + 1], op->w_l1, op->w_ssim, ae + (size_t)m * n);
        orc_pose_to_T(pose_io + 6 * m, ps[m].Tcur);
        memcpy(ps[m].Ttry, ps[m].Tcur, sizeof(ps[m].Tcur));
        ps[m].scur = ps[m].stry = ps[m].s0 = (np == 7 && log_scale_io) ? log_scale_io[m] : 0.0;
        ps[m].lambda = op->lambda0;
    }
    const double pw = (np == 7) ? op->prior_scale : 0.0;
    for (int it = 0; it <= op->n_iters; it++) {
        const int final = it == op->n_iters;
        if (final && !(op->solver == 1 && op->n_iters > 0)) break;
        for (int m = 0; m < N; m++) { memcpy(Tf + 12 * m, ps[m].Ttry, sizeof(double) * 12); lsf[m] = ps[m].stry; }
        window_prepare(&c, Tf, lsf, bits ? bits + (size_t)it * N * n : NULL);
        for (int m = 0; m < N; m++) {
            pstate *p = &ps[m];
            const real *mk = (c.use_mask && m < SB && (argmin && S > 1)) ? mask + (size_t)m * n : NULL;
            const real *Km = win_K(&c, m);
            lin_ext xs;
            const lin_ext *x = window_ext(&c, m, &xs);
            double *st = stats ? stats + ((size_t)m * (op->n_iters + 1) + it) * 4 : NULL;
            const double prior = pw * (p->stry - p->s0) * (p->stry - p->s0);
            const unsigned short *fb = bits ? bits + ((size_t)it * N + m) * n : NULL;
            const int *dec = decide ? decide + (size_t)it * N + m : NULL;
            if (final) { /* LM: cost-only check of the last trial step */
                double cst;
                if (fb) {
                    real *fm = (real *)malloc(sizeof(real) * n);
                    for (int i = 0; i < n; i++) fm[i] = (real)(fb[i] & 1);
                    g_force_bits = fb;
                    cst = cost_masked_x(H, W, pt[4 * m], pt[4 * m + 1], pt[4 * m + 2], pt[4 * m + 3], p->Ttry, Km, p->stry, op, fm, x);
                    g_force_bits = NULL;
                    free(fm);
                } else if (mk || x) {
                    real *om = NULL;
                    if (!mk) {   /* REFERENCE rule, own mask: evaluate it */
                        om = (real *)malloc(sizeof(real) * n);
                        real *d = (real *)malloc(sizeof(real) * n), *va = (real *)malloc(sizeof(real) * n);
                        orc_photometric(H, W, pt[4 * m], pt[4 * m + 1], pt[4 * m + 2], pt[4 * m + 3], p->Ttry, Km, p->stry, op->w_l1, op->w_ssim, d, va, NULL, NULL, NULL, NULL);
                        const int am = op->automask && !(x && x->no_automask);
                        for (int i = 0; i < n; i++) om[i] = (va[i] > 0 && (!am || d[i] < ae[(size_t)m * n + i])) ? 1 : 0;
                        free(d); free(va);
                    }
                    cst = cost_masked_x(H, W, pt[4 * m], pt[4 * m + 1], pt[4 * m + 2], pt[4 * m + 3], p->Ttry, Km, p->stry, op, mk ? mk : om, x);
                    free(om);
                } else cst = orc_cost(H, W, pt[4 * m], pt[4 * m + 1], pt[4 * m + 2], pt[4 * m + 3], p->Ttry, Km, p->stry, op);
                cst += prior;
                if (st) { st[0] = cst; st[1] = cst; st[2] = 0; st[3] = p->lambda; }
                if (dec ? *dec != 0 : cst < p->cur.cost) { memcpy(p->Tcur, p->Ttry, sizeof(p->Tcur)); p->scur = p->stry; }
                continue;
            }
            lin_t tr;
            g_force_bits = fb;
            g_lin_idx = fb ? it : -1;
            g_sel_margin = (mk && fb) ? c.margin + (size_t)(m % B) * n : NULL;
            linearize_masked(H, W, pt[4 * m], pt[4 * m + 1], pt[4 * m + 2], pt[4 * m + 3], p->Ttry, Km, p->stry, op, ae + (size_t)m * n, mk, x,
                             &tr, NULL, NULL, NULL, NULL, NULL);
            g_force_bits = NULL; g_lin_idx = -1; g_sel_margin = NULL;
            if (np == 7) { tr.cost += prior; tr.g[6] += 2 * pw * (p->stry - p->s0); tr.H[6 * np + 6] += 2 * pw; }
            if (rule && op->w_pose_consist > 0) {     /* the partner at the start-of-iteration transforms (Tf): every pair steps on its own */
                double pc, pg[6], pH[36];
                pose_consist_term(Tf + 12 * m, Tf + 12 * (m < SB ? m + SB : m - SB), op->w_pose_consist / (6.0 * SB), op->irls_eps, &pc, pg, pH);
                tr.cost += pc;
                for (int i = 0; i < 6; i++) {
                    tr.g[i] += pg[i];
                    for (int j = 0; j < 6; j++) tr.H[i * np + j] += pH[6 * i + j];
                }
            }
            if (lin_out) lin_out[m] = tr;
            if (st) { st[0] = tr.cost; st[1] = tr.cost_photo; st[2] = tr.n_mask; st[3] = p->lambda; }
            if (op->solver == 0 || !p->have_cur || (dec ? *dec != 0 : tr.cost < p->cur.cost)) {
                if (op->solver == 1 && p->have_cur) p->lambda = fmax(p->lambda * op->lambda_down, op->lambda_min);
                p->cur = tr; memcpy(p->Tcur, p->Ttry, sizeof(p->Tcur)); p->scur = p->stry; p->have_cur = 1;
            } else {
                p->lambda *= op->lambda_up;
            }
            apply_step(op, p->cur.H, p->cur.g, p->lambda, p->Tcur, p->scur, p->Ttry, &p->stry);
        }
        if (lin_out) break;
    }
    if (!lin_out)
        for (int m = 0; m < N; m++) {
            if (!(op->solver == 1 && op->n_iters > 0)) { memcpy(ps[m].Tcur, ps[m].Ttry, sizeof(ps[m].Tcur)); ps[m].scur = ps[m].stry; }
            orc_T_to_pose(ps[m].Tcur, pose_io + 6 * m);
            if (np == 7 && log_scale_io) log_scale_io[m] = ps[m].scur;
        }
    free(ps); free(ae); free(mask); free(Tf); free(lsf); free(pt); free(c.margin); free(c.w0); free(c.cross);
}

void orc_refine_window_forced(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, int argmin, double *pose_io, double *log_scale_io, double *stats,
                       const unsigned short *bits, const int *decide) {
    orc_refine_window_rule(H, W, B, S, tgt, srcs, depth_t, depth_s, K, op, argmin, 0, pose_io, log_scale_io, stats, bits, decide, NULL);
}

void orc_refine_window(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                       const real *K, const orc_opts *op, int argmin, double *pose_io, double *log_scale_io, double *stats) {
    orc_refine_window_rule(H, W, B, S, tgt, srcs, depth_t, depth_s, K, op, argmin, 0, pose_io, log_scale_io, stats, NULL, NULL, NULL);
}

/* one linearisation of a whole window at the given poses: normal equations, cost and mask count of every directed pair */
void orc_linearize_window(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                          const real *K, const orc_opts *op, int argmin, int rule, const double *pose /* [2SB][6] */,
                          const double *log_scale /* [2SB] or NULL */, lin_t *out /* [2SB] */) {
    const int N = 2 * S * B;
    double *p = (double *)malloc(sizeof(double) * 6 * N), *ls = log_scale ? (double *)malloc(sizeof(double) * N) : NULL;
    memcpy(p, pose, sizeof(double) * 6 * N);
    if (ls) memcpy(ls, log_scale, sizeof(double) * N);
    orc_opts o1 = *op;
    if (o1.n_iters < 1) o1.n_iters = 1;
    orc_refine_window_rule(H, W, B, S, tgt, srcs, depth_t, depth_s, K, &o1, argmin, rule, p, ls, NULL, NULL, NULL, out);
    free(p); free(ls);
}

/* ------------------------------------------------------------------------- */
/* dense mode: pose (6) + per-pixel inverse depth of the target, per-pixel Schur complement                           */
/*
 * Unknowns: xi (left SE(3) perturbation) and rho_q = 1 / depth_t(q) for every target pixel q (source depth fixed).
 * Cost: the photometric masked mean C = sum M W diff / sum M (w_dc must be 0 in this mode).
 * Gradient (exact, equal to reference autograd; pinned by golden grad_depth_t):
 *     g_rho[q] = sum_{p : q in window(p)} a_p W_p d diff_p/d rec_q . d rec_q/d rho_q   +  a_q diff_q dW_q/d rho_q
 * Curvature (same generalised-GN model as the pose mode, the window geometry frozen at the pixel itself, so every
 * residual sees ONE depth => the depth block is diagonal and can be eliminated per pixel):
 *     with Jg_p (2 x 6) and alpha_p = d(ix,iy)_p / d rho_p (2 x 1):
 *     H_xx += a Jg' Lam Jg ,  B_p = a Jg' Lam alpha ,  D_p = a alpha' Lam alpha
 * Schur complement on the pose:  S = H_xx - sum_p B_p B_p' / Dd_p ,  gs = g_xi - sum_p B_p g_rho_p / Dd_p ,
 *     Dd_p = (1 + lambda_depth) D_p   (pixels with D_p <= tiny are not updated)
 *     (S + lambda diag S) dxi = -gs ;  drho_p = -(g_rho_p + B_p' dxi) / Dd_p ;  rho clamped to [1/max_depth, 1/min_depth]
 * Outputs of one linearisation: S (6x6), gs (6), cost, n_mask and the per-pixel g_rho, D, B (H*W x 6).
 */
static void linearize_dense_masked(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                         const double T[12], const real *K, const orc_opts *op, const real *auto_err_in, const real *mask_in,
                         double lambda_depth, double w_prior, const real *depth0, lin_t *out, double *g_rho, double *Dq, double *Bq) {
    int n = H * W;
    const int np = 7; /* column 6 of the per-pixel Jacobians is re-purposed as the inverse-depth column */
    cam_t c;
    cam_setup(&c, H, W, K, T, 0.0);
    px_t *px = (px_t *)malloc(sizeof(px_t) * n);
    real *ae = (real *)malloc(sizeof(real) * n), *rec = (real *)malloc(sizeof(real) * 3 * n);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            px_t *P = &px[v * W + u];
            px_eval(&c, src, depth_t, depth_s, u, v, np, P);
            /* scale column (dXp = Xp - t = D R ray) -> inverse-depth column: dXp/drho = -D^2 R ray = -D (Xp - t);
             * the projected (source) depth does not scale with the target depth: drop px_eval's "+pd" term */
            real D = depth_t[v * W + u];
            P->a[6] *= -D; P->b[6] *= -D; P->zc[6] *= -D;
            P->dpd[6] = P->dgx * P->a[6] + P->dgy * P->b[6];
        }
    if (auto_err_in) memcpy(ae, auto_err_in, sizeof(real) * n);
    else photo_err_map(H, W, tgt, src, op->w_l1, op->w_ssim, ae);
    for (int i = 0; i < n; i++)
        for (int ch = 0; ch < 3; ch++) rec[ch * n + i] = px[i].rec[ch];
    const real wl = (real)(op->w_l1 / 3), ws = (real)(op->w_ssim / 3), reps = (real)op->irls_eps;
    double *gx_adj = (double *)calloc((size_t)n * 2, sizeof(double)); /* adjoint wrt sample position (ix, iy) of q */
    double *gxi = (double *)calloc(6, sizeof(double)), *Hxx = (double *)calloc(36, sizeof(double));
    double *own = (double *)calloc((size_t)n, sizeof(double));        /* a_q diff_q dW_q/drho_q */
    real *M = (real *)calloc(n, sizeof(real));
    double *Lam = (double *)calloc((size_t)n * 3, sizeof(double));
    double num = 0, nmask = 0;
    /* pass 1: masks (needs diff), so that the weights a_p are known */
    real *diffm = (real *)malloc(sizeof(real) * n), *Wm = (real *)malloc(sizeof(real) * n);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            int i = v * W + u;
            const px_t *P = &px[i];
            real e = 0;
            for (int ch = 0; ch < 3; ch++) {
                ssim_t s;
                ssim_at(tgt + ch * n, rec + ch * n, H, W, u, v, &s);
                e += wl * clamp01(fabs(rec[ch * n + i] - tgt[ch * n + i])) + ws * s.s;
            }
            real sum = P->cd + P->pd, raw = fabs(P->cd - P->pd) / sum;
            diffm[i] = e; Wm[i] = 1 - clamp01(raw);
            real m = (real)P->nat_valid;
            if (op->automask) m *= (e < ae[i]) ? (real)1 : (real)0;
            if (mask_in) m = mask_in[i];   /* window mode: min-over-sources selection */
            if (g_force_bits) {
                const int fm = g_force_bits[i] & 1;
                flip_note(m, fm, mask_in ? (g_sel_margin ? g_sel_margin[i] < ORC_TIE : 1)
                                         : (P->nat_valid != P->valid) || (op->automask && fabs(e - ae[i]) < ORC_TIE));
                m = (real)fm;
            } else if (!mask_in) m = (real)P->valid * (op->automask ? ((e < ae[i]) ? (real)1 : (real)0) : (real)1);
            if (g_record_bits) g_record_bits[i] = (unsigned short)((g_record_bits[i] & ~1) | (m != 0 ? 1 : 0));
            M[i] = m; nmask += m; num += (double)m * Wm[i] * e;
        }
    /* pass 2: exact gradient by scattering every residual's derivative onto the pixels of its window */
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            int i = v * W + u;
            const px_t *P = &px[i];
            double am = M[i];
            if (am == 0) continue;
            real Wt = Wm[i];
            real sum = P->cd + P->pd, dif = P->cd - P->pd, raw = fabs(dif) / sum;
            real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
            double lxx = 0, lxy = 0, lyy = 0;
            for (int ch = 0; ch < 3; ch++) {
                const real *x = tgt + ch * n, *y = rec + ch * n;
                real r = y[i] - x[i], ar = fabs(r);
                real sgn = (ar <= 1) ? forced_sign(r, (real)1e-6, i, 6 + 2 * ch) : (real)0;
                gx_adj[2 * i] += am * Wt * wl * sgn * P->gx[ch];
                gx_adj[2 * i + 1] += am * Wt * wl * sgn * P->gy[ch];
                if (ar <= 1) {
                    real w1 = wl * Wt / (ar > reps ? ar : reps);
                    lxx += w1 * P->gx[ch] * P->gx[ch]; lxy += w1 * P->gx[ch] * P->gy[ch]; lyy += w1 * P->gy[ch] * P->gy[ch];
                }
                ssim_t s;
                ssim_at(x, y, H, W, u, v, &s);
                if (!s.clamped) {
                    real nn = s.n1 * s.n2, dn = s.d1 * s.d2, ratio = nn / dn, pre = -(real)0.5 / dn / 9;
                    real cA = pre * (2 * s.mux * s.n2 - 2 * s.n1 * s.mux - ratio * (2 * s.muy * s.d2 - 2 * s.d1 * s.muy));
                    real cB = pre * (-ratio * 2 * s.d1), cC = pre * (2 * s.n1);
                    real Sx = 0, Sy = 0;
                    for (int dv = -1; dv <= 1; dv++)
                        for (int du = -1; du <= 1; du++) {
                            int q = refl(v + dv, H) * W + refl(u + du, W);
                            real cf = ws * (cA + cB * y[q] + cC * x[q]);
                            gx_adj[2 * q] += am * Wt * cf * px[q].gx[ch];
                            gx_adj[2 * q + 1] += am * Wt * cf * px[q].gy[ch];
                            Sx += px[q].gx[ch]; Sy += px[q].gy[ch];
                        }
                    const real ninth = (real)1 / 9;
                    real mx = Sx * ninth, my = Sy * ninth, ex = P->gx[ch] - mx, ey = P->gy[ch] - my;
                    real w2 = ws * Wt / s.d2 * (real)1.125, w3 = ws * Wt / s.d1;
                    lxx += w2 * ex * ex + w3 * mx * mx; lxy += w2 * ex * ey + w3 * mx * my; lyy += w2 * ey * ey + w3 * my * my;
                }
            }
            Lam[3 * i] = lxx; Lam[3 * i + 1] = lxy; Lam[3 * i + 2] = lyy;
            /* e dW/dtheta = -diff * d dd/dtheta (own pixel) */
            double kdd = sg * 2.0 / ((double)sum * sum);
            for (int j = 0; j < 6; j++) gxi[j] -= am * diffm[i] * kdd * (P->pd * P->zc[j] - P->cd * P->dpd[j]);
            own[i] = -am * diffm[i] * kdd * (P->pd * P->zc[6] - P->cd * P->dpd[6]);
        }
    /* assemble: pose gradient and per-pixel depth gradient from the adjoint; curvature blocks; Schur complement */
    memset(out, 0, sizeof(*out));
    double a = nmask > 0 ? 1.0 / nmask : 0.0;
    double S[36] = {0}, gs[6] = {0};
    for (int i = 0; i < n; i++) {
        const px_t *P = &px[i];
        double ax = gx_adj[2 * i], ay = gx_adj[2 * i + 1];
        for (int j = 0; j < 6; j++) gxi[j] += ax * P->a[j] + ay * P->b[j];
        double gr = a * (ax * P->a[6] + ay * P->b[6] + own[i]);
        double am = a * M[i], lxx = am * Lam[3 * i], lxy = am * Lam[3 * i + 1], lyy = am * Lam[3 * i + 2];
        double la6 = lxx * P->a[6] + lxy * P->b[6], lb6 = lxy * P->a[6] + lyy * P->b[6];
        double D = la6 * P->a[6] + lb6 * P->b[6], B[6];
        if (w_prior > 0 && depth0) { /* prior on the masked pixels: w sum M ((rho - rho0)/rho0)^2 / sum M */
            double rho = 1.0 / (double)depth_t[i], rho0 = 1.0 / (double)depth0[i];
            gr += am * 2.0 * w_prior * (rho - rho0) / (rho0 * rho0);
            D += am * 2.0 * w_prior / (rho0 * rho0);
            out->cost_dc += am * w_prior * (rho - rho0) * (rho - rho0) / (rho0 * rho0);
        }
        for (int j = 0; j < 6; j++) {
            double la = lxx * P->a[j] + lxy * P->b[j], lb = lxy * P->a[j] + lyy * P->b[j];
            B[j] = la * P->a[6] + lb * P->b[6];
            for (int k = 0; k <= j; k++) Hxx[j * 6 + k] += la * P->a[k] + lb * P->b[k];
        }
        if (g_rho) g_rho[i] = gr;
        if (Dq) Dq[i] = D;
        if (Bq) for (int j = 0; j < 6; j++) Bq[i * 6 + j] = B[j];
        /* a pixel whose own sample is valid but blends with grid_sample's zero padding keeps its depth (not eliminated, not
         * updated): its image gradients are (colour / 1 px), its Gauss-Newton depth step is noise (see dense_kernel.h) */
        if (P->valid && !P->dc_in) { D = 0; if (Dq) Dq[i] = 0; }
        double Dd = (1.0 + lambda_depth) * D;
        if (Dd > 1e-30) {
            for (int j = 0; j < 6; j++) {
                gs[j] -= B[j] * gr / Dd;
                for (int k = 0; k <= j; k++) S[j * 6 + k] -= B[j] * B[k] / Dd;
            }
        }
    }
    for (int j = 0; j < 6; j++) {
        out->g[j] = a * gxi[j] + gs[j];
        for (int k = 0; k <= j; k++) { out->H[j * 6 + k] = Hxx[j * 6 + k] + S[j * 6 + k]; out->H[k * 6 + j] = out->H[j * 6 + k]; }
    }
    out->n_mask = nmask;
    out->cost_photo = a * num;
    out->cost = out->cost_photo + out->cost_dc;
    free(px); free(ae); free(rec); free(gx_adj); free(gxi); free(Hxx); free(own); free(M); free(Lam); free(diffm); free(Wm);
}

/* GN refinement of pose + per-pixel inverse depth (dense BA with per-pixel Schur elimination); depth_io in/out */
void orc_linearize_dense(int H, int W, const real *tgt, const real *src, const real *depth_t, const real *depth_s,
                         const double T[12], const real *K, const orc_opts *op, const real *auto_err_in, double lambda_depth,
                         double w_prior, const real *depth0, lin_t *out, double *g_rho, double *Dq, double *Bq) {
    linearize_dense_masked(H, W, tgt, src, depth_t, depth_s, T, K, op, auto_err_in, NULL, lambda_depth, w_prior, depth0, out, g_rho, Dq, Bq);
}

/* Per-pixel trust region of the depth step, |drho| <= 0.25 rho, then the clamp to the depth range.  Where a sample (or a
 * neighbour's, through the 3x3 SSIM window) touches grid_sample's zero padding the photometric gradient is (colour / 1 px)
 * against an ordinary pixel's curvature: raw Gauss-Newton steps of 30-60 % of the inverse depth at a few dozen pixels per frame
 * (measured), where the linearisation means nothing. */
#define DEPTH_STEP_MAX 0.25
static inline double depth_step(double rho0, double drho, double lo, double hi) {
    const double lim = DEPTH_STEP_MAX * rho0;
    double rho = rho0 + (drho < -lim ? -lim : (drho > lim ? lim : drho));
    return rho < lo ? lo : (rho > hi ? hi : rho);
}

/* Per-pair state of the dense refinement (shared by the pair form and the window form).
 * GN (solver 0): every trial is accepted.  LM (solver 1): the pose block is Marquardt-damped; a trial (pose AND depth map)
 * is accepted when it lowers the cost, otherwise the step is recomputed from the ACCEPTED linearisation -- reduced pose system
 * and per-pixel records (g_rho, D, B) -- with a larger lambda, starting again from the accepted depth map; lambda_depth
 * stays fixed.  After the last iteration LM evaluates the last trial once more and keeps it only if it lowered the cost. */
typedef struct {
    int n;
    double Tcur[12], Ttry[12], lambda;
    int have_cur;
    lin_t cur;
    real *dep_try, *dep_acc; /* trial depth map (caller's buffer) / accepted copy */
    double *gr, *Dq, *Bq;    /* records of the trial linearisation */
    double *gr_a, *Dq_a, *Bq_a; /* ... of the accepted one */
} dense_state;

static void dense_state_init(dense_state *st, int n, real *depth_io, const double pose[6], double lambda0) {
    st->n = n; st->lambda = lambda0; st->have_cur = 0;
    orc_pose_to_T(pose, st->Tcur); memcpy(st->Ttry, st->Tcur, sizeof(st->Tcur));
    st->dep_try = depth_io;
    st->dep_acc = (real *)malloc(sizeof(real) * n);
    st->gr = (double *)malloc(sizeof(double) * n); st->Dq = (double *)malloc(sizeof(double) * n); st->Bq = (double *)malloc(sizeof(double) * n * 6);
    st->gr_a = (double *)malloc(sizeof(double) * n); st->Dq_a = (double *)malloc(sizeof(double) * n); st->Bq_a = (double *)malloc(sizeof(double) * n * 6);
}
static void dense_state_free(dense_state *st) {
    free(st->dep_acc); free(st->gr); free(st->Dq); free(st->Bq); free(st->gr_a); free(st->Dq_a); free(st->Bq_a);
}
/* after a linearisation `tr` at (Ttry, dep_try) with records in st->gr/Dq/Bq: accept / reject, next trial */
static void dense_state_step(dense_state *st, const orc_opts *op, const lin_t *tr, double lambda_depth, double min_depth, double max_depth,
                             double *stats_row, const int *dec) {
    const int n = st->n;
    if (stats_row) { stats_row[0] = tr->cost; stats_row[1] = tr->cost_photo; stats_row[2] = tr->n_mask; stats_row[3] = st->lambda; }
    if (op->solver == 0 || !st->have_cur || (dec ? *dec != 0 : tr->cost < st->cur.cost)) {
        if (op->solver == 1 && st->have_cur) st->lambda = fmax(st->lambda * op->lambda_down, op->lambda_min);
        st->cur = *tr; memcpy(st->Tcur, st->Ttry, sizeof(st->Tcur)); st->have_cur = 1;
        memcpy(st->dep_acc, st->dep_try, sizeof(real) * n);
        memcpy(st->gr_a, st->gr, sizeof(double) * n); memcpy(st->Dq_a, st->Dq, sizeof(double) * n); memcpy(st->Bq_a, st->Bq, sizeof(double) * n * 6);
    } else {
        st->lambda *= op->lambda_up;
    }
    double delta[6], E[12];
    orc_solve_step(6, st->cur.H, st->cur.g, st->lambda, delta);
    orc_se3_exp(delta, E);
    orc_se3_mul(E, st->Tcur, st->Ttry);
    const double lo = 1.0 / max_depth, hi = 1.0 / min_depth;
    for (int i = 0; i < n; i++) {
        double Dd = (1.0 + lambda_depth) * st->Dq_a[i];
        if (!(Dd > 1e-30)) { st->dep_try[i] = st->dep_acc[i]; continue; }
        double bd = 0;
        for (int j = 0; j < 6; j++) bd += st->Bq_a[i * 6 + j] * delta[j];
        st->dep_try[i] = (real)(1.0 / depth_step(1.0 / (double)st->dep_acc[i], -(st->gr_a[i] + bd) / Dd, lo, hi));
    }
}
/* the end: GN keeps the last trial; LM keeps it only if its cost (tr_final) is lower */
static void dense_state_finish(dense_state *st, const orc_opts *op, const lin_t *tr_final, double pose_out[6], double *stats_row, const int *dec) {
    int keep = 1;
    if (op->solver == 1 && op->n_iters > 0) {
        if (stats_row) { stats_row[0] = tr_final->cost; stats_row[1] = tr_final->cost_photo; stats_row[2] = tr_final->n_mask; stats_row[3] = st->lambda; }
        keep = dec ? *dec != 0 : tr_final->cost < st->cur.cost;
    }
    if (keep) memcpy(st->Tcur, st->Ttry, sizeof(st->Tcur));
    else memcpy(st->dep_try, st->dep_acc, sizeof(real) * st->n);
    orc_T_to_pose(st->Tcur, pose_out);
}

/*
 * Dense window mode: the 2 S B directed pairs of a window (stacked order as in orc_refine_window), each refining its pose
 * and ITS OWN copy of its target's depth (forward pairs: a copy of target b's depth per source; inverse pairs: source
 * (s,b)'s depth).  With argmin the forward pairs of a target use the min-over-sources selection, evaluated at every
 * linearisation at the pairs' current TRIAL poses and depth copies.
 * depth_io [2SB][H*W]: in = initial target depth of every pair, out = refined.  depth_src [2SB][H*W]: the (fixed) source depths.
 */
static void dense_window_select(int H, int W, int B, int S, const real **img, const real *depth_io, const real *depth_src, const real *K,
                                const orc_opts *op, const dense_state *st, const real *ae, real *diff, real *valid, real *mask) {
    const int n = H * W;
    for (int b = 0; b < B; b++) {
        for (int s = 0; s < S; s++) {
            int m = s * B + b;
            orc_photometric(H, W, img[2 * m], img[2 * m + 1], depth_io + (size_t)m * n, depth_src + (size_t)m * n, st[m].Ttry, K + 9 * b, 0.0,
                            op->w_l1, op->w_ssim, diff + (size_t)s * n, valid + (size_t)s * n, NULL, NULL, NULL, NULL);
        }
        for (int i = 0; i < n; i++) {
            int smin = 0;
            real dmin = diff[i], amin = ae[(size_t)b * n + i], vany = valid[i];
            for (int s = 1; s < S; s++) {
                if (diff[(size_t)s * n + i] < dmin) { dmin = diff[(size_t)s * n + i]; smin = s; }
                real a = ae[(size_t)(s * B + b) * n + i];
                if (a < amin) amin = a;
                if (valid[(size_t)s * n + i] > vany) vany = valid[(size_t)s * n + i];
            }
            int keep = vany > 0 && (!op->automask || dmin < amin);
            for (int s = 0; s < S; s++) mask[(size_t)(s * B + b) * n + i] = (keep && s == smin) ? 1 : 0;
        }
    }
}

void orc_refine_dense_window_forced(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, const real *depth_src,
                             const real *K, const orc_opts *op, int argmin, double lambda_depth, double w_prior, double min_depth,
                             double max_depth, double *pose_io /* [2SB][6] */, double *stats /* [2SB][n_iters+1][4] or NULL */,
                             const unsigned short *bits /* [n_lin][2SB][H*W] */, const int *decide /* [n_lin][2SB] */) {
    const int n = H * W, SB = S * B, N = 2 * SB, sel = argmin && S > 1;   /* (forced replay: evaluated for the flip statistics only) */
    real *ae = (real *)malloc(sizeof(real) * (size_t)n * N), *d0 = (real *)malloc(sizeof(real) * (size_t)n * N);
    real *mask = sel ? (real *)malloc(sizeof(real) * (size_t)n * SB) : NULL;
    real *diff = sel ? (real *)malloc(sizeof(real) * (size_t)n * S) : NULL, *valid = sel ? (real *)malloc(sizeof(real) * (size_t)n * S) : NULL;
    dense_state *st = (dense_state *)calloc(N, sizeof(dense_state));
    const real **img = (const real **)malloc(sizeof(real *) * N * 2);
    memcpy(d0, depth_io, sizeof(real) * (size_t)n * N);
    for (int m = 0; m < N; m++) {
        int inv = m >= SB, q = inv ? m - SB : m, b = q % B;
        const real *ti = tgt + (size_t)b * 3 * n, *si = srcs + (size_t)q * 3 * n;
        img[2 * m] = inv ? si : ti; img[2 * m + 1] = inv ? ti : si;
        photo_err_map(H, W, img[2 * m], img[2 * m + 1], op->w_l1, op->w_ssim, ae + (size_t)m * n);
        dense_state_init(&st[m], n, depth_io + (size_t)m * n, pose_io + 6 * m, op->lambda0);
    }
    const int lm_final = op->solver == 1 && op->n_iters > 0;
    for (int it = 0; it <= op->n_iters; it++) {
        const int final = it == op->n_iters;
        if (final && !lm_final) break;
        if (sel) dense_window_select(H, W, B, S, img, depth_io, depth_src, K, op, st, ae, diff, valid, mask);
        for (int m = 0; m < N; m++) {
            const real *Km = K + 9 * ((m >= SB ? m - SB : m) % B);
            lin_t L;
            g_force_bits = bits ? bits + ((size_t)it * N + m) * n : NULL;
            g_lin_idx = bits ? it : -1;
            linearize_dense_masked(H, W, img[2 * m], img[2 * m + 1], st[m].dep_try, depth_src + (size_t)m * n, st[m].Ttry, Km, op, ae + (size_t)m * n,
                                   (sel && m < SB) ? mask + (size_t)m * n : NULL, lambda_depth, w_prior, d0 + (size_t)m * n, &L, st[m].gr, st[m].Dq, st[m].Bq);
            g_force_bits = NULL; g_lin_idx = -1;
            double *row = stats ? stats + ((size_t)m * (op->n_iters + 1) + it) * 4 : NULL;
            const int *dec = decide ? decide + (size_t)it * N + m : NULL;
            if (final) dense_state_finish(&st[m], op, &L, pose_io + 6 * m, row, dec);
            else dense_state_step(&st[m], op, &L, lambda_depth, min_depth, max_depth, row, dec);
        }
    }
    for (int m = 0; m < N; m++) {
        if (!lm_final) dense_state_finish(&st[m], op, NULL, pose_io + 6 * m, NULL, NULL);
        dense_state_free(&st[m]);
    }
    free(ae); free(d0); free(mask); free(diff); free(valid); free(st); free(img);
}

void orc_refine_dense_window(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, const real *depth_src,
                             const real *K, const orc_opts *op, int argmin, double lambda_depth, double w_prior, double min_depth,
                             double max_depth, double *pose_io, double *stats) {
    orc_refine_dense_window_forced(H, W, B, S, tgt, srcs, depth_io, depth_src, K, op, argmin, lambda_depth, w_prior, min_depth, max_depth,
                                   pose_io, stats, NULL, NULL);
}

void orc_refine_dense_forced(int H, int W, const real *tgt, const real *src, real *depth_io, const real *depth_s, const real *K,
                      const orc_opts *op, double lambda_depth, double w_prior, double min_depth, double max_depth,
                      double pose_io[6], double *stats, const unsigned short *bits /* [n_lin][H*W] */, const int *decide /* [n_lin] */) {
    int n = H * W;
    real *ae = (real *)malloc(sizeof(real) * n), *d0 = (real *)malloc(sizeof(real) * n);
    memcpy(d0, depth_io, sizeof(real) * n);
    photo_err_map(H, W, tgt, src, op->w_l1, op->w_ssim, ae);
    dense_state st;
    dense_state_init(&st, n, depth_io, pose_io, op->lambda0);
    const int lm_final = op->solver == 1 && op->n_iters > 0;
    for (int it = 0; it <= op->n_iters; it++) {
        const int final = it == op->n_iters;
        if (final && !lm_final) break;
        lin_t L;
        g_force_bits = bits ? bits + (size_t)it * n : NULL;
        g_lin_idx = bits ? it : -1;
        orc_linearize_dense(H, W, tgt, src, st.dep_try, depth_s, st.Ttry, K, op, ae, lambda_depth, w_prior, d0, &L, st.gr, st.Dq, st.Bq);
        g_force_bits = NULL; g_lin_idx = -1;
        const int *dec = decide ? decide + it : NULL;
        if (final) dense_state_finish(&st, op, &L, pose_io, stats ? stats + 4 * it : NULL, dec);
        else dense_state_step(&st, op, &L, lambda_depth, min_depth, max_depth, stats ? stats + 4 * it : NULL, dec);
    }
    if (!lm_final) dense_state_finish(&st, op, NULL, pose_io, NULL, NULL);
    dense_state_free(&st);
    free(ae); free(d0);
}

void orc_refine_dense(int H, int W, const real *tgt, const real *src, real *depth_io, const real *depth_s, const real *K,
                      const orc_opts *op, double lambda_depth, double w_prior, double min_depth, double max_depth,
                      double pose_io[6], double *stats) {
    orc_refine_dense_forced(H, W, tgt, src, depth_io, depth_s, K, op, lambda_depth, w_prior, min_depth, max_depth, pose_io, stats, NULL, NULL);
}

/* ------------------------------------------------------------------------- */
/* JOINT dense mode: the S forward pairs of a target share ONE inverse-depth map (optimize_depth_pred, optimizer.py:194-198,     */
/* 235-247: one disparity per frame that every term of the loss sees)                                                           */
/*
 * Unknowns of target b: the left SE(3) perturbations xi_0 .. xi_{S-1} of its S forward warps and rho_q = 1 / depth_t(q).
 * Cost: the forward term of the reference's loss (optimizer.py:47-73) plus the masked depth prior of the dense mode,
 *     argmin:      C = [ sum_p sum_s M_s(p) W_x(p) diff_s(p)  +  w sum_p m(p) ((rho - rho0)/rho0)^2 ] / K,   K = sum_p sum_s M_s(p)
 *                  M_s = the min-over-sources selection (orc_window_select), m = sum_s M_s; W_x: under the REFERENCE window rule
 *                  (rule 1) the weight map of SOURCE 0 on every pixel (:69), under the PAIR rule (rule 0) the winning source's own
 *     no argmin:   the same with M_s = valid_s (no auto-mask, :71-73) and every source's own weight map W_s
 * (the constant factors of the reference's loss -- 0.25, batch normaliser -- multiply C as a whole and cancel in every step).
 * Gradient: exact (equal to reference autograd of the forward term w.r.t. the poses and the SHARED target depth, golden G13),
 *   adjoint form per source as in linearize_dense_masked, plus the weight terms: -M_s diff_s d dd_x/d(xi_x, rho), x = 0 or s.
 * Curvature: the dense mode's model per (pixel, source) with the window geometry frozen at the pixel, so residual (p, s) sees
 *   xi_s and rho_p only:   H_ss += a Jg_s' Lam_s Jg_s,  B_s(p) = a Jg_s' Lam_s alpha_s,  D(p) = sum_s a alpha_s' Lam_s alpha_s (+ prior)
 * Schur complement on the 6S pose unknowns:
 *     S_ss' = [s == s'] H_ss - sum_p B_s(p) B_s'(p)' / Dd(p),   gS_s = g_s - sum_p B_s(p) g_rho(p) / Dd(p),   Dd = (1 + lambda_depth) D
 *   -- a full 6S x 6S system (12 x 12 for the KITTI window) wherever a pixel counts for more than one source (no argmin); under
 *   the min over the sources every pixel counts for exactly one source and the off-diagonal blocks vanish identically.
 *     (S + lambda diag S) dxi = -gS ;  drho_p = -(g_rho_p + sum_s B_s(p)' dxi_s) / Dd_p ;  rho clamped to [1/max_depth, 1/min_depth]
 * LM (solver 1): accept / reject on C, all S poses and the depth map together (as dense_state_step).
 */
#define JMAXS 4
typedef struct {
    double H[6 * JMAXS * 6 * JMAXS], g[6 * JMAXS];   /* reduced system (np = 6 S, row-major np x np) */
    double cost, cost_photo, cost_prior, K;
    double share[JMAXS], n_mask[JMAXS];               /* per source: sum M_s W diff / K, sum M_s */
} jlin_t;

static void linearize_joint(int H, int W, int S, const real *tgt, const real *const *src, const real *depth_t, const real *const *depth_s,
                            const double *T /* [S][12] */, const real *K, const orc_opts *op, const real *const *ae, int argmin, int rule,
                            const unsigned short *const *bits /* [S] or NULL */, double lambda_depth, double w_prior, const real *depth0,
                            jlin_t *out, double *g_rho /* [n] */, double *Dq /* [n] */, double *Bq /* [n][S][6] */) {
    const int n = H * W, np = 7, NP = 6 * S;
    const real wl = (real)(op->w_l1 / 3), ws = (real)(op->w_ssim / 3), reps = (real)op->irls_eps;
    px_t **px = (px_t **)malloc(sizeof(px_t *) * S);
    real **rec = (real **)malloc(sizeof(real *) * S);
    real *diff = (real *)malloc(sizeof(real) * (size_t)n * S), *valid = (real *)malloc(sizeof(real) * (size_t)n * S), *Wm = (real *)malloc(sizeof(real) * (size_t)n * S);
    real *aeb = (real *)malloc(sizeof(real) * (size_t)n * S), *mask = (real *)malloc(sizeof(real) * (size_t)n * S), *margin = (real *)malloc(sizeof(real) * n);
    for (int s = 0; s < S; s++) {
        cam_t c;
        cam_setup(&c, H, W, K, T + 12 * s, 0.0);
        px[s] = (px_t *)malloc(sizeof(px_t) * n);
        rec[s] = (real *)malloc(sizeof(real) * 3 * n);
        g_force_bits = bits ? bits[s] : NULL;
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                px_t *P = &px[s][v * W + u];
                px_eval(&c, src[s], depth_t, depth_s[s], u, v, np, P);
                real D = depth_t[v * W + u];          /* scale column -> inverse-depth column (as linearize_dense_masked) */
                P->a[6] *= -D; P->b[6] *= -D; P->zc[6] *= -D;
                P->dpd[6] = P->dgx * P->a[6] + P->dgy * P->b[6];
            }
        for (int i = 0; i < n; i++)
            for (int ch = 0; ch < 3; ch++) rec[s][ch * n + i] = px[s][i].rec[ch];
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                int i = v * W + u;
                const px_t *P = &px[s][i];
                real e = 0;
                for (int ch = 0; ch < 3; ch++) {
                    ssim_t q;
                    ssim_at(tgt + ch * n, rec[s] + ch * n, H, W, u, v, &q);
                    e += wl * clamp01(fabs(rec[s][ch * n + i] - tgt[ch * n + i])) + ws * q.s;
                }
                real sum = P->cd + P->pd;
                diff[(size_t)s * n + i] = e; Wm[(size_t)s * n + i] = 1 - clamp01(fabs(P->cd - P->pd) / sum);
                valid[(size_t)s * n + i] = (real)P->nat_valid;
            }
        memcpy(aeb + (size_t)s * n, ae[s], sizeof(real) * n);
    }
    g_force_bits = NULL;
    /* masks as this restatement takes them; a forced replay overrides them below (flip statistics per source) */
    if (argmin && S > 1) window_select_maps(H, W, S, diff, valid, aeb, op->automask, mask, (size_t)n, margin);
    else
        for (int s = 0; s < S; s++)
            for (int i = 0; i < n; i++) {
                const int am = op->automask && (argmin || S == 1);   /* no argmin: optimizer.py:71-73 has no auto-mask */
                mask[(size_t)s * n + i] = (valid[(size_t)s * n + i] > 0 && (!am || diff[(size_t)s * n + i] < aeb[(size_t)s * n + i])) ? 1 : 0;
                margin[i] = (real)1e30;
            }
    for (int s = 0; s < S; s++)
        for (int i = 0; i < n; i++) {
            real *m = &mask[(size_t)s * n + i];
            if (bits) {
                const int fm = bits[s][i] & 1;
                const int am = op->automask && (argmin || S == 1);
                flip_note(*m, fm, (argmin && S > 1) ? margin[i] < ORC_TIE
                                                    : (px[s][i].nat_valid != px[s][i].valid) || (am && fabs(diff[(size_t)s * n + i] - aeb[(size_t)s * n + i]) < ORC_TIE));
                *m = (real)fm;
            } else if (!(argmin && S > 1)) {
                const int am = op->automask && (argmin || S == 1);
                *m = (px[s][i].valid && (!am || diff[(size_t)s * n + i] < aeb[(size_t)s * n + i])) ? 1 : 0;
            }
        }
    memset(out, 0, sizeof(*out));
    double Kn = 0;
    for (int s = 0; s < S; s++)
        for (int i = 0; i < n; i++) { Kn += mask[(size_t)s * n + i]; out->n_mask[s] += mask[(size_t)s * n + i]; }
    out->K = Kn;
    const double a = Kn > 0 ? 1.0 / Kn : 0.0;
    double *gx_adj = (double *)calloc((size_t)n * 2 * S, sizeof(double));   /* per source: adjoint wrt the sample position of q */
    double *gxi = (double *)calloc(NP, sizeof(double)), *Hxx = (double *)calloc((size_t)NP * NP, sizeof(double));
    double *own = (double *)calloc(n, sizeof(double));                       /* weight terms of the depth gradient */
    double *Lam = (double *)calloc((size_t)n * 3 * S, sizeof(double));
    for (int s = 0; s < S; s++) {
        const int x = (argmin && rule) ? 0 : s;             /* whose weight map multiplies this source's pixels: source 0's under the
                                                               REFERENCE rule with argmin (optimizer.py:69), else the source's own */
        g_force_bits = bits ? bits[s] : NULL;                /* sign ties of source s follow the engine's codes of pair s */
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const int i = v * W + u;
                const double am = mask[(size_t)s * n + i];
                if (am == 0) continue;
                const px_t *P = &px[s][i];
                const real Wt = Wm[(size_t)x * n + i];
                out->share[s] += am * Wt * diff[(size_t)s * n + i];
                double lxx = 0, lxy = 0, lyy = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const real *xx = tgt + ch * n, *y = rec[s] + ch * n;
                    real r = y[i] - xx[i], ar = fabs(r);
                    real sgn = (ar <= 1) ? forced_sign(r, (real)1e-6, i, 6 + 2 * ch) : (real)0;
                    gx_adj[((size_t)s * n + i) * 2] += am * Wt * wl * sgn * P->gx[ch];
                    gx_adj[((size_t)s * n + i) * 2 + 1] += am * Wt * wl * sgn * P->gy[ch];
                    if (ar <= 1) {
                        real w1 = wl * Wt / (ar > reps ? ar : reps);
                        lxx += w1 * P->gx[ch] * P->gx[ch]; lxy += w1 * P->gx[ch] * P->gy[ch]; lyy += w1 * P->gy[ch] * P->gy[ch];
                    }
                    ssim_t q;
                    ssim_at(xx, y, H, W, u, v, &q);
                    if (!q.clamped) {
                        real nn = q.n1 * q.n2, dn = q.d1 * q.d2, ratio = nn / dn, pre = -(real)0.5 / dn / 9;
                        real cA = pre * (2 * q.mux * q.n2 - 2 * q.n1 * q.mux - ratio * (2 * q.muy * q.d2 - 2 * q.d1 * q.muy));
                        real cB = pre * (-ratio * 2 * q.d1), cC = pre * (2 * q.n1);
                        real Sx = 0, Sy = 0;
                        for (int dv = -1; dv <= 1; dv++)
                            for (int du = -1; du <= 1; du++) {
                                int qi = refl(v + dv, H) * W + refl(u + du, W);
                                real cf = ws * (cA + cB * y[qi] + cC * xx[qi]);
                                gx_adj[((size_t)s * n + qi) * 2] += am * Wt * cf * px[s][qi].gx[ch];
                                gx_adj[((size_t)s * n + qi) * 2 + 1] += am * Wt * cf * px[s][qi].gy[ch];
                                Sx += px[s][qi].gx[ch]; Sy += px[s][qi].gy[ch];
                            }
                        const real ninth = (real)1 / 9;
                        real mx = Sx * ninth, my = Sy * ninth, ex = P->gx[ch] - mx, ey = P->gy[ch] - my;
                        real w2 = ws * Wt / q.d2 * (real)1.125, w3 = ws * Wt / q.d1;
                        lxx += w2 * ex * ex + w3 * mx * mx; lxy += w2 * ex * ey + w3 * mx * my; lyy += w2 * ey * ey + w3 * my * my;
                    }
                }
                Lam[((size_t)s * n + i) * 3] = lxx; Lam[((size_t)s * n + i) * 3 + 1] = lxy; Lam[((size_t)s * n + i) * 3 + 2] = lyy;
            }
        /* weight terms: -M_s diff_s d dd_x / d theta at the pixel itself (sign of cd_x - pd_x: pair x's code) */
        g_force_bits = bits ? bits[x] : NULL;
        for (int i = 0; i < n; i++) {
            const double am = mask[(size_t)s * n + i];
            if (am == 0) continue;
            const px_t *X = &px[x][i];
            real sum = X->cd + X->pd, dif = X->cd - X->pd, raw = fabs(dif) / sum;
            real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
            double kdd = sg * 2.0 / ((double)sum * sum), e = diff[(size_t)s * n + i];
            for (int j = 0; j < 6; j++) gxi[6 * x + j] -= am * e * kdd * (X->pd * X->zc[j] - X->cd * X->dpd[j]);
            own[i] -= am * e * kdd * (X->pd * X->zc[6] - X->cd * X->dpd[6]);
        }
    }
    g_force_bits = NULL;
    /* assemble: pose gradients and the depth gradient from the adjoints; curvature blocks; Schur complement */
    double *Sm = (double *)calloc((size_t)NP * NP, sizeof(double)), *gs = (double *)calloc(NP, sizeof(double));
    for (int i = 0; i < n; i++) {
        double gr = a * own[i], D = 0, B[6 * JMAXS], mcnt = 0;
        for (int s = 0; s < S; s++) {
            const px_t *P = &px[s][i];
            const double ax = gx_adj[((size_t)s * n + i) * 2], ay = gx_adj[((size_t)s * n + i) * 2 + 1];
            for (int j = 0; j < 6; j++) gxi[6 * s + j] += ax * P->a[j] + ay * P->b[j];
            gr += a * (ax * P->a[6] + ay * P->b[6]);
            const double am = a * mask[(size_t)s * n + i];
            mcnt += mask[(size_t)s * n + i];
            const double lxx = am * Lam[((size_t)s * n + i) * 3], lxy = am * Lam[((size_t)s * n + i) * 3 + 1], lyy = am * Lam[((size_t)s * n + i) * 3 + 2];
            const double la6 = lxx * P->a[6] + lxy * P->b[6], lb6 = lxy * P->a[6] + lyy * P->b[6];
            D += la6 * P->a[6] + lb6 * P->b[6];
            for (int j = 0; j < 6; j++) {
                const double la = lxx * P->a[j] + lxy * P->b[j], lb = lxy * P->a[j] + lyy * P->b[j];
                B[6 * s + j] = la * P->a[6] + lb * P->b[6];
                for (int k = 0; k <= j; k++) Hxx[(size_t)(6 * s + j) * NP + 6 * s + k] += la * P->a[k] + lb * P->b[k];
            }
        }
        if (w_prior > 0 && depth0) {
            const double rho = 1.0 / (double)depth_t[i], rho0 = 1.0 / (double)depth0[i], am = a * mcnt;
            gr += am * 2.0 * w_prior * (rho - rho0) / (rho0 * rho0);
            D += am * 2.0 * w_prior / (rho0 * rho0);
            out->cost_prior += am * w_prior * (rho - rho0) * (rho - rho0) / (rho0 * rho0);
        }
        for (int s = 0; s < S; s++)
            if (px[s][i].valid && !px[s][i].dc_in) D = 0;    /* sampled across the zero padding: the pixel keeps its depth */
        if (g_rho) g_rho[i] = gr;
        if (Dq) Dq[i] = D;
        if (Bq) memcpy(Bq + (size_t)i * 6 * S, B, sizeof(double) * 6 * S);
        const double Dd = (1.0 + lambda_depth) * D;
        if (Dd > 1e-30)
            for (int j = 0; j < NP; j++) {
                gs[j] -= B[j] * gr / Dd;
                for (int k = 0; k <= j; k++) Sm[(size_t)j * NP + k] -= B[j] * B[k] / Dd;
            }
    }
    for (int j = 0; j < NP; j++) {
        out->g[j] = a * gxi[j] + gs[j];
        for (int k = 0; k <= j; k++) { const double v = Hxx[(size_t)j * NP + k] + Sm[(size_t)j * NP + k]; out->H[j * NP + k] = v; out->H[k * NP + j] = v; }
    }
    for (int s = 0; s < S; s++) { out->share[s] *= a; out->cost_photo += out->share[s]; }
    out->cost = out->cost_photo + out->cost_prior;
    for (int s = 0; s < S; s++) { free(px[s]); free(rec[s]); }
    free(px); free(rec); free(diff); free(valid); free(Wm); free(aeb); free(mask); free(margin);
    free(gx_adj); free(gxi); free(Hxx); free(own); free(Lam); free(Sm); free(gs);
}

/* one joint linearisation of ONE target at given poses: the reduced system and the per-pixel records (tests: autograd pin) */
void orc_linearize_dense_joint(int H, int W, int S, const real *tgt, const real *srcs /* [S][3][n] */, const real *depth_t,
                               const real *depth_s /* [S][n] */, const real *K, const orc_opts *op, int argmin, int rule, const double *pose /* [S][6] */,
                               double lambda_depth, double w_prior, const real *depth0, double *Hout /* [6S][6S] */, double *gout /* [6S] */,
                               double *scal /* cost, cost_photo, cost_prior, K, share[S], n_mask[S] */, double *g_rho, double *Dq, double *Bq) {
    const int n = H * W;
    const real *src[JMAXS], *ds[JMAXS], *aep[JMAXS];
    real *ae = (real *)malloc(sizeof(real) * (size_t)n * S);
    double T[12 * JMAXS];
    for (int s = 0; s < S; s++) {
        src[s] = srcs + (size_t)s * 3 * n; ds[s] = depth_s + (size_t)s * n;
        photo_err_map(H, W, tgt, src[s], op->w_l1, op->w_ssim, ae + (size_t)s * n);
        aep[s] = ae + (size_t)s * n;
        orc_pose_to_T(pose + 6 * s, T + 12 * s);
    }
    jlin_t L;
    linearize_joint(H, W, S, tgt, src, depth_t, ds, T, K, op, aep, argmin, rule, NULL, lambda_depth, w_prior, depth0, &L, g_rho, Dq, Bq);
    const int NP = 6 * S;
    memcpy(Hout, L.H, sizeof(double) * NP * NP); memcpy(gout, L.g, sizeof(double) * NP);
    scal[0] = L.cost; scal[1] = L.cost_photo; scal[2] = L.cost_prior; scal[3] = L.K;
    for (int s = 0; s < S; s++) { scal[4 + s] = L.share[s]; scal[4 + S + s] = L.n_mask[s]; }
    free(ae);
}

/* The forward group of a window, jointly: B targets x S sources, ONE depth map per target (depth_io [B][n], in / out), the
 * source depths fixed (depth_s [S*B][n], stacked (s, b) as everywhere).  pose_io [S*B][6] in the stacked order of the forward
 * pairs.  stats [S*B][n_iters+1][4] or NULL: row of pair (s, b) = joint cost of target b, its own share, its own mask count,
 * lambda.  bits [n_lin][S*B][n] / decide [n_lin][S*B] (decide of pair (0, b) is the target's decision): forced replay. */
void orc_refine_dense_joint(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, const real *depth_s, const real *K,
                            const orc_opts *op, int argmin, int rule, double lambda_depth, double w_prior, double min_depth, double max_depth,
                            double *pose_io, double *stats, const unsigned short *bits, const int *decide) {
    const int n = H * W, SB = S * B, NP = 6 * S;
    const int lm_final = op->solver == 1 && op->n_iters > 0;
    const double lo = 1.0 / max_depth, hi = 1.0 / min_depth;
    for (int b = 0; b < B; b++) {
        const real *src[JMAXS], *ds[JMAXS], *aep[JMAXS];
        real *ae = (real *)malloc(sizeof(real) * (size_t)n * S), *d0 = (real *)malloc(sizeof(real) * n);
        real *dep_try = depth_io + (size_t)b * n, *dep_acc = (real *)malloc(sizeof(real) * n);
        double Tcur[12 * JMAXS], Ttry[12 * JMAXS], lambda = op->lambda0;
        double *gr = (double *)malloc(sizeof(double) * n), *Dq = (double *)malloc(sizeof(double) * n), *Bq = (double *)malloc(sizeof(double) * (size_t)n * 6 * S);
        double *gr_a = (double *)malloc(sizeof(double) * n), *Dq_a = (double *)malloc(sizeof(double) * n), *Bq_a = (double *)malloc(sizeof(double) * (size_t)n * 6 * S);
        jlin_t cur, tr;
        int have_cur = 0;
        memset(&cur, 0, sizeof(cur));
        memcpy(d0, dep_try, sizeof(real) * n);
        for (int s = 0; s < S; s++) {
            const int m = s * B + b;
            src[s] = srcs + (size_t)m * 3 * n; ds[s] = depth_s + (size_t)m * n;
            photo_err_map(H, W, tgt + (size_t)b * 3 * n, src[s], op->w_l1, op->w_ssim, ae + (size_t)s * n);
            aep[s] = ae + (size_t)s * n;
            orc_pose_to_T(pose_io + 6 * m, Tcur + 12 * s);
        }
        memcpy(Ttry, Tcur, sizeof(double) * 12 * S);
        for (int it = 0; it <= op->n_iters; it++) {
            const int final = it == op->n_iters;
            if (final && !lm_final) break;
            const unsigned short *fb[JMAXS];
            for (int s = 0; s < S; s++) fb[s] = bits ? bits + ((size_t)it * SB + s * B + b) * n : NULL;
            g_lin_idx = bits ? it : -1;
            linearize_joint(H, W, S, tgt + (size_t)b * 3 * n, src, dep_try, ds, Ttry, K + 9 * b, op, aep, argmin, rule, bits ? fb : NULL, lambda_depth, w_prior, d0,
                            &tr, gr, Dq, Bq);
            g_lin_idx = -1;
            if (stats)
                for (int s = 0; s < S; s++) {
                    double *row = stats + ((size_t)(s * B + b) * (op->n_iters + 1) + it) * 4;
                    row[0] = tr.cost; row[1] = tr.share[s]; row[2] = tr.n_mask[s]; row[3] = lambda;
                }
            const int *dec = decide ? decide + (size_t)it * SB + b : NULL;
            if (final) {
                if (dec ? *dec != 0 : tr.cost < cur.cost) memcpy(Tcur, Ttry, sizeof(double) * 12 * S);
                else memcpy(dep_try, dep_acc, sizeof(real) * n);
                break;
            }
            if (op->solver == 0 || !have_cur || (dec ? *dec != 0 : tr.cost < cur.cost)) {
                if (op->solver == 1 && have_cur) lambda = fmax(lambda * op->lambda_down, op->lambda_min);
                cur = tr; memcpy(Tcur, Ttry, sizeof(double) * 12 * S); have_cur = 1;
                memcpy(dep_acc, dep_try, sizeof(real) * n);
                memcpy(gr_a, gr, sizeof(double) * n); memcpy(Dq_a, Dq, sizeof(double) * n); memcpy(Bq_a, Bq, sizeof(double) * (size_t)n * 6 * S);
            } else lambda *= op->lambda_up;
            /* (S + lambda diag S + 1e-12 I) d = -gS */
            double A[36 * JMAXS * JMAXS], dl[6 * JMAXS];
            for (int i = 0; i < NP; i++) {
                for (int j = 0; j < NP; j++) A[i * NP + j] = cur.H[i * NP + j];
                A[i * NP + i] += lambda * cur.H[i * NP + i] + 1e-12;
                dl[i] = -cur.g[i];
            }
            if (chol_solve(NP, A, dl)) memset(dl, 0, sizeof(dl));
            for (int s = 0; s < S; s++) {
                double E[12];
                orc_se3_exp(dl + 6 * s, E);
                orc_se3_mul(E, Tcur + 12 * s, Ttry + 12 * s);
            }
            for (int i = 0; i < n; i++) {
                const double Dd = (1.0 + lambda_depth) * Dq_a[i];
                if (!(Dd > 1e-30)) { dep_try[i] = dep_acc[i]; continue; }
                double bd = 0;
                for (int j = 0; j < NP; j++) bd += Bq_a[(size_t)i * NP + j] * dl[j];
                dep_try[i] = (real)(1.0 / depth_step(1.0 / (double)dep_acc[i], -(gr_a[i] + bd) / Dd, lo, hi));
            }
        }
        if (!lm_final) memcpy(Tcur, Ttry, sizeof(double) * 12 * S);
        for (int s = 0; s < S; s++) orc_T_to_pose(Tcur + 12 * s, pose_io + 6 * (s * B + b));
        free(ae); free(d0); free(dep_acc); free(gr); free(Dq); free(Bq); free(gr_a); free(Dq_a); free(Bq_a);
    }
}

/* ------------------------------------------------------------------------- */
/* DENSE mode on the REFERENCE's own loss (round 4): poses of all 2 S B directed pairs + ONE inverse-depth map per target,        */
/* compute_optimization_loss as optimize_depth_pred minimises it (optimizer.py:47-90, 194-198, 235-247)                           */
/*
 *   L = c_f / K_f  sum_{b,p} sum_s M_s W_x diff_s                                 forward term (:47-73)
 *     + 0.25 / K_i sum_{s,b,p} M_i W_i diff_i                                     inverse term (:75-81)
 *     + w_dc / (S B HW) sum_{s,b,p} (dd_fwd + dd_inv)                             depth consistency (:83-86)
 *     + w_init / (B HW) sum_{b,p} SSIM(sigma_b, sigma0_b)(p)                      l_depth_init (:89-90), sigma = sigmoid disparity
 *   argmin: M_s = min-over-sources selection, W_x = W_0 (weight map of SOURCE 0 on every selected pixel), c_f = 1, K_f summed over
 *   the batch;  no argmin: M_s = valid_s, W_x = W_s, c_f = 0.25.   K_i: sum of valid x auto-mask over all S B inverse pairs.
 * Unknowns: left perturbations xi_n of all 2 S B warps and rho_b(q) = 1 / depth_t(b, q); the SOURCE depth maps are held at their
 * input (the reference lets them drift too, with no prior on them; the engine refines the map the prior and `disp_opt` are about).
 * The target depth enters the forward pairs as the back-projected depth (local in q) and the inverse pairs as the SAMPLED depth
 * (stn.py:271: through the depth-consistency weight and term only): the gradient below contains both -- the second by the adjoint
 * of the bilinear sample, a scatter over the four taps -- and equals reference autograd (goldens `full`, `fullinit`).
 * Gauss-Newton model: forward group of a target = the joint dense system of linearize_joint (curvature per (pixel, source), depth
 * block diagonal, Schur complement on 6 S poses) + the IRLS curvature of its depth-consistency terms + a DIAGONAL model of the
 * 3x3-coupled prior, D_q += w / r^2 (1 / d2_q + 1 / (9 d1_q)) (d1, d2 = the SSIM denominators at q, r = 1/min_depth - 1/max_depth;
 * the diagonal of the Gauss-Newton matrix of SSIM's mean / covariance decomposition with the pixel's own denominators); the
 * sampled-depth terms and e dW/d. terms are gradient-only, as in every other mode.  Inverse pairs: 6 x 6 pose systems under the
 * window REFERENCE rule (linearize_masked with the batch normaliser), their depth is not an unknown.
 */
typedef struct { double *Dq, *Bq, *Hs, *gs; } dref_src_sys;   /* [SB][n], [SB][n][6], [SB][36] (reduced), [SB][6] (reduced) */
/* Gradient of the same loss w.r.t. the SOURCE inverse-depth maps rho_s(m, q) = 1 / depth_s(m, q), m = (s, b) -- the leaves of the
 * reference's optimize_depth_pred that the engine holds fixed (optimizer.py:194-198: the quarter-resolution disparities of the target AND
 * of every source are optimised).  Their role mirrors the target map's:
 *   (i)  LOCAL in the inverse pair SB + m, whose back-projected depth they are: the photometric term a_i M W diff through the sample
 *        position of every window pixel (SSIM adjoint) and through the pair's own weight W = 1 - dd, and the pair's depth-consistency term;
 *   (ii) SAMPLED by the forward pair m (stn.py:271): through the weight map it provides (source 0's map multiplies every selected pixel
 *        under argmin, optimizer.py:69; its own pixels otherwise) and through its depth-consistency term -- the adjoint of the bilinear
 *        sample, a scatter over the four taps.
 * Natural decisions only (no forced replay); masks / errors / normalisers are the caller's (linearize_dense_ref).  Pinned on reference
 * autograd: golden G13 `full_grad_depth_s` and the source channels of `qinit_grad_q` (tests/test_oracle_vs_golden.py). */
static void dref_source_depth_gradient(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                                       const real *K, const orc_opts *op, int argmin, const double *T, const real *const *ae, const real *imask /* [SB][n] */,
                                       const real *fmask /* [SB][n] */, const real *fdiff /* [SB][n] */, double a_f, double a_i, double bdc,
                                       const unsigned short *bits /* [2SB][n] or NULL: forced replay */, double lambda_depth, double *g_rho_s,
                                       double *Dq_s /* [SB][n] or NULL */, double *Bq_s /* [SB][n][6] or NULL */, double *Hs /* [SB][36]: reduced pose systems, or NULL */,
                                       double *gs /* [SB][6] */) {
    const int n = H * W, SB = S * B;
    const real wl = (real)(op->w_l1 / 3), ws = (real)(op->w_ssim / 3), reps = (real)op->irls_eps;
    const double eps = op->irls_eps;
    (void)ae;
    memset(g_rho_s, 0, sizeof(double) * (size_t)SB * n);
    px_t *px = (px_t *)malloc(sizeof(px_t) * n);
    real *rec = (real *)malloc(sizeof(real) * 3 * n);
    double *adj = (double *)malloc(sizeof(double) * 2 * n), *Lam = (double *)calloc((size_t)3 * n, sizeof(double));
    for (int m = 0; m < SB; m++) {
        const int b = m % B, pn = SB + m;
        const real *ti = srcs + (size_t)m * 3 * n;      /* the inverse pair's target image = source image m */
        const real *ds = depth_s + (size_t)m * n;
        double *gr = g_rho_s + (size_t)m * n;
        /* ---- (ii) first: the forward pair m samples this map: adjoint of the bilinear sample (its sums join the local gradient below) ---- */
        cam_t c;
        cam_setup(&c, H, W, K + 9 * b, T + 12 * m, 0.0);
        const int provider = argmin ? (m / B == 0) : 1;        /* does this pair's weight map multiply photometric terms (optimizer.py:69) */
        g_force_bits = bits ? bits + (size_t)m * n : NULL;
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const int i = v * W + u;
                geo_t g;
                warp_geo(&c, u, v, depth_t[(size_t)b * n + i], &g);
                if (g.oobx || g.ooby) continue;
                real pdv, dgx, dgy;
                bilinear_cell(ds, H, W, g.ix, g.iy, g.adjx, g.adjy, &pdv, &dgx, &dgy);
                const real cd = g.Z, pd = pdv, sum = cd + pd, dif = cd - pd, raw = fabs(dif) / sum;
                if (!(raw >= 0 && raw <= 1)) continue;
                const real sg = forced_sign(dif, (real)1e-6 * sum, i, 4);
                const double ddd = -(double)sg * 2.0 * cd / ((double)sum * sum);              /* d dd / d pd */
                const double dd = clamp01(raw);
                double E = 0;                                                                 /* sum of M diff of the pixels this weight multiplies */
                if (provider) {
                    if (argmin) for (int s2 = 0; s2 < S; s2++) E += fmask[(size_t)(s2 * B + b) * n + i] * fdiff[(size_t)(s2 * B + b) * n + i];
                    else E = fmask[(size_t)m * n + i] * fdiff[(size_t)m * n + i];
                }
                const double coef = (bdc * fmin(1.0, dd / eps) - a_f * E) * ddd;             /* d L / d pd(p) */
                const real fx = floor(g.ix) + (real)g.adjx, fy = floor(g.iy) + (real)g.adjy, wx = g.ix - fx, wy = g.iy - fy;
                const int x0 = (int)fx, y0 = (int)fy;
                const double w4[4] = {(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy};
                for (int t = 0; t < 4; t++) {
                    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) gr[yy * W + xx] -= (double)ds[yy * W + xx] * ds[yy * W + xx] * coef * w4[t];   /* d / d rho = -depth^2 d / d depth */
                }
            }
        /* ---- (i) the inverse pair: per-pixel quantities with the inverse-depth column ---- */
        cam_setup(&c, H, W, K + 9 * b, T + 12 * pn, 0.0);
        g_force_bits = bits ? bits + (size_t)pn * n : NULL;
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                px_t *P = &px[v * W + u];
                px_eval(&c, tgt + (size_t)b * 3 * n, ds, depth_t + (size_t)b * n, u, v, 7, P);
                const real D = ds[v * W + u];
                P->a[6] *= -D; P->b[6] *= -D; P->zc[6] *= -D;
                P->dpd[6] = P->dgx * P->a[6] + P->dgy * P->b[6];
            }
        for (int i = 0; i < n; i++)
            for (int ch = 0; ch < 3; ch++) rec[ch * n + i] = px[i].rec[ch];
        memset(adj, 0, sizeof(double) * 2 * n);
        memset(Lam, 0, sizeof(double) * 3 * n);
        double gxi[6] = {0, 0, 0, 0, 0, 0}, Hxx[36], Sm[36], gsv[6] = {0, 0, 0, 0, 0, 0};
        memset(Hxx, 0, sizeof(Hxx)); memset(Sm, 0, sizeof(Sm));
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const int i = v * W + u;
                const px_t *P = &px[i];
                const double am = imask[(size_t)m * n + i];
                if (am == 0) continue;
                const real sum = P->cd + P->pd, dif = P->cd - P->pd, raw = fabs(dif) / sum, dd = clamp01(raw);
                const real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
                const double kdd = sg * 2.0 / ((double)sum * sum);
                const real Wt = 1 - dd;
                real e = 0;
                double lxx = 0, lxy = 0, lyy = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const real *xx = ti + ch * n, *y = rec + ch * n;
                    const real r = y[i] - xx[i], ar = fabs(r);
                    const real sgn = (ar <= 1) ? forced_sign(r, (real)1e-6, i, 6 + 2 * ch) : (real)0;
                    adj[2 * i] += am * Wt * wl * sgn * P->gx[ch];
                    adj[2 * i + 1] += am * Wt * wl * sgn * P->gy[ch];
                    if (ar <= 1) {
                        real w1 = wl * Wt / (ar > reps ? ar : reps);
                        lxx += w1 * P->gx[ch] * P->gx[ch]; lxy += w1 * P->gx[ch] * P->gy[ch]; lyy += w1 * P->gy[ch] * P->gy[ch];
                    }
                    ssim_t q;
                    ssim_at(xx, y, H, W, u, v, &q);
                    e += wl * clamp01(ar) + ws * q.s;
                    if (!q.clamped) {
                        const real nn = q.n1 * q.n2, dn = q.d1 * q.d2, ratio = nn / dn, pre = -(real)0.5 / dn / 9;
                        const real cA = pre * (2 * q.mux * q.n2 - 2 * q.n1 * q.mux - ratio * (2 * q.muy * q.d2 - 2 * q.d1 * q.muy));
                        const real cB = pre * (-ratio * 2 * q.d1), cC = pre * (2 * q.n1);
                        real Sx = 0, Sy = 0;
                        for (int dv = -1; dv <= 1; dv++)
                            for (int du = -1; du <= 1; du++) {
                                const int qi = refl(v + dv, H) * W + refl(u + du, W);
                                const real cf = ws * (cA + cB * y[qi] + cC * xx[qi]);
                                adj[2 * qi] += am * Wt * cf * px[qi].gx[ch];
                                adj[2 * qi + 1] += am * Wt * cf * px[qi].gy[ch];
                                Sx += px[qi].gx[ch]; Sy += px[qi].gy[ch];
                            }
                        const real ninth = (real)1 / 9;
                        real mx = Sx * ninth, my = Sy * ninth, ex = P->gx[ch] - mx, ey = P->gy[ch] - my;
                        real w2 = ws * Wt / q.d2 * (real)1.125, w3 = ws * Wt / q.d1;
                        lxx += w2 * ex * ex + w3 * mx * mx; lxy += w2 * ex * ey + w3 * mx * my; lyy += w2 * ey * ey + w3 * my * my;
                    }
                }
                Lam[3 * i] = lxx; Lam[3 * i + 1] = lxy; Lam[3 * i + 2] = lyy;
                /* the pair's own weight: -M diff d dd / d theta */
                for (int j = 0; j < 6; j++) gxi[j] -= a_i * am * e * kdd * (P->pd * P->zc[j] - P->cd * P->dpd[j]);
                gr[i] -= a_i * am * e * kdd * (P->pd * P->zc[6] - P->cd * P->dpd[6]);
            }
        for (int i = 0; i < n; i++) {
            const px_t *P = &px[i];
            const double ax = a_i * adj[2 * i], ay = a_i * adj[2 * i + 1];
            for (int j = 0; j < 6; j++) gxi[j] += ax * P->a[j] + ay * P->b[j];
            gr[i] += ax * P->a[6] + ay * P->b[6];
            const double am = a_i * imask[(size_t)m * n + i];
            const double lxx = am * Lam[3 * i], lxy = am * Lam[3 * i + 1], lyy = am * Lam[3 * i + 2];
            /* depth consistency of the inverse pair (every pixel of the image) */
            const real sum = P->cd + P->pd, dif = P->cd - P->pd, raw = fabs(dif) / sum, dd = clamp01(raw);
            const real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
            double ddJ[7];
            for (int j = 0; j < 7; j++) ddJ[j] = sg * 2.0 * (P->pd * P->zc[j] - P->cd * P->dpd[j]) / ((double)sum * sum);
            const double inf = bdc * fmin(1.0, dd / eps), k3 = P->dc_in ? bdc / fmax((double)dd, eps) : 0.0;
            gr[i] += inf * ddJ[6];
            const double la6 = lxx * P->a[6] + lxy * P->b[6], lb6 = lxy * P->a[6] + lyy * P->b[6];
            double D = la6 * P->a[6] + lb6 * P->b[6] + k3 * ddJ[6] * ddJ[6], Bv[6];
            for (int j = 0; j < 6; j++) {
                gxi[j] += inf * ddJ[j];
                const double la = lxx * P->a[j] + lxy * P->b[j], lb = lxy * P->a[j] + lyy * P->b[j];
                Bv[j] = la * P->a[6] + lb * P->b[6] + k3 * ddJ[j] * ddJ[6];
                for (int k = 0; k <= j; k++) Hxx[j * 6 + k] += la * P->a[k] + lb * P->b[k] + k3 * ddJ[j] * ddJ[k];
            }
            if (P->valid && !P->dc_in) D = 0;                   /* sampled across the zero padding: the pixel keeps its depth */
            if (Dq_s) Dq_s[(size_t)m * n + i] = D;
            if (Bq_s) memcpy(Bq_s + ((size_t)m * n + i) * 6, Bv, sizeof(Bv));
            const double Dd = (1.0 + lambda_depth) * D;
            if (Dd > 1e-30)
                for (int j = 0; j < 6; j++) {
                    gsv[j] -= Bv[j] * gr[i] / Dd;
                    for (int k = 0; k <= j; k++) Sm[j * 6 + k] -= Bv[j] * Bv[k] / Dd;
                }
        }
        if (Hs && gs)
            for (int j = 0; j < 6; j++) {
                gs[6 * m + j] = gxi[j] + gsv[j];
                for (int k = 0; k <= j; k++) { const double vv = Hxx[j * 6 + k] + Sm[j * 6 + k]; Hs[36 * m + j * 6 + k] = vv; Hs[36 * m + k * 6 + j] = vv; }
            }
    }
    g_force_bits = NULL;
    free(px); free(rec); free(adj); free(Lam);
}

typedef struct {
    double loss, L_fwd, L_inv, L_dc, L_init, Kf, Ki;
} dref_scal;

static void linearize_dense_ref(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s,
                                const real *depth0, const real *K, const orc_opts *op, int argmin, double w_init, double min_depth, double max_depth,
                                double lambda_depth, const double *T /* [2SB][12] */, const real *const *ae /* [2SB] */, const unsigned short *bits /* [2SB][n] or NULL */,
                                dref_scal *sc, double *g_xi /* [2SB][6] */, double *g_rho /* [B][n] */, double *Hj /* [B][6S x 6S] */, double *gj /* [B][6S] */,
                                double *Dq /* [B][n] */, double *Bq /* [B][n][6S] */, double *Hi /* [SB][36] */, double *gi /* [SB][6] */,
                                double *g_rho_s /* [SB][n] or NULL: d L / d (1 / depth_s) -- see dref_source_depth_gradient */,
                                const dref_src_sys *src_sys /* or NULL: the inverse pairs' systems with the source map as a second unknown */) {
    const int n = H * W, SB = S * B, NP = 6 * S;
    const real wl = (real)(op->w_l1 / 3), ws = (real)(op->w_ssim / 3), reps = (real)op->irls_eps;
    const double bdc = op->w_dc / ((double)SB * n), eps = op->irls_eps;
    const double mind = 1.0 / max_depth, rd = 1.0 / min_depth - 1.0 / max_depth;
    memset(sc, 0, sizeof(*sc));
    memset(g_xi, 0, sizeof(double) * 12 * SB);
    memset(g_rho, 0, sizeof(double) * (size_t)B * n);
    /* ---------- inverse pairs: masks, the batch normaliser K_i ---------- */
    real *imask = (real *)malloc(sizeof(real) * (size_t)SB * n);
    double Ki = 0;
    for (int m = 0; m < SB; m++) {
        const int b = m % B;
        real *d = (real *)malloc(sizeof(real) * n), *va = (real *)malloc(sizeof(real) * n), *am = (real *)malloc(sizeof(real) * n);
        g_force_bits = NULL;
        orc_photometric(H, W, srcs + (size_t)m * 3 * n, tgt + (size_t)b * 3 * n, depth_s + (size_t)m * n, depth_t + (size_t)b * n, T + 12 * (SB + m), K + 9 * b, 0.0,
                        op->w_l1, op->w_ssim, d, va, NULL, NULL, am, NULL);
        for (int i = 0; i < n; i++) {
            real mk = va[i] * (op->automask ? am[i] : 1);
            if (bits) mk = (real)(bits[(size_t)(SB + m) * n + i] & 1);
            imask[(size_t)m * n + i] = mk; Ki += mk;
        }
        free(d); free(va); free(am);
    }
    sc->Ki = Ki;
    const double a_i = Ki > 0 ? 0.25 / Ki : 0.0;
    /* ---------- inverse pairs: pose systems under the window rule, and the scatter of d L / d (sampled target depth) ---------- */
    double *ext = (double *)calloc((size_t)B * n, sizeof(double));       /* d L / d depth_t(b, q) through the inverse pairs' samples */
    for (int m = 0; m < SB; m++) {
        const int b = m % B, pn = SB + m;
        lin_ext x;
        memset(&x, 0, sizeof(x));
        x.norm = Ki; x.scale = 0.25; x.b_dc = bdc;
        orc_opts o6 = *op;
        o6.nparam = 6;
        lin_t L;
        real *Eo = (real *)malloc(sizeof(real) * 3 * n), *Mo = (real *)malloc(sizeof(real) * n);
        g_force_bits = bits ? bits + (size_t)pn * n : NULL;
        linearize_masked(H, W, srcs + (size_t)m * 3 * n, tgt + (size_t)b * 3 * n, depth_s + (size_t)m * n, depth_t + (size_t)b * n, T + 12 * pn, K + 9 * b, 0.0,
                         &o6, ae[pn], bits ? NULL : imask + (size_t)m * n, &x, &L, NULL, NULL, NULL, Eo, Mo);
        for (int j = 0; j < 6; j++) { g_xi[6 * pn + j] = L.g[j]; if (gi) gi[6 * m + j] = L.g[j]; }
        if (Hi) for (int j = 0; j < 36; j++) Hi[36 * m + j] = L.H[(j / 6) * 6 + (j % 6)];
        sc->L_inv += L.cost_photo; sc->L_dc += L.cost_dc;
        /* scatter: term(p) = a_i M W diff + b dd, W = 1 - dd, dd = clamp(|cd - pd| / (cd + pd)); pd = sample of the target depth */
        cam_t c;
        cam_setup(&c, H, W, K + 9 * b, T + 12 * pn, 0.0);
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const int i = v * W + u;
                geo_t g;
                warp_geo(&c, u, v, depth_s[(size_t)m * n + i], &g);
                if (g.oobx || g.ooby) continue;
                real pdv, dgx, dgy;
                bilinear_cell(depth_t + (size_t)b * n, H, W, g.ix, g.iy, g.adjx, g.adjy, &pdv, &dgx, &dgy);
                const real cd = g.Z, pd = pdv, sum = cd + pd, dif = cd - pd, raw = fabs(dif) / sum;
                if (!(raw >= 0 && raw <= 1)) continue;
                const real sg = forced_sign(dif, (real)1e-6 * sum, i, 4);
                const double ddd = -(double)sg * 2.0 * cd / ((double)sum * sum);          /* d dd / d pd */
                const double Wd = (double)Eo[3 * i] + Eo[3 * i + 1];                      /* W (e1 + e2) */
                const double dd = Eo[3 * i + 2], Wt = 1.0 - dd, diff = Wt > 0 ? Wd / Wt : 0.0;
                const double coef = (bdc * fmin(1.0, dd / eps) - a_i * Mo[i] * diff) * ddd;
                const real fx = floor(g.ix) + (real)g.adjx, fy = floor(g.iy) + (real)g.adjy, wx = g.ix - fx, wy = g.iy - fy;
                const int x0 = (int)fx, y0 = (int)fy;
                const double w4[4] = {(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy};
                for (int t = 0; t < 4; t++) {
                    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) ext[(size_t)b * n + yy * W + xx] += coef * w4[t];
                }
            }
        free(Eo); free(Mo);
    }
    g_force_bits = NULL;
    /* ---------- forward pairs: per-pixel quantities, masks, the batch normaliser K_f ---------- */
    px_t **px = (px_t **)malloc(sizeof(px_t *) * SB);
    real **rec = (real **)malloc(sizeof(real *) * SB);
    real *diff = (real *)malloc(sizeof(real) * (size_t)SB * n), *valid = (real *)malloc(sizeof(real) * (size_t)SB * n), *Wm = (real *)malloc(sizeof(real) * (size_t)SB * n);
    real *mask = (real *)calloc((size_t)SB * n, sizeof(real)), *margin = (real *)malloc(sizeof(real) * n);
    for (int m = 0; m < SB; m++) {
        const int b = m % B;
        cam_t c;
        cam_setup(&c, H, W, K + 9 * b, T + 12 * m, 0.0);
        px[m] = (px_t *)malloc(sizeof(px_t) * n);
        rec[m] = (real *)malloc(sizeof(real) * 3 * n);
        g_force_bits = bits ? bits + (size_t)m * n : NULL;
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                px_t *P = &px[m][v * W + u];
                px_eval(&c, srcs + (size_t)m * 3 * n, depth_t + (size_t)b * n, depth_s + (size_t)m * n, u, v, 7, P);
                const real D = depth_t[(size_t)b * n + v * W + u];        /* scale column -> inverse-depth column; the source depth is fixed */
                P->a[6] *= -D; P->b[6] *= -D; P->zc[6] *= -D;
                P->dpd[6] = P->dgx * P->a[6] + P->dgy * P->b[6];
            }
        for (int i = 0; i < n; i++)
            for (int ch = 0; ch < 3; ch++) rec[m][ch * n + i] = px[m][i].rec[ch];
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const int i = v * W + u;
                const px_t *P = &px[m][i];
                real e = 0;
                for (int ch = 0; ch < 3; ch++) {
                    ssim_t q;
                    ssim_at(tgt + (size_t)b * 3 * n + ch * n, rec[m] + ch * n, H, W, u, v, &q);
                    e += wl * clamp01(fabs(rec[m][ch * n + i] - tgt[(size_t)b * 3 * n + ch * n + i])) + ws * q.s;
                }
                diff[(size_t)m * n + i] = e; Wm[(size_t)m * n + i] = 1 - clamp01(fabs(P->cd - P->pd) / (P->cd + P->pd));
                valid[(size_t)m * n + i] = (real)P->nat_valid;
            }
    }
    g_force_bits = NULL;
    for (int b = 0; b < B; b++) {
        if (argmin && S > 1) {
            real *df = (real *)malloc(sizeof(real) * (size_t)S * n), *vf = (real *)malloc(sizeof(real) * (size_t)S * n), *af = (real *)malloc(sizeof(real) * (size_t)S * n);
            for (int s = 0; s < S; s++) {
                memcpy(df + (size_t)s * n, diff + (size_t)(s * B + b) * n, sizeof(real) * n);
                memcpy(vf + (size_t)s * n, valid + (size_t)(s * B + b) * n, sizeof(real) * n);
                memcpy(af + (size_t)s * n, ae[s * B + b], sizeof(real) * n);
            }
            window_select_maps(H, W, S, df, vf, af, op->automask, mask + (size_t)b * n, (size_t)B * n, margin);
            free(df); free(vf); free(af);
        } else {
            for (int s = 0; s < S; s++) {
                const int m = s * B + b, am = op->automask && (argmin || S == 1) && argmin;   /* no argmin: no auto-mask (:71-73) */
                for (int i = 0; i < n; i++)
                    mask[(size_t)m * n + i] = (valid[(size_t)m * n + i] > 0 && (!am || diff[(size_t)m * n + i] < ae[m][i])) ? 1 : 0;
            }
            for (int i = 0; i < n; i++) margin[i] = (real)1e30;
        }
        if (bits)
            for (int s = 0; s < S; s++) {
                const int m = s * B + b;
                for (int i = 0; i < n; i++) {
                    const int fm = bits[(size_t)m * n + i] & 1;
                    flip_note(mask[(size_t)m * n + i], fm, (argmin && S > 1) ? margin[i] < ORC_TIE
                              : (px[m][i].nat_valid != px[m][i].valid) || fabs(diff[(size_t)m * n + i] - ae[m][i]) < ORC_TIE);
                    mask[(size_t)m * n + i] = (real)fm;
                }
            }
    }
    double Kf = 0;
    for (size_t i = 0; i < (size_t)SB * n; i++) Kf += mask[i];
    sc->Kf = Kf;
    const double a_f = Kf > 0 ? (argmin ? 1.0 : 0.25) / Kf : 0.0;
    /* ---------- forward group of every target ---------- */
    double *gx_adj = (double *)malloc(sizeof(double) * (size_t)n * 2 * S), *Lam = (double *)malloc(sizeof(double) * (size_t)n * 3 * S);
    double *own = (double *)malloc(sizeof(double) * n), *gxi = (double *)malloc(sizeof(double) * NP), *Hxx = (double *)malloc(sizeof(double) * (size_t)NP * NP);
    double *Sm = (double *)malloc(sizeof(double) * (size_t)NP * NP), *gs = (double *)malloc(sizeof(double) * NP);
    double *pri_g = (double *)malloc(sizeof(double) * n), *pri_D = (double *)malloc(sizeof(double) * n);
    real *sig = (real *)malloc(sizeof(real) * n), *sig0 = (real *)malloc(sizeof(real) * n);
    for (int b = 0; b < B; b++) {
        const real *tg = tgt + (size_t)b * 3 * n;
        memset(gx_adj, 0, sizeof(double) * (size_t)n * 2 * S); memset(Lam, 0, sizeof(double) * (size_t)n * 3 * S);
        memset(own, 0, sizeof(double) * n); memset(gxi, 0, sizeof(double) * NP); memset(Hxx, 0, sizeof(double) * (size_t)NP * NP);
        memset(Sm, 0, sizeof(double) * (size_t)NP * NP); memset(gs, 0, sizeof(double) * NP);
        memset(pri_g, 0, sizeof(double) * n); memset(pri_D, 0, sizeof(double) * n);
        for (int s = 0; s < S; s++) {
            const int m = s * B + b, x = (argmin ? 0 : s) * B + b;      /* whose weight map multiplies this source's pixels (:69) */
            g_force_bits = bits ? bits + (size_t)m * n : NULL;
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) {
                    const int i = v * W + u;
                    const double am = mask[(size_t)m * n + i];
                    if (am == 0) continue;
                    const px_t *P = &px[m][i];
                    const real Wt = Wm[(size_t)x * n + i];
                    sc->L_fwd += a_f * am * Wt * diff[(size_t)m * n + i];
                    double lxx = 0, lxy = 0, lyy = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        const real *xx = tg + ch * n, *y = rec[m] + ch * n;
                        real r = y[i] - xx[i], ar = fabs(r);
                        real sgn = (ar <= 1) ? forced_sign(r, (real)1e-6, i, 6 + 2 * ch) : (real)0;
                        gx_adj[((size_t)s * n + i) * 2] += am * Wt * wl * sgn * P->gx[ch];
                        gx_adj[((size_t)s * n + i) * 2 + 1] += am * Wt * wl * sgn * P->gy[ch];
                        if (ar <= 1) {
                            real w1 = wl * Wt / (ar > reps ? ar : reps);
                            lxx += w1 * P->gx[ch] * P->gx[ch]; lxy += w1 * P->gx[ch] * P->gy[ch]; lyy += w1 * P->gy[ch] * P->gy[ch];
                        }
                        ssim_t q;
                        ssim_at(xx, y, H, W, u, v, &q);
                        if (!q.clamped) {
                            real nn = q.n1 * q.n2, dn = q.d1 * q.d2, ratio = nn / dn, pre = -(real)0.5 / dn / 9;
                            real cA = pre * (2 * q.mux * q.n2 - 2 * q.n1 * q.mux - ratio * (2 * q.muy * q.d2 - 2 * q.d1 * q.muy));
                            real cB = pre * (-ratio * 2 * q.d1), cC = pre * (2 * q.n1);
                            real Sx = 0, Sy = 0;
                            for (int dv = -1; dv <= 1; dv++)
                                for (int du = -1; du <= 1; du++) {
                                    int qi = refl(v + dv, H) * W + refl(u + du, W);
                                    real cf = ws * (cA + cB * y[qi] + cC * xx[qi]);
                                    gx_adj[((size_t)s * n + qi) * 2] += am * Wt * cf * px[m][qi].gx[ch];
                                    gx_adj[((size_t)s * n + qi) * 2 + 1] += am * Wt * cf * px[m][qi].gy[ch];
                                    Sx += px[m][qi].gx[ch]; Sy += px[m][qi].gy[ch];
                                }
                            const real ninth = (real)1 / 9;
                            real mx = Sx * ninth, my = Sy * ninth, ex = P->gx[ch] - mx, ey = P->gy[ch] - my;
                            real w2 = ws * Wt / q.d2 * (real)1.125, w3 = ws * Wt / q.d1;
                            lxx += w2 * ex * ex + w3 * mx * mx; lxy += w2 * ex * ey + w3 * mx * my; lyy += w2 * ey * ey + w3 * my * my;
                        }
                    }
                    Lam[((size_t)s * n + i) * 3] = lxx; Lam[((size_t)s * n + i) * 3 + 1] = lxy; Lam[((size_t)s * n + i) * 3 + 2] = lyy;
                }
            /* weight terms: -M_s diff_s d dd_x / d theta at the pixel itself (sign of cd_x - pd_x: pair x's code) */
            g_force_bits = bits ? bits + (size_t)x * n : NULL;
            const int sx_ = argmin ? 0 : s;
            for (int i = 0; i < n; i++) {
                const double am = mask[(size_t)m * n + i];
                if (am == 0) continue;
                const px_t *X = &px[x][i];
                real sum = X->cd + X->pd, dif = X->cd - X->pd, raw = fabs(dif) / sum;
                real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
                double kdd = sg * 2.0 / ((double)sum * sum), e = diff[(size_t)m * n + i];
                for (int j = 0; j < 6; j++) gxi[6 * sx_ + j] -= am * e * kdd * (X->pd * X->zc[j] - X->cd * X->dpd[j]);
                own[i] -= am * e * kdd * (X->pd * X->zc[6] - X->cd * X->dpd[6]);
            }
        }
        g_force_bits = NULL;
        /* the prior: SSIM between the current and the initial sigmoid disparity of the target (optimizer.py:89-90) */
        if (w_init > 0 && depth0) {
            const double wp = w_init / ((double)B * n);
            for (int i = 0; i < n; i++) {
                sig[i] = (real)((1.0 / (double)depth_t[(size_t)b * n + i] - mind) / rd);
                sig0[i] = (real)((1.0 / (double)depth0[(size_t)b * n + i] - mind) / rd);
            }
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) {
                    ssim_t q;
                    ssim_at(sig0, sig, H, W, u, v, &q);          /* (symmetric in its arguments; differentiated w.r.t. the second, as everywhere) */
                    sc->L_init += wp * q.s;
                    /* SSIM <= 1, so the loss value (1 - SSIM) / 2 is never below 0 except by rounding -- which is exactly what happens at
                     * the first linearisation, where sigma == sigma0 bit for bit: only the upper clamp switches the derivative off */
                    if ((1 - (q.n1 * q.n2) / (q.d1 * q.d2)) / 2 > 1) continue;
                    real nn = q.n1 * q.n2, dn = q.d1 * q.d2, ratio = nn / dn, pre = -(real)0.5 / dn / 9;
                    real cA = pre * (2 * q.mux * q.n2 - 2 * q.n1 * q.mux - ratio * (2 * q.muy * q.d2 - 2 * q.d1 * q.muy));
                    real cB = pre * (-ratio * 2 * q.d1), cC = pre * (2 * q.n1);
                    for (int dv = -1; dv <= 1; dv++)
                        for (int du = -1; du <= 1; du++) {
                            int qi = refl(v + dv, H) * W + refl(u + du, W);
                            pri_g[qi] += wp * (cA + cB * sig[qi] + cC * sig0[qi]) / rd;          /* d sigma / d rho = 1 / r */
                        }
                    pri_D[v * W + u] = wp / (rd * rd) * (1.0 / q.d2 + 1.0 / (9.0 * q.d1));
                }
        }
        /* l_smooth (optimizer.py:92-93, losses.py:43-61): w_s [mean_x |dx d^| e^{-|dx I|} + mean_y ...] of the mean-normalised disparity
         * d^ = sigma / (mean_image(sigma) + 1e-7); the means run over B H (W-1) and B (H-1) W edges.  Per edge e = (i, j) with weight
         * c_e:  d/d sigma_p  c_e |d^_i - d^_j|  =  c_e s_e (delta_ip - delta_jp) / m  -  c_e |d^_i - d^_j| / (m HW): a local part and a
         * per-image constant -T_b / (m HW), T_b = the image's whole term.  Curvature: the local part's IRLS weights c_e / (m^2 max(|d^_i -
         * d^_j|, eps)), as the diagonal majoriser 2 x (an edge moves with both ends); the normalisation's part is gradient-only. */
        if (op->w_smooth > 0) {
            const real *img = tgt + (size_t)b * 3 * n;
            double mu = 0;
            for (int i = 0; i < n; i++) {
                sig[i] = (real)((1.0 / (double)depth_t[(size_t)b * n + i] - mind) / rd);
                mu += sig[i];
            }
            const double m = mu / n + 1e-7, cx = op->w_smooth / ((double)B * H * (W - 1)), cy = op->w_smooth / ((double)B * (H - 1) * W), eps_s = op->irls_eps;
            double Tb = 0;
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++)
                    for (int dir = 0; dir < 2; dir++) {
                        if (dir == 0 ? u + 1 >= W : v + 1 >= H) continue;
                        const int i = v * W + u, j = dir == 0 ? i + 1 : i + W;
                        double gi_ = 0;
                        for (int ch = 0; ch < 3; ch++) gi_ += fabs((double)img[(size_t)ch * n + i] - (double)img[(size_t)ch * n + j]);
                        const double ce = (dir == 0 ? cx : cy) * exp(-gi_ / 3.0);
                        const double dd_ = ((double)sig[i] - (double)sig[j]) / m, ad = fabs(dd_), den = ad > eps_s ? ad : eps_s;
                        Tb += ce * ad;
                        const double gl = ce * dd_ / den / m;                 /* = c_e s_e / m away from zero */
                        pri_g[i] += gl / rd; pri_g[j] -= gl / rd;
                        const double dl = 2.0 * ce / (m * m * den) / (rd * rd);
                        pri_D[i] += dl; pri_D[j] += dl;
                    }
            for (int i = 0; i < n; i++) pri_g[i] -= Tb / (m * n) / rd;
            sc->L_init += Tb;                                       /* (booked with the prior: both are terms on the map alone) */
        }
        /* assemble */
        for (int i = 0; i < n; i++) {
            const double dep = depth_t[(size_t)b * n + i];
            double gr = a_f * own[i] + pri_g[i] - dep * dep * ext[(size_t)b * n + i], D = pri_D[i], Bv[6 * JMAXS];
            for (int s = 0; s < S; s++) {
                const int m = s * B + b;
                const px_t *P = &px[m][i];
                const double ax = a_f * gx_adj[((size_t)s * n + i) * 2], ay = a_f * gx_adj[((size_t)s * n + i) * 2 + 1];
                for (int j = 0; j < 6; j++) gxi[6 * s + j] += gx_adj[((size_t)s * n + i) * 2] * P->a[j] + gx_adj[((size_t)s * n + i) * 2 + 1] * P->b[j];
                gr += ax * P->a[6] + ay * P->b[6];
                const double am = a_f * mask[(size_t)m * n + i];
                const double lxx = am * Lam[((size_t)s * n + i) * 3], lxy = am * Lam[((size_t)s * n + i) * 3 + 1], lyy = am * Lam[((size_t)s * n + i) * 3 + 2];
                /* depth consistency of THIS forward pair (every pixel of the image) */
                real sum = P->cd + P->pd, dif = P->cd - P->pd, raw = fabs(dif) / sum, dd = clamp01(raw);
                g_force_bits = bits ? bits + (size_t)m * n : NULL;
                real sg = (raw >= 0 && raw <= 1) ? forced_sign(dif, (real)1e-6 * sum, i, 4) : (real)0;
                g_force_bits = NULL;
                double ddJ[7];
                for (int j = 0; j < 7; j++) ddJ[j] = sg * 2.0 * (P->pd * P->zc[j] - P->cd * P->dpd[j]) / ((double)sum * sum);
                sc->L_dc += bdc * dd;
                const double inf = bdc * fmin(1.0, dd / eps), k3 = P->dc_in ? bdc / fmax((double)dd, eps) : 0.0;
                gr += inf * ddJ[6];
                const double la6 = lxx * P->a[6] + lxy * P->b[6], lb6 = lxy * P->a[6] + lyy * P->b[6];
                D += la6 * P->a[6] + lb6 * P->b[6] + k3 * ddJ[6] * ddJ[6];
                for (int j = 0; j < 6; j++) {
                    g_xi[6 * m + j] += inf * ddJ[j];
                    const double la = lxx * P->a[j] + lxy * P->b[j], lb = lxy * P->a[j] + lyy * P->b[j];
                    Bv[6 * s + j] = la * P->a[6] + lb * P->b[6] + k3 * ddJ[j] * ddJ[6];
                    for (int k = 0; k <= j; k++) Hxx[(size_t)(6 * s + j) * NP + 6 * s + k] += la * P->a[k] + lb * P->b[k] + k3 * ddJ[j] * ddJ[k];
                }
            }
            for (int s = 0; s < S; s++)
                if (px[s * B + b][i].valid && !px[s * B + b][i].dc_in) D = 0;    /* sampled across the zero padding: the pixel keeps its depth */
            g_rho[(size_t)b * n + i] = gr;
            if (Dq) Dq[(size_t)b * n + i] = D;
            if (Bq) memcpy(Bq + ((size_t)b * n + i) * NP, Bv, sizeof(double) * NP);
            const double Dd = (1.0 + lambda_depth) * D;
            if (Dd > 1e-30)
                for (int j = 0; j < NP; j++) {
                    gs[j] -= Bv[j] * gr / Dd;
                    for (int k = 0; k <= j; k++) Sm[(size_t)j * NP + k] -= Bv[j] * Bv[k] / Dd;
                }
        }
        for (int s = 0; s < S; s++)
            for (int j = 0; j < 6; j++) g_xi[6 * (s * B + b) + j] += a_f * gxi[6 * s + j];
        if (Hj && gj)
            for (int j = 0; j < NP; j++) {
                gj[(size_t)b * NP + j] = g_xi[6 * ((j / 6) * B + b) + (j % 6)] + gs[j];
                for (int k = 0; k <= j; k++) {
                    const double v = Hxx[(size_t)j * NP + k] + Sm[(size_t)j * NP + k];
                    Hj[(size_t)b * NP * NP + j * NP + k] = v; Hj[(size_t)b * NP * NP + k * NP + j] = v;
                }
            }
    }
    sc->loss = sc->L_fwd + sc->L_inv + sc->L_dc + sc->L_init;
    if (g_rho_s)
        dref_source_depth_gradient(H, W, B, S, tgt, srcs, depth_t, depth_s, K, op, argmin, T, ae, imask, mask, diff, a_f, a_i, bdc, bits, lambda_depth, g_rho_s,
                                   src_sys ? src_sys->Dq : NULL, src_sys ? src_sys->Bq : NULL, src_sys ? src_sys->Hs : NULL, src_sys ? src_sys->gs : NULL);
    for (int m = 0; m < SB; m++) { free(px[m]); free(rec[m]); }
    free(px); free(rec); free(diff); free(valid); free(Wm); free(mask); free(margin); free(imask); free(ext);
    free(gx_adj); free(Lam); free(own); free(gxi); free(Hxx); free(Sm); free(gs); free(pri_g); free(pri_D); free(sig); free(sig0);
}

static void dref_auto_err(int H, int W, int B, int S, const real *tgt, const real *srcs, const orc_opts *op, real *ae /* [2SB][n] */, const real **aep) {
    const int n = H * W, SB = S * B;
    for (int m = 0; m < SB; m++) {
        const int b = m % B;
        photo_err_map(H, W, tgt + (size_t)b * 3 * n, srcs + (size_t)m * 3 * n, op->w_l1, op->w_ssim, ae + (size_t)m * n);
        photo_err_map(H, W, srcs + (size_t)m * 3 * n, tgt + (size_t)b * 3 * n, op->w_l1, op->w_ssim, ae + (size_t)(SB + m) * n);
        aep[m] = ae + (size_t)m * n; aep[SB + m] = ae + (size_t)(SB + m) * n;
    }
}

/* one linearisation at given poses: the loss, its gradients, the Gauss-Newton blocks (tests: reference autograd pin, golden G13) */
void orc_linearize_dense_ref(int H, int W, int B, int S, const real *tgt, const real *srcs, const real *depth_t, const real *depth_s, const real *depth0,
                             const real *K, const orc_opts *op, int argmin, double w_init, double min_depth, double max_depth, double lambda_depth,
                             const double *pose /* [2SB][6] */, double *scal /* [7]: loss, L_fwd, L_inv, L_dc, L_init, K_f, K_i */, double *g_xi, double *g_rho,
                             double *Hj, double *gj, double *Dq, double *Bq, double *Hi, double *gi, double *g_rho_s /* [SB][n] or NULL */) {
    const int n = H * W, SB = S * B;
    real *ae = (real *)malloc(sizeof(real) * (size_t)2 * SB * n);
    const real **aep = (const real **)malloc(sizeof(real *) * 2 * SB);
    double *T = (double *)malloc(sizeof(double) * 24 * SB);
    dref_auto_err(H, W, B, S, tgt, srcs, op, ae, aep);
    for (int m = 0; m < 2 * SB; m++) orc_pose_to_T(pose + 6 * m, T + 12 * m);
    dref_scal sc;
    linearize_dense_ref(H, W, B, S, tgt, srcs, depth_t, depth_s, depth0, K, op, argmin, w_init, min_depth, max_depth, lambda_depth, T, aep, NULL, &sc,
                        g_xi, g_rho, Hj, gj, Dq, Bq, Hi, gi, g_rho_s, NULL);
    if (op->w_pose_consist > 0) {      /* l_pose_consist (optimizer.py:95-96): value and the gradient w.r.t. every pair's left perturbation (pinned on golden G13 `full_pc`) */
        for (int m = 0; m < 2 * SB; m++) {
            double pc, pg[6], pH[36];
            pose_consist_term(T + 12 * m, T + 12 * (m < SB ? m + SB : m - SB), op->w_pose_consist / (6.0 * SB), op->irls_eps, &pc, pg, pH);
            sc.loss += pc;
            for (int i = 0; i < 6; i++) g_xi[6 * m + i] += pg[i];
        }
    }
    scal[0] = sc.loss; scal[1] = sc.L_fwd; scal[2] = sc.L_inv; scal[3] = sc.L_dc; scal[4] = sc.L_init; scal[5] = sc.Kf; scal[6] = sc.Ki;
    free(ae); free(aep); free(T);
}

/* Gauss-Newton on the reference loss: n_iters x { linearise ; joint step of every target's (poses, depth map) ; step of every
 * inverse pair's pose }, fixed damping lambda0 on the pose blocks (Marquardt-scaled) and lambda_depth on the depth block.
 * depth_io [B][n] in / out; pose_io [2SB][6]; stats [n_iters][7] (the scalars of every linearisation) or NULL;
 * bits [n_iters][2SB][n]: forced replay of the engine's decisions. */
/* l_pose_consist in the dense mode on the reference's loss (optimizer.py:95-96 beside :235-268): the term of pose_consist_term added to the
 * REDUCED pose systems (it does not depend on the depth maps) -- forward pair (s, b) to the diagonal 6 x 6 block s of target b's joint system,
 * inverse pair m to its own 6 x 6 system (Hi / gi, or the free-source system Hs / gs).  Every pair sees its partner at this linearisation;
 * returns the term's value (the pairs' halves add up to it). */
static double dref_pose_consist(int B, int S, const orc_opts *op, const double *T, double *Hj, double *gj, double *Hi, double *gi) {
    const int SB = S * B, NP = 6 * S;
    const double c = op->w_pose_consist / (6.0 * SB);
    double total = 0;
    for (int m = 0; m < 2 * SB; m++) {
        double pc, pg[6], pH[36];
        pose_consist_term(T + 12 * m, T + 12 * (m < SB ? m + SB : m - SB), c, op->irls_eps, &pc, pg, pH);
        total += pc;
        if (m < SB) {
            const int s = m / B, b = m % B;
            for (int i = 0; i < 6; i++) {
                gj[(size_t)b * NP + 6 * s + i] += pg[i];
                for (int j = 0; j < 6; j++) Hj[(size_t)b * NP * NP + (size_t)(6 * s + i) * NP + 6 * s + j] += pH[6 * i + j];
            }
        } else {
            for (int i = 0; i < 6; i++) {
                gi[6 * (m - SB) + i] += pg[i];
                for (int j = 0; j < 6; j++) Hi[36 * (m - SB) + 6 * i + j] += pH[6 * i + j];
            }
        }
    }
    return total;
}
static void refine_dense_ref_impl(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, real *depth_s /* in / out when free_sources */,
                                  const real *K, const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                                  double *pose_io, double *stats, const unsigned short *bits, int free_sources) {
    const int n = H * W, SB = S * B, NP = 6 * S;
    const double lo = 1.0 / max_depth, hi = 1.0 / min_depth;
    real *ae = (real *)malloc(sizeof(real) * (size_t)2 * SB * n), *d0 = (real *)malloc(sizeof(real) * (size_t)B * n);
    const real **aep = (const real **)malloc(sizeof(real *) * 2 * SB);
    double *T = (double *)malloc(sizeof(double) * 24 * SB);
    double *g_xi = (double *)malloc(sizeof(double) * 12 * SB), *g_rho = (double *)malloc(sizeof(double) * (size_t)B * n);
    double *Hj = (double *)malloc(sizeof(double) * (size_t)B * NP * NP), *gj = (double *)malloc(sizeof(double) * (size_t)B * NP);
    double *Dq = (double *)malloc(sizeof(double) * (size_t)B * n), *Bq = (double *)malloc(sizeof(double) * (size_t)B * n * NP);
    double *Hi = (double *)malloc(sizeof(double) * 36 * SB), *gi = (double *)malloc(sizeof(double) * 6 * SB);
    /* free_sources: the source maps are unknowns too -- every inverse pair is a group of its own (its pose + the source map it back-projects),
     * linearised by dref_source_depth_gradient: reduced 6 x 6 system Hs / gs, per-pixel g, D, B for the back-substitution */
    dref_src_sys ss = {NULL, NULL, NULL, NULL};
    double *g_rho_s = NULL;
    if (free_sources) {
        g_rho_s = (double *)malloc(sizeof(double) * (size_t)SB * n);
        ss.Dq = (double *)malloc(sizeof(double) * (size_t)SB * n); ss.Bq = (double *)malloc(sizeof(double) * (size_t)SB * n * 6);
        ss.Hs = (double *)malloc(sizeof(double) * 36 * SB); ss.gs = (double *)malloc(sizeof(double) * 6 * SB);
    }
    dref_auto_err(H, W, B, S, tgt, srcs, op, ae, aep);
    memcpy(d0, depth_io, sizeof(real) * (size_t)B * n);
    for (int m = 0; m < 2 * SB; m++) orc_pose_to_T(pose_io + 6 * m, T + 12 * m);
    for (int it = 0; it < op->n_iters; it++) {
        dref_scal sc;
        g_lin_idx = bits ? it : -1;
        linearize_dense_ref(H, W, B, S, tgt, srcs, depth_io, depth_s, d0, K, op, argmin, w_init, min_depth, max_depth, lambda_depth, T, aep,
                            bits ? bits + (size_t)it * 2 * SB * n : NULL, &sc, g_xi, g_rho, Hj, gj, Dq, Bq, Hi, gi, g_rho_s, free_sources ? &ss : NULL);
        g_lin_idx = -1;
        if (op->w_pose_consist > 0) sc.loss += dref_pose_consist(B, S, op, T, Hj, gj, free_sources ? ss.Hs : Hi, free_sources ? ss.gs : gi);
        if (stats) { double *r = stats + 7 * it; r[0] = sc.loss; r[1] = sc.L_fwd; r[2] = sc.L_inv; r[3] = sc.L_dc; r[4] = sc.L_init; r[5] = sc.Kf; r[6] = sc.Ki; }
        for (int b = 0; b < B; b++) {       /* forward group: (S + lambda diag S + 1e-12 I) d = -gS, back-substitution of the depth map */
            double A[36 * JMAXS * JMAXS], dl[6 * JMAXS];
            for (int i = 0; i < NP; i++) {
                for (int j = 0; j < NP; j++) A[i * NP + j] = Hj[(size_t)b * NP * NP + i * NP + j];
                A[i * NP + i] += op->lambda0 * Hj[(size_t)b * NP * NP + i * NP + i] + 1e-12;
                dl[i] = -gj[(size_t)b * NP + i];
            }
            if (chol_solve(NP, A, dl)) memset(dl, 0, sizeof(dl));
            for (int s = 0; s < S; s++) {
                double E[12], Tn[12];
                orc_se3_exp(dl + 6 * s, E);
                orc_se3_mul(E, T + 12 * (s * B + b), Tn);
                memcpy(T + 12 * (s * B + b), Tn, sizeof(Tn));
            }
            for (int i = 0; i < n; i++) {
                const double Dd = (1.0 + lambda_depth) * Dq[(size_t)b * n + i];
                if (!(Dd > 1e-30)) continue;
                double bd = 0;
                for (int j = 0; j < NP; j++) bd += Bq[((size_t)b * n + i) * NP + j] * dl[j];
                depth_io[(size_t)b * n + i] = (real)(1.0 / depth_step(1.0 / (double)depth_io[(size_t)b * n + i], -(g_rho[(size_t)b * n + i] + bd) / Dd, lo, hi));
            }
        }
        for (int m = 0; m < SB; m++) {      /* inverse pairs: their own 6 x 6 systems */
            if (free_sources) {             /* ... reduced by the source map: the same step as a forward group of one source */
                double A[36], dl[6];
                for (int i = 0; i < 6; i++) {
                    for (int j = 0; j < 6; j++) A[i * 6 + j] = ss.Hs[36 * m + i * 6 + j];
                    A[i * 6 + i] += op->lambda0 * ss.Hs[36 * m + i * 6 + i] + 1e-12;
                    dl[i] = -ss.gs[6 * m + i];
                }
                if (chol_solve(6, A, dl)) memset(dl, 0, sizeof(dl));
                double E[12], Tn[12];
                orc_se3_exp(dl, E);
                orc_se3_mul(E, T + 12 * (SB + m), Tn);
                memcpy(T + 12 * (SB + m), Tn, sizeof(Tn));
                for (int i = 0; i < n; i++) {
                    const double Dd = (1.0 + lambda_depth) * ss.Dq[(size_t)m * n + i];
                    if (!(Dd > 1e-30)) continue;
                    double bd = 0;
                    for (int j = 0; j < 6; j++) bd += ss.Bq[((size_t)m * n + i) * 6 + j] * dl[j];
                    depth_s[(size_t)m * n + i] = (real)(1.0 / depth_step(1.0 / (double)depth_s[(size_t)m * n + i], -(g_rho_s[(size_t)m * n + i] + bd) / Dd, lo, hi));
                }
                continue;
            }
            orc_opts o6 = *op;
            o6.nparam = 6; o6.param = 0;
            double Tn[12], sdummy;
            apply_step(&o6, Hi + 36 * m, gi + 6 * m, op->lambda0, T + 12 * (SB + m), 0.0, Tn, &sdummy);
            memcpy(T + 12 * (SB + m), Tn, sizeof(Tn));
        }
    }
    for (int m = 0; m < 2 * SB; m++) orc_T_to_pose(T + 12 * m, pose_io + 6 * m);
    free(ae); free(d0); free(aep); free(T); free(g_xi); free(g_rho); free(Hj); free(gj); free(Dq); free(Bq); free(Hi); free(gi);
    free(g_rho_s); free(ss.Dq); free(ss.Bq); free(ss.Hs); free(ss.gs);
}
void orc_refine_dense_ref(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, const real *depth_s, const real *K,
                          const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                          double *pose_io, double *stats, const unsigned short *bits) {
    refine_dense_ref_impl(H, W, B, S, tgt, srcs, depth_io, (real *)depth_s, K, op, argmin, w_init, lambda_depth, min_depth, max_depth, pose_io, stats, bits, 0);
}
/* the same with the SOURCE maps as unknowns (depth_s_io [SB][n] in / out): the reference's optimize_depth_pred optimises them too, with no prior */
void orc_refine_dense_ref_free(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, real *depth_s_io, const real *K,
                               const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                               double *pose_io, double *stats, const unsigned short *bits) {
    refine_dense_ref_impl(H, W, B, S, tgt, srcs, depth_io, depth_s_io, K, op, argmin, w_init, lambda_depth, min_depth, max_depth, pose_io, stats, bits, 1);
}

/* ------------------------------------------------------------------------- */
/* The reference's PARAMETRISATION of optimize_depth_pred (optimizer.py:194-198, 235-239): the unknown is the QUARTER-resolution      */
/* sigmoid disparity; every epoch upsamples it x4 (F.interpolate, bilinear, align_corners = False) and converts with disp_to_depth.  */
/* sigma and rho = 1 / depth are affine in each other and the interpolation weights of a pixel sum to one, so the quarter-resolution  */
/* inverse depth rho_q with rho = U rho_q is the same parametrisation.                                                                */
/*
 * torch's bilinear kernels (aten/native/UpSample.h area_pixel_compute_source_index, align_corners = False):
 *   upsample x4:   destination x reads source s = max((x + 0.5) / 4 - 0.5, 0): taps i0 = floor(s), i1 = min(i0 + 1, n - 1), weights 1 - l, l = s - i0
 *   downsample /4: destination i reads source 4 i + 1.5: taps 4 i + 1 and 4 i + 2, weights 1/2 (both axes: the mean of the block's 2 x 2 centre)
 * pinned on the reference's own F.interpolate outputs (golden G13 `q_sig`, `q_up`).
 */
static void up4_taps(int x, int nq, int *i0, int *i1, double *l1) {
    double s = (x + 0.5) * 0.25 - 0.5;
    if (s < 0) s = 0;
    int a = (int)floor(s);
    *i0 = a; *i1 = a + 1 < nq ? a + 1 : nq - 1; *l1 = s - a;
}
void orc_down4(int H, int W, const double *full, double *q) {
    const int h = H / 4, w = W / 4;
    for (int cy = 0; cy < h; cy++)
        for (int cx = 0; cx < w; cx++) {
            const double *p = full + (size_t)(4 * cy + 1) * W + 4 * cx + 1;
            q[cy * w + cx] = 0.5 * (0.5 * p[0] + 0.5 * p[1]) + 0.5 * (0.5 * p[W] + 0.5 * p[W + 1]);
        }
}
void orc_up4(int H, int W, const double *q, double *full) {
    const int h = H / 4, w = W / 4;
    for (int y = 0; y < H; y++) {
        int y0, y1; double ly;
        up4_taps(y, h, &y0, &y1, &ly);
        for (int x = 0; x < W; x++) {
            int x0, x1; double lx;
            up4_taps(x, w, &x0, &x1, &lx);
            full[y * W + x] = (1 - ly) * ((1 - lx) * q[y0 * w + x0] + lx * q[y0 * w + x1]) + ly * ((1 - lx) * q[y1 * w + x0] + lx * q[y1 * w + x1]);
        }
    }
}
/* adjoint of orc_up4: q[c] += sum_p U_pc full[p] (q is NOT cleared); stride: `full` holds `stride` values per pixel, `q` per cell */
static void up4_adjoint(int H, int W, int stride, const double *full, const unsigned char *skip, double *q) {
    const int h = H / 4, w = W / 4;
    for (int y = 0; y < H; y++) {
        int y0, y1; double ly;
        up4_taps(y, h, &y0, &y1, &ly);
        for (int x = 0; x < W; x++) {
            if (skip && skip[y * W + x]) continue;
            int x0, x1; double lx;
            up4_taps(x, w, &x0, &x1, &lx);
            const double wts[4] = {(1 - ly) * (1 - lx), (1 - ly) * lx, ly * (1 - lx), ly * lx};
            const int cs[4] = {y0 * w + x0, y0 * w + x1, y1 * w + x0, y1 * w + x1};
            for (int t = 0; t < 4; t++)
                for (int k = 0; k < stride; k++) q[(size_t)cs[t] * stride + k] += wts[t] * full[(size_t)(y * W + x) * stride + k];
        }
    }
}
void orc_up4_adjoint(int H, int W, const double *full, double *q) {
    memset(q, 0, sizeof(double) * (size_t)(H / 4) * (W / 4));
    up4_adjoint(H, W, 1, full, NULL, q);
}

/* Gauss-Newton on the reference loss in the reference's parametrisation: unknowns = the poses of all 2 S B pairs and the QUARTER-
 * resolution inverse depth of every target.  Per linearisation the full-resolution quantities of linearize_dense_ref (gradient g, the
 * diagonal curvature model D, the pose coupling B of every pixel) are carried to the cells by the chain rule through U:
 *     g_c = sum_p U_pc g_p  (exact),   B_c = sum_p U_pc B_p  (exact),   D_c = sum_p U_pc D_p
 * D_c is the row-sum lumping of U' diag(D) U: the rows of U are convex weights, so (sum_c U_pc x_c)^2 <= sum_c U_pc x_c^2 and
 * diag(D_c) majorises U' diag(D) U -- the cell block stays diagonal and the elimination of the depth stays per cell.  Pixels the
 * full-resolution mode freezes (sampled across the zero padding, or no curvature) contribute nothing.  The step of a cell goes through
 * the same trust region as a pixel's (depth_step).  depth_io [B][n] in: the full-resolution input map (its quarter-resolution
 * projection is the start, as optimizer.py:194-196); out: U rho_q as depth.  depth0: centre of the SSIM prior (the full-resolution
 * input, optimizer.py:89-90 `self.target_disparity`).  rho_q_out [B][n/16] or NULL. */
static void refine_dense_ref_q_impl(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, real *depth_s /* in / out when free_sources */,
                                    const real *K, const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                                    double *pose_io, double *stats, const unsigned short *bits, double *rho_q_out, int free_sources) {
    const int n = H * W, SB = S * B, NP = 6 * S, h = H / 4, w = W / 4, nq = h * w;
    const double lo = 1.0 / max_depth, hi = 1.0 / min_depth;
    real *ae = (real *)malloc(sizeof(real) * (size_t)2 * SB * n), *d0 = (real *)malloc(sizeof(real) * (size_t)B * n);
    const real **aep = (const real **)malloc(sizeof(real *) * 2 * SB);
    double *T = (double *)malloc(sizeof(double) * 24 * SB);
    double *g_xi = (double *)malloc(sizeof(double) * 12 * SB), *g_rho = (double *)malloc(sizeof(double) * (size_t)B * n);
    double *Hj = (double *)malloc(sizeof(double) * (size_t)B * NP * NP), *gj = (double *)malloc(sizeof(double) * (size_t)B * NP);
    double *Dq = (double *)malloc(sizeof(double) * (size_t)B * n), *Bq = (double *)malloc(sizeof(double) * (size_t)B * n * NP);
    double *Hi = (double *)malloc(sizeof(double) * 36 * SB), *gi = (double *)malloc(sizeof(double) * 6 * SB);
    double *rq = (double *)malloc(sizeof(double) * (size_t)B * nq), *full = (double *)malloc(sizeof(double) * n);
    double *pix = (double *)malloc(sizeof(double) * (size_t)n * (NP + 2)), *cell = (double *)malloc(sizeof(double) * (size_t)nq * (NP + 2));
    unsigned char *skip = (unsigned char *)malloc(n);
    /* free_sources: the quarter-resolution maps of the SOURCES are unknowns too (optimizer.py:194-198: one tensor of S + 1 channels) -- every
     * inverse pair a group of its pose and the cells of the source map it back-projects; same projection at the start, same cell records */
    dref_src_sys ss = {NULL, NULL, NULL, NULL};
    double *g_rho_s = NULL, *rqs = NULL;
    if (free_sources) {
        g_rho_s = (double *)malloc(sizeof(double) * (size_t)SB * n); rqs = (double *)malloc(sizeof(double) * (size_t)SB * nq);
        ss.Dq = (double *)malloc(sizeof(double) * (size_t)SB * n); ss.Bq = (double *)malloc(sizeof(double) * (size_t)SB * n * 6);
        ss.Hs = (double *)malloc(sizeof(double) * 36 * SB); ss.gs = (double *)malloc(sizeof(double) * 6 * SB);
        for (int m = 0; m < SB; m++) {
            for (int i = 0; i < n; i++) full[i] = 1.0 / (double)depth_s[(size_t)m * n + i];
            orc_down4(H, W, full, rqs + (size_t)m * nq);
            orc_up4(H, W, rqs + (size_t)m * nq, full);
            for (int i = 0; i < n; i++) depth_s[(size_t)m * n + i] = (real)(1.0 / full[i]);
        }
    }
    dref_auto_err(H, W, B, S, tgt, srcs, op, ae, aep);
    memcpy(d0, depth_io, sizeof(real) * (size_t)B * n);
    for (int b = 0; b < B; b++) {       /* the start: quarter-resolution projection of the input, upsampled again */
        for (int i = 0; i < n; i++) full[i] = 1.0 / (double)depth_io[(size_t)b * n + i];
        orc_down4(H, W, full, rq + (size_t)b * nq);
        orc_up4(H, W, rq + (size_t)b * nq, full);
        for (int i = 0; i < n; i++) depth_io[(size_t)b * n + i] = (real)(1.0 / full[i]);
    }
    for (int m = 0; m < 2 * SB; m++) orc_pose_to_T(pose_io + 6 * m, T + 12 * m);
    for (int it = 0; it < op->n_iters; it++) {
        dref_scal sc;
        g_lin_idx = bits ? it : -1;
        /* lambda_depth = infinity: no per-pixel elimination inside -- Hj / gj come back as the pose blocks and pose gradients themselves */
        linearize_dense_ref(H, W, B, S, tgt, srcs, depth_io, depth_s, d0, K, op, argmin, w_init, min_depth, max_depth, INFINITY, T, aep,
                            bits ? bits + (size_t)it * 2 * SB * n : NULL, &sc, g_xi, g_rho, Hj, gj, Dq, Bq, Hi, gi, g_rho_s, free_sources ? &ss : NULL);
        g_lin_idx = -1;
        if (op->w_pose_consist > 0) sc.loss += dref_pose_consist(B, S, op, T, Hj, gj, free_sources ? ss.Hs : Hi, free_sources ? ss.gs : gi);
        if (stats) { double *r = stats + 7 * it; r[0] = sc.loss; r[1] = sc.L_fwd; r[2] = sc.L_inv; r[3] = sc.L_dc; r[4] = sc.L_init; r[5] = sc.Kf; r[6] = sc.Ki; }
        for (int b = 0; b < B; b++) {
            for (int i = 0; i < n; i++) {
                const double D = Dq[(size_t)b * n + i];
                skip[i] = !((1.0 + lambda_depth) * D > 1e-30);
                pix[(size_t)i * (NP + 2)] = g_rho[(size_t)b * n + i]; pix[(size_t)i * (NP + 2) + 1] = D;
                for (int j = 0; j < NP; j++) pix[(size_t)i * (NP + 2) + 2 + j] = Bq[((size_t)b * n + i) * NP + j];
            }
            memset(cell, 0, sizeof(double) * (size_t)nq * (NP + 2));
            up4_adjoint(H, W, NP + 2, pix, skip, cell);
            double A[36 * JMAXS * JMAXS], dl[6 * JMAXS], gs[6 * JMAXS];
            for (int i = 0; i < NP; i++) {
                gs[i] = gj[(size_t)b * NP + i];
                for (int j = 0; j < NP; j++) A[i * NP + j] = Hj[(size_t)b * NP * NP + i * NP + j];
            }
            for (int c = 0; c < nq; c++) {
                const double *q = cell + (size_t)c * (NP + 2), Dd = (1.0 + lambda_depth) * q[1];
                if (!(Dd > 1e-30)) continue;
                for (int j = 0; j < NP; j++) {
                    gs[j] -= q[2 + j] * q[0] / Dd;
                    for (int k = 0; k < NP; k++) A[j * NP + k] -= q[2 + j] * q[2 + k] / Dd;
                }
            }
            for (int i = 0; i < NP; i++) { A[i * NP + i] += op->lambda0 * A[i * NP + i] + 1e-12; dl[i] = -gs[i]; }
            if (chol_solve(NP, A, dl)) memset(dl, 0, sizeof(dl));
            for (int s = 0; s < S; s++) {
                double E[12], Tn[12];
                orc_se3_exp(dl + 6 * s, E);
                orc_se3_mul(E, T + 12 * (s * B + b), Tn);
                memcpy(T + 12 * (s * B + b), Tn, sizeof(Tn));
            }
            for (int c = 0; c < nq; c++) {
                const double *q = cell + (size_t)c * (NP + 2), Dd = (1.0 + lambda_depth) * q[1];
                if (!(Dd > 1e-30)) continue;
                double bd = 0;
                for (int j = 0; j < NP; j++) bd += q[2 + j] * dl[j];
                rq[(size_t)b * nq + c] = depth_step(rq[(size_t)b * nq + c], -(q[0] + bd) / Dd, lo, hi);
            }
            orc_up4(H, W, rq + (size_t)b * nq, full);
            for (int i = 0; i < n; i++) depth_io[(size_t)b * n + i] = (real)(1.0 / full[i]);
        }
        for (int m = 0; m < SB; m++) {      /* inverse pairs: their own 6 x 6 systems */
            if (free_sources) {             /* ... reduced by the CELLS of the source map (ss.Hs / ss.gs come back unreduced: lambda_depth = infinity above) */
                for (int i = 0; i < n; i++) {
                    const double D = ss.Dq[(size_t)m * n + i];
                    skip[i] = !((1.0 + lambda_depth) * D > 1e-30);
                    pix[(size_t)i * 8] = g_rho_s[(size_t)m * n + i]; pix[(size_t)i * 8 + 1] = D;
                    for (int j = 0; j < 6; j++) pix[(size_t)i * 8 + 2 + j] = ss.Bq[((size_t)m * n + i) * 6 + j];
                }
                memset(cell, 0, sizeof(double) * (size_t)nq * 8);
                up4_adjoint(H, W, 8, pix, skip, cell);
                double A[36], dl[6], gsv[6];
                for (int i = 0; i < 6; i++) {
                    gsv[i] = ss.gs[6 * m + i];
                    for (int j = 0; j < 6; j++) A[i * 6 + j] = ss.Hs[36 * m + i * 6 + j];
                }
                for (int c = 0; c < nq; c++) {
                    const double *q = cell + (size_t)c * 8, Dd = (1.0 + lambda_depth) * q[1];
                    if (!(Dd > 1e-30)) continue;
                    for (int j = 0; j < 6; j++) {
                        gsv[j] -= q[2 + j] * q[0] / Dd;
                        for (int k = 0; k < 6; k++) A[j * 6 + k] -= q[2 + j] * q[2 + k] / Dd;
                    }
                }
                for (int i = 0; i < 6; i++) { A[i * 6 + i] += op->lambda0 * A[i * 6 + i] + 1e-12; dl[i] = -gsv[i]; }
                if (chol_solve(6, A, dl)) memset(dl, 0, sizeof(dl));
                double E[12], Tn[12];
                orc_se3_exp(dl, E);
                orc_se3_mul(E, T + 12 * (SB + m), Tn);
                memcpy(T + 12 * (SB + m), Tn, sizeof(Tn));
                for (int c = 0; c < nq; c++) {
                    const double *q = cell + (size_t)c * 8, Dd = (1.0 + lambda_depth) * q[1];
                    if (!(Dd > 1e-30)) continue;
                    double bd = 0;
                    for (int j = 0; j < 6; j++) bd += q[2 + j] * dl[j];
                    rqs[(size_t)m * nq + c] = depth_step(rqs[(size_t)m * nq + c], -(q[0] + bd) / Dd, lo, hi);
                }
                orc_up4(H, W, rqs + (size_t)m * nq, full);
                for (int i = 0; i < n; i++) depth_s[(size_t)m * n + i] = (real)(1.0 / full[i]);
                continue;
            }
            orc_opts o6 = *op;
            o6.nparam = 6; o6.param = 0;
            double Tn[12], sdummy;
            apply_step(&o6, Hi + 36 * m, gi + 6 * m, op->lambda0, T + 12 * (SB + m), 0.0, Tn, &sdummy);
            memcpy(T + 12 * (SB + m), Tn, sizeof(Tn));
        }
    }
    for (int m = 0; m < 2 * SB; m++) orc_T_to_pose(T + 12 * m, pose_io + 6 * m);
    if (rho_q_out) memcpy(rho_q_out, rq, sizeof(double) * (size_t)B * nq);
    free(ae); free(d0); free(aep); free(T); free(g_xi); free(g_rho); free(Hj); free(gj); free(Dq); free(Bq); free(Hi); free(gi);
    free(rq); free(full); free(pix); free(cell); free(skip);
    free(g_rho_s); free(rqs); free(ss.Dq); free(ss.Bq); free(ss.Hs); free(ss.gs);
}
void orc_refine_dense_ref_q(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, const real *depth_s, const real *K,
                            const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                            double *pose_io, double *stats, const unsigned short *bits, double *rho_q_out) {
    refine_dense_ref_q_impl(H, W, B, S, tgt, srcs, depth_io, (real *)depth_s, K, op, argmin, w_init, lambda_depth, min_depth, max_depth, pose_io, stats, bits, rho_q_out, 0);
}
/* the reference's complete leaf set: the quarter-resolution maps of the target AND of the sources (depth_s_io [SB][n] in / out) */
void orc_refine_dense_ref_q_free(int H, int W, int B, int S, const real *tgt, const real *srcs, real *depth_io, real *depth_s_io, const real *K,
                                 const orc_opts *op, int argmin, double w_init, double lambda_depth, double min_depth, double max_depth,
                                 double *pose_io, double *stats, const unsigned short *bits, double *rho_q_out) {
    refine_dense_ref_q_impl(H, W, B, S, tgt, srcs, depth_io, depth_s_io, K, op, argmin, w_init, lambda_depth, min_depth, max_depth, pose_io, stats, bits, rho_q_out, 1);
}

/* ------------------------------------------------------------------------- */
/* DNet ground-plane scale recovery, models/dnet_layers.py:249-327                            */

static void v3_norm(real *v) { /* F.normalize: v / max(|v|, 1e-12) */
    real n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n < (real)1e-12) n = (real)1e-12;
    v[0] /= n; v[1] /= n; v[2] /= n;
}
static void v3_cross_norm(const real *a, const real *b, const real *c, real *o) { /* normalize((a-c) x (b-c)) */
    real u[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]}, w[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]};
    o[0] = u[1] * w[2] - u[2] * w[1]; o[1] = u[2] * w[0] - u[0] * w[2]; o[2] = u[0] * w[1] - u[1] * w[0];
    v3_norm(o);
}

/* per image: camera height map |P.n| and ground mask (dnet_layers.py:259-304,319-322); depth [H,W], K 3x3 */
void orc_ground_height(int H, int W, const real *depth, const real *K, real *height, real *mask) {
    double Kd[9], Ki[9];
    for (int i = 0; i < 9; i++) Kd[i] = K[i];
    mat3_inv(Kd, Ki);
    int n = H * W;
    real *P = (real *)malloc(sizeof(real) * 3 * n), *N = (real *)calloc((size_t)3 * n, sizeof(real));
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++)
            for (int k = 0; k < 3; k++)   /* BackprojectDepth, dnet_layers.py:159-163 */
                P[3 * (v * W + u) + k] = depth[v * W + u] * ((real)Ki[3 * k] * u + (real)Ki[3 * k + 1] * v + (real)Ki[3 * k + 2]);
#define PT(vv, uu) (P + 3 * ((vv) * W + (uu)))
    for (int v = 1; v < H - 1; v++)
        for (int u = 1; u < W - 1; u++) { /* get_surface_normal, dnet_layers.py:259-287 */
            const real *c = PT(v, u);
            real n0[3], n1[3], n2[3], n3[3], m[3];
            v3_cross_norm(PT(v, u - 1), PT(v - 1, u), c, n0);
            v3_cross_norm(PT(v, u + 1), PT(v + 1, u), c, n1);
            v3_cross_norm(PT(v - 1, u - 1), PT(v + 1, u - 1), c, n2);
            v3_cross_norm(PT(v - 1, u + 1), PT(v + 1, u + 1), c, n3);
            for (int k = 0; k < 3; k++) m[k] = (n0[k] + n1[k] + n2[k] + n3[k]) / 4;
            v3_norm(m);
            for (int k = 0; k < 3; k++) N[3 * (v * W + u) + k] = m[k];
        }
    const real thr = (real)cos(5.0 * 3.14159265358979323846 / 180.0);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            /* ReflectionPad2d(1) of the (H-2)x(W-2) normal map, dnet_layers.py:289-290 */
            int rv = v == 0 ? 2 : (v == H - 1 ? H - 3 : v), ru = u == 0 ? 2 : (u == W - 1 ? W - 3 : u);
            const real *nn = N + 3 * (rv * W + ru), *p = PT(v, u);
            real nrm = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            real cs = nn[1] / (nrm > (real)1e-6 ? nrm : (real)1e-6);      /* CosineSimilarity(eps=1e-6) with (0,1,0) */
            int g = ((cs > thr) || (cs < -thr)) && (p[1] > 0);
            mask[v * W + u] = (real)g;
            height[v * W + u] = fabs(p[0] * nn[0] + p[1] * nn[1] + p[2] * nn[2]);
        }
#undef PT
    free(P); free(N);
}

static int cmp_real(const void *a, const void *b) { real x = *(const real *)a, y = *(const real *)b; return (x > y) - (x < y); }

/* ScaleRecovery.forward (dnet_layers.py:306-327): lower median of the masked heights of the WHOLE batch */
double orc_scale_recovery(int nimg, int H, int W, const real *depth, const real *K, double real_cam_height, double *median_out) {
    int n = H * W;
    real *h = (real *)malloc(sizeof(real) * n), *m = (real *)malloc(sizeof(real) * n), *sel = (real *)malloc(sizeof(real) * n * nimg);
    size_t cnt = 0;
    for (int b = 0; b < nimg; b++) {
        orc_ground_height(H, W, depth + (size_t)b * n, K + 9 * b, h, m);
        for (int i = 0; i < n; i++) if (m[i] > 0) sel[cnt++] = h[i];
    }
    double med = NAN;
    if (cnt) { qsort(sel, cnt, sizeof(real), cmp_real); med = sel[(cnt - 1) / 2]; }   /* torch.median: lower median */
    if (median_out) *median_out = med;
    free(h); free(m); free(sel);
    return real_cam_height / med;
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
