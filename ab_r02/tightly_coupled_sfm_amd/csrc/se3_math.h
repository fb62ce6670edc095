// se3_math.h -- double-precision SE(3) / small dense solver shared by the device solve kernel and the
// host-side C ABI utilities (tcsfm_se3_*).  All matrices are row-major; rigid transforms are 3x4.
//
// Conventions follow the reference's hot path:
//   pose 6-vector [tx,ty,tz,rx,ry,rz]; warp transform = pose_vec2mat(-pose) = [Rx(-rx)Ry(-ry)Rz(-rz) | -t]
//   (models/stn.py:81-116,143-158; call sites train_mono.py:69, helpers.py:11)
//   twists are [rho, phi], translation first, as liegroups.SE3 (validate.py:65).
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define TC_HD __host__ __device__ inline
#else
#define TC_HD inline
#endif

#define TC_MAXP 7

namespace tc {

TC_HD void mat3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

TC_HD void hat(const double *w, double *K) {
    K[0] = 0; K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2]; K[4] = 0; K[5] = -w[0];
    K[6] = -w[1]; K[7] = w[0]; K[8] = 0;
}

// R = Rx(ax) Ry(ay) Rz(az)
TC_HD void euler_to_R(const double *ang, double *R) {
    double ca = cos(ang[0]), sa = sin(ang[0]), cb = cos(ang[1]), sb = sin(ang[1]), cc = cos(ang[2]), sc = sin(ang[2]);
    R[0] = cb * cc;                R[1] = -cb * sc;               R[2] = sb;
    R[3] = ca * sc + sa * sb * cc; R[4] = ca * cc - sa * sb * sc; R[5] = -sa * cb;
    R[6] = sa * sc - ca * sb * cc; R[7] = sa * cc + ca * sb * sc; R[8] = ca * cb;
}

TC_HD void pose_to_T(const double *pose, double *T) {
    double ang[3] = {-pose[3], -pose[4], -pose[5]}, R[9];
    euler_to_R(ang, R);
    for (int i = 0; i < 3; i++) {
        T[4 * i] = R[3 * i]; T[4 * i + 1] = R[3 * i + 1]; T[4 * i + 2] = R[3 * i + 2];
        T[4 * i + 3] = -pose[i];
    }
}

TC_HD void T_to_pose(const double *T, double *pose) {
    double sb = T[2] > 1 ? 1 : (T[2] < -1 ? -1 : T[2]);
    pose[0] = -T[3]; pose[1] = -T[7]; pose[2] = -T[11];
    pose[3] = -atan2(-T[6], T[10]);
    pose[4] = -asin(sb);
    pose[5] = -atan2(-T[1], T[0]);
}

// exp: T = [exp(phi^) | J_l(phi) rho]
TC_HD void se3_exp(const double *xi, double *T) {
    const double *rho = xi, *phi = xi + 3;
    double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2], A, B, C;
    if (t2 < 0.09) {
        // |phi| < 0.3 rad (every Gauss-Newton step): Taylor series through t^10, truncation < 1e-16.  No libm trig on the
        // device's serial critical path.
        A = 1 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 - t2 * (1.0 / 39916800)))));
        B = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 - t2 * (1.0 / 479001600)))));
        C = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 - t2 * (1.0 / 6227020800.0)))));
    } else {
        double t = sqrt(t2);
        A = sin(t) / t; B = (1 - cos(t)) / t2; C = (t - sin(t)) / (t2 * t);
    }
    double K[9], K2[9];
    hat(phi, K);
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

TC_HD void se3_log(const double *T, double *xi) {
    double c = 0.5 * (T[0] + T[5] + T[10] - 1);
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    double t = acos(c), t2 = t * t;
    double f = (t < 1e-6) ? 0.5 + t2 / 12 : t / (2 * sin(t));
    double phi[3] = {f * (T[9] - T[6]), f * (T[2] - T[8]), f * (T[4] - T[1])};
    double D = (t < 1e-4) ? 1.0 / 12 + t2 / 720 : 1.0 / t2 - (1 + cos(t)) / (2 * t * sin(t));
    double K[9], K2[9];
    hat(phi, K);
    mat3_mul(K, K, K2);
    for (int i = 0; i < 3; i++) {
        double v = 0;
        for (int j = 0; j < 3; j++) v += (((i == j) ? 1.0 : 0.0) - 0.5 * K[3 * i + j] + D * K2[3 * i + j]) * T[4 * j + 3];
        xi[i] = v;
        xi[3 + i] = phi[i];
    }
}

TC_HD void se3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
        C[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
    }
}

TC_HD void se3_inv(const double *A, double *B) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) B[4 * i + j] = A[4 * j + i];
        B[4 * i + 3] = -(A[i] * A[3] + A[4 + i] * A[7] + A[8 + i] * A[11]);
    }
}

// A (6x6) = d(xi_left)/d(pose):  T(pose + dp) ~= exp((A dp)^) T(pose)
//   dphi = -J_e dr,  drho = -dt - [t']x J_e dr,  t' = -t,  J_e = [e_x | Rx e_y | Rx Ry e_z] at angles -r
TC_HD void euler_left_jacobian(const double *pose, double *A) {
    double ax = -pose[3], ay = -pose[4];
    double ca = cos(ax), sa = sin(ax), cb = cos(ay), sb = sin(ay);
    double Je[9] = {1, 0, sb, 0, ca, -sa * cb, 0, sa, ca * cb};
    double tp[3] = {-pose[0], -pose[1], -pose[2]}, Tx[9], TJ[9];
    hat(tp, Tx);
    mat3_mul(Tx, Je, TJ);
    for (int i = 0; i < 36; i++) A[i] = 0;
    for (int i = 0; i < 3; i++) {
        A[6 * i + i] = -1;
        for (int j = 0; j < 3; j++) {
            A[6 * i + 3 + j] = -TJ[3 * i + j];
            A[6 * (3 + i) + 3 + j] = -Je[3 * i + j];
        }
    }
}

// Cholesky solve of the N x N SPD system (row-major, overwritten); returns false when not SPD.
// N is a template parameter and every loop is unrolled so that, on the device, all arrays live in registers
// (runtime-indexed local arrays go to scratch memory: the first version of k_solve spent ~50 us there).
template <int N>
TC_HD bool chol_solve(double *A, double *b) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * N + k] * A[j * N + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * N + j] = d;
        const double id = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = s * id;
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= A[i * N + k] * b[k];
        b[i] = s / A[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= A[k * N + i] * b[k];
        b[i] = s / A[i * N + i];
    }
    return true;
}

// (H + lambda diag(H) + 1e-12 I) d = -g
template <int N>
TC_HD void damped_step(const double *H, const double *g, double lambda, double *delta) {
    double A[N * N], b[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < N; j++) A[i * N + j] = H[i * N + j];
        A[i * N + i] += lambda * H[i * N + i] + 1e-12;
        b[i] = -g[i];
    }
    const bool ok = chol_solve<N>(A, b);
#pragma unroll
    for (int i = 0; i < N; i++) delta[i] = ok ? b[i] : 0.0;
}

// One update of (T, log_scale) from the normal equations under the chosen parameterisation
// param 0: T <- exp(delta) T ; param 1: additive on the reference [t, euler] vector (H_p = A'HA, g_p = A'g)
// T <- exp(delta) T for an already solved step (the device solves the system lane-parallel)
TC_HD void retract_se3(const double *delta, const double *Tin, double *Tout) {
    double E[12];
    se3_exp(delta, E);
    se3_mul(E, Tin, Tout);
}

// ws: caller-provided workspace of 3 N N doubles for the additive-Euler branch (LDS on the device, so that the
// common SE(3) branch keeps everything in registers)
template <int N>
TC_HD void apply_step(int param, const double *H, const double *g, double lambda, const double *Tin, double sin_,
                      double *Tout, double *sout, double *ws) {
    double delta[N];
    if (param == 0) {
        damped_step<N>(H, g, lambda, delta);
        double E[12];
        se3_exp(delta, E);
        se3_mul(E, Tin, Tout);
    } else {
        double pose[6], A6[36], gp[N];
        double *Af = ws, *Hp = ws + N * N, *HA = ws + 2 * N * N;
        T_to_pose(Tin, pose);
        euler_left_jacobian(pose, A6);
#pragma unroll
        for (int i = 0; i < N * N; i++) Af[i] = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) Af[i * N + j] = A6[6 * i + j];
        if (N == 7) Af[6 * N + 6] = 1;
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < N; j++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < N; k++) s += H[i * N + k] * Af[k * N + j];
                HA[i * N + j] = s;
            }
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < N; k++) s += Af[k * N + i] * g[k];
            gp[i] = s;
#pragma unroll
            for (int j = 0; j < N; j++) {
                double h = 0;
#pragma unroll
                for (int k = 0; k < N; k++) h += Af[k * N + i] * HA[k * N + j];
                Hp[i * N + j] = h;
            }
        }
        damped_step<N>(Hp, gp, lambda, delta);
#pragma unroll
        for (int i = 0; i < 6; i++) pose[i] += delta[i];
        pose_to_T(pose, Tout);
    }
    *sout = sin_ + (N == 7 ? delta[N - 1] : 0.0);
}

}  // namespace tc
