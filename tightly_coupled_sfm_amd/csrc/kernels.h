// kernels.h -- gfx950 device code of the photometric refinement engine.
//
// Kernels (one HIP stream, no host sync between them):
//   k_pack        once per refine call: planar fp32 inputs -> two float4 images per pair
//                   tgtpack = (tgt r,g,b, auto_err)   auto_err = iteration-invariant auto-mask error (train_mono.py:84)
//                   srcpack = (src r,g,b, depth_s)    one 16-B gather per bilinear tap instead of four 4-B gathers;
//                                                     stored with a 1-texel zero border so that taps need no masks
//                 (window form: the fwd / inv directed pairs of train_mono.py:54-62 are formed here by indexing)
//   k_linearize   THE hot kernel, once per Gauss-Newton iteration: fused
//                   backproject (stn.py:33-48) -> rigid transform + project (stn.py:198-231) -> bilinear warp of
//                   RGB+depth with d/d(ix,iy) (stn.py:266,271) -> L1 + 3x3 SSIM (losses.py:27-41, train_mono.py:87)
//                   -> masks / depth-consistency weight (train_mono.py:89-92) -> exact gradient rows, structure-tensor
//                   curvature -> per-workgroup J'J / J'r partial sums.  Reads 36 B/pixel, writes one partial-sum record
//                   per workgroup; measured VALU-issue bound (~1050 instructions per 64-pixel wave), see DESIGN.md.
//                   MODE_COST / MODE_MAPS variants: scalar cost only / the reference's residual maps.
//   k_solve       once per iteration, one workgroup per pair: deterministic fp64 reduction of the partial sums,
//                   LM/GN logic, lane-parallel Gauss-Jordan on the 6x6 / 7x7 system, SE(3) retraction, emits the fp32
//                   constants of the next iteration.
//   k_warp        inverse_warp2 drop-in (stn.py:234-273), planar in / planar out (+ the next PoseNet input, train_mono.py:73-77).
//   k_ssim, k_disp_to_depth: SSIM_Loss / disp_to_depth drop-ins.  dense_kernel.h, scale_kernel.h: dense mode, DNet scale.
#pragma once
#include <hip/hip_runtime.h>
#include "se3_math.h"
#include "wave_reduce.h"

namespace tc {

// ---------------------------------------------------------------------------------------------------------------
// per-pair constants of one linearisation (fp32, written by k_init / k_solve, read through the scalar cache)
struct PairConst {
    float A[9], kt[3];         // A = K (R - I) K^-1 and K t, both formed in fp64: K[R|t] K^-1 pix = pix + A pix, so the
                               // projection OFFSET (flow) is computed without cancellation (see warp_geo)
    float R[9], t[3];          // [R|t] = pose_vec2mat(-pose) (left-multiplied by the GN updates)
    float fx, fy, cx, cy;      // pinhole intrinsics
    float ki0, ki2, ki4, ki5;  // K^-1 = [ki0 0 ki2; 0 ki4 ki5; 0 0 1]
    float es;                  // exp(log depth-scale)
    int img;                   // which packed image pair this problem reads (loss-surface sweeps share one)
    int pad[2];
};

// per-pair optimiser state (fp64)
struct PairState {
    double Tcur[12], Ttry[12];
    double scur, stry, s0;   // log depth-scale: accepted, trial, initial (prior centre)
    double lambda, cost_cur;
    double M8[64];             // accepted linearisation as the augmented 8x8 system [H | -g] (undamped), one entry per lane
    double K[9];
    int have_cur, pad;
};

struct LinParams {
    const float4 *tgtpack;  // [Nimg][H][W]  rgb + auto_err
    // Window forms of the pose modes (round 5, `tshare`): every image of a window is the TARGET of one directed pair and the SOURCE of its
    // partner (pair n <-> n +- tshare_sb), so k_pack writes each image ONCE -- the bordered (rgb, depth) pack the partner gathers from -- plus
    // ONE auto-mask error plane per couple (in depth_t's buffer, at the forward pair's index; the error of (x, y) and of (y, x) is the same
    // number).  The pair then reads its target colours and depth from srcpack[partner] and the error from that plane: the same two loads per
    // pixel as tgtpack + depth_t, 36 instead of 72 bytes written per pixel and window, and the target rows share cache lines with the partner's taps.
    int tshare, tshare_sb;
    const float4 *srcpack;  // [Nimg][H+2][W+2]  rgb + depth_s, 1-texel zero border
    const float *depth_t;   // [Nimg][H][W]
    const PairConst *pc;    // [N]
    float *blockrec;        // [N][nblk][nacc]   one partial-sum record per workgroup (write-through stores)
    int *tickets;           // [N][ngrp]         arrival counters of the 16-workgroup reduction groups (zero between launches)
    float *partials;        // [N][ngrp][nacc] group records (a few dozen per pair): what the solve kernel reads
    // maps mode outputs (may be null)
    float *o_diff, *o_valid, *o_weight, *o_auto_err, *o_auto_mask, *o_rec;
    int H, W, tiles_x, tiles_y, nacc;
    int ngrp, ngrp_pad;     // reduction groups per pair, and that rounded up to a multiple of 64
    float wl, ws;           // w_l1/3, w_ssim/3
    float eps;              // irls_eps
    int automask;
    int shared_image;       // all problems read packed image pair 0 (loss-surface sweeps); kept OUT of PairConst so that the
                            // first image loads do not wait for the scalar loads of the pair constants
    int direct;             // 1: no in-launch group reduction -- k_solve sums the workgroup records itself (small grids)
    int one_generation;     // 1: the whole grid is resident at once (<= 2 workgroups per CU): its workgroups run their phases in lockstep and the
                            // two-position waves of phase 1 are the critical path to the barrier -- they raise their priority (a full chip
                            // prefers them unprioritised: measured both ways, profiles/r04_lds_pipeline_ab.txt).  Scheduling only: same bits.
    // dense window modes: residual maps [ext_S * ext_B][H][W] of the forward pairs (k_linearize<MODE_MAPS> at the current poses); the
    // min-over-sources selection formed from them (ext_selected) replaces the own mask of pairs n < n_ext
    const float *ext_diff, *ext_valid;
    int n_ext, ext_B, ext_S;
    int sel_B, sel_S;       // k_linearize<SEL>: window geometry; pairs n < sel_B * sel_S are the forward pairs n = s * sel_B + b
    // Frame-level pack cache (sequence calls, tcsfm_refine_sequence): the packed source image (rgb + depth, zero border) and the
    // converted depth plane exist ONCE per frame of the ring (k_frame_pack, when the frame's copy lands) instead of once per
    // directed pair and call; pair n reads the slots k_pack_cached noted for it.  Null: per-pair srcpack / depth_t as before.
    const int *pair_src, *pair_dep;     // [N] ring slot of pair n's source pack / of its target's depth plane
    int rule;               // TCSFM_WINDOW_REFERENCE: the forward pairs of sources s > 0 weigh their pixels with source 0's depth-
                            // consistency weight map, source 0 carries the cross term (optimizer.py:69; normalisers: k_solve)
    int fwd_noauto;         // pairs n < fwd_noauto take no auto-mask (REFERENCE rule without argmin, optimizer.py:71-73)
    unsigned short *trace;  // tcsfm_debug_trace: [N][H*W] decisions of THIS launch (bit 0 = pixel counts, bit 1 = warp valid,
                            // bits 2 / 3 = parity of the bilinear cell floor(ix) / floor(iy), bits 4-5 = sign code of cd - pd,
                            // bits 6-11 = sign codes of rec_c - tgt_c; codes 0 zero / 1 positive / 2 negative), or null
    unsigned long long *stamp;  // tcsfm_profile_*: [workgroups of this launch][2] = start / end of every workgroup in
                                // s_memrealtime ticks (100 MHz), or null.  Duration of the launch as the GPU sees it = latest end -
                                // earliest start (taken on the host), free of the ~2-4 us a HIP event pair adds around a 10 us kernel.
    // FRONT (k_linearize<.., FRONT = true>; dense mode on the reference's loss, dense_ref_kernel.h, round 5): ONE launch over all 2 S B
    // directed pairs of a call opens every linearisation.  Pairs n < front_fwd are the FORWARD pairs: their mask only -- own mask, or the
    // min-over-sources selection under SEL -> sel_out (what k_dense_joint reads instead of recomputing the selection from residual maps)
    // and its count -> norms[2 g]; pairs n >= front_fwd are the INVERSE pairs: the complete 6 x 6 linearisation, their mask count ->
    // norms[2 g + 1], and the adjoint of their bilinear samples of the target depth -> ext2 (two fixed-point sums per target pixel, so
    // that the scatter needs no batch normaliser and rides in this launch).  g = normaliser group of the pair's target (coalesced calls:
    // one group per call, norm_B targets each; 0: one group).
    int front_fwd, front_Bt;    // S B; targets of the launch: the target of pair n is (n % front_fwd) % front_Bt
    int norm_B;
    float *sel_out;             // [front_fwd][H*W] 1 / 0 (or null: nobody reads the selection)
    const float *sel_in;        // k_dense_joint: that map (or null: own masks / ext_selected)
    int *norms;                 // [groups][2] K_f, K_i: zero when the launch starts
    long long *ext2;            // [front_Bt][H*W][2]: sum of h(dd) ddd w_tap and of M diff ddd w_tap over the inverse pixels that sample the
                                // target pixel, 2^-40 fixed point (integer atomics: order-independent, bit-reproducible); zero when the launch starts
    // FRONT with free source maps (front_light = 1, opts.free_source_depths): the inverse pairs are linearised by the joint kernel as groups
    // of one source, so EVERY row is light here -- mask, count, scatter -- and the forward rows scatter too: the adjoint of the forward pairs'
    // samples of the SOURCE maps -> ext2_src (what k_dref_scatter_src computed).  The sign of cd - pd is then the plain fp32 difference's,
    // as the joint kernel that linearises both groups takes it.
    int front_light;
    long long *ext2_src;        // [front_fwd][H*W][2], zero when the launch starts
};
constexpr double DREF_FIX = 1099511627776.0;     // 2^40: fixed-point scale of the scatter sums

__device__ __forceinline__ unsigned sign_code(float x) { return x > 0.f ? 1u : (x < 0.f ? 2u : 0u); }

// in-kernel launch bracket (see LinParams::stamp): one plain 8-byte store per workgroup at each end, only while profiling
// (an atomic min / max on one word per launch was tried first: 480 workgroups x 2 same-address atomics x ~12 ns = +4 us per launch)
__device__ __forceinline__ void stamp_begin(unsigned long long *stamp, int tid) {
    if (stamp != nullptr && tid == 0) stamp[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x)] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void stamp_end(unsigned long long *stamp, int tid) {
    if (stamp != nullptr && tid == 0) stamp[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = (unsigned long long)wall_clock64();
}

constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ int refl_idx(int i, int n) {  // ReflectionPad2d(1), losses.py:22 (clamped for safety)
    i = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
    return min(max(i, 0), n - 1);
}
__device__ __forceinline__ float clamp01(float a) { return fminf(fmaxf(a, 0.f), 1.f); }
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: the kernel is VALU-bound
__device__ __forceinline__ float frcp(float a) { return __builtin_amdgcn_rcpf(a); }

// ---------------------------------------------------------------------------------------------------------------
// warp geometry of one target pixel
struct Geo {
    float rx, ry;        // sample position RELATIVE to the pixel's own integer coordinates: ix = u + rx, iy = v + ry
    float Z, iz;         // computed depth (clamped at 1e-3, stn.py:215) and 1/Z
    float uz, vz;        // projected pixel coordinates p0/Z, p1/Z
    float X0, X1, X2;    // point in the source camera frame
    float q2;            // Z - D before the clamp (D = scaled target depth): the small part of the computed depth (see dc_diff)
    bool oobx, ooby, zcl;
};

// pixel2cam (stn.py:33-48) -> [R|t] -> cam2pixel2 (stn.py:198-231) -> grid_sample un-normalisation (stn.py:266).
// Same mathematics as the reference, evaluated in a cancellation-free order for fp32:
//   p = K[R|t](D K^-1 pix) = D (pix + A pix) + K t     =>  p0 - u p2 = D((A pix)_0 - u (A pix)_2) + kt_0 - u kt_2
// so the flow u_proj - u = (p0 - u Z)/Z is formed from small quantities only, and the bilinear weights come from
// the fractional part of a small number instead of from ix ~ 10^2..10^3 (ulp 3e-5 px).  Agreement with the float64
// oracle improves ~300x; versus the reference's own fp32 evaluation only exact ties can differ.
__device__ __forceinline__ void warp_geo(const PairConst &c, int W, int H, int ui, int vi, float depth, Geo &g) {
    const float u = (float)ui, v = (float)vi;
    float D = c.es * depth;
    float a0 = c.A[0] * u + c.A[1] * v + c.A[2];
    float a1 = c.A[3] * u + c.A[4] * v + c.A[5];
    float a2 = c.A[6] * u + c.A[7] * v + c.A[8];
    float q0 = D * a0 + c.kt[0], q1 = D * a1 + c.kt[1], q2 = D * a2 + c.kt[2];
    float p2 = D + q2;
    g.q2 = q2;
    g.zcl = p2 < 1e-3f;
    g.Z = g.zcl ? 1e-3f : p2;
    g.iz = frcp(g.Z);
    // numerators of the flow: p0 - u Z, p1 - v Z
    float fu = g.zcl ? (u * D + q0) - u * g.Z : q0 - u * q2;
    float fv = g.zcl ? (v * D + q1) - v * g.Z : q1 - v * q2;
    float flx = fu * g.iz, fly = fv * g.iz;
    g.uz = u + flx;
    g.vz = v + fly;
    // |x_norm| > 1  <=>  u_proj outside [0, W-1]   (stn.py:223-227; detached sentinel -> zero sample, zero gradient)
    g.oobx = (flx > (float)(W - 1 - ui)) || (flx < -u);
    g.ooby = (fly > (float)(H - 1 - vi)) || (fly < -v);
    // ix = u_proj W/(W-1) - 0.5 = u + [u/(W-1) - 0.5 + flow W/(W-1)]
    const float iw = frcp((float)(W - 1)), ih = frcp((float)(H - 1));
    g.rx = (u * iw - 0.5f) + flx * ((float)W * iw);
    g.ry = (v * ih - 0.5f) + fly * ((float)H * ih);
    // point in the source camera frame (Jacobians only)
    float r0 = c.ki0 * u + c.ki2, r1 = c.ki4 * v + c.ki5;
    float x0 = r0 * D, x1 = r1 * D, x2 = D;
    g.X0 = c.R[0] * x0 + c.R[1] * x1 + c.R[2] * x2 + c.t[0];
    g.X1 = c.R[3] * x0 + c.R[4] * x1 + c.R[5] * x2 + c.t[1];
    g.X2 = c.R[6] * x0 + c.R[7] * x1 + c.R[8] * x2 + c.t[2];
}

// bilinear sample of a float4 image at (ui + rx, vi + ry) with zero padding; also d/dix and d/diy
// (grid_sampler_2d forward/backward semantics).  oob => the reference's sentinel: everything is zero.
// The image is stored with a 1-texel ZERO BORDER ((H+2) x (W+2), k_pack): taps are clamped into the border instead of being
// masked one by one, and an out-of-bounds sample is sent there as a whole -- 1 select instead of 16 selects + 8 compares.
struct Tap {
    float4 v00, v01, v10, v11;
    float wx, wy;
    bool inside;     // all four taps are real source pixels (none comes from the zero border)
};
// address computation + the four 16-byte gathers (asynchronous: nothing here waits for them)
__device__ __forceinline__ void tap4_fetch(const float4 *__restrict__ img, int W, int H, int ui, int vi, float rx, float ry, bool oob, Tap &t) {
    float fx = floorf(rx), fy = floorf(ry);
    t.wx = rx - fx; t.wy = ry - fy;
    int xi = ui + (int)fx, yi = vi + (int)fy;
    t.inside = !oob && xi >= 0 && xi < W - 1 && yi >= 0 && yi < H - 1;
    xi = oob ? -2 : xi;                                         // both columns clamp to the left border: all four taps are zero
    const int x0 = min(max(xi, -1), W) + 1, x1 = min(max(xi + 1, -1), W) + 1;   // bordered coordinates 0 .. W+1
    const int y0 = min(max(yi, -1), H) + 1, y1 = min(max(yi + 1, -1), H) + 1;
    // 32-bit byte offsets from the (wave-uniform) image base: the loads take the base in an SGPR pair and the offset in one VGPR instead
    // of a sign-extended 64-bit address per tap (a bordered image is (H + 2)(W + 2) 16 B < 4 GB)
    const unsigned WB = (unsigned)(W + 2);
    const char *base = reinterpret_cast<const char *>(img);
    const unsigned r0 = (unsigned)y0 * WB, r1 = (unsigned)y1 * WB;
    t.v00 = *reinterpret_cast<const float4 *>(base + ((r0 + (unsigned)x0) << 4)); t.v01 = *reinterpret_cast<const float4 *>(base + ((r0 + (unsigned)x1) << 4));
    t.v10 = *reinterpret_cast<const float4 *>(base + ((r1 + (unsigned)x0) << 4)); t.v11 = *reinterpret_cast<const float4 *>(base + ((r1 + (unsigned)x1) << 4));
}
// bilinear value and d/dix, d/diy of the four channels
__device__ __forceinline__ void tap4_lerp(const Tap &t, float4 &val, float4 &gx, float4 &gy) {
    const float wx = t.wx, wy = t.wy, ax = 1.f - wx, ay = 1.f - wy;
#define TC_LERP(f)                                                                             \
    val.f = ax * ay * t.v00.f + wx * ay * t.v01.f + ax * wy * t.v10.f + wx * wy * t.v11.f;     \
    gx.f = ay * (t.v01.f - t.v00.f) + wy * (t.v11.f - t.v10.f);                                \
    gy.f = ax * (t.v10.f - t.v00.f) + wx * (t.v11.f - t.v01.f);
    TC_LERP(x) TC_LERP(y) TC_LERP(z) TC_LERP(w)
#undef TC_LERP
}
// computed - projected depth (train_mono.py:91) WITHOUT cancellation: both are O(1) numbers that agree to 1e-2 .. 1e-4 on consistent
// depth maps, so their fp32 difference would carry 1e-4 .. 1e-3 relative error -- which the depth-consistency term's IRLS weight
// 1 / max(dd, eps) turns into 1e-5 on the normal equations (measured; GN iterations under a strong depth-consistency weight
// amplify it past the 1e-4 parity bar).  With Z = D + q2, D = es d_t and pd = es (v00 + delta), delta = the bilinear
// interpolation of the taps' differences to tap 00:   cd - pd = es ((d_t - v00) - delta) + q2   -- every term small and exact
// to its own ulp (d_t - v00 is exact by Sterbenz when the two depths are within a factor of two).
__device__ __forceinline__ float dc_diff(const PairConst &c, const Geo &g, const Tap &t, float depth_t, float pd) {
    const float wx = t.wx, wy = t.wy;
    const float e01 = t.v01.w - t.v00.w, e10 = t.v10.w - t.v00.w, e11 = t.v11.w - t.v00.w;
    const float dlt = wx * ((1.f - wy) * e01) + wy * ((1.f - wx) * e10) + (wx * wy) * e11;
    return g.zcl ? g.Z - pd : c.es * ((depth_t - t.v00.w) - dlt) + g.q2;
}

__device__ __forceinline__ void tap4(const float4 *__restrict__ img, int W, int H, int ui, int vi, float rx, float ry, bool oob,
                                     float4 &val, float4 &gx, float4 &gy) {
    Tap t;
    tap4_fetch(img, W, H, ui, vi, rx, ry, oob, t);
    tap4_lerp(t, val, gx, gy);
}

// Jacobian of the sample position and of Z w.r.t. the left SE(3) perturbation [rho, phi] (+ log depth-scale)
//   dXp/drho_j = e_j ; dXp/dphi_j = e_j x Xp ; dXp/dsigma = Xp - t   ; pinhole K
template <int NP>
__device__ __forceinline__ void geo_jac(const PairConst &c, const Geo &g, int W, int H, float *a, float *b, float *zc) {
    const float cw = (float)W * frcp((float)(W - 1)), ch = (float)H * frcp((float)(H - 1));
    float dp0[TC_MAXP], dp1[TC_MAXP], dp2[TC_MAXP];
    dp0[0] = c.fx;  dp1[0] = 0.f;   dp2[0] = 0.f;
    dp0[1] = 0.f;   dp1[1] = c.fy;  dp2[1] = 0.f;
    dp0[2] = c.cx;  dp1[2] = c.cy;  dp2[2] = 1.f;
    dp0[3] = c.cx * g.X1;               dp1[3] = -c.fy * g.X2 + c.cy * g.X1; dp2[3] = g.X1;
    dp0[4] = c.fx * g.X2 - c.cx * g.X0; dp1[4] = -c.cy * g.X0;               dp2[4] = -g.X0;
    dp0[5] = -c.fx * g.X1;              dp1[5] = c.fy * g.X0;                dp2[5] = 0.f;
    if (NP == 7) {
        float q0 = g.X0 - c.t[0], q1 = g.X1 - c.t[1], q2 = g.X2 - c.t[2];
        dp0[6] = c.fx * q0 + c.cx * q2; dp1[6] = c.fy * q1 + c.cy * q2; dp2[6] = q2;
    }
    float sa = g.oobx ? 0.f : cw * g.iz, sb = g.ooby ? 0.f : ch * g.iz, zf = g.zcl ? 0.f : 1.f;
    // Columns 0, 1, 5 (translation along x / y, rotation about the optical axis) do not change the depth: their dz is a literal zero,
    // not a product the compiler has to keep (x * 0 does not fold under IEEE rules).  Same values as the general expression.
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const bool nz = (j == 2 || j == 3 || j == 4 || j == 6);
        float dz = nz ? zf * dp2[j] : 0.f;
        zc[j] = dz;
        a[j] = nz ? sa * (dp0[j] - g.uz * dz) : sa * dp0[j];
        b[j] = nz ? sb * (dp1[j] - g.vz * dz) : sb * dp1[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// per-pair fp64 state -> fp32 constants of one linearisation; pair initialisation

__device__ inline void write_const(const PairState &S, const double *T, double s, int img, PairConst &c) {
    const double *K = S.K;
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double Ki[9] = {1.0 / fx, 0, -cx / fx, 0, 1.0 / fy, -cy / fy, 0, 0, 1};
    double RmI[9], KR[9], A[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) RmI[3 * i + j] = T[4 * i + j] - (i == j ? 1.0 : 0.0);
    mat3_mul(K, RmI, KR);
    mat3_mul(KR, Ki, A);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            c.A[3 * i + j] = (float)A[3 * i + j];
            c.R[3 * i + j] = (float)T[4 * i + j];
        }
        c.kt[i] = (float)(K[3 * i] * T[3] + K[3 * i + 1] * T[7] + K[3 * i + 2] * T[11]);
        c.t[i] = (float)T[4 * i + 3];
    }
    c.fx = (float)fx; c.fy = (float)fy; c.cx = (float)cx; c.cy = (float)cy;
    c.ki0 = (float)(1.0 / fx); c.ki2 = (float)(-cx / fx); c.ki4 = (float)(1.0 / fy); c.ki5 = (float)(-cy / fy);
    c.es = (s == 0.0) ? 1.f : (float)exp(s);
    c.img = img;
}

// Lane-parallel form of write_const for the solve kernel: lanes 0..8 one entry of A and R each, lanes 0..2 kt and t.
// Intrinsics and image index never change after init_pair and are not rewritten.  T: 3x4 transform in LDS.
// 1/x in double precision without the ~35-instruction IEEE division sequence: v_rcp_f64 seed + two Newton steps
// (quadratic: 2^-26 -> 2^-52).  The solve kernel is one long dependent fp64 chain; every division on it costs ~0.1 us.
__device__ __forceinline__ double rcp64(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

template <int NP>
__device__ __forceinline__ void write_const_lanes(int lane, const double *K, const double *T, double s, PairConst &c) {
    if (lane < 9) {
        const int i = lane / 3, j = lane - 3 * i;
        const double ifx = rcp64(K[0]), ify = rcp64(K[4]), cx = K[2], cy = K[5];
        const double Ki[9] = {ifx, 0, -cx * ifx, 0, ify, -cy * ify, 0, 0, 1};
        double KR[3];   // row i of K (R - I)
#pragma unroll
        for (int m = 0; m < 3; m++)
            KR[m] = K[3 * i] * (T[m] - (m == 0 ? 1.0 : 0.0)) + K[3 * i + 1] * (T[4 + m] - (m == 1 ? 1.0 : 0.0)) +
                    K[3 * i + 2] * (T[8 + m] - (m == 2 ? 1.0 : 0.0));
        c.A[lane] = (float)(KR[0] * Ki[j] + KR[1] * Ki[3 + j] + KR[2] * Ki[6 + j]);
        c.R[lane] = (float)T[4 * i + j];
    }
    if (lane < 3) {
        c.kt[lane] = (float)(K[3 * lane] * T[3] + K[3 * lane + 1] * T[7] + K[3 * lane + 2] * T[11]);
        c.t[lane] = (float)T[4 * lane + 3];
    }
    if (NP == 7 && lane == 0) c.es = (s == 0.0) ? 1.f : (float)exp(s);
}

// Coalesced calls (tcsfm_refine_window_queued, round 4): up to TC_MAX_COAL queued B-window calls of one shape run as ONE launch sequence
// over their 2 S B ncall directed pairs; every call keeps its own buffers, the kernels reach them through this pointer table.
// Batch pair n in the stacked order over Bt = ncall cB targets (forward n = s Bt + b, inverse S Bt + s Bt + b); target b belongs to
// call b / cB as its local target b % cB.  Results per window are the bits of the call run on its own (the kernels are batch-independent).
constexpr int TC_MAX_COAL = 16;
struct CoalTab {
    int ncall, cB, cS, pad;
    const float *tgt[TC_MAX_COAL], *src[TC_MAX_COAL], *dt[TC_MAX_COAL], *ds[TC_MAX_COAL], *K[TC_MAX_COAL], *pose[TC_MAX_COAL];
    const float *ls[TC_MAX_COAL];       // per call: initial log depth-scales of its pairs (TCSFM_REFINE_POSE_SCALE), or null (0)
};
struct CoalIdx { int call, bl, s, inv, li; };      // li: the pair's index in ITS call's stacked order
__device__ __forceinline__ CoalIdx coal_index(int ncall, int cB, int cS, int n) {
    const int Bt = ncall * cB, SB = cS * Bt;
    CoalIdx r;
    r.inv = n >= SB;
    const int q = r.inv ? n - SB : n;
    r.s = q / Bt;
    const int b = q - r.s * Bt;
    r.call = b / cB; r.bl = b - r.call * cB;
    r.li = (r.inv ? cS * cB : 0) + r.s * cB + r.bl;
    return r;
}

struct InitParams {
    const float *pose, *log_scale, *K;  // [N,6], [N] or null, [Nimg,3,3]
    PairState *st;
    PairConst *pc;
    int N, shared_image;                // shared_image: all problems read image pair 0 (loss-surface sweep)
    float lambda0;
    int K_mod;                          // window form: pair n uses intrinsics K[n % K_mod] (0: one matrix per pair)
    int *err;                           // host-mapped status word: set to 1 when a pair's intrinsics are not pinhole (or null)
    double *pose_lin;                   // l_pose_consist (or null): [2][N][12] transform of every pair at the linearisation, double-buffered by iteration parity
};

__device__ inline void init_pair(const InitParams &P, int n, const CoalTab *ct = nullptr) {
    PairState &S = P.st[n];
    int img = P.shared_image ? 0 : n;
    const int kidx = P.K_mod > 0 ? n % P.K_mod : img;
    const float *Kp = P.K + kidx * 9, *pp = P.pose + n * 6;
    if (ct != nullptr) {      // coalesced calls: intrinsics and initial pose from the pair's own call
        const CoalIdx ci = coal_index(ct->ncall, ct->cB, ct->cS, n);
        Kp = ct->K[ci.call] + ci.bl * 9; pp = ct->pose[ci.call] + ci.li * 6;
    }
    for (int i = 0; i < 9; i++) S.K[i] = (double)Kp[i];
    double pose[6];
    for (int i = 0; i < 6; i++) pose[i] = (double)pp[i];
    {   // device-side guard of the pinhole contract (the host validates a given intrinsics buffer only once)
        const double *K = S.K;
        if (K[1] != 0.0 || K[3] != 0.0 || K[6] != 0.0 || K[7] != 0.0 || K[8] != 1.0 || K[0] == 0.0 || K[4] == 0.0) {
            for (int i = 0; i < 6; i++) pose[i] = __longlong_as_double(0x7ff8000000000000LL);  // NaN: fail loudly ...
            if (P.err) *reinterpret_cast<volatile int *>(P.err) = 1;   // ... and report TCSFM_E_INTRINSICS at the next call / synchronize
        }
    }
    pose_to_T(pose, S.Tcur);
    for (int i = 0; i < 12; i++) S.Ttry[i] = S.Tcur[i];
    if (P.pose_lin)
        for (int i = 0; i < 12; i++) P.pose_lin[(size_t)n * 12 + i] = S.Tcur[i];          // (iteration 0 reads buffer 0)
    double ls0 = P.log_scale ? (double)P.log_scale[n] : 0.0;
    if (ct != nullptr) { const CoalIdx ci = coal_index(ct->ncall, ct->cB, ct->cS, n); ls0 = ct->ls[ci.call] ? (double)ct->ls[ci.call][ci.li] : 0.0; }
    S.scur = S.stry = S.s0 = ls0;
    S.lambda = (double)P.lambda0;
    S.cost_cur = 0.0;
    S.have_cur = 0;
    write_const(S, S.Ttry, S.stry, img, P.pc[n]);
}

__global__ void k_init(InitParams P) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < P.N) init_pair(P, n);
}

// ---------------------------------------------------------------------------------------------------------------
// k_pack
// Window form with explicit source positions: source (s, b) is image win_off.off[s] + b of the `src` / `depth_s` arrays instead of the
// standard s B + b.  The sequence calls use it: the targets and every source of consecutive windows are runs of ONE frame ring
// (target = frame w + t0, source s = frame w + off[s]), so B windows with any number of sources are refined by pointer.
constexpr int TC_MAX_SRC_OFF = 8;
struct WinOff {
    int on;                       // 0: standard layout
    int off[TC_MAX_SRC_OFF];
};
__device__ __forceinline__ int win_src_image(const WinOff &w, int q, int B) { return w.on ? w.off[q / B] + (q - (q / B) * B) : q; }

struct PackParams {
    const float *tgt, *src, *depth_t, *depth_s;  // planar inputs [N,3,H,W] / [N,1,H,W]
    float4 *tgtpack, *srcpack;
    float *depth_out;                            // [N,H,W] depth_t (converted if depth_is_disp)
    float *depth_out2;                           // optional second copy of the same (dense mode: the prior centre depth0), or null
    float *depth_out3;                           // optional third copy (dense mode on the reference's loss: the CALLER's depth output, whose inverse
                                                 // slots -- the source maps, not unknowns there -- are final at once), or null
    int *zero_ints; int zero_n;                  // optional: words zeroed by this launch (the batch counters of that mode), or null
    int tshare;                                  // window forms of the pose modes (LinParams::tshare): no tgtpack, no depth plane -- the bordered packs
                                                 // of both images and ONE auto-mask error plane per couple (in depth_out, at the forward pair's index)
    float *c_depth_out3[TC_MAX_COAL];            // coalesced calls: depth_out3 per call (pair li of call c at c_depth_out3[c] + li H W); used when c_out3 != 0
    int c_out3;
    int H, W, N;
    float wl, ws;                                // w_l1/3, w_ssim/3
    int depth_is_disp;
    float min_disp, max_disp;
    InitParams init;                             // init.N > 0: pair initialisation fused into this launch (one thread per pair)
    // window form (win_B > 0): tgt [B,3,H,W], src [S,B,3,H,W], depth_t [B,1,H,W], depth_s [S,B,1,H,W]; directed pairs in
    // the stacked order of train_mono.py:54-62 -- n = s B + b forward (tgt b <- src s), S B + s B + b inverse -- are formed
    // here by indexing, the caller never materialises the repeated / concatenated tensors
    int win_B, win_S;
    WinOff win_off;
};

// (w_l1 |y-x|.clamp + w_ssim SSIM(x,y)).mean(C) at one pixel straight from planar global memory.
// The 9 reflect-padded neighbour offsets are formed once (3 row + 3 column indices) and shared by the 6 planes: the first
// version recomputed refl_idx per load and spent most of its instructions on addresses.
__device__ inline float photo_err_planar(const float *__restrict__ x, const float *__restrict__ y, int H, int W, int u, int v,
                                         float wl, float ws) {
    float acc = 0.f;
    const int hw = H * W;
    const int r0 = refl_idx(v - 1, H) * W, r1 = v * W, r2 = refl_idx(v + 1, H) * W;
    const int c0 = refl_idx(u - 1, W), c1 = u, c2 = refl_idx(u + 1, W);
    const int off[9] = {r0 + c0, r0 + c1, r0 + c2, r1 + c0, r1 + c1, r1 + c2, r2 + c0, r2 + c1, r2 + c2};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float *xc = x + c * hw, *yc = y + c * hw;
        float xv[9], yv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { xv[k] = xc[off[k]]; yv[k] = yc[off[k]]; }
        const float x0 = xv[4], y0 = yv[4];
        float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float a = xv[k] - x0, b = yv[k] - y0;  // shifted by the centre value: fp32-safe variances
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
        const float n9 = 1.f / 9.f;
        float mdx = sx * n9, mdy = sy * n9, mux = x0 + mdx, muy = y0 + mdy;
        float sigx = sxx * n9 - mdx * mdx, sigy = syy * n9 - mdy * mdy, sigxy = sxy * n9 - mdx * mdy;
        float n = (2.f * mux * muy + SSIM_C1) * (2.f * sigxy + SSIM_C2);
        // (the two squares are rounded on their own: mux mux + muy muy must not contract into ONE fma, whose result depends on which of
        // the two images is called x -- the error of (x, y) and of (y, x) is the same number, bit for bit, and the window forms compute
        // it once for a forward pair and its inverse)
        float d = ((__fmul_rn(mux, mux) + __fmul_rn(muy, muy)) + SSIM_C1) * (sigx + sigy + SSIM_C2);
        acc += wl * clamp01(fabsf(y0 - x0)) + ws * clamp01((1.f - n * frcp(d)) * 0.5f);
    }
    return acc;
}

// The same value from an LDS tile (round 4): a workgroup owns a PT_W x PT_H tile of pixels, stages the tile + 1-pixel reflect halo of the six
// colour planes once (about 9 coalesced loads per thread instead of 54 cache hits) and every thread reads its 3 x 3 window from LDS.  Same
// neighbours (the halo is filled through refl_idx, as photo_err_planar indexes), same operation order -> the same bits.
#ifndef TC_PACK_TW
#define TC_PACK_TW 64
#endif
constexpr int PT_W = TC_PACK_TW, PT_H = 256 / PT_W, PT_CW = PT_W + 2, PT_CH = PT_H + 2, PT_N = PT_CW * PT_CH;      // (A/B: -DTC_PACK_TW=32 -> 32 x 8 tiles)
__device__ __forceinline__ void pack_stage_tile(float (*tile)[PT_N], const float *__restrict__ x, const float *__restrict__ y, int H, int W, int x0, int y0) {
    const int hw = H * W;
    for (int e = threadIdx.x; e < PT_N; e += 256) {
        const int ly = e / PT_CW, lx = e - ly * PT_CW;
        const int gi = refl_idx(y0 + ly - 1, H) * W + refl_idx(x0 + lx - 1, W);
#pragma unroll
        for (int c = 0; c < 3; c++) { tile[c][e] = x[c * hw + gi]; tile[3 + c][e] = y[c * hw + gi]; }
    }
}
__device__ __forceinline__ float photo_err_tile(const float (*tile)[PT_N], int tx, int ty, float wl, float ws, float *xc3, float *yc3) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float xv[9], yv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int q = (ty + k / 3) * PT_CW + tx + (k % 3);
            xv[k] = tile[c][q]; yv[k] = tile[3 + c][q];
        }
        const float x0 = xv[4], y0 = yv[4];
        xc3[c] = x0; yc3[c] = y0;
        float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float a = xv[k] - x0, b = yv[k] - y0;  // shifted by the centre value: fp32-safe variances
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
        const float n9 = 1.f / 9.f;
        float mdx = sx * n9, mdy = sy * n9, mux = x0 + mdx, muy = y0 + mdy;
        float sigx = sxx * n9 - mdx * mdx, sigy = syy * n9 - mdy * mdy, sigxy = sxy * n9 - mdx * mdy;
        float n = (2.f * mux * muy + SSIM_C1) * (2.f * sigxy + SSIM_C2);
        float d = ((__fmul_rn(mux, mux) + __fmul_rn(muy, muy)) + SSIM_C1) * (sigx + sigy + SSIM_C2);      // (symmetric in x, y: see photo_err_planar)
        acc += wl * clamp01(fabsf(y0 - x0)) + ws * clamp01((1.f - n * frcp(d)) * 0.5f);
    }
    return acc;
}

// source image of pair n with its 1-texel zero border (see tap4); edge pixels also write the border texels next to them
__device__ __forceinline__ void pack_write_src(const PackParams &P, int n, int u, int v, const float4 &val) {
    const int WB = P.W + 2;
    float4 *sp = P.srcpack + (size_t)n * (P.H + 2) * WB;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    sp[(v + 1) * WB + u + 1] = val;
    if (u == 0) sp[(v + 1) * WB] = zero;
    if (u == P.W - 1) sp[(v + 1) * WB + P.W + 1] = zero;
    if (v == 0) { sp[u + 1] = zero; if (u == 0) sp[0] = zero; if (u == P.W - 1) sp[P.W + 1] = zero; }
    if (v == P.H - 1) {
        sp[(P.H + 1) * WB + u + 1] = zero;
        if (u == 0) sp[(P.H + 1) * WB] = zero;
        if (u == P.W - 1) sp[(P.H + 1) * WB + P.W + 1] = zero;
    }
}

// Pair form (win_B == 0, no table): blockIdx.y = directed pair n.  Window forms (win_B > 0, or the coalesced calls' table): blockIdx.y =
// the forward pair q < S B; the thread packs pair q AND its inverse S B + q -- the same two images with their roles swapped, and
// photo_err_planar(x, y) == photo_err_planar(y, x) bit for bit -- so every input value is loaded once and the 3x3 error evaluated once
// (round 4: k_pack 8.6 -> see DESIGN section 4; the bits equal those of the pair form on the same images).
__device__ __forceinline__ void pack_body(const PackParams &P, const CoalTab *ct) {
    __shared__ float tile[6][PT_N];
    const int tiles_x = (P.W + PT_W - 1) / PT_W;
#ifndef TC_PACK_XCD
#define TC_PACK_XCD 1
#endif
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin, so with tiles in launch order the tiles above and below a tile --
    // which share two of its six staged rows -- sit in OTHER XCDs' L2 and every halo row is fetched from memory again.  Workgroup i takes
    // tile (i % 8) * (tiles / 8) + i / 8 (remainders spread over the first XCDs): the tiles of one XCD form a contiguous band of the image.
    int tile_id = blockIdx.x;
    if (TC_PACK_XCD) {
        const int nt = gridDim.x, q = nt >> 3, r = nt & 7, x = tile_id & 7, k = tile_id >> 3;
        tile_id = x < r ? x * (q + 1) + k : r * (q + 1) + (x - r) * q + k;
    }
    const int tyi = tile_id / tiles_x, txi = tile_id - tyi * tiles_x;
    const int x0 = txi * PT_W, y0 = tyi * PT_H, tx = threadIdx.x & (PT_W - 1), ty = threadIdx.x / PT_W;
    int n = blockIdx.y;
    const int hw = P.H * P.W;
    const bool both = ct != nullptr || P.win_B > 0;
    int n_inv = 0;
    const float *t = P.tgt + (size_t)n * 3 * hw, *s = P.src + (size_t)n * 3 * hw;
    const float *dtp = P.depth_t + (size_t)n * hw, *dsp = P.depth_s + (size_t)n * hw;
    if (ct != nullptr) {      // coalesced calls: the pair's images live in its own call's buffers (window layout of that call)
        const CoalIdx ci = coal_index(ct->ncall, ct->cB, ct->cS, n);
        n_inv = ct->cS * ct->ncall * ct->cB + n;
        t = ct->tgt[ci.call] + (size_t)ci.bl * 3 * hw; s = ct->src[ci.call] + (size_t)(ci.s * ct->cB + ci.bl) * 3 * hw;
        dtp = ct->dt[ci.call] + (size_t)ci.bl * hw; dsp = ct->ds[ci.call] + (size_t)(ci.s * ct->cB + ci.bl) * hw;
    } else if (P.win_B > 0) {
        const int b = n % P.win_B, qi = win_src_image(P.win_off, n, P.win_B);
        n_inv = P.win_S * P.win_B + n;
        t = P.tgt + (size_t)b * 3 * hw; s = P.src + (size_t)qi * 3 * hw;
        dtp = P.depth_t + (size_t)b * hw; dsp = P.depth_s + (size_t)qi * hw;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2) {            // independent of the packing below
        const int ni = threadIdx.x == 0 ? n : n_inv;
        if ((threadIdx.x == 0 || both) && ni < P.init.N) init_pair(P.init, ni, ct);
    }
    if (P.zero_ints != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x >= 64 && (int)threadIdx.x - 64 < P.zero_n) P.zero_ints[threadIdx.x - 64] = 0;
    pack_stage_tile(tile, t, s, P.H, P.W, x0, y0);
    __syncthreads();
    const int u = x0 + tx, v = y0 + ty, idx = v * P.W + u;
    if (u >= P.W || v >= P.H) return;
    float tc[3], sc[3];
    float ae = photo_err_tile(tile, tx, ty, P.wl, P.ws, tc, sc);
    float dt = dtp[idx], ds = dsp[idx];
    if (P.depth_is_disp) {  // disp_to_depth, learning_helpers.py:77-86
        dt = 1.f / (P.min_disp + (P.max_disp - P.min_disp) * dt);
        ds = 1.f / (P.min_disp + (P.max_disp - P.min_disp) * ds);
    }
    const float t0 = tc[0], t1 = tc[1], t2 = tc[2], s0 = sc[0], s1 = sc[1], s2 = sc[2];
    if (P.tshare) {         // (both: see PackParams::tshare)
        pack_write_src(P, n, u, v, make_float4(s0, s1, s2, ds));
        pack_write_src(P, n_inv, u, v, make_float4(t0, t1, t2, dt));
        P.depth_out[(size_t)n * hw + idx] = ae;
        return;
    }
    P.tgtpack[(size_t)n * hw + idx] = make_float4(t0, t1, t2, ae);
    pack_write_src(P, n, u, v, make_float4(s0, s1, s2, ds));
    P.depth_out[(size_t)n * hw + idx] = dt;
    if (P.depth_out2) P.depth_out2[(size_t)n * hw + idx] = dt;
    float *o3 = P.depth_out3 ? P.depth_out3 + (size_t)n * hw : nullptr, *o3i = P.depth_out3 ? P.depth_out3 + (size_t)n_inv * hw : nullptr;
    if (ct != nullptr && P.c_out3) {
        const CoalIdx ci = coal_index(ct->ncall, ct->cB, ct->cS, n);
        o3 = P.c_depth_out3[ci.call] + (size_t)ci.li * hw; o3i = P.c_depth_out3[ci.call] + (size_t)(ct->cS * ct->cB + ci.li) * hw;
    }
    if (o3) o3[idx] = dt;
    if (both) {             // the inverse pair: target and source swapped, the same error
        P.tgtpack[(size_t)n_inv * hw + idx] = make_float4(s0, s1, s2, ae);
        pack_write_src(P, n_inv, u, v, make_float4(t0, t1, t2, dt));
        P.depth_out[(size_t)n_inv * hw + idx] = ds;
        if (P.depth_out2) P.depth_out2[(size_t)n_inv * hw + idx] = ds;
        if (o3i) o3i[idx] = ds;
    }
}
__global__ __launch_bounds__(256) void k_pack(PackParams P) { pack_body(P, nullptr); }
__global__ __launch_bounds__(256) void k_pack_coal(PackParams P, CoalTab T) { pack_body(P, &T); }

// ---------------------------------------------------------------------------------------------------------------
// Frame-level pack cache of the sequence calls (run_sequential_optimization.py:186-247 streams a sequence; every frame belongs to
// S + 1 windows, as a target once and as a source S times).  k_frame_pack runs ONCE per frame, on the copy stream right behind
// the frame's H2D copy: planar rgb + depth -> the bordered (rgb, depth) float4 image the warp gathers from + the (converted)
// depth plane.  k_pack_cached then forms, per window call, only what is pair-specific: tgtpack = (target rgb, auto_err of THIS
// (target, source) pair) and the pair's slots -- 16 B/pixel/pair written instead of 36.  Same values, bit for bit, as k_pack.
struct FramePackParams {
    const float *img, *depth;     // [F][3][H][W], [F][1][H][W] planar (ring slots)
    float4 *fpack;                // [F][H+2][W+2]
    float *fdepth;                // [F][H][W]
    int H, W, depth_is_disp;
    float min_disp, max_disp;
};
__global__ __launch_bounds__(256) void k_frame_pack(FramePackParams P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int hw = P.H * P.W;
    if (idx >= hw) return;
    const int v = idx / P.W, u = idx - v * P.W;
    const float *s = P.img + (size_t)f * 3 * hw;
    float d = P.depth[(size_t)f * hw + idx];
    if (P.depth_is_disp) d = 1.f / (P.min_disp + (P.max_disp - P.min_disp) * d);       // disp_to_depth, learning_helpers.py:77-86
    const int WB = P.W + 2;
    float4 *sp = P.fpack + (size_t)f * (P.H + 2) * WB;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    sp[(v + 1) * WB + u + 1] = make_float4(s[idx], s[hw + idx], s[2 * hw + idx], d);
    if (u == 0) sp[(v + 1) * WB] = zero;
    if (u == P.W - 1) sp[(v + 1) * WB + P.W + 1] = zero;
    if (v == 0) { sp[u + 1] = zero; if (u == 0) sp[0] = zero; if (u == P.W - 1) sp[P.W + 1] = zero; }
    if (v == P.H - 1) {
        sp[(P.H + 1) * WB + u + 1] = zero;
        if (u == 0) sp[(P.H + 1) * WB] = zero;
        if (u == P.W - 1) sp[(P.H + 1) * WB + P.W + 1] = zero;
    }
    P.fdepth[(size_t)f * hw + idx] = d;
}

struct PackCachedParams {
    const float4 *fpack;          // [F][H+2][W+2] frame packs (ring)
    float4 *tgtpack;              // [N][H][W] out
    int *pair_src, *pair_dep;     // [N] out
    int H, W, N, win_B, win_S;
    int slot0, tpos;              // ring slot of the call's first frame; position of the target inside a window
    WinOff win_off;               // source s of window b = frame slot0 + off[s] + b
    float wl, ws;
    InitParams init;
};
// photo_err_planar on two bordered frame packs: same loads (values), same operation order -> the same bits
__device__ inline float photo_err_packs(const float4 *__restrict__ x, const float4 *__restrict__ y, int H, int W, int u, int v, float wl, float ws) {
    const int WB = W + 2;
    const int r0 = (refl_idx(v - 1, H) + 1) * WB, r1 = (v + 1) * WB, r2 = (refl_idx(v + 1, H) + 1) * WB;
    const int c0 = refl_idx(u - 1, W) + 1, c1 = u + 1, c2 = refl_idx(u + 1, W) + 1;
    const int off[9] = {r0 + c0, r0 + c1, r0 + c2, r1 + c0, r1 + c1, r1 + c2, r2 + c0, r2 + c1, r2 + c2};
    float4 xv[9], yv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { xv[k] = x[off[k]]; yv[k] = y[off[k]]; }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        auto ch = [&](const float4 &q) { return c == 0 ? q.x : (c == 1 ? q.y : q.z); };
        const float x0 = ch(xv[4]), y0 = ch(yv[4]);
        float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float a = ch(xv[k]) - x0, b = ch(yv[k]) - y0;
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
        const float n9 = 1.f / 9.f;
        float mdx = sx * n9, mdy = sy * n9, mux = x0 + mdx, muy = y0 + mdy;
        float sigx = sxx * n9 - mdx * mdx, sigy = syy * n9 - mdy * mdy, sigxy = sxy * n9 - mdx * mdy;
        float n = (2.f * mux * muy + SSIM_C1) * (2.f * sigxy + SSIM_C2);
        // (the two squares are rounded on their own: mux mux + muy muy must not contract into ONE fma, whose result depends on which of
        // the two images is called x -- the error of (x, y) and of (y, x) is the same number, bit for bit, and the window forms compute
        // it once for a forward pair and its inverse)
        float d = ((__fmul_rn(mux, mux) + __fmul_rn(muy, muy)) + SSIM_C1) * (sigx + sigy + SSIM_C2);
        acc += wl * clamp01(fabsf(y0 - x0)) + ws * clamp01((1.f - n * frcp(d)) * 0.5f);
    }
    return acc;
}
// blockIdx.y = forward pair q < S B; the thread also serves the inverse pair S B + q (the same error, see pack_body)
__global__ __launch_bounds__(256) void k_pack_cached(PackCachedParams P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    const int hw = P.H * P.W;
    const int SB = P.win_S * P.win_B, b = n % P.win_B, s = n / P.win_B, n_inv = SB + n;
    const int ts = P.slot0 + P.tpos + b, ss = P.slot0 + P.win_off.off[s] + b;     // window b: its target / its source s
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        const bool inv = threadIdx.x == 1;
        const int ni = inv ? n_inv : n;
        P.pair_src[ni] = inv ? ts : ss; P.pair_dep[ni] = inv ? ss : ts;           // THIS pair's source / target frame
        if (ni < P.init.N) init_pair(P.init, ni);
    }
    if (idx >= hw) return;
    const int v = idx / P.W, u = idx - v * P.W;
    const size_t fs = (size_t)(P.H + 2) * (P.W + 2);
    const float4 *t = P.fpack + (size_t)ts * fs, *sp = P.fpack + (size_t)ss * fs;
    const float ae = photo_err_packs(t, sp, P.H, P.W, u, v, P.wl, P.ws);
    const float4 tc = t[(v + 1) * (P.W + 2) + u + 1], sc = sp[(v + 1) * (P.W + 2) + u + 1];
    P.tgtpack[(size_t)n * hw + idx] = make_float4(tc.x, tc.y, tc.z, ae);
    P.tgtpack[(size_t)n_inv * hw + idx] = make_float4(sc.x, sc.y, sc.z, ae);
}

// Per-pixel min over the S sources of one target (compute_optimization_loss, optimizer.py:47-69; oracle orc_window_select), dense window
// modes: does forward pair n = s B + b keep pixel gi?  From the residual maps of all sources of its target at the current poses: the
// source with the smallest error (first minimum, as torch.min), under the union validity and the auto-mask of the minima.  Every
// workgroup that needs the answer computes it (a few loads per pixel) -- a separate selection launch cost more than that.
__device__ __forceinline__ bool ext_selected(const LinParams &P, int n, int gi, int hw) {
    const int b = n % P.ext_B, s_own = n / P.ext_B;
    int smin = 0;
    float dmin = 0.f, amin = 0.f, vany = 0.f;
    for (int s = 0; s < P.ext_S; s++) {
        const size_t o = (size_t)(s * P.ext_B + b) * hw + gi;
        const float d = P.ext_diff[o], a = P.tgtpack[o].w, v = P.ext_valid[o];
        if (s == 0 || d < dmin) { dmin = d; smin = s; }
        amin = (s == 0) ? a : fminf(amin, a);
        vany = fmaxf(vany, v);
    }
    return vany > 0.f && (!P.automask || dmin < amin) && s_own == smin;
}

// SSIM_Loss.forward, losses.py:27-41, on C planes of N images: x, y [N*C, H, W] -> out (same shape)
__global__ __launch_bounds__(256) void k_ssim(const float *x, const float *y, float *out, int H, int W) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = H * W;
    if (idx >= hw) return;
    int v = idx / W, u = idx - v * W;
    const float *xc = x + (size_t)blockIdx.y * hw, *yc = y + (size_t)blockIdx.y * hw;
    float x0 = xc[idx], y0 = yc[idx];
    float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int dv = -1; dv <= 1; dv++)
        for (int du = -1; du <= 1; du++) {
            int j = refl_idx(v + dv, H) * W + refl_idx(u + du, W);
            float a = xc[j] - x0, b = yc[j] - y0;
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    const float n9 = 1.f / 9.f;
    float mdx = sx * n9, mdy = sy * n9, mux = x0 + mdx, muy = y0 + mdy;
    float sigx = sxx * n9 - mdx * mdx, sigy = syy * n9 - mdy * mdy, sigxy = sxy * n9 - mdx * mdy;
    float n = (2.f * mux * muy + SSIM_C1) * (2.f * sigxy + SSIM_C2);
    float d = (mux * mux + muy * muy + SSIM_C1) * (sigx + sigy + SSIM_C2);
    out[(size_t)blockIdx.y * hw + idx] = clamp01((1.f - n / d) * 0.5f);
}

__global__ void k_disp_to_depth(const float *disp, float *scaled, float *depth, long long n, float min_disp, float max_disp) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = min_disp + (max_disp - min_disp) * disp[i];
    if (scaled) scaled[i] = s;
    if (depth) depth[i] = 1.f / s;
}

// ---------------------------------------------------------------------------------------------------------------
// k_warp: inverse_warp2 drop-in on planar inputs
struct WarpParams {
    const float *src, *depth_t, *depth_s;
    const PairConst *pc;
    float *rec, *valid, *pd, *cd;
    const float *tgt;   // optional: with posenet_in, the target image to be masked by the warp validity
    float *posenet_in;  // optional [N,6,H,W]: (tgt * valid, img_rec) = the next PoseNet input of solve_pose_iteratively
    int H, W;           //                     (train_mono.py:74-76), written by the warp itself: no extra HBM round trip
    int win_B, win_S;   // window form (win_B > 0): tgt [B,3,H,W], src [S,B,3,H,W], depth_t [B,1,H,W], depth_s [S,B,1,H,W]; pair n as in k_pack
    WinOff win_off;
};

__device__ __forceinline__ float tap1(const float *__restrict__ img, int W, int H, int ui, int vi, float rx, float ry, bool oob) {
    float fx = floorf(rx), fy = floorf(ry);
    float wx = rx - fx, wy = ry - fy;
    int xi = ui + (int)fx, yi = vi + (int)fy;
    bool x0in = (xi >= 0) && (xi < W), x1in = (xi >= -1) && (xi < W - 1);
    bool y0in = (yi >= 0) && (yi < H), y1in = (yi >= -1) && (yi < H - 1);
    int x0 = min(max(xi, 0), W - 1), x1 = min(max(xi + 1, 0), W - 1);
    int y0 = min(max(yi, 0), H - 1), y1 = min(max(yi + 1, 0), H - 1);
    float v00 = (x0in && y0in && !oob) ? img[y0 * W + x0] : 0.f, v01 = (x1in && y0in && !oob) ? img[y0 * W + x1] : 0.f;
    float v10 = (x0in && y1in && !oob) ? img[y1 * W + x0] : 0.f, v11 = (x1in && y1in && !oob) ? img[y1 * W + x1] : 0.f;
    return (1.f - wx) * (1.f - wy) * v00 + wx * (1.f - wy) * v01 + (1.f - wx) * wy * v10 + wx * wy * v11;
}

__global__ __launch_bounds__(256) void k_warp(WarpParams P) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int n = blockIdx.y;
    const int hw = P.H * P.W;
    if (idx >= hw) return;
    int v = idx / P.W, u = idx - v * P.W;
    const PairConst &c = P.pc[n];
    const float *tgt = P.tgt ? P.tgt + (size_t)n * 3 * hw : nullptr, *src = P.src + (size_t)n * 3 * hw;
    const float *dtp = P.depth_t + (size_t)n * hw, *dsp = P.depth_s + (size_t)n * hw;
    if (P.win_B > 0) {   // the directed pairs of a window, formed by indexing (train_mono.py:54-62)
        const int SB = P.win_S * P.win_B, inv = n >= SB, q = inv ? n - SB : n, b = q % P.win_B, qi = win_src_image(P.win_off, q, P.win_B);
        const float *ti = P.tgt + (size_t)b * 3 * hw, *si = P.src + (size_t)qi * 3 * hw;
        const float *td = P.depth_t + (size_t)b * hw, *sd = P.depth_s + (size_t)qi * hw;
        tgt = inv ? si : ti; src = inv ? ti : si; dtp = inv ? sd : td; dsp = inv ? td : sd;
    }
    Geo g;
    warp_geo(c, P.W, P.H, u, v, dtp[idx], g);
    const bool oob = g.oobx || g.ooby;
    if (P.rec)
        for (int ch = 0; ch < 3; ch++)
            P.rec[((size_t)n * 3 + ch) * hw + idx] = tap1(src + (size_t)ch * hw, P.W, P.H, u, v, g.rx, g.ry, oob);
    if (P.valid) P.valid[(size_t)n * hw + idx] = oob ? 0.f : 1.f;
    if (P.posenet_in) {
        const float vm = oob ? 0.f : 1.f;
        for (int ch = 0; ch < 3; ch++) {
            P.posenet_in[((size_t)n * 6 + ch) * hw + idx] = tgt[(size_t)ch * hw + idx] * vm;
            P.posenet_in[((size_t)n * 6 + 3 + ch) * hw + idx] = tap1(src + (size_t)ch * hw, P.W, P.H, u, v, g.rx, g.ry, oob);
        }
    }
    if (P.pd) P.pd[(size_t)n * hw + idx] = c.es * tap1(dsp, P.W, P.H, u, v, g.rx, g.ry, oob);
    if (P.cd) P.cd[(size_t)n * hw + idx] = g.Z;
}

// ---------------------------------------------------------------------------------------------------------------
// k_linearize
//
// Workgroup = one TW x TH tile of target pixels of one pair.  Phase 1 warps the tile plus a 1-pixel halo
// (reflect-mapped at the image border, so the halo IS the ReflectionPad2d of losses.py:22) and stages per pixel
//   y[3] (warped source), x[3] (target), gx[3], gy[3] (d rec/d ix,iy), a[NP], b[NP] (d ix, d iy / d theta)
// in LDS as an array of 112-byte records (7 x float4: conflict-free ds_read_b128 for consecutive lanes).
// Phase 2 evaluates SSIM / L1 / masks and the exact gradient rows for the tile's own pixels from the 3x3 LDS
// neighbourhood and accumulates J'J and J'r in registers; one wave-reduce + LDS reduce per workgroup, one
// partial-sum record per workgroup to HBM (deterministic: no atomics).

constexpr int RG = 16;       // workgroups per in-launch reduction group
constexpr int LDS_REC = 28;  // floats per staged pixel: 24 used (six 16-byte rows, NP = 7 included), stride 28 for the banks (28 l mod 64 visits every bank group once)

template <int NP>
struct AccLayout {
    static constexpr int NH = NP * (NP + 1) / 2;
    static constexpr int OFF_HP = 0, OFF_GP = NH, OFF_HD = NH + NP, OFF_GD = 2 * NH + NP, OFF_S = 2 * NH + 2 * NP;
    static constexpr int NACC = 2 * NH + 2 * NP + 3;  // + sum(M W diff), sum(M), sum(dd)
};

enum { MODE_COST = 0, MODE_LIN = 1, MODE_MAPS = 2 };

// Forced 128-bit LDS accesses.  Left to itself hipcc splits partially-used or register-scattered float4 accesses of the
// 112-byte records into ds_read2_b64 / ds_write2_b32 / ds_read_b96, whose banking conflicts 2- to 4-way on that stride
// (PMC: 49 % of LDS cycles were bank-conflict cycles); ds_read_b128 / ds_write_b128 are conflict-free on it.
// The waits are inside the asm statements because hipcc does not track asm loads (cdna_hip_programming.md section 5.7).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void lds_read3(const float4 *p, float4 &a, float4 &b, float4 &c) {
    f32x4 x, y, z;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(lds_addr(p)) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w); c = make_float4(z.x, z.y, z.z, z.w);
}
__device__ __forceinline__ void lds_read3b(const float4 *p, float4 &a, float4 &b, float4 &c) {  // record floats 12..23
    f32x4 x, y, z;
    asm volatile("ds_read_b128 %0, %3 offset:48\n\tds_read_b128 %1, %3 offset:64\n\tds_read_b128 %2, %3 offset:80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(lds_addr(p)) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w); c = make_float4(z.x, z.y, z.z, z.w);
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_read3v(const float4 *p, f32x4 &a, f32x4 &b, f32x4 &c) {   // record floats 0..11
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(lds_addr(p)) : "memory");
}
__device__ __forceinline__ void lds_read3bv(const float4 *p, f32x4 &a, f32x4 &b, f32x4 &c) {  // record floats 12..23
    asm volatile("ds_read_b128 %0, %3 offset:48\n\tds_read_b128 %1, %3 offset:64\n\tds_read_b128 %2, %3 offset:80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(lds_addr(p)) : "memory");
}
// Packed multiply / FMA with ONE half of `s` broadcast over both halves of `v` (HI = 0: s.x, 1: s.y).  hipcc broadcasts a low half through
// op_sel_hi but copies a HIGH half into a low one first (one v_mov per use); the modifiers select it in place.
template <int HI>
__device__ __forceinline__ f2 pk_mul_b(f2 s, f2 v) {
    f2 r;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(s), "v"(v));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(v));
    return r;
}
template <int HI>
__device__ __forceinline__ void pk_fma_b(f2 &d, f2 s, f2 v) {      // d += s.{x|y} * v
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(s), "v"(v));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(d) : "v"(s), "v"(v));
}
// a - b on both halves in ONE instruction (hipcc lowers a float2 subtraction to two v_sub_f32)
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void lds_read02v(const float4 *p, f32x4 &a, f32x4 &c) {   // record floats 0..3 and 8..11
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(c) : "v"(lds_addr(p)) : "memory");
}
__device__ __forceinline__ void lds_read6v(const float4 *p, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e, f32x4 &f) {  // floats 0..23
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f) : "v"(lds_addr(p)) : "memory");
}
// Split form: issue the six reads of a record, then wait for the first three (LDS returns in order) and for the rest separately, so
// that the arithmetic on the colour / gradient part overlaps the arrival of the Jacobian part.  The "+v" operands tie the uses of the
// registers to the wait (hipcc does not track asm loads).
__device__ __forceinline__ void lds_issue6v(const float4 *p, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e, f32x4 &f) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f) : "v"(lds_addr(p)) : "memory");
}
__device__ __forceinline__ void lds_wait3of6(f32x4 &a, f32x4 &b, f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }
__device__ __forceinline__ void lds_wait0(f32x4 &a, f32x4 &b, f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }
__device__ __forceinline__ void lds_issue02v(const float4 *p, f32x4 &a, f32x4 &c) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:32" : "=&v"(a), "=&v"(c) : "v"(lds_addr(p)) : "memory");
}
// Reads at a compile-time byte offset from a base address (the offset rides in the instruction: no address arithmetic; DS offsets are
// unsigned 16-bit) and counted waits for the software-pipelined neighbour passes: colour part (floats 0..11) / Jacobian part (12..23).
template <int OFF>
__device__ __forceinline__ void lds_issue3c_at(unsigned addr, f32x4 &a, f32x4 &b, f32x4 &c) {
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(addr), "i"(OFF), "i"(OFF + 16), "i"(OFF + 32) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_issue3j_at(unsigned addr, f32x4 &a, f32x4 &b, f32x4 &c) {
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(addr), "i"(OFF + 48), "i"(OFF + 64), "i"(OFF + 80) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_waitn(f32x4 &a, f32x4 &b, f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void lds_waitn(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e, f32x4 &f) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_issue02v_at(unsigned addr, f32x4 &a, f32x4 &c) {       // record floats 0..3 and 8..11
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(a), "=&v"(c) : "v"(addr), "i"(OFF), "i"(OFF + 32) : "memory");
}
__device__ __forceinline__ void lds_wait2(f32x4 &a, f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(c) : : "memory"); }
__device__ __forceinline__ void lds_wait0(f32x4 &a, f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(c) : : "memory"); }
__device__ __forceinline__ float4 lds_read1(const float4 *p) {
    f32x4 x;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(lds_addr(p)) : "memory");
    return make_float4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void lds_write1(float4 *p, float a, float b, float c, float d) {
    f32x4 x = {a, b, c, d};
    asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr(p)), "v"(x) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// Per-channel photometric terms of one pixel from its 3x3 window statistics (centre-shifted sums), written once for
// T = float and T = float2 (two channels per VALU instruction).
typedef int i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vsel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ f2 vsel(i2 c, f2 a, f2 b) { return c ? a : b; }
__device__ __forceinline__ float vrcp(float a) { return frcp(a); }
__device__ __forceinline__ f2 vrcp(f2 a) { return (f2){frcp(a.x), frcp(a.y)}; }
__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
__device__ __forceinline__ f2 vabs(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ f2 vmin(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
template <class T> __device__ __forceinline__ T vsplat(float a);
template <> __device__ __forceinline__ float vsplat<float>(float a) { return a; }
template <> __device__ __forceinline__ f2 vsplat<f2>(float a) { return (f2){a, a}; }

// clamp(a, 0, 1) in one instruction (v_med3_f32), and wl * sign(r) for 0 < |r| <= 1, else 0, with integer compares / bit operations
// (full-rate) in place of three float compares and three selects: |r| lies in (0, 1] exactly when its bit pattern lies in
// [1, 0x3f800000] (non-negative floats order like their bit patterns; NaN patterns are larger: no sign, as before).
__device__ __forceinline__ float vclamp01(float a) { return __builtin_amdgcn_fmed3f(a, 0.f, 1.f); }
__device__ __forceinline__ f2 vclamp01(f2 a) { return (f2){__builtin_amdgcn_fmed3f(a.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(a.y, 0.f, 1.f)}; }
__device__ __forceinline__ float l1_sign(float r, float ar, float wl) {
    const bool in = (__float_as_uint(ar) - 1u) < 0x3f800000u;
    return in ? __uint_as_float(__float_as_uint(wl) ^ (__float_as_uint(r) & 0x80000000u)) : 0.f;
}
__device__ __forceinline__ f2 l1_sign(f2 r, f2 ar, float wl) { return (f2){l1_sign(r.x, ar.x, wl), l1_sign(r.y, ar.y, wl)}; }

template <class T>
struct ChanTerms {
    T e1, e2;          // w_l1/3 |y-x|.clamp(0,1),  w_ssim/3 SSIM            (train_mono.py:87, losses.py:27-41)
    T cA, cB, cC;      // d e2 / d y_q = cA + cB (y_q - y_c) + cC (x_q - x_c)  for the 9 window pixels q
    T id1, id2;        // curvature weights of the mean / covariance parts of SSIM
    T l1x, l1y;        // d e1 / d(ix, iy) of the centre sample
    T lxx, lxy, lyy;   // L1 part of the 2x2 curvature
};

template <class T>
__device__ __forceinline__ void ssim_l1_channel(T xc, T yc, T gxc, T gyc, T Sx, T Sy, T Sxx, T Syy, T Sxy, float ws, float wl,
                                                float eps, ChanTerms<T> &o) {
    const float n9 = 1.f / 9.f;
    const T zero = vsplat<T>(0.f), one = vsplat<T>(1.f);
    T mdx = Sx * n9, mdy = Sy * n9;
    T mux = xc + mdx, muy = yc + mdy;
    T sigx = Sxx * n9 - mdx * mdx, sigy = Syy * n9 - mdy * mdy, sigxy = Sxy * n9 - mdx * mdy;
    T n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sigxy + SSIM_C2;
    T d1 = mux * mux + muy * muy + SSIM_C1, d2 = sigx + sigy + SSIM_C2;
    T idn = vrcp(d1 * d2), ratio = n1 * n2 * idn;
    T raw = (one - ratio) * 0.5f;
    const T rawc = vclamp01(raw);
    auto cl = rawc != raw;                         // clamped: the value is a constant there (no gradient)
    o.e2 = ws * rawc;
    T pre = vsel(cl, zero, idn * (-0.5f * n9 * ws));
    o.cB = pre * (ratio * d1) * -2.f;
    o.cC = pre * n1 * 2.f;
    o.cA = pre * 2.f * (mux * n2 - ratio * muy * d2) - o.cB * mdy - o.cC * mdx;
    T wi = vsel(cl, zero, idn * ws);
    o.id1 = wi * d2; o.id2 = 1.125f * wi * d1;
    // L1 term
    T rr = yc - xc, ar = vabs(rr);
    auto inr = ar <= one;
    o.e1 = wl * vmin(ar, one);
    T sgn = l1_sign(rr, ar, wl);                   // wl sign(rr) inside 0 < |rr| <= 1, else 0
    o.l1x = sgn * gxc; o.l1y = sgn * gyc;
    T w1 = vsel(inr, wl * vrcp(vmax(ar, vsplat<T>(eps))), zero);
    o.lxx = w1 * gxc * gxc; o.lxy = w1 * gxc * gyc; o.lyy = w1 * gyc * gyc;
}

// TILE-SHIFTED form (round 5, k_linearize with TC_TILE_SHIFT): the staged colours of BOTH images carry a per-channel constant shift c (the
// target colour of the tile's first pixel), so the window sums need no per-neighbour subtraction of the centre value (27 packed
// subtractions per pixel): xs, ys = the centre's shifted values, Sx .. Sxy = plain sums of the shifted window values.  The same
// mathematics; the variances cancel against |window mean - c| (the colour variation inside a 32 x 16 tile) instead of |window mean -
// centre| -- measured against the float64 oracle before it was kept (profiles/r05_tile_shift_ab.txt).
template <class T>
__device__ __forceinline__ void ssim_l1_channel_ts(T xs, T ys, T c, T gxc, T gyc, T Sx, T Sy, T Sxx, T Syy, T Sxy, float ws, float wl,
                                                   float eps, ChanTerms<T> &o) {
    const float n9 = 1.f / 9.f;
    const T zero = vsplat<T>(0.f), one = vsplat<T>(1.f);
    T mx = Sx * n9, my = Sy * n9;                  // shifted window means
    T mdx = mx - xs, mdy = my - ys;                // mean - centre
    T mux = mx + c, muy = my + c;
    T sigx = Sxx * n9 - mx * mx, sigy = Syy * n9 - my * my, sigxy = Sxy * n9 - mx * my;
    T n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sigxy + SSIM_C2;
    T d1 = mux * mux + muy * muy + SSIM_C1, d2 = sigx + sigy + SSIM_C2;
    T idn = vrcp(d1 * d2), ratio = n1 * n2 * idn;
    T raw = (one - ratio) * 0.5f;
    const T rawc = vclamp01(raw);
    auto cl = rawc != raw;                         // clamped: the value is a constant there (no gradient)
    o.e2 = ws * rawc;
    T pre = vsel(cl, zero, idn * (-0.5f * n9 * ws));
    o.cB = pre * (ratio * d1) * -2.f;
    o.cC = pre * n1 * 2.f;
    o.cA = pre * 2.f * (mux * n2 - ratio * muy * d2) - o.cB * mdy - o.cC * mdx;
    T wi = vsel(cl, zero, idn * ws);
    o.id1 = wi * d2; o.id2 = 1.125f * wi * d1;
    // L1 term (differences of the two images: the shift cancels)
    T rr = ys - xs, ar = vabs(rr);
    auto inr = ar <= one;
    o.e1 = wl * vmin(ar, one);
    T sgn = l1_sign(rr, ar, wl);
    o.l1x = sgn * gxc; o.l1y = sgn * gyc;
    T w1 = vsel(inr, wl * vrcp(vmax(ar, vsplat<T>(eps))), zero);
    o.lxx = w1 * gxc * gxc; o.lxy = w1 * gxc * gyc; o.lyy = w1 * gyc * gyc;
}
template <class T>
__device__ __forceinline__ T ssim_l1_value_ts(T xs, T ys, T c, T Sx, T Sy, T Sxx, T Syy, T Sxy, float ws, float wl) {
    const float n9 = 1.f / 9.f;
    const T one = vsplat<T>(1.f);
    T mx = Sx * n9, my = Sy * n9, mux = mx + c, muy = my + c;
    T sigx = Sxx * n9 - mx * mx, sigy = Syy * n9 - my * my, sigxy = Sxy * n9 - mx * my;
    T n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sigxy + SSIM_C2;
    T d1 = mux * mux + muy * muy + SSIM_C1, d2 = sigx + sigy + SSIM_C2;
    T raw = (one - n1 * n2 * vrcp(d1 * d2)) * 0.5f;
    return ws * vclamp01(raw) + wl * vmin(vabs(ys - xs), one);
}

// value of the per-channel photometric error only (no gradient coefficients): the residual of ANOTHER source, for the
// min-over-sources selection
template <class T>
__device__ __forceinline__ T ssim_l1_value(T xc, T yc, T Sx, T Sy, T Sxx, T Syy, T Sxy, float ws, float wl) {
    const float n9 = 1.f / 9.f;
    const T zero = vsplat<T>(0.f), one = vsplat<T>(1.f);
    T mdx = Sx * n9, mdy = Sy * n9, mux = xc + mdx, muy = yc + mdy;
    T sigx = Sxx * n9 - mdx * mdx, sigy = Syy * n9 - mdy * mdy, sigxy = Sxy * n9 - mdx * mdy;
    T n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sigxy + SSIM_C2;
    T d1 = mux * mux + muy * muy + SSIM_C1, d2 = sigx + sigy + SSIM_C2;
    T raw = (one - n1 * n2 * vrcp(d1 * d2)) * 0.5f;
    return ws * vclamp01(raw) + wl * vmin(vabs(yc - xc), one);
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup reduction of NLIVE per-thread values -> one record per workgroup -> deterministic in-launch group reduction.
//   v[]: compacted live values [H photo | g photo | (H dc | g dc) | 3 scalars]; dead accumulators are stored as 0.
template <int NP, int NLIVE, bool LIN, bool DC, int NT>
__device__ __forceinline__ void block_reduce_publish(const LinParams &P, const float *v, float *red, int n, int bid, int nblk, int tid) {
    using L = AccLayout<NP>;
    constexpr int NPH = L::NH + NP;
    const int wave = tid >> 6, lane = tid & 63;
#ifdef TC_PROBE_NOREDUCE      // timing-only build (WRONG sums): what the wave butterfly costs (profiles/r04_reduce_probe.txt)
    { float s_ = 0.f;
#pragma unroll
      for (int i = 0; i < NLIVE; i++) s_ += v[i];
      if (lane < NLIVE) red[wave * L::NACC + lane] = s_; }
#else
    wave_reduce_store<NLIVE>(v, red + wave * L::NACC, lane);
#endif
    __syncthreads();
    // A single workgroup can only pull ~6 GB/s of freshly written records (measured: 110 KB = 480 records in 19 us), so the
    // solve kernel must not read one record per workgroup.  Groups of RG consecutive workgroups reduce themselves: every
    // workgroup publishes its record, takes a ticket, and the LAST arriver of the group sums the group's records in index
    // order (fixed order => bit-reproducible, no float atomics) into one group record.
    // Protocol (cdna_hip_programming.md Guideline 16, counter form): write-through (sc1) record stores -> every storing
    // wave drains vmcnt -> workgroup barrier -> one relaxed agent-scope ticket; reducer: agent-scope acquire -> drain ->
    // barrier -> plain loads.  The reducer zeroes the ticket for the next launch (tickets are also zeroed at create
    // and after a failed call).
    float *myrec = P.blockrec + ((size_t)n * nblk + bid) * L::NACC;
    for (int i = tid; i < L::NACC; i += NT) {
        // accumulator index -> compacted live index (or -1 for a dead accumulator, stored as 0)
        int li = -1;
        if (i >= L::OFF_S) li = NLIVE - 3 + (i - L::OFF_S);
        else if (LIN && (DC || i < NPH)) li = i;
        float s = 0.f;
        if (li >= 0)
            for (int w = 0; w < NT / 64; w++) s += red[w * L::NACC + li];
        if (P.direct) myrec[i] = s;   // plain store: the kernel boundary publishes it
        else __hip_atomic_store(&myrec[i], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.direct) return;             // wave-uniform: k_solve reads one record per workgroup (a few hundred at most)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    const int grp = bid / RG, gfirst = grp * RG, gcount = min(RG, nblk - gfirst);
    if (tid == 0) {
        int t = __hip_atomic_fetch_add(&P.tickets[n * P.ngrp + grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gcount - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float *grec = P.blockrec + ((size_t)n * nblk + gfirst) * L::NACC;
    for (int i = tid; i < L::NACC; i += NT) {
        float w[RG];
#pragma unroll
        for (int b = 0; b < RG; b++) w[b] = grec[(size_t)(b < gcount ? b : 0) * L::NACC + i];  // RG UNCONDITIONAL loads in flight
        float s = 0.f;                                         // (a predicated load makes hipcc branch + vmcnt(0) per element)
#pragma unroll
        for (int b = 0; b < RG; b++) s += (b < gcount) ? w[b] : 0.f;   // fixed-order sum
        P.partials[((size_t)n * P.ngrp + grp) * L::NACC + i] = s;
    }
    if (tid == 0) P.tickets[n * P.ngrp + grp] = 0;
}

// TRACE: the parity-test build of the kernel that records its discrete decisions (tcsfm_debug_trace); a template parameter so that
// the production instantiation carries none of it (not even the branches: the kernel sits at its 128-VGPR budget)
// ADJ: the ADJOINT form of the 3x3-coupled SSIM gradient (round 4; the dense kernels have used it since round 2).  Instead of every
// residual pixel p visiting the full record (colours, image gradients, geometric Jacobian: 96 bytes) of its 9 window pixels and
// forming 9 x (2 x 6) products (pass B below: ~20 packed instructions per neighbour), p leaves its nine SSIM coefficients
// m W (cA, cB, cC)_c in LDS; every position q of the tile AND its ring then sums the coefficient records of the residual pixels that
// see it (9 x 5 additions), forms d C / d(ix, iy) at q with its OWN colours and image gradients and applies its OWN Jacobian once.
// The gradient is the same sum in another order (exact; H is untouched).  The coefficient records alias the colour part of the
// staged records (dead after pass A once every thread holds its own colours in registers), so LDS stays at 68.5 KB; the price is
// two more workgroup barriers and the window sums of the image gradients (curvature model) moving into pass A (+3 per neighbour).
// Measurement hook (scripts/passa_probe.sh; appendix R4): a build with -DTC_PROBE_PASSA_VISITS=5 visits 5 of the 9 window positions in
// pass A -- WRONG statistics, timing only -- which removes more pass-A work (48 VALU + 8 LDS reads per wave, no extra barrier, no
// extra LDS) than an exactly separable form could: a measured upper bound on what that form can buy.  Production builds: 9.
#ifndef TC_PROBE_PASSA_VISITS
#define TC_PROBE_PASSA_VISITS 9
#endif
// Software-pipelined neighbour passes (round 4, second session): 1 = production; 0 = the rolled loops they replace (A/B builds).
// TC_PASSB_J2 = 1 double-buffers the Jacobian parts of pass B as well: measured no faster (profiles/r04_lds_pipeline_ab.txt), default off.
#ifndef TC_PASSA_PIPELINED
#define TC_PASSA_PIPELINED 1
#endif
#ifndef TC_PASSB_PIPELINED
#define TC_PASSB_PIPELINED 1
#endif
#ifndef TC_PASSB_J2
#define TC_PASSB_J2 0
#endif
// Tile-constant colour shift of the staged records (round 5; see ssim_l1_channel_ts): 1 = production candidate, 0 = centre-shifted sums as in round 4
#ifndef TC_TILE_SHIFT
#define TC_TILE_SHIFT 0
#endif
// TSH: the shared-pack form (LinParams::tshare) folded at compile time -- the production instantiations; with the flag read at run time the
// two sides of its branches cost the chip-filling launch 5 % (profiles/r05_shared_pack_ab.txt).  The TRACE / ADJ builds (tests, A/B) keep the
// run-time flag; the host launches <TSH = true> exactly when P.tshare is set and neither TRACE nor ADJ is (launch_lin_a).
template <int NP, bool DC, int MODE, int TW, int TH, int NT, bool SEL = false, bool TRACE = false, bool ADJ = false, bool FRONT = false, bool TSH = false>
__global__ __launch_bounds__(NT, 4) void k_linearize(LinParams P) {
    constexpr int CW = TW + 2, CH = TH + 2, NCOMP = CW * CH, NCEN = TW * TH;
    constexpr int PPT = (NCEN + NT - 1) / NT;  // centre pixels per thread
    static_assert(NCEN % NT == 0, "tile must be a multiple of the workgroup");
    static_assert(!FRONT || (MODE == MODE_LIN && !ADJ && NP == 6), "FRONT: the 6-DoF linearisation in its production form");
    using L = AccLayout<NP>;
    __shared__ float4 lds[NCOMP * (LDS_REC / 4)];
    __shared__ float red[(NT / 64) * L::NACC];
    __shared__ unsigned adj_any;                              // ADJ: bit w = wave w has a pixel that counts
    __shared__ int front_cnt, front_org[2];                   // FRONT: the tile's mask count; origin of the scatter window
    constexpr bool ADJL = ADJ && MODE == MODE_LIN;
    if (ADJL && threadIdx.x == 0) adj_any = 0u;              // (ordered before its first use by the phase-1 barrier)
    if (FRONT && threadIdx.x == 0) front_cnt = 0;

    // XCD-aware tile order: consecutive workgroups land on different XCDs (round-robin dispatch), so give each
    // of the 8 XCDs a contiguous band of tiles -> halo / source-texel reuse stays inside one XCD's L2.
    const int nblk = P.tiles_x * P.tiles_y;
    int bid = blockIdx.x;
    {
        int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // FRONT: rows [0, S B) are the INVERSE pairs (the heavier role is dispatched first), the rows behind them the forward pairs -- all S B of
    // them, or under SEL one row per TARGET (pair (0, b)), which decides the selection for all S sources of its target
    // (tshare: rows 2k and 2k + 1 are forward pair k and its inverse partner -- the two read the same two image packs, and rows that are
    // dispatched one after the other find each other's lines in L2; with the pairs in index order a merged launch runs them S B rows apart)
    const int n = FRONT ? ((int)blockIdx.y < P.front_fwd ? P.front_fwd + (int)blockIdx.y : (int)blockIdx.y - P.front_fwd)
                        : (P.tshare ? (((int)blockIdx.y & 1) ? P.tshare_sb + ((int)blockIdx.y >> 1) : ((int)blockIdx.y >> 1)) : (int)blockIdx.y);
    const PairConst &c = P.pc[n];
    const int H = P.H, W = P.W, hw = H * W;
    const int tyi = bid / P.tiles_x, txi = bid - tyi * P.tiles_x;
    const int x00 = txi * TW, y00 = tyi * TH;
    const int img = P.shared_image ? 0 : n;
    const bool ffwd = FRONT && n < P.front_fwd;                              // FRONT: a forward pair -- mask and count only (workgroup-uniform)
    const bool flight = FRONT && (ffwd || P.front_light != 0);               // FRONT: a row without a linearisation
    bool front_m = false;                                                    // FRONT: this thread's pixel counts
    int f_tap = 0;                                                           // FRONT, inverse pairs: top-left tap (x + 1) | (y + 1) << 16 of the own pixel's
    float f_wx = 0.f, f_wy = 0.f, f_dc = 0.f, f_ph = 0.f;                    // sample, its bilinear weights; scatter coefficients h(dd) ddd and M diff ddd
    const bool tsh = TSH || ((TRACE || ADJ) && P.tshare != 0);               // wave-uniform (see LinParams); a compile-time constant in the production builds
    const int tsh_part = n < P.tshare_sb ? n + P.tshare_sb : n - P.tshare_sb, tsh_ae = n < P.tshare_sb ? n : n - P.tshare_sb;
    const float4 *tgtpack = tsh ? P.srcpack + (size_t)tsh_part * (H + 2) * (W + 2) : P.tgtpack + (size_t)img * hw;
    constexpr bool TS = TC_TILE_SHIFT && TC_PASSA_PIPELINED && TC_PROBE_PASSA_VISITS == 9 && !(ADJ && MODE == MODE_LIN);
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f;                                   // TS: the workgroup's colour shift = the target colour at its first pixel
    if (TS) {                                                                // (a workgroup-uniform address: scalar load)
        const float4 c0_ = tgtpack[(size_t)min(y00, H - 1) * W + min(x00, W - 1)];
        cs0 = c0_.x; cs1 = c0_.y; cs2 = c0_.z;
    }
    const bool cached = P.pair_src != nullptr;                               // wave-uniform
    const float4 *srcpack = P.srcpack + (size_t)(cached ? P.pair_src[n] : img) * (H + 2) * (W + 2);   // zero-bordered (tap4)
    const float *depth_t = P.depth_t + (size_t)(tsh ? tsh_ae : (cached ? P.pair_dep[n] : img)) * hw;      // (tshare: the couple's auto-mask error plane)
    const int tid = threadIdx.x;
    stamp_begin(P.stamp, tid);

    // centre-only values carried in registers from phase 1 to phase 2
    float c_pd[PPT], c_cd[PPT], c_dif[PPT], c_dgx[PPT], c_dgy[PPT], c_ae[PPT], c_zc[PPT][NP];
    bool c_valid[PPT], c_in[PPT], c_dcin[PPT];

    // ---------------- window mode: the residual of the OTHER sources at this tile (min over sources, optimizer.py:47-69) -------
    // For a forward pair n = s B + b the other sources of target b are warped with THEIR poses (and their copy of the target
    // depth), colours only, and their photometric error at the tile's pixels is reduced to four numbers per pixel: the smallest
    // error among the sources before / after s (torch.min keeps the FIRST minimum), the union of their validity and the
    // smallest auto-mask threshold.  Same arithmetic as the maps pass + selection it replaces (the dense modes keep the maps pass: ext_selected).
    float sel_before = 3.0e38f, sel_after = 3.0e38f, sel_valid = 0.f, sel_ae = 3.0e38f;
    float sel_d1 = 3.0e38f, sel_d2 = 3.0e38f;      // FRONT: the errors of sources 1 and 2 on their own (the row of pair (0, b) decides for every source)
    float sel_w0 = 1.f;      // REFERENCE rule: depth-consistency weight of SOURCE 0 at this pixel (pairs of the other sources)
    const bool sel_pair = SEL && n < P.sel_B * P.sel_S;
    const int s_own = (SEL && sel_pair) ? n / P.sel_B : 0;
    if (SEL && sel_pair) {
        const int b_ = n % P.sel_B;
        for (int so = 0; so < P.sel_S; so++) {
            if (so == s_own) continue;
            const int no = so * P.sel_B + b_;
            const PairConst &co = P.pc[no];
            // (tshare: pair `no` is a forward pair here -- its target pack is its inverse partner's source pack, its error plane its own)
            const float4 *tpo = tsh ? P.srcpack + (size_t)(no + P.tshare_sb) * (H + 2) * (W + 2) : P.tgtpack + (size_t)no * hw;
            const float4 *spo = P.srcpack + (size_t)(cached ? P.pair_src[no] : no) * (H + 2) * (W + 2);
            const float *dto = P.depth_t + (size_t)(cached ? P.pair_dep[no] : no) * hw;
            for (int ci = tid; ci < NCOMP; ci += NT) {       // tile + ring, colours only (+ validity and auto-mask threshold)
                const int ly = ci / CW, lx = ci - ly * CW;
                const int px = refl_idx(x00 + lx - 1, W), py = refl_idx(y00 + ly - 1, H), gi = py * W + px;
                float4 tp;
                float dpo;
                if (tsh) { tp = tpo[(py + 1) * (W + 2) + px + 1]; dpo = tp.w; tp.w = dto[gi]; }
                else { tp = tpo[gi]; dpo = dto[gi]; }
                Geo g;
                warp_geo(co, W, H, px, py, dpo, g);
                float4 val, gx, gy;
                Tap to;
                tap4_fetch(spo, W, H, px, py, g.rx, g.ry, g.oobx || g.ooby, to);
                tap4_lerp(to, val, gx, gy);
                if (FRONT && P.front_light && lx >= 1 && lx <= TW && ly >= 1 && ly <= TH && x00 + lx - 1 < W && y00 + ly - 1 < H && !(g.oobx || g.ooby)) {
                    // free source maps: forward pair (so, b) samples source map (so, b) here; under the min over the sources only source 0's
                    // weight multiplies the photometric term, so this pair's sample enters through its depth-consistency term alone
                    const float pdo = co.es * val.w, cdo = g.Z, iso = frcp(cdo + pdo), dfo = cdo - pdo, rwo = fabsf(dfo) * iso;
                    if (rwo >= 0.f && rwo <= 1.f) {
                        const float sgo = dfo > 0.f ? 1.f : (dfo < 0.f ? -1.f : 0.f);
                        const float fdc = fminf(1.f, fminf(rwo, 1.f) * frcp(P.eps)) * (-sgo * 2.f * cdo * iso * iso * co.es);
                        const int xi = px + (int)floorf(g.rx), yi = py + (int)floorf(g.ry);
                        const float w4[4] = {(1.f - to.wx) * (1.f - to.wy), to.wx * (1.f - to.wy), (1.f - to.wx) * to.wy, to.wx * to.wy};
                        long long *es_ = P.ext2_src + (size_t)no * hw * 2;
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const int xx = xi + (k4 & 1), yy = yi + (k4 >> 1);
                            if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                                const long long a0 = (long long)llrint((double)(fdc * w4[k4]) * DREF_FIX);
                                if (a0 != 0) atomicAdd(reinterpret_cast<unsigned long long *>(es_ + ((size_t)yy * W + xx) * 2), (unsigned long long)a0);
                            }
                        }
                    }
                }
                float4 *rec = lds + ci * (LDS_REC / 4);
                lds_write1(rec + 0, val.x - cs0, val.y - cs1, tp.x - cs0, tp.y - cs1);
                lds_write1(rec + 2, val.z - cs2, tp.z - cs2, (g.oobx || g.ooby) ? 0.f : 1.f, tp.w);
                if (P.rule && so == 0) {   // that source's depth-consistency weight (train_mono.py:91-92), with its depth sample
                    const float pdo = co.es * val.w;
                    lds_write1(rec + 1, 1.f - clamp01(fabsf(g.Z - pdo) * frcp(g.Z + pdo)), 0.f, 0.f, 0.f);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            {
                const int cy = tid / TW + 1, cx = tid - (tid / TW) * TW + 1;
                const float4 *ctr = lds + (cy * CW + cx) * (LDS_REC / 4);
                f32x4 q0, q2;
                lds_read02v(ctr, q0, q2);
                const f2 yc01 = q0.lo, xc01 = q0.hi, yx2c = q2.lo;
                const float4 *nb = ctr - (CW + 1) * (LDS_REC / 4);
                f2 Sy01 = {0.f, 0.f}, Sx01 = {0.f, 0.f}, Syy01 = {0.f, 0.f}, Sxx01 = {0.f, 0.f}, Sxy01 = {0.f, 0.f}, S2 = {0.f, 0.f}, SS2 = {0.f, 0.f};
                float Sxy2 = 0.f;
#if TC_PASSA_PIPELINED
                {   // software-pipelined like pass A of phase 2 (same operations in the same order)
                    constexpr int RB = LDS_REC * 4, ROWB = CW * LDS_REC * 4;
                    const unsigned base = lds_addr(nb);
                    f32x4 u0, u2, w0, w2;
                    auto more = [&](const f32x4 &n0, const f32x4 &n2) {
                        f2 ey, ex, e2v;
                        if (TS) { ey = n0.lo; ex = n0.hi; e2v = n2.lo; }
                        else { ey = pk_sub(n0.lo, yc01); ex = pk_sub(n0.hi, xc01); e2v = pk_sub(n2.lo, yx2c); }
                        Sy01 += ey; Sx01 += ex; Syy01 += ey * ey; Sxx01 += ex * ex; Sxy01 += ex * ey;
                        S2 += e2v; SS2 += e2v * e2v; Sxy2 += e2v.x * e2v.y;
                    };
                    lds_issue02v_at<0>(base, u0, u2);
                    lds_issue02v_at<RB>(base, w0, w2);
                    lds_wait2(u0, u2); more(u0, u2); lds_issue02v_at<2 * RB>(base, u0, u2);
                    lds_wait2(w0, w2); more(w0, w2); lds_issue02v_at<ROWB>(base, w0, w2);
                    lds_wait2(u0, u2); more(u0, u2); lds_issue02v_at<ROWB + RB>(base, u0, u2);
                    lds_wait2(w0, w2); more(w0, w2); lds_issue02v_at<ROWB + 2 * RB>(base, w0, w2);
                    lds_wait2(u0, u2); more(u0, u2); lds_issue02v_at<2 * ROWB>(base, u0, u2);
                    lds_wait2(w0, w2); more(w0, w2); lds_issue02v_at<2 * ROWB + RB>(base, w0, w2);
                    lds_wait2(u0, u2); more(u0, u2); lds_issue02v_at<2 * ROWB + 2 * RB>(base, u0, u2);
                    lds_wait2(w0, w2); more(w0, w2);
                    lds_wait0(u0, u2); more(u0, u2);
                }
#else
#pragma unroll 1
                for (int kk = 0; kk < 9; kk++) {
                    f32x4 n0, n2;
                    lds_read02v(nb, n0, n2);
                    nb += (kk == 2 || kk == 5) ? (CW - 2) * (LDS_REC / 4) : (LDS_REC / 4);
                    f2 ey = pk_sub(n0.lo, yc01), ex = pk_sub(n0.hi, xc01), e2v = pk_sub(n2.lo, yx2c);
                    Sy01 += ey; Sx01 += ex; Syy01 += ey * ey; Sxx01 += ex * ex; Sxy01 += ex * ey;
                    S2 += e2v; SS2 += e2v * e2v; Sxy2 += e2v.x * e2v.y;
                }
#endif
                const f2 e01 = TS ? ssim_l1_value_ts<f2>(xc01, yc01, f2{cs0, cs1}, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl)
                                  : ssim_l1_value<f2>(xc01, yc01, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl);
                const float d_o = e01.x + e01.y + (TS ? ssim_l1_value_ts<float>(yx2c.y, yx2c.x, cs2, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl)
                                                      : ssim_l1_value<float>(yx2c.y, yx2c.x, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl));
                if (so < s_own) sel_before = fminf(sel_before, d_o); else sel_after = fminf(sel_after, d_o);
                if (FRONT) { if (so == 1) sel_d1 = d_o; else sel_d2 = d_o; }
                sel_valid = fmaxf(sel_valid, q2.z);
                sel_ae = fminf(sel_ae, q2.w);
                if (P.rule && so == 0) sel_w0 = lds_read1(ctr + 1).x;
            }
            __syncthreads();   // the records are overwritten by the next source / by phase 1
        }
    }

    // ---------------- phase 1: warp + stage (centres, and the halo ring on the first waves) ----------------
    // The NHALO ring pixels are a second pixel for the first NHALO threads.  Those waves run BOTH pixels as one software-
    // pipelined sequence -- loads of both, then both warps and gather issues, then both interpolations -- so the dependent
    // load chain (pixel data -> warp -> source gather) of the ring pixel overlaps the centre pixel's instead of following it
    // while the rest of the workgroup waits at the barrier.
    constexpr int NHALO = NCOMP - NCEN;
    static_assert(PPT == 1 && NHALO <= NT, "one centre pixel per thread, one ring round");
    constexpr int HALO_THREADS = (NHALO + 63) / 64 * 64;   // wave-uniform split
    struct Stage { int lx, ly, px, py; float4 tp; float dep; Geo g; Tap t; };
    auto s_load = [&](Stage &S) {
        S.px = refl_idx(x00 + S.lx - 1, W); S.py = refl_idx(y00 + S.ly - 1, H);
        const unsigned gi = (unsigned)(S.py * W + S.px);          // 32-bit offsets from the wave-uniform bases (see tap4_fetch)
        if (tsh) {      // target colours + depth from the partner's bordered pack, the auto-mask error from the couple's plane
            const unsigned gb = (unsigned)((S.py + 1) * (W + 2) + S.px + 1);
            S.tp = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tgtpack) + (gb << 4));
            S.dep = S.tp.w;
            S.tp.w = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(depth_t) + (gi << 2));
        } else {
            S.tp = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tgtpack) + (gi << 4));
            S.dep = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(depth_t) + (gi << 2));
        }
    };
    auto s_warp = [&](Stage &S) {
        warp_geo(c, W, H, S.px, S.py, S.dep, S.g);
        tap4_fetch(srcpack, W, H, S.px, S.py, S.g.rx, S.g.ry, S.g.oobx || S.g.ooby, S.t);
    };
    auto s_store = [&](Stage &S, bool write, bool centre) {
        float4 val, gx, gy;
        tap4_lerp(S.t, val, gx, gy);
        float a[NP], b[NP], zc[NP];
        if (FRONT && flight) {
#pragma unroll
            for (int j = 0; j < NP; j++) { a[j] = 0.f; b[j] = 0.f; zc[j] = 0.f; }
        } else if (MODE == MODE_LIN) geo_jac<NP>(c, S.g, W, H, a, b, zc);   // cost / maps passes need neither Jacobians nor image gradients
        float4 *rec = lds + (S.ly * CW + S.lx) * (LDS_REC / 4);
        // record: [y0 y1 x0 x1][gx0 gy0 gx1 gy1][y2 x2 gx2 gy2][a0 b1 a2 a3][a4 a5 b2 b3][b4 b5 (a6 b6)] : channel pairs and
        // Jacobian pairs sit on aligned register pairs after ds_read_b128, so phase 2 runs on v_pk_*_f32 without shuffles.
        // a1 = d ix / d rho_y and b0 = d iy / d rho_x are STRUCTURALLY zero (pinhole K: a translation along y moves the sample along
        // iy only, see geo_jac), so they are not staged: (a0, b1) travel as one pair -- pass B spends one packed FMA on columns (0,1)
        // instead of two -- and the seventh column of the pose + depth-scale mode takes their place (six 16-byte rows for NP = 7 too)
        if (write) {
            lds_write1(rec + 0, val.x - cs0, val.y - cs1, S.tp.x - cs0, S.tp.y - cs1);      // (cs = 0 without the tile shift)
            if (MODE == MODE_LIN) {
                lds_write1(rec + 1, gx.x, gy.x, gx.y, gy.y);     // (gx, gy) pairs per channel: pass B forms (sx, sy) with packed FMAs
                lds_write1(rec + 2, val.z - cs2, S.tp.z - cs2, gx.z, gy.z);
                if (!(FRONT && flight)) {
                    lds_write1(rec + 3, a[0], b[1], a[2], a[3]);
                    lds_write1(rec + 4, a[4], a[5], b[2], b[3]);
                    lds_write1(rec + 5, b[4], b[5], NP == 7 ? a[NP - 1] : 0.f, NP == 7 ? b[NP - 1] : 0.f);
                }
            } else {
                lds_write1(rec + 2, val.z - cs2, S.tp.z - cs2, 0.f, 0.f);
            }
        }
        if (centre) {
            c_in[0] = (x00 + S.lx - 1 < W) && (y00 + S.ly - 1 < H);
            if (FRONT) {
                f_tap = ((S.px + (int)floorf(S.g.rx) + 1) & 0xffff) | ((S.py + (int)floorf(S.g.ry) + 1) << 16);
                f_wx = S.t.wx; f_wy = S.t.wy;
            }
            if (TRACE && MODE != MODE_MAPS && P.trace != nullptr && c_in[0] && !flight)    // bilinear cell parity now, mask / validity bits in phase 2
                P.trace[(size_t)n * hw + (size_t)S.py * W + S.px] =
                    (unsigned short)((((S.px + (int)floorf(S.g.rx)) & 1) << 2) | (((S.py + (int)floorf(S.g.ry)) & 1) << 3));
            c_pd[0] = c.es * val.w; c_dgx[0] = c.es * gx.w; c_dgy[0] = c.es * gy.w; c_cd[0] = S.g.Z;
            c_dif[0] = dc_diff(c, S.g, S.t, S.dep, c_pd[0]);
            c_dcin[0] = S.t.inside;
            c_ae[0] = S.tp.w; c_valid[0] = !(S.g.oobx || S.g.ooby);
            if (MODE == MODE_LIN) {
#pragma unroll
                for (int j = 0; j < NP; j++) c_zc[0][j] = zc[j];
            }
        }
    };
    {
        Stage A;
        A.ly = tid / TW + 1; A.lx = tid - (tid / TW) * TW + 1;
        if (tid < HALO_THREADS) {
            if (P.one_generation) __builtin_amdgcn_s_setprio(3);     // (wave-uniform; see LinParams)
            Stage B;
            const int hi = min(tid, NHALO - 1);   // ring enumeration: top row, bottom row, then left/right columns
            if (hi < CW) { B.ly = 0; B.lx = hi; }
            else if (hi < 2 * CW) { B.ly = CH - 1; B.lx = hi - CW; }
            else { const int k = hi - 2 * CW; B.ly = 1 + (k >> 1); B.lx = (k & 1) ? CW - 1 : 0; }
            s_load(A); s_load(B);
            s_warp(A); s_warp(B);
            s_store(A, true, true); s_store(B, tid < NHALO, false);
            if (P.one_generation) __builtin_amdgcn_s_setprio(0);
        } else {
            s_load(A); s_warp(A); s_store(A, true, true);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm LDS writes above are invisible to hipcc's own waitcnt tracking
    __syncthreads();

    // ---------------- phase 2: residuals, gradient rows, curvature, accumulation ----------------
    // Packed fp32 throughout: channels (0,1) travel as one f2, channel 2 shares an f2 with its partner quantity, Jacobian
    // columns travel as pairs (01)(23)(45).  H is accumulated as 12 (+4 for NP=7) row-pair f2's: rows j, column pairs p <= j/2.
    constexpr int NHP = (NP == 6) ? 12 : 16;     // f2 accumulators of the lower triangle (some upper entries ride along)
    f2 aH2[NHP], aG2[3], dH2[DC ? NHP : 1], dG2[DC ? 3 : 1];
    float aH66 = 0.f, aG6 = 0.f, dH66 = 0.f, dG6 = 0.f;   // NP == 7: the (6,6) entry and g[6]
    float sMWd = 0.f, sM = 0.f, sdd = 0.f;
#pragma unroll
    for (int i = 0; i < NHP; i++) { aH2[i] = (f2){0.f, 0.f}; if (DC) dH2[i] = (f2){0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 3; i++) { aG2[i] = (f2){0.f, 0.f}; if (DC) dG2[i] = (f2){0.f, 0.f}; }

#pragma unroll
    for (int k = 0; k < PPT; k++) {
        int ci = tid + k * NT;
        int ly = ci / TW + 1, lx = ci - (ci / TW) * TW + 1;
        const float4 *ctr = lds + (ly * CW + lx) * (LDS_REC / 4);
        f32x4 q0, q1 = {0.f, 0.f, 0.f, 0.f}, q2;
        if (MODE == MODE_LIN) lds_read3v(ctr, q0, q1, q2);
        else lds_read02v(ctr, q0, q2);               // the gradient part of the records is only staged for linearisations
        const f2 yc01 = q0.lo, xc01 = q0.hi, gxc01 = {q1.x, q1.z}, gyc01 = {q1.y, q1.w}, yx2c = q2.lo, g2c = q2.hi;   // (channel pairs for the per-channel terms)
        const float yc[3] = {yc01.x, yc01.y, yx2c.x}, xc[3] = {xc01.x, xc01.y, yx2c.y};
        const float gxc[3] = {gxc01.x, gxc01.y, g2c.x}, gyc[3] = {gyc01.x, gyc01.y, g2c.y};

        // pass A: SSIM statistics over the 3x3 window, shifted by the centre value (fp32-safe variances); 11 VALU per neighbour.
        // The first neighbour INITIALISES the accumulators (no zero-fill instructions), the other eight are added.  Only the colour
        // part of the records is read here (32 of their 96 bytes); the gradient window sums the curvature needs are gathered in
        // pass B, which reads the gradients anyway.  Production form: software-pipelined (below); the rolled loop (one neighbour
        // live at a time, the pointer walking the window with one vector add per step) remains for the adjoint form and the A/B builds.
        const float4 *nbA = ctr - (CW + 1) * (LDS_REC / 4);
        f2 Sy01, Sx01, Syy01, Sxx01, Sxy01, S2, SS2;
        float Sxy2;
        f2 aGx01 = {0.f, 0.f}, aGy01 = {0.f, 0.f}, aG2s = {0.f, 0.f};   // ADJ: 3x3 sums of the image gradients (curvature model), taken here
#if TC_PASSA_PIPELINED
        if (!ADJL && TC_PROBE_PASSA_VISITS == 9) {
            // Software-pipelined form: the reads of window position k + 1 are in flight while position k is accumulated (two register
            // sets, counted waits: LDS returns in order), and every position is a compile-time offset from the window's first record,
            // so the walk costs no address arithmetic.  Same operations on the same values in the same order as the rolled loop below.
            constexpr int RB = LDS_REC * 4, ROWB = CW * LDS_REC * 4;      // bytes per record / per row of records
            const unsigned base = lds_addr(nbA);
            f32x4 u0, u2, w0, w2;
            auto first = [&](const f32x4 &n0, const f32x4 &n2) {
                if (TS) { Sy01 = n0.lo; Sx01 = n0.hi; S2 = n2.lo; }       // tile-shifted records: the window values as they are
                else { Sy01 = pk_sub(n0.lo, yc01); Sx01 = pk_sub(n0.hi, xc01); S2 = pk_sub(n2.lo, yx2c); }            // (y2 - y2c, x2 - x2c)
                Syy01 = Sy01 * Sy01; Sxx01 = Sx01 * Sx01; Sxy01 = Sx01 * Sy01;
                SS2 = S2 * S2; Sxy2 = S2.x * S2.y;
            };
            auto more = [&](const f32x4 &n0, const f32x4 &n2) {
                f2 ey, ex, e2v;
                if (TS) { ey = n0.lo; ex = n0.hi; e2v = n2.lo; }
                else { ey = pk_sub(n0.lo, yc01); ex = pk_sub(n0.hi, xc01); e2v = pk_sub(n2.lo, yx2c); }
                Sy01 += ey; Sx01 += ex; Syy01 += ey * ey; Sxx01 += ex * ex; Sxy01 += ex * ey;
                S2 += e2v; SS2 += e2v * e2v; Sxy2 += e2v.x * e2v.y;
            };
            lds_issue02v_at<0>(base, u0, u2);
            lds_issue02v_at<RB>(base, w0, w2);
            lds_wait2(u0, u2); first(u0, u2);
            lds_issue02v_at<2 * RB>(base, u0, u2);
            lds_wait2(w0, w2); more(w0, w2);
            lds_issue02v_at<ROWB>(base, w0, w2);
            lds_wait2(u0, u2); more(u0, u2);
            lds_issue02v_at<ROWB + RB>(base, u0, u2);
            lds_wait2(w0, w2); more(w0, w2);
            lds_issue02v_at<ROWB + 2 * RB>(base, w0, w2);
            lds_wait2(u0, u2); more(u0, u2);
            lds_issue02v_at<2 * ROWB>(base, u0, u2);
            lds_wait2(w0, w2); more(w0, w2);
            lds_issue02v_at<2 * ROWB + RB>(base, w0, w2);
            lds_wait2(u0, u2); more(u0, u2);
            lds_issue02v_at<2 * ROWB + 2 * RB>(base, u0, u2);
            lds_wait2(w0, w2); more(w0, w2);
            lds_wait0(u0, u2); more(u0, u2);
        } else
#endif
        {
        {
            f32x4 n0, n1 = {0.f, 0.f, 0.f, 0.f}, n2;
            if (ADJL) lds_read3v(nbA, n0, n1, n2); else lds_read02v(nbA, n0, n2);
            nbA += LDS_REC / 4;
            Sy01 = pk_sub(n0.lo, yc01); Sx01 = pk_sub(n0.hi, xc01);
            Syy01 = Sy01 * Sy01; Sxx01 = Sx01 * Sx01; Sxy01 = Sx01 * Sy01;
            S2 = pk_sub(n2.lo, yx2c);            // (y2 - y2c, x2 - x2c)
            SS2 = S2 * S2; Sxy2 = S2.x * S2.y;
            if (ADJL) { aGx01 = n1.lo; aGy01 = n1.hi; aG2s = n2.hi; }
        }
#pragma unroll 1
        for (int kk = 1; kk < TC_PROBE_PASSA_VISITS; kk++) {
            f32x4 n0, n1 = {0.f, 0.f, 0.f, 0.f}, n2;
            if (ADJL) lds_read3v(nbA, n0, n1, n2); else lds_read02v(nbA, n0, n2);
            nbA += (kk == 2 || kk == 5) ? (CW - 2) * (LDS_REC / 4) : (LDS_REC / 4);
            f2 ey = pk_sub(n0.lo, yc01), ex = pk_sub(n0.hi, xc01);
            Sy01 += ey; Sx01 += ex; Syy01 += ey * ey; Sxx01 += ex * ex; Sxy01 += ex * ey;
            f2 e2v = pk_sub(n2.lo, yx2c);
            S2 += e2v; SS2 += e2v * e2v; Sxy2 += e2v.x * e2v.y;
            if (ADJL) { aGx01 += n1.lo; aGy01 += n1.hi; aG2s += n2.hi; }
        }
        }
        // per-channel SSIM value / gradient coefficients / curvature weights and the L1 term: channels (0,1) as one packed
        // evaluation, channel 2 as a scalar one (same code, ssim_l1_channel<T>)
        ChanTerms<f2> t01;
        ChanTerms<float> t2;
        if (TS) {
            ssim_l1_channel_ts<f2>(xc01, yc01, f2{cs0, cs1}, gxc01, gyc01, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl, P.eps, t01);
            ssim_l1_channel_ts<float>(yx2c.y, yx2c.x, cs2, g2c.x, g2c.y, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl, P.eps, t2);
        } else {
            ssim_l1_channel<f2>(xc01, yc01, gxc01, gyc01, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl, P.eps, t01);
            ssim_l1_channel<float>(yx2c.y, yx2c.x, g2c.x, g2c.y, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl, P.eps, t2);
        }
        const float cA[3] = {t01.cA.x, t01.cA.y, t2.cA}, cB[3] = {t01.cB.x, t01.cB.y, t2.cB}, cC[3] = {t01.cC.x, t01.cC.y, t2.cC};
        const float e1 = t01.e1.x + t01.e1.y + t2.e1, e2 = t01.e2.x + t01.e2.y + t2.e2;
        const float l1x = t01.l1x.x + t01.l1x.y + t2.l1x, l1y = t01.l1y.x + t01.l1y.y + t2.l1y;
        float lxx = t01.lxx.x + t01.lxx.y + t2.lxx, lxy = t01.lxy.x + t01.lxy.y + t2.lxy, lyy = t01.lyy.x + t01.lyy.y + t2.lyy;
        float diff = e1 + e2;

        // the pixel's mask, known before the expensive part: valid x auto-mask, or the min-over-sources selection
        const bool inimg = c_in[k];
        bool m = inimg && c_valid[k] && (!(P.automask && n >= P.fwd_noauto) || diff < c_ae[k]);
        bool sel_keep = false;
        float sel_dothers = 0.f;
        if (SEL && sel_pair) {   // keep the pixel for the source with the smallest error (first minimum), under the union
                                 // validity and the auto-mask of the minima
            sel_dothers = fminf(sel_before, sel_after);
            const float dmin = fminf(diff, sel_dothers);
            sel_keep = (c_valid[k] || sel_valid > 0.f) && (!P.automask || dmin < fminf(c_ae[k], sel_ae));
            m = inimg && sel_keep && (diff < sel_before) && (diff <= sel_after);
        }

        if (FRONT) front_m = m;
        if (FRONT && P.front_light) {      // free source maps: the scatter coefficients of this row's own sample (plain fp32 sign, as the joint kernel's)
            const float cd_ = c_cd[k], pd_ = c_pd[k], is_ = frcp(cd_ + pd_), df_ = cd_ - pd_, rw_ = fabsf(df_) * is_;
            if (inimg && c_valid[k] && rw_ >= 0.f && rw_ <= 1.f) {
                const float sg_ = df_ > 0.f ? 1.f : (df_ < 0.f ? -1.f : 0.f);
                const float ddd_ = -sg_ * 2.f * cd_ * is_ * is_ * c.es;                  // d dd / d (sampled depth)
                f_dc = fminf(1.f, fminf(rw_, 1.f) * frcp(P.eps)) * ddd_;
                // photometric part: an inverse pair's own masked error; a forward pair's E = the error this sample's weight multiplies -- under
                // the min over the sources (this row decides for all of them) the winning source's error wherever a source is selected
                float E_ = m ? diff : 0.f;
                if (SEL && ffwd) E_ = (inimg && sel_keep) ? fminf(diff, fminf(sel_before, sel_after)) : 0.f;
                f_ph = E_ * ddd_;
            }
        }
        if (FRONT && flight) {     // a row without a linearisation: its mask (the selection), the count and the scatter below
            if (SEL && ffwd) {     // the row of pair (0, b): the decision for EVERY source of target b (first minimum, as torch.min / ext_selected)
                front_m = inimg && sel_keep;                  // exactly one source keeps the pixel
                if (inimg) {
                    const size_t o = (size_t)(y00 + ly - 1) * W + (x00 + lx - 1);
                    const bool m1 = sel_keep && sel_d1 < diff && sel_d1 <= sel_d2, m2 = sel_keep && sel_d2 < diff && sel_d2 < sel_d1;
                    P.sel_out[(size_t)n * hw + o] = m ? 1.f : 0.f;
                    P.sel_out[(size_t)(P.sel_B + n) * hw + o] = m1 ? 1.f : 0.f;
                    if (P.sel_S > 2) P.sel_out[(size_t)(2 * P.sel_B + n) * hw + o] = m2 ? 1.f : 0.f;
                }
            }
            continue;
        }

        // depth consistency, train_mono.py:91-92
        float cd = c_cd[k], pd = c_pd[k];
        float sum = cd + pd, dif = c_dif[k], isum = frcp(sum);
        float raw = fabsf(dif) * isum;
        float dd = clamp01(raw), Wt = 1.f - dd;
        // REFERENCE window rule (optimizer.py:69): Wp = the weight on the photometric term -- source 0's map for every forward
        // pair; wext: it is not this pair's own (no e dW/d theta term); crossf: source 0's weight also multiplies the pixels the
        // other sources won, their error times d W_0 / d theta enters source 0's gradient
        float Wp = Wt, crossf = 0.f;
        bool wext = false;
        if (SEL && sel_pair && P.rule) {
            if (s_own != 0) { Wp = sel_w0; wext = true; }
            else crossf = (inimg && sel_keep && !m) ? sel_dothers : 0.f;
        }

        f2 de2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // d(e2)/d theta, column pairs (01)(23)(45)
        float de6 = 0.f;
        // Pass B only serves masked-in pixels (its results are multiplied by the mask): a wave whose 64 pixels are ALL masked out --
        // the other source won them (min over the sources: the sources win in coherent regions), the auto-mask or the warp's
        // validity dropped them -- skips it (wave-uniform branch; the skipped terms would have been multiplied by zero).
        float adj_sx = 0.f, adj_sy = 0.f;      // ADJ: d C / d(ix, iy) of this thread's own position, SSIM part (carries m W of the residual pixels)
        f2 ring_g2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // ADJ: contribution of this thread's ring position (first NHALO threads)
        float ring_g6 = 0.f;
        if (ADJL) {
            const unsigned long long anyw = __builtin_amdgcn_ballot_w64(m);
            if (anyw != 0ull) {     // GN curvature of the SSIM term from the gradient sums of pass A (as at the end of pass B below)
                const float n9 = 1.f / 9.f;
                const f2 mx = f2{aGx01.x, aGy01.x} * n9, my = f2{aGx01.y, aGy01.y} * n9, ex = gxc01 - mx, ey = gyc01 - my;   // (the sums travel as (gx, gy) pairs of channels 0 / 1)
                const f2 qxx = t01.id2 * ex * ex + t01.id1 * mx * mx, qxy = t01.id2 * ex * ey + t01.id1 * mx * my,
                         qyy = t01.id2 * ey * ey + t01.id1 * my * my;
                const float mx2 = aG2s.x * n9, my2 = aG2s.y * n9, ex2 = g2c.x - mx2, ey2 = g2c.y - my2;
                lxx += qxx.x + qxx.y + t2.id2 * ex2 * ex2 + t2.id1 * mx2 * mx2;
                lxy += qxy.x + qxy.y + t2.id2 * ex2 * ey2 + t2.id1 * mx2 * my2;
                lyy += qyy.x + qyy.y + t2.id2 * ey2 * ey2 + t2.id1 * my2 * my2;
                if ((tid & 63) == 0) atomicOr(&adj_any, 1u << (tid >> 6));
            }
            // coefficient record of this residual pixel: m W (cA, cB, cC) per channel, re-centred from its own colours to 1/2 so that it
            // serves every position q:  d e2_p / d y_q = cA + cB (y_q - 1/2) + cC (x_q - 1/2)
            const float wq = m ? Wp : 0.f;
            const f2 h01 = {0.5f, 0.5f};
            const f2 wA01 = wq * (t01.cA + t01.cB * (h01 - yc01) + t01.cC * (h01 - xc01)), wB01 = wq * t01.cB, wC01 = wq * t01.cC;
            const float wA2 = wq * (t2.cA + t2.cB * (0.5f - yx2c.x) + t2.cC * (0.5f - yx2c.y)), wB2 = wq * t2.cB, wC2 = wq * t2.cC;
            // ring position of this thread (first NHALO threads): its colours / image gradients leave LDS before the records are reused
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0;
            int rlx = 0, rly = 0;
            const bool ring_thread = tid < HALO_THREADS;      // wave-uniform
            if (ring_thread) {
                const int hi = min(tid, NHALO - 1);
                if (hi < CW) { rly = 0; rlx = hi; }
                else if (hi < 2 * CW) { rly = CH - 1; rlx = hi - CW; }
                else { const int kq = hi - 2 * CW; rly = 1 + (kq >> 1); rlx = (kq & 1) ? CW - 1 : 0; }
                lds_read3v(lds + (rly * CW + rlx) * (LDS_REC / 4), r0, r1, r2);
            }
            __syncthreads();                                  // every pass-A read of the colour records is done
            {
                float4 *cr = const_cast<float4 *>(ctr);
                lds_write1(cr + 0, wA01.x, wA01.y, wB01.x, wB01.y);
                lds_write1(cr + 1, wC01.x, wC01.y, wA2, wB2);
                lds_write1(cr + 2, wC2, 0.f, 0.f, 0.f);
                if (ring_thread && tid < NHALO) {             // ring positions carry no residual of this tile
                    float4 *rr = lds + (rly * CW + rlx) * (LDS_REC / 4);
                    lds_write1(rr + 0, 0.f, 0.f, 0.f, 0.f); lds_write1(rr + 1, 0.f, 0.f, 0.f, 0.f); lds_write1(rr + 2, 0.f, 0.f, 0.f, 0.f);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned anyb = adj_any;
            const int wv = tid >> 6;
            // own position: residual pixels of rows ly-1 .. ly+1 live in waves wv-1 .. wv+1 (a wave = two tile rows)
            if ((anyb & (7u << wv) >> 1) != 0u) {
                f2 sA01 = {0.f, 0.f}, sB01 = {0.f, 0.f}, sC01 = {0.f, 0.f}, sAB2 = {0.f, 0.f};
                float sC2 = 0.f;
                const float4 *nb = ctr - (CW + 1) * (LDS_REC / 4);
#pragma unroll 1
                for (int kk = 0; kk < 9; kk++) {
                    f32x4 c0, c1, c2;
                    lds_read3v(nb, c0, c1, c2);
                    nb += (kk == 2 || kk == 5) ? (CW - 2) * (LDS_REC / 4) : (LDS_REC / 4);
                    sA01 += c0.lo; sB01 += c0.hi; sC01 += c1.lo; sAB2 += c1.hi; sC2 += c2.x;
                }
                const f2 lam01 = sA01 + sB01 * (yc01 - h01) + sC01 * (xc01 - h01);
                const float lam2 = sAB2.x + sAB2.y * (yx2c.x - 0.5f) + sC2 * (yx2c.y - 0.5f);
                const f2 tx = lam01 * gxc01, ty = lam01 * gyc01;
                adj_sx = tx.x + tx.y + lam2 * g2c.x; adj_sy = ty.x + ty.y + lam2 * g2c.y;
            }
            if (ring_thread && anyb != 0u) {                  // the ring position: only residual pixels inside the tile see it
                f2 sA01 = {0.f, 0.f}, sB01 = {0.f, 0.f}, sC01 = {0.f, 0.f}, sAB2 = {0.f, 0.f};
                float sC2 = 0.f;
#pragma unroll 1
                for (int kk = 0; kk < 9; kk++) {
                    const int ny = rly + kk / 3 - 1, nx = rlx + (kk - (kk / 3) * 3) - 1;
                    const bool ok = nx >= 1 && nx <= TW && ny >= 1 && ny <= TH;
                    f32x4 c0, c1, c2;
                    lds_read3v(lds + ((ok ? ny : rly) * CW + (ok ? nx : rlx)) * (LDS_REC / 4), c0, c1, c2);   // (own record: zeros)
                    sA01 += c0.lo; sB01 += c0.hi; sC01 += c1.lo; sAB2 += c1.hi; sC2 += c2.x;
                }
                const f2 lam01 = sA01 + sB01 * (r0.lo - h01) + sC01 * (r0.hi - h01);
                const float lam2 = sAB2.x + sAB2.y * (r2.x - 0.5f) + sC2 * (r2.y - 0.5f);
                const f2 rs2 = lam01.x * r1.lo + lam01.y * r1.hi + lam2 * r2.hi;
                const float live = tid < NHALO ? 1.f : 0.f;   // (threads NHALO .. HALO_THREADS-1 repeat the last ring position)
                const float rsx = live * rs2.x, rsy = live * rs2.y;
                f32x4 q3, q4, q5;
                lds_read3bv(lds + (rly * CW + rlx) * (LDS_REC / 4), q3, q4, q5);
                ring_g2[0] = rsx * f2{q3.x, 0.f} + rsy * f2{0.f, q3.y}; ring_g2[1] = rsx * q3.hi + rsy * q4.hi; ring_g2[2] = rsx * q4.lo + rsy * q5.lo;
                if (NP == 7) ring_g6 = rsx * q5.z + rsy * q5.w;
            }
        }
        if (!ADJL && MODE == MODE_LIN && __builtin_amdgcn_ballot_w64(m) != 0ull) {
            // pass B: exact SSIM gradient rows (neighbour geometry included); 15 VALU per neighbour (13 packed)
            // d SSIM_p / d y_q = cA + cB (y_q - y_c) + cC (x_q - x_c): the centre shift goes into the constant once per pixel
            // instead of three packed subtractions per neighbour
            const f2 cB01 = {cB[0], cB[1]}, cC01 = {cC[0], cC[1]};
            const f2 cA01 = f2{cA[0], cA[1]} - cB01 * yc01 - cC01 * xc01;
            const float cA2 = cA[2] - cB[2] * yx2c.x - cC[2] * yx2c.y;
            f2 Gx01 = {0.f, 0.f}, Gy01 = {0.f, 0.f}, G2 = {0.f, 0.f};   // 3x3 sums of the image gradients (curvature model)
            const float4 *nb = ctr - (CW + 1) * (LDS_REC / 4);
#if TC_PASSB_PIPELINED
            {
                // Software-pipelined over the nine window positions (unrolled; every position is a compile-time offset from the window's
                // first record): two register sets for the colour / gradient parts -- position k + 2's is requested as soon as position
                // k's has been consumed -- and one for the Jacobian parts, requested a position ahead.  Issue order C0 C1 J0 | C2 J1 | C3 J2 ...;
                // LDS returns in order, so ONE counted wait per position (lgkmcnt(3): everything but the newest colour request) covers
                // the Jacobians of this position and the colours of the next.  Same operations on the same values in the same order
                // as the rolled loop below: bit-identical results.
                constexpr int RB = LDS_REC * 4, ROWB = CW * LDS_REC * 4;
                const unsigned base = lds_addr(nb);
                f32x4 A0, A1, A2, B0, B1, B2, J3, J4, J5;
                f2 sxy;
                auto colour = [&](const f32x4 &n0, const f32x4 &n1, const f32x4 &n2) {
                    Gx01 += n1.lo; Gy01 += n1.hi; G2 += n2.hi;
                    f2 cf = cA01 + cB01 * n0.lo + cC01 * n0.hi;
                    float cf2 = cA2 + cB[2] * n2.x + cC[2] * n2.y;
                    sxy = pk_mul_b<0>(cf, n1.lo);            // (sx, sy) = cf_0 (gx0, gy0) + cf_1 (gx1, gy1) + cf_2 (gx2, gy2): three packed instructions
                    pk_fma_b<1>(sxy, cf, n1.hi);
                    sxy += cf2 * n2.hi;
                };
                auto rows = [&](const f32x4 &n3, const f32x4 &n4, const f32x4 &n5) {
                    de2[0] += sxy * n3.lo;                          // (sx a0, sy b1): a1 = b0 = 0
                    pk_fma_b<0>(de2[1], sxy, n3.hi); pk_fma_b<1>(de2[1], sxy, n4.hi);   // sx / sy broadcast from their halves of the pair
                    pk_fma_b<0>(de2[2], sxy, n4.lo); pk_fma_b<1>(de2[2], sxy, n5.lo);
                    if (NP == 7) de6 += sxy.x * n5.z + sxy.y * n5.w;
                };
#define TC_POS(k) (((k) / 3) * ROWB + ((k) % 3) * RB)
#if TC_PASSB_J2
                // Jacobian parts double-buffered as well (two positions ahead): issue order C0 C1 J0 J1 | C2 J2 | C3 J3 | ...; at most 12 reads in flight
                f32x4 K3, K4, K5;
                lds_issue3c_at<TC_POS(0)>(base, A0, A1, A2);
                lds_issue3c_at<TC_POS(1)>(base, B0, B1, B2);
                lds_issue3j_at<TC_POS(0)>(base, J3, J4, J5);
                lds_issue3j_at<TC_POS(1)>(base, K3, K4, K5);
                lds_waitn<9>(A0, A1, A2);
                colour(A0, A1, A2); lds_issue3c_at<TC_POS(2)>(base, A0, A1, A2); lds_waitn<6>(J3, J4, J5, B0, B1, B2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(2)>(base, J3, J4, J5);
                colour(B0, B1, B2); lds_issue3c_at<TC_POS(3)>(base, B0, B1, B2); lds_waitn<9>(K3, K4, K5);             rows(K3, K4, K5); lds_issue3j_at<TC_POS(3)>(base, K3, K4, K5);
#define TC_STEP(CA, CB, CC, JA, JB, JC, k) lds_waitn<9>(CA, CB, CC); colour(CA, CB, CC); lds_issue3c_at<TC_POS((k) + 2)>(base, CA, CB, CC); \
                lds_waitn<9>(JA, JB, JC); rows(JA, JB, JC); lds_issue3j_at<TC_POS((k) + 2)>(base, JA, JB, JC);
                TC_STEP(A0, A1, A2, J3, J4, J5, 2)
                TC_STEP(B0, B1, B2, K3, K4, K5, 3)
                TC_STEP(A0, A1, A2, J3, J4, J5, 4)
                TC_STEP(B0, B1, B2, K3, K4, K5, 5)
                TC_STEP(A0, A1, A2, J3, J4, J5, 6)
#undef TC_STEP
                // outstanding: C7 J7 C8 J8
                lds_waitn<9>(B0, B1, B2); colour(B0, B1, B2); lds_waitn<6>(K3, K4, K5); rows(K3, K4, K5);
                lds_waitn<3>(A0, A1, A2); colour(A0, A1, A2); lds_waitn<0>(J3, J4, J5); rows(J3, J4, J5);
#else
                lds_issue3c_at<TC_POS(0)>(base, A0, A1, A2);
                lds_issue3c_at<TC_POS(1)>(base, B0, B1, B2);
                lds_issue3j_at<TC_POS(0)>(base, J3, J4, J5);
                lds_waitn<6>(A0, A1, A2);
                colour(A0, A1, A2); lds_issue3c_at<TC_POS(2)>(base, A0, A1, A2); lds_waitn<3>(J3, J4, J5, B0, B1, B2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(1)>(base, J3, J4, J5);
                colour(B0, B1, B2); lds_issue3c_at<TC_POS(3)>(base, B0, B1, B2); lds_waitn<3>(J3, J4, J5, A0, A1, A2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(2)>(base, J3, J4, J5);
                colour(A0, A1, A2); lds_issue3c_at<TC_POS(4)>(base, A0, A1, A2); lds_waitn<3>(J3, J4, J5, B0, B1, B2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(3)>(base, J3, J4, J5);
                colour(B0, B1, B2); lds_issue3c_at<TC_POS(5)>(base, B0, B1, B2); lds_waitn<3>(J3, J4, J5, A0, A1, A2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(4)>(base, J3, J4, J5);
                colour(A0, A1, A2); lds_issue3c_at<TC_POS(6)>(base, A0, A1, A2); lds_waitn<3>(J3, J4, J5, B0, B1, B2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(5)>(base, J3, J4, J5);
                colour(B0, B1, B2); lds_issue3c_at<TC_POS(7)>(base, B0, B1, B2); lds_waitn<3>(J3, J4, J5, A0, A1, A2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(6)>(base, J3, J4, J5);
                colour(A0, A1, A2); lds_issue3c_at<TC_POS(8)>(base, A0, A1, A2); lds_waitn<3>(J3, J4, J5, B0, B1, B2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(7)>(base, J3, J4, J5);
                colour(B0, B1, B2);                                               lds_waitn<0>(J3, J4, J5, A0, A1, A2); rows(J3, J4, J5); lds_issue3j_at<TC_POS(8)>(base, J3, J4, J5);
                colour(A0, A1, A2);                                               lds_waitn<0>(J3, J4, J5);             rows(J3, J4, J5);
#endif
#undef TC_POS
            }
#else
#pragma unroll 1
            for (int kk = 0; kk < 9; kk++) {
                f32x4 n0, n1, n2, n3, n4, n5;
                lds_issue6v(nb, n0, n1, n2, n3, n4, n5);  // one LDS round trip per neighbour; the colour part is used while the Jacobians arrive
                lds_wait3of6(n0, n1, n2);
                Gx01 += n1.lo; Gy01 += n1.hi; G2 += n2.hi;
                f2 cf = cA01 + cB01 * n0.lo + cC01 * n0.hi;
                float cf2 = cA2 + cB[2] * n2.x + cC[2] * n2.y;
                f2 sxy = pk_mul_b<0>(cf, n1.lo);
                pk_fma_b<1>(sxy, cf, n1.hi);
                sxy += cf2 * n2.hi;
                const float sx = sxy.x, sy = sxy.y;
                lds_wait0(n3, n4, n5);
                de2[0] += sxy * n3.lo;                          // (sx a0, sy b1): a1 = b0 = 0
                de2[1] += sx * n3.hi; de2[1] += sy * n4.hi;     // separate statements: each contracts to one v_pk_fma_f32
                de2[2] += sx * n4.lo; de2[2] += sy * n5.lo;
                if (NP == 7) de6 += sx * n5.z + sy * n5.w;
                nb += (kk == 2 || kk == 5) ? (CW - 2) * (LDS_REC / 4) : (LDS_REC / 4);
            }
#endif
            {   // GN curvature of the SSIM term: Cov/d2 + mean mean'/d1, Cov ~ 9/8 (g - mean)(g - mean)' (centre sample)
                const float n9 = 1.f / 9.f;
                const f2 mx = f2{Gx01.x, Gy01.x} * n9, my = f2{Gx01.y, Gy01.y} * n9, ex = gxc01 - mx, ey = gyc01 - my;   // (Gx01 / Gy01 hold the (gx, gy) sums of channel 0 / 1)
                const f2 qxx = t01.id2 * ex * ex + t01.id1 * mx * mx, qxy = t01.id2 * ex * ey + t01.id1 * mx * my,
                         qyy = t01.id2 * ey * ey + t01.id1 * my * my;
                const float mx2 = G2.x * n9, my2 = G2.y * n9, ex2 = g2c.x - mx2, ey2 = g2c.y - my2;
                lxx += qxx.x + qxx.y + t2.id2 * ex2 * ex2 + t2.id1 * mx2 * mx2;
                lxy += qxy.x + qxy.y + t2.id2 * ex2 * ey2 + t2.id1 * mx2 * my2;
                lyy += qyy.x + qyy.y + t2.id2 * ey2 * ey2 + t2.id1 * my2 * my2;
            }
        }

        if (TRACE && MODE != MODE_MAPS && P.trace != nullptr && inimg) {   // parity tests replay these decisions in the float64 oracle
            unsigned short *tb = P.trace + (size_t)n * hw + (size_t)(y00 + ly - 1) * W + (x00 + lx - 1);   // (this thread's own phase-1 word)
            *tb = (unsigned short)(*tb | (m ? 1 : 0) | (c_valid[k] ? 2 : 0) | (sign_code(dif) << 4) | (sign_code(yc[0] - xc[0]) << 6) |
                                   (sign_code(yc[1] - xc[1]) << 8) | (sign_code(yc[2] - xc[2]) << 10));
        }

        if (MODE == MODE_MAPS) {
            if (inimg) {
                size_t o = (size_t)n * hw + (size_t)(y00 + ly - 1) * W + (x00 + lx - 1);
                if (P.o_diff) P.o_diff[o] = diff;
                if (P.o_valid) P.o_valid[o] = c_valid[k] ? 1.f : 0.f;
                if (P.o_weight) P.o_weight[o] = Wt;
                if (P.o_auto_err) P.o_auto_err[o] = c_ae[k];
                if (P.o_auto_mask) P.o_auto_mask[o] = diff < c_ae[k] ? 1.f : 0.f;
                if (P.o_rec) {
                    size_t o3 = (size_t)n * 3 * hw + (size_t)(y00 + ly - 1) * W + (x00 + lx - 1);
                    P.o_rec[o3] = yc[0] + cs0; P.o_rec[o3 + hw] = yc[1] + cs1; P.o_rec[o3 + 2 * hw] = yc[2] + cs2;
                }
            }
            continue;
        }

        if (inimg) sdd += dd;
        if (m) { sMWd += Wp * diff; sM += 1.f; }
        if (MODE == MODE_LIN) {
            // own geometric Jacobian (centre record), as column pairs
            f32x4 q3, q4, q5;
            lds_read3bv(ctr, q3, q4, q5);
            const f2 a2[3] = {f2{q3.x, 0.f}, q3.hi, q4.lo}, b2[3] = {f2{0.f, q3.y}, q4.hi, q5.lo};   // ([0]: not used below, see ab0)
            float sg = (raw >= 0.f && raw <= 1.f) ? (dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f)) : 0.f;
            float kdd = sg * 2.f * isum * isum;
            float mf = m ? 1.f : 0.f;
            if (FRONT) {      // d L / d pd of this inverse pixel, split by what multiplies it: b_dc h(dd) ddd  and  -a_i M diff ddd  (dense_ref_kernel.h)
                const float ddd = -sg * 2.f * cd * isum * isum * c.es;                  // d dd / d (sampled depth)
                f_dc = (inimg && c_valid[k]) ? fminf(1.f, dd * frcp(P.eps)) * ddd : 0.f;
                f_ph = m ? diff * ddd : 0.f;
            }
            const float wxx = mf * Wp * lxx, wxy = mf * Wp * lxy, wyy = mf * Wp * lyy;
            const float dsub = (SEL && wext) ? 0.f : diff;
            // Columns (0,1) of the Jacobian are (a0, 0) and (0, b1) (structural zeros, see the record layout): their products are formed from the
            // pair ab0 = (a0, b1) with ONE packed instruction where the general column pairs need two; the curvature rows travel as pairs
            // lalb_j = (la_j, lb_j) so that row j x columns (0,1) is the elementwise product lalb_j * ab0.  Same values as the general
            // expressions (the dropped terms are exact zeros; every remaining product / FMA keeps its operands and order).
            const f2 ab0 = q3.lo;
            f2 ddJ2[3], lalb[6];
            const f2 wA = {wxx, wxy}, wB = {wxy, wyy};
            {   // p = 0
                const f2 dpd = f2{c_dgx[k], c_dgy[k]} * ab0;
                ddJ2[0] = kdd * (-(cd * dpd));
                if (ADJL) {
                    const float sxt = adj_sx + mf * Wp * l1x, syt = adj_sy + mf * Wp * l1y;
                    aG2[0] += f2{sxt, syt} * ab0 - (mf * dsub) * ddJ2[0] + ring_g2[0];
                } else {
                    f2 row = Wp * (de2[0] + f2{l1x, l1y} * ab0) - dsub * ddJ2[0];   // d(W (e1+e2))/d theta
                    aG2[0] += mf * row;
                }
                if (SEL) aG2[0] -= crossf * ddJ2[0];
                lalb[0] = pk_mul_b<0>(ab0, wA);       // b0 = 0
                lalb[1] = pk_mul_b<1>(ab0, wB);       // a1 = 0
            }
#pragma unroll
            for (int p = 1; p < 3; p++) {
                const f2 zc2 = {c_zc[k][2 * p], c_zc[k][2 * p + 1]};
                f2 dpd = c_dgx[k] * a2[p] + c_dgy[k] * b2[p];
                ddJ2[p] = kdd * (pd * zc2 - cd * dpd);
                if (ADJL) {      // adjoint form: the SSIM part arrives as d C / d(ix, iy) of this position (m W inside), L1 joins it
                    const float sxt = adj_sx + mf * Wp * l1x, syt = adj_sy + mf * Wp * l1y;
                    aG2[p] += sxt * a2[p] + syt * b2[p] - (mf * dsub) * ddJ2[p] + ring_g2[p];
                } else {
                    f2 row = Wp * (de2[p] + l1x * a2[p] + l1y * b2[p]) - dsub * ddJ2[p];   // d(W (e1+e2))/d theta
                    aG2[p] += mf * row;
                }
                if (SEL) aG2[p] -= crossf * ddJ2[p];
                // (la_j, lb_j) = (wxx a_j + wxy b_j, wxy a_j + wyy b_j) for j = 2p, 2p + 1: the scalar a_j / b_j is broadcast from its half
                lalb[2 * p] = pk_mul_b<0>(b2[p], wB);     pk_fma_b<0>(lalb[2 * p], a2[p], wA);
                lalb[2 * p + 1] = pk_mul_b<1>(b2[p], wB); pk_fma_b<1>(lalb[2 * p + 1], a2[p], wA);
            }
            float ddJ6 = 0.f;
            f2 lalb6 = {0.f, 0.f};
            float a6 = 0.f, b6 = 0.f;
            if (NP == 7) {
                a6 = q5.z; b6 = q5.w;
                float dpd = c_dgx[k] * a6 + c_dgy[k] * b6 + pd;
                ddJ6 = kdd * (pd * c_zc[k][NP - 1] - cd * dpd);
                if (ADJL) aG6 += (adj_sx + mf * Wp * l1x) * a6 + (adj_sy + mf * Wp * l1y) * b6 - (mf * dsub) * ddJ6 + ring_g6;
                else aG6 += mf * (Wp * (de6 + l1x * a6 + l1y * b6) - dsub * ddJ6);
                if (SEL) aG6 -= crossf * ddJ6;
                lalb6 = wA * a6 + wB * b6;
            }
            // H row j (la_j, lb_j) x column pairs p <= j/2
            {
                int h = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    aH2[h] += lalb[j] * ab0; h++;                          // columns (0,1): (la_j a0, lb_j b1)
#pragma unroll
                    for (int p = 1; p <= (j >> 1); p++) { aH2[h] += lalb[j].x * a2[p]; pk_fma_b<1>(aH2[h], lalb[j], b2[p]); h++; }
                }
                if (NP == 7) {
                    aH2[h] += lalb6 * ab0; h++;
#pragma unroll
                    for (int p = 1; p < 3; p++) { aH2[h] += lalb6.x * a2[p]; pk_fma_b<1>(aH2[h], lalb6, b2[p]); h++; }
                    aH66 += lalb6.x * a6 + lalb6.y * b6;
                }
            }
            if (DC) {
                // IRLS curvature 1/max(dd,eps) -- over pixels whose projected depth is a real depth sample only (a footprint that
                // touches the zero padding gives a blend with 0: a handful of such border pixels would be 90 % of this curvature, see
                // the oracle); the gradient keeps every pixel and is Huberised inside dd < eps (sign(cd-pd) is rounding noise there)
                float k3 = (inimg && c_dcin[k]) ? frcp(fmaxf(dd, P.eps)) : 0.f, inf = inimg ? fminf(1.f, dd * frcp(P.eps)) : 0.f;
                int h = 0;
#pragma unroll
                for (int p = 0; p < 3; p++) dG2[p] += inf * ddJ2[p];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const float kj = k3 * ((j & 1) ? ddJ2[j >> 1].y : ddJ2[j >> 1].x);
#pragma unroll
                    for (int p = 0; p <= (j >> 1); p++) { dH2[h] += kj * ddJ2[p]; h++; }
                }
                if (NP == 7) {
                    dG6 += inf * ddJ6;
                    const float kj = k3 * ddJ6;
#pragma unroll
                    for (int p = 0; p < 3; p++) { dH2[h] += kj * ddJ2[p]; h++; }
                    dH66 += kj * ddJ6;
                }
            }
        }
    }
    if (MODE == MODE_MAPS) { stamp_end(P.stamp, tid); return; }
    // FRONT: the ADJOINT of this tile's bilinear samples (an inverse row: of the target depth -> ext2; a forward row with free source maps: of
    // its source map -> ext2_src).  The taps of the tile land in a window of the sampled image displaced by the tile's flow: they are summed
    // in LDS first -- 64-bit fixed-point adds on a (TW + 2 M) x (TH + 2 M) window placed by the tap of the tile's centre pixel, two sums per
    // entry -- then every non-zero entry goes out with one global atomic per sum; a tap outside the window goes to global memory directly.
    // Called when every wave is past a workgroup barrier behind phase 2 (the staged records are dead: the window aliases them).
    auto front_scatter = [&](long long *ext) {
        constexpr int FM = 6, WW = TW + 2 * FM, WH = TH + 2 * FM, NWIN = WW * WH;
        static_assert(2 * NWIN * sizeof(unsigned long long) <= sizeof(lds), "the scatter window aliases the staged records");
        unsigned long long *win = reinterpret_cast<unsigned long long *>(lds);
        for (int i = tid; i < 2 * NWIN; i += NT) win[i] = 0ull;
        const int tx0 = (f_tap & 0xffff) - 1, ty0 = (f_tap >> 16) - 1;
        if (tid == (TH / 2) * TW + TW / 2) { front_org[0] = tx0 - TW / 2 - FM; front_org[1] = ty0 - TH / 2 - FM; }
        __syncthreads();
        const int ox = front_org[0], oy = front_org[1];
        if (f_dc != 0.f || f_ph != 0.f) {
            const float w4[4] = {(1.f - f_wx) * (1.f - f_wy), f_wx * (1.f - f_wy), (1.f - f_wx) * f_wy, f_wx * f_wy};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = tx0 + (k & 1), yy = ty0 + (k >> 1);
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {         // (a tap in the zero border is no pixel of the sampled map)
                    const long long a0 = (long long)llrint((double)(f_dc * w4[k]) * DREF_FIX), a1 = (long long)llrint((double)(f_ph * w4[k]) * DREF_FIX);
                    const int wxl = xx - ox, wyl = yy - oy;
                    if (wxl >= 0 && wxl < WW && wyl >= 0 && wyl < WH) {
                        if (a0 != 0) atomicAdd(&win[wyl * WW + wxl], (unsigned long long)a0);
                        if (a1 != 0) atomicAdd(&win[NWIN + wyl * WW + wxl], (unsigned long long)a1);
                    } else {
                        unsigned long long *e = reinterpret_cast<unsigned long long *>(ext + ((size_t)yy * W + xx) * 2);
                        if (a0 != 0) atomicAdd(e, (unsigned long long)a0);
                        if (a1 != 0) atomicAdd(e + 1, (unsigned long long)a1);
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < NWIN; i += NT) {
            const unsigned long long a0 = win[i], a1 = win[NWIN + i];
            const int yy = oy + i / WW, xx = ox + i % WW;
            if ((a0 | a1) != 0ull && xx >= 0 && xx < W && yy >= 0 && yy < H) {
                unsigned long long *e = reinterpret_cast<unsigned long long *>(ext + ((size_t)yy * W + xx) * 2);
                if (a0 != 0ull) atomicAdd(e, a0);
                if (a1 != 0ull) atomicAdd(e + 1, a1);
            }
        }
    };
    if (FRONT) {      // the tile's mask count -> the batch normaliser of the pair's group (one integer atomic per workgroup)
        const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(front_m));
        if ((tid & 63) == 0 && cnt != 0) atomicAdd(&front_cnt, cnt);
        if (flight) {
            __syncthreads();
            const int b_ = (n % P.front_fwd) % P.front_Bt;
            if (tid == 0 && front_cnt != 0) atomicAdd(P.norms + 2 * (P.norm_B > 0 ? b_ / P.norm_B : 0) + (ffwd ? 0 : 1), front_cnt);
            if (P.front_light) front_scatter(ffwd ? P.ext2_src + (size_t)n * hw * 2 : P.ext2 + (size_t)b_ * hw * 2);
            stamp_end(P.stamp, tid);
            return;
        }
    }

    // unpack the row-pair accumulators into the triangular layout the reduction / solve kernel use
    float aHP[L::NH], aGP[NP], aHD[DC ? L::NH : 1], aGD[DC ? NP : 1];
    {
        int h2 = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int npair = (j == 6) ? 3 : (j >> 1) + 1;
#pragma unroll
            for (int i = 0; i <= j; i++) {
                float vp = 0.f, vd = 0.f;
                if (i < 6) {
                    vp = (i & 1) ? aH2[h2 + (i >> 1)].y : aH2[h2 + (i >> 1)].x;
                    if (DC) vd = (i & 1) ? dH2[h2 + (i >> 1)].y : dH2[h2 + (i >> 1)].x;
                } else { vp = aH66; vd = dH66; }
                aHP[j * (j + 1) / 2 + i] = vp;
                if (DC) aHD[j * (j + 1) / 2 + i] = vd;
            }
            h2 += npair;
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            aGP[j] = (j & 1) ? aG2[j >> 1].y : aG2[j >> 1].x;
            if (DC) aGD[j] = (j & 1) ? dG2[j >> 1].y : dG2[j >> 1].x;
        }
        if (NP == 7) { aGP[NP - 1] = aG6; if (DC) aGD[NP - 1] = dG6; }
    }

    // ---------------- workgroup reduction -> group record (shared with the dense kernel) ----------------
    {
        constexpr int NPH = L::NH + NP;
        constexpr int NLIVE = (MODE == MODE_LIN) ? (DC ? L::NACC : NPH + 3) : 3;
        float v[NLIVE];
        if (MODE == MODE_LIN) {
#pragma unroll
            for (int i = 0; i < L::NH; i++) v[i] = aHP[i];
#pragma unroll
            for (int i = 0; i < NP; i++) v[L::NH + i] = aGP[i];
            if (DC) {
#pragma unroll
                for (int i = 0; i < L::NH; i++) v[NPH + i] = aHD[i];
#pragma unroll
                for (int i = 0; i < NP; i++) v[NPH + L::NH + i] = aGD[i];
            }
        }
        v[NLIVE - 3] = sMWd; v[NLIVE - 2] = sM; v[NLIVE - 1] = sdd;
        block_reduce_publish<NP, NLIVE, (MODE == MODE_LIN), DC, NT>(P, v, red, n, bid, nblk, tid);
    }
    if (FRONT) {      // FRONT, a linearised inverse pair: its count and its scatter (every wave is past the barrier inside the reduction)
        const int b_ = (n % P.front_fwd) % P.front_Bt;
        if (tid == 0 && front_cnt != 0) atomicAdd(P.norms + 2 * (P.norm_B > 0 ? b_ / P.norm_B : 0) + 1, front_cnt);
        front_scatter(P.ext2 + (size_t)b_ * hw * 2);
    }
    stamp_end(P.stamp, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// k_init / k_solve / k_finish (fp64 per-pair logic; mirrors oracle/tcsfm_oracle.c orc_refine)

struct SolveParams {
    const float *partials;  // [N][ngrp][nacc] group records
    PairState *st;
    PairConst *pc;
    float *stats;           // [N][n_iters+1][TCSFM_NSTAT] or null
    double *lin_out;        // linearize debug: [N][np*np + np + 4] or null
    int ngrp, nacc, np, has_dc;
    int it, n_iters, solver, param, mode;  // mode 0: iteration step, 1: final LM cost check, 2: export only
    double b_dc;            // w_dc / (H W)
    double lambda_up, lambda_down, lambda_min;
    double prior_scale;     // weight of (log_scale - s0)^2 (np == 7)
    int shared_image;
    float *pose_out, *log_scale_out;  // written by the last launch of a refine call (null otherwise)
    long long *dbg;                   // diagnostic builds only: s_memrealtime stamps of the solve phases (null in production)
    double *delta_out;                // dense mode: [N][8] pose increment of this iteration for k_dense_update (else null)
    int *accept_out;                  // dense LM: [N] 1 = this launch accepted the trial (mode 0) / kept the last step (mode 1)
    int *trace_decide;                // tcsfm_debug_trace: [N] the same decision of THIS launch, or null
    // TCSFM_WINDOW_REFERENCE (compute_optimization_loss, optimizer.py:47-86): the photometric sums of a pair are normalised by the
    // mask count summed over ALL forward (n < grp_fwd) or ALL inverse (grp_fwd <= n < n_pairs) pairs of the call, times `scale`
    int rule, grp_fwd, n_pairs;
    double scale_fwd, scale_inv;      // 1 (argmin) or 0.25 (no argmin, :73) / 0.25 (:79)
    // l_pose_consist (optimizer.py:95-96; window REFERENCE rule, 6-DoF Gauss-Newton): c = weight / (6 S B), 0 = off; the partner of pair n
    // is n +- grp_fwd; pose_lin [2][n_pairs][12]: every pair's transform at linearisation `it` in buffer it & 1 (written by the pair
    // initialisation and by the previous launch: a pair never reads what its partner writes in the SAME launch)
    double w_pc, pc_eps;                // pc_eps: floor of |r| in the IRLS weights (opts.irls_eps)
    double *pose_lin;
    // the same term with this launch's pairs a SLICE of the call's pairs (the inverse pairs of the dense mode on the reference's loss, solved
    // as 0 .. S B - 1): pc_np > 0 = pairs per buffer, pair n sits at pc_self0 + n and its partner at pc_part0 + n
    int pc_np, pc_self0, pc_part0;
    // coalesced calls (CoalTab): the refined pose of batch pair n goes to ITS call's output, at the pair's index in that call
    int c_ncall, c_B, c_S, c_pad;
    float *c_pose_out[TC_MAX_COAL];
    float *c_ls_out[TC_MAX_COAL];       // (np == 7) per call, or null
    // rule with the batch counts already summed (k_linearize<FRONT>, dense mode on the reference's loss): norms [groups][2] = K_f, K_i; the
    // group of pair n is ((n % norm_Bt) / norm_B) (norm_B = 0: one group) -- replaces the sum over the group's records
    const int *norms;
    int norm_Bt, norm_B;
    int c_n0;                           // coalesced calls: batch index of this launch's pair 0 (k_solve_front solves the inverse pairs as 0 .. S Bt - 1)
};

// fp32 extraction of the reference 6-vector from the fp64 transform (inverse of pose_to_T); angles are small, fp32
// inverse-trig keeps ~1e-7 relative accuracy and stays off the fp64 serial path
__device__ inline void T_to_pose_f32(const double *T, float *pose) {
    const float sbr = (float)T[2];
    float sb = sbr < -1.f ? -1.f : (sbr > 1.f ? 1.f : sbr);   // (comparisons, not fminf / fmaxf: a NaN pose must stay NaN in every component)
    pose[0] = (float)-T[3]; pose[1] = (float)-T[7]; pose[2] = (float)-T[11];
    pose[3] = -atan2f((float)-T[6], (float)T[10]);
    pose[4] = -asinf(sb);
    pose[5] = -atan2f((float)-T[1], (float)T[0]);
}

// One workgroup per pair.
//   1. all 256 threads: deterministic fp64 reduction of the workgroup partial records -> tot[] (LDS)
//   2. wave 0, one lane per entry of the augmented 8x8 system [H | -g]: assembly, LM accept/reject bookkeeping,
//      Marquardt damping and an unpivoted Gauss-Jordan elimination (SPD system) with 3 cross-lane reads per pivot
//   3. lane 0: SE(3) retraction (series exp, no trig), next iteration's fp32 constants, pose output
// The per-pair logic mirrors orc_refine() of the CPU oracle (which factorises with Cholesky instead).
// NT: threads that enter (256: k_solve; 1024: the pair role of k_solve_front, whose record sums are spread over four times the threads).
// LEAN: the SE(3) chart without the pose-consistency term (what the dense mode on the reference's loss asks of it): the rare branches that
// set the register budget of the general kernel are compiled out (k_solve_front runs 1024-thread workgroups: 128 VGPRs).
template <int NP, int NT, bool LEAN = false, bool PC = !LEAN>
__device__ __forceinline__ void solve_body(const SolveParams &P, const int n, const int tid) {
    using L = AccLayout<NP>;
    constexpr int NPH = L::NH + NP;
    __shared__ double tot[L::NACC];
    __shared__ double part[NT];
    __shared__ double ws[3 * NP * NP];
    __shared__ double dl[8];
    __shared__ double eul[NP * NP + NP];
    __shared__ double Ts[40];
    constexpr int NST = (int)(sizeof(PairState) / sizeof(double));
    static_assert(sizeof(PairState) % sizeof(double) == 0 && NST <= NT, "PairState must be a whole number of doubles");
    __shared__ double sst[NST];
#define TC_STAMP(i) if (P.dbg && tid == 0 && n == 0) P.dbg[i] = wall_clock64();
    TC_STAMP(0)
    // the pair's optimiser state is fetched NOW, together with the partial records, so that the serial phases below never
    // wait on a global load (each first touch used to cost a miss in the middle of the dependent chain)
    if (tid < NST) sst[tid] = reinterpret_cast<const double *>(&P.st[n])[tid];
    __shared__ double pcT[12], pcA[36], pcD[6], pcG[6];
    const bool pc_on = PC && NP == 6 && P.rule && P.w_pc > 0.0 && P.pose_lin != nullptr;
    if (pc_on && tid >= 64 && tid < 76) {
        const int partner = P.pc_np > 0 ? P.pc_part0 + n : (n < P.grp_fwd ? n + P.grp_fwd : n - P.grp_fwd);
        pcT[tid - 64] = P.pose_lin[((size_t)(P.it & 1) * (P.pc_np > 0 ? P.pc_np : P.n_pairs) + partner) * 12 + (tid - 64)];
    }
    __shared__ double kpart[NT];
    if (P.rule && !P.norms) {   // batch-summed mask count of this pair's group: every record of every pair of the group, fixed order
        const int g0 = n < P.grp_fwd ? 0 : P.grp_fwd, g1 = n < P.grp_fwd ? P.grp_fwd : P.n_pairs, cnt = (g1 - g0) * P.ngrp;
        const float *p = P.partials + (size_t)g0 * P.ngrp * L::NACC + L::OFF_S + 1;
        double s = 0.0;
        for (int i = tid; i < cnt; i += NT) s += (double)p[(size_t)i * L::NACC];
        kpart[tid] = s;
    }
    {
        // Deterministic fp64 reduction of the pair's P.ngrp partial records (group records, or one record per workgroup in
        // direct mode).  Only the live accumulators are read; they are spread over 256 threads as (accumulator, record
        // subset) so that ALL loads of a thread are in flight before its first add -- the records were written by other CUs,
        // every load is a miss (~0.4 us each if serialised; a predicated load compiles to branch + vmcnt(0) per element).
        const int nlive = P.has_dc ? L::NACC : NPH + 3;
        const int apad = nlive <= 32 ? 32 : (nlive <= 64 ? 64 : 128), parts = NT / apad;
        const int c = tid & (apad - 1), q = tid / apad;
        const int acc = (P.has_dc || c < NPH) ? c : L::OFF_S + (c - NPH);   // compact live index -> accumulator
        if (tid < L::NACC) tot[tid] = 0.0;
        double s = 0.0;
        if (c < nlive) {
            const float *p = P.partials + (size_t)n * P.ngrp * L::NACC + acc;
#ifndef TC_SOLVE_NB
#define TC_SOLVE_NB 32
#endif
            constexpr int NB = NT >= 1024 ? 16 : TC_SOLVE_NB;       // loads in flight per thread and batch (1024 threads: 16 x 16 subsets cover 256 records)
            for (int r0 = q; r0 < P.ngrp; r0 += NB * parts) {
                float v[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) { const int r = r0 + j * parts; v[j] = p[(size_t)(r < P.ngrp ? r : 0) * L::NACC]; }
#pragma unroll
                for (int j = 0; j < NB; j++) s += (r0 + j * parts < P.ngrp) ? (double)v[j] : 0.0;
            }
        }
        part[tid] = s;
        __syncthreads();
        if (q == 0 && c < nlive) {
            double t = 0.0;
            for (int k = 0; k < parts; k++) t += part[k * apad + c];   // fixed order
            tot[acc] = t;
        }
    }
    __syncthreads();
    TC_STAMP(1)
    if (tid >= 64) return;  // wave 0 only from here: no workgroup barriers below

    PairState &S = P.st[n];                                        // global: writes (the next launch reads them)
    const PairState &Lc = *reinterpret_cast<const PairState *>(sst);  // LDS copy taken at kernel start: reads
    const int r = tid >> 3, c = tid & 7;
    const double nmask = tot[L::OFF_S + 1];
    double an = nmask > 0 ? rcp64(nmask) : 0.0;
    if (P.rule) {
        double kn = 0.0;
        if (P.norms) kn = (double)P.norms[2 * (P.norm_B > 0 ? (n % P.norm_Bt) / P.norm_B : 0) + (n < P.grp_fwd ? 0 : 1)];
        else for (int i = 0; i < NT; i++) kn += kpart[i];   // (every lane the same fixed-order sum: LDS broadcast reads)
        an = kn > 0 ? (n < P.grp_fwd ? P.scale_fwd : P.scale_inv) * rcp64(kn) : 0.0;
    }
    const double cost_photo = an * tot[L::OFF_S], cost_dc = P.b_dc * tot[L::OFF_S + 2];
    double cost = cost_photo + cost_dc;
    const double bdc = P.has_dc ? P.b_dc : 0.0;
    // this lane's entry of [H | -g]
    double M = 0.0;
    if (r < NP && c < NP) {
        const int hi = r > c ? r : c, lo = r > c ? c : r, h = hi * (hi + 1) / 2 + lo;
        M = an * tot[L::OFF_HP + h] + bdc * tot[L::OFF_HD + h];
    } else if (r < NP && c == 7) {
        M = -(an * tot[L::OFF_GP + r] + bdc * tot[L::OFF_GD + r]);
    }
    if (NP == 7 && P.mode != 2) {  // scale prior (not part of the exported raw normal equations)
        const double ds = Lc.stry - Lc.s0;
        cost += P.prior_scale * ds * ds;
        if (r == 6 && c == 6) M += 2.0 * P.prior_scale;
        if (r == 6 && c == 7) M -= 2.0 * P.prior_scale * ds;
    }
    if (pc_on) {
        // l_pose_consist: c sum_j |p_n + p_partner|_j with both poses at this linearisation; gradient c r / max(|r|, eps) and the
        // block-Jacobi majoriser 2 c / max(|r|, eps) in pose coordinates, carried to the left perturbation by dp = A^-1 dxi,
        // A^-1 = [[-I, Tx], [0, -Je^-1]] (oracle pose_consist_term; se3_math.h euler_left_jacobian is A)
        double pm[6], pp[6];
        T_to_pose(Lc.Ttry, pm); T_to_pose(pcT, pp);
        const double cx = cos(-pm[3]), sx = sin(-pm[3]), cy = cos(-pm[4]), sy = sin(-pm[4]);
        const double Je[9] = {1, 0, sy, 0, cx, -sx * cy, 0, sx, cx * cy};
        const double id = 1.0 / cy;
        const double Ji[9] = {(Je[4] * Je[8] - Je[5] * Je[7]) * id, -(Je[1] * Je[8] - Je[2] * Je[7]) * id, (Je[1] * Je[5] - Je[2] * Je[4]) * id,
                              -(Je[3] * Je[8] - Je[5] * Je[6]) * id, (Je[0] * Je[8] - Je[2] * Je[6]) * id, -(Je[0] * Je[5] - Je[2] * Je[3]) * id,
                              (Je[3] * Je[7] - Je[4] * Je[6]) * id, -(Je[0] * Je[7] - Je[1] * Je[6]) * id, (Je[0] * Je[4] - Je[1] * Je[3]) * id};
        const double tp[3] = {-pm[0], -pm[1], -pm[2]};
        const double Tx[9] = {0, -tp[2], tp[1], tp[2], 0, -tp[0], -tp[1], tp[0], 0};
        if (tid < 36) {
            const int i = tid / 6, j = tid - 6 * i;
            double v = 0.0;
            if (i < 3 && j < 3) v = (i == j) ? -1.0 : 0.0;
            else if (i < 3) v = Tx[3 * i + (j - 3)];
            else if (j >= 3) v = -Ji[3 * (i - 3) + (j - 3)];
            pcA[tid] = v;
        }
        const double eps_pc = P.pc_eps;
        double pcc = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double rj = pm[j] + pp[j], a = fabs(rj), den = a > eps_pc ? a : eps_pc;
            pcc += 0.5 * P.w_pc * a;
            if (tid == j) { pcG[j] = P.w_pc * rj / den; pcD[j] = 2.0 * P.w_pc / den; }
        }
        cost += pcc;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (r < 6 && c < 6) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) v += pcA[6 * k + r] * pcD[k] * pcA[6 * k + c];
            M += v;
        } else if (r < 6 && c == 7) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) v += pcA[6 * k + r] * pcG[k];
            M -= v;
        }
    }
    if (P.mode == 2) {  // export for tcsfm_linearize / tcsfm_loss_surface
        double *o = P.lin_out + (size_t)n * (NP * NP + NP + 4);
        if (r < NP && c < NP) o[r * NP + c] = M;
        if (r < NP && c == 7) o[NP * NP + r] = -M;
        if (tid == 0) { o[NP * NP + NP] = cost; o[NP * NP + NP + 1] = cost_photo; o[NP * NP + NP + 2] = cost_dc; o[NP * NP + NP + 3] = nmask; }
        return;
    }
    double lambda = Lc.lambda;
    if (P.stats && tid == 0) {
        float *st = P.stats + ((size_t)n * (P.n_iters + 1) + P.it) * TCSFM_NSTAT;
        st[0] = (float)cost; st[1] = (float)cost_photo; st[2] = (float)nmask; st[3] = (float)lambda;
        T_to_pose_f32(Lc.Ttry, st + TCSFM_STAT_POSE);   // the iterate this linearisation was evaluated at
    }
    // Ts (LDS): [0..11] exp(delta), [12..23] the accepted transform, [24..35] the new trial transform, [36] its log scale
    bool final_pose = false;
    const double *Tfin = Ts + 24;
    double sfin = 0.0;
    if (P.mode == 1) {  // LM: keep the last step only if it lowered the cost
        const bool keep = cost < Lc.cost_cur;
        if (P.accept_out && tid == 0) P.accept_out[n] = keep ? 1 : 0;
        if (P.trace_decide && tid == 0) P.trace_decide[n] = keep ? 1 : 0;
        if (tid < 12) { const double v = keep ? Lc.Ttry[tid] : Lc.Tcur[tid]; Ts[24 + tid] = v; if (keep) S.Tcur[tid] = v; }
        sfin = keep ? Lc.stry : Lc.scur;
        if (tid == 0 && keep) S.scur = sfin;
        final_pose = true;
    } else {
        const bool accept = (P.solver == 0) || !Lc.have_cur || (cost < Lc.cost_cur);  // wave-uniform
        if (accept) {
            if (P.solver == 1 && Lc.have_cur) lambda = fmax(lambda * P.lambda_down, P.lambda_min);
            S.M8[tid] = M;
        } else {
            lambda *= P.lambda_up;
            M = Lc.M8[tid];
        }
        if (!LEAN && P.param != 0) {   // undamped system in dense NP x NP form for the additive-Euler branch below
            if (r < NP && c < NP) eul[r * NP + c] = M;
            if (r < NP && c == 7) eul[NP * NP + r] = -M;
        }
        TC_STAMP(2)
        // Marquardt damping, then Gauss-Jordan on [H + lambda diag(H) + 1e-12 I | -g]
        if (r == c && r < NP) M += lambda * M + 1e-12;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const double piv = __shfl(M, k * 8 + k, 64);
            const double rowv = __shfl(M, k * 8 + c, 64);
            const double colv = __shfl(M, r * 8 + k, 64);
            ok = ok && (piv > 0.0);
            const double f = colv * rcp64(piv);
            if (r != k) M -= f * rowv;
        }
        const double diag = __shfl(M, r * 8 + r, 64);
        if (c == 7 && r < NP) dl[r] = ok ? M * rcp64(diag) : 0.0;
        const double sc = accept ? Lc.stry : Lc.scur;
        if (tid < 12) { const double v = accept ? Lc.Ttry[tid] : Lc.Tcur[tid]; Ts[12 + tid] = v; if (accept) S.Tcur[tid] = v; }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): dl[] / Ts[] written before they are read (single wave)
        __builtin_amdgcn_wave_barrier();
        TC_STAMP(3)
        if (tid == 0) {   // serial part: bookkeeping and the exponential of the step
            if (P.accept_out) P.accept_out[n] = accept ? 1 : 0;
            if (P.trace_decide) P.trace_decide[n] = accept ? 1 : 0;
            if (accept) { S.scur = sc; S.cost_cur = cost; S.have_cur = 1; }
            S.lambda = lambda;
            if (P.delta_out)
                for (int i = 0; i < NP; i++) P.delta_out[n * 8 + i] = dl[i];
            if (LEAN || P.param == 0) {
                double d6[6];
#pragma unroll
                for (int i = 0; i < 6; i++) d6[i] = dl[i];
                se3_exp(d6, Ts);
                Ts[36] = sc + (NP == 7 ? dl[NP - 1] : 0.0);
            } else {
                // additive Euler parameterisation: re-solve in pose coordinates from the undamped system (rare path, serial;
                // everything stays in LDS so that this branch does not set the register budget of the kernel)
                double stry;
                apply_step<NP>(1, eul, eul + NP * NP, lambda, Ts + 12, sc, Ts + 24, &stry, ws);
                Ts[36] = stry;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if ((LEAN || P.param == 0) && tid < 12) {   // T_try = exp(delta) T_accepted, one entry per lane (se3_mul's operation order)
            const int i = tid >> 2, j = tid & 3;
            double v = Ts[4 * i] * Ts[12 + j] + Ts[4 * i + 1] * Ts[16 + j] + Ts[4 * i + 2] * Ts[20 + j];
            if (j == 3) v += Ts[4 * i + 3];
            Ts[24 + tid] = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        TC_STAMP(4)
        const double stry = Ts[36];
        const bool last_gn = (P.solver == 0 && P.it == P.n_iters - 1);   // GN: the last step is always taken
        if (tid < 12) {
            const double v = Ts[24 + tid]; S.Ttry[tid] = v; if (last_gn) S.Tcur[tid] = v;
            if (P.pose_lin) P.pose_lin[((size_t)((P.it + 1) & 1) * (P.pc_np > 0 ? P.pc_np : P.n_pairs) + (P.pc_np > 0 ? P.pc_self0 + n : n)) * 12 + tid] = v;
        }
        if (tid == 0) { S.stry = stry; if (last_gn) S.scur = stry; }
        write_const_lanes<NP>(tid, Lc.K, Ts + 24, stry, P.pc[n]);
        TC_STAMP(5)
        final_pose = last_gn;
        sfin = stry;
    }
    if (final_pose && P.pose_out && tid == 0) {  // last launch of a refine call: emit the reference 6-vector
        float pose[6];
        T_to_pose_f32(Tfin, pose);
        float *po = P.pose_out + n * 6;
        if (P.c_ncall > 0) { const CoalIdx ci = coal_index(P.c_ncall, P.c_B, P.c_S, n + P.c_n0); po = P.c_pose_out[ci.call] + ci.li * 6; }
#pragma unroll
        for (int i = 0; i < 6; i++) po[i] = pose[i];
        if (P.c_ncall > 0) {
            const CoalIdx ci = coal_index(P.c_ncall, P.c_B, P.c_S, n + P.c_n0);
            if (NP == 7 && P.c_ls_out[ci.call]) P.c_ls_out[ci.call][ci.li] = (float)sfin;
        } else if (P.log_scale_out) P.log_scale_out[n] = (float)sfin;
        if (P.stats && P.mode == 0) {                  // GN: last row = final iterate (its cost is not evaluated)
            float *st = P.stats + ((size_t)n * (P.n_iters + 1) + P.n_iters) * TCSFM_NSTAT + TCSFM_STAT_POSE;
#pragma unroll
            for (int i = 0; i < 6; i++) st[i] = pose[i];
        }
    }
    TC_STAMP(6)
#undef TC_STAMP
}
#ifndef TC_SOLVE_NT
#define TC_SOLVE_NT 256
#endif
template <int NP>
__global__ __launch_bounds__(TC_SOLVE_NT) void k_solve(SolveParams P) { solve_body<NP, TC_SOLVE_NT>(P, blockIdx.x, threadIdx.x); }
// The common case -- SE(3) chart, no pose-consistency term -- with 1024 threads (round 5): the 240 workgroup records of a KITTI pair are summed
// from ONE batch of loads per thread (16 record subsets of 15) instead of two dependent batches of 32, and the rare branches that set the
// general kernel's register budget are compiled out.  Same sums in a different association: results differ from k_solve<NP> in the last bits,
// so a call uses ONE of the two throughout (launch_solve decides on the options, not on the size).
template <int NP>
__global__ __launch_bounds__(1024) void k_solve_lean(SolveParams P) { solve_body<NP, 1024, true>(P, blockIdx.x, threadIdx.x); }

// dense sequence calls: the refined depth maps of one call, stacked [pair index j][window b] by the window form, into the caller's
// per-window order [window b][pair index j]
__global__ __launch_bounds__(256) void k_maps_to_window_order(const float *__restrict__ src, float *__restrict__ dst, int nbw, int N, int hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;      // m = j * nbw + b
    if (i >= hw) return;
    const int j = m / nbw, b = m - j * nbw;
    dst[((size_t)b * N + j) * hw + i] = src[(size_t)m * hw + i];
}

struct FinishParams {
    const PairState *st;
    float *pose_out, *log_scale_out;
    int N;
};

__global__ void k_finish(FinishParams P) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= P.N) return;
    double pose[6];
    T_to_pose(P.st[n].Tcur, pose);
    for (int i = 0; i < 6; i++) P.pose_out[n * 6 + i] = (float)pose[i];
    if (P.log_scale_out) P.log_scale_out[n] = (float)P.st[n].scur;
}

}  // namespace tc
