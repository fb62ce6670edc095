// dense_ref_kernel.h -- dense mode on the REFERENCE's own loss (round 4): compute_optimization_loss as optimize_depth_pred minimises it
// (optimizer.py:47-90, 194-198, 235-247), mirrors linearize_dense_ref / orc_refine_dense_ref of the oracle.
//
//   L = c_f / K_f  sum M_s W_x diff_s  +  0.25 / K_i sum M_i W_i diff_i  +  w_dc / (S B HW) sum (dd_fwd + dd_inv)  +  w_init / (B HW) sum SSIM(sigma, sigma0)
//
// Unknowns: the poses of all 2 S B directed pairs and ONE inverse-depth map per target.
// Round 5: FOUR dependent launches per Gauss-Newton iteration (five with the quarter-resolution unknown) instead of eight to eleven:
//   k_linearize<6, DC, LIN, .., FRONT>  all 2 S B pairs: the forward pairs' mask / selection and its count K_f; the inverse pairs' 6 x 6
//                            linearisation, their count K_i and the adjoint scatter of their samples of the target depth (two fixed-point
//                            sums per target pixel: no normaliser needed when they are scattered)                        (kernels.h)
//   k_dense_joint<S, .., REF>  the forward group (reads K_f, K_i, the selection map and the scattered sums)              (joint_kernel.h)
//   [k_qres_schur<S>           quarter-resolution unknown: cell records and their Schur terms]
//   k_solve_front<S>         the target groups' 6S x 6S systems AND the inverse pairs' 6 x 6 systems in one launch
//   k_dense_joint_update<S>  back-substitution (k_qres_step_up<S>: cell step + x4 upsampling in one launch)
// The chain below (residual maps, counts, scatter as launches of their own) remains for the free-source-map mode and the exports:
// per linearisation, on one stream:
//   k_linearize<MODE_MAPS>   residual maps (diff, valid) of all 2 S B pairs at the current poses / depth            (kernels.h)
//   k_dref_count             the batch-summed mask counts: K_f of the forward pairs' min-over-sources selection, K_i of the inverse pairs
//   k_dref_scatter           the ADJOINT of the inverse pairs' bilinear samples of the target depth -- d L / d pd of every inverse pixel onto
//                            its four taps, summed per tile in LDS and written with 64-bit fixed-point atomics (integer addition is
//                            order-independent: the result is bit-reproducible)
//   k_linearize<6, DC> + k_solve<6> (window rule REFERENCE)   the inverse pairs' 6 x 6 pose systems                  (kernels.h)
//   k_dense_joint<S, .., REF>  the forward group: source 0's weight map on every selected pixel with its cross term, depth consistency
//                            with its inverse-depth column, the SSIM prior, the scattered sums; per-pixel Schur elimination  (joint_kernel.h)
//   k_solve_joint<S>, k_dense_joint_update<S>   6S x 6S solve, back-substitution; the new map also goes into the inverse pairs' packs
#pragma once
#include "joint_kernel.h"

namespace tc {

struct DrefPrepassParams {
    const float *diff, *valid;    // [2SB][H*W] residual maps of all pairs (k_linearize<MODE_MAPS>)
    int *norms;                   // [2] K_f, K_i (zeroed before k_dref_count)
    long long *ext;               // [B][H*W][2] fixed-point scatter sums (zero before k_dref_scatter): depth-consistency part, photometric part
    int B, S, argmin, automask;
    float eps;
    float b_dc;                   // w_dc / (S B H W)
    int plain_dif;                // k_dref_scatter: the inverse pairs are linearised by k_dense_joint (free source maps), whose sign of cd - pd is the
                                  // plain fp32 difference's -- not k_linearize's cancellation-free dc_diff: the adjoint takes the linearising kernel's
};

// The scattered sum of a target pixel is  sum_p (b_dc h - a_i M diff) ddd w_tap  with a_i = 0.25 / K_i.  Its two parts are accumulated
// SEPARATELY and without their factors -- sum h ddd w and sum M diff ddd w: O(1) numbers for the 2^-40 fixed point whatever the image size,
// and no batch normaliser is needed while scattering (round 5: the scatter rides in the launch that counts K_i); the consumer
// (k_dense_joint) applies b_dc and a_i.

// pass 1: the batch-summed mask counts.  DREF_CNT_WG workgroups per directed pair stride over its pixels, count in registers (ballot +
// popcount per wave), combine in LDS and issue ONE integer atomic per workgroup: a few hundred same-address atomics per linearisation
// (one per wave of pixels -- 7 680 of them at 640x192 -- serialised on the two counters and took 80 us)
constexpr int DREF_CNT_WG = 64;
__global__ __launch_bounds__(256) void k_dref_count(LinParams P, DrefPrepassParams D) {
    __shared__ int wsum;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int hw = P.H * P.W, SB = D.S * D.B;
    if (tid == 0) wsum = 0;
    __syncthreads();
    int cnt = 0;                  // wave-uniform
    for (int base = blockIdx.x * 256; base < hw; base += DREF_CNT_WG * 256) {
        const int idx = base + tid;
        const bool live = idx < hw;
        const int gi = live ? idx : 0;
        bool count = false;
        if (live) {
            const bool valid = P.ext_valid[(size_t)n * hw + gi] > 0.5f;
            const float diff = P.ext_diff[(size_t)n * hw + gi], ae = P.tgtpack[(size_t)n * hw + gi].w;
            if (n < SB) count = (D.argmin && D.S > 1) ? ext_selected(P, n, gi, hw) : (valid && (!(D.automask && D.argmin) || diff < ae));
            else count = valid && (!D.automask || diff < ae);
        }
        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(count));
    }
    if ((tid & 63) == 0 && cnt != 0) atomicAdd(&wsum, cnt);
    __syncthreads();
    if (tid == 0 && wsum != 0) atomicAdd(D.norms + (n < SB ? 0 : 1), wsum);
}

// pass 2: the adjoint of the inverse pairs' bilinear samples of the target depth.  One workgroup = one 32 x 8 tile of one inverse pair;
// its taps land in a window of the TARGET image displaced by the tile's flow: the contributions are first summed in LDS (64-bit
// fixed-point adds on a (32 + 2M) x (8 + 2M) window placed by the flow of the tile's centre pixel), then every non-zero window entry
// goes out with ONE global atomic -- a third of the atomics of scattering tap by tap, and neighbouring lanes write neighbouring
// addresses.  A tap outside the window (depth edges) goes to global memory directly.  Integer adds: order-independent, bit-reproducible.
constexpr int DREF_TW = 32, DREF_TH = 8, DREF_M = 6;
__global__ __launch_bounds__(DREF_TW * DREF_TH) void k_dref_scatter(LinParams P, DrefPrepassParams D) {
    constexpr int WW = DREF_TW + 2 * DREF_M, WH = DREF_TH + 2 * DREF_M, NWIN = WW * WH;
    __shared__ unsigned long long win[NWIN];
    const int H = P.H, W = P.W, hw = H * W, SB = D.S * D.B;
    const int m = blockIdx.y, n = SB + m, b = m % D.B;
    const int tiles_x = (W + DREF_TW - 1) / DREF_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int tid = threadIdx.x, lx = tid % DREF_TW, ly = tid / DREF_TW;
    const int u = tx * DREF_TW + lx, v = ty * DREF_TH + ly;
    const PairConst &c = P.pc[n];
    const float *depth_t = P.depth_t + (size_t)n * hw;
    for (int i = tid; i < NWIN; i += DREF_TW * DREF_TH) win[i] = 0ull;
    // window origin: the tile's own origin displaced by the flow of its centre pixel (every thread evaluates it: no broadcast, no barrier)
    int ox, oy;
    {
        const int cu = min(tx * DREF_TW + DREF_TW / 2, W - 1), cv = min(ty * DREF_TH + DREF_TH / 2, H - 1);
        Geo gc;
        warp_geo(c, W, H, cu, cv, depth_t[cv * W + cu], gc);
        const float fx = fminf(fmaxf(gc.rx, -4096.f), 4096.f), fy = fminf(fmaxf(gc.ry, -4096.f), 4096.f);
        ox = tx * DREF_TW + (int)floorf(fx) - DREF_M; oy = ty * DREF_TH + (int)floorf(fy) - DREF_M;
    }
    // This launch runs BEHIND k_dref_count: K_i is known, so the two parts are combined into ONE sum per tap -- in the slot whose factor the
    // consumer applies: slot 1 (factor -a_i) holds  sum (M diff - (b_dc / a_i) h) ddd w  when inverse pixels count, slot 0 (factor b_dc) holds
    // sum h ddd w  otherwise.  (k_linearize<FRONT> scatters before the counts exist and fills both slots.)
    const float Ki = (float)D.norms[1];
    const int slot = Ki > 0.f ? 1 : 0;
    const float ratio_dc = Ki > 0.f ? -D.b_dc * Ki * 4.f : 1.f, w_photo = Ki > 0.f ? -1.f : 0.f;       // (signs: slot 1 is SUBTRACTED by the consumer)
    __syncthreads();
    long long *ext = D.ext + (size_t)b * hw * 2;
    if (u < W && v < H) {
        const int gi = v * W + u;
        const bool valid = P.ext_valid[(size_t)n * hw + gi] > 0.5f;
        if (valid) {
            const float diff = P.ext_diff[(size_t)n * hw + gi];
            const bool count = !D.automask || diff < P.tgtpack[(size_t)n * hw + gi].w;
            Geo g;
            warp_geo(c, W, H, u, v, depth_t[gi], g);
            Tap t;
            tap4_fetch(P.srcpack + (size_t)n * (H + 2) * (W + 2), W, H, u, v, g.rx, g.ry, false, t);
            float4 val, gx, gy;
            tap4_lerp(t, val, gx, gy);
            const float pd = c.es * val.w, cd = g.Z, sum = cd + pd, isum = frcp(sum);
            const float dif = D.plain_dif ? cd - pd : dc_diff(c, g, t, depth_t[gi], pd), raw = fabsf(dif) * isum;
            if (raw >= 0.f && raw <= 1.f) {
                const float sg = dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f);
                const float ddd = -sg * 2.f * cd * isum * isum * c.es;                  // d dd / d (sampled depth)
                const float dd = fminf(raw, 1.f);
                const float coef = (ratio_dc * fminf(1.f, dd * frcp(D.eps)) - (count ? w_photo * diff : 0.f)) * ddd;
                const int xi = u + (int)floorf(g.rx), yi = v + (int)floorf(g.ry);
                const float wx = t.wx, wy = t.wy;
                const float w4[4] = {(1.f - wx) * (1.f - wy), wx * (1.f - wy), (1.f - wx) * wy, wx * wy};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int xx = xi + (k & 1), yy = yi + (k >> 1);
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {         // (a tap in the zero border is no pixel of the target)
                        const long long a = (long long)llrint((double)(coef * w4[k]) * DREF_FIX);
                        const int wxl = xx - ox, wyl = yy - oy;
                        if (a != 0) {
                            if (wxl >= 0 && wxl < WW && wyl >= 0 && wyl < WH) atomicAdd(&win[wyl * WW + wxl], (unsigned long long)a);
                            else atomicAdd(reinterpret_cast<unsigned long long *>(ext + ((size_t)yy * W + xx) * 2 + slot), (unsigned long long)a);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < NWIN; i += DREF_TW * DREF_TH) {
        const unsigned long long a = win[i];
        const int yy = oy + i / WW, xx = ox + i % WW;
        if (a != 0ull && xx >= 0 && xx < W && yy >= 0 && yy < H) atomicAdd(reinterpret_cast<unsigned long long *>(ext + ((size_t)yy * W + xx) * 2 + slot), a);
    }
}


// The mirror image of k_dref_scatter for the SOURCE maps (tcsfm_linearize_dense_window_sources: the gradient of the reference's loss w.r.t.
// the depth maps the engine holds fixed; oracle: dref_source_depth_gradient).  Forward pair m = (s, b) samples source map m (stn.py:271):
// through its depth-consistency term and through the weight map it provides -- source 0's map multiplies EVERY selected pixel under argmin
// (optimizer.py:69), a pair's own pixels otherwise.  d L / d pd(p) = (b_dc h(dd) - a_f E(p)) d dd / d pd, E = the sum of M diff over the
// pixels this weight multiplies, scattered over the four taps in units of a_f = c_f / K_f (b_dc when no pixel counts).  A verification
// path, not a timed one: one thread per pixel, one 64-bit fixed-point atomic per tap (order-independent: bit-reproducible).
__global__ __launch_bounds__(256) void k_dref_scatter_src(LinParams P, DrefPrepassParams D, long long *ext_src /* [SB][H*W][2] */, float c_f) {
    const int H = P.H, W = P.W, hw = H * W;
    const int m = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int b = m % D.B, s = m / D.B, v = idx / W, u = idx - v * W;
    const PairConst &c = P.pc[m];
    const float dep = P.depth_t[(size_t)m * hw + idx];
    Geo g;
    warp_geo(c, W, H, u, v, dep, g);
    if (g.oobx || g.ooby) return;
    Tap t;
    tap4_fetch(P.srcpack + (size_t)m * (H + 2) * (W + 2), W, H, u, v, g.rx, g.ry, false, t);
    float4 val, gx, gy;
    tap4_lerp(t, val, gx, gy);
    const float pd = c.es * val.w, cd = g.Z, isum = frcp(cd + pd);
    // the sign of cd - pd is a decision the FORWARD group's kernel (k_dense_joint) takes on the plain fp32 difference (exactly zero is not rare
    // there: sg = 0); the adjoint of the same term must take the same one -- dc_diff's cancellation-free value would call a pixel with
    // |cd - pd| / (cd + pd) ~ 1e-8 negative where the group's own pose / map gradient used zero (scripts/diag/free_source_pixel.py)
    const float dif = cd - pd, raw = fabsf(dif) * isum;
    if (!(raw >= 0.f && raw <= 1.f)) return;
    const float sg = dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f);
    const float ddd = -sg * 2.f * cd * isum * isum * c.es;                  // d dd / d (sampled depth)
    const float dd = fminf(raw, 1.f);
    float E = 0.f;
    if (!D.argmin || s == 0) {
        const int s_lo = D.argmin ? 0 : s, s_hi = D.argmin ? D.S : s + 1;
        for (int s2 = s_lo; s2 < s_hi; s2++) {
            const int n2 = s2 * D.B + b;
            const bool valid = P.ext_valid[(size_t)n2 * hw + idx] > 0.5f;
            const float diff = P.ext_diff[(size_t)n2 * hw + idx], ae = P.tgtpack[(size_t)n2 * hw + idx].w;
            const bool count = (D.argmin && D.S > 1) ? ext_selected(P, n2, idx, hw) : (valid && (!(D.automask && D.argmin) || diff < ae));   // (k_dref_count's rule)
            if (count) E += diff;
        }
    }
    // (K_f is known here: one combined sum per tap, in the slot whose factor the consumer applies -- see k_dref_scatter)
    const float Kf = (float)D.norms[0];
    const int slot = Kf > 0.f ? 1 : 0;
    const float coef = Kf > 0.f ? (E - (D.b_dc * Kf / c_f) * fminf(1.f, dd * frcp(D.eps))) * ddd : fminf(1.f, dd * frcp(D.eps)) * ddd;
    const int xi = u + (int)floorf(g.rx), yi = v + (int)floorf(g.ry);
    const float wx = t.wx, wy = t.wy;
    const float w4[4] = {(1.f - wx) * (1.f - wy), wx * (1.f - wy), (1.f - wx) * wy, wx * wy};
    long long *ext = ext_src + (size_t)m * hw * 2;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xx = xi + (k & 1), yy = yi + (k >> 1);
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const long long a = (long long)llrint((double)(coef * w4[k]) * DREF_FIX);
            if (a != 0) atomicAdd(reinterpret_cast<unsigned long long *>(ext + ((size_t)yy * W + xx) * 2 + slot), (unsigned long long)a);
        }
    }
}
// d loss / d rho_s = a_i x the per-pixel record of the joint kernel run on the inverse pairs (a_i = 0.25 / K_i)
__global__ __launch_bounds__(256) void k_dref_export_grho_src(const float *jrec, int jrec_stride, const int *norms, float *out, int hw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (idx >= hw) return;
    const float Ki = (float)norms[1];
    out[(size_t)m * hw + idx] = Ki > 0.f ? (0.25f / Ki) * jrec[((size_t)m * hw + idx) * jrec_stride] : 0.f;
}

// l_smooth prepass (optimizer.py:92-93; losses.py:43-61 get_smooth_loss): per target the mean of its sigmoid disparity and the value of its
// whole term  T_b = (1 / m) [w_x sum_x-edges e^{-|dI|} |d sigma| + w_y sum_y-edges ...],  m = mean + 1e-7 -- the joint kernel needs both
// before it can form the term's gradient (the mean-normalisation couples every pixel of the image to every edge).  One workgroup per
// target, fp64 accumulation, fixed-order tree: bit-reproducible.
struct DrefSmoothParams {
    const float *depth;       // [.][H*W]: slot b = the target's current map
    const float4 *tgtpack;    // [.][H*W]: slot b = the target's colours (forward pair (0, b))
    float *out;               // [B][2]: m, T_b
    int H, W;
    float sig_lo, sig_ir, wx, wy;
};
__global__ __launch_bounds__(1024) void k_dref_smooth(DrefSmoothParams P) {
    const int b = blockIdx.x, tid = threadIdx.x, hw = P.H * P.W;
    const float *d = P.depth + (size_t)b * hw;
    const float4 *t = P.tgtpack + (size_t)b * hw;
    __shared__ double r0[1024], r1[1024], r2[1024];
    double s = 0.0, sx = 0.0, sy = 0.0;
    for (int i = tid; i < hw; i += 1024) {
        const int v = i / P.W, u = i - v * P.W;
        const float sg = (frcp(d[i]) - P.sig_lo) * P.sig_ir;
        const float4 c = t[i];
        s += (double)sg;
        if (u + 1 < P.W) {
            const float4 q = t[i + 1];
            const float gi_ = (fabsf(c.x - q.x) + fabsf(c.y - q.y) + fabsf(c.z - q.z)) * (1.f / 3.f);
            sx += (double)(__expf(-gi_) * fabsf(sg - (frcp(d[i + 1]) - P.sig_lo) * P.sig_ir));
        }
        if (v + 1 < P.H) {
            const float4 q = t[i + P.W];
            const float gi_ = (fabsf(c.x - q.x) + fabsf(c.y - q.y) + fabsf(c.z - q.z)) * (1.f / 3.f);
            sy += (double)(__expf(-gi_) * fabsf(sg - (frcp(d[i + P.W]) - P.sig_lo) * P.sig_ir));
        }
    }
    r0[tid] = s; r1[tid] = sx; r2[tid] = sy;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { r0[tid] += r0[tid + o]; r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double m = r0[0] / (double)hw + 1e-7;
        P.out[2 * b] = (float)m;
        P.out[2 * b + 1] = (float)(((double)P.wx * r1[0] + (double)P.wy * r2[0]) / m);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's PARAMETRISATION of optimize_depth_pred (optimizer.py:194-198, 235-239; TCSFM_DEPTH_QUARTER): the unknown of a target
// is its QUARTER-resolution map; every linearisation sees its x4 bilinear upsampling (F.interpolate, align_corners = False).  Inverse
// depth and sigmoid disparity are affine in each other and the interpolation weights of a pixel sum to one: rho = U rho_q.
// Mirrors orc_refine_dense_ref_q of the oracle (which is pinned on the reference's F.interpolate outputs and autograd, golden G13 `qinit`).
//   k_qres_init      rho_q = the /4 bilinear downsampling of the input map (the mean of every 4x4 block's 2x2 centre)
//   k_qres_upsample  rho = U rho_q -> depth, into every forward pair's depth slot and the inverse pairs' source packs
//   k_qres_schur     after k_dense_joint (which leaves the pixel records g, D, B and eliminates nothing): one WAVE per cell, one lane per
//                    pixel of the cell's 8x8 footprint -- g_c = sum U g, D_c = sum U D (the row-sum lumping of U' diag(D) U, which it
//                    majorises: the cell block stays diagonal), B_c = sum U B -- the cell record, and the cell's Schur terms
//                    -B_c B_c' / D_c, -B_c g_c / D_c accumulated per workgroup into records k_solve_joint sums beside the tiles'
//   k_qres_step      rho_q += -(g_c + B_c' dxi) / D_c through the pixels' trust region (depth_step)
struct QresParams {
    const float *jrec;        // [B][H*W][JREC] pixel records of k_dense_joint
    float *qrec;              // [B][nq][JREC] cell records
    float *rho_q;             // [B][nq]
    float *jblockrec;         // [B][rec_stride][NACC]: this kernel's records start at rec_first
    const double *delta;      // [B][6 JMAXS]
    float *depth;             // [.][H*W] slots of the forward pairs n = s B + b
    float4 *srcpack_inv;      // packs of the inverse pairs (channel w = the target depth they sample)
    int *norms_zero;          // the batch counters (norms_n of them), zeroed for the next linearisation (or null)
    int norms_n;
    float *rho_q_next;        // k_qres_step_up: the new cell values (ping-pong: the launch reads rho_q of neighbouring cells while it writes)
    float *depth_out;         // k_qres_step_up: optional second destination of the forward slots (the caller's buffer: last iteration), or null
    int c_ncall, c_B;         //   ... coalesced calls: per call (as JointUpdateParams), used when c_ncall > 0
    float *c_depth_out[TC_MAX_COAL];
    int H, W, B, S, rec_stride, rec_first;
    float rho_lo, rho_hi;
};
constexpr int QRES_CELLS_PER_WG = 16;      // 4 waves x 4 cells: the cells of a wave are loaded together (a first version walked 16 cells per wave
                                           // one dependent load after the other: 25 us per linearisation at 240x320)

// weight of cell c in the x4 upsampling taps of pixel x (torch area_pixel_compute_source_index, align_corners = False)
__device__ __forceinline__ float up4_weight(int x, int c, int nq) {
    float s = ((float)x + 0.5f) * 0.25f - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int i0 = (int)s, i1 = i0 + 1 < nq ? i0 + 1 : nq - 1;
    const float l1 = s - (float)i0;
    return (i0 == c ? 1.f - l1 : 0.f) + (i1 == c ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void k_qres_init(QresParams P) {
    const int h = P.H / 4, w = P.W / 4, nq = h * w;
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= nq) return;
    const int cy = c / w, cx = c - cy * w;
    const float *d = P.depth + (size_t)b * P.H * P.W + (size_t)(4 * cy + 1) * P.W + 4 * cx + 1;      // slot of forward pair (0, b)
    P.rho_q[(size_t)b * nq + c] = 0.5f * (0.5f / d[0] + 0.5f / d[1]) + 0.5f * (0.5f / d[P.W] + 0.5f / d[P.W + 1]);
}

__global__ __launch_bounds__(256) void k_qres_upsample(QresParams P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, hw = P.H * P.W;
    if (P.norms_zero && idx < P.norms_n && b == 0) P.norms_zero[idx] = 0;
    if (idx >= hw) return;
    const int h = P.H / 4, w = P.W / 4;
    const int v = idx / P.W, u = idx - v * P.W;
    float sy = ((float)v + 0.5f) * 0.25f - 0.5f, sx = ((float)u + 0.5f) * 0.25f - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, y1 = y0 + 1 < h ? y0 + 1 : h - 1, x0 = (int)sx, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float *q = P.rho_q + (size_t)b * h * w;
    const float rho = (1.f - ly) * ((1.f - lx) * q[y0 * w + x0] + lx * q[y0 * w + x1]) + ly * ((1.f - lx) * q[y1 * w + x0] + lx * q[y1 * w + x1]);
    const float dep = 1.f / rho;
    for (int s = 0; s < P.S; s++) {
        P.depth[(size_t)(s * P.B + b) * hw + idx] = dep;
        P.srcpack_inv[((size_t)(s * P.B + b) * (P.H + 2) + v + 1) * (P.W + 2) + u + 1].w = dep;
    }
}

template <int NS>
__device__ __forceinline__ void qres_schur_body(const QresParams &P, const int bx, const int b) {
    using JL = JointLayout<NS>;
    constexpr int NV = 2 + 6 * NS, NE = JL::NHJ + JL::NP, EPL = (NE + 63) / 64;
    __shared__ float cellv[4][32];
    __shared__ float wacc[4][EPL * 64];
    const int h = P.H / 4, w = P.W / 4, nq = h * w;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int dy = lane >> 3, dx = lane & 7;
    // this lane's entries of the Schur sums: e < NHJ = the lower-triangle entry (r, c) of the pose block, then the right-hand side's
    int er[EPL], ec[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        const int e = lane + 64 * k;
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) r++;
        er[k] = e < JL::NHJ ? r : e - JL::NHJ; ec[k] = e < JL::NHJ ? e - r * (r + 1) / 2 : -1;
    }
    float acc[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) acc[k] = 0.f;
    constexpr int CPW = QRES_CELLS_PER_WG / 4;
    const int c_first = (bx * 4 + wave) * CPW;
    // all CPW cells' pixel records are requested before the first is used (unconditional loads from clamped addresses; weight 0 outside)
    float r[CPW][JL::JREC], wt[CPW];
#pragma unroll
    for (int ci = 0; ci < CPW; ci++) {
        const int c = c_first + ci < nq ? c_first + ci : nq - 1;
        const int cy = c / w, cx = c - cy * w;
        const int py = 4 * cy - 2 + dy, px = 4 * cx - 2 + dx;
        const bool in = c_first + ci < nq && py >= 0 && py < P.H && px >= 0 && px < P.W;
        const int qy = py < 0 ? 0 : (py >= P.H ? P.H - 1 : py), qx = px < 0 ? 0 : (px >= P.W ? P.W - 1 : px);
        wt[ci] = in ? up4_weight(py, cy, h) * up4_weight(px, cx, w) : 0.f;
        const float4 *rt = reinterpret_cast<const float4 *>(P.jrec + ((size_t)b * P.H * P.W + (size_t)qy * P.W + qx) * JL::JREC);
#pragma unroll
        for (int k = 0; k < JL::JREC / 4; k++) { const float4 q = rt[k]; r[ci][4 * k] = q.x; r[ci][4 * k + 1] = q.y; r[ci][4 * k + 2] = q.z; r[ci][4 * k + 3] = q.w; }
    }
#pragma unroll
    for (int ci = 0; ci < CPW; ci++) {
        const int c = c_first + ci;
        if (c >= nq) break;                                  // (wave-uniform)
        float v[NV];
        const float wq = r[ci][1] > 0.f ? wt[ci] : 0.f;       // (pixels the full-resolution mode freezes contribute nothing)
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = wq * r[ci][j];
        wave_reduce_store<NV>(v, cellv[wave], lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float g = cellv[wave][0], Dd = cellv[wave][1];
        const bool on = Dd > 1e-30f;
        const float iD = on ? 1.f / Dd : 0.f;
        if (lane < JL::JREC) P.qrec[((size_t)b * nq + c) * JL::JREC + lane] = lane == 1 ? (on ? Dd : 0.f) : (lane < NV ? cellv[wave][lane] : 0.f);
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            const int e = lane + 64 * k;
            if (e < NE) acc[k] -= cellv[wave][2 + er[k]] * (ec[k] >= 0 ? cellv[wave][2 + ec[k]] : g) * iD;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // (cellv is rewritten by the next cell)
    }
#pragma unroll
    for (int k = 0; k < EPL; k++) wacc[wave][lane + 64 * k] = acc[k];
    __syncthreads();
    float *rec = P.jblockrec + ((size_t)b * P.rec_stride + P.rec_first + bx) * JL::NACC;
    for (int e = tid; e < JL::NACC; e += 256)
        rec[e] = e < NE ? (wacc[0][e] + wacc[1][e]) + (wacc[2][e] + wacc[3][e]) : 0.f;
}

template <int NS>
__global__ __launch_bounds__(256) void k_qres_schur(QresParams P) { qres_schur_body<NS>(P, blockIdx.x, blockIdx.y); }
// ... of the targets' cells (rows [0, Pa.B)) and of the source maps' cells (free source maps; rows behind them) in one launch
template <int NS>
__global__ __launch_bounds__(256) void k_qres_schur2(QresParams Pa, QresParams Pb) {
    if ((int)blockIdx.y < Pa.B) qres_schur_body<NS>(Pa, blockIdx.x, blockIdx.y);
    else qres_schur_body<1>(Pb, blockIdx.x, (int)blockIdx.y - Pa.B);
}

template <int NS>
__global__ __launch_bounds__(256) void k_qres_step(QresParams P) {
    using JL = JointLayout<NS>;
    const int nq = (P.H / 4) * (P.W / 4);
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= nq) return;
    const float4 *rt = reinterpret_cast<const float4 *>(P.qrec + ((size_t)b * nq + c) * JL::JREC);
    float r[JL::JREC];
#pragma unroll
    for (int k = 0; k < JL::JREC / 4; k++) { const float4 q = rt[k]; r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w; }
    if (!(r[1] > 0.f)) return;
    float bd = 0.f;
#pragma unroll
    for (int j = 0; j < 6 * NS; j++) bd += r[2 + j] * (float)P.delta[b * 6 * JMAXS + j];
    float *q = P.rho_q + (size_t)b * nq + c;
    *q = depth_step(*q, -(r[0] + bd) / r[1], P.rho_lo, P.rho_hi);
}

// k_qres_step + k_qres_upsample as ONE launch (round 5).  A workgroup owns a 32 x 8 tile of pixels = 8 x 2 cells; the x4 bilinear
// upsampling of its pixels reads the 10 x 4 cells around them: the first 40 threads take the step of one of those cells each (the same
// arithmetic as k_qres_step: a cell is stepped by up to four workgroups, every time to the same bits), the new values go to LDS and -- the
// workgroup's OWN 8 x 2 cells only -- to rho_q_next (ping-pong: neighbouring workgroups still read rho_q); then every pixel interpolates
// from LDS and writes its depth into the forward pairs' slots and the inverse pairs' packs, as k_qres_upsample.
constexpr int QSU_TW = 32, QSU_TH = 8, QSU_CW = QSU_TW / 4 + 2, QSU_CH = QSU_TH / 4 + 2;
template <int NS>
__device__ __forceinline__ void qres_step_up_body(const QresParams &P, const int bx, const int b) {
    using JL = JointLayout<NS>;
    __shared__ float rq[QSU_CH * QSU_CW];
    const int h = P.H / 4, w = P.W / 4, nq = h * w, hw = P.H * P.W;
    const int tiles_x = (P.W + QSU_TW - 1) / QSU_TW;
    const int tyi = bx / tiles_x, txi = bx - tyi * tiles_x;
    const int tid = threadIdx.x;
    const int cx0 = txi * (QSU_TW / 4), cy0 = tyi * (QSU_TH / 4);
    if (P.norms_zero && bx == 0 && b == 0 && tid < P.norms_n) P.norms_zero[tid] = 0;
    if (tid < QSU_CH * QSU_CW) {
        const int ly = tid / QSU_CW, lx = tid - ly * QSU_CW;
        const int cyr = cy0 - 1 + ly, cxr = cx0 - 1 + lx;
        const int cy = min(max(cyr, 0), h - 1), cx = min(max(cxr, 0), w - 1), c = cy * w + cx;
        const float4 *rt = reinterpret_cast<const float4 *>(P.qrec + ((size_t)b * nq + c) * JL::JREC);
        float r[JL::JREC];
#pragma unroll
        for (int k = 0; k < JL::JREC / 4; k++) { const float4 q = rt[k]; r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w; }
        float v = P.rho_q[(size_t)b * nq + c];
        if (r[1] > 0.f) {
            float bd = 0.f;
#pragma unroll
            for (int j = 0; j < 6 * NS; j++) bd += r[2 + j] * (float)P.delta[b * 6 * JMAXS + j];
            v = depth_step(v, -(r[0] + bd) / r[1], P.rho_lo, P.rho_hi);
        }
        rq[tid] = v;
        if (cyr == cy && cxr == cx && ly >= 1 && ly <= QSU_TH / 4 && lx >= 1 && lx <= QSU_TW / 4) P.rho_q_next[(size_t)b * nq + c] = v;      // an own cell, inside the map
    }
    __syncthreads();
    const int oy = tid / QSU_TW, ox = tid - oy * QSU_TW;
    const int u = txi * QSU_TW + ox, v = tyi * QSU_TH + oy;
    if (u >= P.W || v >= P.H) return;
    float sy = ((float)v + 0.5f) * 0.25f - 0.5f, sx = ((float)u + 0.5f) * 0.25f - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, y1 = y0 + 1 < h ? y0 + 1 : h - 1, x0 = (int)sx, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float ly_ = sy - (float)y0, lx_ = sx - (float)x0;
    const int ry0 = (y0 - cy0 + 1) * QSU_CW, ry1 = (y1 - cy0 + 1) * QSU_CW, rx0 = x0 - cx0 + 1, rx1 = x1 - cx0 + 1;
    const float rho = (1.f - ly_) * ((1.f - lx_) * rq[ry0 + rx0] + lx_ * rq[ry0 + rx1]) + ly_ * ((1.f - lx_) * rq[ry1 + rx0] + lx_ * rq[ry1 + rx1]);
    const float dep = 1.f / rho;
    const int idx = v * P.W + u;
    for (int s = 0; s < P.S; s++) {
        P.depth[(size_t)(s * P.B + b) * hw + idx] = dep;
        if (P.c_ncall > 0) P.c_depth_out[b / P.c_B][(size_t)(s * P.c_B + b % P.c_B) * hw + idx] = dep;
        else if (P.depth_out) P.depth_out[(size_t)(s * P.B + b) * hw + idx] = dep;
        P.srcpack_inv[((size_t)(s * P.B + b) * (P.H + 2) + v + 1) * (P.W + 2) + u + 1].w = dep;
    }
}
template <int NS>
__global__ __launch_bounds__(QSU_TW * QSU_TH) void k_qres_step_up(QresParams P) { qres_step_up_body<NS>(P, blockIdx.x, blockIdx.y); }
// ... for the targets' cells and the source maps' cells (free source maps) in one launch
template <int NS>
__global__ __launch_bounds__(QSU_TW * QSU_TH) void k_qres_step_up2(QresParams Pa, QresParams Pb) {
    if ((int)blockIdx.y < Pa.B) qres_step_up_body<NS>(Pa, blockIdx.x, blockIdx.y);
    else qres_step_up_body<1>(Pb, blockIdx.x, (int)blockIdx.y - Pa.B);
}

}  // namespace tc
