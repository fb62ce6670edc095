// dense_ref_kernel.h -- dense mode on the REFERENCE's own loss (round 4): compute_optimization_loss as optimize_depth_pred minimises it
// (optimizer.py:47-90, 194-198, 235-247), mirrors linearize_dense_ref / orc_refine_dense_ref of the oracle.
//
//   L = c_f / K_f  sum M_s W_x diff_s  +  0.25 / K_i sum M_i W_i diff_i  +  w_dc / (S B HW) sum (dd_fwd + dd_inv)  +  w_init / (B HW) sum SSIM(sigma, sigma0)
//
// Unknowns: the poses of all 2 S B directed pairs and ONE inverse-depth map per target.  Per linearisation, on one stream:
//   k_linearize<MODE_MAPS>   residual maps (diff, valid) of all 2 S B pairs at the current poses / depth            (kernels.h)
//   k_dref_prepass           forward pairs: the batch-summed count K_f of the min-over-sources selection; inverse pairs: their count K_i
//                            and the ADJOINT of their bilinear samples of the target depth -- d L / d pd of every inverse pixel scattered
//                            onto its four taps (64-bit fixed-point atomics: integer addition is order-independent, the result is
//                            bit-reproducible)
//   k_linearize<6, DC> + k_solve<6> (window rule REFERENCE)   the inverse pairs' 6 x 6 pose systems                  (kernels.h)
//   k_dense_joint<S, .., REF>  the forward group: source 0's weight map on every selected pixel with its cross term, depth consistency
//                            with its inverse-depth column, the SSIM prior, the scattered sums; per-pixel Schur elimination  (joint_kernel.h)
//   k_solve_joint<S>, k_dense_joint_update<S>   6S x 6S solve, back-substitution; the new map also goes into the inverse pairs' packs
#pragma once
#include "joint_kernel.h"

namespace tc {

struct DrefPrepassParams {
    const float *diff, *valid;    // [2SB][H*W] residual maps of all pairs (k_linearize<MODE_MAPS>)
    int *norms;                   // [2] K_f, K_i (zeroed before the launch)
    long long *ext;               // [B][H*W][2] fixed-point scatter sums (zeroed before the launch): sum M diff ddd w_tap, sum h ddd w_tap
    int B, S, argmin, automask;
    float eps;
};

// one thread per pixel of one directed pair; the LinParams carry the packs, the pair constants and (ext_*) the maps for the selection
__global__ __launch_bounds__(256) void k_dref_prepass(LinParams P, DrefPrepassParams D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    const int H = P.H, W = P.W, hw = H * W, SB = D.S * D.B;
    const bool live = idx < hw;
    const int gi = live ? idx : 0;
    bool count = false;
    if (n < SB) {               // forward pair: does it keep the pixel (min over the sources / own validity)?
        if (live) count = (D.argmin && D.S > 1) ? ext_selected(P, n, gi, hw)
                                                : (P.ext_valid[(size_t)n * hw + gi] > 0.5f &&
                                                   (!(D.automask && D.argmin) || P.ext_diff[(size_t)n * hw + gi] < P.tgtpack[(size_t)n * hw + gi].w));
    } else if (live) {          // inverse pair: own mask (validity x auto-mask) and the adjoint of its depth sample
        const int m = n - SB, b = m % D.B;
        const float diff = P.ext_diff[(size_t)n * hw + gi];
        const bool valid = P.ext_valid[(size_t)n * hw + gi] > 0.5f;
        count = valid && (!D.automask || diff < P.tgtpack[(size_t)n * hw + gi].w);
        if (valid) {
            const int v = gi / W, u = gi - v * W;
            const PairConst &c = P.pc[n];
            Geo g;
            warp_geo(c, W, H, u, v, P.depth_t[(size_t)n * hw + gi], g);
            Tap t;
            tap4_fetch(P.srcpack + (size_t)n * (H + 2) * (W + 2), W, H, u, v, g.rx, g.ry, false, t);
            float4 val, gx, gy;
            tap4_lerp(t, val, gx, gy);
            const float pd = c.es * val.w, cd = g.Z, sum = cd + pd, isum = frcp(sum);
            const float dif = dc_diff(c, g, t, P.depth_t[(size_t)n * hw + gi], pd), raw = fabsf(dif) * isum;
            if (raw >= 0.f && raw <= 1.f) {
                const float sg = dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f);
                const float ddd = -sg * 2.f * cd * isum * isum * c.es;                  // d dd / d (sampled depth)
                const float dd = fminf(raw, 1.f);
                const float e1 = count ? diff * ddd : 0.f, e2 = fminf(1.f, dd * frcp(D.eps)) * ddd;
                // the four taps in bordered coordinates (tap4_fetch); a tap in the zero border is no pixel of the target
                const float fx = floorf(g.rx), fy = floorf(g.ry);
                const int xi = u + (int)fx, yi = v + (int)fy;
                const float wx = t.wx, wy = t.wy;
                const float w4[4] = {(1.f - wx) * (1.f - wy), wx * (1.f - wy), (1.f - wx) * wy, wx * wy};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int xx = xi + (k & 1), yy = yi + (k >> 1);
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                        long long *e = D.ext + ((size_t)b * hw + (size_t)yy * W + xx) * 2;
                        const long long a1 = (long long)llrint((double)(e1 * w4[k]) * DREF_FIX), a2 = (long long)llrint((double)(e2 * w4[k]) * DREF_FIX);
                        if (a1 != 0) atomicAdd(reinterpret_cast<unsigned long long *>(e), (unsigned long long)a1);
                        if (a2 != 0) atomicAdd(reinterpret_cast<unsigned long long *>(e + 1), (unsigned long long)a2);
                    }
                }
            }
        }
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(count);
    if ((threadIdx.x & 63) == 0 && bal != 0ull) atomicAdd(D.norms + (n < SB ? 0 : 1), __builtin_popcountll(bal));
}

}  // namespace tc
