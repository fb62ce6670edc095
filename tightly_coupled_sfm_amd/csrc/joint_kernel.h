// joint_kernel.h -- JOINT dense mode: the S forward pairs of a target share ONE inverse-depth map (the reference's
// `optimize_depth_pred`: one disparity per frame that every term of the loss sees, optimizer.py:194-198,235-247).
// Mirrors linearize_joint / orc_refine_dense_joint of the oracle.
//
// Unknowns of target b: the S left SE(3) perturbations of its forward warps and rho_q = 1 / depth_t(q).  Cost: the forward term of
// the reference's loss (optimizer.py:47-73) + the masked depth prior of the dense mode.  Per pixel the depth block is a scalar and is
// eliminated; what is left is ONE reduced camera system of 6S x 6S per target (12 x 12 for the KITTI window):
//     S_ss' = [s == s'] H_ss - sum_p B_s(p) B_s'(p)' / Dd(p),   gS_s = g_s - sum_p B_s(p) g_rho(p) / Dd(p)
// Under the min over the sources every pixel counts for exactly one source, so the off-diagonal blocks vanish identically and are
// not accumulated (JointParams::argmin); without it the shared depth couples the poses.
//
// k_dense_joint        one workgroup = one 32x16 tile of ONE TARGET, looping over its S sources with the LDS staging of
//                      k_dense_linearize reused per source (phase 1: tile + 2-pixel halo warped with the SHARED depth map; phase 2a:
//                      residuals + adjoint coefficients on tile + 1-pixel halo; phase 2b: adjoint gather, this source's pose gradient
//                      and curvature block, its share of the depth gradient / curvature).  Per-pixel cross-source state (g_rho, D,
//                      mask count, weight terms) stays in registers, B_s goes straight into the pixel's back-substitution record;
//                      every source's sums are reduced over the workgroup inside the loop, the Schur terms after it.
// k_solve_joint        one workgroup per target: fp64 sum of the workgroup records, LM bookkeeping, (6S+1)-column Gauss-Jordan in
//                      LDS, SE(3) retraction of every source's transform, next PairConst of every forward pair.
// k_dense_joint_update back-substitution drho = -(g_rho + sum_s B_s' dxi_s) / Dd of the shared map, written to every forward pair's
//                      depth slot (the selection pass reads them per pair) and, at the end, to the caller.
#pragma once
#include "dense_kernel.h"

namespace tc {

constexpr int JMAXS = 3;                       // sources per target the joint kernels are instantiated for (2 .. JMAXS)
template <int NS> struct JointLayout {
    static constexpr int NP = 6 * NS;
    static constexpr int NHJ = NP * (NP + 1) / 2;
    static constexpr int OFF_G = NHJ, OFF_S = NHJ + NP, OFF_SHARE = OFF_S + 3, OFF_NM = OFF_SHARE + NS, NACC = OFF_NM + NS;
    static constexpr int JREC = (2 + 6 * NS + 3) / 4 * 4;     // floats per pixel record: g_rho, Dd, B[NS][6], padded to float4s
    __host__ __device__ static constexpr int tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
};

struct JointParams {
    float *jrec;             // [B][H*W][JREC]
    const float *depth0;     // [.][H*W] prior centre of target b at index b (the slot of forward pair (0, b))
    float *jblockrec;        // [B][nblk][NACC] one record per workgroup
    float lambda_depth, w_prior;
    int B, S, argmin;        // every pixel is weighted by the depth-consistency map of the source it counts for (see tcsfm.h)
    int automask;            // own masks (no argmin): optimizer.py:71-73 has no auto-mask there -> 0 from the host when S > 1
    // REF (k_dense_joint<.., REF = true>): the forward group under the reference's COMPLETE loss (optimizer.py:47-90, dense_ref_kernel.h):
    const int *norms;        // [groups][2] batch-summed mask counts K_f (forward selection) and K_i (inverse pairs) of THIS linearisation
                             // (k_linearize<FRONT> / k_dref_count); group of target b = b / norm_B (coalesced calls: one group per call; 0: one group)
    int norm_B;
    long long *ext2;         // [B][H*W][2] fixed-point (2^-40) adjoint sums of the inverse pairs' samples of the target depth: the depth-consistency
                             // part sum h(dd) ddd w and the photometric part sum M diff ddd w (k_linearize<FRONT> / k_dref_scatter); consumed AND zeroed here
    float c_f;               // factor on the forward term: 1 with the min over the sources, 0.25 without (:73)
    float b_dc;              // per-pixel weight of the depth-consistency terms: w_dc / (S B H W) (:83-86)
    float w_init_px;         // per-pixel weight of the SSIM prior between current and initial sigmoid disparity: w_init / (B H W) (:89-90)
    float sig_lo, sig_ir;    // sigmoid disparity = (rho - sig_lo) * sig_ir: 1 / max_depth and 1 / (1 / min_depth - 1 / max_depth)
    // quarter-resolution parametrisation (TCSFM_DEPTH_QUARTER, dense_ref_kernel.h): the depth is NOT eliminated per pixel here -- the
    // pixel records go to k_qres_schur, whose workgroup records follow this kernel's in the same per-target array
    int qres;
    int rec_stride;          // workgroup records per target in jblockrec (0: the tile count)
    // l_smooth (optimizer.py:92-93, losses.py:43-61): edge-aware smoothness of the mean-normalised sigmoid disparity of the target
    const float *smooth;     // [B][2] per target: m = mean(sigma) + 1e-7 and T_b = the target's whole term (k_dref_smooth), or null
    float w_smooth_x, w_smooth_y;      // weight / (B H (W-1)), weight / (B (H-1) W); 0: off
    // the kernel run on the INVERSE pairs as S = 1 groups (their back-projected depth = the source map: tcsfm_linearize_dense_window_sources):
    // the scattered sums then come from the FORWARD pairs' samples of that map, in units of ext_c / ext_norm[0] = the forward factor a_f
    const int *ext_norm;     // null: the photometric part of ext2 carries the inverse term's factor 0.25 / K_i; else ext_c / ext_norm[0]
    float ext_c;
    long long *dbg;          // diagnostic runs only (TCSFM_DEBUG_STAMPS=1): s_memrealtime stamps of the phases of workgroup (0, 0); null in production
};

// reduce N (<= 32) per-thread values over the workgroup and ADD them to LDS accumulators acc[slot(k)], k = 0..N-1
template <int N, int NT, class SlotFn>
__device__ __forceinline__ void joint_reduce_add(const float *v, float *red, float *acc, SlotFn slot, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    wave_reduce_store<N>(v, red + wave * 32, lane);
    __syncthreads();
    if (tid < N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) s += red[w * 32 + tid];
        acc[slot(tid)] += s;
    }
    __syncthreads();
}

// Waves per SIMD the register allocation aims at: 2 (256 VGPRs).  TC_JOINT_OCC=3 (168 VGPRs: three 256-thread workgroups per CU, which the 50 KB
// of LDS of the 32 x 8 tiling would allow) was measured: with S >= 2 the kernel then spills 90 registers and runs at HALF the speed (47 vs 26 us
// per launch, 226 vs 178 us per window at minibatch 6; profiles/r05_joint_occupancy_ab.txt).  S = 1 needs 153 registers and gets three waves anyway.
#ifndef TC_JOINT_OCC
#define TC_JOINT_OCC 2
#endif
#ifndef TC_JOINT_LEAN_PIPE
#define TC_JOINT_LEAN_PIPE 3
#endif
#ifndef TC_JOINT_NONREF_PIPE
#define TC_JOINT_NONREF_PIPE 1
#endif
// SM: the l_smooth term compiled in (off in the reference's drivers: the lean instantiation leaves its code and registers out).
// LEAN (NS = 2 on the 256-thread tiling without l_smooth -- the KITTI window of the mirror's default mode and of the library's own joint mode): the loop over the two sources is
// unrolled, so every `s == 0` / `s > 0` branch folds and the cross-source state that only one of the two bodies needs is not carried through
// the other; together with the Jacobian rebuilt in phase 2b (below) the kernel fits 168 VGPRs = THREE workgroups per CU (was 256 / two).
template <int NS, int NT, bool REF, bool SM> struct JointShape {
    static constexpr bool LEAN = NS == 2 && NT <= 256 && !SM;
    static constexpr int OCC = NT > 256 ? 2 : ((NS == 1 || LEAN) ? (TC_JOINT_OCC > 3 ? TC_JOINT_OCC : 3) : TC_JOINT_OCC);
    // software-pipelined window reads of phases 2a / 2b (as k_dense_linearize).  In the LEAN form, whose 168 registers are taken, they cost nine
    // more spilled registers and still win (joint launch 23.6 -> 22.5 us, minibatch-6 -1.6 %: profiles/r05_dense_pipe_ab.txt); TC_JOINT_LEAN_PIPE
    // (bit 0 = phase 2a, bit 1 = phase 2b) keeps the rolled loops there for A/B
    // The reference-loss instantiations that already fill their 256 registers (S = 3, or S = 2 with l_smooth) keep the rolled loops: pipelined they spill.
    static constexpr bool ROOM = NS == 1 || (!REF && TC_JOINT_NONREF_PIPE);
    static constexpr bool PIPE_A = TC_DENSE_PIPE_A && (ROOM || (LEAN && (TC_JOINT_LEAN_PIPE & 1)));
    static constexpr bool PIPE_B = TC_DENSE_PIPE_B && (ROOM || (LEAN && (TC_JOINT_LEAN_PIPE & 2)));
};
// LDS of one workgroup, carved from ONE array of float4 owned by the kernel: the two-role launch (k_dense_joint2) runs a body per role on the same storage
template <int NS, int TW, int TH, int NT, bool REF> struct JointSmem {
    static constexpr int N2 = (TW + 4) * (TH + 4), N1 = (TW + 2) * (TH + 2);
    static constexpr int OFF_AUX = N2 * 3, OFF_COEF = OFF_AUX + N2, OFF_RED = OFF_COEF + N1 * 3;                 // in float4
    static constexpr int F_RED = (NT / 64) * 32, F_ACC = (JointLayout<NS>::NACC + 3) / 4 * 4, F_W0 = REF ? (N2 + 3) / 4 * 4 : 4, F_SGQ = REF ? (2 * N2 + 3) / 4 * 4 : 4;      // in floats
    static constexpr int N4 = OFF_RED + (F_RED + F_ACC + F_W0 + F_SGQ) / 4;
};
template <int NS, int TW, int TH, int NT, bool TRACE, bool REF, bool SM>
__device__ __forceinline__ void dense_joint_body(const LinParams &P, const JointParams &J, const int by, float4 *smem) {
    using JL = JointLayout<NS>;
    using SMEM = JointSmem<NS, TW, TH, NT, REF>;
    constexpr int NP = 6;
    constexpr int W2 = TW + 4, H2 = TH + 4, N2 = W2 * H2;
    constexpr int W1 = TW + 2, H1 = TH + 2, N1 = W1 * H1;
    constexpr int NCEN = TW * TH;
    static_assert(NCEN == NT, "one tile pixel per thread");
    float4 *rec1 = smem;                                        // [N2 * 3]
    float4 *aux = smem + SMEM::OFF_AUX;                         // [N2]
    float4 *coef = smem + SMEM::OFF_COEF;                       // [N1 * 3]
    float *red = reinterpret_cast<float *>(smem + SMEM::OFF_RED);      // [(NT / 64) * 32]
    float *acc = red + SMEM::F_RED;                             // [NACC]
    float *w0 = acc + SMEM::F_ACC;                              // REF: depth-consistency weight of SOURCE 0 on tile + 2-pixel halo (optimizer.py:69)
    float *sgq = w0 + SMEM::F_W0;                               // REF: (sigma, sigma0) on tile + 2-pixel halo: the l_depth_init prior (optimizer.py:89-90)
    // REF: everything is accumulated in units of the forward term's factor a_f = c_f / K_f (k_solve_joint multiplies by it)
    float r_dc = 0.f, r_init = 0.f, r_eph = 0.f, iaf_ = 0.f;
    if (REF) {
        const int *nr = J.norms + 2 * (J.norm_B > 0 ? by / J.norm_B : 0);      // the normaliser group of this target
        const float Kf = (float)nr[0], Ki = (float)nr[1];
        const float iaf = Kf > 0.f ? Kf / J.c_f : 0.f;                 // 1 / a_f
        iaf_ = iaf;
        r_dc = J.b_dc * iaf; r_init = J.w_init_px * iaf;
        r_eph = (Ki > 0.f ? 0.25f / Ki : 0.f) * iaf;                   // factor of the scattered photometric sums (the inverse term's a_i) over a_f
        if (J.ext_norm) { const float Ke = (float)J.ext_norm[2 * (J.norm_B > 0 ? by / J.norm_B : 0)]; r_eph = (Ke > 0.f ? J.ext_c / Ke : 0.f) * iaf; }
    }
    const bool ref_w0 = REF && J.argmin;             // every source's pixels carry source 0's weight map
    const bool ref_prior = REF && J.w_init_px > 0.f;
    const bool ref_smooth = REF && SM && J.smooth != nullptr && (J.w_smooth_x > 0.f || J.w_smooth_y > 0.f);
    const bool ref_sig = ref_prior || ref_smooth;          // (sigma, sigma0) staged on tile + 2-pixel halo

    const int nblk = P.tiles_x * P.tiles_y;
    int bid = blockIdx.x;
    {
        int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int b = by;
    const int H = P.H, W = P.W, hw = H * W;
    const int tyi = bid / P.tiles_x, txi = bid - tyi * P.tiles_x;
    const int x00 = txi * TW, y00 = tyi * TH;
    const float *depth_t = P.depth_t + (size_t)b * hw;          // the SHARED map: slot of forward pair (0, b)
    const int tid = threadIdx.x;
    stamp_begin(P.stamp, tid);
#define TC_JSTAMP(i) if (J.dbg && tid == 0 && blockIdx.x == 0 && by == 0 && (i) < 8) J.dbg[i] = wall_clock64();
    TC_JSTAMP(0)
    for (int i = tid; i < JL::NACC; i += NT) acc[i] = 0.f;

    const int oy = tid / TW, ox = tid - oy * TW;
    const int gxo = x00 + ox, gyo = y00 + oy;
    const bool inimg = gxo < W && gyo < H;
    // the pixel's record (formed where it is used: a pointer held from here to the end of the kernel costs two registers through every phase)
    auto jrp = [&]() -> float * { return J.jrec + ((size_t)b * hw + (inimg ? gyo * W + gxo : 0)) * JL::JREC; };

    // per-pixel state across the sources
    float o_depth = 1.f, g_rho = 0.f, Dsum = 0.f, mcnt = 0.f;
    bool o_pad = false;      // some source's sample at this pixel is valid but blends with the zero padding: the pixel keeps its depth
    float ddJ0[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // REF: d dd_0 / d(xi_0, rho) at this pixel (the weight term of every source's pixels)
    float extra_cost = 0.f;                                // REF: depth-consistency and prior cost of this pixel (units of a_f)
    float o_sig = 0.f, o_sig0 = 0.f, o_pd1 = 1.f, o_pd2 = 1.f;   // REF prior: own sigmoid disparities and SSIM denominators
    bool o_pcl = true;                                     // REF prior: SSIM value clamped (no gradient / curvature)
    float o_sm_g = 0.f, o_sm_D = 0.f;                      // REF l_smooth: local gradient / diagonal curvature of this pixel w.r.t. sigma (true units)

    constexpr int NRING = N2 - NCEN;
    static_assert(NRING <= NT, "one ring round");
    constexpr int RING_THREADS = (NRING + 63) / 64 * 64;
    struct Stage { int lx, ly, px, py; float4 tp; float dep, dep0; Geo g; Tap t; };
    constexpr int SRC_UNROLL = JointShape<NS, NT, REF, SM>::LEAN ? 2 : 1;

#pragma unroll SRC_UNROLL
    for (int s = 0; s < NS; s++) {
        const int n = s * J.B + b;
        const PairConst &c = P.pc[n];
        const float4 *tgtpack = P.tgtpack + (size_t)n * hw;
        const float4 *srcpack = P.srcpack + (size_t)n * (H + 2) * (W + 2);
        float a[7], bb[7], zc[7];
        Geo o_g;                 // the own pixel's warp geometry: phase 2b rebuilds the 7-column Jacobian from it (16 registers not carried through phase 2a)
        float o_pd = 0.f, o_cd = 1.f, o_dgx = 0.f, o_dgy = 0.f;
        bool o_dcin = false;     // this source's projected-depth sample is a real depth sample (no zero padding in its footprint)

        // ---------------- phase 1: tile + 2-pixel halo, warped with the shared depth ----------------
        auto s_load = [&](Stage &S) {
            S.px = refl_idx(x00 + S.lx - 2, W); S.py = refl_idx(y00 + S.ly - 2, H);
            const int gi = S.py * W + S.px;
            S.tp = tgtpack[gi]; S.dep = depth_t[gi];
            S.dep0 = (REF && ref_sig && s == 0) ? J.depth0[(size_t)b * hw + gi] : 1.f;
        };
        auto s_warp = [&](Stage &S) {
            warp_geo(c, W, H, S.px, S.py, S.dep, S.g);
            tap4_fetch(srcpack, W, H, S.px, S.py, S.g.rx, S.g.ry, S.g.oobx || S.g.ooby, S.t);
        };
        auto s_store = [&](Stage &S, bool write, bool own) {
            float4 val, gx, gy;
            tap4_lerp(S.t, val, gx, gy);
            const bool oob = S.g.oobx || S.g.ooby;
            float pd = c.es * val.w, cd = S.g.Z;
            float Wt = 1.f - clamp01(fabsf(cd - pd) * frcp(cd + pd));
            if (write) {
                float Wuse = Wt;
                if (REF && ref_w0) {            // the same thread stages the same position for every source: no barrier needed
                    if (s == 0) w0[S.ly * W2 + S.lx] = Wt; else Wuse = w0[S.ly * W2 + S.lx];
                }
                if (REF && ref_sig && s == 0) {
                    sgq[2 * (S.ly * W2 + S.lx)] = (frcp(S.dep) - J.sig_lo) * J.sig_ir;
                    sgq[2 * (S.ly * W2 + S.lx) + 1] = (frcp(S.dep0) - J.sig_lo) * J.sig_ir;
                }
                float4 *rec = rec1 + (S.ly * W2 + S.lx) * 3;
                lds_write1(rec + 0, val.x, val.y, S.tp.x, S.tp.y);
                lds_write1(rec + 1, gx.x, gx.y, gy.x, gy.y);
                lds_write1(rec + 2, val.z, S.tp.z, gx.z, gy.z);
                lds_write1(aux + S.ly * W2 + S.lx, Wuse, oob ? 0.f : 1.f, S.tp.w, 0.f);
            }
            if (own) {
                if (TRACE && P.trace != nullptr && inimg)
                    P.trace[(size_t)n * hw + (size_t)S.py * W + S.px] =
                        (unsigned short)((((S.px + (int)floorf(S.g.rx)) & 1) << 2) | (((S.py + (int)floorf(S.g.ry)) & 1) << 3));
                o_g = S.g;
                o_pd = pd; o_cd = cd; o_dgx = c.es * gx.w; o_dgy = c.es * gy.w; o_depth = S.dep;
                o_pad = o_pad || (!oob && !S.t.inside);
                o_dcin = !oob && S.t.inside;
            }
        };
        {
            Stage A;
            A.lx = ox + 2; A.ly = oy + 2;
            if (tid < RING_THREADS) {
                Stage B;
                const int hi = min(tid, NRING - 1);
                if (hi < 2 * W2) { B.ly = hi / W2; B.lx = hi - B.ly * W2; }
                else if (hi < 4 * W2) { const int k = hi - 2 * W2; B.ly = H2 - 2 + k / W2; B.lx = k - (k / W2) * W2; }
                else { const int k = hi - 4 * W2; B.ly = 2 + (k >> 2); const int q = k & 3; B.lx = q < 2 ? q : W2 - 4 + q; }
                s_load(A); s_load(B);
                s_warp(A); s_warp(B);
                s_store(A, true, true); s_store(B, tid < NRING, false);
            } else {
                s_load(A); s_warp(A); s_store(A, true, true);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        TC_JSTAMP(1 + 3 * s)

        // ---------------- phase 2a: residual + adjoint coefficients for tile + 1-pixel halo ----------------
        float o_valid = 0.f, o_diff = 0.f, o_w = 0.f, o_m = 0.f, o_lxx = 0.f, o_lxy = 0.f, o_lyy = 0.f, o_l1x = 0.f, o_l1y = 0.f;
        float o_gx[3] = {0, 0, 0}, o_gy[3] = {0, 0, 0}, o_y[3] = {0, 0, 0}, o_x[3] = {0, 0, 0};
        constexpr int RA = (N1 + NT - 1) / NT;
#pragma unroll
        for (int r = 0; r < RA; r++) {
            int lx, ly;
            bool active = true;
            if (r == 0) { lx = ox + 1; ly = oy + 1; }
            else {
                int hi = tid + (r - 1) * NT;
                active = hi < N1 - NCEN;
                if (hi < W1) { ly = 0; lx = hi; }
                else if (hi < 2 * W1) { ly = H1 - 1; lx = hi - W1; }
                else { int k = hi - 2 * W1; ly = 1 + (k >> 1); lx = (k & 1) ? W1 - 1 : 0; }
            }
            if (!active) continue;
            const int gx_ = x00 + lx - 1, gy_ = y00 + ly - 1;
            const bool real = gx_ >= 0 && gx_ < W && gy_ >= 0 && gy_ < H;
            const float4 *ctr = rec1 + ((ly + 1) * W2 + lx + 1) * 3;
            f32x4 q0, q1, q2;
            lds_read3v(ctr, q0, q1, q2);
            const f2 yc01 = q0.lo, xc01 = q0.hi, gxc01 = q1.lo, gyc01 = q1.hi, yx2c = q2.lo, g2c = q2.hi;
            const float yc[3] = {yc01.x, yc01.y, yx2c.x}, xc[3] = {xc01.x, xc01.y, yx2c.y};
            const float gxc[3] = {gxc01.x, gxc01.y, g2c.x}, gyc[3] = {gyc01.x, gyc01.y, g2c.y};
            float4 ax = lds_read1(aux + (ly + 1) * W2 + lx + 1);
            f2 Sy01, Sx01, Syy01, Sxx01, Sxy01, Gx01, Gy01, S2, SS2, G2;
            float Sxy2;
            const float4 *nbA = ctr - (W2 + 1) * 3;
            auto first = [&](const f32x4 &n0, const f32x4 &n1, const f32x4 &n2) {
                Sy01 = pk_sub(n0.lo, yc01); Sx01 = pk_sub(n0.hi, xc01);
                Syy01 = Sy01 * Sy01; Sxx01 = Sx01 * Sx01; Sxy01 = Sx01 * Sy01;
                Gx01 = n1.lo; Gy01 = n1.hi;
                S2 = pk_sub(n2.lo, yx2c);
                SS2 = S2 * S2; Sxy2 = S2.x * S2.y; G2 = n2.hi;
            };
            auto more = [&](const f32x4 &n0, const f32x4 &n1, const f32x4 &n2) {
                f2 ey = pk_sub(n0.lo, yc01), ex = pk_sub(n0.hi, xc01);
                Sy01 += ey; Sx01 += ex; Syy01 += ey * ey; Sxx01 += ex * ex; Sxy01 += ex * ey;
                Gx01 += n1.lo; Gy01 += n1.hi;
                f2 e2v = pk_sub(n2.lo, yx2c);
                S2 += e2v; SS2 += e2v * e2v; Sxy2 += e2v.x * e2v.y; G2 += n2.hi;
            };
            if constexpr (JointShape<NS, NT, REF, SM>::PIPE_A) {   // software-pipelined window reads with scheduling fences, as in k_dense_linearize (dense_kernel.h): same operations, same order
                constexpr int RB = 48, ROWB = W2 * 48;
                const unsigned base = lds_addr(nbA);
                f32x4 u0, u1, u2, w0, w1, w2;
                auto fence = [&]() { asm volatile("" : "+v"(Sy01), "+v"(Sx01), "+v"(Syy01), "+v"(Sxx01), "+v"(Sxy01), "+v"(Gx01), "+v"(Gy01), "+v"(S2), "+v"(SS2), "+v"(Sxy2), "+v"(G2)); };
                lds_issue3c_at<0>(base, u0, u1, u2);
                lds_issue3c_at<RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); first(u0, u1, u2); fence(); lds_issue3c_at<2 * RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); more(w0, w1, w2); fence(); lds_issue3c_at<ROWB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); more(u0, u1, u2); fence(); lds_issue3c_at<ROWB + RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); more(w0, w1, w2); fence(); lds_issue3c_at<ROWB + 2 * RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); more(u0, u1, u2); fence(); lds_issue3c_at<2 * ROWB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); more(w0, w1, w2); fence(); lds_issue3c_at<2 * ROWB + RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); more(u0, u1, u2); fence(); lds_issue3c_at<2 * ROWB + 2 * RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); more(w0, w1, w2); fence();
                lds_waitn<0>(u0, u1, u2); more(u0, u1, u2);
            } else {
                {
                    f32x4 n0, n1, n2;
                    lds_read3v(nbA, n0, n1, n2);
                    nbA += 3;
                    first(n0, n1, n2);
                }
#pragma unroll 1
                for (int kk = 1; kk < 9; kk++) {
                    f32x4 n0, n1, n2;
                    lds_read3v(nbA, n0, n1, n2);
                    nbA += (kk == 2 || kk == 5) ? (W2 - 2) * 3 : 3;
                    more(n0, n1, n2);
                }
            }
            ChanTerms<f2> t01;
            ChanTerms<float> t2;
            ssim_l1_channel<f2>(xc01, yc01, gxc01, gyc01, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl, P.eps, t01);
            ssim_l1_channel<float>(yx2c.y, yx2c.x, g2c.x, g2c.y, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl, P.eps, t2);
            const float cB[3] = {t01.cB.x, t01.cB.y, t2.cB}, cC[3] = {t01.cC.x, t01.cC.y, t2.cC};
            float cA[3] = {t01.cA.x, t01.cA.y, t2.cA};
#pragma unroll
            for (int ch = 0; ch < 3; ch++) cA[ch] += cB[ch] * (0.5f - yc[ch]) + cC[ch] * (0.5f - xc[ch]);
            const float e1 = t01.e1.x + t01.e1.y + t2.e1, e2 = t01.e2.x + t01.e2.y + t2.e2;
            const float l1x = t01.l1x.x + t01.l1x.y + t2.l1x, l1y = t01.l1y.x + t01.l1y.y + t2.l1y;
            float lxx = t01.lxx.x + t01.lxx.y + t2.lxx, lxy = t01.lxy.x + t01.lxy.y + t2.lxy, lyy = t01.lyy.x + t01.lyy.y + t2.lyy;
            {
                const float n9 = 1.f / 9.f;
                const f2 mx = Gx01 * n9, my = Gy01 * n9, ex = gxc01 - mx, ey = gyc01 - my;
                const f2 qxx = t01.id2 * ex * ex + t01.id1 * mx * mx, qxy = t01.id2 * ex * ey + t01.id1 * mx * my,
                         qyy = t01.id2 * ey * ey + t01.id1 * my * my;
                const float mx2 = G2.x * n9, my2 = G2.y * n9, ex2 = g2c.x - mx2, ey2 = g2c.y - my2;
                lxx += qxx.x + qxx.y + t2.id2 * ex2 * ex2 + t2.id1 * mx2 * mx2;
                lxy += qxy.x + qxy.y + t2.id2 * ex2 * ey2 + t2.id1 * mx2 * my2;
                lyy += qyy.x + qyy.y + t2.id2 * ey2 * ey2 + t2.id1 * my2 * my2;
            }
            float diff = e1 + e2;
            float m = (real && ax.y > 0.5f && (!J.automask || diff < ax.z)) ? 1.f : 0.f;
            if (P.sel_in != nullptr)        // min over the sources, decided by k_linearize<FRONT> at this linearisation: does forward pair n keep this pixel?
                m = (real && P.sel_in[(size_t)n * hw + gy_ * W + gx_] > 0.5f) ? 1.f : 0.f;
            else if (P.ext_diff != nullptr) // ... or from the residual maps of a MODE_MAPS pass
                m = (real && ext_selected(P, n, gy_ * W + gx_, hw)) ? 1.f : 0.f;
            float w = m * ax.x;    // M_s W_x
            float pA = 0.f, pB = 0.f, pC = 0.f;
            if (REF && ref_prior && s == 0) {
                // l_depth_init: SSIM_Loss(sigma, sigma0) at this position (losses.py:27-41 on one channel, reflect-padded as the region is);
                // d s_p / d sigma_q = pA + pB (sigma_q - 1/2) + pC (sigma0_q - 1/2) for the nine q of its window
                const float *sc = sgq + 2 * ((ly + 1) * W2 + lx + 1);
                const float yc_ = sc[0], xc_ = sc[1];
                float Sy = 0.f, Sx = 0.f, Syy = 0.f, Sxx = 0.f, Sxy = 0.f;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        const float ey = sc[2 * (dy * W2 + dx)] - yc_, ex = sc[2 * (dy * W2 + dx) + 1] - xc_;
                        Sy += ey; Sx += ex; Syy += ey * ey; Sxx += ex * ex; Sxy += ex * ey;
                    }
                const float n9 = 1.f / 9.f;
                const float mdx = Sx * n9, mdy = Sy * n9, mux = xc_ + mdx, muy = yc_ + mdy;
                const float sigx = Sxx * n9 - mdx * mdx, sigy = Syy * n9 - mdy * mdy, sigxy = Sxy * n9 - mdx * mdy;
                const float n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sigxy + SSIM_C2;
                const float d1 = mux * mux + muy * muy + SSIM_C1, d2 = sigx + sigy + SSIM_C2;
                const float idn = frcp(d1 * d2), ratio = n1 * n2 * idn, raw = (1.f - ratio) * 0.5f;
                const bool cl = raw > 1.f;      // (the lower clamp is rounding only -- SSIM <= 1 -- and sigma == sigma0 at the first linearisation: oracle)
                if (real && !cl) {
                    const float pre = idn * (-0.5f * n9);
                    pB = pre * (ratio * d1) * -2.f;
                    pC = pre * n1 * 2.f;
                    pA = pre * 2.f * (mux * n2 - ratio * muy * d2) - pB * mdy - pC * mdx + pB * (0.5f - yc_) + pC * (0.5f - xc_);
                }
                if (r == 0) {
                    o_sig = yc_; o_sig0 = xc_; o_pd1 = d1; o_pd2 = d2; o_pcl = cl;
                    if (real) extra_cost += r_init * clamp01(raw);
                }
            }
            if (REF && ref_smooth && s == 0 && r == 0 && real) {
                // l_smooth at the own pixel: for each of its (up to four) edges  c_e |d^_p - d^_q|,  d^ = sigma / m:  local gradient
                // c_e ((sigma_p - sigma_q) / m) / max(|.|, eps) / m  and the diagonal majoriser 2 c_e / (m^2 max(|.|, eps)); the per-image
                // constant of the normalisation is added after the sources (oracle: linearize_dense_ref)
                const float m_ = J.smooth[2 * b], im = frcp(m_);
                const int q0 = (ly + 1) * W2 + lx + 1;
                const float sp = sgq[2 * q0];
                const float4 c0 = rec1[q0 * 3], c2 = rec1[q0 * 3 + 2];
                const int dq[4] = {1, -1, W2, -W2};
                const bool ok4[4] = {gx_ + 1 < W, gx_ >= 1, gy_ + 1 < H, gy_ >= 1};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int q = q0 + dq[e];
                    const float4 n0 = rec1[q * 3], n2 = rec1[q * 3 + 2];
                    const float gi_ = (fabsf(c0.z - n0.z) + fabsf(c0.w - n0.w) + fabsf(c2.y - n2.y)) * (1.f / 3.f);
                    const float ce = (e < 2 ? J.w_smooth_x : J.w_smooth_y) * __expf(-gi_);
                    const float dd_ = (sp - sgq[2 * q]) * im, den = fmaxf(fabsf(dd_), P.eps);
                    const float t_ = ok4[e] ? ce * frcp(den) * im : 0.f;
                    o_sm_g += t_ * dd_;
                    o_sm_D += 2.f * t_ * im;
                }
            }
            float4 *cr = coef + (ly * W1 + lx) * 3;
            lds_write1(cr + 0, w * cA[0], w * cA[1], w * cA[2], w * cB[0]);
            lds_write1(cr + 1, w * cB[1], w * cB[2], w * cC[0], w * cC[1]);
            lds_write1(cr + 2, w * cC[2], pA, pB, pC);
            if (r == 0) {
                o_valid = ax.y;
                o_diff = diff; o_w = w; o_m = m; o_lxx = lxx; o_lxy = lxy; o_lyy = lyy; o_l1x = l1x; o_l1y = l1y;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { o_gx[ch] = gxc[ch]; o_gy[ch] = gyc[ch]; o_y[ch] = yc[ch]; o_x[ch] = xc[ch]; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        TC_JSTAMP(2 + 3 * s)

        // ---------------- phase 2b: adjoint gather; this source's gradient, curvature block and depth terms ----------------
        float v[29];      // H_ss (21, pre-Schur) | g_s (6) | share_s | n_mask_s
        float vx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 29; i++) v[i] = 0.f;
        if (inimg) {
            float sg;
            {
                float sum = o_cd + o_pd, dif = o_cd - o_pd, isum = frcp(sum), raw = fabsf(dif) * isum;
                sg = (raw >= 0.f && raw <= 1.f) ? (dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f)) : 0.f;
                if (TRACE && P.trace != nullptr) {
                    unsigned short *tb = P.trace + (size_t)n * hw + gyo * W + gxo;
                    *tb = (unsigned short)(*tb | (o_m > 0.f ? 1 : 0) | (o_valid > 0.5f ? 2 : 0) | (sign_code(dif) << 4) | (sign_code(o_y[0] - o_x[0]) << 6) |
                                           (sign_code(o_y[1] - o_x[1]) << 8) | (sign_code(o_y[2] - o_x[2]) << 10));
                }
                sg *= 2.f * isum * isum;                     // d dd / d theta = sg (pd zc - cd dpd)
            }
            const float mxl = (gxo == 1) ? 2.f : 1.f, mxr = (gxo == W - 2) ? 2.f : 1.f;
            const float myu = (gyo == 1) ? 2.f : 1.f, myd = (gyo == H - 2) ? 2.f : 1.f;
            const float yq[3] = {o_y[0] - 0.5f, o_y[1] - 0.5f, o_y[2] - 0.5f}, xq[3] = {o_x[0] - 0.5f, o_x[1] - 0.5f, o_x[2] - 0.5f};
            float sA[3] = {0, 0, 0}, sB[3] = {0, 0, 0}, sC[3] = {0, 0, 0};
            float sP[3] = {0, 0, 0};          // REF: the prior's coefficient sums (source 0's records carry them)
            auto gather = [&](const f32x4 &c0, const f32x4 &c1, const f32x4 &c2, const float fm) {
                sA[0] += fm * c0.x; sA[1] += fm * c0.y; sA[2] += fm * c0.z;
                sB[0] += fm * c0.w; sB[1] += fm * c1.x; sB[2] += fm * c1.y;
                sC[0] += fm * c1.z; sC[1] += fm * c1.w; sC[2] += fm * c2.x;
                if (REF) { sP[0] += fm * c2.y; sP[1] += fm * c2.z; sP[2] += fm * c2.w; }
            };
            if constexpr (JointShape<NS, NT, REF, SM>::PIPE_B) {
                constexpr int RB = 48, ROWB = W1 * 48;
                const unsigned base = lds_addr(coef + (oy * W1 + ox) * 3);
                f32x4 u0, u1, u2, w0, w1, w2;
                auto fence = [&]() {
                    asm volatile("" : "+v"(sA[0]), "+v"(sA[1]), "+v"(sA[2]), "+v"(sB[0]), "+v"(sB[1]), "+v"(sB[2]), "+v"(sC[0]), "+v"(sC[1]), "+v"(sC[2]));
                    if (REF) asm volatile("" : "+v"(sP[0]), "+v"(sP[1]), "+v"(sP[2]));
                };
                lds_issue3c_at<0>(base, u0, u1, u2);
                lds_issue3c_at<RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); gather(u0, u1, u2, mxl * myu); fence(); lds_issue3c_at<2 * RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); gather(w0, w1, w2, 1.f * myu); fence(); lds_issue3c_at<ROWB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); gather(u0, u1, u2, mxr * myu); fence(); lds_issue3c_at<ROWB + RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); gather(w0, w1, w2, mxl * 1.f); fence(); lds_issue3c_at<ROWB + 2 * RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); gather(u0, u1, u2, 1.f * 1.f); fence(); lds_issue3c_at<2 * ROWB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); gather(w0, w1, w2, mxr * 1.f); fence(); lds_issue3c_at<2 * ROWB + RB>(base, w0, w1, w2);
                lds_waitn<3>(u0, u1, u2); gather(u0, u1, u2, mxl * myd); fence(); lds_issue3c_at<2 * ROWB + 2 * RB>(base, u0, u1, u2);
                lds_waitn<3>(w0, w1, w2); gather(w0, w1, w2, 1.f * myd); fence();
                lds_waitn<0>(u0, u1, u2); gather(u0, u1, u2, mxr * myd);
            } else {
#pragma unroll 1
                for (int r = 0; r < 3; r++) {
                    const float fy = r == 0 ? myu : (r == 2 ? myd : 1.f);
                    const float4 *row = coef + ((oy + r) * W1 + ox) * 3;
#pragma unroll
                    for (int cx = 0; cx < 3; cx++) {
                        float4 c0, c1, c2;
                        lds_read3(row + cx * 3, c0, c1, c2);
                        const float fm = (cx == 0 ? mxl : (cx == 2 ? mxr : 1.f)) * fy;
                        gather(f32x4{c0.x, c0.y, c0.z, c0.w}, f32x4{c1.x, c1.y, c1.z, c1.w}, f32x4{c2.x, c2.y, c2.z, c2.w}, fm);
                    }
                }
            }
            geo_jac<7>(c, o_g, W, H, a, bb, zc);      // (rebuilt here, behind the coefficient gather: see o_g)
            a[6] *= -o_depth; bb[6] *= -o_depth; zc[6] *= -o_depth;      // scale column -> inverse-depth column
            float sx = o_w * o_l1x, sy = o_w * o_l1y;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float lam = sA[ch] + sB[ch] * yq[ch] + sC[ch] * xq[ch];
                sx += lam * o_gx[ch]; sy += lam * o_gy[ch];
            }
            // d dd_s / d(xi_s, rho) = sg (pd zc - cd dpd) of THIS source at this pixel
            float ddJ[7];
#pragma unroll
            for (int j = 0; j < 7; j++) ddJ[j] = sg * (o_pd * zc[j] - o_cd * (o_dgx * a[j] + o_dgy * bb[j]));
            if (REF && ref_w0 && s == 0) {
#pragma unroll
                for (int j = 0; j < 7; j++) ddJ0[j] = ddJ[j];
            }
            // weight term -M_s diff_s d dd_x / d theta: own source's map (x = s), or under the reference's rule source 0's (x = 0): for the
            // pixels of the other sources it then lands in source 0's pose gradient (cross, below) and in the shared depth
            const float kdd = o_m * o_diff;
            const bool wown = !(REF && ref_w0 && s > 0);
            float grow[7];
#pragma unroll
            for (int j = 0; j < 7; j++) grow[j] = sx * a[j] + sy * bb[j] - (wown ? kdd * ddJ[j] : 0.f);
            float cross[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!wown) {
#pragma unroll
                for (int j = 0; j < 6; j++) cross[j] = -kdd * ddJ0[j];
                grow[6] -= kdd * ddJ0[6];
            }
            const float wxx = o_w * o_lxx, wxy = o_w * o_lxy, wyy = o_w * o_lyy;
            float la[7], lb[7];
#pragma unroll
            for (int j = 0; j < 7; j++) { la[j] = wxx * a[j] + wxy * bb[j]; lb[j] = wxy * a[j] + wyy * bb[j]; }
            Dsum += la[6] * a[6] + lb[6] * bb[6];
            g_rho += grow[6];
            mcnt += o_m;
            float Bq[6];
#pragma unroll
            for (int j = 0; j < 6; j++) Bq[j] = la[j] * a[6] + lb[j] * bb[6];
            // REF: depth consistency of this forward pair (optimizer.py:83-86; every pixel of the image): Huberised gradient, IRLS
            // curvature over real depth samples only -- as k_linearize has it for the poses, here with the inverse-depth column
            float kdc = 0.f;
            if (REF && r_dc > 0.f) {
                const float dd = clamp01(fabsf(o_cd - o_pd) * frcp(o_cd + o_pd));
                const float inf = r_dc * fminf(1.f, dd * frcp(P.eps));
                kdc = o_dcin ? r_dc * frcp(fmaxf(dd, P.eps)) : 0.f;
                extra_cost += r_dc * dd;
#pragma unroll
                for (int j = 0; j < 7; j++) grow[j] += inf * ddJ[j];
                g_rho += inf * ddJ[6];
                Dsum += kdc * ddJ[6] * ddJ[6];
#pragma unroll
                for (int j = 0; j < 6; j++) Bq[j] += kdc * ddJ[j] * ddJ[6];
            }
            if (REF && ref_prior && s == 0) {      // the prior's gradient (adjoint of the 3x3 window) and its diagonal curvature model
                const float pl = sP[0] + sP[1] * (o_sig - 0.5f) + sP[2] * (o_sig0 - 0.5f);
                g_rho += r_init * J.sig_ir * pl;
                if (!o_pcl) Dsum += r_init * J.sig_ir * J.sig_ir * (frcp(o_pd2) + (1.f / 9.f) * frcp(o_pd1));
            }
            if (REF && ref_smooth && s == 0) {     // l_smooth: local part minus the per-image constant T_b / (m HW), in units of a_f
                const float iaf_s = iaf_;
                const float m_ = J.smooth[2 * b], Gb = J.smooth[2 * b + 1] * frcp(m_ * (float)hw);
                g_rho += iaf_s * J.sig_ir * (o_sm_g - Gb);
                Dsum += iaf_s * J.sig_ir * J.sig_ir * o_sm_D;
                if (bid == 0 && tid == 0) extra_cost += iaf_s * J.smooth[2 * b + 1];       // the term's value, booked once per target
            }
            // B_s goes straight into the pixel's record (read back for the Schur terms below and by the back-substitution)
            float *jr = jrp();
            jr[2 + 6 * s + 0] = Bq[0]; jr[2 + 6 * s + 1] = Bq[1]; jr[2 + 6 * s + 2] = Bq[2];
            jr[2 + 6 * s + 3] = Bq[3]; jr[2 + 6 * s + 4] = Bq[4]; jr[2 + 6 * s + 5] = Bq[5];
            int h = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                v[21 + j] = grow[j];
#pragma unroll
                for (int i = 0; i <= j; i++) { v[h] = la[j] * a[i] + lb[j] * bb[i] + ((REF) ? kdc * ddJ[j] * ddJ[i] : 0.f); h++; }
            }
            v[27] = o_w * o_diff;
            v[28] = o_m;
            if (REF && ref_w0 && s > 0) {
#pragma unroll
                for (int j = 0; j < 6; j++) vx[j] = cross[j];
            }
        }
        joint_reduce_add<29, NT>(v, red, acc, [&](int k) {
            if (k >= 27) return k == 27 ? JL::OFF_SHARE + s : JL::OFF_NM + s;
            if (k >= 21) return JL::OFF_G + 6 * s + (k - 21);
            int j = 0;                                       // k = j (j + 1) / 2 + i
            while ((j + 1) * (j + 2) / 2 <= k) j++;
            return JL::tri(6 * s + j, 6 * s + (k - j * (j + 1) / 2));
        }, tid);          // (its barriers also protect rec1 / aux / coef for the next source)
        if (REF && ref_w0 && s > 0)          // the cross term: source 0's weight map on this source's pixels -> source 0's pose gradient
            joint_reduce_add<6, NT>(vx, red, acc, [&](int k) { return JL::OFF_G + k; }, tid);
        TC_JSTAMP(3 + 3 * s)
    }

    // ---------------- after the sources: prior, per-pixel Schur elimination of the shared depth ----------------
    float iD = 0.f, Bv[NS][6];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int j = 0; j < 6; j++) Bv[s][j] = 0.f;
    float prior_cost = 0.f;
    if (inimg) {
        if (REF) {      // the inverse pairs see this depth through their bilinear samples of it: the adjoint sums (fixed point), d L / d pd = b_dc S_dc - a_i S_ph
            longlong2 *ep = reinterpret_cast<longlong2 *>(J.ext2 + ((size_t)b * hw + gyo * W + gxo) * 2);
            const longlong2 e2 = *ep;
            *ep = make_longlong2(0, 0);                                  // consumed: the next linearisation's scatter starts from zero (no memset launch)
            const float Edc = (float)((double)e2.x * (1.0 / DREF_FIX)), Eph = (float)((double)e2.y * (1.0 / DREF_FIX));
            g_rho -= o_depth * o_depth * (r_dc * Edc - r_eph * Eph);     // d depth / d rho = -depth^2; in units of a_f
            prior_cost = extra_cost;
        }
        float D = Dsum;
        if (J.w_prior > 0.f) {
            float rho = frcp(o_depth), rho0 = frcp(J.depth0[(size_t)b * hw + gyo * W + gxo]);
            float ir2 = frcp(rho0 * rho0), dr = rho - rho0;
            g_rho += mcnt * 2.f * J.w_prior * dr * ir2;
            D += mcnt * 2.f * J.w_prior * ir2;
            prior_cost += mcnt * J.w_prior * dr * dr * ir2;
        }
        const float Dd = (1.f + J.lambda_depth) * D;
        const bool elim = Dd > 1e-30f && !o_pad;         // (dense_kernel.h: pixels sampled across the zero padding keep their depth)
        iD = (elim && !J.qres) ? frcp(Dd) : 0.f;         // (quarter resolution: eliminated per CELL by k_qres_schur, nothing here)
        float *jr = jrp();
        jr[0] = g_rho; jr[1] = elim ? Dd : 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 6; j++) Bv[s][j] = jr[2 + 6 * s + j];          // (this thread's own stores: L1 / L2 hits)
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {       // diagonal blocks and right-hand sides
        float v[28];
        int h = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            v[21 + j] = -Bv[s][j] * g_rho * iD;
#pragma unroll
            for (int i = 0; i <= j; i++) { v[h] = -Bv[s][j] * Bv[s][i] * iD; h++; }
        }
        v[27] = (s == 0) ? prior_cost : 0.f;
        joint_reduce_add<28, NT>(v, red, acc, [&](int k) {
            if (k == 27) return JL::OFF_S;                       // the prior's cost rides in the sum(M W diff) slot
            if (k >= 21) return JL::OFF_G + 6 * s + (k - 21);
            int j = 0;
            while ((j + 1) * (j + 2) / 2 <= k) j++;
            return JL::tri(6 * s + j, 6 * s + (k - j * (j + 1) / 2));
        }, tid);
    }
    if (!J.argmin || (REF && r_dc > 0.f)) {   // off-diagonal blocks: the shared depth couples the poses (18 values per reduction); under the
                                              // reference's loss the depth-consistency terms of ALL sources see the depth of every pixel
#pragma unroll
        for (int s = 1; s < NS; s++)
#pragma unroll
            for (int t = 0; t < s; t++)
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    float v[18];
#pragma unroll
                    for (int k = 0; k < 18; k++) v[k] = -Bv[s][half * 3 + k / 6] * Bv[t][k % 6] * iD;
                    joint_reduce_add<18, NT>(v, red, acc, [&](int k) { return JL::tri(6 * s + half * 3 + k / 6, 6 * t + k % 6); }, tid);
                }
    }
    // share of source 0 carries the whole photometric numerator's slot layout of k_solve: OFF_S = sum M W diff (+ prior), OFF_S+1 = K
    if (tid == 0) {
        float num = 0.f, kk = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++) { num += acc[JL::OFF_SHARE + s]; kk += acc[JL::OFF_NM + s]; }
        acc[JL::OFF_S] += num; acc[JL::OFF_S + 1] = kk;
    }
    __syncthreads();
    float *myrec = J.jblockrec + ((size_t)b * (J.rec_stride > 0 ? J.rec_stride : nblk) + bid) * JL::NACC;
    for (int i = tid; i < JL::NACC; i += NT) myrec[i] = acc[i];
    TC_JSTAMP(7)
#undef TC_JSTAMP
    stamp_end(P.stamp, tid);
}

template <int NS, int TW, int TH, int NT, bool TRACE = false, bool REF = false, bool SM = REF>
__global__ __launch_bounds__(NT, (JointShape<NS, NT, REF, SM>::OCC)) void k_dense_joint(LinParams P, JointParams J) {
    __shared__ float4 smem[JointSmem<NS, TW, TH, NT, REF>::N4];
    dense_joint_body<NS, TW, TH, NT, TRACE, REF, SM>(P, J, (int)blockIdx.y, smem);
}
// Free source maps under the reference's loss: the forward groups (S sources per target; rows [0, Ja.B)) and the inverse pairs as groups of one
// source (rows behind them) in ONE launch -- independent work on disjoint records, LDS overlaid by role (third session of round 5; was two launches)
template <int NS, int TW, int TH, int NT, bool TRACE = false>
__global__ __launch_bounds__(NT, (JointShape<NS, NT, true, false>::OCC)) void k_dense_joint2(LinParams Pa, JointParams Ja, LinParams Pb, JointParams Jb) {
    constexpr int NA = JointSmem<NS, TW, TH, NT, true>::N4, NB = JointSmem<1, TW, TH, NT, true>::N4;
    __shared__ float4 smem[NA > NB ? NA : NB];
    if ((int)blockIdx.y < Ja.B) dense_joint_body<NS, TW, TH, NT, TRACE, true, false>(Pa, Ja, (int)blockIdx.y, smem);
    else dense_joint_body<1, TW, TH, NT, TRACE, true, false>(Pb, Jb, (int)blockIdx.y - Ja.B, smem);
}

// ---------------------------------------------------------------------------------------------------------------
struct JointState {                 // per target (fp64): LM bookkeeping + the accepted reduced system [S | -gS] (NP x (NP+1))
    double lambda, cost_cur;
    int have_cur, pad;
    double M[6 * JMAXS * (6 * JMAXS + 1)];
};

struct JointSolveParams {
    const float *jblockrec;   // [B][nblk][NACC]
    JointState *js;           // [B]
    PairState *st;            // [N] (forward pairs n = s B + b hold their transforms)
    PairConst *pc;
    float *stats;             // [Ntot][n_iters+1][TCSFM_NSTAT] or null (rows of the forward pairs)
    int nblk, B, it, n_iters, solver, mode;       // mode 0: step, 1: final LM check
    double lambda_up, lambda_down, lambda_min, lambda0;
    float *pose_out;          // [Ntot][6] or null: written by the last launch
    double *delta_out;        // [B][6 JMAXS] pose step of this iteration (back-substitution)
    int *accept_out;          // [B] LM decision (or null)
    int *trace_decide;        // [N] slot of every forward pair of the target, or null
    const int *norms;         // REF (or null): [groups][2] batch-summed mask counts; the records are in units of c_f / K_f of the target's group
    int norm_B;               //   targets per group (coalesced calls; 0: one group)
    double c_f;
    double *export_out;       // linearisation export (tcsfm_linearize_dense_window): [B][2 + 6 JMAXS] = cost, factor a_f, g (per source 6), or null
    // coalesced calls (CoalTab): the refined pose of forward pair n = s B + b goes to ITS call's output, at the pair's index in that call
    int c_ncall, c_B, c_S, c_pad;
    float *c_pose_out[TC_MAX_COAL];
    // Split record sum (round 5): nsplit > 1 workgroups per target each sum a contiguous share of the target's workgroup records (the fetch
    // of freshly written records by ONE workgroup is rate-bound: 186 KB for a KITTI target on 32 x 8 tiles took ~7 us), leave their
    // partial sums in jpart [B][nsplit][NACC] (fp64, written through) and take a ticket; the LAST arriver adds the partials in index
    // order (deterministic) and solves.  No workgroup waits for another.  jtick [B]: zero between launches (the last arriver resets it).
    int nsplit;
    double *jpart;
    int *jtick;
    // l_pose_consist (optimizer.py:95-96; kernels templated PC): c = weight / (6 S B); pose_lin [2][pc_np][12] holds every pair's transform
    // at linearisation `it` in buffer it & 1 (SolveParams, kernels.h); forward pair n = s B + b sits at pc_self0 + n, its partner at pc_part0 + n
    double w_pc, pc_eps;
    double *pose_lin;
    int pc_np, pc_self0, pc_part0;
    long long *dbg;           // diagnostic runs only (TCSFM_DEBUG_STAMPS=2): wall-clock stamps of the phases of target 0's solve; null in production
};

constexpr int JSOLVE_NT = 1024;
// PC: the pose-consistency term is added to the diagonal 6 x 6 blocks of the reduced system (it does not depend on the depth map); as in
// k_solve every pair sees its partner at THIS linearisation and the block-Jacobi majoriser 2 c / max(|r|, eps) (oracle dref_pose_consist)
template <int NS, bool PC = false>
__device__ __forceinline__ void solve_joint_body(const JointSolveParams &P, const int b, const int tid, const int split_k = 0) {
    using JL = JointLayout<NS>;
    constexpr int NP = JL::NP, NC = NP + 1;
    // thread = (accumulator, record subset).  S = 1 (32 accumulators): 32 subsets, so a target's 300-480 tile records are ONE batch of at most 16 loads per thread
    // (third session of round 5: with 128 accumulator slots only a quarter of the threads loaded, in two dependent batches: 5.5 of the launch's 9.3 us)
    constexpr int APAD = JL::NACC <= 32 ? 32 : (JL::NACC <= 128 ? 128 : 256), PARTS = JSOLVE_NT / APAD;
    constexpr int NBATCH = PARTS >= 32 ? 16 : 32;        // loads in flight per thread and batch
    static_assert(JL::NACC <= 256 && NP * NC <= JSOLVE_NT, "k_solve_joint: one thread per accumulator column and per matrix entry");
    __shared__ double tot[JL::NACC];
    __shared__ double part[JSOLVE_NT];
    __shared__ double M[NP * NC];
    __shared__ double dl[NP];
    __shared__ double Ts[NS][40];
    __shared__ int s_flag[2];
    const int NTS = JSOLVE_NT;
    // the optimiser state is fetched NOW, beside the records: the serial phases below never wait on a global load (as in k_solve)
    __shared__ double Ks[NS][12];
#define TC_SSTAMP(i) if (P.dbg && tid == 0 && b == 0 && split_k == 0) P.dbg[i] = wall_clock64();
    TC_SSTAMP(0)
    JointState &S = P.js[b];
    const double S_lambda = S.lambda, S_cost_cur = S.cost_cur;
    const int S_have_cur = S.have_cur;
    double pre_try = 0.0, pre_cur = 0.0;
    if (tid < 64 && (tid >> 4) < NS) {
        const PairState &ps0 = P.st[(tid >> 4) * P.B + b];
        const int sub = tid & 15;
        if (sub < 12) { pre_try = ps0.Ttry[sub]; pre_cur = ps0.Tcur[sub]; }
        if (sub < 9) Ks[tid >> 4][sub] = ps0.K[sub];
    }
    {
        // deterministic fp64 sum of the target's workgroup records, as in k_solve: thread = (accumulator, record subset), every
        // subset's loads issued in batches of 8 so that they are all in flight, subsets combined in fixed order
        const int c = tid & (APAD - 1), q = tid / APAD;
        const int G = P.nsplit > 1 ? P.nsplit : 1;
        const int chunk = (P.nblk + G - 1) / G, r_lo = split_k * chunk, r_hi = min(P.nblk, r_lo + chunk);      // this workgroup's share of the records
        double s = 0.0;
        if (c < JL::NACC) {
            const float *p = P.jblockrec + (size_t)b * P.nblk * JL::NACC + c;
            for (int r0 = r_lo + q; r0 < r_hi; r0 += NBATCH * PARTS) {      // NBATCH loads per thread in flight
                float w[NBATCH];
#pragma unroll
                for (int k = 0; k < NBATCH; k++) { const int r = r0 + k * PARTS; w[k] = p[(size_t)(r < r_hi ? r : r_lo) * JL::NACC]; }
#pragma unroll
                for (int k = 0; k < NBATCH; k++) s += (r0 + k * PARTS < r_hi) ? (double)w[k] : 0.0;
            }
        }
        part[tid] = s;
        __syncthreads();
        if (q == 0 && c < JL::NACC) {
            double t = 0.0;
            for (int k = 0; k < PARTS; k++) t += part[k * APAD + c];
            tot[c] = t;
            if (G > 1) __hip_atomic_store(&P.jpart[((size_t)b * G + split_k) * JL::NACC + c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // written through
        }
        if (G > 1) {
            // protocol of block_reduce_publish (kernels.h): write-through stores -> every storing wave drains vmcnt -> workgroup barrier -> one
            // relaxed agent-scope ticket; the last arriver: agent-scope acquire -> barrier -> plain loads, fixed order
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int t = __hip_atomic_fetch_add(&P.jtick[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_flag[1] = (t == G - 1);
            }
            __syncthreads();
            if (!s_flag[1]) return;               // (workgroup-uniform: the other shares' workgroups are done)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                P.jtick[b] = 0;
            }
            __syncthreads();
            if (tid < JL::NACC) {
                double t = 0.0;
                for (int k = 0; k < G; k++) t += P.jpart[((size_t)b * G + k) * JL::NACC + tid];
                tot[tid] = t;
            }
        }
    }
    __syncthreads();
    TC_SSTAMP(1)
    __shared__ double pcA[PC ? NS : 1][36], pcG[PC ? NS : 1][6], pcD[PC ? NS : 1][6], pcH[PC ? NS : 1][36], pcg[PC ? NS : 1][6], pcC[PC ? NS : 1];
    double pc_cost = 0.0;
    if constexpr (PC) {
        const int g = tid >> 6, l = tid & 63;
        if (g < NS && l == 0) {       // pose coordinates, A^-1 = [[-I, Tx], [0, -Je^-1]], IRLS gradient and majoriser (serial: one lane per source)
            const int n = g * P.B + b;
            const double *Tm = P.st[n].Ttry, *Tp = P.pose_lin + ((size_t)(P.it & 1) * P.pc_np + P.pc_part0 + n) * 12;
            double tm[12], tp_[12], pm[6], pp[6];
            for (int i = 0; i < 12; i++) { tm[i] = Tm[i]; tp_[i] = Tp[i]; }
            T_to_pose(tm, pm); T_to_pose(tp_, pp);
            const double cx = cos(-pm[3]), sx = sin(-pm[3]), cy = cos(-pm[4]), sy = sin(-pm[4]);
            const double Je[9] = {1, 0, sy, 0, cx, -sx * cy, 0, sx, cx * cy};
            const double id = 1.0 / cy;
            const double Ji[9] = {(Je[4] * Je[8] - Je[5] * Je[7]) * id, -(Je[1] * Je[8] - Je[2] * Je[7]) * id, (Je[1] * Je[5] - Je[2] * Je[4]) * id,
                                  -(Je[3] * Je[8] - Je[5] * Je[6]) * id, (Je[0] * Je[8] - Je[2] * Je[6]) * id, -(Je[0] * Je[5] - Je[2] * Je[3]) * id,
                                  (Je[3] * Je[7] - Je[4] * Je[6]) * id, -(Je[0] * Je[7] - Je[1] * Je[6]) * id, (Je[0] * Je[4] - Je[1] * Je[3]) * id};
            const double tq[3] = {-pm[0], -pm[1], -pm[2]};
            const double Tx[9] = {0, -tq[2], tq[1], tq[2], 0, -tq[0], -tq[1], tq[0], 0};
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    double v = 0.0;
                    if (i < 3 && j < 3) v = (i == j) ? -1.0 : 0.0;
                    else if (i < 3) v = Tx[3 * i + (j - 3)];
                    else if (j >= 3) v = -Ji[3 * (i - 3) + (j - 3)];
                    pcA[g][6 * i + j] = v;
                }
            double cc = 0.0;
            for (int j = 0; j < 6; j++) {
                const double rj = pm[j] + pp[j], a = fabs(rj), den = a > P.pc_eps ? a : P.pc_eps;
                cc += 0.5 * P.w_pc * a;
                pcG[g][j] = P.w_pc * rj / den; pcD[g][j] = 2.0 * P.w_pc / den;
            }
            pcC[g] = cc;
        }
        __syncthreads();
        if (g < NS && l < 36) {
            const int i = l / 6, j = l - 6 * i;
            double v = 0.0;
            for (int k = 0; k < 6; k++) v += pcA[g][6 * k + i] * pcD[g][k] * pcA[g][6 * k + j];
            pcH[g][l] = v;
        } else if (g < NS && l < 42) {
            const int i = l - 36;
            double v = 0.0;
            for (int k = 0; k < 6; k++) v += pcA[g][6 * k + i] * pcG[g][k];
            pcg[g][i] = v;
        }
        __syncthreads();
        for (int g2 = 0; g2 < NS; g2++) pc_cost += pcC[g2];
    }
    double Kn = tot[JL::OFF_S + 1], an = Kn > 0 ? 1.0 / Kn : 0.0;
    if (P.norms) { Kn = (double)P.norms[2 * (P.norm_B > 0 ? b / P.norm_B : 0)]; an = Kn > 0 ? P.c_f / Kn : 0.0; }      // the reference's batch normaliser (optimizer.py:69)
    const double cost = an * tot[JL::OFF_S] + pc_cost;
    if (P.export_out) {       // one linearisation, exported (no step): cost, a_f and the pose gradients of the group's records
        if (tid == 0) { P.export_out[(size_t)b * (2 + 6 * JMAXS)] = cost; P.export_out[(size_t)b * (2 + 6 * JMAXS) + 1] = an; }
        if (tid < NP) P.export_out[(size_t)b * (2 + 6 * JMAXS) + 2 + tid] = an * tot[JL::OFF_G + tid];
        return;
    }
    double lambda = P.it == 0 && P.mode == 0 ? P.lambda0 : S_lambda;
    const bool have_cur = !(P.it == 0 && P.mode == 0) && S_have_cur;
    const double cost_cur = S_cost_cur;
    if (P.stats && tid < NS) {      // row of forward pair (tid, b): joint cost, own share, own mask count, lambda, the iterate
        const int n = tid * P.B + b;
        float *st = P.stats + ((size_t)n * (P.n_iters + 1) + P.it) * TCSFM_NSTAT;
        st[0] = (float)cost; st[1] = (float)(an * tot[JL::OFF_SHARE + tid]); st[2] = (float)tot[JL::OFF_NM + tid]; st[3] = (float)lambda;
        T_to_pose_f32(P.st[n].Ttry, st + TCSFM_STAT_POSE);
    }
    if (P.mode == 1) {              // LM: keep the last trial (all S poses; the depth map follows accept_out) only if it lowered the cost
        const bool keep = cost < cost_cur;
        if (tid == 0 && P.accept_out) P.accept_out[b] = keep ? 1 : 0;
        if (tid < NS) {
            const int n = tid * P.B + b;
            if (P.trace_decide) P.trace_decide[n] = keep ? 1 : 0;
            PairState &ps = P.st[n];
            if (keep) for (int i = 0; i < 12; i++) ps.Tcur[i] = ps.Ttry[i];
            if (P.pose_out) {
                float pose[6];
                T_to_pose_f32(keep ? ps.Ttry : ps.Tcur, pose);
                for (int i = 0; i < 6; i++) P.pose_out[n * 6 + i] = pose[i];
            }
        }
        return;
    }
    const bool accept = (P.solver == 0) || !have_cur || (cost < cost_cur);
    if (accept && P.solver == 1 && have_cur) lambda = fmax(lambda * P.lambda_down, P.lambda_min);
    if (!accept) lambda *= P.lambda_up;
    // [S | -gS] of the accepted linearisation
    for (int i = tid; i < NP * NC; i += NTS) {
        const int r = i / NC, c = i - r * NC;
        double v;
        if (accept) {
            v = (c < NP) ? an * tot[JL::tri(r, c)] : -an * tot[JL::OFF_G + r];
            if constexpr (PC) {
                if (c == NP) v -= pcg[r / 6][r % 6];
                else if (r / 6 == c / 6) v += pcH[r / 6][(r % 6) * 6 + c % 6];
            }
            S.M[i] = v;
        }
        else v = S.M[i];
        M[i] = v;
    }
    __syncthreads();
    TC_SSTAMP(2)
    if (tid < NP) M[tid * NC + tid] += lambda * M[tid * NC + tid] + 1e-12;     // Marquardt damping
    if (tid == 0) { s_flag[0] = 1; }
    __syncthreads();
    // Gauss-Jordan (unpivoted: the system is SPD) by wave 0 alone: a wave's LDS accesses execute in program order, so the pivots
    // need no workgroup barrier (12-18 pivots x 2 barriers of 16 waves cost more than the arithmetic); lane = entries lane, lane + 64, ..
    if (tid < 64) {
        constexpr int EPL = (NP * NC + 63) / 64;
        double *Mv = M;      // (not volatile: the memory legaliser waits after every volatile access; the fences below order the wave's LDS traffic)
        for (int k = 0; k < NP; k++) {
            const double piv = Mv[k * NC + k];
            if (tid == 0 && !(piv > 0.0)) s_flag[0] = 0;
            const double ipiv = rcp64(piv);
            double upd[EPL];
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = tid + 64 * e, r = i / NC, c = i - r * NC;
                upd[e] = (i < NP * NC && r != k) ? Mv[i] - Mv[r * NC + k] * ipiv * Mv[k * NC + c] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = tid + 64 * e, r = i / NC;
                if (i < NP * NC && r != k) Mv[i] = upd[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    TC_SSTAMP(3)
    if (tid < NP) dl[tid] = s_flag[0] ? M[tid * NC + NP] / M[tid * NC + tid] : 0.0;
    __syncthreads();
    TC_SSTAMP(4)
    if (tid == 0) {
        S.lambda = lambda;
        if (accept) { S.cost_cur = cost; S.have_cur = 1; }
        if (P.accept_out) P.accept_out[b] = accept ? 1 : 0;
        if (P.delta_out) for (int i = 0; i < NP; i++) P.delta_out[b * 6 * JMAXS + i] = dl[i];
    }
    // per source: T_try = exp(d_s) T_accepted and the next constants, on 16 lanes of wave 0 each (as k_solve: the exponential is
    // serial on one lane, the product and the constants one entry per lane)
    if (tid < 64) {
        const int g = tid >> 4, sub = tid & 15;
        const bool on = g < NS;
        const int n = (on ? g : 0) * P.B + b;
        PairState &ps = P.st[n];
        double *T = Ts[on ? g : 0];
        if (on && sub < 12) {
            const double v = accept ? pre_try : pre_cur;
            T[12 + sub] = v;
            if (accept) ps.Tcur[sub] = v;
        }
        if (on && sub == 0) {
            if (P.trace_decide) P.trace_decide[n] = accept ? 1 : 0;
            double d6[6], E[12];
            for (int i = 0; i < 6; i++) d6[i] = dl[6 * g + i];
            se3_exp(d6, E);
            for (int i = 0; i < 12; i++) T[i] = E[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool last_gn = (P.solver == 0 && P.it == P.n_iters - 1);
        if (on && sub < 12) {           // se3_mul's operation order
            const int i = sub >> 2, j = sub & 3;
            double v = T[4 * i] * T[12 + j] + T[4 * i + 1] * T[16 + j] + T[4 * i + 2] * T[20 + j];
            if (j == 3) v += T[4 * i + 3];
            T[24 + sub] = v;
            ps.Ttry[sub] = v;
            if (last_gn) ps.Tcur[sub] = v;
            if constexpr (PC) P.pose_lin[((size_t)((P.it + 1) & 1) * P.pc_np + P.pc_self0 + n) * 12 + sub] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (on) write_const_lanes<6>(sub, Ks[g], T + 24, 0.0, P.pc[n]);
        if (on && sub == 0 && last_gn && P.pose_out) {
            float pose[6];
            T_to_pose_f32(T + 24, pose);
            float *po = P.pose_out + n * 6;
            if (P.c_ncall > 0) { const CoalIdx ci = coal_index(P.c_ncall, P.c_B, P.c_S, n); po = P.c_pose_out[ci.call] + ci.li * 6; }
            for (int i = 0; i < 6; i++) po[i] = pose[i];
            if (P.stats) {
                float *st = P.stats + ((size_t)n * (P.n_iters + 1) + P.n_iters) * TCSFM_NSTAT + TCSFM_STAT_POSE;
                for (int i = 0; i < 6; i++) st[i] = pose[i];
            }
        }
    }
    TC_SSTAMP(5)
#undef TC_SSTAMP
}

template <int NS>
__global__ __launch_bounds__(JSOLVE_NT) void k_solve_joint(JointSolveParams P) {
    const int G = P.nsplit > 1 ? P.nsplit : 1;
    solve_joint_body<NS>(P, (int)blockIdx.x / G, threadIdx.x, (int)blockIdx.x % G);
}

// Dense mode on the reference's loss (round 5): the target groups' reduced systems AND the inverse pairs' 6 x 6 systems of one iteration
// in ONE launch -- they are independent (different workgroups, different state), so a second dependent launch bought nothing but its
// ~6 us.  Workgroups [0, B): solve_joint_body; [B, B + n_inv): solve_body of inverse pair blockIdx.x - B (its record sums spread over
// the 1024 threads).
// PC: with the pose-consistency term (both roles).
template <int NS, bool PC = false>
__global__ __launch_bounds__(JSOLVE_NT) void k_solve_front(JointSolveParams Pj, SolveParams Pi) {
    const int G = Pj.nsplit > 1 ? Pj.nsplit : 1;      // (workgroups [0, B G): the targets' systems, G record shares each)
    if ((int)blockIdx.x < Pj.B * G) solve_joint_body<NS, PC>(Pj, (int)blockIdx.x / G, threadIdx.x, (int)blockIdx.x % G);
    else solve_body<6, JSOLVE_NT, true, PC>(Pi, (int)blockIdx.x - Pj.B * G, threadIdx.x);
}

// Free source maps (round 5): the forward groups (NS sources each) and the inverse groups (one source each: the inverse pairs with their
// source map) of one iteration are independent systems -- one launch solves both.  Workgroups [0, Pa.B): Pa; [Pa.B, Pa.B + Pb.B): Pb.
template <int NS, bool PC = false>
__global__ __launch_bounds__(JSOLVE_NT) void k_solve_joint2(JointSolveParams Pa, JointSolveParams Pb) {
    const int Ga = Pa.nsplit > 1 ? Pa.nsplit : 1, Gb = Pb.nsplit > 1 ? Pb.nsplit : 1;
    if ((int)blockIdx.x < Pa.B * Ga) solve_joint_body<NS, PC>(Pa, (int)blockIdx.x / Ga, threadIdx.x, (int)blockIdx.x % Ga);
    else { const int x = (int)blockIdx.x - Pa.B * Ga; solve_joint_body<1, PC>(Pb, x / Gb, threadIdx.x, x % Gb); }
}

// back-substitution of the shared map; LM: promote / roll back first (as k_dense_update_lm)
struct JointUpdateParams {
    const float *jrec;        // [B][H*W][JREC] records of the linearisation just evaluated
    float *jrec_acc;          // LM: accepted records (or null)
    float *depth_acc;         // LM: accepted depth [B][H*W] (or null)
    const double *delta;      // [B][6 JMAXS]
    const int *accept;        // LM: [B] (or null)
    float *depth;             // [.][H*W]: slots of the forward pairs n = s B + b, all S written
    float *depth_out;         // optional: [.][H*W] same layout (the caller's buffer), or null
    int hw, B, S, mode;       // mode 0: step; 1: final LM decision (keep the trial or fall back to the accepted map), no step
    float rho_lo, rho_hi;
    int *norms_zero;          // REF (or null): the batch counters, consumed by the solve kernels before this launch: zeroed for the next linearisation
    int norms_n;              //   how many of them (2 per normaliser group)
    float4 *srcpack_inv;      // REF (or null): packs of the inverse pairs (pair S B + s B + b samples target b's depth: channel w), refreshed
    int W, H;                 //   with the new map so that the inverse pairs of the next linearisation see the depth the forward pairs see
    // coalesced calls: depth_out per call (forward slot (s, b) of the batch = slot s c_B + b % c_B of call b / c_B); used when c_ncall > 0
    int c_ncall, c_B;
    float *c_depth_out[TC_MAX_COAL];
};

template <int NS>
__device__ __forceinline__ void joint_update_body(const JointUpdateParams &P, const int idx, const int b) {
    using JL = JointLayout<NS>;
    if (P.norms_zero && idx < P.norms_n && b == 0) P.norms_zero[idx] = 0;
    if (idx >= P.hw) return;
    const size_t o = (size_t)b * P.hw + idx;
    float dep;
    if (P.mode == 1) {
        dep = P.accept[b] ? P.depth[o] : P.depth_acc[o];
    } else {
        float r[JL::JREC];
        float base;
        const bool lm = P.jrec_acc != nullptr;
        if (!lm || P.accept[b]) {
            const float4 *rt = reinterpret_cast<const float4 *>(P.jrec + o * JL::JREC);
#pragma unroll
            for (int k = 0; k < JL::JREC / 4; k++) { const float4 q = rt[k]; r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w; }
            base = P.depth[o];
            if (lm) {
                float4 *ra = reinterpret_cast<float4 *>(P.jrec_acc + o * JL::JREC);
#pragma unroll
                for (int k = 0; k < JL::JREC / 4; k++) ra[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
                P.depth_acc[o] = base;
            }
        } else {
            const float4 *ra = reinterpret_cast<const float4 *>(P.jrec_acc + o * JL::JREC);
#pragma unroll
            for (int k = 0; k < JL::JREC / 4; k++) { const float4 q = ra[k]; r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w; }
            base = P.depth_acc[o];
        }
        dep = base;
        if (r[1] > 0.f) {
            float bd = 0.f;
#pragma unroll
            for (int j = 0; j < 6 * NS; j++) bd += r[2 + j] * (float)P.delta[b * 6 * JMAXS + j];
            dep = 1.f / depth_step(1.f / base, -(r[0] + bd) / r[1], P.rho_lo, P.rho_hi);
        }
    }
    for (int s = 0; s < P.S; s++) {
        P.depth[(size_t)(s * P.B + b) * P.hw + idx] = dep;
        if (P.c_ncall > 0) P.c_depth_out[b / P.c_B][(size_t)(s * P.c_B + b % P.c_B) * P.hw + idx] = dep;
        else if (P.depth_out) P.depth_out[(size_t)(s * P.B + b) * P.hw + idx] = dep;
    }
    if (P.srcpack_inv) {
        const int v = idx / P.W, u = idx - v * P.W;
        for (int s = 0; s < P.S; s++)
            P.srcpack_inv[((size_t)(s * P.B + b) * (P.H + 2) + v + 1) * (P.W + 2) + u + 1].w = dep;
    }
}
template <int NS>
__global__ __launch_bounds__(256) void k_dense_joint_update(JointUpdateParams P) {
    joint_update_body<NS>(P, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
// ... of the forward groups' maps (rows [0, Pa.B)) and of the inverse groups' source maps (rows behind them) in one launch
template <int NS>
__global__ __launch_bounds__(256) void k_dense_joint_update2(JointUpdateParams Pa, JointUpdateParams Pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.y < Pa.B) joint_update_body<NS>(Pa, idx, blockIdx.y);
    else joint_update_body<1>(Pb, idx, (int)blockIdx.y - Pa.B);
}

}  // namespace tc
