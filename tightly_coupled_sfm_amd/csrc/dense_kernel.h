// dense_kernel.h -- dense mode: pose (6) + per-pixel inverse depth of the target, per-pixel Schur complement
// (BASELINE.json config 5; north star "block-sparse normal equations").  Mirrors orc_linearize_dense / orc_refine_dense.
//
// k_dense_linearize uses the ADJOINT form of the 3x3-coupled SSIM gradient: instead of every residual pixel p visiting
// the geometry of its 9 window pixels (k_linearize pass B), every pixel q gathers d C / d rec_q from the 9 residuals that
// see it and applies its OWN geometric Jacobian once -- which is also exactly what the per-pixel depth gradient needs.
//   phase 1   tile + 2-pixel halo : warp, bilinear tap, depth-consistency weight              -> LDS rec1 / aux
//   phase 2a  tile + 1-pixel halo : SSIM statistics, L1, mask; adjoint coefficient records      -> LDS coef
//   phase 2b  tile                : gather 9 coefficient records (reflect-pad multiplicities), own L1 / dW terms,
//                                   pose gradient, depth gradient, curvature blocks H_xx, B_q, D_q, per-pixel Schur
//                                   elimination; per-pixel (g_rho, Dd, B[6]) record to HBM for the back-substitution
// The reduced pose system S, g_S leaves the kernel through the same deterministic reduction as k_linearize and is
// solved by k_solve; k_dense_update then back-substitutes the depth increments.
#pragma once
#include "kernels.h"

namespace tc {

struct DenseParams {
    float *dense_rec;       // [N][H*W][8]  g_rho, Dd, B[6]   (unnormalised: the common 1/sum(M) cancels per pixel)
    const float *depth0;    // [N][H*W]     initial depth (prior centre)
    float lambda_depth;     // Marquardt damping of the depth block
    float w_prior;          // weight of the masked prior  w sum M ((rho - rho0)/rho0)^2 / sum M
    // Fused back-substitution (Gauss-Newton, pair form): the depth map this linearisation works on is the PREVIOUS iteration's map
    // (LinParams::depth_t) advanced on the fly by the previous iteration's per-pixel records and pose step -- the arithmetic of
    // k_dense_update, evaluated for the tile and its halo -- and written once, for the tile's own pixels, to depth_next.  Saves
    // one launch per iteration (5 us + a kernel boundary of a ~25 us iteration at B=1).  prev_rec == nullptr: first iteration.
    const float *prev_rec;      // [N][H*W][8] records of the previous linearisation (a different buffer than dense_rec)
    const double *prev_delta;   // [N][8] pose step of the previous iteration (written by k_solve)
    float *depth_next;          // [N][H*W] the advanced depth map (a different buffer than LinParams::depth_t), or nullptr
    float rho_lo, rho_hi;       // clamp of the inverse depth: 1/max_depth, 1/min_depth
};

// back-substitution of one pixel: drho = -(g_rho + B' dxi) / Dd ; rho clamped to [rho_lo, rho_hi]  (shared by the fused form in
// k_dense_linearize and by k_dense_update, so that both produce the same bits)
// Per-pixel trust region of the depth step: |drho| <= DEPTH_STEP_MAX * rho.  A pixel whose sample -- or whose neighbours' samples,
// through the 3x3 SSIM window -- touch grid_sample's zero padding gets a photometric gradient of (colour / 1 px) against the
// curvature of an ordinary pixel: raw Gauss-Newton steps of 30-60 % of the inverse depth were measured at such pixels (a few dozen
// per 640x192 frame), i.e. outside the range in which the linearisation means anything, and 1e-4 of such a step is 1e-4 of the depth.
constexpr float DEPTH_STEP_MAX = 0.25f;
__device__ __forceinline__ float depth_step(float rho0, float drho, float rho_lo, float rho_hi) {
    const float lim = DEPTH_STEP_MAX * rho0;
    float rho = rho0 + fminf(fmaxf(drho, -lim), lim);
    return fminf(fmaxf(rho, rho_lo), rho_hi);
}
__device__ __forceinline__ float dense_advance(float dep, const float4 &r0, const float4 &r1, const float *d, float rho_lo, float rho_hi) {
    if (!(r0.y > 0.f)) return dep;
    float bd = r0.z * d[0] + r0.w * d[1] + r1.x * d[2] + r1.y * d[3] + r1.z * d[4] + r1.w * d[5];
    return 1.f / depth_step(1.f / dep, -(r0.x + bd) / r0.y, rho_lo, rho_hi);
}

// Software-pipelined window reads of phases 2a / 2b (round 5, third session; bit-identical to the rolled loops, kept for A/B: -DTC_DENSE_PIPELINED=0)
#ifndef TC_DENSE_PIPELINED
#define TC_DENSE_PIPELINED 1
#endif
#ifndef TC_DENSE_PIPE_A
#define TC_DENSE_PIPE_A TC_DENSE_PIPELINED
#endif
#ifndef TC_DENSE_PIPE_B
#define TC_DENSE_PIPE_B TC_DENSE_PIPELINED
#endif
// waves per SIMD the register allocation must leave room for: 4 = two 512-thread workgroups per CU (the LDS allows exactly two)
#ifndef TC_DENSE_OCC
#define TC_DENSE_OCC 4
#endif
template <int TW, int TH, int NT, bool TRACE = false>
__global__ __launch_bounds__(NT, TC_DENSE_OCC) void k_dense_linearize(LinParams P, DenseParams Dn) {
    constexpr int NP = 6;
    using L = AccLayout<NP>;
    constexpr int W2 = TW + 4, H2 = TH + 4, N2 = W2 * H2;   // phase-1 region (2-pixel halo)
    constexpr int W1 = TW + 2, H1 = TH + 2, N1 = W1 * H1;   // phase-2a region (1-pixel halo)
    constexpr int NCEN = TW * TH;
    static_assert(NCEN == NT, "one tile pixel per thread");
    __shared__ float4 rec1[N2 * 3];   // [y0 y1 x0 x1][gx0 gx1 gy0 gy1][y2 x2 gx2 gy2]: channel pairs on aligned register pairs
    __shared__ float4 aux[N2];        // W, valid, auto_err, -
    __shared__ float4 coef[N1 * 3];   // w (cA'', cB, cC) per channel: [A0 A1 A2 B0 | B1 B2 C0 C1 | C2 - - -]
    __shared__ float red[(NT / 64) * L::NACC];

    const int nblk = P.tiles_x * P.tiles_y;
    int bid = blockIdx.x;
    {
        int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n = blockIdx.y;
    const PairConst &c = P.pc[n];
    const int H = P.H, W = P.W, hw = H * W;
    const int tyi = bid / P.tiles_x, txi = bid - tyi * P.tiles_x;
    const int x00 = txi * TW, y00 = tyi * TH;
    const float4 *tgtpack = P.tgtpack + (size_t)n * hw;
    const float4 *srcpack = P.srcpack + (size_t)n * (H + 2) * (W + 2);   // zero-bordered (tap4)
    const float *depth_t = P.depth_t + (size_t)n * hw;
    const int tid = threadIdx.x;
    stamp_begin(P.stamp, tid);

    // own pixel (tile coordinates) and values carried from phase 1 to phase 2b
    const int oy = tid / TW, ox = tid - oy * TW;
    const int gxo = x00 + ox, gyo = y00 + oy;
    const bool inimg = gxo < W && gyo < H;
    float a[7], b[7], zc[7];
    Geo o_g;                 // own pixel's warp geometry (phase 1 -> phase 2b)
    float o_pd = 0.f, o_cd = 1.f, o_dgx = 0.f, o_dgy = 0.f, o_depth = 1.f;
    bool o_pad = false;      // the pixel's own sample is valid but its bilinear footprint touches the zero padding (see `elim` below)

    // ---------------- phase 1 ----------------
    // Own pixel for every thread; the N2 - NCEN pixels of the 2-pixel ring are a second pixel for the first threads.  As in
    // k_linearize those waves run both pixels as one software-pipelined sequence (loads, warps + gather issues,
    // interpolations) so that the ring pixel's dependent load chain overlaps the own pixel's.
    constexpr int NRING = N2 - NCEN;
    static_assert(NRING <= NT, "one ring round");
    constexpr int RING_THREADS = (NRING + 63) / 64 * 64;
    struct Stage { int lx, ly, px, py; float4 tp; float dep; float4 r0, r1; Geo g; Tap t; };
    const bool fused = Dn.prev_rec != nullptr;      // wave-uniform
    float dstep[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fused) {
#pragma unroll
        for (int j = 0; j < 6; j++) dstep[j] = (float)Dn.prev_delta[n * 8 + j];
    }
    auto s_load = [&](Stage &S) {
        S.px = refl_idx(x00 + S.lx - 2, W); S.py = refl_idx(y00 + S.ly - 2, H);
        const int gi = S.py * W + S.px;
        S.tp = tgtpack[gi]; S.dep = depth_t[gi];
        if (fused) {
            const float4 *r = reinterpret_cast<const float4 *>(Dn.prev_rec + ((size_t)n * hw + gi) * 8);
            S.r0 = r[0]; S.r1 = r[1];
        }
    };
    auto s_warp = [&](Stage &S) {
        if (fused) S.dep = dense_advance(S.dep, S.r0, S.r1, dstep, Dn.rho_lo, Dn.rho_hi);
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
            float4 *rec = rec1 + (S.ly * W2 + S.lx) * 3;
            lds_write1(rec + 0, val.x, val.y, S.tp.x, S.tp.y);
            lds_write1(rec + 1, gx.x, gx.y, gy.x, gy.y);
            lds_write1(rec + 2, val.z, S.tp.z, gx.z, gy.z);
            lds_write1(aux + S.ly * W2 + S.lx, Wt, oob ? 0.f : 1.f, S.tp.w, 0.f);
        }
        if (own) {
            if (Dn.depth_next != nullptr && inimg) Dn.depth_next[(size_t)n * hw + (size_t)S.py * W + S.px] = S.dep;
            if (TRACE && P.trace != nullptr && inimg)    // bilinear cell parity now, mask / validity bits in phase 2b
                P.trace[(size_t)n * hw + (size_t)S.py * W + S.px] =
                    (unsigned short)((((S.px + (int)floorf(S.g.rx)) & 1) << 2) | (((S.py + (int)floorf(S.g.ry)) & 1) << 3));
            o_g = S.g;      // (the 7-column Jacobian is rebuilt from this in phase 2b: 14 registers less through phase 2a, which the pipelined window reads use)
            o_pd = pd; o_cd = cd; o_dgx = c.es * gx.w; o_dgy = c.es * gy.w; o_depth = S.dep;
            o_pad = !oob && !S.t.inside;
        }
    };
    {
        Stage A;
        A.lx = ox + 2; A.ly = oy + 2;
        if (tid < RING_THREADS) {
            Stage B;
            const int hi = min(tid, NRING - 1);   // index among the ring pixels, raster order skipping the tile interior
            // rows 0,1 and H2-2,H2-1 are full rows of W2; the middle TH rows contribute 4 pixels each (2 left, 2 right)
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

    // ---------------- phase 2a: residual + adjoint coefficients for tile + 1-pixel halo ----------------
    float o_valid = 0.f;
    float o_diff = 0.f, o_w = 0.f, o_m = 0.f, o_lxx = 0.f, o_lxy = 0.f, o_lyy = 0.f, o_l1x = 0.f, o_l1y = 0.f;
    float o_gx[3] = {0, 0, 0}, o_gy[3] = {0, 0, 0}, o_y[3] = {0, 0, 0}, o_x[3] = {0, 0, 0};
    constexpr int RA = (N1 + NT - 1) / NT;
#pragma unroll
    for (int r = 0; r < RA; r++) {
        int lx, ly;                          // coordinates in the 1-halo region
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
        const bool real = gx_ >= 0 && gx_ < W && gy_ >= 0 && gy_ < H;   // residual pixels must exist (no reflection here)
        const float4 *ctr = rec1 + ((ly + 1) * W2 + lx + 1) * 3;
        f32x4 q0, q1, q2;
        lds_read3v(ctr, q0, q1, q2);
        const f2 yc01 = q0.lo, xc01 = q0.hi, gxc01 = q1.lo, gyc01 = q1.hi, yx2c = q2.lo, g2c = q2.hi;
        const float yc[3] = {yc01.x, yc01.y, yx2c.x}, xc[3] = {xc01.x, xc01.y, yx2c.y};
        const float gxc[3] = {gxc01.x, gxc01.y, g2c.x}, gyc[3] = {gyc01.x, gyc01.y, g2c.y};
        float4 ax = lds_read1(aux + (ly + 1) * W2 + lx + 1);
        // window statistics, packed exactly as pass A of k_linearize
        f2 Sy01, Sx01, Syy01, Sxx01, Sxy01, Gx01, Gy01, S2, SS2, G2;
        float Sxy2;
        const float4 *nbA = ctr - (W2 + 1) * 3;
        // the first neighbour initialises the accumulators (as in k_linearize), the other eight are added -- same operations on the same values
        // in the same order in both forms below
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
#if TC_DENSE_PIPE_A
        {   // software-pipelined (round 5, third session): the three reads of window position k + 1 are in flight while position k is accumulated
            // (two register sets, counted waits: LDS returns in order; every position a compile-time offset from the window's first record)
            constexpr int RB = 48, ROWB = W2 * 48;
            const unsigned base = lds_addr(nbA);
            f32x4 u0, u1, u2, w0, w1, w2;
            // (the fence pins a position's accumulation in front of the reads that reuse its registers: left free, the scheduler keeps several
            // positions' records alive at once and the kernel no longer fits four waves per SIMD)
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
        }
#else
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
#endif
        ChanTerms<f2> t01;
        ChanTerms<float> t2;
        ssim_l1_channel<f2>(xc01, yc01, gxc01, gyc01, Sx01, Sy01, Sxx01, Syy01, Sxy01, P.ws, P.wl, P.eps, t01);
        ssim_l1_channel<float>(yx2c.y, yx2c.x, g2c.x, g2c.y, S2.y, S2.x, SS2.y, SS2.x, Sxy2, P.ws, P.wl, P.eps, t2);
        // d s/d y_q = cA' + cB (y_q - 1/2) + cC (x_q - 1/2): re-centred from the centre pixel to 1/2 so that p's record serves every q
        const float cB[3] = {t01.cB.x, t01.cB.y, t2.cB}, cC[3] = {t01.cC.x, t01.cC.y, t2.cC};
        float cA[3] = {t01.cA.x, t01.cA.y, t2.cA};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) cA[ch] += cB[ch] * (0.5f - yc[ch]) + cC[ch] * (0.5f - xc[ch]);
        const float e1 = t01.e1.x + t01.e1.y + t2.e1, e2 = t01.e2.x + t01.e2.y + t2.e2;
        const float l1x = t01.l1x.x + t01.l1x.y + t2.l1x, l1y = t01.l1y.x + t01.l1y.y + t2.l1y;
        float lxx = t01.lxx.x + t01.lxx.y + t2.lxx, lxy = t01.lxy.x + t01.lxy.y + t2.lxy, lyy = t01.lyy.x + t01.lyy.y + t2.lyy;
        {   // GN curvature of the SSIM term (as in k_linearize)
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
        float m = (real && ax.y > 0.5f && (!P.automask || diff < ax.z)) ? 1.f : 0.f;
        if (P.ext_diff != nullptr && n < P.n_ext)   // window mode: min-over-sources selection instead of the pair's own mask
            m = (real && ext_selected(P, n, gy_ * W + gx_, hw)) ? 1.f : 0.f;
        float w = m * ax.x;    // M_p W_p
        float4 *cr = coef + (ly * W1 + lx) * 3;
        lds_write1(cr + 0, w * cA[0], w * cA[1], w * cA[2], w * cB[0]);
        lds_write1(cr + 1, w * cB[1], w * cB[2], w * cC[0], w * cC[1]);
        lds_write1(cr + 2, w * cC[2], 0.f, 0.f, 0.f);
        if (r == 0) {
            o_valid = ax.y;
            o_diff = diff; o_w = w; o_m = m; o_lxx = lxx; o_lxy = lxy; o_lyy = lyy; o_l1x = l1x; o_l1y = l1y;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) { o_gx[ch] = gxc[ch]; o_gy[ch] = gyc[ch]; o_y[ch] = yc[ch]; o_x[ch] = xc[ch]; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---------------- phase 2b: adjoint gather, gradients, curvature blocks, per-pixel Schur elimination ----------------
    float v[L::NH + NP + 3];
#pragma unroll
    for (int i = 0; i < L::NH + NP + 3; i++) v[i] = 0.f;
    if (inimg) {
        if (TRACE && P.trace != nullptr) {   // parity tests replay these decisions in the float64 oracle
            unsigned short *tb = P.trace + (size_t)n * hw + gyo * W + gxo;     // (this thread's own phase-1 word)
            *tb = (unsigned short)(*tb | (o_m > 0.f ? 1 : 0) | (o_valid > 0.5f ? 2 : 0) | (sign_code(o_cd - o_pd) << 4) | (sign_code(o_y[0] - o_x[0]) << 6) |
                                   (sign_code(o_y[1] - o_x[1]) << 8) | (sign_code(o_y[2] - o_x[2]) << 10));
        }
        // reflect-pad multiplicity: how many window slots of residual pixel p map to this pixel q (losses.py:22).  It differs from 1
        // only next to the image border and factorises into a column and a row factor: formed ONCE per pixel (the first version
        // evaluated four compares and two selects per neighbour -- a third of the gather's instructions).
        const float mxl = (gxo == 1) ? 2.f : 1.f, mxr = (gxo == W - 2) ? 2.f : 1.f;
        const float myu = (gyo == 1) ? 2.f : 1.f, myd = (gyo == H - 2) ? 2.f : 1.f;
        const float yq[3] = {o_y[0] - 0.5f, o_y[1] - 0.5f, o_y[2] - 0.5f}, xq[3] = {o_x[0] - 0.5f, o_x[1] - 0.5f, o_x[2] - 0.5f};
        float sA[3] = {0, 0, 0}, sB[3] = {0, 0, 0}, sC[3] = {0, 0, 0};     // multiplicity-weighted sums of the 9 coefficient records
        auto gather = [&](const f32x4 &c0, const f32x4 &c1, const f32x4 &c2, const float fm) {
            sA[0] += fm * c0.x; sA[1] += fm * c0.y; sA[2] += fm * c0.z;
            sB[0] += fm * c0.w; sB[1] += fm * c1.x; sB[2] += fm * c1.y;
            sC[0] += fm * c1.z; sC[1] += fm * c1.w; sC[2] += fm * c2.x;
        };
#if TC_DENSE_PIPE_B
        {   // the nine coefficient records, raster order as the rolled loop below, the next record in flight under the current one's FMAs
            constexpr int RB = 48, ROWB = W1 * 48;
            const unsigned base = lds_addr(coef + (oy * W1 + ox) * 3);
            f32x4 u0, u1, u2, w0, w1, w2;
            auto fence = [&]() { asm volatile("" : "+v"(sA[0]), "+v"(sA[1]), "+v"(sA[2]), "+v"(sB[0]), "+v"(sB[1]), "+v"(sB[2]), "+v"(sC[0]), "+v"(sC[1]), "+v"(sC[2])); };
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
        }
#else
#pragma unroll 1
        for (int r = 0; r < 3; r++) {                      // rows rolled (a full unroll keeps all 27 LDS reads live: 172 VGPRs)
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
#endif
        geo_jac<7>(c, o_g, W, H, a, b, zc);
        // scale column (dXp = Xp - t) -> inverse-depth column: dXp/drho = -depth (Xp - t)
        a[6] *= -o_depth; b[6] *= -o_depth; zc[6] *= -o_depth;
        float lam[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) lam[ch] = sA[ch] + sB[ch] * yq[ch] + sC[ch] * xq[ch];
        // d C / d (ix, iy) of this pixel: SSIM adjoint + own L1 term (both carry M W of the residual pixel)
        float sx = o_w * o_l1x, sy = o_w * o_l1y;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { sx += lam[ch] * o_gx[ch]; sy += lam[ch] * o_gy[ch]; }
        // depth-consistency weight derivative (own pixel): -M diff d dd/d theta
        float sum = o_cd + o_pd, dif = o_cd - o_pd, isum = frcp(sum), raw = fabsf(dif) * isum;
        float sg = (raw >= 0.f && raw <= 1.f) ? (dif > 0.f ? 1.f : (dif < 0.f ? -1.f : 0.f)) : 0.f;
        float kdd = o_m * o_diff * sg * 2.f * isum * isum;
        float grow[7];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            float dpd = o_dgx * a[j] + o_dgy * b[j];
            grow[j] = sx * a[j] + sy * b[j] - kdd * (o_pd * zc[j] - o_cd * dpd);
        }
        // curvature blocks with Lam' = M W Lam
        float wxx = o_w * o_lxx, wxy = o_w * o_lxy, wyy = o_w * o_lyy;
        float la[7], lb[7];
#pragma unroll
        for (int j = 0; j < 7; j++) { la[j] = wxx * a[j] + wxy * b[j]; lb[j] = wxy * a[j] + wyy * b[j]; }
        float D = la[6] * a[6] + lb[6] * b[6];
        float g_rho = grow[6];
        if (Dn.w_prior > 0.f) {   // masked prior on the relative inverse-depth change
            float rho = frcp(o_depth), rho0 = frcp(Dn.depth0[(size_t)n * hw + gyo * W + gxo]);
            float ir2 = frcp(rho0 * rho0), dr = rho - rho0;
            g_rho += o_m * 2.f * Dn.w_prior * dr * ir2;
            D += o_m * 2.f * Dn.w_prior * ir2;
            v[L::NH + NP] += o_m * Dn.w_prior * dr * dr * ir2;     // prior cost rides in the sum(M W diff) slot
        }
        float Bq[6];
#pragma unroll
        for (int j = 0; j < 6; j++) Bq[j] = la[j] * a[6] + lb[j] * b[6];
        const float Dd = (1.f + Dn.lambda_depth) * D;
        // A pixel whose own sample is a blend with grid_sample's zero padding (valid up to half a pixel outside the source image,
        // stn.py:268-271) has image gradients of (colour / 1 px): its depth gradient is 50x an ordinary pixel's against ordinary
        // curvature, and the Gauss-Newton step there is noise (measured: 30-60 % of rho per iteration, chaotic from one iteration to
        // the next).  Such pixels keep their depth: no elimination, no back-substitution; their photometric terms stay in the pose system.
        const bool elim = Dd > 1e-30f && !o_pad;
        const float iD = elim ? frcp(Dd) : 0.f;
        int h = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            v[L::NH + j] = grow[j] - Bq[j] * g_rho * iD;
#pragma unroll
            for (int i = 0; i <= j; i++) { v[h] = la[j] * a[i] + lb[j] * b[i] - Bq[j] * Bq[i] * iD; h++; }
        }
        v[L::NH + NP] += o_w * o_diff;
        v[L::NH + NP + 1] = o_m;
        float *dr = Dn.dense_rec + ((size_t)n * hw + gyo * W + gxo) * 8;
        reinterpret_cast<float4 *>(dr)[0] = make_float4(g_rho, elim ? Dd : 0.f, Bq[0], Bq[1]);
        reinterpret_cast<float4 *>(dr)[1] = make_float4(Bq[2], Bq[3], Bq[4], Bq[5]);
    }
    block_reduce_publish<NP, L::NH + NP + 3, true, false, NT>(P, v, red, n, bid, nblk, tid);
    stamp_end(P.stamp, tid);
}

// back-substitution: drho_q = -(g_rho_q + B_q' dxi) / Dd_q ;  rho clamped to [1/max_depth, 1/min_depth]
struct DenseUpdateParams {
    const float *dense_rec;   // [N][H*W][8]
    const double *delta;      // [N][8] pose increment of this iteration (written by k_solve)
    const float *depth;       // [N][H*W] in
    float *depth_out;         // [N][H*W] out (may be the same buffer)
    int hw;
    float rho_lo, rho_hi;
    // coalesced calls (CoalTab, kernels.h): the refined map of batch pair n goes to ITS call's output, at the pair's index in that call
    int c_ncall, c_B, c_S, c_pad;
    float *c_depth_out[TC_MAX_COAL];
};

__global__ __launch_bounds__(256) void k_dense_update(DenseUpdateParams P) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int n = blockIdx.y;
    if (idx >= P.hw) return;
    const float4 *r = reinterpret_cast<const float4 *>(P.dense_rec + ((size_t)n * P.hw + idx) * 8);
    const float4 r0 = r[0], r1 = r[1];
    float d[6];
#pragma unroll
    for (int j = 0; j < 6; j++) d[j] = (float)P.delta[n * 8 + j];
    float *dst = P.depth_out + (size_t)n * P.hw;
    if (P.c_ncall > 0) { const CoalIdx ci = coal_index(P.c_ncall, P.c_B, P.c_S, n); dst = P.c_depth_out[ci.call] + (size_t)ci.li * P.hw; }
    dst[idx] = dense_advance(P.depth[(size_t)n * P.hw + idx], r0, r1, d, P.rho_lo, P.rho_hi);
}

// LM variant of the back-substitution: an accepted trial first becomes the accepted state (depth map and per-pixel records),
// then the next trial depth is formed from the ACCEPTED state with the step k_solve derived from the accepted system.
struct DenseLmParams {
    const float *rec_try;     // [N][H*W][8] records of the linearisation just evaluated
    float *rec_acc;           // [N][H*W][8] accepted records
    float *depth_acc;         // [N][H*W]    accepted depth
    float *depth;             // [N][H*W]    trial depth in/out
    const double *delta;      // [N][8]
    const int *accept;        // [N]
    int hw;
    float rho_lo, rho_hi;
};

__global__ __launch_bounds__(256) void k_dense_update_lm(DenseLmParams P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (idx >= P.hw) return;
    const size_t o = (size_t)n * P.hw + idx;
    float4 *ra = reinterpret_cast<float4 *>(P.rec_acc + o * 8);
    float4 r0, r1;
    float base;
    if (P.accept[n]) {
        const float4 *rt = reinterpret_cast<const float4 *>(P.rec_try + o * 8);
        r0 = rt[0]; r1 = rt[1]; base = P.depth[o];
        ra[0] = r0; ra[1] = r1; P.depth_acc[o] = base;
    } else {
        r0 = ra[0]; r1 = ra[1]; base = P.depth_acc[o];
    }
    float dep = base;
    if (r0.y > 0.f) {
        const double *d = P.delta + n * 8;
        float bd = r0.z * (float)d[0] + r0.w * (float)d[1] + r1.x * (float)d[2] + r1.y * (float)d[3] + r1.z * (float)d[4] + r1.w * (float)d[5];
        dep = 1.f / depth_step(1.f / base, -(r0.x + bd) / r0.y, P.rho_lo, P.rho_hi);
    }
    P.depth[o] = dep;
}

// after the final LM cost check: a pair whose last step was rejected falls back to its accepted depth map
__global__ __launch_bounds__(256) void k_dense_final_lm(const int *keep, const float *depth_acc, float *depth, int hw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (idx >= hw || keep[n]) return;
    depth[(size_t)n * hw + idx] = depth_acc[(size_t)n * hw + idx];
}

}  // namespace tc
