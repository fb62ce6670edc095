// posenet_kernel.h -- the reference's PoseNet (models/pose_models.py:88-147) on gfx950: seven weight-standardised stride-2
// convolutions (conv2d_wn, :10-25) each followed by GroupNorm(16) + ReLU (conv_gn, :62-82), a 1x1 head, spatial mean, x 0.01.
// It is called `iterations` times per window inside the coupled pose loop (train_mono.py:64,77), on the (tgt | src) /
// (tgt * valid | img_rec) 6-channel inputs the warp kernel assembles.
//
// Design (MI355X, not a cuDNN/MIOpen call chain):
//   * weights are frozen at test time: weight standardisation is done ONCE when the weights are loaded (k_pn_prep), which also
//     lays them out for the matrix cores;
//   * every convolution is an implicit GEMM  out[pixel][cout] = sum_k patch[pixel][k] w[k][cout]  on the fp32 matrix
//     instruction v_mfma_f32_16x16x4_f32 (exact fp32: the bf16 forms would not hold the 1e-5 parity bar).  One wave owns 16
//     output pixels x up to 64 output channels.  Operands go global -> registers with 16-byte loads and NO LDS staging: the A
//     operand of four consecutive K-steps is ONE float4 per lane (four consecutive input channels of the lane's pixel at the
//     current tap), the B operand one float4 per lane and 16-channel block from a weight image stored in exactly that order
//     (K is summed in a permuted order; any fixed order is a valid GEMM).  Activations are NHWC so that those float4 are contiguous;
//   * GroupNorm needs whole-image statistics, i.e. a grid-wide dependency between a convolution and its consumer: k_pn_stats
//     (one workgroup per image and group) reduces them deterministically into per-channel scale / shift, and the CONSUMER applies
//     normalisation + ReLU on the fly while loading its A operand -- the normalised activation is never written;
//   * the last layers have 120 / 30 / 10 output pixels per image and K up to 2304: they are split over K across workgroups
//     (partial sums reduced, in fixed order, by the same k_pn_stats pass);
//   * the first layer reads the caller's planar NCHW images directly (two 3-channel pointers per image: no concatenated
//     [N,6,H,W] tensor is needed for the first call of the loop), input normalisation (x - 0.45) / 0.22 fused.
#pragma once
#include <hip/hip_runtime.h>

namespace tc {

typedef float pn_f4 __attribute__((ext_vector_type(4)));

struct PnLayer {           // geometry of one convolution layer (host-filled)
    int cin, cout, ks, pad; // stride is 2 everywhere
    int ih, iw, oh, ow;     // input / output spatial size
    int ksplit;             // K-split factor (1 = none)
    int kgroups;            // number of 16-wide K groups: layer 1: 21 (6 ch x 7 rows x 8 padded columns / 16), else ks*ks*cin/16
};

// ---------------------------------------------------------------------------------------------------------------
// weight preparation: conv2d_wn's standardisation (pose_models.py:17-23: subtract the per-filter mean, divide by the UNBIASED
// per-filter standard deviation + 1e-5) and the matrix-core layout.  One workgroup per output channel.
//   generic layers (cin % 16 == 0): w4[((tap * cin/16 + c16) * 4 + kq) * cout + co] = float4 over t of w[co][ci = c16*16 + 4 kq + t][tap]
//   first layer (cin = 6, 7x7):      w4[(grp * 4 + kq) * cout + co] = float4 over t of w[co][ci][ky][kx = 4 (kq & 1) + t], combo = 2 grp + (kq >> 1)
//                                    = ci * 7 + ky, kx = 7 is the zero pad
__global__ __launch_bounds__(256) void k_pn_prep(const float *w, pn_f4 *w4, int cin, int cout, int ks, int first, int standardize) {
    const int co = blockIdx.x, tid = threadIdx.x, n = cin * ks * ks;
    const float *wc = w + (size_t)co * n;
    __shared__ double red[256];
    __shared__ float s_mean, s_inv;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)wc[i];
    red[tid] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float mean = (float)(red[0] / n);
    __syncthreads();
    double q = 0.0;
    for (int i = tid; i < n; i += 256) { const double d = (double)(wc[i] - mean); q += d * d; }
    red[tid] = q; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) { s_mean = standardize ? mean : 0.f; s_inv = standardize ? 1.f / ((float)sqrt(red[0] / (n - 1)) + 1e-5f) : 1.f; }
    __syncthreads();
    const float m = s_mean, inv = s_inv;
    auto W = [&](int ci, int ky, int kx) { return (wc[(ci * ks + ky) * ks + kx] - m) * inv; };
    if (first) {
        const int ngrp = (cin * ks + 1) / 2;                       // 21 for 6 x 7
        for (int e = tid; e < ngrp * 4; e += 256) {
            const int grp = e >> 2, kq = e & 3, combo = 2 * grp + (kq >> 1);
            pn_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (combo < cin * ks) {
                const int ci = combo / ks, ky = combo - ci * ks;
                for (int t = 0; t < 4; t++) { const int kx = 4 * (kq & 1) + t; if (kx < ks) v[t] = W(ci, ky, kx); }
            }
            w4[(size_t)(grp * 4 + kq) * cout + co] = v;
        }
    } else {
        const int c16n = cin / 16;
        for (int e = tid; e < ks * ks * c16n * 4; e += 256) {
            const int kq = e & 3, c16 = (e >> 2) % c16n, tap = (e >> 2) / c16n, ky = tap / ks, kx = tap - ky * ks;
            pn_f4 v;
            for (int t = 0; t < 4; t++) v[t] = W(c16 * 16 + 4 * kq + t, ky, kx);
            w4[(size_t)((tap * c16n + c16) * 4 + kq) * cout + co] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct PnConvParams {
    // first layer: two planar 3-channel images per sample.  Pair-stacked callers pass both with a per-sample stride; the window
    // form (win_B > 0) indexes the B targets / S*B sources like k_pack: sample n = s B + b forward (tgt b | src (s,b)), inverse swapped
    const float *imgA, *imgB;     // [*,3,IH,IW] planar
    long long strideA, strideB;   // floats between consecutive samples
    int win_B, win_S;
    WinOff win_off;
    // generic layers: NHWC raw convolution output of the previous layer + its per-sample, per-channel GroupNorm scale / shift
    const float *in;              // [N][IH][IW][CIN]
    const float *scsh;            // [N][CIN][2]  (scale, shift): a = relu(x * scale + shift)
    const pn_f4 *w4;              // prepared weights
    const float *bias;            // [COUT] or null
    float *out;                   // [ksplit][N][OH][OW][COUT] raw output (+ bias when ksplit == 1)
    float *part;                  // ksplit == 1: [N][tiles][COUT][2] per-workgroup (sum, sum of squares) of the outputs, for GroupNorm
    PnLayer L;
    int N;
};

// One wave = PB blocks of 16 output pixels x (16 NB) output channels; a workgroup = 4 waves = 64 PB consecutive output pixels of ONE sample.
// grid = (ceil(OH OW / (64 PB)), COUT / (16 NB), N * ksplit).  PB = 2 (round 4, the many-images regime, layers with >= 64 pixels): every
// weight / scale-shift float4 a lane loads feeds two pixel blocks -- 0.375 instead of 0.625 loads per MFMA in layer 2, 0.25 instead of 0.44
// in the 64-channel-block layers -- and the per-group address work is shared.
template <int NB, bool FIRST, int PB = 1>
__global__ __launch_bounds__(256) void k_pn_conv(PnConvParams P) {
    static_assert(!FIRST || PB == 1, "the first layer's generic form has one pixel block per wave");
    const PnLayer &L = P.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int ksp = blockIdx.z % L.ksplit, n = blockIdx.z / L.ksplit;
    const int npix = L.oh * L.ow;
    const int pix0 = ((blockIdx.x * 4 + wave) * PB) * 16;         // first output pixel of this wave's first block
    const int pix = pix0 + m;                                     // this lane's output pixel in block 0 (A operand row)
    const bool pvalid = pix < npix;
    const int oy = pvalid ? pix / L.ow : 0, ox = pvalid ? pix - (pix / L.ow) * L.ow : 0;
    const int cbase = blockIdx.y * 16 * NB;
    pn_f4 acc[PB][NB];
#pragma unroll
    for (int p = 0; p < PB; p++)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[p][b] = (pn_f4){0.f, 0.f, 0.f, 0.f};
    // K groups of this split: contiguous ranges.  A wave whose pixel blocks all lie past the image (the last layers have 30 and 10 pixels
    // per sample: three of a workgroup's four waves) has nothing to multiply: it skips the loop (round 4; it used to run every K group on
    // zeros) and only takes part in the epilogue's statistics barrier
    const int g0 = (L.kgroups * ksp) / L.ksplit, g1 = pix0 < npix ? (L.kgroups * (ksp + 1)) / L.ksplit : g0;
    // The K loop is latency-bound, not matrix-bound (a layer has only a few hundred waves, far fewer than would hide a global load
    // behind other waves' MFMAs): the loads of GC consecutive K groups are issued together, then their 16 GC MFMAs run -- the
    // compiler keeps the whole batch of loads in flight (one wait per batch instead of one per group).
    if (FIRST) {
        constexpr int GC = 7;                                    // 21 groups = 3 batches
        const int SB = P.win_S * P.win_B;
        const float *pa, *pb;
        if (P.win_B > 0) {
            const int inv = n >= SB, q = inv ? n - SB : n, b = q % P.win_B;
            const float *t = P.imgA + (size_t)b * P.strideA, *s = P.imgB + (size_t)win_src_image(P.win_off, q, P.win_B) * P.strideB;
            pa = inv ? s : t; pb = inv ? t : s;
        } else { pa = P.imgA + (size_t)n * P.strideA; pb = P.imgB + (size_t)n * P.strideB; }
        const int hw = L.ih * L.iw;
        for (int gb = g0; gb < g1; gb += GC) {
            pn_f4 a[GC], b4[GC][NB];
#pragma unroll
            for (int u = 0; u < GC; u++) {
                const int g = gb + u;
                const bool gok = g < g1;
                const int combo = 2 * (gok ? g : g0) + (kq >> 1);    // (ci, ky)
                const int ci = combo / 7, ky = combo - ci * 7;
                const int iy = oy * 2 + ky - 3, ix0 = ox * 2 - 3 + 4 * (kq & 1);
                const bool rowok = gok && pvalid && iy >= 0 && iy < L.ih;
                const float *row = (ci < 3 ? pa + (size_t)ci * hw : pb + (size_t)(ci - 3) * hw) + (size_t)(rowok ? iy : 0) * L.iw;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int ix = ix0 + t;
                    const bool ok = rowok && ix >= 0 && ix < L.iw;
                    const float v = row[ok ? ix : 0];
                    a[u][t] = ok ? (v - 0.45f) * (1.f / 0.22f) : 0.f;   // (imgs - 0.45) / 0.22, pose_models.py:125; zero padding of the normalised image
                }
#pragma unroll
                for (int b = 0; b < NB; b++) b4[u][b] = P.w4[(size_t)((gok ? g : g0) * 4 + kq) * L.cout + cbase + b * 16 + m];
            }
            __builtin_amdgcn_sched_barrier(0);       // all loads of the batch are issued before the first MFMA (hipcc otherwise re-serialises them)
#pragma unroll
            for (int u = 0; u < GC; u++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int b = 0; b < NB; b++) acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b4[u][b][t], acc[0][b], 0, 0, 0);
        }
    } else {
        // Addressing without divisions (round 4): the K group index is walked as (ky, kx, c16) counters, every address is a per-lane base
        // (fixed for the whole loop) plus a wave-uniform offset, and the zero-padding test of a tap is a bit of two per-lane masks.  The first
        // version divided the group index by c16n and ks for EVERY group -- two ~30-instruction scalar divisions in front of every batch of
        // loads, on the wave's critical path, as long as the group's MFMAs -- and rebuilt every 64-bit address on the vector ALU
        // (1 586 VALU and 1 430 SALU per wave beside 200 MFMAs in layer 2: profiles/r02b_posenet16_pmc_summary.json).
        constexpr int GC = NB * PB >= 8 ? 2 : (NB * PB == 4 ? 3 : 4);      // (batch sizes that keep the kernel at four waves per SIMD)
        const int c16n = L.cin / 16;
        const float *in = P.in + (size_t)n * L.ih * L.iw * L.cin;
        const pn_f4 *scsh = reinterpret_cast<const pn_f4 *>(P.scsh + (size_t)n * L.cin * 2) + 2 * kq;   // [(scale, shift) pairs]: 2 channels per float4
        unsigned rowm[PB], colm[PB];                      // bit k: input row iy0 + k / column ix0 + k lies inside the image (ks <= 7)
        int lane_in[PB];                                  // this lane's input offset of tap (0, 0), channel block 0 (may be negative)
#pragma unroll
        for (int p = 0; p < PB; p++) {
            const int px = pix + 16 * p;
            const bool pv = px < npix;
            const int oy_ = pv ? px / L.ow : 0, ox_ = pv ? px - (px / L.ow) * L.ow : 0;
            const int iy0 = oy_ * 2 - L.pad, ix0 = ox_ * 2 - L.pad;
            rowm[p] = 0u; colm[p] = 0u;
            for (int k = 0; k < L.ks; k++) {
                rowm[p] |= (pv && iy0 + k >= 0 && iy0 + k < L.ih) ? (1u << k) : 0u;
                colm[p] |= (ix0 + k >= 0 && ix0 + k < L.iw) ? (1u << k) : 0u;
            }
            lane_in[p] = (iy0 * L.iw + ix0) * L.cin + 4 * kq;
        }
        const pn_f4 *wl = P.w4 + (size_t)kq * L.cout + cbase + m;       // this lane's weight base: group g, block b at wl[g * 4 * cout + b * 16]
        // start of this K split, once per wave
        int tap = g0 / c16n, c16 = g0 - tap * c16n, ky = tap / L.ks, kx = tap - ky * L.ks;
        for (int gb = g0; gb < g1; gb += GC) {
            pn_f4 a[GC][PB], s01[GC], s23[GC], b4[GC][NB];
            bool ok[GC][PB];
#pragma unroll
            for (int u = 0; u < GC; u++) {
                const bool live = gb + u < g1;                                   // (wave-uniform)
                const int g = live ? gb + u : g0;
                const int uoff = live ? (ky * L.iw + kx) * L.cin + c16 * 16 : 0;    // wave-uniform part of the input offset
#pragma unroll
                for (int p = 0; p < PB; p++) {
                    ok[u][p] = live && ((rowm[p] >> (live ? ky : 0)) & (colm[p] >> (live ? kx : 0)) & 1u);
                    a[u][p] = *reinterpret_cast<const pn_f4 *>(in + (ok[u][p] ? lane_in[p] + uoff : 0));
                }
                const int cq = live ? c16 * 8 : 0;                                // (c16 * 16 + 4 kq) / 2 float4 of (scale, shift) pairs
                s01[u] = scsh[cq]; s23[u] = scsh[cq + 1];                         // (sc0, sh0, sc1, sh1), (sc2, sh2, sc3, sh3)
#pragma unroll
                for (int b = 0; b < NB; b++) b4[u][b] = wl[(size_t)g * 4 * L.cout + b * 16];
                if (live) { c16++; if (c16 == c16n) { c16 = 0; kx++; if (kx == L.ks) { kx = 0; ky++; } } }
            }
            __builtin_amdgcn_sched_barrier(0);       // all loads of the batch are issued before the first use
#pragma unroll
            for (int u = 0; u < GC; u++)       // GroupNorm + ReLU of the producer, zero padding AFTER it; a group past the end contributes 0
#pragma unroll
                for (int p = 0; p < PB; p++) {
                    a[u][p][0] = ok[u][p] ? fmaxf(a[u][p][0] * s01[u][0] + s01[u][1], 0.f) : 0.f;
                    a[u][p][1] = ok[u][p] ? fmaxf(a[u][p][1] * s01[u][2] + s01[u][3], 0.f) : 0.f;
                    a[u][p][2] = ok[u][p] ? fmaxf(a[u][p][2] * s23[u][0] + s23[u][1], 0.f) : 0.f;
                    a[u][p][3] = ok[u][p] ? fmaxf(a[u][p][3] * s23[u][2] + s23[u][3], 0.f) : 0.f;
                }
#pragma unroll
            for (int u = 0; u < GC; u++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int p = 0; p < PB; p++)
#pragma unroll
                        for (int b = 0; b < NB; b++) acc[p][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][p][t], b4[u][b][t], acc[p][b], 0, 0, 0);
        }
    }
    // C/D layout of the 16x16 tile: column (output channel) = lane & 15, row (pixel) = 4 (lane >> 4) + reg
    float *out = P.out + ((size_t)ksp * P.N + n) * npix * L.cout;
    __shared__ float wsum[4][16 * NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int co = cbase + b * 16 + m;
        const float bs = (P.bias != nullptr && L.ksplit == 1) ? P.bias[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < PB; p++) {
            const int prow0 = pix0 + 16 * p + 4 * kq;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (prow0 + r < npix) {
                    const float v = acc[p][b][r] + bs;
                    out[(size_t)(prow0 + r) * L.cout + co] = v;
                    s1 += v; s2 += v * v;
                }
        }
        // GroupNorm partial sums of this workgroup's 64 PB pixels, per channel: the 4 row groups of the wave (lanes m, m+16, m+32,
        // m+48), then the 4 waves through LDS, both in fixed order
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (kq == 0) { wsum[wave][b * 16 + m][0] = s1; wsum[wave][b * 16 + m][1] = s2; }
    }
    if (L.ksplit == 1 && P.part != nullptr) {
        __syncthreads();
        const int c = threadIdx.x;
        if (c < 16 * NB) {
            const float t1 = (wsum[0][c][0] + wsum[1][c][0]) + (wsum[2][c][0] + wsum[3][c][0]);
            const float t2 = (wsum[0][c][1] + wsum[1][c][1]) + (wsum[2][c][1] + wsum[3][c][1]);
            float *pp = P.part + (((size_t)n * gridDim.x + blockIdx.x) * L.cout + cbase + c) * 2;
            pp[0] = t1; pp[1] = t2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm statistics of one layer's raw output (+ the fixed-order reduction of its K-split partial sums, + bias for split
// layers), one workgroup per (sample, group): mean and biased variance over the group's channels and all pixels (torch
// GroupNorm), eps 1e-5, folded with the affine parameters into per-channel  scale = rstd gamma,  shift = beta - mean scale.
// ---------------------------------------------------------------------------------------------------------------
// First layer (6 -> 16 channels, 7x7, stride 2) with the input patch staged in LDS.  In the generic form every lane fetched its
// 4 x 21 input values with bounds-checked scalar loads from global memory and each value was fetched ~12 times per workgroup: the
// layer cost 24 us whatever the matrix cores did.  Here a workgroup owns 64 consecutive output pixels of TWO output rows: the
// 9 input rows x 134 columns x 6 channels behind them are loaded once, coalesced, normalised ((x - 0.45) / 0.22, zero outside the
// image -- the padding of the NORMALISED image, pose_models.py:125) into LDS; the 21 weight float4 of a lane are loaded up front and
// serve both rows (two independent accumulator chains per wave); the K loop runs on LDS reads and MFMAs only.  Same K order and
// the same arithmetic per output as k_pn_conv<1, true>.
// grid = (ceil(OH / 2) * ceil(OW / 64), 1, N); GroupNorm partial sums per workgroup as in k_pn_conv (tiles = gridDim.x).
constexpr int PN1_COLS = 136;      // 2 * 63 + 7 + 1 = 134 staged columns, padded (even: 8-byte aligned pairs)
constexpr int PN1_ROWS = 9;        // two output rows per workgroup: input rows 2 oy0 - 3 .. 2 oy0 + 5
__global__ __launch_bounds__(256) void k_pn_conv1(PnConvParams P) {
    const PnLayer &L = P.L;
    __shared__ __attribute__((aligned(16))) float patch[6 * PN1_ROWS * PN1_COLS];
    __shared__ float wsum[4][16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int chunks = (L.ow + 63) / 64;
    const int oyp = blockIdx.x / chunks, ox0 = (blockIdx.x - oyp * chunks) * 64;
    const int oy0 = 2 * oyp;                             // output rows oy0 and oy0 + 1 (the second may be past the end)
    const int n = blockIdx.z;
    const int SB = P.win_S * P.win_B;
    const float *pa, *pb;
    if (P.win_B > 0) {
        const int inv = n >= SB, q = inv ? n - SB : n, b = q % P.win_B;
        const float *t = P.imgA + (size_t)b * P.strideA, *s_ = P.imgB + (size_t)win_src_image(P.win_off, q, P.win_B) * P.strideB;
        pa = inv ? s_ : t; pb = inv ? t : s_;
    } else { pa = P.imgA + (size_t)n * P.strideA; pb = P.imgB + (size_t)n * P.strideB; }
    const int hw = L.ih * L.iw;
    // stage the patch: rows iy = 2 oy0 - 3 + row, columns ix = 2 ox0 - 3 + col
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    {   // all of a thread's loads are issued before the first is used (a rolled loop would pay one memory latency per element).
        // Round 4: a thread keeps ONE column and walks rows, so that plane, row and the row's bounds test are compile-time / wave-uniform and
        // only the column's test and offset are per lane (the first version decoded (plane, row, column) from a linear element index with two
        // divisions per element: ~750 of the kernel's 1 000 non-matrix VALU instructions per wave).  Columns 0..127: thread = (column, half),
        // half 0 walks the 27 rows of image A's three planes, half 1 those of image B (27 = 3 planes x 9 rows); the last 8 columns x 54 rows
        // go round once more, two elements per thread.
        static_assert(PN1_ROWS == 9 && PN1_COLS == 136, "staging pattern below");
        const int colA = tid & 127, half = tid >> 7;
        const float *srcA = half ? pb : pa;
        const int ixA = ix0 + colA;
        const bool okxA = ixA >= 0 && ixA < L.iw;
        float vA[27], vB[2];
        bool okA[27], okB[2];
#pragma unroll
        for (int j = 0; j < 27; j++) {
            const int pl = j / 9, ry = j - 9 * pl, iy = iy0 + ry;           // (compile-time plane / row; iy wave-uniform)
            okA[j] = okxA && iy >= 0 && iy < L.ih;
            vA[j] = srcA[(size_t)pl * hw + (okA[j] ? iy * L.iw + ixA : 0)];
        }
        int rrB[2], colB[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = tid + 256 * j;
            rrB[j] = e >> 3; colB[j] = 128 + (e & 7);
            const bool live = rrB[j] < 6 * PN1_ROWS;
            const int rr = live ? rrB[j] : 0, ci = rr / 9, ry = rr - 9 * ci, iy = iy0 + ry, ix = ix0 + colB[j];
            okB[j] = live && iy >= 0 && iy < L.ih && ix >= 0 && ix < L.iw;
            const float *src = (ci < 3 ? pa + (size_t)ci * hw : pb + (size_t)(ci - 3) * hw);
            vB[j] = src[okB[j] ? iy * L.iw + ix : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 27; j++)
            patch[((3 * half + j / 9) * PN1_ROWS + (j % 9)) * PN1_COLS + colA] = okA[j] ? (vA[j] - 0.45f) * (1.f / 0.22f) : 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (rrB[j] < 6 * PN1_ROWS) patch[rrB[j] * PN1_COLS + colB[j]] = okB[j] ? (vB[j] - 0.45f) * (1.f / 0.22f) : 0.f;
    }
    // weights of this lane: group g, quarter kq, output channel m (21 float4, L2-resident: every wave reads the same 21 KB).  Round 4: they are
    // loaded in three chunks of seven groups, the next chunk in flight under the current chunk's MFMAs, and only after the staging values have
    // gone to LDS -- 56 instead of 84 registers of weights that no longer coexist with the staging set (132 -> about 90 VGPRs: the kernel runs
    // five waves per SIMD, what its 30 KB of LDS allow, instead of three)
    pn_f4 bA[7], bB[7];
    auto wload = [&](pn_f4 *b, int g0_) {
#pragma unroll
        for (int g = 0; g < 7; g++) b[g] = P.w4[(size_t)((g0_ + g) * 4 + kq) * L.cout + m];
    };
    wload(bA, 0);
    __syncthreads();
    const int oxl = wave * 16 + m;                       // this lane's output pixel within the chunk (A operand row)
    pn_f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // output rows oy0 / oy0 + 1: two independent MFMA chains, one set of weights
    auto wmul = [&](const pn_f4 *b, int g0_) {
#pragma unroll
        for (int gg = 0; gg < 7; gg++) {
            // (ci, ky) of combo = 2 g + (kq >> 1): two compile-time row offsets, one select (no division by 7 per group)
            const int g = g0_ + gg, c0 = 2 * g, c1 = 2 * g + 1;          // (constants once the loops are unrolled)
            const int roff = (kq >> 1) ? ((c1 / 7) * PN1_ROWS + (c1 % 7)) * PN1_COLS : ((c0 / 7) * PN1_ROWS + (c0 % 7)) * PN1_COLS;
            const float *row = patch + roff + 2 * oxl + 4 * (kq & 1);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const float2 lo = *reinterpret_cast<const float2 *>(row + 2 * r * PN1_COLS), hi = *reinterpret_cast<const float2 *>(row + 2 * r * PN1_COLS + 2);
                const pn_f4 a = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int t = 0; t < 4; t++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[gg][t], acc[r], 0, 0, 0);
            }
        }
    };
    wload(bB, 7);
    __builtin_amdgcn_sched_barrier(0);
    wmul(bA, 0);
    __builtin_amdgcn_sched_barrier(0);
    wload(bA, 14);
    __builtin_amdgcn_sched_barrier(0);
    wmul(bB, 7);
    __builtin_amdgcn_sched_barrier(0);
    wmul(bA, 14);
    // C/D layout: column (output channel) = lane & 15, row (pixel) = 4 (lane >> 4) + reg
    float *out = P.out + (size_t)n * L.oh * L.ow * L.cout;
    const float bs = P.bias != nullptr ? P.bias[m] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ox = ox0 + wave * 16 + 4 * kq + q, oy = oy0 + r;
            if (ox < L.ow && oy < L.oh) {
                const float v = acc[r][q] + bs;
                out[((size_t)oy * L.ow + ox) * L.cout + m] = v;
                s1 += v; s2 += v * v;
            }
        }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (kq == 0) { wsum[wave][m][0] = s1; wsum[wave][m][1] = s2; }
    if (P.part != nullptr) {
        __syncthreads();
        if (tid < 16) {
            const float t1 = (wsum[0][tid][0] + wsum[1][tid][0]) + (wsum[2][tid][0] + wsum[3][tid][0]);
            const float t2 = (wsum[0][tid][1] + wsum[1][tid][1]) + (wsum[2][tid][1] + wsum[3][tid][1]);
            float *pp = P.part + (((size_t)n * gridDim.x + blockIdx.x) * L.cout + tid) * 2;
            pp[0] = t1; pp[1] = t2;
        }
    }
}

struct PnStatsParams {
    const float *part;      // ksplit == 1: [N][tiles][cout][2] partial sums written by the convolution's epilogue
    int tiles;
    float *out;             // [ksplit][N][npix][cout] raw (reduced in place into split 0 when ksplit > 1)
    const float *bias;      // [cout] or null (added here when ksplit > 1)
    const float *gamma, *beta;   // [cout] GroupNorm affine, or null (1, 0)
    float *scsh;            // [N][cout][2]
    int N, npix, cout, ksplit;
};

__global__ __launch_bounds__(256) void k_pn_stats(PnStatsParams P) {
    const int n = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int cg = P.cout / 16;                               // channels per group
    const size_t plane = (size_t)P.N * P.npix * P.cout;
    float *x = P.out + (size_t)n * P.npix * P.cout;
    __shared__ double r1[256], r2[256];
    double s = 0.0, q = 0.0;
    const int total = P.npix * cg;
    if (P.ksplit == 1 && P.part != nullptr) {       // fixed-order sum of the workgroup partials of this group's channels
        const float *pp = P.part + (size_t)n * P.tiles * P.cout * 2;
        for (int e = tid; e < P.tiles * cg; e += 256) {
            const int t = e / cg, c = g * cg + (e - t * cg);
            s += (double)pp[((size_t)t * P.cout + c) * 2]; q += (double)pp[((size_t)t * P.cout + c) * 2 + 1];
        }
    } else
    for (int e0 = tid; e0 < total; e0 += 1024) {       // 4 elements per trip: their loads (all K-split planes) are independent
        float v[4];
        float *xp[4];
        int cc[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = e0 + 256 * u;
            ok[u] = e < total;
            const int p = ok[u] ? e / cg : 0, c = g * cg + (ok[u] ? e - p * cg : 0);
            xp[u] = x + (size_t)p * P.cout + c;
            cc[u] = c;
            v[u] = xp[u][0];
        }
        if (P.ksplit > 1) {          // partial sums of the K split, added in fixed order (k ascending)
            for (int k0 = 1; k0 < P.ksplit; k0 += 8) {       // 8 planes x 4 elements in flight, then added in plane order
                float a[8][4];
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int u = 0; u < 4; u++) a[j][u] = (k0 + j < P.ksplit) ? xp[u][(size_t)(k0 + j) * plane] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] += a[j][u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (P.bias) v[u] += P.bias[cc[u]];
                if (ok[u]) xp[u][0] = v[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (ok[u]) { s += (double)v[u]; q += (double)v[u] * (double)v[u]; }
    }
    r1[tid] = s; r2[tid] = q; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; } __syncthreads(); }
    const double mean = r1[0] / total, var = fmax(r2[0] / total - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (tid < cg) {
        const int c = g * cg + tid;
        const float sc = rstd * (P.gamma ? P.gamma[c] : 1.f);
        P.scsh[((size_t)n * P.cout + c) * 2] = sc;
        P.scsh[((size_t)n * P.cout + c) * 2 + 1] = (P.beta ? P.beta[c] : 0.f) - (float)mean * sc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// head: GroupNorm + ReLU of conv7, 1x1 convolution 256 -> 6 (+ bias), spatial mean, x 0.01 (pose_models.py:135-137); optionally
// accumulates into the running pose (full_poses += correction, train_mono.py:78) and records the iterate.  One workgroup per sample.
struct PnHeadParams {
    const float *x;         // [N][npix][256] raw conv7
    const float *scsh;      // [N][256][2]
    const float *w, *b;     // [6][256], [6]
    float *pose;            // [N][6]
    float *stacked;         // optional [N][iters][6] (row `it` receives the pose after this call), or null
    int npix, accumulate, it, iters;
};

__global__ __launch_bounds__(256) void k_pn_head(PnHeadParams P) {
    const int n = blockIdx.x, c = threadIdx.x;
    __shared__ float feat[256];
    const float sc = P.scsh[((size_t)n * 256 + c) * 2], sh = P.scsh[((size_t)n * 256 + c) * 2 + 1];
    float s = 0.f;
    for (int p = 0; p < P.npix; p++) s += fmaxf(P.x[((size_t)n * P.npix + p) * 256 + c] * sc + sh, 0.f);
    feat[c] = s / (float)P.npix;                              // the mean commutes with the 1x1 convolution
    __syncthreads();
    __shared__ float part[6][4];
    const int o = c >> 6 < 4 ? c & 63 : 0;
    // 6 outputs: wave w sums a quarter of the channels for every output, fixed order
    const int wave = c >> 6, lane = c & 63;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    const int ch = wave * 64 + lane;
#pragma unroll
    for (int j = 0; j < 6; j++) acc[j] = feat[ch] * P.w[j * 256 + ch];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float v = acc[j];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if (lane == 0) part[j][wave] = v;
    }
    (void)o;
    __syncthreads();
    if (c < 6) {
        float v = 0.01f * (part[c][0] + part[c][1] + part[c][2] + part[c][3] + P.b[c]);
        if (P.accumulate) v += P.pose[n * 6 + c];
        P.pose[n * 6 + c] = v;
        if (P.stacked) P.stacked[((size_t)n * P.iters + P.it) * 6 + c] = v;
    }
}

}  // namespace tc
