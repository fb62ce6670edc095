// scale_kernel.h -- DNet ground-plane scale recovery (models/dnet_layers.py:249-327).
//   k_ground   per pixel: backproject, 8-neighbour surface normal (4 normalised cross products, mean, normalise,
//              ReflectionPad2d(1) of the interior normal map), ground mask (|cos(n, y)| > cos 5 deg, y > 0), camera height
//              |P.n|; masked heights are emitted as order-preserving uint keys
//   k_sel_hist / k_sel_pick  exact lower median of all masked heights of the batch (torch.median) by an MSB-first
//              radix select: 4 rounds of 256-bin histograms with INTEGER atomics (deterministic), no sort
#pragma once
#include <hip/hip_runtime.h>

namespace tc {

struct GroundParams {
    const float *depth;   // [N][H*W]
    const float *K;       // [N][9] pinhole
    float *height, *mask; // optional outputs [N][H*W]
    unsigned *keys;       // [N][H*W] float bits of the height where masked, 0xFFFFFFFF elsewhere
    int H, W;
    int *err;             // host-mapped status word: set to 1 when an image's intrinsics are not pinhole (or null)
};

__device__ __forceinline__ void bp(const float *depth, int W, float ifx, float icx, float ify, float icy, int v, int u, float *p) {
    float d = depth[v * W + u];
    p[0] = d * (ifx * (float)u + icx); p[1] = d * (ify * (float)v + icy); p[2] = d;
}
__device__ __forceinline__ void cross_norm(const float *a, const float *b, const float *c, float *o) {
    float u0 = a[0] - c[0], u1 = a[1] - c[1], u2 = a[2] - c[2], w0 = b[0] - c[0], w1 = b[1] - c[1], w2 = b[2] - c[2];
    o[0] = u1 * w2 - u2 * w1; o[1] = u2 * w0 - u0 * w2; o[2] = u0 * w1 - u1 * w0;
    float n = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
    o[0] /= n; o[1] /= n; o[2] /= n;
}

__global__ __launch_bounds__(256) void k_ground(GroundParams P) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y, hw = P.H * P.W, H = P.H, W = P.W;
    if (idx >= hw) return;
    int v = idx / W, u = idx - v * W;
    const float *K = P.K + n * 9, *depth = P.depth + (size_t)n * hw;
    if (K[1] != 0.f || K[3] != 0.f || K[6] != 0.f || K[7] != 0.f || K[8] != 1.f || K[0] == 0.f || K[4] == 0.f) {
        // device-side guard of the pinhole contract (the host validates a given device buffer only once): no ground anywhere ->
        // the median and the scale come out NaN, and the status word turns the next call / synchronize into TCSFM_E_INTRINSICS
        if (P.err && idx == 0) *reinterpret_cast<volatile int *>(P.err) = 1;
        if (P.height) P.height[(size_t)n * hw + idx] = __uint_as_float(0x7fc00000u);
        if (P.mask) P.mask[(size_t)n * hw + idx] = 0.f;
        P.keys[(size_t)n * hw + idx] = 0xFFFFFFFFu;
        return;
    }
    const float ifx = 1.f / K[0], icx = -K[2] / K[0], ify = 1.f / K[4], icy = -K[5] / K[4];
    // normal of the reflect-mapped interior pixel (dnet_layers.py:289-290)
    int rv = v == 0 ? 2 : (v == H - 1 ? H - 3 : v), ru = u == 0 ? 2 : (u == W - 1 ? W - 3 : u);
    float c[3], a[3], b[3], n0[3], n1[3], n2[3], n3[3];
    bp(depth, W, ifx, icx, ify, icy, rv, ru, c);
    bp(depth, W, ifx, icx, ify, icy, rv, ru - 1, a); bp(depth, W, ifx, icx, ify, icy, rv - 1, ru, b); cross_norm(a, b, c, n0);
    bp(depth, W, ifx, icx, ify, icy, rv, ru + 1, a); bp(depth, W, ifx, icx, ify, icy, rv + 1, ru, b); cross_norm(a, b, c, n1);
    bp(depth, W, ifx, icx, ify, icy, rv - 1, ru - 1, a); bp(depth, W, ifx, icx, ify, icy, rv + 1, ru - 1, b); cross_norm(a, b, c, n2);
    bp(depth, W, ifx, icx, ify, icy, rv - 1, ru + 1, a); bp(depth, W, ifx, icx, ify, icy, rv + 1, ru + 1, b); cross_norm(a, b, c, n3);
    float m[3] = {(n0[0] + n1[0] + n2[0] + n3[0]) * 0.25f, (n0[1] + n1[1] + n2[1] + n3[1]) * 0.25f, (n0[2] + n1[2] + n2[2] + n3[2]) * 0.25f};
    float nn = fmaxf(sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]), 1e-12f);
    m[0] /= nn; m[1] /= nn; m[2] /= nn;
    float p[3];
    bp(depth, W, ifx, icx, ify, icy, v, u, p);
    float nrm = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    float cs = m[1] / fmaxf(nrm, 1e-6f);
    const float thr = 0.99619469809174555f;   // cos(5 deg)
    bool g = ((cs > thr) || (cs < -thr)) && (p[1] > 0.f);
    float h = fabsf(p[0] * m[0] + p[1] * m[1] + p[2] * m[2]);
    if (P.height) P.height[(size_t)n * hw + idx] = h;
    if (P.mask) P.mask[(size_t)n * hw + idx] = g ? 1.f : 0.f;
    P.keys[(size_t)n * hw + idx] = g ? __float_as_uint(h) : 0xFFFFFFFFu;
}

// state: [0] = k (rank still to find inside the current prefix bucket), [1] = prefix, [2] = count, [3] = unused
__global__ __launch_bounds__(256) void k_sel_hist(const unsigned *keys, int hw, int weight0, const unsigned *state, int shift, unsigned *hist) {
    __shared__ unsigned sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int n = blockIdx.y;
    const unsigned prefix = state[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        unsigned key = keys[(size_t)n * hw + i];
        if (key == 0xFFFFFFFFu) continue;
        if (shift < 24 && (key >> (shift + 8)) != (prefix >> (shift + 8))) continue;
        atomicAdd(&sh[(key >> shift) & 255u], n == 0 ? (unsigned)weight0 : 1u);   // image 0 may stand for several (batch padding)
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void k_sel_pick(unsigned *hist, unsigned *state, int shift, float real_cam_height, float *scale_out, float *median_out) {
    if (threadIdx.x != 0) return;
    unsigned k = state[0];
    if (shift == 24) {
        unsigned cnt = 0;
        for (int b = 0; b < 256; b++) cnt += hist[b];
        state[2] = cnt;
        k = cnt ? (cnt - 1) / 2 : 0;             // torch.median: lower median
    }
    unsigned cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; b++) {
        if (cum + hist[b] > k) { bin = b; break; }
        cum += hist[b];
    }
    state[0] = k - cum;
    state[1] = (shift == 24 ? 0u : state[1]) | ((unsigned)bin << shift);
    for (int b = 0; b < 256; b++) hist[b] = 0;
    if (shift == 0) {
        float med = state[2] ? __uint_as_float(state[1]) : __uint_as_float(0x7FC00000u);
        if (median_out) *median_out = med;
        *scale_out = real_cam_height / med;      // dnet_layers.py:325
    }
}

// ---------------------------------------------------------------------------------------------------------------
// get_smooth_loss, losses.py:43-61: edge-aware smoothness of the mean-normalised disparity,
//   mean |d_x (disp / mean disp)| exp(-mean_c |d_x img|)  +  the same in y.
// k_smooth_mean: per-image mean of the disparity (one workgroup per image, fp64, fixed order);
// k_smooth: per-pixel terms, one (sum_x, sum_y) partial per workgroup (fixed-order LDS tree); the host adds the partials in double.
__global__ __launch_bounds__(1024) void k_smooth_mean(const float *disp, int hw, double *mean) {
    const int n = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = tid; i < hw; i += 1024) s += (double)disp[(size_t)n * hw + i];
    red[tid] = s; __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) mean[n] = red[0] / hw;
}

__global__ __launch_bounds__(256) void k_smooth(const float *disp, const float *img, const double *mean, int H, int W, float *partial) {
    const int idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y, hw = H * W;
    float sx = 0.f, sy = 0.f;
    if (idx < hw) {
        const int v = idx / W, u = idx - v * W;
        const float inv = 1.f / ((float)mean[n] + 1e-7f);
        const float *d = disp + (size_t)n * hw, *im = img + (size_t)n * 3 * hw;
        const float d0 = d[idx] * inv;
        if (u < W - 1) {
            const float g = (fabsf(im[idx] - im[idx + 1]) + fabsf(im[hw + idx] - im[hw + idx + 1]) + fabsf(im[2 * hw + idx] - im[2 * hw + idx + 1])) * (1.f / 3.f);
            sx = fabsf(d0 - d[idx + 1] * inv) * __expf(-g);
        }
        if (v < H - 1) {
            const float g = (fabsf(im[idx] - im[idx + W]) + fabsf(im[hw + idx] - im[hw + idx + W]) + fabsf(im[2 * hw + idx] - im[2 * hw + idx + W])) * (1.f / 3.f);
            sy = fabsf(d0 - d[idx + W] * inv) * __expf(-g);
        }
    }
    __shared__ float rx[256], ry[256];
    rx[threadIdx.x] = sx; ry[threadIdx.x] = sy; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { rx[threadIdx.x] += rx[threadIdx.x + o]; ry[threadIdx.x] += ry[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { partial[((size_t)n * gridDim.x + blockIdx.x) * 2] = rx[0]; partial[((size_t)n * gridDim.x + blockIdx.x) * 2 + 1] = ry[0]; }
}

}  // namespace tc
