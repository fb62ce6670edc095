// grid_barrier.hip -- what would ONE cooperative launch for all Gauss-Newton iterations of a B=1 call cost per iteration?
//
// VERDICT r01 item 3 proposes: 480 workgroups (240 tiles x 2 directed pairs, all co-resident at 2 per CU), a grid barrier per
// iteration, and EVERY workgroup redundantly reducing its pair's 240 records and solving the 6x6 (no last-arriver tail).
// Before restructuring k_linearize / k_solve around that, this measures the primitive on the real launch shape:
//   per iteration:  [stand-in for the linearisation: `spin_us` of dependent VALU work]  ->  record (32 floats) written through
//                   -> grid barrier -> every workgroup reads its pair's 240 records (30 KB) and sums them in fp64, fixed order
// Barrier variants:  0 = none (lower bound: record write + redundant reduce only, reads stale data)
//                    1 = one agent-scope counter per pair (atomic add, spin on an atomic load)
//                    2 = one flag per workgroup (plain store after a release fence), every workgroup polls its pair's 240 flags
//                    3 = as 2, polling back to back (no s_sleep between polls)
// Every spin is bounded (1 ms of s_memrealtime): a wave that does not see the barrier open sets an error word and leaves.
// Output: one JSON line per variant: microseconds per iteration above the stand-in work (GPU's own clock: earliest workgroup start
// -> latest workgroup end, as tcsfm_profile_kernel_time does), median of `reps` launches.
//
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scripts/experiments/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 512, WG_PER_PAIR = 240, PAIRS = 2, NREC = 32, ITERS = 4;
constexpr long long TIMEOUT_TICKS = 100000;     // 1 ms at 100 MHz

struct Params {
    float *rec;                 // [2 (ping-pong)][PAIRS][WG_PER_PAIR][NREC]
    unsigned *counter;          // [PAIRS]: monotonically increasing arrival counter (variant 1)
    unsigned *flag;             // [PAIRS][WG_PER_PAIR]: last iteration this workgroup has published (variant 2)
    double *out;                // [PAIRS][WG_PER_PAIR][NREC]: every workgroup's redundant sum (checked on the host)
    unsigned long long *stamp;  // [workgroups][2]
    int *err;
    int spin_iters;             // stand-in work: dependent fma chain length
    unsigned epoch0;            // flags / counters are never reset between launches: launch k works on epochs epoch0+1..epoch0+ITERS
};

template <int VARIANT>
__global__ __launch_bounds__(NT, 2) void k_fused(Params P) {
    extern __shared__ float lds[];              // 70 KB: the production kernel's footprint, so that 2 workgroups share a CU
    __shared__ double part[8][NREC];
    const int tid = threadIdx.x, wg = blockIdx.x, pair = blockIdx.y;
    const int gid = pair * WG_PER_PAIR + wg;
    if (tid == 0) P.stamp[2 * gid] = wall_clock64();
    float acc = (float)tid * 1e-3f;
    for (int it = 0; it < ITERS; it++) {
        // stand-in for the tile's linearisation
        for (int k = 0; k < P.spin_iters; k++) acc = fmaf(acc, 1.0000001f, 1e-9f);
        lds[tid] = acc;
        __syncthreads();
        // the workgroup's record, written through to memory (agent scope)
        float *rec = P.rec + ((size_t)(it & 1) * PAIRS * WG_PER_PAIR + gid) * NREC;
        if (tid < NREC) __hip_atomic_store(rec + tid, lds[tid] + (float)(it + gid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned epoch = P.epoch0 + it + 1;
        if (VARIANT == 1) {
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                __hip_atomic_fetch_add(P.counter + pair, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = epoch * WG_PER_PAIR;
                const long long t0 = wall_clock64();
                while ((int)(__hip_atomic_load(P.counter + pair, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                    if (wall_clock64() - t0 > TIMEOUT_TICKS) { *P.err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        } else if (VARIANT == 2 || VARIANT == 3) {
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                __hip_atomic_store(P.flag + gid, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 64) {                                   // wave 0 polls the pair's 240 flags, 4 per lane
                const long long t0 = wall_clock64();
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int w = tid + 64 * j;
                        if (w < WG_PER_PAIR)
                            ok &= (int)(__hip_atomic_load(P.flag + pair * WG_PER_PAIR + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
                    }
                    if (__all(ok)) break;
                    if (wall_clock64() - t0 > TIMEOUT_TICKS) { if (tid == 0) *P.err = 1; break; }
                    if (VARIANT == 2) __builtin_amdgcn_s_sleep(1);
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
            __syncthreads();
        } else {
            __syncthreads();
        }
        // redundant reduction of the pair's records: 512 threads = 16 record subsets x 32 entries, fixed order, then 16 -> 1
        {
            const float *base = P.rec + ((size_t)(it & 1) * PAIRS + pair) * WG_PER_PAIR * NREC;
            const int e = tid & 31, sub = tid >> 5;           // 16 subsets of 15 records
            float v[15];
#pragma unroll
            for (int j = 0; j < 15; j++)
                v[j] = __hip_atomic_load(base + (size_t)(sub * 15 + j) * NREC + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 15; j++) s += (double)v[j];
            double *sh = reinterpret_cast<double *>(lds + 1024);
            sh[sub * 32 + e] = s;
            __syncthreads();
            if (tid < 32) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 16; j++) t += sh[j * 32 + tid];
                part[0][tid] = t;
                if (it == ITERS - 1) P.out[(size_t)gid * NREC + tid] = t;
                acc += (float)t * 1e-30f;                     // the next "linearisation" depends on the solve
            }
            __syncthreads();
            acc += (float)part[0][tid & 31] * 1e-30f;
        }
    }
    if (acc == 123.456f) P.out[0] = acc;
    __syncthreads();
    if (tid == 0) P.stamp[2 * gid + 1] = wall_clock64();
}

template <int VARIANT>
static double run(Params P, int reps, std::vector<unsigned long long> &h, unsigned &epoch, bool check) {
    const int nwg = PAIRS * WG_PER_PAIR;
    std::vector<double> us;
    for (int r = 0; r < reps + 3; r++) {
        P.epoch0 = epoch;
        hipLaunchKernelGGL(k_fused<VARIANT>, dim3(WG_PER_PAIR, PAIRS), dim3(NT), 70 * 1024, 0, P);
        CHK(hipDeviceSynchronize());
        epoch += ITERS;
        CHK(hipMemcpy(h.data(), P.stamp, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        for (int i = 0; i < nwg; i++) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); }
        if (r >= 3) us.push_back((double)(hi - lo) * 0.01);
    }
    if (check && VARIANT != 0) {           // every workgroup of a pair must hold the same sums
        std::vector<double> o((size_t)nwg * NREC);
        CHK(hipMemcpy(o.data(), P.out, sizeof(double) * o.size(), hipMemcpyDeviceToHost));
        for (int p = 0; p < PAIRS; p++)
            for (int w = 1; w < WG_PER_PAIR; w++)
                for (int e = 0; e < NREC; e++)
                    if (o[((size_t)p * WG_PER_PAIR + w) * NREC + e] != o[(size_t)p * WG_PER_PAIR * NREC + e]) { fprintf(stderr, "variant %d: workgroups disagree\n", VARIANT); exit(2); }
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main() {
    Params P{};
    const int nwg = PAIRS * WG_PER_PAIR;
    CHK(hipMalloc(&P.rec, sizeof(float) * 2 * nwg * NREC));
    CHK(hipMalloc(&P.counter, sizeof(unsigned) * PAIRS));
    CHK(hipMalloc(&P.flag, sizeof(unsigned) * nwg));
    CHK(hipMalloc(&P.out, sizeof(double) * nwg * NREC));
    CHK(hipMalloc(&P.stamp, sizeof(unsigned long long) * 2 * nwg));
    CHK(hipMalloc(&P.err, sizeof(int)));
    CHK(hipMemset(P.rec, 0, sizeof(float) * 2 * nwg * NREC));
    CHK(hipMemset(P.flag, 0, sizeof(unsigned) * nwg));
    CHK(hipMemset(P.err, 0, sizeof(int)));
    CHK(hipFuncSetAttribute((const void *)k_fused<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    CHK(hipFuncSetAttribute((const void *)k_fused<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    CHK(hipFuncSetAttribute((const void *)k_fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    CHK(hipFuncSetAttribute((const void *)k_fused<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    int per_cu = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_fused<1>, NT, 70 * 1024));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    if (per_cu * prop.multiProcessorCount < nwg) { fprintf(stderr, "grid would not be co-resident (%d x %d < %d)\n", per_cu, prop.multiProcessorCount, nwg); return 3; }
    std::vector<unsigned long long> h(2 * nwg);
    const int reps = 41;
    for (int spin : {0, 400, 2000}) {   // 0: the primitive alone; 400 dependent fma ~ 8-9 us of work per iteration (the real kernel's scale); 2000 ~ 42 us
        P.spin_iters = spin;
        double t[4];
        unsigned epoch = 0;
        CHK(hipMemset(P.counter, 0, sizeof(unsigned) * PAIRS));
        CHK(hipMemset(P.flag, 0, sizeof(unsigned) * nwg));
        t[0] = run<0>(P, reps, h, epoch, false);
        epoch = 0; CHK(hipMemset(P.counter, 0, sizeof(unsigned) * PAIRS));
        t[1] = run<1>(P, reps, h, epoch, true);
        epoch = 0; CHK(hipMemset(P.flag, 0, sizeof(unsigned) * nwg));
        t[2] = run<2>(P, reps, h, epoch, true);
        epoch = 0; CHK(hipMemset(P.flag, 0, sizeof(unsigned) * nwg));
        t[3] = run<3>(P, reps, h, epoch, true);
        int err = 0;
        CHK(hipMemcpy(&err, P.err, sizeof(int), hipMemcpyDeviceToHost));
        printf("{\"workgroups\": %d, \"threads\": %d, \"iterations\": %d, \"stand_in_fma_chain\": %d, \"us_per_launch\": {\"no_barrier\": %.2f, \"counter\": %.2f, \"flags\": %.2f, \"flags_nosleep\": %.2f}, "
               "\"us_per_iteration\": {\"no_barrier\": %.2f, \"counter\": %.2f, \"flags\": %.2f, \"flags_nosleep\": %.2f}, "
               "\"barrier_us_per_iteration\": {\"counter\": %.2f, \"flags\": %.2f, \"flags_nosleep\": %.2f}, \"timeouts\": %d}\n",
               nwg, NT, ITERS, spin, t[0], t[1], t[2], t[3], t[0] / ITERS, t[1] / ITERS, t[2] / ITERS, t[3] / ITERS, (t[1] - t[0]) / ITERS, (t[2] - t[0]) / ITERS,
               (t[3] - t[0]) / ITERS, err);
    }
    return 0;
}
