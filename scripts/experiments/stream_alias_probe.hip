// stream_alias_probe.hip -- which HIP streams of a process really run concurrently?
// ROCclr maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, default 4, per priority level); two streams that land
// in the same queue serialise.  For every pair (i, j) of K streams: a 300 us spin kernel on i, then a tiny kernel on j; if j's kernel
// finishes before i's, the two streams are concurrent.  Prints the K x K matrix ('.' = concurrent, 'X' = serialised).
//   hipcc -O3 --offload-arch=gfx950 scripts/experiments/stream_alias_probe.hip -o /tmp/probe && /tmp/probe [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(long long ticks, long long *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    if (out) *out = wall_clock64();
}
__global__ void k_stamp(long long *out) { *out = wall_clock64(); }
int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<hipStream_t> s(K);
    int lo, hi;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < K; i++) {
        if (i == K - 1) CHK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, hi));   // the last one: high priority
        else CHK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    }
    long long *d;
    CHK(hipMalloc(&d, 2 * sizeof(long long)));
    printf("%d streams (the last one high priority); row i = spinning stream, column j = probing stream\n", K);
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < K; j++) {
            if (i == j) { printf("  -"); continue; }
            long long h[2];
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], 30000LL, d);        // 300 us
            hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s[j], d + 1);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("  %c", h[1] < h[0] ? '.' : 'X');
        }
        printf("\n");
    }
    return 0;
}
