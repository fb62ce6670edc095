// valu_rate.hip -- what does one SIMD of gfx950 sustain in wave64 VALU instructions per clock?
//
// k_linearize is VALU-bound (PMC: SQ_ACTIVE_INST_VALU ~ 4.1 quad-clocks... per instruction); to price its instruction stream
// against a PEAK the peak itself has to be measured, per instruction class and per occupancy:
//   classes : v_fma_f32 | v_pk_fma_f32 | v_pk_mul_f32 | v_pk_add_f32 | v_rcp_f32 | v_mov_b32 DPP (row_shr) | v_cndmask_b32 |
//             v_cvt/v_floor mix is not needed: everything else in the kernel is plain 32-bit VALU = the v_fma_f32 class
//   waves per SIMD : 1, 2, 4 (256-thread workgroups, occupancy limited through dynamic LDS)
// Every lane runs NI independent chains (no dependent-issue stalls), REP x UNROLL instructions per chain.
// Output: one JSON line per (class, waves/SIMD): clocks per wave-instruction per SIMD at the measured shader clock.
//
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scripts/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NI = 8;        // independent chains per lane
constexpr int UNROLL = 32;   // instructions per chain per loop trip

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CLS>
__global__ __launch_bounds__(256) void k_rate(float *out, int reps, long long *clk) {
    extern __shared__ float dyn[];
    float a[NI], b[NI];
    f2 p[NI], p2[NI];
    unsigned long long q[NI], sm[2] = {0, 0};
    const unsigned long long smask = (unsigned long long)reps * 0x5555555555ULL;   // wave-uniform, not a compile-time constant
    const float sk = (float)reps * 1e-9f;
    const float s = 1.0000001f, t = 1e-9f;
    const f2 s2 = {s, s}, t2 = {t, t};
#pragma unroll
    for (int i = 0; i < NI; i++) { a[i] = (float)(threadIdx.x + i) * 1e-3f; b[i] = a[i] + 2.f; p[i] = (f2){a[i], a[i] + 1.f}; p2[i] = p[i] + 1.f; q[i] = threadIdx.x + i; }
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b[1]) : "vcc");     // a defined VCC for the classes that read it
    long long c0 = 0, r0 = 0;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (CLS == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
                if (CLS == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(s2), "v"(t2));
                if (CLS == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
                if (CLS == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t2));
                if (CLS == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (CLS == 5) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (CLS == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : );
                if (CLS == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                if (CLS == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                if (CLS == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "s"(smask));
                if (CLS == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));          // move out of a second register set
                if (CLS == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 13) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) % NI]));
                if (CLS == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 15) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
                if (CLS == 16) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
                if (CLS == 17) asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
                if (CLS == 18) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
                if (CLS == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 20) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if (CLS == 21) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
                if (CLS == 22) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(s) : "vcc");
                if (CLS == 23) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(sm[i & 1]) : "v"(a[i]), "v"(s));
                if (CLS == 24) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (CLS == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
                if (CLS == 26) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(p[i]) : "v"(p2[i]));
                if (CLS == 27) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                if (CLS == 28) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(t), "v"(s));
                if (CLS == 29) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 30) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                if (CLS == 31) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(sk));           // SGPR operand
                if (CLS == 32) asm volatile("v_mul_f32 %0, 0x3e4ccccd, %0" : "+v"(a[i]));                  // literal operand
                if (CLS == 33) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");   // compare + select pair
                if (CLS == 34) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (CLS == 35) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sk));
            }
        }
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) acc += a[i] + b[i] + p[i].x + p[i].y + p2[i].x + (float)q[i] + (float)sm[i & 1];
    if (acc == 12345.678f) out[0] = acc + dyn[0];   // never true: keeps the chains alive
}

template <int CLS>
void run(const char *name, float *out, long long *clk) {
    for (int wps : {1, 2, 4}) {
        const int lds = (wps == 1 ? 96 : (wps == 2 ? 64 : 36)) * 1024;   // 1 / 2 / 4 workgroups of 4 waves per CU
        CHK(hipFuncSetAttribute((const void *)k_rate<CLS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int blocks = 256 * wps * 4, reps = 400;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<CLS>, dim3(blocks), dim3(256), lds, 0, out, reps, clk);   // warm-up (+ clock ramp)
        hipLaunchKernelGGL(k_rate<CLS>, dim3(blocks), dim3(256), lds, 0, out, reps, clk);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<CLS>, dim3(blocks), dim3(256), lds, 0, out, reps, clk);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2];
        CHK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / (double)h[1] * 0.1;                // s_memrealtime ticks at 100 MHz
        const double winsts = (double)blocks * 4 /*waves*/ * reps * UNROLL * NI;
        const double simd_clk = ms * 1e-3 * ghz * 1e9 * 1024.0;              // clocks summed over the 1024 SIMDs
        printf("{\"class\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"shader_GHz\": %.3f, \"clk_per_wave_inst_per_simd\": %.3f, "
               "\"G_wave_insts_per_s\": %.1f}\n", name, wps, ms, ghz, simd_clk / winsts, winsts / (ms * 1e-3) / 1e9);
        fflush(stdout);
    }
}

int main() {
    float *out; long long *clk;
    CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&clk, 16));
    run<0>("v_fma_f32", out, clk);
    run<1>("v_pk_fma_f32", out, clk);
    run<2>("v_pk_mul_f32", out, clk);
    run<3>("v_pk_add_f32", out, clk);
    run<4>("v_rcp_f32", out, clk);
    run<5>("v_mov_b32_dpp", out, clk);
    run<6>("v_cndmask_b32 vcc", out, clk);
    run<7>("v_add_f32", out, clk);
    run<8>("v_mul_f32", out, clk);
    run<9>("v_max_f32", out, clk);
    run<10>("v_cndmask_b32_e64 sgpr-pair", out, clk);
    run<11>("v_mov_b32", out, clk);
    run<12>("v_add_u32", out, clk);
    run<13>("v_lshl_add_u64", out, clk);
    run<14>("v_min_i32", out, clk);
    run<15>("v_cvt_f32_i32", out, clk);
    run<16>("v_fmac_f32", out, clk);
    run<17>("s_nop+v_permlane32_swap", out, clk);
    run<18>("v_ashrrev_i32", out, clk);
    run<19>("v_mul_lo_u32", out, clk);
    run<20>("v_floor_f32", out, clk);
    run<21>("v_cvt_i32_f32", out, clk);
    run<22>("v_cmp_gt_f32 vcc", out, clk);
    run<23>("v_cmp_gt_f32_e64 sgpr-pair", out, clk);
    run<24>("v_add_f32_dpp", out, clk);
    run<25>("v_mad_u32_u24", out, clk);
    run<26>("v_pk_mov_b32", out, clk);
    run<27>("v_min_f32", out, clk);
    run<28>("v_med3_f32", out, clk);
    run<29>("v_and_b32", out, clk);
    run<30>("v_sub_f32", out, clk);
    run<31>("v_fma_f32 sgpr-operand", out, clk);
    run<32>("v_mul_f32 literal", out, clk);
    run<33>("v_cmp+v_cndmask pair (2 insts)", out, clk);
    run<34>("v_max_i32", out, clk);
    run<35>("v_mov_b32 from sgpr", out, clk);
    return 0;
}
