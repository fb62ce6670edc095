// wave_reduce.h -- value-halving butterfly: reduce NV per-lane values over the 64 lanes of a wave (gfx950).
//
// A plain wave reduction costs 6 cross-lane steps PER VALUE.  Here a lane keeps one half of its values at each step and
// hands the other half to its partner, so the work halves every step (32 -> 16 -> 8 -> 4 -> 2 -> 1): ~75 VALU for 32
// values instead of ~240.  The two cross-row steps use CDNA4's v_permlane32_swap / v_permlane16_swap (one instruction
// exchanges a register pair between partner lanes), the in-row steps use DPP.  Every lane ends with ONE fully reduced
// value whose index is slotNN(lane).  Verified against a host sum on MI355X (tests/test_gpu_parity.py covers it through
// the normal equations).
#pragma once
#include <hip/hip_runtime.h>

namespace tc {

// partner exchange inside a 16-lane row via DPP.  XOR = 1,2 use quad_perm, 4 and 8 use two bank-masked row shifts.
template <int XOR>
__device__ __forceinline__ float dpp_xor(float x) {
    const int xi = __float_as_int(x);
    int r;
    if (XOR == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, false);        // quad_perm:[1,0,3,2]
    else if (XOR == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
    else if (XOR == 4) {
        r = __builtin_amdgcn_update_dpp(0, xi, 0x104, 0xF, 0x5, false);                 // row_shl:4 -> lanes 0-3, 8-11 read lane+4
        r = __builtin_amdgcn_update_dpp(r, xi, 0x114, 0xF, 0xA, false);                 // row_shr:4 -> lanes 4-7, 12-15 read lane-4
    } else {
        r = __builtin_amdgcn_update_dpp(0, xi, 0x108, 0xF, 0x3, false);                 // row_shl:8 -> lanes 0-7 read lane+8
        r = __builtin_amdgcn_update_dpp(r, xi, 0x118, 0xF, 0xC, false);                 // row_shr:8 -> lanes 8-15 read lane-8
    }
    return __int_as_float(r);
}

// One halving step over the lane bit `XOR` (1,2,4,8 via DPP; 16,32 via the gfx950 permlane swaps).
// In: v[0..N).  Out: v[0..N/2) = own kept half + partner's same half.  Lanes with the bit clear keep the lower half.
template <int N, int XOR>
__device__ __forceinline__ void halve(float *v, int lane) {
    if (XOR >= 16) {
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + N / 2]);
            auto r = (XOR == 32) ? __builtin_amdgcn_permlane32_swap(a, b, false, false)
                                 : __builtin_amdgcn_permlane16_swap(a, b, false, false);
            v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    } else {
        // lanes with the bit set keep the upper half.  The lane pattern is a compile-time constant, so the selects take it as a
        // 64-bit scalar mask (v_cndmask_b32_e64 with an SGPR pair: half rate) instead of a compare into VCC + v_cndmask_b32_e32
        // (measured 8.2 clocks per select behind a compare, scripts/valu_rate.hip).
        constexpr unsigned long long HI = XOR == 1 ? 0xAAAAAAAAAAAAAAAAull : XOR == 2 ? 0xCCCCCCCCCCCCCCCCull
                                        : XOR == 4 ? 0xF0F0F0F0F0F0F0F0ull : 0xFF00FF00FF00FF00ull;
        (void)lane;
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            float keep, send;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(keep) : "v"(v[i]), "v"(v[i + N / 2]), "s"(HI));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(send) : "v"(v[i + N / 2]), "v"(v[i]), "s"(HI));
            v[i] = keep + dpp_xor<XOR>(send);
        }
    }
}

// Reduce 32 values per lane over the wave.  On return every lane holds ONE fully reduced value in v[0]; its index is
// slot32(lane).  Steps: xor32, xor16 (one swap instruction per value pair), xor1, xor2, xor4 (halving), xor8 (plain).
__device__ __forceinline__ int slot32(int lane) {
    return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + (lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1);
}
__device__ __forceinline__ void wave_reduce32(float *v, int lane) {
    halve<32, 32>(v, lane);
    halve<16, 16>(v, lane);
    halve<8, 1>(v, lane);
    halve<4, 2>(v, lane);
    halve<2, 4>(v, lane);
    v[0] += dpp_xor<8>(v[0]);
}
// 64 values: one more halving level (xor 8 halves instead of plain-reducing); index slot64(lane)
__device__ __forceinline__ int slot64(int lane) {
    return ((lane >> 5) & 1) * 32 + ((lane >> 4) & 1) * 16 + (lane & 1) * 8 + ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 +
           ((lane >> 3) & 1);
}
__device__ __forceinline__ void wave_reduce64(float *v, int lane) {
    halve<64, 32>(v, lane);
    halve<32, 16>(v, lane);
    halve<16, 1>(v, lane);
    halve<8, 2>(v, lane);
    halve<4, 4>(v, lane);
    halve<2, 8>(v, lane);
}
// 8 values: index slot8(lane) = bits (5,4,0); the xor 2, 4, 8 steps are plain reductions.  Every width combines the lanes in the same
// order (32, 16, 1, 2, 4, 8), so a value's sum does not depend on which chunk carries it.
__device__ __forceinline__ int slot8(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + (lane & 1); }
__device__ __forceinline__ void wave_reduce8(float *v, int lane) {
    halve<8, 32>(v, lane);
    halve<4, 16>(v, lane);
    halve<2, 1>(v, lane);
    v[0] += dpp_xor<2>(v[0]);
    v[0] += dpp_xor<4>(v[0]);
    v[0] += dpp_xor<8>(v[0]);
}
// 4 values (cost-only mode): index slot4(lane) = bits (5,4); the in-row part is a plain reduction
__device__ __forceinline__ int slot4(int lane) { return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1); }
__device__ __forceinline__ void wave_reduce4(float *v, int lane) {
    halve<4, 32>(v, lane);
    halve<2, 16>(v, lane);
    v[0] += dpp_xor<1>(v[0]);
    v[0] += dpp_xor<2>(v[0]);
    v[0] += dpp_xor<4>(v[0]);
    v[0] += dpp_xor<8>(v[0]);
}

// Reduce NLIVE (compile-time) values in chunks of 64 / 32 / 4 and store the reduced values to dst[0..NLIVE).
template <int NLIVE, int BASE = 0>
__device__ __forceinline__ void wave_reduce_store(const float *vals, float *dst, int lane) {
    constexpr int REM = NLIVE - BASE;
    if constexpr (REM > 32 && REM <= 40) {          // 33..40 values (pose + depth scale: 38): a 32-chunk and an 8-chunk, not a 64-chunk
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = vals[BASE + i];
        wave_reduce32(v, lane);
        const int s = slot32(lane);
        if (!(lane & 8)) dst[BASE + s] = v[0];
        wave_reduce_store<NLIVE, BASE + 32>(vals, dst, lane);
    } else if constexpr (REM > 32) {
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; i++) v[i] = (i < REM) ? vals[BASE + i] : 0.f;
        wave_reduce64(v, lane);
        const int s = slot64(lane);
        if (s < REM) dst[BASE + s] = v[0];
        wave_reduce_store<NLIVE, BASE + 64>(vals, dst, lane);
    } else if constexpr (REM > 4 && REM <= 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (i < REM) ? vals[BASE + i] : 0.f;
        wave_reduce8(v, lane);
        const int s = slot8(lane);
        if (s < REM && !(lane & 14)) dst[BASE + s] = v[0];
    } else if constexpr (REM > 4) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = (i < REM) ? vals[BASE + i] : 0.f;
        wave_reduce32(v, lane);
        const int s = slot32(lane);
        if (s < REM && !(lane & 8)) dst[BASE + s] = v[0];
    } else if constexpr (REM > 0) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (i < REM) ? vals[BASE + i] : 0.f;
        wave_reduce4(v, lane);
        const int s = slot4(lane);
        if (s < REM && !(lane & 15)) dst[BASE + s] = v[0];
    }
}

}  // namespace tc
