// sela_device.h -- wave-level building blocks shared by the encode and decode kernels (gfx950).
//
// Everything here assumes wavefront = 64 lanes and that the translation unit is compiled with
// -ffp-contract=off (the FP64 contract of SURVEY.md App. A: separate v_mul_f64 / v_add_f64 in
// the reference's accumulation order; IEEE div/sqrt).
#ifndef SELA_DEVICE_H_
#define SELA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SELA_TABLE_QUAL static __device__ const
#include "sela_format.h"
#include "sela_hip.h"

namespace sela {

constexpr int kWave = 64;
constexpr int kBlock = SELA_BLOCK;          // 2048 samples per channel per frame
constexpr int kPerLane = kBlock / kWave;    // 32
constexpr int kMaxOrder = SELA_MAX_LPC_ORDER;
constexpr int kCoefWordsCap = 32;           // 100 values * (1 + 6 + 1) bits = 25 words worst case
constexpr int kResWordsCap = 2208;          // > 34 bits per sample; 16-bit audio needs <= ~1216
constexpr int kSlotWords = kCoefWordsCap + kResWordsCap; // 2240 words per (frame, signal) slot

// One record per (frame, signal) written by the block-encode kernel.
struct BlockMeta {
    uint8_t order;
    uint8_t coef_k;
    uint8_t res_k;
    uint8_t flags;
    uint16_t coef_words;
    uint16_t res_words;
};
static_assert(sizeof(BlockMeta) == 8, "BlockMeta is read as one 8-byte word");

// Order LDS traffic between the lanes of ONE wave (a wave executes in lockstep and the LDS serves a
// wave's instructions in order, so this only has to stop the compiler from moving accesses and make
// it wait for outstanding ones).  Used instead of __syncthreads() because in the decoder several
// waves of a workgroup run this code with different trip counts.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- cross-lane moves (DPP on the GFX9 family: the shift crosses all four rows) ----------------
// lane l <- lane l+1 ; lane 63 receives `fill`.
__device__ __forceinline__ int wave_shl1(int fill, int v)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
// lane l <- lane l+1 ; lane 63 receives 0 (bound_ctrl: no register has to be pre-set)
__device__ __forceinline__ uint32_t wave_shl1_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint64_t wave_shl1_zero(uint64_t x)
{
    return ((uint64_t)wave_shl1_zero((uint32_t)(x >> 32)) << 32) | wave_shl1_zero((uint32_t)x);
}
__device__ __forceinline__ double wave_shl1_zero(double v)
{
    return __builtin_bit_cast(double, wave_shl1_zero(__builtin_bit_cast(uint64_t, v)));
}
__device__ __forceinline__ double wave_shl1(double fill, double v)
{
    const uint64_t f = __builtin_bit_cast(uint64_t, fill), x = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)wave_shl1((int)(uint32_t)f, (int)(uint32_t)x);
    const uint32_t hi = (uint32_t)wave_shl1((int)(uint32_t)(f >> 32), (int)(uint32_t)(x >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Every lane receives the value of lane I of its own row of 16 lanes (one v_mov_b64_dpp row_newbcast).
template <int I>
__device__ __forceinline__ double row_broadcast(double v)
{
    const long long x = __builtin_bit_cast(long long, v);
    const long long y = __builtin_amdgcn_mov_dpp(x, 0x150 + I /* row_newbcast:I */, 0xf, 0xf, true);
    return __builtin_bit_cast(double, y);
}

__device__ __forceinline__ double read_first_lane(double v)
{
    const uint64_t x = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint64_t read_first_lane(uint64_t x)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// ---- wave reductions / scans (butterfly over ds_bpermute; exact integer arithmetic) ----------------
__device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
// Exclusive prefix sum over the 64 lanes.
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, int lane)
{
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    return incl - v;
}

// ---- format arithmetic --------------------------------------------------------------------------
// zig-zag of src/rice/rice_encoder.cpp:15, evaluated in int32 then widened with sign extension
// (values that do not fit 32 bits raise SELA_HIP_FLAG_RICE_RANGE at the call site).
__device__ __forceinline__ uint32_t zigzag32(int32_t x)
{
    const uint32_t t = (uint32_t)x << 1;
    return x < 0 ? (0u - t - 1u) : t;
}
// requiredInts = ceil((float)requiredBits / 32), src/rice/rice_encoder.cpp:37,63
__device__ __forceinline__ uint32_t words_for_bits(uint64_t bits)
{
    return (uint32_t)ceilf((float)bits / 32.0f);
}

// Dequantise one reflection coefficient (src/lpc/linear_predictor.cpp:23-26). idx = q + 64 clamped.
__device__ __forceinline__ double dequant(int i, int32_t q, uint32_t& flags)
{
    int idx = q + 64;
    if (idx < 0 || idx > 127) {
        flags |= SELA_HIP_FLAG_Q_RANGE;
        idx = idx < 0 ? 0 : 127;
    }
    return i == 0 ? SELA_DEQUANT_FIRST[idx] : (i == 1 ? SELA_DEQUANT_SECOND[idx] : SELA_DEQUANT_HIGHER[idx]);
}

// (int64)(2^35 * v) with the x86 result for out-of-range input (src/lpc/linear_predictor.cpp:59).
__device__ __forceinline__ int64_t q35_trunc(double v, uint32_t& flags)
{
    const double scaled = 34359738368.0 * v;
    if (!(scaled > -9223372036854775808.0 && scaled < 9223372036854775808.0)) {
        flags |= SELA_HIP_FLAG_COEF_OVERFLOW;
        return INT64_MIN;
    }
    return (int64_t)scaled;
}

// ---- step-up recursion, shared by encoder and decoder -------------------------------------------
// Reflection coefficients kd[0..order) (LDS) -> Q35 predictor a[0..order] (LDS, int64).
// src/lpc/linear_predictor.cpp:30-61.  Stage i updates every m < i from the OLD values,
//   t[m] <- t[m] + kd[i] * t[i-1-m]
// which is the reference's pairwise update (and its odd-i middle element) written per element;
// the reads of a stage all precede its writes, so the lanes are independent.  One wave.
__device__ inline void step_up(const double* kd, double* t, int64_t* a, int order, int lane, uint32_t& flags)
{
    for (int i = 0; i < order; i++) {
        const double ki = kd[i];
        const int m0 = lane, m1 = lane + 64;
        double n0 = 0.0, n1 = 0.0;
        if (m0 < i)
            n0 = t[m0] + ki * t[i - 1 - m0];
        if (m1 < i)
            n1 = t[m1] + ki * t[i - 1 - m1];
        wave_sync();
        if (m0 < i)
            t[m0] = n0;
        if (m1 < i)
            t[m1] = n1;
        if (lane == 0)
            t[i] = ki;
        wave_sync();
    }
    if (lane == 0)
        a[0] = 0;
    for (int m = lane; m < order; m += 64)
        a[m + 1] = q35_trunc(-t[m], flags);
    wave_sync();
}

} // namespace sela
#endif // SELA_DEVICE_H_
