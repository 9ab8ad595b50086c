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
#include "sela_hip_debug.h"

namespace sela {

constexpr int kWave = 64;
constexpr int kBlock = SELA_BLOCK;          // 2048 samples per channel per frame
constexpr int kPerLane = kBlock / kWave;    // 32
constexpr int kMaxOrder = SELA_MAX_LPC_ORDER;
constexpr int kCoefWordsCap = 32;           // 100 values * (1 + 6 + 1) bits = 25 words worst case
constexpr int kResWordsCap = 2208;          // > 34 bits per sample; 16-bit audio needs <= ~1216
constexpr int kSlotWords = kCoefWordsCap + kResWordsCap; // 2240 words per (frame, signal) slot
constexpr int kGroupFrames = 8;             // frames whose bytes one block (the last of theirs to finish) places and writes

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
// In BlockMeta::flags only (never in d_status: frame_words masks them): which form of the residue filter the block took
// (sela_encode_tail.inc) -- neither bit: one pass of FP64 taps.  They reuse the bits of two flags an encoder block never
// raises (RICE_OVERRUN is the decoder's, SHORT_BLOCK the any-length route's).  Read by sela_hip_debug_block_forms (tests).
constexpr uint32_t kBlockFormTwoPass = SELA_HIP_FLAG_RICE_OVERRUN, kBlockFormPlain = SELA_HIP_FLAG_SHORT_BLOCK;
constexpr uint32_t kBlockFormBits = kBlockFormTwoPass | kBlockFormPlain;

// What the host pipeline hands an encode launch (launch_encode): where the launch reports to the host, where the
// job's stream position lives, and -- when the PCM is still in page-locked host memory -- where to fetch it from.
struct EncodeHostLink {
    uint64_t* mirror;        // page-locked, as the device sees it: a copy of frame_offsets[0..n_frames] + one status word
    const uint64_t* pos_in;  // device word: offset of the launch's first frame in the job's stream
    uint64_t* pos_out;       // device word (another one): the launch leaves the offset behind its last frame here
    const int16_t* host_pcm; // page-locked stereo PCM as the device sees it (k_stage_in copies it in), or null: the PCM is on the device
    uint64_t* pcm_ready;     // device: [n_frames] <- launch ticket | checksum, frame copied in
    uint32_t stage_workgroups; // of k_stage_in, each with a CU to itself
    hipStream_t stage_stream;  // where k_stage_in runs (not the encode launch's stream: the two may run side by side)
    uint64_t* stage_started;   // four device words: launch ticket | stager workgroups that have a CU; the stagers' frame counter; the gate's mark; groups the launch has finished
    int wait_naps;             // bound of a block's wait for its frame, in naps of 2048 cycles; -1: the default (~0.5 s).  Tests set 0.
};

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

// s_setprio with a run-time level (the instruction takes an immediate)
__device__ __forceinline__ void set_wave_priority(int level)
{
    switch (level & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
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

// ---- wave reductions / scans (exact integer arithmetic) ----------------------------------------------
// Sum of one 32-bit value per lane (the total must fit 32 bits) by DPP adds -- row_shr 1/2/4/8 (a prefix
// sum within each row of 16), then row_bcast 15/31 under row masks: six full-rate instructions instead of six
// ds_bpermute round trips; the total lands in lane 63.
__device__ __forceinline__ uint32_t wave_sum_small(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    // lane 15 of every row now holds its row's total
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// OR of one 32-bit value per lane (same ladder; the result is wave-uniform).
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Maximum of one unsigned 32-bit value per lane (same ladder; the result is wave-uniform).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Exact wave sum of per-lane values below 2^40 (sums of 32 Rice quotients are below 2^37): two 20-bit
// limbs, each summed without overflow.
__device__ __forceinline__ uint64_t wave_sum_40(uint64_t v)
{
    const uint32_t lo = wave_sum_small((uint32_t)v & 0xFFFFFu);
    const uint32_t hi = wave_sum_small((uint32_t)(v >> 20));
    return ((uint64_t)hi << 20) + lo;
}

// Exclusive prefix sum over the 64 lanes (totals must fit 32 bits): the same DPP ladder -- prefix sums
// within each row of 16, then the totals of the rows before are broadcast in.
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, int lane)
{
    (void)lane;
    uint32_t incl = v;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
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
// src/lpc/linear_predictor.cpp:30-61.  Stage i turns every t[m], m < i, into t[m] + k_i * t[i-1-m]
// (the reference's pairwise update and its odd-i middle element, written per element from the OLD
// values) and appends t[i] = k_i.  The whole recursion stays in registers, on the MONIC polynomial
// A = [1, t[0], t[1], ...]: lane m (+64 in a second register) holds A[m] and, beside it, the mirrored
// element R[m] = A[i+1-m] it is about to meet.  With the leading 1 in the arrays every element obeys ONE rule,
//     A'[m] = A[m] + k_i * R[m]        (m = i+1:  0 + k_i * 1 = k_i, the appended coefficient;  m = 0:  1 + k_i * 0)
//     R'[m] = R[m-1] + k_i * A[m-1]    (the same multiply-add with the roles swapped, moved up one lane; R'[0] = 0)
// so a stage is one scalar read of k_i, four FP64 operations and one lane shift -- no lane is selected or masked
// (round 2 kept t[] without its leading 1 and selected k_i into two lanes per stage: 15 instructions instead of 8).
// Every A[m] sees the reference's operations on the reference's operands (t + k * t', the product rounded
// first); 0 + k * 1 is k for every finite k (a -0 becomes +0: it can only ever meet other zeros, and
// (int64)(2^35 * -t) is 0 for both).  Elements beyond the current degree are +0.0 (0 + k*0 = +0).
__device__ __forceinline__ double read_lane(double v, int src_lane /* wave-uniform */)
{
    const uint64_t x = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src_lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// lane l <- lane l-1 ; lane 0 receives 0
__device__ __forceinline__ double wave_shr1_zero(double v)
{
    const uint64_t x = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// lane l <- lane l-1 ; lane 0 receives `fill`
__device__ __forceinline__ double wave_shr1(double fill, double v)
{
    const uint64_t f = __builtin_bit_cast(uint64_t, fill), x = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)f, (int)(uint32_t)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(f >> 32), (int)(uint32_t)(x >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// k_lo / k_hi: reflection coefficient `lane` / `lane + 64` (0 beyond the order)
__device__ inline void step_up_regs(double k_lo, double k_hi, int64_t* a, int order, int lane, uint32_t& flags)
{
    double t_lo = lane == 0 ? 1.0 : 0.0, r_lo = lane == 1 ? 1.0 : 0.0; // degree 0: A = [1], R[m] = A[1 - m]
    double t_hi = 0.0, r_hi = 0.0;
    // while the mirror of the new degree (i + 2 elements) fits lanes 0..63, the second register is all zeros
    const int n_lo = order < 62 ? order : 62;
    for (int i = 0; i < n_lo; i++) {
        const double ki = read_lane(k_lo, i);
        const double tn = t_lo + ki * r_lo;
        const double rn = r_lo + ki * t_lo;
        t_lo = tn;
        r_lo = wave_shr1_zero(rn);
    }
    for (int i = 62; i < order; i++) {
        const double ki = i < 64 ? read_lane(k_lo, i) : read_lane(k_hi, i - 64);
        const double tn_lo = t_lo + ki * r_lo, rn_lo = r_lo + ki * t_lo;
        const double tn_hi = t_hi + ki * r_hi, rn_hi = r_hi + ki * t_hi;
        t_lo = tn_lo;
        t_hi = tn_hi;
        const double carry = read_lane(rn_lo, 63); // crosses from the first register into the second
        r_lo = wave_shr1_zero(rn_lo);
        r_hi = wave_shr1(carry, rn_hi);
    }
    if (lane == 0)
        a[0] = 0;
    else if (lane <= order)
        a[lane] = q35_trunc(-t_lo, flags);
    if (lane + 64 <= order)
        a[lane + 64] = q35_trunc(-t_hi, flags);
    wave_sync();
}

__device__ inline void step_up(const double* kd, int64_t* a, int order, int lane, uint32_t& flags)
{
    const double k_lo = lane < order ? kd[lane] : 0.0;
    const double k_hi = lane + 64 < order ? kd[lane + 64] : 0.0;
    step_up_regs(k_lo, k_hi, a, order, lane, flags);
}

} // namespace sela
#endif // SELA_DEVICE_H_
