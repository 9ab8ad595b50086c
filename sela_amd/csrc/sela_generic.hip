// sela_generic.hip -- the frame path for ANY block length and 32-bit samples (gfx950).
//
// The reference's frame API is length-agnostic and 32-bit: data::WavFrame carries int32 samples per channel
// (src/include/data/wav_frame.hpp:8-16), lpc::ResidueGenerator loops over samples.size() (src/lpc/residue_generator.cpp:12-45,
// 98-119), a subframe brings its own samplesPerChannel (src/include/data/sela_sub_frame.hpp:27,
// src/frame/frame_decoder.cpp:24-25,48-49) and frame::FrameDecoder returns untruncated int32 (frame_decoder.cpp:64-71).
// The fast kernels (sela_encode.hip, sela_decode.hip) are built around the one shape the reference's CLI produces --
// 2048 samples of 16-bit PCM -- and everything else comes here: the same arithmetic, bit for bit, with the length a run-time
// value (1 .. 65535: the u16 field) and nothing assumed about the samples.  One wave per block / subframe: the fast kernels'
// loops with a run-time length (round 6), global-memory scratch between the kernels; the route for callers of the frame classes
// with other shapes, not the one bench.py's headline times.
//
// Encode: k_generic_analyse (samples -> order, q[], residues, the two Rice plans) -> k_generic_plan (stereo decision, frame
// sizes, offsets) -> k_generic_pack (the chosen candidates' Rice streams) -> k_generic_assemble (on-disk bytes).
// Decode: k_decode_subframes32 (sela_decode32.hip: the fast decoder's lane-parallel parse and tuned synthesis, any length, 32-bit)
// or, for what that kernel will not judge, k_generic_decode (one wave per subframe: headers, a serial Rice walk, synthesis)
// -> k_generic_combine (independent subframes first, then dependent ones in subframe order: src/frame/frame_decoder.cpp:17-69).
#include <hip/hip_runtime.h>

#include <atomic>

#include "sela_device.h"
#include "sela_generic.h"

namespace sela {

namespace {

__device__ __forceinline__ uint64_t wave_sum_wrap(uint64_t v) // sum over the 64 lanes mod 2^64 (wave-uniform result)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// (int32)double as the x86 build of the reference converts (cvttsd2si: INT32_MIN for anything outside)
__device__ __forceinline__ int32_t trunc_to_i32(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT32_MIN;
    return (int32_t)v;
}

// OR the low `nbits` (1..32) of v into a zeroed word buffer (global memory or LDS) at bit position pos
__device__ __forceinline__ void or_bits(uint32_t* buf, uint64_t pos, uint32_t v, uint32_t nbits)
{
    const uint64_t w = pos >> 5;
    const uint32_t sh = (uint32_t)pos & 31;
    atomicOr(&buf[w], v << sh);
    if (sh + nbits > 32)
        atomicOr(&buf[w + 1], v >> (32 - sh));
}

// one Golomb-Rice codeword (src/rice/rice_encoder.cpp:41-53): u >> k ones, a zero, the low k bits MSB first
__device__ inline void put_codeword(uint32_t* buf, uint64_t pos, uint32_t u, uint32_t k)
{
    uint32_t ones = u >> k;
    const uint32_t rem = k ? __brev(u << (32 - k)) : 0u;
    while (ones >= 32) {
        or_bits(buf, pos, 0xFFFFFFFFu, 32);
        pos += 32;
        ones -= 32;
    }
    or_bits(buf, pos, (1u << ones) - 1u, ones + 1); // the ones and their terminator (a zero: nothing to OR)
    pos += ones + 1;
    if (k)
        or_bits(buf, pos, rem, k);
}

// pack the stream v[0..n) with parameter k into zeroed words (src/rice/rice_encoder.cpp:35-71).
// Round 6: 1024 values at a time, lane L owning 16 CONSECUTIVE values (staged through LDS so that the loads are coalesced and the
// lanes' reads fall on different banks) -- ONE scan of the lanes' bit counts per stretch, every codeword of at most 32 bits OR-ed
// into the LDS window without a branch -- and the window leaves as whole words: plain stores, but for the stretch's first and
// last word, which its neighbours share (an atomic OR each).  A stretch whose quotients are long (a lane's bits beyond 2^26) or
// whose bits do not fit the window goes 64 codewords at a time as the first version did for every round (rice_pack_round).
constexpr uint32_t kPackPerLane = 16;                         // consecutive values a lane owns in a stretch
constexpr uint32_t kPackStretch = 64 * kPackPerLane;         // 1024 values
constexpr uint32_t kPackWindow = 1152;  // words: 36,864 bits for 1024 codewords (36 bits each), and room for the 1024 staged values
                                        // (1088 words: one pad word per 16).  4.6 KB per wave: eight waves per SIMD (a stretch of 2048
                                        // values in a window of 9.2 KB, four waves per SIMD: 110 us for 7750 subframes of 2048 samples)

// 64 codewords at bit position `base` of the stream: lane = codeword
__device__ inline uint64_t rice_pack_round(uint32_t u, bool valid, uint32_t k, uint32_t* out, uint64_t base, int lane, uint32_t* win)
{
    const uint64_t len = valid ? (uint64_t)(u >> k) + 1 + k : 0;
    uint64_t incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d, 64);
        const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d, 64);
        if (lane >= d)
            incl += ((uint64_t)hi << 32) | lo;
    }
    const uint32_t tlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)incl, 63);
    const uint32_t thi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(incl >> 32), 63);
    const uint64_t total = ((uint64_t)thi << 32) | tlo; // the round's bits (wave-uniform)
    const uint64_t first_word = base >> 5, span = ((base + total + 31) >> 5) - first_word;
    if (span <= kPackWindow) {
        for (uint32_t w = lane; w < (uint32_t)span; w += 64)
            win[w] = 0;
        wave_sync();
        if (valid)
            put_codeword(win, (base & 31) + incl - len, u, k);
        wave_sync();
        for (uint32_t w = lane; w < (uint32_t)span; w += 64) {
            const uint32_t x = win[w];
            if (x == 0)
                continue; // (the words are zeroed)
            if (w == 0 || w + 1 == (uint32_t)span)
                atomicOr(&out[first_word + w], x);
            else
                out[first_word + w] = x;
        }
        wave_sync();
    } else if (valid) {
        put_codeword(out, base + incl - len, u, k); // piece by piece, straight into memory
    }
    return total;
}

// (a stretch's values as they come from memory: lane L holds values L, L + 64, ..; beyond n: 0)
__device__ __forceinline__ void pack_fetch(const int32_t* v, uint32_t n, uint32_t i0, int lane, int32_t (&raw)[kPackPerLane])
{
#pragma unroll
    for (int t = 0; t < (int)kPackPerLane; t++) {
        const uint32_t i = i0 + (uint32_t)lane + 64u * (uint32_t)t;
        raw[t] = i < n ? v[i] : 0;
    }
}

// kFetched: the first stretch's values are in `raw` already.  A wave of k_generic_pack is a chain of trips to memory -- two thousand
// vector instructions in 27 us, SQ_WAIT_ANY 70 % of its cycles -- and every trip taken early is one less: the kernel asks for the
// block's record, its coefficients and its first 1024 residues together, and a stretch is asked for before the one in front of it is
// packed: 80 -> 60 us for 7750 subframes of 2048 (more waves per SIMD do not help it: its 160 registers are live values; forced to
// 128 it spills 9 and takes 57 us, to 96 it spills 735 and takes 444).
template <bool kFetched>
__device__ inline void rice_pack_stream(const int32_t* v, uint32_t n, uint32_t k, uint32_t* out, int lane, uint32_t* win, int32_t (&raw)[kPackPerLane])
{
    uint64_t base = 0;
    if (!kFetched)
        pack_fetch(v, n, 0, lane, raw);
    for (uint32_t i0 = 0; i0 < n; i0 += kPackStretch) {
        const uint32_t in_stretch = min(kPackStretch, n - i0);
        // stage: value i of the stretch at word i + i / 16
#pragma unroll
        for (int t = 0; t < (int)kPackPerLane; t++) {
            const uint32_t i = (uint32_t)lane + 64u * (uint32_t)t;
            if (i < in_stretch)
                win[i + (i >> 4)] = zigzag32(raw[t]);
        }
        if (i0 + kPackStretch < n)
            pack_fetch(v, n, i0 + kPackStretch, lane, raw);
        wave_sync();
        uint32_t u[kPackPerLane];
        const uint32_t mine = in_stretch > kPackPerLane * (uint32_t)lane ? min(kPackPerLane, in_stretch - kPackPerLane * (uint32_t)lane) : 0u;
        uint32_t bits = 0, longest = 0;
#pragma unroll
        for (int t = 0; t < (int)kPackPerLane; t++) {
            u[t] = (uint32_t)t < mine ? win[(kPackPerLane + 1) * lane + t] : 0u;
            const uint32_t q = u[t] >> k;
            longest = max(longest, q);
            bits += (uint32_t)t < mine ? q + 1 + k : 0u; // (may wrap when a quotient is long: then `longest` says so)
        }
        wave_sync(); // the staged values are in registers: the window is free
        const bool easy = !__any(longest >= (1u << 20)); // a lane's bits below 2^25, the stretch's below 2^32
        uint32_t at = 0, total = 0;
        if (easy) {
            at = wave_exclusive_scan(bits, lane);
            total = (uint32_t)__builtin_amdgcn_readlane((int)(at + bits), 63);
        }
        const uint64_t first_word = base >> 5;
        const uint32_t span = easy ? (uint32_t)((((base & 31) + total + 31) >> 5)) : 0u;
        if (easy && span + 1 <= kPackWindow) {
            for (uint32_t w = lane; w <= span; w += 64) // (one word to spare: or_bits_both touches the word behind a codeword's last)
                win[w] = 0;
            wave_sync();
            uint32_t pos = (uint32_t)(base & 31) + at;
#pragma unroll
            for (int t = 0; t < (int)kPackPerLane; t++) {
                if ((uint32_t)t < mine) {
                    const uint32_t ones = u[t] >> k, len = ones + 1 + k;
                    if (len <= 32) {
                        const uint32_t rem = __builtin_amdgcn_ubfe(__brev(u[t]), 32 - k, k); // the low k bits, the first one the most significant
                        const uint64_t wide = (uint64_t)(((1u << ones) - 1u) | (rem << ((ones + 1) & 31))) << (pos & 31);
                        atomicOr(&win[pos >> 5], (uint32_t)wide);
                        atomicOr(&win[(pos >> 5) + 1], (uint32_t)(wide >> 32));
                    } else {
                        put_codeword(win, pos, u[t], k);
                    }
                    pos += len;
                }
            }
            wave_sync();
            for (uint32_t w = lane; w < span; w += 64) {
                const uint32_t x = win[w];
                if (x == 0)
                    continue; // (the words are zeroed)
                if (w == 0 || w + 1 == span)
                    atomicOr(&out[first_word + w], x);
                else
                    out[first_word + w] = x;
            }
            wave_sync();
            base += total;
        } else { // long quotients, or more bits than the window holds: 64 codewords at a time, codeword i of the stretch in lane i % 64
            // (rice_pack_round uses the window too, so round by round from the registers)
            for (uint32_t r0 = 0; r0 < in_stretch; r0 += 64) {
                // codeword r0 + lane lives in lane (r0 + lane) / 16's register (r0 + lane) % 16
                const uint32_t src_lane = (r0 + (uint32_t)lane) / kPackPerLane;
                uint32_t mine_u = 0;
#pragma unroll
                for (int t = 0; t < (int)kPackPerLane; t++) {
                    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * src_lane), (int)u[t]);
                    mine_u = (((r0 + (uint32_t)lane) % kPackPerLane) == (uint32_t)t) ? got : mine_u;
                }
                base += rice_pack_round(mine_u, r0 + lane < in_stretch, k, out, base, lane, win);
            }
        }
    }
}

} // namespace

// ---- analysis: one wave per (frame, signal) -------------------------------------------------------------------------------
// lpc::ResidueGenerator::process (src/lpc/residue_generator.cpp:121-134) as written, with samples.size() = n:
//   x[j] = s[j] / 32767 (:12-18);  mean = (sequential sum) / n (:27-30);  ac[lag] = sequential sum over j = lag .. n-1 of
//   (x[j] - mean) * (x[j - lag] - mean), lag = 0..100 (:33-38);  Schur (:47-68);  order (:70-78);  quantise (:80-96);
//   dequantise + step-up (linear_predictor.cpp:16-61);  residues (:98-119);  then both Rice plans (rice_encoder.cpp:20-33, 37).
//
// Round 6: the loops of the fast kernels (sela_encode.hip) with the length a run-time value -- the first version walked
// every chain through v_readlane and moved its windows by DPP, 45 k vector instructions for a block of 2048 samples.
//   mean            a chunk of 64 quotients goes to LDS and comes back by same-address reads: one v_add_f64 per sample.
//   autocorrelation lane L owns lags 2L and 2L + 1 (51 lanes): its two window values are c[j - 2L] and the one before, so a
//                   step needs ONE new value per lane -- from a ring of 256 centred samples in LDS (+ 64 mirrored behind its
//                   end: the 64 reads of a chunk have compile-time offsets) -- the wave-uniform c[j] by a same-address read of
//                   the same ring, two multiplies and two adds: every product rounded before it is added, every accumulator
//                   in ascending j.
//   Schur           in registers: lane L holds columns 2L, 2L + 1 of gen0 / gen1; gen1[j + 1] of the odd column is the next
//                   lane's register (one DPP move of 64 bits per stage).
//   residues        1024 samples at a time, lane L owning 16 CONSECUTIVE samples and sliding a statically addressed window of
//                   16 registers over its history (one new LDS word and 16 multiply-adds per tap) -- in FP64 (v_fma_f64, full
//                   rate) where that is exact: 2^34 + sum |a[j]| x max |s| < 2^53 bounds every partial sum of integers, in any
//                   order.  A block beyond that bound (21-bit noise with a long predictor; nothing 16-bit) takes the 64-bit
//                   wrap-around taps in a window that moves one lane per tap, as the first version did for every block.
//   Rice parameter  by convexity (rice_plan_convex), not by all twenty sums.
// Occupancy is what this kernel lives on -- its mean, its Schur stages and its accumulators are dependent chains -- and the residue
// filter's register window is what bounds it: 32 samples per lane (128 VGPRs of window and sums, 168 in all, 12 KB of LDS) allow three
// waves per SIMD, 16 per lane (125 / 7.9 KB) four or, with 16 registers spilled outside the loops, five: 763 / 667 / 655 us for 11,625
// blocks of 2048 samples.
#define SELA_GEN_WAVES 5
constexpr int kGenPad = 128;                                   // "no sample" / the samples before a stretch, in front of it
constexpr int kGenPerLane = 16;                                // consecutive samples a lane owns in a stretch of the residue filter
constexpr int kGenPadShift = kGenPerLane == 32 ? 5 : 4;        // index i stored at i + (i >> shift): the lanes' strides fall on different banks
constexpr int kGenStretch = kGenPerLane * kWave;               // samples per pass of the residue filter
constexpr int kGenSBufWords = (kGenPad + kGenStretch) + ((kGenPad + kGenStretch) >> kGenPadShift);
static_assert(kGenPerLane == 16 || kGenPerLane == 32, "the window is a power of two of registers");
constexpr int kGenRing = 256, kGenMirror = 64;

struct AnalyseLds {
    union {
        struct {
            alignas(16) double ring[kGenRing + kGenMirror];
            alignas(16) double chunk[64];
        } ac;
        int32_t st[kGenSBufWords];
    };
    double kk[104];   // reflection coefficients, then the dequantised ones
    int64_t a[104];   // Q35 predictor
    double af[104];   // the same as doubles (FP64 taps)
    int32_t q[128];
};

typedef const volatile __attribute__((address_space(3))) double* LdsDoubles; // (volatile: the reads stay ds_read_b64, 2 LDS cycles each)
// Two neighbours at once.  Lane L's window starts 2 L doubles below lane 0's, a stride of 16 bytes: as ds_read_b64 the two halves
// of a half-wave fall on the same banks (SQ_LDS_BANK_CONFLICT 23 % of SQ_LDS_IDX_ACTIVE), as ds_read_b128 the 64 lanes read 1 KB
// of consecutive LDS, four conflict-free passes for two steps' values.
typedef double __attribute__((ext_vector_type(2))) DoublePair;
typedef const volatile __attribute__((address_space(3))) DoublePair* LdsPairs;

// x = s / 32767 (src/lpc/residue_generator.cpp:12-18): q0 = s * RN(1/32767), one residual fma, one correction fma equal the
// correctly rounded quotient for every |s| <= 70000 (exhaustive check: tests/test_host_logic.py); anything larger divides.
__device__ __forceinline__ double scale_any(int32_t v, bool small /* wave-uniform: every lane's |v| <= 70000 */)
{
    if (small) {
        constexpr double r = 1.0 / SELA_SAMPLE_SCALE;
        const double x = (double)v;
        const double q0 = x * r;
        const double e = __builtin_fma(-SELA_SAMPLE_SCALE, q0, x);
        return __builtin_fma(e, r, q0);
    }
    return (double)v / SELA_SAMPLE_SCALE;
}

// Taps j0 + JJ + 1 .. j0 + kGenPerLane of the residue filter in FP64 (the window of sela_encode.hip's fir_taps_f64): tap j uses
// win[(t - j) mod kGenPerLane] = s[kGenPerLane lane + t - j] and loads the one new element s[kGenPerLane lane - j].
template <int JJ>
__device__ __forceinline__ void gen_taps_f64(int j0, int order, int lane, const int32_t* sT, const double* a_f, double (&win)[kGenPerLane], double (&acc)[kGenPerLane])
{
    const int j = j0 + JJ + 1;
    if (j > order)
        return;
    const double aj = read_first_lane(a_f[j]);
    const int e = kGenPad + kGenPerLane * lane - j;
    win[(kGenPerLane - JJ - 1) & (kGenPerLane - 1)] = (double)sT[e + (e >> kGenPadShift)];
#pragma unroll
    for (int t = 0; t < kGenPerLane; t++)
        acc[t] = __builtin_fma(aj, win[(t - JJ - 1) & (kGenPerLane - 1)], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (JJ < kGenPerLane - 1)
        gen_taps_f64<JJ + 1>(j0, order, lane, sT, a_f, win, acc);
}

// rice::RiceEncoder::calculateOptimumRiceParam (src/rice/rice_encoder.cpp:20-33) for the values load(0 .. n): the FIRST k in
// [0, 20) that minimises bits(k) = sum(u >> k) + n (1 + k).  The reference evaluates all 20 candidates; the same answer needs
// a few of them because bits() is convex in k (the argument of rice_plan, sela_encode.hip): the first minimum is the smallest
// k with T(k) - T(k + 1) <= n, found by walking from a guess near log2(mean u).  Exact 64-bit sums.  t0 = T(0) = sum of the
// zig-zagged values.
template <typename Load>
__device__ __forceinline__ void rice_plan_convex(Load load, uint32_t n, int lane, uint64_t t0, uint32_t& best_k, uint64_t& best_bits)
{
    constexpr uint32_t kLast = SELA_MAX_RICE_PARAM - 1;
    auto T = [&](uint32_t k) -> uint64_t {
        uint64_t part = 0;
        for (uint32_t i = lane; i < n; i += 64)
            part += zigzag32(load(i)) >> k;
        return wave_sum_wrap(part);
    };
    const uint64_t mean = n ? t0 / n : 0;
    uint32_t k = mean ? 63u - (uint32_t)__clzll(mean) : 0u;
    k = k > kLast - 1 ? kLast - 1 : k;
    uint64_t ta = k ? T(k) : t0;
    uint64_t tb = T(k + 1);
    if (ta - tb <= n) { // bits(k + 1) >= bits(k): the first minimum is at or below k
        while (k > 0) {
            const uint64_t tc = k != 1 ? T(k - 1) : t0;
            if (tc - ta > n)
                break;
            k--;
            tb = ta;
            ta = tc;
        }
    } else { // still descending: move up
        for (;;) {
            k++;
            ta = tb;
            if (k == kLast)
                break;
            tb = T(k + 1);
            if (ta - tb <= n)
                break;
        }
    }
    best_k = k;
    best_bits = ta + (uint64_t)n * (1 + k);
}

template <bool kIn16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SELA_GEN_WAVES, SELA_GEN_WAVES))) void k_generic_analyse(const void* __restrict__ input, uint32_t n_frames, uint32_t channels,
    uint32_t n_sig, uint32_t n, int32_t* __restrict__ sig_ws, int32_t* __restrict__ res_ws, int32_t* __restrict__ q_ws, GenericMeta* __restrict__ meta, uint32_t force_wrap_taps /* tests: every block on the 64-bit wrap-around taps */)
{
    __shared__ __attribute__((aligned(16))) AnalyseLds lds;
    const uint32_t b = blockIdx.x;
    if (b >= n_frames * n_sig)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = b / n_sig, sg = b % n_sig;
    int32_t* const s = sig_ws + (size_t)b * n;
    int32_t* const r = res_ws + (size_t)b * n;
    uint32_t flags = 0;

    // ---- the signal: a channel, or channel 0 - channel 1 of an exactly-stereo frame (src/frame/frame_encoder.cpp:18-24);
    //      x = s / 32767 and its sequential sum (:27-29) ---------------------------------------------------------------------
    auto load_sample = [&](uint32_t j) -> int32_t {
        if (kIn16) {
            const int16_t* pcm = static_cast<const int16_t*>(input) + (size_t)f * n * channels;
            return sg < channels ? (int32_t)pcm[(size_t)j * channels + sg] : (int32_t)pcm[(size_t)j * channels] - (int32_t)pcm[(size_t)j * channels + 1];
        }
        const int32_t* pl = static_cast<const int32_t*>(input) + (size_t)f * channels * n;
        return sg < channels ? pl[(size_t)sg * n + j] : (int32_t)((uint32_t)pl[j] - (uint32_t)pl[(size_t)n + j]);
    };
    double sum = 0.0;
    uint32_t mag_lane = 0; // the largest |sample| of this lane
    {
        int32_t v_next = (uint32_t)lane < n ? load_sample(lane) : 0;
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool valid = j < n;
            const int32_t v = v_next;
            if (j0 + 64 < n)
                v_next = j + 64 < n ? load_sample(j + 64) : 0;
            if (valid)
                s[j] = v;
            const uint32_t mag = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
            mag_lane = max(mag_lane, mag);
            const double x = valid ? scale_any(v, !__any(mag > 70000u)) : 0.0;
            lds.ac.chunk[lane] = x;
            wave_sync();
            const LdsDoubles ch = (LdsDoubles)lds.ac.chunk;
            if (n - j0 >= 64u) {
                const LdsPairs cp = (LdsPairs)lds.ac.chunk;
#pragma unroll
                for (int l = 0; l < 32; l++) {
                    const DoublePair two = cp[l];
                    sum += two.x;
                    sum += two.y;
                }
            } else {
                const int cnt = (int)(n - j0);
                for (int l = 0; l < cnt; l++)
                    sum += ch[l];
            }
            wave_sync();
        }
    }
    const double mean = sum / (double)n;
    const uint32_t s_mag = wave_max_u32(mag_lane);
    const bool small_samples = s_mag <= 70000u;
    // the other lanes' s[] is read below: the stores have left the CU, nothing older is served from its vector cache
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    // ---- autocorrelation (:33-38): lags 2 lane and 2 lane + 1 ---------------------------------------------------------------
    double acc_e = 0.0, acc_o = 0.0;
    {
        for (int i = lane; i < kGenRing + kGenMirror; i += 64)
            lds.ac.ring[i] = 0.0; // c before the block: a product with it adds +-0, which leaves an accumulator as it is
        wave_sync();
        double B = 0.0; // c[j - 1 - 2 lane]
        int32_t v_next = (uint32_t)lane < n ? s[lane] : 0;
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            const bool valid = j0 + lane < n;
            const int32_t v = v_next;
            if (j0 + 64 < n)
                v_next = j0 + 64 + lane < n ? s[j0 + 64 + lane] : 0;
            const double c_mine = valid ? scale_any(v, small_samples) - mean : 0.0;
            const uint32_t slab = (j0 >> 6) & 3u;
            lds.ac.ring[64 * slab + lane] = c_mine;
            if (slab == 0)
                lds.ac.ring[kGenRing + lane] = c_mine;
            wave_sync();
            const LdsDoubles rl = (LdsDoubles)lds.ac.ring + ((j0 - 2u * (uint32_t)lane) & (uint32_t)(kGenRing - 1)); // c[j0 - 2 lane]
            if (n - j0 >= 64u) {
                // (the window values are fetched kAcAhead pairs ahead of their use: an LDS read takes ~100 cycles to come back)
                constexpr int kAcAhead = 4;
                const LdsPairs rp = (LdsPairs)rl; // j0 and 2 lane are even and the ring is 16-byte aligned
                DoublePair ahead[kAcAhead];
#pragma unroll
                for (int u = 0; u < kAcAhead; u++)
                    ahead[u] = rp[u];
                // the wave-uniform multiplier c[j] comes by a same-address LDS read as well (two v_readlane of the register that holds the
                // chunk cost 4.1 k of the block's 23.3 k vector instructions: 0.775 -> 0.756 ms at 11,625 blocks)
                const LdsPairs cp = (LdsPairs)((LdsDoubles)lds.ac.ring + (j0 & (uint32_t)(kGenRing - 1)));
                DoublePair mult[kAcAhead];
#pragma unroll
                for (int u = 0; u < kAcAhead; u++)
                    mult[u] = cp[u];
#pragma unroll
                for (int u = 0; u < 32; u++) {
                    const DoublePair cj = mult[u % kAcAhead];
                    if (u + kAcAhead < 32)
                        mult[u % kAcAhead] = cp[u + kAcAhead];
                    const DoublePair A = ahead[u % kAcAhead];
                    if (u + kAcAhead < 32)
                        ahead[u % kAcAhead] = rp[u + kAcAhead];
                    {
                        const double pe = cj.x * A.x, po = cj.x * B; // step j0 + 2 u
                        acc_e += pe;
                        acc_o += po;
                    }
                    {
                        const double pe = cj.y * A.y, po = cj.y * A.x; // step j0 + 2 u + 1
                        acc_e += pe;
                        acc_o += po;
                    }
                    B = A.y;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const int cnt = (int)(n - j0);
                for (int t = 0; t < cnt; t++) {
                    const double cj = read_lane(c_mine, t);
                    const double A = rl[t];
                    const double pe = cj * A, po = cj * B;
                    acc_e += pe;
                    acc_o += po;
                    B = A;
                }
            }
            wave_sync();
        }
    }

    // ---- normalise (:41-44), Schur recursion (:47-68), always 100 stages ------------------------------------------------------
    // gen0 = gen1 = ac[1 .. 100]: lane L holds columns 2L (= ac[2L + 1], its own odd lag) and 2L + 1 (= ac[2L + 2], the next
    // lane's even lag).  A stage: gen1'[j] = gen1[j + 1] + k gen0[j]; gen0'[j] = gen1[j + 1] k + gen0[j] for the columns still
    // alive (what lies beyond them is never read by a live column).
    {
        const double ac0 = read_lane(acc_e, 0);
        const double n_e = acc_e / ac0, n_o = acc_o / ac0;
        double g1a = n_o, g1b = wave_shl1(0.0, n_e);
        double g0a = g1a, g0b = g1b;
        double err = 1.0;
        double g = read_lane(g1a, 0);
        double k = -g / err;
        err += g * k;
        if (lane == 0)
            lds.kk[0] = k;
#pragma unroll 1
        for (int i = 1; i < kMaxOrder; i++) {
            const double next = wave_shl1(0.0, g1a); // gen1[2L + 2]
            const double n1a = g1b + k * g0a, n0a = g1b * k + g0a;
            const double n1b = next + k * g0b, n0b = next * k + g0b;
            g1a = n1a, g0a = n0a, g1b = n1b, g0b = n0b;
            g = read_lane(g1a, 0);
            k = -g / err;
            err += g * k;
            if (lane == 0)
                lds.kk[i] = k;
        }
    }
    wave_sync();
    const double k_lo = lds.kk[lane];
    const double k_hi = lane + 64 < kMaxOrder ? lds.kk[lane + 64] : 0.0;
    wave_sync();

    // ---- order (:70-78), quantise (:80-96) ----------------------------------------------------------------------------------
    int order;
    {
        const unsigned long long b_lo = __ballot(fabs(k_lo) > SELA_ORDER_THRESHOLD);
        const unsigned long long b_hi = __ballot(lane + 64 < kMaxOrder && fabs(k_hi) > SELA_ORDER_THRESHOLD);
        order = b_hi ? 128 - __clzll(b_hi) : (b_lo ? 64 - __clzll(b_lo) : 1);
    }
    {
        const double sqrt2 = SELA_SQRT2;
        double v_lo;
        if (lane == 0)
            v_lo = floor(64 * (-1 + (sqrt2 * sqrt(k_lo + 1))));
        else if (lane == 1)
            v_lo = floor(64 * (-1 + (sqrt2 * sqrt(-k_lo + 1))));
        else
            v_lo = floor(64 * k_lo);
        const double v_hi = floor(64 * k_hi);
        const int32_t q_lo = isnan(v_lo) ? 0 : trunc_to_i32(v_lo);
        const int32_t q_hi = isnan(v_hi) ? 0 : trunc_to_i32(v_hi);
        if (lane < order) {
            lds.q[lane] = q_lo;
            lds.kk[lane] = order <= 1 ? 0.0 : dequant(lane, q_lo, flags);
        }
        if (lane + 64 < order) {
            lds.q[lane + 64] = q_hi;
            lds.kk[lane + 64] = dequant(lane + 64, q_hi, flags);
        }
    }
    wave_sync();
    step_up(lds.kk, lds.a, order, lane, flags);
    for (int i = lane; i < kMaxOrder; i += 64)
        q_ws[(size_t)b * kMaxOrder + i] = i < order ? lds.q[i] : 0;

    // ---- residues (:98-119): r[0] = s[0]; r[i] = s[i] - (int32)((2^34 + sum_{j=1..min(i,order)} a[j] s[i-j]) >> 35) ----------
    if ((uint32_t)order >= n)
        flags |= SELA_HIP_FLAG_SHORT_BLOCK; // the reference's warm-up loop reads samples[1 .. order] (:104-110): past its vector
    // which form: FP64 taps are exact while 2^34 + sum |a[j]| x max |s| < 2^53
    bool fp64_taps;
    {
        bool modest = true; // every |a[j]| below 2^39: their sum fits 2^46 and each is two exact FP64 halves
        uint64_t a_abs = 0;
        for (int j = 1 + lane; j <= order; j += 64) {
            const int64_t aj = lds.a[j];
            const uint64_t mag = aj < 0 ? 0 - (uint64_t)aj : (uint64_t)aj;
            modest &= mag < ((uint64_t)1 << 39);
            a_abs += modest ? mag : 0u;
            const int64_t top = aj >> 20;
            lds.af[j] = ((double)(int32_t)(top >> 16) * 65536.0 + (double)(uint32_t)((uint64_t)top & 0xFFFFu)) * 1048576.0 + (double)(uint32_t)((uint64_t)aj & 0xFFFFFu);
        }
        const uint64_t a_sum = wave_sum_wrap(a_abs);
        const uint64_t room = ((uint64_t)1 << 53) - ((uint64_t)1 << (SELA_Q_SHIFT - 1));
        fp64_taps = !__any(!modest) && (a_sum == 0 || (uint64_t)s_mag < room / a_sum) && !force_wrap_taps;
        wave_sync();
    }
    uint64_t t0_lane = 0; // sum of this lane's zig-zagged residues
    bool wide = false;
    if (fp64_taps) {
        int32_t* const sT = lds.st;
        for (int m = lane; m < kGenPad; m += 64)
            sT[m + (m >> kGenPadShift)] = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += kGenStretch) {
            if (i0) { // the stretch before this one left its last 128 samples behind: they become this one's history
                int32_t keep[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int e = kGenStretch + lane + 64 * h; // = kGenPad + (kGenStretch - kGenPad) + ...
                    keep[h] = sT[e + (e >> kGenPadShift)];
                }
                wave_sync();
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int e = lane + 64 * h;
                    sT[e + (e >> kGenPadShift)] = keep[h];
                }
            }
#pragma unroll 8
            for (int t = 0; t < kGenPerLane; t++) {
                const uint32_t i = i0 + lane + 64 * t;
                const int e = kGenPad + lane + 64 * t;
                sT[e + (e >> kGenPadShift)] = i < n ? s[i] : 0;
            }
            wave_sync();
            const int32_t* mine_s = sT + (kGenPad + kGenPerLane * lane) + ((kGenPad + kGenPerLane * lane) >> kGenPadShift); // &s[i0 + kGenPerLane lane]: no pad word inside
            double win_f[kGenPerLane], acc_f[kGenPerLane];
#pragma unroll
            for (int t = 0; t < kGenPerLane; t++) {
                acc_f[t] = (double)((int64_t)1 << (SELA_Q_SHIFT - 1));
                win_f[t] = (double)mine_s[t];
            }
#pragma unroll 1
            for (int j0 = 0; j0 < order; j0 += kGenPerLane)
                gen_taps_f64<0>(j0, order, lane, sT, lds.af, win_f, acc_f);
            int32_t rr[kGenPerLane];
#pragma unroll
            for (int t = 0; t < kGenPerLane; t++) { // floor(sum / 2^35): |.| < 2^18
                const int32_t pred = (int32_t)__builtin_floor(acc_f[t] * (1.0 / (double)((int64_t)1 << SELA_Q_SHIFT)));
                rr[t] = (int32_t)((uint32_t)mine_s[t] - (uint32_t)pred);
            }
            const uint32_t first = i0 + (uint32_t)kGenPerLane * (uint32_t)lane;
#pragma unroll
            for (int t = 0; t < kGenPerLane; t++)
                if (first + t < n) {
                    r[first + t] = rr[t];
                    t0_lane += zigzag32(rr[t]);
                    wide |= (rr[t] >= (1 << 30)) || (rr[t] < -(1 << 30));
                }
            wave_sync(); // (the samples are re-staged by the next stretch)
        }
    } else {
        // 64-bit wrap-around taps: a round's 64 samples, the 64 before them and the 64 before those in registers; tap j wants
        // sample i - j = the value j lanes down, so a window starts as the round's own samples and moves up one lane per tap,
        // lane 0 fed from the rounds before (zero before the block: a tap that reaches there adds 0).
        const uint32_t o = (uint32_t)order;
        int32_t before1 = 0, before2 = 0; // samples i0 - 64 + lane, i0 - 128 + lane
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const bool valid = i0 + lane < n;
            const int32_t mine = valid ? s[i0 + lane] : 0;
            uint64_t temp = (uint64_t)1 << (SELA_Q_SHIFT - 1);
            int32_t win = mine;
#pragma unroll 4
            for (uint32_t j = 1; j <= o; j++) {
                const int32_t feed = j <= 64u ? __builtin_amdgcn_readlane(before1, (int)(64u - j)) : __builtin_amdgcn_readlane(before2, (int)(128u - j));
                win = __builtin_amdgcn_update_dpp(feed, win, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                temp += (uint64_t)lds.a[j] * (uint64_t)(int64_t)win;
            }
            if (valid) {
                const int32_t rt = (int32_t)((uint32_t)mine - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
                r[i0 + lane] = rt;
                t0_lane += zigzag32(rt);
                wide |= (rt >= (1 << 30)) || (rt < -(1 << 30));
            }
            before2 = before1;
            before1 = mine;
        }
    }
    // the lanes read each other's residues back for the Rice plan
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    // ---- the two Rice plans -------------------------------------------------------------------------------------------------
    uint32_t ck, rk;
    uint64_t cbits, rbits;
    {
        const int32_t* const q = lds.q;
        uint64_t c0 = 0;
        for (int i = lane; i < order; i += 64)
            c0 += zigzag32(q[i]);
        rice_plan_convex([&](uint32_t i) { return q[i]; }, (uint32_t)order, lane, wave_sum_wrap(c0), ck, cbits);
        rice_plan_convex([&](uint32_t i) { return r[i]; }, n, lane, wave_sum_wrap(t0_lane), rk, rbits);
    }
    if (__any(wide))
        flags |= SELA_HIP_FLAG_RICE_RANGE;
    // requiredInts = ceil((float)bits / 32) (rice_encoder.cpp:37,63): exact below 2^24 bits, and a count the u16 field of the
    // subframe can carry (65535 words = 2,097,120 bits) is far below that
    const uint32_t cwords = cbits <= 32ull * kCoefWordsCap ? words_for_bits(cbits) : 0xFFFFFFFFu;
    const uint32_t rwords = rbits <= 32ull * 65535ull ? words_for_bits(rbits) : 0xFFFFFFFFu;
    if (cwords == 0xFFFFFFFFu || rwords == 0xFFFFFFFFu)
        flags |= SELA_HIP_FLAG_WORDS_CAP;
    flags = wave_or(flags);
    if (lane == 0) {
        GenericMeta m;
        m.order = (uint32_t)order, m.coef_k = ck, m.coef_words = cwords, m.res_k = rk, m.res_words = rwords, m.flags = flags, m.form = fp64_taps ? 0u : 1u;
        meta[b] = m;
    }
}

// ---- plan: the stereo decision, frame sizes, offsets -- one workgroup -----------------------------------------------------
// src/frame/frame_encoder.cpp:64-72: the difference candidate wins iff its words (coefficients + residues) are FEWER.
// 1024 threads over tiles of 4096 frames.  Within a tile frame f is sized by thread f mod 1024 -- the records' loads of one pass
// are independent and coalesced (a thread that walks consecutive frames waits for memory once per frame: the second version,
// 46 us at 3875 frames; the first version's single thread walking 256 partial sums: 60) -- its subframes' word counts go to LDS;
// then every thread sums a contiguous run of the tile's frames, the runs are scanned across the workgroup (shuffles within a
// wave, the sixteen waves' totals through LDS), and the offsets and the subframes' word bases are written frame by frame again.
constexpr int kPlanThreads = 1024, kPlanTile = 4096;
__global__ __launch_bounds__(kPlanThreads) void k_generic_plan(const GenericMeta* __restrict__ meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig,
    uint64_t base_bytes, uint64_t* __restrict__ frame_offsets /* [n_frames + 1], absolute */, uint64_t* __restrict__ word_base /* [n_frames * channels + 1] */,
    uint32_t* __restrict__ chosen /* [n_frames * channels]: signal index */, uint32_t* __restrict__ status, uint64_t* __restrict__ total_words_out)
{
    constexpr int kWaves = kPlanThreads / 64;
    __shared__ uint64_t wave_bytes[kWaves], wave_words[kWaves];
    __shared__ uint32_t frame_words[kPlanTile]; // words of the frames of one tile (their bytes follow: 4 + 12 channels + 4 words)
    __shared__ uint32_t frame_first_hi[kPlanTile];
    __shared__ uint32_t all_flags;
    const uint32_t t = threadIdx.x;
    const int lane = t % 64, wave = t / 64;
    if (t == 0)
        all_flags = 0;
    uint64_t base_b = 0, base_w = 0; // bytes / words of the tiles before this one (the same in every thread)
    uint32_t my_flags = 0;
    for (uint32_t tile0 = 0; tile0 < n_frames; tile0 += kPlanTile) {
        const uint32_t tile_n = min((uint32_t)kPlanTile, n_frames - tile0);
        __syncthreads(); // (the tile before has been read)
        for (uint32_t i = t; i < tile_n; i += kPlanThreads) {
            const uint32_t f = tile0 + i;
            uint32_t words = 0;
            // (values, not pointers, are selected: a record chosen by pointer is loaded again through it -- a chain of dependent
            // loads, 32 us for 3875 frames)
            const GenericMeta* const fm = meta + (size_t)f * n_sig;
            if (channels == 2) {
                const uint32_t w0 = fm[0].coef_words + fm[0].res_words, w1 = fm[1].coef_words + fm[1].res_words, w2 = fm[2].coef_words + fm[2].res_words;
                const uint32_t f0 = fm[0].flags, f1 = fm[1].flags, f2 = fm[2].flags;
                const bool diff = (uint64_t)fm[2].coef_words + fm[2].res_words < (uint64_t)fm[1].coef_words + fm[1].res_words;
                my_flags |= f0 | f1 | f2; // (both candidates were computed by the reference too: either's trouble is the frame's)
                chosen[(size_t)f * 2] = 0;
                chosen[(size_t)f * 2 + 1] = diff ? 2u : 1u;
                const uint32_t fs = diff ? f2 : f1, ws = diff ? w2 : w1;
                words = ((f0 & SELA_HIP_FLAG_WORDS_CAP) ? 0u : w0) + ((fs & SELA_HIP_FLAG_WORDS_CAP) ? 0u : ws);
            } else {
                for (uint32_t c = 0; c < channels; c++) {
                    const uint32_t fl = fm[c].flags, w = fm[c].coef_words + fm[c].res_words;
                    my_flags |= fl;
                    chosen[(size_t)f * channels + c] = c;
                    words += (fl & SELA_HIP_FLAG_WORDS_CAP) ? 0u : w; // (<= 255 x 131,070 words: fits)
                }
            }
            frame_words[i] = words;
        }
        __syncthreads();
        const uint32_t per = (tile_n + kPlanThreads - 1) / kPlanThreads;
        const uint32_t begin = min(t * per, tile_n), end = min(begin + per, tile_n);
        uint64_t my_words = 0;
        for (uint32_t i = begin; i < end; i++)
            my_words += frame_words[i];
        const uint64_t my_bytes = (uint64_t)(end - begin) * (4 + (uint64_t)channels * SELA_SUBFRAME_HEADER_BYTES) + 4 * my_words;
        auto scan64 = [&](uint64_t v) -> uint64_t { // inclusive, within the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
                const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
                if (lane >= d)
                    v += ((uint64_t)hi << 32) | lo;
            }
            return v;
        };
        const uint64_t incl_b = scan64(my_bytes), incl_w = scan64(my_words);
        if (lane == 63)
            wave_bytes[wave] = incl_b, wave_words[wave] = incl_w;
        __syncthreads();
        uint64_t at_w = base_w + incl_w - my_words, tile_b = 0, tile_w = 0;
        for (int w = 0; w < kWaves; w++) {
            const uint64_t wb = wave_bytes[w], ww = wave_words[w];
            if (w < wave)
                at_w += ww;
            tile_b += wb, tile_w += ww;
        }
        // every frame's first word: an exclusive scan of the run in place (frame_words[i] <- words before frame i in the stream) ...
        for (uint32_t i = begin; i < end; i++) {
            const uint32_t w = frame_words[i];
            frame_words[i] = (uint32_t)(at_w - base_w); // (relative to the tile: 4096 frames x 255 x 131,070 words fit 2^32 only just -- kept in 64 bits below)
            frame_first_hi[i] = (uint32_t)((at_w - base_w) >> 32);
            at_w += w;
        }
        __syncthreads();
        // ... and then frame by frame again as in the first pass -- thread f mod 1024, independent loads (the third version: the
        // run's owner walked its frames' subframes through `chosen`, a chain of dependent loads: 38 us at 3875 frames)
        for (uint32_t i = t; i < tile_n; i += kPlanThreads) {
            const uint32_t f = tile0 + i;
            const uint64_t first = base_w + (((uint64_t)frame_first_hi[i] << 32) | frame_words[i]);
            uint64_t w_at = first;
            const GenericMeta* const fm = meta + (size_t)f * n_sig;
            if (channels == 2) {
                const uint32_t w0 = fm[0].coef_words + fm[0].res_words, f0 = fm[0].flags;
                word_base[(size_t)f * 2] = w_at;
                word_base[(size_t)f * 2 + 1] = w_at + ((f0 & SELA_HIP_FLAG_WORDS_CAP) ? 0u : w0);
            } else {
                for (uint32_t c = 0; c < channels; c++) {
                    const uint32_t fl = fm[c].flags, w = fm[c].coef_words + fm[c].res_words;
                    word_base[(size_t)f * channels + c] = w_at;
                    w_at += (fl & SELA_HIP_FLAG_WORDS_CAP) ? 0u : w;
                }
            }
            // bytes before frame f = (frames before it) x (4 + 12 channels) + 4 x (words before it)
            frame_offsets[f] = base_bytes + (uint64_t)f * (4 + (uint64_t)channels * SELA_SUBFRAME_HEADER_BYTES) + 4 * first;
        }
        base_b += tile_b, base_w += tile_w;
    }
    if (my_flags)
        atomicOr(&all_flags, my_flags);
    __syncthreads();
    if (t == 0) {
        frame_offsets[n_frames] = base_bytes + base_b;
        word_base[(size_t)n_frames * channels] = base_w;
        *total_words_out = base_w;
        atomicOr(&status[0], all_flags);
    }
}

// ---- pack: the chosen candidates' two Rice streams, one wave per subframe ---------------------------------------------------
__global__ __launch_bounds__(64) void k_generic_pack(const GenericMeta* __restrict__ meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n,
    const int32_t* __restrict__ res_ws, const int32_t* __restrict__ q_ws, const uint32_t* __restrict__ chosen, const uint64_t* __restrict__ word_base,
    uint32_t* __restrict__ words /* zeroed */, uint64_t words_cap)
{
    const uint32_t sub = blockIdx.x;
    if (sub >= n_frames * channels)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = sub / channels;
    const size_t b = (size_t)f * n_sig + chosen[sub];
    // everything that only needs to know WHICH block is asked for in one go: the block's record, its coefficients (all 100 places of
    // its row: how many count is in the record), the first stretch of its residues
    int32_t first[kPackPerLane], coefs[kPackPerLane];
    pack_fetch(res_ws + b * n, n, 0, lane, first);
    pack_fetch(q_ws + b * kMaxOrder, kMaxOrder, 0, lane, coefs);
    const GenericMeta m = meta[b];
    if (m.flags & (SELA_HIP_FLAG_WORDS_CAP | SELA_HIP_FLAG_RICE_RANGE))
        return;
    if (word_base[sub] + m.coef_words + m.res_words > words_cap) // (the host sized the words by an estimate: it packs again at the plan's size)
        return;
    __shared__ uint32_t win[kPackWindow];
    uint32_t* const out = words + word_base[sub];
    rice_pack_stream<true>(q_ws + b * kMaxOrder, m.order, m.coef_k, out, lane, win, coefs);
    rice_pack_stream<true>(res_ws + b * n, n, m.res_k, out + m.coef_words, lane, win, first);
}

// ---- assemble: the on-disk bytes (src/file/sela_file.cpp:115-135), one workgroup per subframe -------------------------------
constexpr int kAsmThreads = 256;
__global__ __launch_bounds__(kAsmThreads) void k_generic_assemble(const GenericMeta* __restrict__ meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig,
    uint32_t n, const uint32_t* __restrict__ chosen, const uint64_t* __restrict__ word_base, const uint32_t* __restrict__ words,
    const uint64_t* __restrict__ frame_offsets, uint64_t base_bytes, uint8_t* __restrict__ frames /* the launch's first byte */, uint64_t frames_cap)
{
    const uint32_t sub = blockIdx.x;
    if (sub >= n_frames * channels)
        return;
    const uint32_t f = sub / channels, c = sub % channels;
    if (frame_offsets[f + 1] - base_bytes > frames_cap) // (the call reports ECAPACITY from frame_offsets[n_frames])
        return;
    uint64_t at = frame_offsets[f] - base_bytes + 4;
    for (uint32_t i = 0; i < c; i++) // (<= 254 steps; every thread the same)
        at += SELA_SUBFRAME_HEADER_BYTES + 4 * (word_base[(size_t)f * channels + i + 1] - word_base[(size_t)f * channels + i]);
    const uint32_t sgn = chosen[sub];
    const GenericMeta m = meta[(size_t)f * n_sig + sgn];
    const bool broken = (m.flags & (SELA_HIP_FLAG_WORDS_CAP | SELA_HIP_FLAG_RICE_RANGE)) != 0;
    const uint32_t cw = broken ? 0 : m.coef_words, rw = broken ? 0 : m.res_words;
    uint8_t* const dst = frames + at;
    const uint32_t* const src = words + word_base[sub];
    const uint32_t t = threadIdx.x;
    if (c == 0 && t < 4)
        frames[frame_offsets[f] - base_bytes + t] = (uint8_t)(SELA_SYNC_WORD >> (8 * t));
    if (t == 0) {
        dst[0] = (uint8_t)c;
        dst[1] = sgn >= channels ? 1 : 0;                      // subFrameType
        dst[2] = (uint8_t)(sgn >= channels ? c - 1 : c);       // parentChannelNumber
        dst[3] = (uint8_t)m.coef_k;
        dst[4] = (uint8_t)cw, dst[5] = (uint8_t)(cw >> 8);
        dst[6] = (uint8_t)m.order;
        uint8_t* const h = dst + 7 + 4 * (size_t)cw;
        h[0] = (uint8_t)m.res_k;
        h[1] = (uint8_t)rw, h[2] = (uint8_t)(rw >> 8);
        h[3] = (uint8_t)n, h[4] = (uint8_t)(n >> 8);
    }
    for (size_t i = t; i < 4 * (size_t)cw; i += kAsmThreads)
        dst[7 + i] = (uint8_t)(src[i >> 2] >> (8 * (i & 3)));
    uint8_t* const rdst = dst + 12 + 4 * (size_t)cw;
    for (size_t i = t; i < 4 * (size_t)rw; i += kAsmThreads)
        rdst[i] = (uint8_t)(src[cw + (i >> 2)] >> (8 * (i & 3)));
}

// ---- decode: one wave per subframe -------------------------------------------------------------------------------------------
// src/frame/frame_decoder.cpp:19-36 / :42-61 for one subframe: rice::RiceDecoder on the coefficients (n = order) and on the
// residues (n = samplesPerChannel) -- src/rice/rice_decoder.cpp:21-52: the codewords' lengths walked serially, wave-uniform, the
// remainder fields fetched by all lanes at once (RiceScan below) -- and
// lpc::SampleGenerator (src/lpc/sample_generator.cpp:11-39) in transposed form: lane l carries the part of sample i + 1 + l's
// prediction that is already known, P[l] = sum a[i + 1 + l - i'] s[i'] over the samples i' <= i; a new sample adds a[l + 1] s[i]
// to every lane after the lanes have moved down by one.  (Integer arithmetic mod 2^64: the order of the additions is free.)
// The parse is serial only in the codewords' LENGTHS.  RiceScan walks those, wave-uniform, on a 64-bit register window that is
// refilled word by word from a lane-held buffer of the frame's next 64 aligned words (one coalesced load per 2048 bits, no
// load on the chain); each lane keeps the start and the unary count of ONE codeword of a chunk of 64, and then all lanes
// fetch their remainder fields at once.  Bits at or beyond the stream's end read as zero (a stream that ends early raises
// RICE_OVERRUN), exactly as the byte-wise reader of the first version did.
struct RiceScan {
    const uint32_t* words; // the frame's aligned words
    uint32_t n_words;
    uint64_t end;          // the stream's end, in bits from the frame's first byte
    uint64_t pos;          // bit position of the window's first bit
    uint64_t cur;          // the next `avail` bits of the stream, LSB first
    uint32_t avail, wpos;  // wpos: the next word to append
    uint32_t buf, wbase;   // lane l holds word wbase + l (masked to the stream)
    int lane;

    __device__ __forceinline__ uint32_t masked_word(uint32_t w) const
    {
        const uint64_t first = 32ull * w;
        if (w >= n_words || first >= end)
            return 0u;
        const uint32_t v = words[w];
        return first + 32 > end ? v & ((1u << (uint32_t)(end - first)) - 1u) : v;
    }
    __device__ __forceinline__ void load_buf(uint32_t base)
    {
        wbase = base;
        buf = masked_word(base + (uint32_t)lane);
    }
    __device__ __forceinline__ void refill()
    {
        while (avail <= 32) {
            if (wpos - wbase >= 64u)
                load_buf(wpos);
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)buf, (int)(wpos - wbase));
            cur |= (uint64_t)w << avail;
            avail += 32;
            wpos++;
        }
    }
    __device__ __forceinline__ void start(const uint32_t* frame_words, uint32_t frame_word_count, uint64_t first_bit, uint64_t end_bit, int lane_)
    {
        words = frame_words, n_words = frame_word_count, end = end_bit, lane = lane_;
        pos = first_bit;
        wpos = (uint32_t)(first_bit >> 5);
        load_buf(wpos);
        cur = 0, avail = 0;
        refill();
        const uint32_t skip = (uint32_t)first_bit & 31; // the stream starts inside its first aligned word
        cur >>= skip, avail -= skip;
        refill();
    }
    __device__ __forceinline__ void consume(uint32_t n) // n <= 32
    {
        cur >>= n;
        avail -= n;
        pos += n;
        refill();
    }
    // one codeword: where it starts and how many ones it has; the window moves past its k remainder bits
    __device__ __forceinline__ void next(uint32_t k, uint64_t& at, uint32_t& ones)
    {
        at = pos;
        ones = 0;
        for (;;) {
            const uint32_t low = (uint32_t)cur;
            const uint32_t t = low == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~low);
            ones += t;
            consume(t);
            if (t < 32)
                break; // (beyond the stream's end everything is zero: the run ends there at the latest)
        }
        consume(1 + k);
    }
};

// the k bits at bit position p of the frame, the first one the most significant (src/rice/rice_decoder.cpp:37-40); bits at or
// beyond `end` are zero.  Per lane (every lane its own position).
__device__ __forceinline__ uint32_t remainder_bits(const uint8_t* base, uint64_t p, uint64_t end, uint32_t k)
{
    if (k == 0)
        return 0u;
    const uint64_t at = p >> 3;
    uint64_t w = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
        w |= (uint64_t)(8 * (at + i) < end ? base[at + i] : 0u) << (8 * i);
    uint32_t v = (uint32_t)(w >> (p & 7));
    if (p + 32 > end) // (the stream ends inside these 32 bits; its end is a whole number of bytes, so only bytes were masked above)
        v &= p >= end ? 0u : (uint32_t)(((uint64_t)1 << (end - p)) - 1);
    return __brev(v) >> (32 - k);
}

// up to 64 values of a stream, value i0 + lane in each lane (lanes beyond `count` get 0)
__device__ inline int32_t rice_chunk(RiceScan& sc, const uint8_t* base, uint32_t k, uint32_t count, bool& overrun)
{
    uint64_t my_at = 0;
    uint32_t my_ones = 0;
    for (uint32_t i = 0; i < count; i++) {
        uint64_t at;
        uint32_t ones;
        sc.next(k, at, ones);
        if ((uint32_t)sc.lane == i)
            my_at = at, my_ones = ones;
    }
    const bool live = (uint32_t)sc.lane < count;
    const uint64_t field = my_at + my_ones + 1;
    overrun |= live && field + k > sc.end;
    const uint32_t rem = live ? remainder_bits(base, field, sc.end, k) : 0u;
    const uint32_t u = (my_ones << k) | rem; // (uint32 arithmetic: src/rice/rice_decoder.cpp:35)
    return live ? (int32_t)((u >> 1) ^ (0u - (u & 1u))) : 0;
}

__global__ __launch_bounds__(64) void k_generic_decode(const uint8_t* __restrict__ frames, const uint64_t* __restrict__ frame_offsets, uint64_t base_bytes,
    uint32_t n_frames, uint32_t channels, uint32_t stride, int32_t* __restrict__ dec_ws /* [n_frames][channels][stride] by subframe position */,
    GenericSubInfo* __restrict__ info /* [n_frames][channels] */, uint32_t* __restrict__ status)
{
    __shared__ int64_t a_lds[kMaxOrder + 1];
    __shared__ int32_t q_lds[256];
    const uint32_t sub = blockIdx.x;
    if (sub >= n_frames * channels)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = sub / channels, c = sub % channels;
    const uint8_t* const fb = frames + (frame_offsets[f] - base_bytes);
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    GenericSubInfo si;
    si.channel = si.type = si.parent = 0, si.n = 0, si.ok = 0;
    uint32_t flags = 0;
    // ---- the headers up to this subframe (src/file/sela_file.cpp:58-91) ----
    bool ok = fbytes >= 4 && fbytes < (1ull << 31) && fb[0] == 0x00 && fb[1] == 0xFF && fb[2] == 0x55 && fb[3] == 0xAA;
    uint64_t p = 4;
    uint32_t ck = 0, cw = 0, order = 0, rk = 0, rw = 0, n = 0;
    for (uint32_t i = 0; ok && i <= c; i++) {
        if (p + 12 > fbytes) {
            ok = false;
            break;
        }
        si.channel = fb[p], si.type = fb[p + 1], si.parent = fb[p + 2];
        ck = fb[p + 3], cw = fb[p + 4] | ((uint32_t)fb[p + 5] << 8), order = fb[p + 6];
        const uint64_t p2 = p + 7 + 4 * (uint64_t)cw;
        if (p2 + 5 > fbytes) {
            ok = false;
            break;
        }
        rk = fb[p2], rw = fb[p2 + 1] | ((uint32_t)fb[p2 + 2] << 8), n = fb[p2 + 3] | ((uint32_t)fb[p2 + 4] << 8);
        const uint64_t next = p2 + 5 + 4 * (uint64_t)rw;
        if (next > fbytes) {
            ok = false;
            break;
        }
        if (i < c)
            p = next;
    }
    ok = ok && order <= (uint32_t)kMaxOrder && ck < 32 && rk < 32 && n <= stride;
    if (!ok) {
        if (lane == 0) {
            info[sub] = si;
            atomicOr(&status[0], (uint32_t)SELA_HIP_FLAG_BAD_FRAME);
            atomicAdd(&status[1], 1u);
        }
        return;
    }
    si.n = n, si.ok = 1;
    if (n == 0 || n <= order) // lpc::SampleGenerator writes samples[0] and samples[1 .. order] whatever the length (src/lpc/sample_generator.cpp:14-22): past its vector
        flags |= SELA_HIP_FLAG_SHORT_BLOCK;
    const uint32_t* const fw = reinterpret_cast<const uint32_t*>(fb); // (frames are whole words at word-aligned offsets)
    const uint32_t n_fw = (uint32_t)(fbytes >> 2);
    bool overrun = false;
    RiceScan sc;
    // ---- the coefficients ----
    {
        const uint64_t first = 8 * (p + 7);
        sc.start(fw, n_fw, first, first + 32ull * cw, lane);
        for (uint32_t i0 = 0; i0 < order; i0 += 64) {
            const int32_t v = rice_chunk(sc, fb, ck, order - i0 < 64u ? order - i0 : 64u, overrun);
            if (i0 + lane < order)
                q_lds[i0 + lane] = v;
        }
    }
    wave_sync();
    {
        const uint32_t o = order;
        const int32_t q_lo = (uint32_t)lane < o ? q_lds[lane] : 0;
        const int32_t q_hi = (uint32_t)lane + 64 < o ? q_lds[lane + 64] : 0;
        const double k_lo = (uint32_t)lane < o ? (o <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
        const double k_hi = (uint32_t)lane + 64 < o ? dequant(lane + 64, q_hi, flags) : 0.0;
        step_up_regs(k_lo, k_hi, a_lds, (int)o, lane, flags);
    }
    // lane l: a[l + 1], a[l + 65] (0 beyond the order)
    const uint64_t a_lo = (uint32_t)lane + 1 <= order ? (uint64_t)a_lds[lane + 1] : 0;
    const uint64_t a_hi = (uint32_t)lane + 65 <= order ? (uint64_t)a_lds[lane + 65] : 0;
    // ---- residues -> samples, 64 at a time ----
    {
        const uint64_t first = 8 * (p + 12 + 4 * (uint64_t)cw);
        sc.start(fw, n_fw, first, first + 32ull * rw, lane);
    }
    int32_t* const out = dec_ws + (size_t)sub * stride;
    uint64_t p_lo = 0, p_hi = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t cnt = n - i0 < 64u ? n - i0 : 64u;
        const int32_t mine = rice_chunk(sc, fb, rk, cnt, overrun);
        int32_t made = 0;
        for (uint32_t l = 0; l < cnt; l++) {
            const int32_t res = __builtin_amdgcn_readlane(mine, (int)l);
            const uint64_t sum = read_first_lane(p_lo);
            const uint64_t temp = ((uint64_t)1 << (SELA_Q_SHIFT - 1)) - sum;
            const int32_t smp = (int32_t)((uint32_t)res - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
            if ((uint32_t)lane == l)
                made = smp;
            if (order > 64) { // two registers: lags 1..64 and 65..100
                const uint64_t carry = read_first_lane(p_hi);
                // lane l <- lane l + 1
                const uint32_t nlo = (uint32_t)wave_shl1((int)(uint32_t)carry, (int)(uint32_t)p_lo);
                const uint32_t nhi = (uint32_t)wave_shl1((int)(uint32_t)(carry >> 32), (int)(uint32_t)(p_lo >> 32));
                p_lo = (((uint64_t)nhi << 32) | nlo) + a_lo * (uint64_t)(int64_t)smp;
                p_hi = wave_shl1_zero(p_hi) + a_hi * (uint64_t)(int64_t)smp;
            } else {
                p_lo = wave_shl1_zero(p_lo) + a_lo * (uint64_t)(int64_t)smp;
            }
        }
        if (i0 + lane < n)
            out[i0 + lane] = made;
    }
    if (__any(overrun))
        flags |= SELA_HIP_FLAG_RICE_OVERRUN;
    flags = wave_or(flags);
    if (lane == 0) {
        info[sub] = si;
        if (flags)
            atomicOr(&status[0], flags);
    }
}

// ---- combine: src/frame/frame_decoder.cpp:17-69 over the decoded subframes of a frame ----------------------------------------
// One workgroup per (frame, slice of kCombineSlice samples): every slice walks the frame's subframes in the reference's order --
// independent ones first, then dependent ones IN SUBFRAME ORDER, a later subframe of a channel overwriting an earlier one, an
// unknown type skipped -- and moves its own samples only (a thread meets the same samples in every pass: the passes need no
// barrier for the data, only for the channels' counts).  A frame of 65535 samples is 16 workgroups instead of one.
// kOut16: interleaved int16 at sample_offsets[f] (src/file/wav_file.cpp:244-266 narrows so); every channel of the frame must
// have come out with the first one's length, or the frame counts as malformed.
constexpr int kCombineThreads = 256;
constexpr uint32_t kCombineSlice = 4096;
template <bool kOut16>
__global__ __launch_bounds__(kCombineThreads) void k_generic_combine(const int32_t* __restrict__ dec_ws, const GenericSubInfo* __restrict__ info, uint32_t n_frames,
    uint32_t channels, uint32_t stride, int32_t* __restrict__ all /* [n_frames][channels][stride] by channel */, uint32_t* __restrict__ counts,
    const uint64_t* __restrict__ sample_offsets, int16_t* __restrict__ pcm_out, uint32_t* __restrict__ status)
{
    __shared__ uint32_t cnt[256];
    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const uint32_t t = threadIdx.x;
    const uint32_t lo = blockIdx.y * kCombineSlice, hi = lo + kCombineSlice; // this workgroup's samples
    const bool first_slice = blockIdx.y == 0;
    const GenericSubInfo* const inf = info + (size_t)f * channels;
    int32_t* const fa = all + (size_t)f * channels * stride;
    const int32_t* const fd = dec_ws + (size_t)f * channels * stride;
    for (uint32_t c = t; c < channels; c += kCombineThreads)
        cnt[c] = 0;
    __syncthreads();
    bool bad = false;
    for (uint32_t c = 0; c < channels; c++) { // :17-37
        const GenericSubInfo si = inf[c];
        if (!si.ok || si.type != 0)
            continue;
        if (si.channel >= channels) {
            bad = true;
            continue;
        }
        const uint32_t end = min(si.n, hi);
        for (uint32_t i = lo + t; i < end; i += kCombineThreads)
            fa[(size_t)si.channel * stride + i] = fd[(size_t)c * stride + i];
        __syncthreads();
        if (t == 0)
            cnt[si.channel] = si.n;
        __syncthreads();
    }
    for (uint32_t c = 0; c < channels; c++) { // :40-69
        const GenericSubInfo si = inf[c];
        if (!si.ok || si.type != 1)
            continue;
        if (si.channel >= channels || si.parent >= channels || cnt[si.parent] < si.n) {
            bad = true;
            continue;
        }
        const uint32_t end = min(si.n, hi);
        for (uint32_t i = lo + t; i < end; i += kCombineThreads)
            fa[(size_t)si.channel * stride + i] = (int32_t)((uint32_t)fa[(size_t)si.parent * stride + i] - (uint32_t)fd[(size_t)c * stride + i]);
        __syncthreads();
        if (t == 0)
            cnt[si.channel] = si.n;
        __syncthreads();
    }
    if (kOut16) {
        const uint32_t n = (uint32_t)(sample_offsets[f + 1] - sample_offsets[f]);
        for (uint32_t c = 0; c < channels; c++)
            bad |= cnt[c] != n;
        if (!bad && lo < n) {
            const uint32_t end = min(n, hi);
            int16_t* const o = pcm_out + sample_offsets[f] * channels;
            if (channels == 1) {
                for (uint32_t i = lo + t; i < end; i += kCombineThreads)
                    o[i] = (int16_t)(uint16_t)fa[i];
            } else if (channels == 2) { // (a sample pair per thread: one 32-bit store)
                uint32_t* const o2 = reinterpret_cast<uint32_t*>(o);
                for (uint32_t i = lo + t; i < end; i += kCombineThreads)
                    o2[i] = ((uint32_t)fa[i] & 0xFFFFu) | ((uint32_t)fa[(size_t)stride + i] << 16);
            } else {
                for (size_t i = (size_t)lo * channels + t; i < (size_t)end * channels; i += kCombineThreads) {
                    const uint32_t smp = (uint32_t)(i / channels), c = (uint32_t)(i % channels);
                    o[i] = (int16_t)(uint16_t)fa[(size_t)c * stride + smp];
                }
            }
        }
    } else if (first_slice) {
        for (uint32_t c = t; c < channels; c += kCombineThreads)
            counts[(size_t)f * channels + c] = cnt[c];
    }
    if (bad && t == 0 && first_slice) {
        atomicOr(&status[0], (uint32_t)SELA_HIP_FLAG_BAD_FRAME);
        atomicAdd(&status[1], 1u);
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
size_t generic_encode_workspace_bytes(uint32_t n_frames, uint32_t channels, uint32_t n)
{
    const size_t n_sig = channels == 2 ? 3 : channels, blocks = (size_t)n_frames * n_sig;
    return blocks * n * (4 + 4) + blocks * kMaxOrder * 4 + blocks * sizeof(GenericMeta) + ((size_t)n_frames * channels + 1) * (8 + 4) + 1024;
}

static std::atomic<int> g_force_wrap_taps{0};
void set_generic_wrap_taps(int on) { g_force_wrap_taps.store(on, std::memory_order_relaxed); }

hipError_t launch_generic_analyse(const void* d_input, bool in16, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n, int32_t* d_sig,
    int32_t* d_res, int32_t* d_q, GenericMeta* d_meta, hipStream_t stream)
{
    const uint32_t blocks = n_frames * n_sig;
    if (blocks == 0)
        return hipSuccess;
    const uint32_t wrap = g_force_wrap_taps.load(std::memory_order_relaxed) ? 1u : 0u;
    if (in16)
        hipLaunchKernelGGL(k_generic_analyse<true>, dim3(blocks), dim3(64), 0, stream, d_input, n_frames, channels, n_sig, n, d_sig, d_res, d_q, d_meta, wrap);
    else
        hipLaunchKernelGGL(k_generic_analyse<false>, dim3(blocks), dim3(64), 0, stream, d_input, n_frames, channels, n_sig, n, d_sig, d_res, d_q, d_meta, wrap);
    return hipGetLastError();
}

hipError_t launch_generic_plan(const GenericMeta* d_meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint64_t base_bytes, uint64_t* d_frame_offsets,
    uint64_t* d_word_base, uint32_t* d_chosen, uint32_t* d_status, uint64_t* d_total_words, hipStream_t stream)
{
    hipLaunchKernelGGL(k_generic_plan, dim3(1), dim3(kPlanThreads), 0, stream, d_meta, n_frames, channels, n_sig, base_bytes, d_frame_offsets, d_word_base, d_chosen,
        d_status, d_total_words);
    return hipGetLastError();
}

hipError_t launch_generic_emit(const GenericMeta* d_meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n, const int32_t* d_res, const int32_t* d_q,
    const uint32_t* d_chosen, const uint64_t* d_word_base, uint32_t* d_words /* zeroed */, uint64_t words_cap, const uint64_t* d_frame_offsets, uint64_t base_bytes,
    uint8_t* d_frames, uint64_t frames_cap, hipStream_t stream)
{
    const uint32_t subs = n_frames * channels;
    if (subs == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_generic_pack, dim3(subs), dim3(64), 0, stream, d_meta, n_frames, channels, n_sig, n, d_res, d_q, d_chosen, d_word_base, d_words, words_cap);
    hipLaunchKernelGGL(k_generic_assemble, dim3(subs), dim3(kAsmThreads), 0, stream, d_meta, n_frames, channels, n_sig, n, d_chosen, d_word_base, d_words,
        d_frame_offsets, base_bytes, d_frames, frames_cap);
    return hipGetLastError();
}

hipError_t launch_generic_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels, uint32_t stride,
    int32_t* d_dec, GenericSubInfo* d_info, int32_t* d_all, uint32_t* d_counts, const uint64_t* d_sample_offsets, int16_t* d_pcm_out, uint32_t* d_status,
    bool fast_first, bool standard_path, hipStream_t stream)
{
    const uint32_t subs = n_frames * channels;
    if (subs == 0)
        return hipSuccess;
    if (fast_first) {
        const hipError_t e = launch_decode_subframes32(d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec, d_info, d_status, standard_path, stream);
        if (e != hipSuccess)
            return e;
    } else
        hipLaunchKernelGGL(k_generic_decode, dim3(subs), dim3(64), 0, stream, d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec, d_info, d_status);
    const dim3 grid(n_frames, (stride + kCombineSlice - 1) / kCombineSlice);
    if (d_pcm_out)
        hipLaunchKernelGGL(k_generic_combine<true>, grid, dim3(kCombineThreads), 0, stream, d_dec, d_info, n_frames, channels, stride, d_all, d_counts,
            d_sample_offsets, d_pcm_out, d_status);
    else
        hipLaunchKernelGGL(k_generic_combine<false>, grid, dim3(kCombineThreads), 0, stream, d_dec, d_info, n_frames, channels, stride, d_all, d_counts,
            d_sample_offsets, d_pcm_out, d_status);
    return hipGetLastError();
}

} // namespace sela
