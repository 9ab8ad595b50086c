// sela_generic.hip -- the frame path for ANY block length and 32-bit samples (gfx950).
//
// The reference's frame API is length-agnostic and 32-bit: data::WavFrame carries int32 samples per channel
// (src/include/data/wav_frame.hpp:8-16), lpc::ResidueGenerator loops over samples.size() (src/lpc/residue_generator.cpp:12-45,
// 98-119), a subframe brings its own samplesPerChannel (src/include/data/sela_sub_frame.hpp:27,
// src/frame/frame_decoder.cpp:24-25,48-49) and frame::FrameDecoder returns untruncated int32 (frame_decoder.cpp:64-71).
// The fast kernels (sela_encode.hip, sela_decode.hip) are built around the one shape the reference's CLI produces --
// 2048 samples of 16-bit PCM -- and everything else comes here: the same arithmetic, bit for bit, with the length a run-time
// value (1 .. 65535: the u16 field) and nothing assumed about the samples.  One wave per block / subframe, register
// windows and global-memory scratch instead of an LDS plan: the route for callers of the frame classes with odd shapes, not
// the one bench.py times.
//
// Encode: k_generic_analyse (samples -> order, q[], residues, the two Rice plans) -> k_generic_plan (stereo decision, frame
// sizes, offsets) -> k_generic_pack (the chosen candidates' Rice streams) -> k_generic_assemble (on-disk bytes).
// Decode: k_generic_decode (one wave per subframe: headers, Rice parse, synthesis, 32-bit) -> k_generic_combine
// (independent subframes first, then dependent ones in subframe order: src/frame/frame_decoder.cpp:17-69).
#include <hip/hip_runtime.h>

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

// rice::RiceEncoder::calculateOptimumRiceParam (src/rice/rice_encoder.cpp:20-33) for values v[0..n) of one stream: all 20
// candidates, the first minimum.  `wide`: a value whose int32 zig-zag overflows (undefined in the reference).
__device__ inline void rice_plan_stream(const int32_t* v, uint32_t n, int lane, uint32_t& best_k, uint64_t& best_bits, bool& wide)
{
    // (the twenty candidates in two halves, a pass over the values each, one reduction at a time: with all twenty sums and their
    // reductions in flight at once the kernel needed 180 registers -- two waves per SIMD)
    constexpr int kHalf = SELA_MAX_RICE_PARAM / 2;
    best_k = 0;
    best_bits = ~0ull;
    bool w = false;
#pragma unroll 1
    for (int k0 = 0; k0 < SELA_MAX_RICE_PARAM; k0 += kHalf) {
        uint64_t sum[kHalf];
#pragma unroll
        for (int k = 0; k < kHalf; k++)
            sum[k] = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            const int32_t x = v[i];
            w |= (x >= (1 << 30)) || (x < -(1 << 30));
            const uint32_t u = zigzag32(x) >> k0;
#pragma unroll
            for (int k = 0; k < kHalf; k++)
                sum[k] += u >> k;
        }
#pragma unroll
        for (int k = 0; k < kHalf; k++) {
            const uint64_t bits = wave_sum_wrap(sum[k]) + (uint64_t)n * (uint64_t)(1 + k0 + k);
            if (bits < best_bits) // strict: the FIRST minimum
                best_bits = bits, best_k = (uint32_t)(k0 + k);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wide = __any(w);
}

// pack the stream v[0..n) with parameter k into zeroed words.  The 64 codewords of a round are OR-ed together in an LDS window
// (`win`, kPackWindow words) and go out as whole words: plain stores, but for the round's first and last word, which the
// rounds before and behind it share (an atomic OR each; first version: every codeword piece an atomic OR in global memory,
// 0.85 ms of a 3.6 ms encode at 3875 stereo frames).  A round longer than the window (unary runs of thousands of bits) goes
// piece by piece as before.
constexpr uint32_t kPackWindow = 512; // words: 16,384 bits for 64 codewords
__device__ inline void rice_pack_stream(const int32_t* v, uint32_t n, uint32_t k, uint32_t* out, int lane, uint32_t* win)
{
    uint64_t base = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const bool valid = i0 + lane < n;
        const uint32_t u = valid ? zigzag32(v[i0 + lane]) : 0u;
        const uint64_t len = valid ? (uint64_t)(u >> k) + 1 + k : 0;
        // exclusive scan of 64-bit lengths over the lanes
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
            put_codeword(out, base + incl - len, u, k);
        }
        base += total;
    }
}

} // namespace

// ---- analysis: one wave per (frame, signal) -------------------------------------------------------------------------------
// lpc::ResidueGenerator::process (src/lpc/residue_generator.cpp:121-134) as written, with samples.size() = n:
//   x[j] = s[j] / 32767 (:12-18);  mean = (sequential sum) / n (:27-30);  ac[lag] = sequential sum over j = lag .. n-1 of
//   (x[j] - mean) * (x[j - lag] - mean), lag = 0..100 -- a lane per lag (two for lanes 0..36), all lanes walking j upwards
//   together, so every accumulator sees the reference's order of additions (:33-38);  Schur (:47-68);  order (:70-78);
//   quantise (:80-96);  dequantise + step-up (linear_predictor.cpp:16-61);  residues, a sample per lane (:98-119);
// then both Rice plans (rice_encoder.cpp:20-33, 37).
template <bool kIn16>
__global__ __launch_bounds__(64) void k_generic_analyse(const void* __restrict__ input, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n,
    int32_t* __restrict__ sig_ws, int32_t* __restrict__ res_ws, int32_t* __restrict__ q_ws, GenericMeta* __restrict__ meta)
{
    __shared__ double g0[128], g1[128], kk[128];
    __shared__ int64_t a_lds[kMaxOrder + 1];
    __shared__ int32_t q_lds[kMaxOrder];
    const uint32_t b = blockIdx.x;
    if (b >= n_frames * n_sig)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = b / n_sig, sg = b % n_sig;
    int32_t* const s = sig_ws + (size_t)b * n;
    int32_t* const r = res_ws + (size_t)b * n;
    uint32_t flags = 0;

    // ---- the signal: a channel, or channel 0 - channel 1 of an exactly-stereo frame (src/frame/frame_encoder.cpp:18-24);
    //      x = s / 32767 and its sequential sum --------------------------------------------------------------------------------
    double sum = 0.0;
    for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool valid = j < n;
        int32_t v = 0;
        if (valid) {
            if (kIn16) {
                const int16_t* pcm = static_cast<const int16_t*>(input) + (size_t)f * n * channels;
                v = sg < channels ? (int32_t)pcm[(size_t)j * channels + sg] : (int32_t)pcm[(size_t)j * channels] - (int32_t)pcm[(size_t)j * channels + 1];
            } else {
                const int32_t* pl = static_cast<const int32_t*>(input) + (size_t)f * channels * n;
                v = sg < channels ? pl[(size_t)sg * n + j] : (int32_t)((uint32_t)pl[j] - (uint32_t)pl[(size_t)n + j]);
            }
            s[j] = v;
        }
        const double x = valid ? (double)v / SELA_SAMPLE_SCALE : 0.0;
        const int cnt = n - j0 < 64u ? (int)(n - j0) : 64;
        for (int l = 0; l < cnt; l++)
            sum += read_lane(x, l);
    }
    const double mean = sum / (double)n;
    __threadfence(); // the other lanes' s[] is read below

    // ---- autocorrelation: lags lane and lane + 64 ---------------------------------------------------------------------------
    // In registers: w_lo / w_hi hold c[j - lane] / c[j - 64 - lane] (c = x - mean; zero before the block: a product with it adds
    // +-0, which leaves an accumulator as it is -- c is finite).  A step moves both windows up one lane (DPP), feeds lane 0 with the
    // new sample (a scalar, read from the lane that loaded it) and with what leaves the first window, and adds c[j] * window: every
    // accumulator sees its products in the reference's order.  64 samples per load, no memory access inside a round.
    // (First version: c[] in global scratch, three loads per step -- 4.6 ms of analysis for 3875 stereo frames, this form 1.4.)
    double acc_lo = 0.0, acc_hi = 0.0;
    {
        double w_lo = 0.0, w_hi = 0.0;
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            const bool valid = j0 + lane < n;
            const double c_mine = valid ? (double)s[j0 + lane] / SELA_SAMPLE_SCALE - mean : 0.0;
            auto step = [&](int t) {
                const double cj = read_lane(c_mine, t);
                const double leaving = read_lane(w_lo, 63);
                w_lo = wave_shr1(cj, w_lo);
                w_hi = wave_shr1(leaving, w_hi);
                const double p_lo = cj * w_lo;
                const double p_hi = cj * w_hi;
                acc_lo += p_lo;
                acc_hi += p_hi;
            };
            if (n - j0 >= 64u) {
#pragma unroll
                for (int t = 0; t < 64; t++)
                    step(t);
            } else {
                const int cnt = (int)(n - j0);
                for (int t = 0; t < cnt; t++)
                    step(t);
            }
        }
    }
    const double ac0 = read_lane(acc_lo, 0);
    // normalise (:41-44): ac[i] /= ac[0] for i >= 1; ac[0] = 1.0
    const double ac_lo = lane == 0 ? 1.0 : acc_lo / ac0;
    const double ac_hi = acc_hi / ac0;

    // ---- Schur recursion (:47-68), always 100 stages ------------------------------------------------------------------------
    // gen0 = gen1 = ac[1..100]
    if (lane >= 1)
        g0[lane - 1] = g1[lane - 1] = ac_lo;
    if (lane + 64 <= kMaxOrder)
        g0[lane + 63] = g1[lane + 63] = ac_hi;
    wave_sync();
    double err = 1.0;
    {
        const double k0 = -g1[0] / err;
        err += g1[0] * k0;
        if (lane == 0)
            kk[0] = k0;
        double kprev = k0;
        for (int i = 1; i < kMaxOrder; i++) {
            const int count = kMaxOrder - i;
            double n1_lo = 0, n0_lo = 0, n1_hi = 0, n0_hi = 0;
            if (lane < count) {
                const double a = g1[lane + 1], c = g0[lane];
                n1_lo = a + kprev * c;
                n0_lo = a * kprev + c;
            }
            if (lane + 64 < count) {
                const double a = g1[lane + 65], c = g0[lane + 64];
                n1_hi = a + kprev * c;
                n0_hi = a * kprev + c;
            }
            wave_sync();
            if (lane < count)
                g1[lane] = n1_lo, g0[lane] = n0_lo;
            if (lane + 64 < count)
                g1[lane + 64] = n1_hi, g0[lane + 64] = n0_hi;
            wave_sync();
            const double ki = -g1[0] / err;
            err += g1[0] * ki;
            if (lane == 0)
                kk[i] = ki;
            kprev = ki;
        }
    }
    wave_sync();
    const double k_lo = kk[lane];
    const double k_hi = lane + 64 < kMaxOrder ? kk[lane + 64] : 0.0;

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
            q_lds[lane] = q_lo;
            kk[lane] = order <= 1 ? 0.0 : dequant(lane, q_lo, flags);
        }
        if (lane + 64 < order) {
            q_lds[lane + 64] = q_hi;
            kk[lane + 64] = dequant(lane + 64, q_hi, flags);
        }
    }
    wave_sync();
    step_up(kk, a_lds, order, lane, flags);
    for (int i = lane; i < kMaxOrder; i += 64)
        q_ws[(size_t)b * kMaxOrder + i] = i < order ? q_lds[i] : 0;

    // ---- residues (:98-119): r[0] = s[0]; r[i] = s[i] - (int32)((2^34 + sum_{j=1..min(i,order)} a[j] s[i-j]) >> 35) ----------
    if ((uint32_t)order >= n)
        flags |= SELA_HIP_FLAG_SHORT_BLOCK; // the reference's warm-up loop reads samples[1 .. order] (:104-110): past its vector
    // In registers as well: a round's 64 samples, the 64 before them and the 64 before those; tap j wants sample i - j = the
    // value j lanes down, so a window starts as the round's own samples and moves up one lane per tap, lane 0 fed from the
    // rounds before (zero before the block: a tap that reaches there adds 0, which is what "taps = min(i, order)" means).
    {
        const uint32_t o = (uint32_t)order;
        int32_t before1 = 0, before2 = 0; // samples i0 - 64 + lane, i0 - 128 + lane
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const bool valid = i0 + lane < n;
            const int32_t mine = valid ? s[i0 + lane] : 0;
            uint64_t temp = (uint64_t)1 << (SELA_Q_SHIFT - 1);
            int32_t win = mine;
#pragma unroll 4
            for (uint32_t j = 1; j <= o; j++) {
                // lane 0's tap j is sample i0 - j: lane 64 - j of the round before, lane 128 - j of the one before that
                const int32_t feed = j <= 64u ? __builtin_amdgcn_readlane(before1, (int)(64u - j)) : __builtin_amdgcn_readlane(before2, (int)(128u - j));
                win = __builtin_amdgcn_update_dpp(feed, win, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                temp += (uint64_t)a_lds[j] * (uint64_t)(int64_t)win;
            }
            if (valid)
                r[i0 + lane] = (int32_t)((uint32_t)mine - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
            before2 = before1;
            before1 = mine;
        }
    }
    __threadfence();

    // ---- the two Rice plans -------------------------------------------------------------------------------------------------
    uint32_t ck, rk;
    uint64_t cbits, rbits;
    bool cwide, rwide;
    rice_plan_stream(q_ws + (size_t)b * kMaxOrder, (uint32_t)order, lane, ck, cbits, cwide);
    rice_plan_stream(r, n, lane, rk, rbits, rwide);
    if (cwide || rwide)
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
        m.order = (uint32_t)order, m.coef_k = ck, m.coef_words = cwords, m.res_k = rk, m.res_words = rwords, m.flags = flags;
        meta[b] = m;
    }
}

// ---- plan: the stereo decision, frame sizes, offsets -- one workgroup -----------------------------------------------------
// src/frame/frame_encoder.cpp:64-72: the difference candidate wins iff its words (coefficients + residues) are FEWER.
constexpr int kPlanThreads = 256;
__global__ __launch_bounds__(kPlanThreads) void k_generic_plan(const GenericMeta* __restrict__ meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig,
    uint64_t base_bytes, uint64_t* __restrict__ frame_offsets /* [n_frames + 1], absolute */, uint64_t* __restrict__ word_base /* [n_frames * channels + 1] */,
    uint32_t* __restrict__ chosen /* [n_frames * channels]: signal index */, uint32_t* __restrict__ status, uint64_t* __restrict__ total_words_out)
{
    __shared__ uint64_t part_bytes[kPlanThreads], part_words[kPlanThreads];
    __shared__ uint32_t part_flags[kPlanThreads];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_frames + kPlanThreads - 1) / kPlanThreads;
    const uint32_t f_begin = min(t * per, n_frames), f_end = min(f_begin + per, n_frames);
    auto frame_plan = [&](uint32_t f, uint64_t& bytes, uint64_t& words, uint32_t& flags, bool write, uint64_t words_before) {
        bytes = 4;
        words = 0;
        for (uint32_t c = 0; c < channels; c++) {
            uint32_t sgn = c;
            const GenericMeta* m = meta + (size_t)f * n_sig + c;
            if (channels == 2 && c == 1) {
                const GenericMeta* d = meta + (size_t)f * n_sig + 2;
                flags |= d->flags; // (both candidates were computed by the reference too: either's trouble is the frame's)
                if ((uint64_t)d->coef_words + d->res_words < (uint64_t)m->coef_words + m->res_words)
                    sgn = 2, m = d;
            }
            flags |= m->flags;
            if (write) {
                chosen[(size_t)f * channels + c] = sgn;
                word_base[(size_t)f * channels + c] = words_before + words;
            }
            const uint64_t w = (m->flags & SELA_HIP_FLAG_WORDS_CAP) ? 0 : (uint64_t)m->coef_words + m->res_words;
            words += w;
            bytes += SELA_SUBFRAME_HEADER_BYTES + 4 * w;
        }
    };
    uint64_t my_bytes = 0, my_words = 0;
    uint32_t my_flags = 0;
    for (uint32_t f = f_begin; f < f_end; f++) {
        uint64_t bt, w;
        frame_plan(f, bt, w, my_flags, false, 0);
        my_bytes += bt, my_words += w;
    }
    part_bytes[t] = my_bytes, part_words[t] = my_words, part_flags[t] = my_flags;
    __syncthreads();
    if (t == 0) {
        uint64_t run_b = 0, run_w = 0;
        uint32_t fl = 0;
        for (int i = 0; i < kPlanThreads; i++) {
            const uint64_t pb = part_bytes[i], pw = part_words[i];
            part_bytes[i] = run_b, part_words[i] = run_w;
            run_b += pb, run_w += pw;
            fl |= part_flags[i];
        }
        frame_offsets[n_frames] = base_bytes + run_b;
        word_base[(size_t)n_frames * channels] = run_w;
        *total_words_out = run_w;
        atomicOr(&status[0], fl);
    }
    __syncthreads();
    uint64_t at_b = part_bytes[t], at_w = part_words[t];
    for (uint32_t f = f_begin; f < f_end; f++) {
        uint64_t bt, w;
        uint32_t fl = 0;
        frame_offsets[f] = base_bytes + at_b;
        frame_plan(f, bt, w, fl, true, at_w);
        at_b += bt, at_w += w;
    }
}

// ---- pack: the chosen candidates' two Rice streams, one wave per subframe ---------------------------------------------------
__global__ __launch_bounds__(64) void k_generic_pack(const GenericMeta* __restrict__ meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n,
    const int32_t* __restrict__ res_ws, const int32_t* __restrict__ q_ws, const uint32_t* __restrict__ chosen, const uint64_t* __restrict__ word_base,
    uint32_t* __restrict__ words /* zeroed */)
{
    const uint32_t sub = blockIdx.x;
    if (sub >= n_frames * channels)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = sub / channels;
    const size_t b = (size_t)f * n_sig + chosen[sub];
    const GenericMeta m = meta[b];
    if (m.flags & (SELA_HIP_FLAG_WORDS_CAP | SELA_HIP_FLAG_RICE_RANGE))
        return;
    __shared__ uint32_t win[kPackWindow];
    uint32_t* const out = words + word_base[sub];
    rice_pack_stream(q_ws + b * kMaxOrder, m.order, m.coef_k, out, lane, win);
    rice_pack_stream(res_ws + b * n, n, m.res_k, out + m.coef_words, lane, win);
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

// ---- combine: src/frame/frame_decoder.cpp:17-69 over the decoded subframes of a frame, one workgroup per frame ----------------
// kOut16: interleaved int16 at sample_offsets[f] (src/file/wav_file.cpp:244-266 narrows so); every channel of the frame must
// have come out with the first one's length, or the frame counts as malformed.
constexpr int kCombineThreads = 256;
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
        for (uint32_t i = t; i < si.n; i += kCombineThreads)
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
        for (uint32_t i = t; i < si.n; i += kCombineThreads)
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
        if (!bad) {
            int16_t* const o = pcm_out + sample_offsets[f] * channels;
            for (size_t i = t; i < (size_t)n * channels; i += kCombineThreads) {
                const uint32_t smp = (uint32_t)(i / channels), c = (uint32_t)(i % channels);
                o[i] = (int16_t)(uint16_t)fa[(size_t)c * stride + smp];
            }
        }
    } else {
        for (uint32_t c = t; c < channels; c += kCombineThreads)
            counts[(size_t)f * channels + c] = cnt[c];
    }
    if (bad && t == 0) {
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

hipError_t launch_generic_analyse(const void* d_input, bool in16, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n, int32_t* d_sig,
    int32_t* d_res, int32_t* d_q, GenericMeta* d_meta, hipStream_t stream)
{
    const uint32_t blocks = n_frames * n_sig;
    if (blocks == 0)
        return hipSuccess;
    if (in16)
        hipLaunchKernelGGL(k_generic_analyse<true>, dim3(blocks), dim3(64), 0, stream, d_input, n_frames, channels, n_sig, n, d_sig, d_res, d_q, d_meta);
    else
        hipLaunchKernelGGL(k_generic_analyse<false>, dim3(blocks), dim3(64), 0, stream, d_input, n_frames, channels, n_sig, n, d_sig, d_res, d_q, d_meta);
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
    const uint32_t* d_chosen, const uint64_t* d_word_base, uint32_t* d_words /* zeroed */, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint8_t* d_frames,
    uint64_t frames_cap, hipStream_t stream)
{
    const uint32_t subs = n_frames * channels;
    if (subs == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_generic_pack, dim3(subs), dim3(64), 0, stream, d_meta, n_frames, channels, n_sig, n, d_res, d_q, d_chosen, d_word_base, d_words);
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
    if (d_pcm_out)
        hipLaunchKernelGGL(k_generic_combine<true>, dim3(n_frames), dim3(kCombineThreads), 0, stream, d_dec, d_info, n_frames, channels, stride, d_all, d_counts,
            d_sample_offsets, d_pcm_out, d_status);
    else
        hipLaunchKernelGGL(k_generic_combine<false>, dim3(n_frames), dim3(kCombineThreads), 0, stream, d_dec, d_info, n_frames, channels, stride, d_all, d_counts,
            d_sample_offsets, d_pcm_out, d_status);
    return hipGetLastError();
}

} // namespace sela
