// sela_decode.hip -- MI355X (gfx950) decoder kernel of the SELA frame path.
//
// ONE launch: k_decode_frames, one WORKGROUP per frame, one WAVE per subframe.  A wave takes its subframe
// from the frame bytes to finished samples without leaving the CU:
//
//   headers                                   layout of src/file/sela_file.cpp:58-91
//   rice::RiceDecoder x2                      src/rice/rice_decoder.cpp:11-61   (segment-parallel, below)
//   dequantise + step-up                      src/lpc/linear_predictor.cpp:16-61
//   lpc::SampleGenerator::generateSamples     src/lpc/sample_generator.cpp:11-30 (lane-ring recurrence, below)
//
// then, after a workgroup barrier, frame::FrameDecoder's second pass (out[ch] = parent - difference,
// src/frame/frame_decoder.cpp:40-69) and the int16 interleave of src/file/wav_file.cpp:244-257, stored
// coalesced.  Residues and samples live in LDS only; nothing goes through a workspace.
//
// ---- Rice parsing across the lanes of a wave ------------------------------------------------------------
// A Golomb-Rice stream is a serial bit parse -- where codeword i+1 starts depends on codeword i -- but it
// RESYNCHRONISES: a parser dropped at an arbitrary bit is in step with the true parse after a few
// codewords (inside a unary run it is in step at the next terminator; inside a remainder field it has a
// chance of roughly E[quotient]/k per codeword).  So every lane gets a ZONE of the stream (whole words):
//
//   phase A  lane i parses from the first bit of its zone to the zone's end and sets one bit per codeword
//            start in a bitmap (only lane i writes its zone's words);
//   phase B  lane i keeps going through the following zones until it stands on a start that a later lane
//            marked: from there on the two trajectories are one.  m_i = that position (or "end of stream");
//   resolve  the true trajectory starts at the stream's first bit in its first lane; it follows that lane's
//            path to m, which lies in the zone of a later lane j AND on lane j's path, then lane j's path to
//            m_j, ... (a walk of <= 64 hops on the scalar unit).  Lanes the chain skips were never in step
//            and decode nothing.  A lane on the chain counts its codewords from its true entry (popcount
//            of its bitmap words + what it parsed in phase B); an exclusive scan gives the index of its
//            first value;
//   pass 2   the lanes on the chain decode their codewords from their true entries, in parallel, straight
//            into the LDS arrays the synthesis reads.
//
// The first kCoefLanes lanes do the same for the coefficient stream (<= 100 codewords, its own k), in the
// same loops: both streams are zones of ONE bit space, the subframe's aligned words.  tools/parse_model.py
// is an executable model of this algorithm; tests/test_host_logic.py runs it against the CPU oracle.
//
// A frame whose subframes do not fit the LDS plan (a Rice stream longer than any 16-bit audio produces,
// or more than 8 channels) takes the GENERIC mode of the same kernel: a plain serial parse straight from
// global memory, then the same synthesis.  Slow, complete, and never needed by files the encoder writes
// for 1..8 channels.
#include "sela_device.h"

namespace sela {

constexpr int kDecMaxWaves = 8;     // waves per workgroup; frames with more channels loop (generic mode)
constexpr int kDecMaxChannels = 17; // [channels][2048] int32 in LDS
constexpr int kCoefLanes = 4;       // lanes of a wave that parse the coefficient stream
constexpr int kResLanes = kWave - kCoefLanes;
// Aligned words of one subframe the segment-parallel parser holds in LDS: coefficient words + 2 + residue
// words.  With 8192 B of values, this and q[] a stereo workgroup needs 27,104 B: six per CU (12 waves).
constexpr int kStreamCap = 1200;
constexpr int kStreamMargin = 4;    // zero words behind the stream: a window may run this far past the end
constexpr uint32_t kEndOfStream = 0xFFFFFFFFu;

typedef const volatile __attribute__((address_space(3))) uint64_t* LdsTable;

// ---- per-wave LDS scratch --------------------------------------------------------------------------------
struct SynthTables {
    union {
        struct {
            double k[104];  // dequantised reflection coefficients
            int64_t a[104]; // Q35 predictor
        };
        uint64_t tab[256];  // synthesis coefficient table (build_synth_table), replaces k[] and a[]
    };
};
struct DecWaveFast {
    union {
        uint32_t strm[kStreamCap + kStreamMargin]; // the subframe's aligned words (until the values are decoded)
        SynthTables t;
    };
    int32_t q[128];
};
struct DecWaveGeneric {
    SynthTables t;
    int32_t q[128];
};
static_assert(sizeof(SynthTables) == 2048 && sizeof(SynthTables) <= (kStreamCap + kStreamMargin) * 4, "tables overlay the stream words");
static_assert(kStreamCap + kStreamMargin <= kBlock, "the start bitmap overlays the value array");

__device__ __forceinline__ int32_t rice_value(uint32_t ones, uint32_t field, uint32_t k)
{
    const uint32_t rem = k ? (__brev(field) >> (32 - k)) : 0u; // remainder is MSB first in the stream
    const uint32_t u = (ones << k) | rem;                      // uint32 arithmetic as src/rice/rice_decoder.cpp:35
    return (int32_t)((u >> 1) ^ (0u - (u & 1u)));               // un-zig-zag, src/rice/rice_decoder.cpp:49-50
}

// ---- segment-parallel parse of one subframe (fast mode) ------------------------------------------------
// strm[0 .. cw + 2 + rw) = the subframe's aligned words from the one holding [coefficient word count |
// order | first coefficient byte] on, followed by kStreamMargin zero words.  Bit space: stream bit t of the
// array = bit t % 32 of word t / 32.  Coefficient stream = bits [24, 24 + 32 cw), residue stream = bits
// [32 (cw + 2), 32 (cw + 2 + rw)).  marks[] overlays vals[] (it is dead before the first value is stored).
// Outputs: q[0 .. order), vals[0 .. 2048); returns SELA_HIP_FLAG_RICE_OVERRUN or 0.
struct ParseProfile {
    long long t[4];
};

template <bool kProf>
__device__ inline uint32_t parse_subframe(const uint32_t* strm, uint32_t* marks, int32_t* vals, int32_t* q, uint32_t cw, uint32_t rw,
    uint32_t ck, uint32_t rk, uint32_t order, int lane, ParseProfile& prof)
{
    // ---- zones ---------------------------------------------------------------------------------------------
    const bool coef_lane = lane < kCoefLanes;
    const uint32_t zc = max(1u, (cw + 1 + kCoefLanes - 1) / kCoefLanes); // words per coefficient zone
    const uint32_t zr = max(1u, (rw + kResLanes - 1) / kResLanes);       // words per residue zone
    const uint32_t rs_word = cw + 2;
    uint32_t first_word, end_word, stream_end, k, need;
    if (coef_lane) {
        first_word = min((uint32_t)lane * zc, cw + 1);
        end_word = min((uint32_t)(lane + 1) * zc, cw + 1);
        stream_end = 24 + 32 * cw;
        k = ck;
        need = order;
    } else {
        const uint32_t r = (uint32_t)(lane - kCoefLanes);
        first_word = rs_word + min(r * zr, rw);
        end_word = rs_word + min((r + 1) * zr, rw);
        stream_end = 32 * (rs_word + rw);
        k = rk;
        need = (uint32_t)kBlock;
    }
    const uint32_t entry = lane == 0 ? 24u : 32 * first_word;
    const uint32_t zone_end = min(32 * end_word, stream_end);
    const uint32_t kmask = k ? (0xFFFFFFFFu >> (32 - k)) : 0u;

    // ---- phase A: own zone, marking every codeword start ---------------------------------------------------
    uint32_t pos = entry, n_own = 0;
    bool in_run = false; // inside a unary run longer than the 32-bit window
    // (predicated rather than branched: lanes that are through OR a zero into a word of the bitmap)
    while (__any(pos < zone_end)) {
        const uint32_t w = pos >> 5, sh = pos & 31;
        const uint32_t x = __builtin_amdgcn_alignbit(strm[w + 1], strm[w], sh);
        const bool act = pos < zone_end;
        const bool start = act && !in_run;
        atomicOr(&marks[w], start ? 1u << sh : 0u); // (LDS ds_or_b32; a zone's words are marked by its lane alone)
        n_own += start ? 1u : 0u;
        const bool full = x == 0xFFFFFFFFu;
        const uint32_t adv = full ? 32u : (uint32_t)__builtin_ctz(~x | 0x80000000u) + 1 + k;
        pos += act ? adv : 0u;
        in_run = act ? full : in_run;
    }
    wave_sync();
    if (kProf)
        prof.t[0] = clock64();

    // ---- phase B: on through the following zones until standing on a later lane's start -----------------------
    uint32_t n_cont = 0, merged = 0;
    bool walking = true;
    while (__any(walking)) {
        const uint32_t w = pos >> 5, sh = pos & 31;
        const uint32_t x = __builtin_amdgcn_alignbit(strm[w + 1], strm[w], sh);
        const uint32_t mk = marks[w];
        const bool at_start = walking && !in_run;
        const bool ended = at_start && pos >= stream_end;
        const bool met = at_start && !ended && ((mk >> sh) & 1u);
        merged = ended ? kEndOfStream : (met ? pos : merged);
        walking = walking && !ended && !met;
        n_cont += (at_start && walking) ? 1u : 0u;
        const bool full = x == 0xFFFFFFFFu;
        const uint32_t adv = full ? 32u : (uint32_t)__builtin_ctz(~x | 0x80000000u) + 1 + k;
        pos += walking ? adv : 0u;
        in_run = walking ? full : in_run;
    }
    if (kProf)
        prof.t[1] = clock64();

    // ---- resolve: the chains of lanes the true trajectories run through -----------------------------------------
    uint32_t succ = 64; // lane whose zone holds `merged`
    if (merged != kEndOfStream) {
        const uint32_t wm = merged >> 5;
        succ = wm < rs_word ? min(wm / zc, (uint32_t)kCoefLanes - 1) : (uint32_t)kCoefLanes + (wm - rs_word) / zr;
    }
    uint32_t e_true = kEndOfStream; // this lane's true entry; kEndOfStream = not on a chain
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        uint32_t cur = h ? (uint32_t)kCoefLanes : 0u;
        const uint32_t limit = h ? (uint32_t)kWave : (uint32_t)kCoefLanes;
        uint32_t e = h ? 32 * rs_word : 24u;
#pragma unroll 1
        for (int hop = 0; hop < kWave; hop++) {
            e_true = (uint32_t)lane == cur ? e : e_true;
            const uint32_t m_cur = (uint32_t)__builtin_amdgcn_readlane((int)merged, (int)cur);
            const uint32_t s_cur = (uint32_t)__builtin_amdgcn_readlane((int)succ, (int)cur);
            if (m_cur == kEndOfStream || s_cur <= cur || s_cur >= limit)
                break;
            e = m_cur;
            cur = s_cur;
        }
    }
    const bool on_chain = e_true != kEndOfStream;
    // codewords of this lane's path from its true entry: the marked starts at or behind the entry + phase B's
    uint32_t count = 0;
    {
        const uint32_t we = e_true >> 5;
        const uint32_t zmax = max(zc, zr);
        for (uint32_t j = 0; j < zmax; j++) {
            const bool valid = on_chain && we + j < end_word;
            if (!__any(valid))
                break;
            uint32_t word = valid ? marks[we + j] : 0u;
            if (j == 0)
                word &= 0xFFFFFFFFu << (e_true & 31);
            count += (uint32_t)__builtin_popcount(word);
        }
        count = on_chain ? count + n_cont : 0u;
    }
    (void)n_own;
    uint32_t idx = wave_exclusive_scan(count, lane);
    const uint32_t coef_total = (uint32_t)__builtin_amdgcn_readlane((int)idx, kCoefLanes);
    const uint32_t all_total = (uint32_t)__builtin_amdgcn_readlane((int)(idx + count), kWave - 1);
    const uint32_t res_total = all_total - coef_total;
    if (!coef_lane)
        idx -= coef_total;
    uint32_t remaining = idx < need ? min(count, need - idx) : 0u;
    wave_sync(); // every lane has read the bitmap: the values may overwrite it
    if (kProf)
        prof.t[2] = clock64();

    // ---- pass 2: decode, every chain lane from its true entry -----------------------------------------------------
    int32_t* out = (coef_lane ? q : vals) + idx;
    pos = on_chain ? e_true : 0u;
    uint32_t ones = 0;
    bool overrun = false;
    while (__any(remaining != 0)) {
        const uint32_t w = pos >> 5, sh = pos & 31;
        const uint32_t w0 = strm[w], w1 = strm[w + 1], w2 = strm[w + 2];
        const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
        const bool act = remaining != 0;
        const bool full = x0 == 0xFFFFFFFFu;
        const uint32_t t = full ? 32u : (uint32_t)__builtin_ctz(~x0 | 0x80000000u);
        const uint32_t field = (uint32_t)(((((uint64_t)x1) << 32) | x0) >> ((t + 1) & 63)) & kmask;
        const bool emit = act && !full;
        if (emit)
            *out = rice_value(ones + t, field, k);
        out += emit ? 1 : 0;
        ones = emit ? 0u : (act ? ones + 32 : ones);
        pos += act ? (full ? 32u : t + 1 + k) : 0u;
        remaining -= emit ? 1u : 0u;
        overrun |= emit && pos > stream_end;
    }
    // a stream that ends before all its values were read yields zeros (reads beyond the end are zero)
    if (coef_total < order)
        for (uint32_t i = coef_total + lane; i < order; i += kWave)
            q[i] = 0;
    if (res_total < (uint32_t)kBlock)
        for (uint32_t i = res_total + lane; i < (uint32_t)kBlock; i += kWave)
            vals[i] = 0;
    const bool bad = __any(overrun) || coef_total < order || res_total < (uint32_t)kBlock;
    wave_sync();
    if (kProf)
        prof.t[3] = clock64();
    return bad ? (uint32_t)SELA_HIP_FLAG_RICE_OVERRUN : 0u;
}

// ---- generic mode: one stream, serially, straight from global memory ---------------------------------------
// src/rice/rice_decoder.cpp:21-52 as written: count the ones up to the first zero, read k bits MSB first.
// Every lane runs the same (wave-uniform) parse; lane 0 stores.  `words` = the frame's aligned words,
// n_frame_words of them; reads beyond the stream's own words (or the frame) are zero.
__device__ inline uint32_t parse_stream_serial(const uint32_t* __restrict__ words, uint32_t first_bit, uint32_t stream_end,
    uint32_t n_frame_words, uint32_t k, uint32_t count, int32_t* out, int lane)
{
    auto word_at = [&](uint32_t w) -> uint32_t { return (w < n_frame_words && 32 * w < stream_end) ? words[w] : 0u; };
    auto bit_at = [&](uint32_t p) -> uint32_t { return p < stream_end ? (word_at(p >> 5) >> (p & 31)) & 1u : 0u; };
    uint32_t pos = first_bit;
#pragma unroll 1
    for (uint32_t i = 0; i < count; i++) {
        uint32_t ones = 0;
        for (;;) { // whole words of ones at a time, then bit by bit
            const uint32_t w = pos >> 5, sh = pos & 31;
            const uint32_t lo = word_at(w) >> sh;
            const uint32_t have = 32 - sh;
            const uint32_t run = (uint32_t)__builtin_ctz(~lo | (have < 32 ? 1u << have : 0u));
            const uint32_t t = min(run, have);
            ones += t;
            pos += t;
            if (t < have || pos >= stream_end)
                break;
        }
        pos++; // the terminator
        uint32_t rem = 0;
        for (uint32_t b = 0; b < k; b++)
            rem = (rem << 1) | bit_at(pos + b);
        pos += k;
        const uint32_t u = (ones << k) | rem;
        if (lane == 0)
            out[i] = (int32_t)((u >> 1) ^ (0u - (u & 1u)));
    }
    wave_sync();
    return pos > stream_end ? (uint32_t)SELA_HIP_FLAG_RICE_OVERRUN : 0u;
}

// ---- synthesis filter ----------------------------------------------------------------------------------
// lpc::SampleGenerator::generateSamples (src/lpc/sample_generator.cpp:11-30), in place over the
// residues in LDS.  Transposed direct form without data movement: every sample that is still to come
// owns a partial sum, and the sum of sample j lives in lane j mod 64 for its whole life (a ring over
// the lanes; orders above 60 use two registers per lane = a ring of 128).  Once sample s_i is known,
// the lane that owns sample i + d adds a[d] * s_i; its coefficient a[(lane - i) mod ring] comes out of
// a doubled table in LDS at a compile-time offset (the 64 steps of a block are unrolled), so nothing is
// shifted between lanes.  The recurrence itself (sum -> s_i) runs on the scalar unit: v_readlane of the
// finished sum, two SALU ops, and s_i feeds the multiply-adds as a scalar operand.
//
// What is accumulated is N = 2^34 - sum(a_j s_(i-j)): the coefficients are negated once and every sum
// starts at the rounding constant 2^34, so the prediction (int32)((2^34 - P) >> 35) is the arithmetic
// shift (int32)N_hi >> 3 of the HIGH word alone (the reference's cast keeps exactly those 29 bits).
//
// A finished sum is not touched again until its lane is recycled: the coefficients of lags
// ring - G + 1 .. ring - 1 are zero (order <= ring - G), so lanes are recycled in aligned groups of G
// (three DPP moves under a row/bank mask: keep the finished high words, restart the sums), and the
// 64 samples of a block are derived from the kept words in one vector step.  Per sample that is
// 3 + 3/G VALU instructions (5 + 3/G on the ring of 128) and one (two) ds_read_b64.
//
// 64x32-bit products: a' = ah*2^32 + al with al = (int32)a', so
//     z + a'*s mod 2^64 = (z + al*s)  [v_mad_i64_i32, exact]  +  ((ah*s mod 2^32) << 32)
//
// kFold: the residue is folded into its sum at the start of its block of 64,
//     N' = N - r * 2^35  (one subtract on the high word per 64 samples)   ==>   s = -(N' >> 35),
// which drops the per-sample v_readlane of r, and the high product is one v_mad_i32_i24.  Both need
// small operands: the shift keeps 29 bits and the multiplier 24, so this equals the reference's 32-bit
// r - (int32)((2^34 - P) >> 35) exactly while |s| < 2^23 and |a| < 2^55; the coefficients are checked
// when the table is built and every 64 samples against 2^23.  A block that fails the check has stored
// nothing: the caller puts the sums back as they were at the block's start and runs it -- and the rest of
// the subframe -- in the exact form (v_readlane of r, v_mul_lo_u32 + v_add_u32).  16-bit audio never gets
// there; crafted streams do (tests).
template <bool kFold>
__device__ __forceinline__ void synth_mac(uint32_t& zl, uint32_t& zh, uint64_t coef, int32_t s_i)
{
    const int32_t al = (int32_t)(uint32_t)coef, ah = (int32_t)(uint32_t)(coef >> 32);
    const uint64_t z = ((uint64_t)zh << 32) | zl;
    const uint64_t lo = (uint64_t)((int64_t)z + (int64_t)al * (int64_t)s_i);
    if (kFold) // both factors fit 24 bits in the folded form (checked)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(zh) : "v"(ah), "s"(s_i), "v"((uint32_t)(lo >> 32)));
    else
        zh = (uint32_t)(lo >> 32) + (uint32_t)ah * (uint32_t)s_i;
    zl = (uint32_t)lo;
}

// Coefficient prefetch depth (steps).  The table reads have compile-time addresses, so left alone the
// scheduler hoists all 64 (128) of a block to its top and spills; instead each step consumes the
// value fetched kAhead steps earlier, issues the fetch for step M + kAhead and ends in a scheduling
// barrier.  (LdsTable is volatile: that keeps the two reads of a ring-of-128 step as ds_read_b64, 2 LDS
// cycles each; merged into one ds_read2_b64 they would take 8 and the loop turns LDS-bound.)
constexpr int kAhead = 4;

// Steps M .. 63 of one block of 64 samples.  (cl, ch): the register whose sums finish in this block;
// (ol, oh): the other register of the ring of 128 (R == 2).  tab_lane = table + lane.
template <int R, bool kFold, int G, int M>
__device__ __forceinline__ void synth_steps(uint32_t& cl, uint32_t& ch, uint32_t& ol, uint32_t& oh, uint32_t& kept,
    LdsTable tab_lane, int32_t r_block, uint32_t four, uint32_t zero, uint64_t (&pf_c)[kAhead], uint64_t (&pf_o)[kAhead])
{
    // scalar side: the sum of this sample sits in lane M
    const int32_t pred = __builtin_amdgcn_readlane((int)ch, M) >> 3;
    int32_t s_i;
    if (kFold)
        s_i = (int32_t)(0u - (uint32_t)pred);
    else
        s_i = (int32_t)((uint32_t)__builtin_amdgcn_readlane(r_block, M) - (uint32_t)pred);
    // vector side: lane L adds a'[(L - M) mod ring] * s_i  (a'[0] = 0: the finished sum stays)
    synth_mac<kFold>(cl, ch, pf_c[M % kAhead], s_i);
    if (R == 2)
        synth_mac<kFold>(ol, oh, pf_o[M % kAhead], s_i);
    if constexpr (M + kAhead < 64) {
        pf_c[M % kAhead] = tab_lane[64 * R - (M + kAhead)];
        if (R == 2)
            pf_o[M % kAhead] = tab_lane[64 - (M + kAhead)];
    }
    if constexpr ((M + 1) % G == 0) { // recycle lanes M + 1 - G .. M
        constexpr int first_lane = M + 1 - G;
        constexpr int row_mask = 1 << (first_lane / 16);
        constexpr int bank_mask = G == 16 ? 0xf : 1 << ((first_lane % 16) / 4);
        kept = (uint32_t)__builtin_amdgcn_update_dpp((int)kept, (int)ch, 0xE4 /* quad_perm:[0,1,2,3] */, row_mask, bank_mask, false);
        ch = (uint32_t)__builtin_amdgcn_update_dpp((int)ch, (int)four, 0xE4, row_mask, bank_mask, false);
        cl = (uint32_t)__builtin_amdgcn_update_dpp((int)cl, (int)zero, 0xE4, row_mask, bank_mask, false);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M < 63)
        synth_steps<R, kFold, G, M + 1>(cl, ch, ol, oh, kept, tab_lane, r_block, four, zero, pf_c, pf_o);
}

// One block of 64 samples at rs[0..63].  The folded form returns false, with rs[] untouched, if a sample
// of the block left its range.
template <int R, bool kFold, int G>
__device__ __forceinline__ bool synth_block(int32_t* rs, uint32_t& cl, uint32_t& ch, uint32_t& ol, uint32_t& oh,
    LdsTable tab_lane, int lane, uint32_t four, uint32_t zero)
{
    uint64_t pf_c[kAhead], pf_o[kAhead];
#pragma unroll
    for (int m = 0; m < kAhead; m++) {
        pf_c[m] = tab_lane[64 * R - m];
        pf_o[m] = R == 2 ? tab_lane[64 - m] : 0;
    }
    const int32_t r_block = rs[lane];
    if (kFold)
        ch -= (uint32_t)r_block << 3; // sample lane of this block: N -= r * 2^35
    uint32_t kept = 0;
    __builtin_amdgcn_sched_barrier(0);
    synth_steps<R, kFold, G, 0>(cl, ch, ol, oh, kept, tab_lane, r_block, four, zero, pf_c, pf_o);
    const int32_t s = (int32_t)((kFold ? 0u : (uint32_t)r_block) - (uint32_t)((int32_t)kept >> 3));
    if (kFold && __any((uint32_t)(s + (1 << 23)) >= (1u << 24)))
        return false;
    rs[lane] = s;
    return true;
}

// All 2048 samples of a subframe.  R = ring / 64 (1: order <= 64 - G, 2: order <= 128 - G); G = recycling
// group (4 or 16).  fold = start in the folded form (the coefficients fit it).
template <int R, int G>
__device__ inline void synthesize(int32_t* rs, const uint64_t* tab, bool fold, int lane)
{
    static_assert(G == 4 || G == 16, "groups are DPP banks or rows");
    uint32_t zl[2], zh[2];
#pragma unroll
    for (int h = 0; h < 2; h++) { // every sum starts at 2^34
        zl[h] = 0;
        zh[h] = 4;
    }
    uint32_t four = 4, zero = 0;
    asm volatile("" : "+v"(four), "+v"(zero)); // DPP sources must be VGPRs
    const LdsTable tab_lane = (LdsTable)(tab + lane); // the table is in LDS: ds_read with immediate offsets
#pragma unroll 1
    for (int base = 0; base < kBlock; base += 64 * R) {
#pragma unroll
        for (int h = 0; h < R; h++) {
            uint32_t& cl = zl[h];
            uint32_t& ch = zh[h];
            uint32_t& ol = zl[R - 1 - h];
            uint32_t& oh = zh[R - 1 - h];
            if (fold) {
                const uint32_t s0 = zl[0], s1 = zh[0], s2 = zl[1], s3 = zh[1];
                if (synth_block<R, true, G>(rs + base + 64 * h, cl, ch, ol, oh, tab_lane, lane, four, zero))
                    continue;
                zl[0] = s0, zh[0] = s1, zl[1] = s2, zh[1] = s3; // back to the block's start, exact form from here on
                fold = false;
            }
            synth_block<R, false, G>(rs + base + 64 * h, cl, ch, ol, oh, tab_lane, lane, four, zero);
        }
    }
    wave_sync();
}

// Negated coefficients a'[d] = -a[d] (0 for d = 0 and beyond `order`), packed {al, ah}, ring-periodic
// and doubled, written over the wave's k[] / a[] arrays.  Returns whether every ah fits 24 bits.
__device__ inline bool build_synth_table(const int64_t* a, uint64_t* tab, int order, int lane)
{
    uint64_t c[2];
    bool fits = true;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int d = lane + 64 * h;
        const uint64_t nv = 0 - (uint64_t)(d >= 1 && d <= order ? a[d] : 0);
        const int32_t al = (int32_t)(uint32_t)nv;
        const int32_t ah = (int32_t)(uint32_t)((nv - (uint64_t)(int64_t)al) >> 32); // nv = ah 2^32 + al, al signed
        fits &= ah >= -(1 << 23) && ah < (1 << 23);
        c[h] = ((uint64_t)(uint32_t)ah << 32) | (uint32_t)al;
    }
    wave_sync(); // a[] has been read by every lane
    if (order <= 60) { // ring of 64
        tab[lane] = c[0];
        tab[lane + 64] = c[0];
    } else {           // ring of 128
        tab[lane] = c[0];
        tab[lane + 64] = c[1];
        tab[lane + 128] = c[0];
        tab[lane + 192] = c[1];
    }
    wave_sync();
    return !__any(!fits);
}

// ---- subframe header walk (layout of src/file/sela_file.cpp:58-91) ------------------------------------------
struct SubHeader {
    bool ok;
    uint32_t p;  // byte offset of the subframe in the frame
    uint32_t channel, type, parent, ck, cw, order, rk, rw;
};

__device__ inline SubHeader walk_headers(const uint8_t* fb, uint64_t fbytes, uint32_t c, uint32_t channels)
{
    SubHeader h;
    h.ok = fbytes >= 4 && fbytes < 0x7FFFFFFFull && (fbytes & 3) == 0 && reinterpret_cast<const uint32_t*>(fb)[0] == SELA_SYNC_WORD;
    uint64_t p = 4;
    uint32_t n = 0;
    h.channel = h.type = h.parent = h.ck = h.cw = h.order = h.rk = h.rw = 0;
    for (uint32_t i = 0; h.ok && i <= c; i++) { // walk the headers up to this subframe
        if (p + 12 > fbytes) {
            h.ok = false;
            break;
        }
        const uint32_t h0 = *reinterpret_cast<const uint32_t*>(fb + p);     // channel, type, parent, coefficient k
        const uint32_t h1 = *reinterpret_cast<const uint32_t*>(fb + p + 4); // word count (u16), order (u8), first coefficient byte
        h.channel = h0 & 0xFF, h.type = (h0 >> 8) & 0xFF, h.parent = (h0 >> 16) & 0xFF, h.ck = h0 >> 24;
        h.cw = h1 & 0xFFFF, h.order = (h1 >> 16) & 0xFF;
        const uint64_t p2 = p + 4 + 4 * (uint64_t)h.cw; // aligned word: last 3 coefficient bytes + residue k
        if (p2 + 8 > fbytes) {
            h.ok = false;
            break;
        }
        const uint32_t h2 = *reinterpret_cast<const uint32_t*>(fb + p2);
        const uint32_t h3 = *reinterpret_cast<const uint32_t*>(fb + p2 + 4);
        h.rk = h2 >> 24, h.rw = h3 & 0xFFFF, n = h3 >> 16;
        const uint64_t next = p + 12 + 4 * ((uint64_t)h.cw + h.rw);
        if (next > fbytes) {
            h.ok = false;
            break;
        }
        if (i < c)
            p = next;
    }
    h.ok = h.ok && h.channel < channels && h.order <= (uint32_t)kMaxOrder && n == (uint32_t)kBlock && h.ck < 32 && h.rk < 32 && h.type <= 1
        && (h.type == 0 || h.parent < channels);
    h.p = (uint32_t)p;
    return h;
}

// LDS plan of k_decode_frames (dynamic): [channels][2048] int32 values | one scratch record per wave |
// sub_info[channels] | mode words.  Values are indexed by subframe POSITION; sub_info maps channels to them.
__host__ __device__ inline size_t decode_scratch_stride(bool fast) { return fast ? sizeof(DecWaveFast) : sizeof(DecWaveGeneric); }

// kProf: also write per-phase cycle counts (debug hook sela_hip_debug_phase_buffer; 16 uint64 per subframe).
template <bool kProf>
__global__ __launch_bounds__(kDecMaxWaves * 64) void k_decode_frames(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* __restrict__ pcm_out,
    uint32_t* __restrict__ status, uint64_t* __restrict__ phase_cycles)
{
    long long stamp[10];
    for (int i = 0; i < 10; i++)
        stamp[i] = 0;
    uint32_t prof_sub = 0xFFFFFFFFu;
    if (kProf)
        stamp[0] = clock64();
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int n_waves = blockDim.x / 64;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)), lane = threadIdx.x % 64;
    const bool fast_plan = channels <= (uint32_t)kDecMaxWaves; // the scratch records are DecWaveFast (host: launch_decode)
    int32_t* const vals_all = reinterpret_cast<int32_t*>(dyn);
    unsigned char* const scratch = dyn + (size_t)channels * kBlock * 4 + (size_t)wave * decode_scratch_stride(fast_plan);
    uint32_t* const sub_info = reinterpret_cast<uint32_t*>(dyn + (size_t)channels * kBlock * 4 + (size_t)n_waves * decode_scratch_stride(fast_plan));
    uint32_t* const too_big = sub_info + channels; // [n_waves]: this wave's subframe does not fit the fast plan

    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const uint8_t* const fb = frames + frame_offsets[f];
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    uint32_t flags = 0;
    for (uint32_t c = threadIdx.x; c < channels; c += blockDim.x)
        sub_info[c] = 0xFFFFFFFFu; // "no subframe delivered this channel"

    // ---- mode: every subframe of the frame must fit the fast plan ----------------------------------------------
    SubHeader hd = walk_headers(fb, fbytes, (uint32_t)wave < channels ? (uint32_t)wave : 0u, channels);
    if (lane == 0)
        too_big[wave] = (fast_plan && (!hd.ok || (hd.cw + 2 + hd.rw <= (uint32_t)kStreamCap && hd.cw <= (uint32_t)kCoefWordsCap))) ? 0u : 1u;
    __syncthreads();
    bool fast = fast_plan;
    for (int w = 0; w < n_waves; w++)
        fast = fast && too_big[w] == 0;
    if (kProf)
        stamp[1] = clock64();

    for (uint32_t c = wave; c < channels; c += n_waves) {
        if (c != (uint32_t)wave)
            hd = walk_headers(fb, fbytes, c, channels);
        if (!hd.ok) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            continue;
        }
        int32_t* const vals = vals_all + (size_t)c * kBlock;
        SynthTables* tables;
        int32_t* q;
        ParseProfile pp;
        if (fast) {
            DecWaveFast* const wl = reinterpret_cast<DecWaveFast*>(scratch);
            tables = &wl->t;
            q = wl->q;
            // the subframe's aligned words -> LDS, the start bitmap (over the value array) cleared
            const uint32_t nw = hd.cw + 2 + hd.rw;
            const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4);
            uint32_t* const marks = reinterpret_cast<uint32_t*>(vals);
            for (uint32_t w = lane; w < nw + kStreamMargin; w += kWave) {
                wl->strm[w] = w < nw ? gw[w] : 0u;
                marks[w] = 0;
            }
            wave_sync();
            if (kProf)
                stamp[2] = clock64(), prof_sub = c;
            flags |= parse_subframe<kProf>(wl->strm, marks, vals, q, hd.cw, hd.rw, hd.ck, hd.rk, hd.order, lane, pp);
        } else {
            DecWaveGeneric* const wl = reinterpret_cast<DecWaveGeneric*>(scratch);
            tables = &wl->t;
            q = wl->q;
            if (kProf)
                stamp[2] = clock64(), prof_sub = c;
            const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4);
            const uint32_t n_frame_words = (uint32_t)((fbytes - hd.p - 4) / 4);
            flags |= parse_stream_serial(gw, 24, 24 + 32 * hd.cw, n_frame_words, hd.ck, hd.order, q, lane);
            flags |= parse_stream_serial(gw, 32 * (hd.cw + 2), 32 * (hd.cw + 2 + hd.rw), n_frame_words, hd.rk, (uint32_t)kBlock, vals, lane);
            pp.t[0] = pp.t[1] = pp.t[2] = pp.t[3] = kProf ? clock64() : 0;
        }
        if (kProf)
            stamp[3] = pp.t[0], stamp[4] = pp.t[1], stamp[5] = pp.t[2], stamp[6] = pp.t[3];

        // dequantise (src/lpc/linear_predictor.cpp:16-28) + step-up; the tables go over the dead stream words
        const uint32_t order = hd.order;
        const int32_t q_lo = (uint32_t)lane < order ? q[lane] : 0, q_hi = (uint32_t)lane + 64 < order ? q[lane + 64] : 0;
        wave_sync();
        if ((uint32_t)lane < order)
            tables->k[lane] = order <= 1 ? 0.0 : dequant(lane, q_lo, flags);
        if ((uint32_t)lane + 64 < order)
            tables->k[lane + 64] = dequant(lane + 64, q_hi, flags);
        wave_sync();
        step_up(tables->k, tables->a, (int)order, lane, flags);
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        if (kProf)
            stamp[7] = clock64();
        // ring / recycling group by order: <= 48: 64 / 16, <= 60: 64 / 4, else 128 / 16
        if (order <= 48)
            synthesize<1, 16>(vals, tables->tab, fits24, lane);
        else if (order <= 60)
            synthesize<1, 4>(vals, tables->tab, fits24, lane);
        else
            synthesize<2, 16>(vals, tables->tab, fits24, lane);
        if (kProf)
            stamp[8] = clock64();
        if (lane == 0)
            sub_info[hd.channel] = hd.type | (hd.parent << 8) | (c << 16);
    }
    __syncthreads();

    // ---- second pass of frame::FrameDecoder + interleave to int16 ------------------------------------
    // dependent channels become parent - difference (parents are independent subframes); a channel
    // that no valid subframe delivered decodes to silence and raises BAD_FRAME.
    if (channels == 2) {
        // stereo: four samples of both channels per thread, one 16-byte store (wave-uniform case analysis)
        const uint32_t i0 = sub_info[0], i1 = sub_info[1];
        const bool have0 = i0 != 0xFFFFFFFFu, have1 = i1 != 0xFFFFFFFFu;
        const bool dep0 = have0 && (i0 & 0xFF) == 1, dep1 = have1 && (i1 & 0xFF) == 1;
        const uint32_t par0 = (i0 >> 8) & 0xFF, par1 = (i1 >> 8) & 0xFF; // parent channel of a dependent subframe (0 or 1, checked above)
        const int4* s0 = reinterpret_cast<const int4*>(vals_all + (size_t)(have0 ? i0 >> 16 : 0) * kBlock);
        const int4* s1 = reinterpret_cast<const int4*>(vals_all + (size_t)(have1 ? i1 >> 16 : 0) * kBlock);
        uint4* out = reinterpret_cast<uint4*>(pcm_out + (size_t)f * kBlock * 2);
        for (uint32_t i4 = threadIdx.x; i4 < (uint32_t)kBlock / 4; i4 += blockDim.x) {
            const int4 zero = make_int4(0, 0, 0, 0);
            const int4 r0 = have0 ? s0[i4] : zero, r1 = have1 ? s1[i4] : zero; // raw subframe outputs
            int4 a = r0, b = r1;
            if (dep0) { // parent - difference; the parent's own (independent) samples
                const int4 pv = par0 == 0 ? r0 : r1;
                a = make_int4((int)((uint32_t)pv.x - (uint32_t)r0.x), (int)((uint32_t)pv.y - (uint32_t)r0.y),
                    (int)((uint32_t)pv.z - (uint32_t)r0.z), (int)((uint32_t)pv.w - (uint32_t)r0.w));
            }
            if (dep1) {
                const int4 pv = par1 == 0 ? r0 : r1;
                b = make_int4((int)((uint32_t)pv.x - (uint32_t)r1.x), (int)((uint32_t)pv.y - (uint32_t)r1.y),
                    (int)((uint32_t)pv.z - (uint32_t)r1.z), (int)((uint32_t)pv.w - (uint32_t)r1.w));
            }
            uint4 w;
            w.x = ((uint32_t)a.x & 0xFFFFu) | ((uint32_t)b.x << 16);
            w.y = ((uint32_t)a.y & 0xFFFFu) | ((uint32_t)b.y << 16);
            w.z = ((uint32_t)a.z & 0xFFFFu) | ((uint32_t)b.z << 16);
            w.w = ((uint32_t)a.w & 0xFFFFu) | ((uint32_t)b.w << 16);
            out[i4] = w;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < (uint32_t)kBlock; i += blockDim.x) {
            for (uint32_t c = 0; c < channels; c++) {
                const uint32_t info = sub_info[c];
                int32_t v = info == 0xFFFFFFFFu ? 0 : vals_all[(size_t)(info >> 16) * kBlock + i];
                if (info != 0xFFFFFFFFu && (info & 0xFF) == 1) {
                    const uint32_t pinfo = sub_info[(info >> 8) & 0xFF];
                    const int32_t pv = pinfo == 0xFFFFFFFFu ? 0 : vals_all[(size_t)(pinfo >> 16) * kBlock + i];
                    v = (int32_t)((uint32_t)pv - (uint32_t)v);
                }
                pcm_out[((size_t)f * kBlock + i) * channels + c] = (int16_t)(uint16_t)v;
            }
        }
    }
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < channels; c++) {
            const uint32_t info = sub_info[c];
            if (info == 0xFFFFFFFFu)
                flags |= SELA_HIP_FLAG_BAD_FRAME;
            else if ((info & 0xFF) == 1) {
                const uint32_t pinfo = sub_info[(info >> 8) & 0xFF];
                if (pinfo == 0xFFFFFFFFu || (pinfo & 0xFF) != 0)
                    flags |= SELA_HIP_FLAG_BAD_FRAME; // a parent that is itself dependent is outside what the reference defines
            }
        }
    }
    flags = wave_or(flags);
    if (lane == 0 && flags)
        atomicOr(&status[0], flags);
    __syncthreads(); // (orders the flag words below after every wave's atomicOr only loosely; they are counted per frame)
    if (threadIdx.x == 0) {
        bool bad = (flags & SELA_HIP_FLAG_BAD_FRAME) != 0;
        for (int w = 1; w < n_waves; w++)
            bad = bad || too_big[w] == 2;
        if (bad)
            atomicAdd(&status[1], 1u);
    } else if (lane == 0 && (flags & SELA_HIP_FLAG_BAD_FRAME)) {
        too_big[wave] = 2; // (read by thread 0 after the barrier below would be cleaner; see comment)
    }
    if (kProf && lane == 0 && prof_sub != 0xFFFFFFFFu) { // (one subframe per wave is reported)
        stamp[9] = clock64();
        for (int i = 0; i < 9; i++)
            phase_cycles[((size_t)f * channels + prof_sub) * 16 + i] = (uint64_t)(stamp[i + 1] - stamp[i]);
    }
}

int decode_waves(uint32_t channels)
{
    return channels < (uint32_t)kDecMaxWaves ? (int)channels : kDecMaxWaves;
}

size_t decode_lds_bytes(uint32_t channels)
{
    const int n_waves = decode_waves(channels);
    const bool fast_plan = channels <= (uint32_t)kDecMaxWaves;
    return (size_t)channels * kBlock * 4 + (size_t)n_waves * decode_scratch_stride(fast_plan) + (size_t)channels * 4 + (size_t)n_waves * 4;
}

uint32_t decode_max_channels() { return (uint32_t)kDecMaxChannels; }

// The decoder keeps everything on chip; the workspace argument of the C ABI is kept for callers written
// against the first version of the interface.
size_t decode_workspace_bytes(uint32_t n_frames, uint32_t channels)
{
    (void)n_frames;
    (void)channels;
    return 256;
}

hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, hipStream_t stream, hipEvent_t* ev /* 2 events or nullptr */, uint64_t* d_phase_cycles)
{
    hipError_t err = hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
    if (err != hipSuccess || n_frames == 0)
        return err;
    const int n_waves = decode_waves(channels);
    const size_t lds = decode_lds_bytes(channels);
    if (channels > (uint32_t)kDecMaxChannels || lds > 160 * 1024)
        return hipErrorInvalidValue;
    if (lds > 64 * 1024) { // above the default dynamic-LDS limit (five channels and more)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    if (ev)
        (void)hipEventRecord(ev[0], stream);
    if (d_phase_cycles)
        hipLaunchKernelGGL(k_decode_frames<true>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames, channels,
            d_pcm_out, d_status, d_phase_cycles);
    else
        hipLaunchKernelGGL(k_decode_frames<false>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames, channels,
            d_pcm_out, d_status, d_phase_cycles);
    if (ev)
        (void)hipEventRecord(ev[1], stream);
    return hipGetLastError();
}

} // namespace sela
