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
// coalesced.
//
// The synthesis recurrence is a dependent chain of ~90 cycles per sample, so the kernel lives on occupancy:
// what a subframe keeps in LDS is cut to 5.7 KB (28 waves per CU).  The Rice stream is NOT copied on chip --
// it is read where it lies through a bounds-checked buffer resource (reads past the subframe's words
// return zero) -- the parser leaves one 16-bit POSITION per codeword instead of a 32-bit value, the
// residues of a block of 64 samples are decoded from those positions just in time (the loads are in
// flight during the previous block's 64 steps), and the finished samples go over the positions as int16
// (the WAV writer truncates to 16 bits anyway, and (parent - difference) mod 2^16 only needs 16 bits).
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
//            and list nothing.  A lane on the chain counts its codewords from its true entry (popcount
//            of its bitmap words + what it parsed in phase B); an exclusive scan gives the index of its
//            first value;
//   pass 2   the lanes on the chain walk their codewords once more from their true entries, in parallel, and
//            write every start position to pos[] (over the bitmap, which is dead by then).
//
// All three walks take four codewords per memory round trip (a 160-bit window per lane) while every lane's
// codewords are short, and one at a time otherwise (unary runs longer than the window, k = 31).  The first
// kCoefLanes lanes do the same for the coefficient stream (<= 100 codewords, its own k), in the same loops:
// both streams are zones of ONE bit space, the subframe's aligned words.  tools/parse_model.py is an
// executable model of this algorithm; tests/test_host_logic.py runs it against the CPU oracle.
//
// A frame with a subframe that does not fit the LDS plan (a Rice stream longer than any 16-bit audio produces)
// takes the GENERIC mode of the same kernel: a plain serial parse into the workspace, then the same synthesis.
// Slow, complete, and never needed by files the encoder writes.  Frames of more than 8 channels (up to the 255 the
// header's field carries) are decoded by k_decode_frames_wide, eight subframes at a time.
#include <atomic>

#include "sela_device.h"

namespace sela {

constexpr int kDecMaxWaves = 8;     // waves per workgroup; frames with more channels take k_decode_frames_wide
constexpr int kDecMaxChannels = 255; // what the 8-bit channel field of the .sela header can say (src/file/sela_file.cpp:40)
constexpr int kCoefLanes = 4;       // lanes of a wave that parse the coefficient stream
constexpr int kResLanes = kWave - kCoefLanes;
// Aligned words of one subframe the segment-parallel parser takes: coefficient words + 2 + residue words
// (start bitmap: one bit per stream bit; positions must fit 16 bits).
constexpr int kStreamCap = 1072;
constexpr int kStreamMargin = 4;    // a window may run this many words past the end (they read as zero)
constexpr uint32_t kEndOfStream = 0xFFFFFFFFu;

typedef const volatile __attribute__((address_space(3))) uint64_t* LdsTable;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- per-wave LDS scratch --------------------------------------------------------------------------------
struct SynthTables {
    union {
        int64_t a[104];     // Q35 predictor (the reflection coefficients never touch LDS here: dequantised into registers)
        uint64_t tab[192];  // synthesis coefficient table (build_synth_table), replaces a[]: entries 1 .. 191 are read
    };
};
struct DecWaveScratch {
    SynthTables t;
    // (the parsed coefficient values q[0 .. order) live inside t, see coef_values(): between the parse, whose scratch lies
    // in front of them, and the dequantisation, which reads them into registers before a[] is written)
};
constexpr int kCoefValuesAt = 896; // byte offset in SynthTables: behind the parse's positions (0..256), chain flags (512..577) and entries (640..896)
static_assert(kCoefValuesAt + 104 * 4 <= (int)sizeof(SynthTables) && kCoefValuesAt >= 104 * 8, "the coefficient values lie behind a[] inside the table's space");
__device__ __forceinline__ int32_t* coef_values(DecWaveScratch* s)
{
    return reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(&s->t) + kCoefValuesAt);
}
// per subframe POSITION (not per wave: the combine pass reads every channel)
union DecSubframeLds {
    uint32_t marks[kStreamCap + kStreamMargin]; // start bitmap (parse)
    uint16_t pos[kBlock];                       // bit position of every residue codeword (after the parse)
    int16_t smp[kBlock];                        // finished samples, written over the positions block by block
};
static_assert(sizeof(DecSubframeLds) == (kStreamCap + kStreamMargin) * 4 && sizeof(DecSubframeLds) % 16 == 0, "LDS plan");
static_assert(32 * (kStreamCap + kStreamMargin) <= 65536, "positions are 16-bit");

__device__ __forceinline__ int32_t rice_value(uint32_t ones, uint32_t field, uint32_t k)
{
    const uint32_t rem = k ? (__brev(field) >> (32 - k)) : 0u; // remainder is MSB first in the stream
    const uint32_t u = (ones << k) | rem;                      // uint32 arithmetic as src/rice/rice_decoder.cpp:35
    return (int32_t)((u >> 1) ^ (0u - (u & 1u)));               // un-zig-zag, src/rice/rice_decoder.cpp:49-50
}

// ---- the subframe's aligned words, where they lie ------------------------------------------------------------
// A raw-dword buffer resource over [words, words + n_words): reads beyond return 0 (hardware bounds check),
// which is exactly the zero padding the parser wants behind a stream.  Rebuilt from scalars at every use site
// that a function call separates from its creation (a resource that travelled through arguments is no longer
// known to be wave-uniform).
struct StreamWords {
    const uint32_t* words;
    uint32_t n_words;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_rsrc(const StreamWords& s)
{
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(read_first_lane(reinterpret_cast<uint64_t>(s.words))), 0,
        4u * (uint32_t)__builtin_amdgcn_readfirstlane((int)s.n_words), 0x00020000);
}
// 64 stream bits at bit position p
__device__ __forceinline__ void window64(__amdgpu_buffer_rsrc_t rs, uint32_t p, uint32_t& x0, uint32_t& x1)
{
    const uint32_t b = 4 * (p >> 5), sh = p & 31;
    const uint32_t w0 = __builtin_amdgcn_raw_buffer_load_b32(rs, b, 0, 0), w1 = __builtin_amdgcn_raw_buffer_load_b32(rs, b + 4, 0, 0),
                   w2 = __builtin_amdgcn_raw_buffer_load_b32(rs, b + 8, 0, 0);
    x0 = __builtin_amdgcn_alignbit(w1, w0, sh);
    x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
}

// Four codewords at bit position p, provided they are all short: off[j] = offset of codeword j's start,
// off[4] = offset behind the fourth.  simple = every run length < 31 and every codeword <= 31 bits (the
// funnel shifts below take their amounts mod 32, and the 160-bit window then always covers the next start).
__device__ __forceinline__ void analyse4(__amdgpu_buffer_rsrc_t rs, uint32_t p, uint32_t k, uint32_t (&off)[5], bool& simple)
{
    const uint32_t b = 4 * (p >> 5), sh = p & 31;
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, b, 0, 0);
    const uint32_t w4 = __builtin_amdgcn_raw_buffer_load_b32(rs, b + 16, 0, 0);
    uint32_t n0 = __builtin_amdgcn_alignbit(w.y, w.x, sh), n1 = __builtin_amdgcn_alignbit(w.z, w.y, sh);
    uint32_t n2 = __builtin_amdgcn_alignbit(w.w, w.z, sh), n3 = __builtin_amdgcn_alignbit(w4, w.w, sh);
    uint32_t used = 0;
    simple = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t t = (uint32_t)__builtin_ctz(~n0 | 0x80000000u); // <= 31
        const uint32_t len = t + 1 + k;
        simple = simple && len <= 31;
        off[j] = used;
        used += len;
        if (j < 3) { // (a codeword that is not simple garbles the rest of a group that is then not used)
            n0 = __builtin_amdgcn_alignbit(n1, n0, len);
            n1 = __builtin_amdgcn_alignbit(n2, n1, len);
            if (j < 2)
                n2 = __builtin_amdgcn_alignbit(n3, n2, len);
            if (j < 1)
                n3 >>= len & 31;
        }
    }
    off[4] = used;
}

// One step of the careful walk: over (part of) one codeword at p.  A run of 32 ones and more is taken 32
// bits at a time (in_run).
__device__ __forceinline__ void single_step(__amdgpu_buffer_rsrc_t rs, uint32_t p, uint32_t k, uint32_t& adv, bool& full)
{
    const uint32_t b = 4 * (p >> 5), sh = p & 31;
    const uint32_t x = __builtin_amdgcn_alignbit(__builtin_amdgcn_raw_buffer_load_b32(rs, b + 4, 0, 0), __builtin_amdgcn_raw_buffer_load_b32(rs, b, 0, 0), sh);
    full = x == 0xFFFFFFFFu;
    adv = full ? 32u : (uint32_t)__builtin_ctz(~x | 0x80000000u) + 1 + k;
}

// ---- segment-parallel parse of one subframe (fast mode) ------------------------------------------------
// Bit space: stream bit t of the subframe's aligned words = bit t % 32 of word t / 32.  Coefficient stream =
// bits [24, 24 + 32 cw), residue stream = bits [32 (cw + 2), 32 (cw + 2 + rw)).  marks[] and pos_out[] are the
// same LDS bytes (the bitmap is dead before the first position is stored).  Outputs: pos_out[0 .. 2048) =
// start of every residue codeword, q[0 .. order) = the coefficients (decoded here: there are few);
// returns SELA_HIP_FLAG_RICE_OVERRUN or 0.  cpos: 128 uint16 of scratch.
struct ParseProfile {
    long long t[4];
};

// The value of the codeword at bit position p (runs of 32 ones and more are followed word by word).
__device__ __forceinline__ int32_t decode_at(__amdgpu_buffer_rsrc_t rs, uint32_t p, uint32_t k, uint32_t kmask, bool valid)
{
    uint32_t x0, x1, ones = 0;
    window64(rs, p, x0, x1);
    while (__any(valid && x0 == 0xFFFFFFFFu)) {
        const bool more = valid && x0 == 0xFFFFFFFFu;
        ones += more ? 32u : 0u;
        p += more ? 32u : 0u;
        window64(rs, p, x0, x1);
    }
    const uint32_t t = (uint32_t)__builtin_ctz(~x0 | 0x80000000u);
    const uint32_t field = (uint32_t)(((((uint64_t)x1) << 32) | x0) >> (t + 1)) & kmask;
    return rice_value(ones + t, field, k);
}

template <bool kProf>
__device__ __forceinline__ uint32_t parse_subframe(const StreamWords& sw, uint32_t* marks, uint16_t* pos_out, uint16_t* cpos, int32_t* q,
    uint32_t cw, uint32_t rw, uint32_t ck, uint32_t rk, uint32_t order, int lane, ParseProfile& prof)
{
    const __amdgpu_buffer_rsrc_t rs = stream_rsrc(sw);
    // ---- zones ---------------------------------------------------------------------------------------------
    const bool coef_lane = lane < kCoefLanes;
    const uint32_t zc = max(1u, (cw + 1 + kCoefLanes - 1) / kCoefLanes); // words per coefficient zone
    const uint32_t zr = max(1u, (rw + kResLanes - 1) / kResLanes);       // words per residue zone
    const uint32_t rs_word = cw + 2;
    const uint32_t last_mark_word = cw + 2 + rw + kStreamMargin - 1;
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

    // ---- phase A: own zone, marking every codeword start ---------------------------------------------------
    // (predicated rather than branched: lanes that are through OR a zero into the first word of the bitmap)
    uint32_t pos = entry;
    bool in_run = false; // inside a unary run longer than the window
    while (__any(pos < zone_end)) {
        const bool act = pos < zone_end;
        uint32_t off[5];
        bool simple;
        analyse4(rs, pos, k, off, simple);
        if (!__any(act && (in_run || !simple))) { // four codewords per round trip
            uint32_t adv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + off[j];
                const bool a = act && p < zone_end;
                atomicOr(&marks[a ? p >> 5 : 0u], a ? 1u << (p & 31) : 0u); // (LDS ds_or_b32; a zone's words are marked by its lane alone)
                adv = a ? off[j + 1] : adv;
            }
            pos += adv;
        } else { // one (part of a) codeword at a time
            uint32_t adv;
            bool full;
            single_step(rs, pos, k, adv, full);
            const bool start = act && !in_run;
            atomicOr(&marks[start ? pos >> 5 : 0u], start ? 1u << (pos & 31) : 0u);
            pos += act ? adv : 0u;
            in_run = act ? full : in_run;
        }
    }
    wave_sync();
    if (kProf)
        prof.t[0] = clock64();

    // ---- phase B: on through the following zones until standing on a later lane's start -----------------------
    uint32_t n_cont = 0, merged = 0;
    bool walking = true;
    while (__any(walking)) {
        uint32_t off[5];
        bool simple;
        analyse4(rs, pos, k, off, simple);
        if (!__any(walking && (in_run || !simple))) {
            uint32_t mk[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                mk[j] = marks[min((pos + off[j]) >> 5, last_mark_word)];
            uint32_t adv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + off[j];
                const bool ended = walking && p >= stream_end;
                const bool met = walking && !ended && ((mk[j] >> (p & 31)) & 1u);
                merged = ended ? kEndOfStream : (met ? p : merged);
                walking = walking && !ended && !met;
                n_cont += walking ? 1u : 0u;
                adv = walking ? off[j + 1] : adv;
            }
            pos += adv;
        } else {
            uint32_t adv;
            bool full;
            single_step(rs, pos, k, adv, full);
            const uint32_t mk = marks[min(pos >> 5, last_mark_word)];
            const bool at_start = walking && !in_run;
            const bool ended = at_start && pos >= stream_end;
            const bool met = at_start && !ended && ((mk >> (pos & 31)) & 1u);
            merged = ended ? kEndOfStream : (met ? pos : merged);
            walking = walking && !ended && !met;
            n_cont += (at_start && walking) ? 1u : 0u;
            pos += walking ? adv : 0u;
            in_run = walking ? full : in_run;
        }
    }
    if (kProf)
        prof.t[1] = clock64();

    // ---- resolve: the chains of lanes the true trajectories run through -----------------------------------------
    uint32_t succ = 64; // lane whose zone holds `merged`
    if (merged != kEndOfStream) {
        const uint32_t wm = merged >> 5;
        succ = wm < rs_word ? min(wm / zc, (uint32_t)kCoefLanes - 1) : (uint32_t)kCoefLanes + (wm - rs_word) / zr;
    }
    // The lanes a chain runs through = the orbit of its first lane under succ.  Pointer doubling: after round r
    // the set holds every lane within 2^(r+1) hops (six rounds cover the wave); a lane joins when a member's
    // pointer lands on it (a byte flag in LDS), and the pointers are squared with ds_bpermute.  (The walk
    // hop by hop on the scalar unit, 64 x readlane -> compare -> branch, cost 19 k cycles of latency.)
    if (coef_lane ? succ >= (uint32_t)kCoefLanes : false)
        succ = 64;
    succ = succ > (uint32_t)lane ? succ : 64u; // (always true of a zone further on; keeps the orbit finite whatever the stream holds)
    uint8_t* const flag = reinterpret_cast<uint8_t*>(cpos) + 512;   // 65 bytes of the scratch (the tables' space, dead until the parse is over)
    uint32_t* const entry_of = reinterpret_cast<uint32_t*>(cpos) + 160; // 64 words behind them
    flag[lane] = 0;
    if (lane == 0)
        flag[64] = 0;
    bool on_chain = lane == 0 || lane == kCoefLanes;
    uint32_t jump = succ;
    wave_sync();
#pragma unroll
    for (int r = 0; r < 6; r++) {
        if (on_chain && jump < 64)
            flag[jump] = 1;
        wave_sync();
        on_chain = on_chain || flag[lane] != 0;
        const uint32_t next = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * min(jump, 63u)), (int)jump);
        jump = jump < 64 ? next : 64u;
        wave_sync();
    }
    // a chain lane's true entry = where its predecessor merged
    if (on_chain && succ < 64)
        entry_of[succ] = merged;
    wave_sync();
    uint32_t e_true = kEndOfStream; // this lane's true entry; kEndOfStream = not on a chain
    if (on_chain)
        e_true = lane == 0 ? 24u : (lane == kCoefLanes ? 32 * rs_word : entry_of[lane]);
    wave_sync();
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
    uint32_t idx = wave_exclusive_scan(count, lane);
    const uint32_t coef_total = (uint32_t)__builtin_amdgcn_readlane((int)idx, kCoefLanes);
    const uint32_t all_total = (uint32_t)__builtin_amdgcn_readlane((int)(idx + count), kWave - 1);
    const uint32_t res_total = all_total - coef_total;
    if (!coef_lane)
        idx -= coef_total;
    uint32_t remaining = idx < need ? min(count, need - idx) : 0u;
    wave_sync(); // every lane has read the bitmap: the positions may overwrite it
    if (kProf)
        prof.t[2] = clock64();

    // ---- pass 2: list the starts, every chain lane from its true entry -----------------------------------------------
    uint16_t* out = (coef_lane ? cpos : pos_out) + idx;
    pos = on_chain ? e_true : 0u;
    in_run = false;
    bool overrun = false;
    while (__any(remaining != 0)) {
        const bool act = remaining != 0;
        uint32_t off[5];
        bool simple;
        analyse4(rs, pos, k, off, simple);
        if (!__any(act && (in_run || !simple))) {
            uint32_t adv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool a = (uint32_t)j < remaining;
                if (a)
                    out[j] = (uint16_t)(pos + off[j]);
                adv = a ? off[j + 1] : adv;
            }
            const uint32_t take = min(remaining, 4u);
            out += take;
            remaining -= take;
            pos += adv;
            overrun |= act && pos > stream_end;
        } else {
            uint32_t adv;
            bool full;
            single_step(rs, pos, k, adv, full);
            const bool start = act && !in_run;
            if (start)
                *out = (uint16_t)pos;
            out += start ? 1 : 0;
            pos += act ? adv : 0u;
            in_run = act ? full : in_run;
            remaining -= (act && !full) ? 1u : 0u;
            overrun |= act && !full && pos > stream_end;
        }
    }
    // a stream that ends before all its values were read: the missing codewords read as zero bits, i.e. they
    // "start" behind the stream's end, where the resource returns zeros
    if (res_total < (uint32_t)kBlock)
        for (uint32_t i = res_total + lane; i < (uint32_t)kBlock; i += kWave)
            pos_out[i] = (uint16_t)(32 * (rs_word + rw + 1));
    wave_sync();
    // the coefficients themselves (<= 100 values: two rounds)
    {
        const uint32_t ckmask = ck ? (0xFFFFFFFFu >> (32 - ck)) : 0u;
        for (uint32_t i0 = 0; i0 < order; i0 += kWave) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool valid = i < min(order, coef_total);
            const int32_t v = decode_at(rs, valid ? cpos[i] : 0u, ck, ckmask, valid);
            if (i < order)
                q[i] = valid ? v : 0;
        }
    }
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
        for (;;) { // up to a whole word of ones at a time
            const uint32_t sh = pos & 31, have = 32 - sh;
            const uint32_t lo = word_at(pos >> 5) >> sh;
            const uint32_t t = (uint32_t)__builtin_ctzll(~(uint64_t)lo | ((uint64_t)1 << have)); // <= have
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
// lpc::SampleGenerator::generateSamples (src/lpc/sample_generator.cpp:11-30).  Transposed direct form
// without data movement: every sample that is still to come owns a partial sum, and the sum of sample j
// lives in lane j mod 64 for its whole life (a ring over the lanes; orders above 60 use two registers per
// lane = a ring of 128).  Once sample s_i is known, the lane that owns sample i + d adds a[d] * s_i; its
// coefficient a[(lane - i) mod ring] comes out of a doubled table in LDS at a compile-time offset (the 64
// steps of a block are unrolled), so nothing is shifted between lanes.  The recurrence itself (sum -> s_i)
// runs on the scalar unit: v_readlane of the finished sum, one SALU op (two in the exact form), and the result
// feeds the multiply-adds as a scalar operand.
//
// What is accumulated is N = 2^34 - sum(a_j s_(i-j)) = 2^34 + sum(a_j (-s_(i-j))): every sum starts at the
// rounding constant 2^34 and the multiplier of a step is MINUS its sample, so the prediction
// (int32)((2^34 - P) >> 35) is the arithmetic shift (int32)N_hi >> 3 of the HIGH word alone (the reference's cast
// keeps exactly those 29 bits).
//
// A finished sum is not touched again until its lane is recycled: the coefficients of lags
// ring - G + 1 .. ring - 1 are zero (order <= ring - G), so lanes are recycled in aligned groups of G
// (three DPP moves under a row/bank mask: keep the finished high words, restart the sums), and the
// 64 samples of a block are derived from the kept words in one vector step.  Per sample that is
// 3 + 3/G VALU instructions (5 + 3/G on the ring of 128) and one (two) ds_read_b64.
//
// 64x32-bit products: a' = ah*2^32 + al with al = (int32)a', so
//     z + a*m mod 2^64 = (z + al*m)  [v_mad_i64_i32, exact]  +  ((ah*m mod 2^32) << 32)
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
// kShift: also hand back (new high word) >> 3, the next step's multiplier if the next step's sum is in this register
// (kVecShift, see synth_steps).
template <bool kFold, bool kShift>
__device__ __forceinline__ void synth_mac(uint32_t& zl, uint32_t& zh, uint64_t coef, int32_t s_i, int32_t& shifted)
{
    const int32_t al = (int32_t)(uint32_t)coef, ah = (int32_t)(uint32_t)(coef >> 32);
    const uint64_t z = ((uint64_t)zh << 32) | zl;
    const uint64_t lo = (uint64_t)((int64_t)z + (int64_t)al * (int64_t)s_i);
    if (kFold) { // both factors fit 24 bits in the folded form (checked)
        if (kShift)
            asm("v_mad_i32_i24 %0, %2, %3, %4\n\tv_ashrrev_i32 %1, 3, %0" : "=v"(zh), "=v"(shifted) : "v"(ah), "s"(s_i), "v"((uint32_t)(lo >> 32)));
        else
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(zh) : "v"(ah), "s"(s_i), "v"((uint32_t)(lo >> 32)));
    } else {
        zh = (uint32_t)(lo >> 32) + (uint32_t)ah * (uint32_t)s_i;
        if (kShift) {
            shifted = (int32_t)zh >> 3;
            asm volatile("" : "+v"(shifted)); // (stays a vector shift: the compiler would move it behind the readlane)
        }
    }
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
//
// kVecShift: where the >> 3 of the prediction happens.  false: on the scalar unit, behind the v_readlane (three vector
// instructions per step: what a SIMD shared by seven waves, bound by vector issue, wants).  true: on the vector unit, in
// front of it (four, but the step's dependency chain loses its detour through the scalar ALU: 36 instead of 50 cycles per
// sample for a wave that has its SIMD to itself -- tools/chain_ubench.py -- which is how small batches and the last
// workgroups of a launch run).  Same bits either way.
template <int R, bool kFold, int G, bool kVecShift, int M>
__device__ __forceinline__ void synth_steps(uint32_t& cl, uint32_t& ch, uint32_t& ol, uint32_t& oh, uint32_t& kept,
    LdsTable tab_lane, int32_t r_block, uint32_t four, uint32_t zero, uint64_t (&pf_c)[kAhead], uint64_t (&pf_o)[kAhead], int32_t& shifted)
{
    // scalar side: the sum of this sample sits in lane M.  What goes back into the sums is -a_d * s_i; the table holds
    // +a_d, so the multiplier is -s_i: in the folded form that IS the shifted sum (s_i = -pred: one scalar operation
    // between the readlane and the multiply-adds instead of two), in the exact form pred - r_i.
    // (kVecShift: `shifted` = ch >> 3 as of the end of the step before -- the lanes a step recycles are behind it)
    const int32_t pred = kVecShift ? __builtin_amdgcn_readlane(shifted, M) : __builtin_amdgcn_readlane((int)ch, M) >> 3;
    int32_t m_i;
    if (kFold)
        m_i = pred;
    else
        m_i = (int32_t)((uint32_t)pred - (uint32_t)__builtin_amdgcn_readlane(r_block, M));
    // vector side: lane L adds a[(L - M) mod ring] * (-s_i)  (a[0] = 0: the finished sum stays)
    synth_mac<kFold, kVecShift>(cl, ch, pf_c[M % kAhead], m_i, shifted);
    if (R == 2) {
        int32_t unused;
        synth_mac<kFold, false>(ol, oh, pf_o[M % kAhead], m_i, unused);
    }
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
        synth_steps<R, kFold, G, kVecShift, M + 1>(cl, ch, ol, oh, kept, tab_lane, r_block, four, zero, pf_c, pf_o, shifted);
}

// One block of 64 samples with residues r_block (one per lane).  The folded form returns false if a sample
// of the block left its range (s is then meaningless).
template <int R, bool kFold, int G, bool kVecShift>
__device__ __forceinline__ bool synth_block(int32_t r_block, int32_t& s, uint32_t& cl, uint32_t& ch, uint32_t& ol, uint32_t& oh,
    LdsTable tab_lane, uint32_t four, uint32_t zero)
{
    uint64_t pf_c[kAhead], pf_o[kAhead];
#pragma unroll
    for (int m = 0; m < kAhead; m++) {
        pf_c[m] = tab_lane[64 * R - m];
        pf_o[m] = R == 2 ? tab_lane[64 - m] : 0;
    }
    if (kFold)
        ch -= (uint32_t)r_block << 3; // sample lane of this block: N -= r * 2^35
    uint32_t kept = 0;
    int32_t shifted = 0;
    if (kVecShift) {
        shifted = (int32_t)ch >> 3;
        asm volatile("" : "+v"(shifted)); // (stays a vector shift: the compiler would move it behind the readlane)
    }
    __builtin_amdgcn_sched_barrier(0);
    synth_steps<R, kFold, G, kVecShift, 0>(cl, ch, ol, oh, kept, tab_lane, r_block, four, zero, pf_c, pf_o, shifted);
    s = (int32_t)((kFold ? 0u : (uint32_t)r_block) - (uint32_t)((int32_t)kept >> 3));
    // (the multiplier of a folded step is -s: both s and -s must fit the 24-bit operand)
    return !kFold || !__any((uint32_t)(s + (1 << 23) - 1) >= (1u << 24) - 1u);
}

// All 2048 samples of a subframe.  R = ring / 64 (1: order <= 64 - G, 2: order <= 128 - G); G = recycling
// group (4 or 16).  fold = start in the folded form (the coefficients fit it).  Residues: decoded just in time
// from the codeword positions in pos_smp[] (ws == nullptr), the words fetched one block ahead -- or read from
// the workspace array ws[] (generic mode).  Samples go to pos_smp[] as int16, over the positions of the
// block just consumed.
template <int R, int G, bool kVecShift, bool kOut32 = false>
__device__ __attribute__((noinline)) void synthesize( // (a real call: six of these inlined into three kernels cost the kernels their registers)
    const uint32_t* words, uint32_t n_words, uint32_t k, uint16_t* pos_smp, const int32_t* ws,
    const uint64_t* tab, bool fold, int lane)
{
    // kOut32 (the stage on its own, k_stage_lpc_decode): pos_smp is really an int32_t* in global memory and takes the samples
    // as the 32-bit values lpc::SampleGenerator returns (ws != nullptr there: no positions are read from it).  A template
    // parameter rather than one more argument: the frame kernels' instantiations stay exactly what they were.
    static_assert(G == 4 || G == 16, "groups are DPP banks or rows");
    const StreamWords sw = { words, n_words };
    const __amdgpu_buffer_rsrc_t rs = stream_rsrc(sw);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    const uint32_t kmask = k ? (0xFFFFFFFFu >> (32 - k)) : 0u;
    const bool jit = read_first_lane(reinterpret_cast<uint64_t>(ws)) == 0;
    uint32_t zl[2], zh[2];
#pragma unroll
    for (int h = 0; h < 2; h++) { // every sum starts at 2^34
        zl[h] = 0;
        zh[h] = 4;
    }
    uint32_t four = 4, zero = 0;
    asm volatile("" : "+v"(four), "+v"(zero)); // DPP sources must be VGPRs
    const LdsTable tab_lane = (LdsTable)(tab + lane); // the table is in LDS: ds_read with immediate offsets
    // the block ahead: its codeword's position and three stream words (or its residue, generic mode)
    uint32_t p = 0, w0 = 0, w1 = 0, w2 = 0;
    int32_t r_ws = 0;
    auto issue = [&](int blk) {
        if (jit) {
            p = pos_smp[64 * blk + lane];
            const uint32_t b = 4 * (p >> 5);
            w0 = __builtin_amdgcn_raw_buffer_load_b32(rs, b, 0, 0);
            w1 = __builtin_amdgcn_raw_buffer_load_b32(rs, b + 4, 0, 0);
            w2 = __builtin_amdgcn_raw_buffer_load_b32(rs, b + 8, 0, 0);
        } else {
            r_ws = ws[64 * blk + lane];
        }
    };
    auto land = [&]() -> int32_t {
        if (!jit)
            return r_ws;
        uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, p & 31), x1 = __builtin_amdgcn_alignbit(w2, w1, p & 31), ones = 0;
        while (__any(x0 == 0xFFFFFFFFu)) { // runs of 32 ones and more (rare): follow them word by word
            const bool more = x0 == 0xFFFFFFFFu;
            ones += more ? 32u : 0u;
            p += more ? 32u : 0u;
            window64(rs, p, x0, x1);
        }
        const uint32_t t = (uint32_t)__builtin_ctz(~x0);
        const uint32_t field = (uint32_t)(((((uint64_t)x1) << 32) | x0) >> (t + 1)) & kmask;
        return rice_value(ones + t, field, k);
    };
    // one block: (cl, ch) = the register whose sums finish in it, (ol, oh) = the other register of a ring of 128
    auto run_block = [&](int blk, uint32_t& cl, uint32_t& ch, uint32_t& ol, uint32_t& oh) {
        const int32_t r_block = land();
        if (blk + 1 < kBlock / 64)
            issue(blk + 1); // in flight during this block's 64 steps
        int32_t s;
        bool done = false;
        if (fold) {
            const uint32_t s0 = cl, s1 = ch, s2 = ol, s3 = oh;
            done = synth_block<R, true, G, kVecShift>(r_block, s, cl, ch, ol, oh, tab_lane, four, zero);
            if (!done) { // back to the block's start, exact form from here on
                cl = s0, ch = s1;
                if (R == 2)
                    ol = s2, oh = s3;
                fold = false;
            }
        }
        if (!done)
            synth_block<R, false, G, kVecShift>(r_block, s, cl, ch, ol, oh, tab_lane, four, zero);
        if constexpr (kOut32)
            reinterpret_cast<int32_t*>(pos_smp)[64 * blk + lane] = s;
        else
            reinterpret_cast<int16_t*>(pos_smp)[64 * blk + lane] = (int16_t)(uint16_t)(uint32_t)s;
    };
    issue(0);
#pragma unroll 1
    for (int pair = 0; pair < kBlock / 64; pair += R) {
        run_block(pair, zl[0], zh[0], zl[R - 1], zh[R - 1]);
        if (R == 2)
            run_block(pair + 1, zl[1], zh[1], zl[0], zh[0]);
    }
    wave_sync();
}

// ring / recycling group by order: <= 48: 64 / 16, <= 60: 64 / 4, else 128 / 16
template <bool kVecShift, bool kOut32 = false>
__device__ __forceinline__ void synthesize_by_order(uint32_t order, const uint32_t* words, uint32_t n_words, uint32_t k, uint16_t* pos_smp,
    const int32_t* ws, const uint64_t* tab, bool fold, int lane)
{
    if (order <= 48)
        synthesize<1, 16, kVecShift, kOut32>(words, n_words, k, pos_smp, ws, tab, fold, lane);
    else if (order <= 60)
        synthesize<1, 4, kVecShift, kOut32>(words, n_words, k, pos_smp, ws, tab, fold, lane);
    else
        synthesize<2, 16, kVecShift, kOut32>(words, n_words, k, pos_smp, ws, tab, fold, lane);
}

// The coefficients a[d] (0 for d = 0 and beyond `order`), packed {al, ah}, ring-periodic and doubled, written
// over the wave's k[] / a[] arrays.  Returns whether every ah fits 24 bits.  (The sums accumulate
// 2^34 - sum a_d s = 2^34 + sum a_d (-s): the multiplier carries the sign, see synth_steps.)
__device__ inline bool build_synth_table(const int64_t* a, uint64_t* tab, int order, int lane)
{
    uint64_t c[2];
    bool fits = true;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int d = lane + 64 * h;
        const uint64_t nv = (uint64_t)(d >= 1 && d <= order ? a[d] : 0);
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
        tab[lane + 128] = c[0]; // (a step reads entries lane + 128 - M and lane + 64 - M, M = 0 .. 63: 1 .. 191)
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

// LDS plan of k_decode_frames (dynamic): one DecSubframeLds per subframe POSITION | one DecWaveScratch per wave |
// sub_info[channels] (channel -> type | parent << 8 | position << 16) | too_big[n_waves].
__host__ __device__ inline size_t decode_lds_bytes_for(uint32_t channels, int n_waves)
{
    return (size_t)channels * sizeof(DecSubframeLds) + (size_t)n_waves * sizeof(DecWaveScratch) + (size_t)channels * 4 + (size_t)n_waves * 4;
}

// kProf: also write per-phase cycle counts (debug hook sela_hip_debug_phase_buffer; 16 uint64 per subframe).
template <bool kProf>
__global__ __launch_bounds__(kDecMaxWaves * 64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_decode_frames(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* __restrict__ pcm_out,
    uint32_t* __restrict__ status, int32_t* __restrict__ ws_residues, uint64_t* __restrict__ phase_cycles,
    uint8_t* __restrict__ frame_flags /* or null: one byte per (frame, wave), written only when not zero */,
    uint32_t vec_shift_from /* frames from this one on run the synthesis with the shift on the vector side (synth_steps) */,
    uint32_t synth_priorities /* s_setprio through the synthesis by the subframe's order class: byte 0 orders <= 48, 1 <= 60, 2 above; 0: none */)
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
    DecSubframeLds* const sub = reinterpret_cast<DecSubframeLds*>(dyn);
    DecWaveScratch* const scratch = reinterpret_cast<DecWaveScratch*>(dyn + (size_t)channels * sizeof(DecSubframeLds)) + wave;
    uint32_t* const sub_info = reinterpret_cast<uint32_t*>(dyn + (size_t)channels * sizeof(DecSubframeLds) + (size_t)n_waves * sizeof(DecWaveScratch));
    uint32_t* const too_big = sub_info + channels; // [n_waves]: this wave's subframe does not fit the fast plan

    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const bool vec_shift = f >= vec_shift_from;
    const uint8_t* const fb = frames + frame_offsets[f];
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    uint32_t flags = 0;
    for (uint32_t c = threadIdx.x; c < channels; c += blockDim.x)
        sub_info[c] = 0xFFFFFFFFu; // "no subframe delivered this channel"

    // ---- mode: every subframe of the frame must fit the fast plan ----------------------------------------------
    const bool fast_plan = channels <= (uint32_t)kDecMaxWaves;
    SubHeader hd = walk_headers(fb, fbytes, (uint32_t)wave < channels ? (uint32_t)wave : 0u, channels);
    if (lane == 0)
        too_big[wave] = (fast_plan && (!hd.ok || (hd.cw + 2 + hd.rw <= (uint32_t)kStreamCap && hd.order <= 2 * (uint32_t)kWave))) ? 0u : 1u;
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
        DecSubframeLds* const sl = sub + c;
        const uint32_t nw = hd.cw + 2 + hd.rw;
        const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4); // the subframe's aligned words
        const int32_t* ws_c = nullptr;
        ParseProfile pp;
        if (fast) {
            for (uint32_t w = lane; w < nw + kStreamMargin; w += kWave) // the start bitmap
                sl->marks[w] = 0;
            wave_sync();
            if (kProf)
                stamp[2] = clock64(), prof_sub = c;
            const StreamWords sw = { gw, nw };
            flags |= parse_subframe<kProf>(sw, sl->marks, sl->pos, reinterpret_cast<uint16_t*>(&scratch->t), coef_values(scratch), hd.cw, hd.rw, hd.ck, hd.rk,
                hd.order, lane, pp);
        } else {
            if (kProf)
                stamp[2] = clock64(), prof_sub = c;
            int32_t* const wres = ws_residues + ((size_t)f * channels + c) * kBlock;
            const uint32_t n_frame_words = (uint32_t)((fbytes - hd.p - 4) / 4);
            flags |= parse_stream_serial(gw, 24, 24 + 32 * hd.cw, n_frame_words, hd.ck, hd.order, coef_values(scratch), lane);
            flags |= parse_stream_serial(gw, 32 * (hd.cw + 2), 32 * (hd.cw + 2 + hd.rw), n_frame_words, hd.rk, (uint32_t)kBlock, wres, lane);
            __threadfence(); // lane 0's stores to the workspace are read back by every lane
            ws_c = wres;
            pp.t[0] = pp.t[1] = pp.t[2] = pp.t[3] = kProf ? clock64() : 0;
        }
        if (kProf)
            stamp[3] = pp.t[0], stamp[4] = pp.t[1], stamp[5] = pp.t[2], stamp[6] = pp.t[3];

        // dequantise (src/lpc/linear_predictor.cpp:16-28) + step-up
        SynthTables* const tables = &scratch->t;
        const uint32_t order = hd.order;
        const int32_t q_lo = (uint32_t)lane < order ? coef_values(scratch)[lane] : 0, q_hi = (uint32_t)lane + 64 < order ? coef_values(scratch)[lane + 64] : 0;
        wave_sync();
        const double k_lo = (uint32_t)lane < order ? (order <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
        const double k_hi = (uint32_t)lane + 64 < order ? dequant(lane + 64, q_hi, flags) : 0.0;
        step_up_regs(k_lo, k_hi, tables->a, (int)order, lane, flags);
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        if (kProf)
            stamp[7] = clock64();
        if (synth_priorities)
            set_wave_priority((int)((synth_priorities >> (order <= 48 ? 0 : (order <= 60 ? 8 : 16))) & 0xFF));
        if (vec_shift)
            synthesize_by_order<true>(order, gw, nw, hd.rk, sl->pos, ws_c, tables->tab, fits24, lane);
        else
            synthesize_by_order<false>(order, gw, nw, hd.rk, sl->pos, ws_c, tables->tab, fits24, lane);
        if (synth_priorities)
            __builtin_amdgcn_s_setprio(0);
        if (kProf)
            stamp[8] = clock64();
        if (lane == 0)
            sub_info[hd.channel] = hd.type | (hd.parent << 8) | (c << 16);
    }
    __syncthreads();

    // ---- second pass of frame::FrameDecoder + interleave to int16 ------------------------------------
    // dependent channels become parent - difference (parents are independent subframes); a channel
    // that no valid subframe delivered decodes to silence and raises BAD_FRAME.  All of it mod 2^16: the
    // reference truncates to int16 when it writes the WAV (src/file/wav_file.cpp:248-251).
    if (channels == 2) {
        // stereo: four samples of both channels per thread, one 16-byte store (wave-uniform case analysis)
        const uint32_t i0 = sub_info[0], i1 = sub_info[1];
        const bool have0 = i0 != 0xFFFFFFFFu, have1 = i1 != 0xFFFFFFFFu;
        const bool dep0 = have0 && (i0 & 0xFF) == 1, dep1 = have1 && (i1 & 0xFF) == 1;
        const uint32_t par0 = (i0 >> 8) & 0xFF, par1 = (i1 >> 8) & 0xFF; // parent channel of a dependent subframe (0 or 1, checked above)
        const uint2* s0 = reinterpret_cast<const uint2*>(sub[have0 ? i0 >> 16 : 0].smp);
        const uint2* s1 = reinterpret_cast<const uint2*>(sub[have1 ? i1 >> 16 : 0].smp);
        uint4* out = reinterpret_cast<uint4*>(pcm_out + (size_t)f * kBlock * 2);
        for (uint32_t i4 = threadIdx.x; i4 < (uint32_t)kBlock / 4; i4 += blockDim.x) {
            const uint2 zero = make_uint2(0, 0);
            const uint2 r0 = have0 ? s0[i4] : zero, r1 = have1 ? s1[i4] : zero; // raw subframe outputs, two samples per word
            // per 16-bit half: parent - difference (the parent's own, independent samples)
            auto sub16 = [](uint32_t a, uint32_t b) -> uint32_t { return ((a - (b & 0xFFFFu)) & 0xFFFFu) | ((a & 0xFFFF0000u) - (b & 0xFFFF0000u)); };
            uint2 a = r0, b = r1;
            if (dep0) {
                const uint2 pv = par0 == 0 ? r0 : r1;
                a = make_uint2(sub16(pv.x, r0.x), sub16(pv.y, r0.y));
            }
            if (dep1) {
                const uint2 pv = par1 == 0 ? r0 : r1;
                b = make_uint2(sub16(pv.x, r1.x), sub16(pv.y, r1.y));
            }
            uint4 w;
            w.x = (a.x & 0xFFFFu) | (b.x << 16);
            w.y = (a.x >> 16) | (b.x & 0xFFFF0000u);
            w.z = (a.y & 0xFFFFu) | (b.y << 16);
            w.w = (a.y >> 16) | (b.y & 0xFFFF0000u);
            out[i4] = w;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < (uint32_t)kBlock; i += blockDim.x) {
            for (uint32_t c = 0; c < channels; c++) {
                const uint32_t info = sub_info[c];
                uint32_t v = info == 0xFFFFFFFFu ? 0u : (uint32_t)(uint16_t)sub[info >> 16].smp[i];
                if (info != 0xFFFFFFFFu && (info & 0xFF) == 1) {
                    const uint32_t pinfo = sub_info[(info >> 8) & 0xFF];
                    const uint32_t pv = pinfo == 0xFFFFFFFFu ? 0u : (uint32_t)(uint16_t)sub[pinfo >> 16].smp[i];
                    v = pv - v;
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
    if (lane == 0 && flags) {
        if (frame_flags) // (the host pipeline: page-locked host memory, zeroed by the host; a plain store, no atomics over the link)
            frame_flags[(size_t)f * n_waves + wave] = (uint8_t)flags;
        else {
            atomicOr(&status[0], flags);
            if (wave == 0 && (flags & SELA_HIP_FLAG_BAD_FRAME))
                atomicAdd(&status[1], 1u);
        }
    }
    if (kProf && lane == 0 && prof_sub != 0xFFFFFFFFu) { // (one subframe per wave is reported)
        stamp[9] = clock64();
        for (int i = 0; i < 9; i++)
            phase_cycles[((size_t)f * channels + prof_sub) * 16 + i] = (uint64_t)(stamp[i + 1] - stamp[i]);
    }
}


// ---- more than eight channels: k_decode_frames_wide ---------------------------------------------------------------
// frame::FrameDecoder::process takes any channel count the header's 8-bit field carries (src/frame/frame_decoder.cpp:
// 11-72).  k_decode_frames keeps every channel of a frame in LDS until the parent - difference pass, which bounds it
// (one wave and one 4.5 KB record per channel).  Here the eight waves of a workgroup take the frame's subframes in
// rounds, each wave with a record of its own: a subframe goes from the frame bytes to finished samples exactly as above
// (the segment-parallel parse when it fits the plan, the serial parse into the workspace otherwise -- decided per
// SUBFRAME: no other wave depends on this one's record) and leaves its raw samples in the output, strided.  The second
// pass (src/frame/frame_decoder.cpp:40-69) then runs over the output: a difference-coded channel becomes parent -
// difference, where the parent's samples are the raw ones an independent subframe left there.  The reference's encoder
// writes difference subframes for exactly-stereo input only (src/frame/frame_encoder.cpp:18), so for these frames the
// pass has nothing to do -- but files are input, not promises.
struct HeaderCursor {
    uint32_t index; // subframe the cursor stands in front of
    uint64_t p;     // its byte offset in the frame
    bool ok;
};

__device__ inline SubHeader walk_on(const uint8_t* fb, uint64_t fbytes, HeaderCursor& cur, uint32_t c, uint32_t channels)
{
    SubHeader h;
    h.ok = cur.ok;
    h.channel = h.type = h.parent = h.ck = h.cw = h.order = h.rk = h.rw = 0;
    uint32_t n = 0;
    uint64_t p = cur.p;
    while (h.ok && cur.index <= c) { // over the headers up to and including subframe c
        p = cur.p;
        if (p + 12 > fbytes) {
            h.ok = false;
            break;
        }
        const uint32_t h0 = *reinterpret_cast<const uint32_t*>(fb + p);
        const uint32_t h1 = *reinterpret_cast<const uint32_t*>(fb + p + 4);
        h.channel = h0 & 0xFF, h.type = (h0 >> 8) & 0xFF, h.parent = (h0 >> 16) & 0xFF, h.ck = h0 >> 24;
        h.cw = h1 & 0xFFFF, h.order = (h1 >> 16) & 0xFF;
        const uint64_t p2 = p + 4 + 4 * (uint64_t)h.cw;
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
        cur.p = next;
        cur.index++;
    }
    cur.ok = h.ok; // (a frame is walked front to back: behind a broken header there is nothing to find)
    h.ok = h.ok && h.channel < channels && h.order <= (uint32_t)kMaxOrder && n == (uint32_t)kBlock && h.ck < 32 && h.rk < 32 && h.type <= 1
        && (h.type == 0 || h.parent < channels);
    h.p = (uint32_t)p;
    return h;
}

constexpr uint32_t kNoSubframe = 0xFFFFFFFFu;

__global__ __launch_bounds__(kDecMaxWaves * 64) void k_decode_frames_wide(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* __restrict__ pcm_out,
    uint32_t* __restrict__ status, int32_t* __restrict__ ws_residues, uint8_t* __restrict__ frame_flags, uint32_t vec_shift_from)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)), lane = threadIdx.x % 64;
    DecSubframeLds* const sl = reinterpret_cast<DecSubframeLds*>(dyn) + wave;
    DecWaveScratch* const scratch = reinterpret_cast<DecWaveScratch*>(dyn + (size_t)kDecMaxWaves * sizeof(DecSubframeLds)) + wave;
    uint32_t* const sub_info = reinterpret_cast<uint32_t*>(dyn + (size_t)kDecMaxWaves * (sizeof(DecSubframeLds) + sizeof(DecWaveScratch)));
    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const bool vec_shift = f >= vec_shift_from;
    const uint8_t* const fb = frames + frame_offsets[f];
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    uint32_t flags = 0;
    for (uint32_t c = threadIdx.x; c < channels; c += blockDim.x)
        sub_info[c] = kNoSubframe; // "no subframe delivered this channel"
    __syncthreads();
    HeaderCursor cur;
    cur.index = 0;
    cur.p = 4;
    cur.ok = fbytes >= 4 && fbytes < 0x7FFFFFFFull && (fbytes & 3) == 0 && reinterpret_cast<const uint32_t*>(fb)[0] == SELA_SYNC_WORD;
    int16_t* const out_frame = pcm_out + (size_t)f * kBlock * channels;
    for (uint32_t c = wave; c < channels; c += kDecMaxWaves) {
        const SubHeader hd = walk_on(fb, fbytes, cur, c, channels);
        if (!hd.ok) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            continue;
        }
        const uint32_t nw = hd.cw + 2 + hd.rw;
        const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4); // the subframe's aligned words
        const int32_t* ws_c = nullptr;
        ParseProfile pp;
        if (nw <= (uint32_t)kStreamCap) { // (hd.order <= 100 <= 2 waves' worth: checked by walk_on)
            for (uint32_t w = lane; w < nw + kStreamMargin; w += kWave) // the start bitmap
                sl->marks[w] = 0;
            wave_sync();
            const StreamWords sw = { gw, nw };
            flags |= parse_subframe<false>(sw, sl->marks, sl->pos, reinterpret_cast<uint16_t*>(&scratch->t), coef_values(scratch), hd.cw, hd.rw, hd.ck, hd.rk,
                hd.order, lane, pp);
        } else {
            int32_t* const wres = ws_residues + ((size_t)f * channels + c) * kBlock;
            const uint32_t n_frame_words = (uint32_t)((fbytes - hd.p - 4) / 4);
            flags |= parse_stream_serial(gw, 24, 24 + 32 * hd.cw, n_frame_words, hd.ck, hd.order, coef_values(scratch), lane);
            flags |= parse_stream_serial(gw, 32 * (hd.cw + 2), 32 * (hd.cw + 2 + hd.rw), n_frame_words, hd.rk, (uint32_t)kBlock, wres, lane);
            // lane 0's stores to the workspace are read back by every lane of this wave: they have left the CU, and nothing
            // older is served from its vector cache (not __threadfence(): its release half writes back the whole L2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ws_c = wres;
        }
        SynthTables* const tables = &scratch->t;
        const uint32_t order = hd.order;
        const int32_t q_lo = (uint32_t)lane < order ? coef_values(scratch)[lane] : 0, q_hi = (uint32_t)lane + 64 < order ? coef_values(scratch)[lane + 64] : 0;
        wave_sync();
        const double k_lo = (uint32_t)lane < order ? (order <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
        const double k_hi = (uint32_t)lane + 64 < order ? dequant(lane + 64, q_hi, flags) : 0.0;
        step_up_regs(k_lo, k_hi, tables->a, (int)order, lane, flags);
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        if (vec_shift)
            synthesize_by_order<true>(order, gw, nw, hd.rk, sl->pos, ws_c, tables->tab, fits24, lane);
        else
            synthesize_by_order<false>(order, gw, nw, hd.rk, sl->pos, ws_c, tables->tab, fits24, lane);
        // the raw samples of this subframe, where its channel lies in the output (mod 2^16: src/file/wav_file.cpp:248-251)
        for (int i = lane; i < kBlock; i += kWave)
            out_frame[(size_t)i * channels + hd.channel] = sl->smp[i];
        wave_sync(); // (the record is reused by this wave's next subframe)
        if (lane == 0)
            sub_info[hd.channel] = hd.type | (hd.parent << 8);
    }
    // ---- second pass of frame::FrameDecoder, over the output -----------------------------------------------------------
    // Writers and readers are waves of ONE workgroup, i.e. of one CU behind one L2: the stores only have to have left the
    // CU (the wait is written out -- a fence's own wait is dropped when the compiler believes nothing is outstanding,
    // MI355X_MICROARCH.md "inter-workgroup visibility"), and nothing older may be served from its vector cache.  (A release
    // fence at agent scope would write back the whole L2 for nothing.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t c = 0; c < channels; c++) {
        const uint32_t info = sub_info[c]; // (wave-uniform)
        if (info == kNoSubframe) { // a channel that no valid subframe delivered decodes to silence
            for (uint32_t i = threadIdx.x; i < (uint32_t)kBlock; i += blockDim.x)
                out_frame[(size_t)i * channels + c] = 0;
            if (threadIdx.x == 0)
                flags |= SELA_HIP_FLAG_BAD_FRAME;
        } else if ((info & 0xFF) == 1) {
            const uint32_t parent = (info >> 8) & 0xFF;
            const uint32_t pinfo = sub_info[parent];
            // (a parent that is itself difference-coded is outside what the reference defines: flagged, and its raw samples taken)
            if (threadIdx.x == 0 && (pinfo == kNoSubframe || (pinfo & 0xFF) != 0))
                flags |= SELA_HIP_FLAG_BAD_FRAME;
            for (uint32_t i = threadIdx.x; i < (uint32_t)kBlock; i += blockDim.x) {
                const uint32_t pv = pinfo == kNoSubframe ? 0u : (uint32_t)(uint16_t)out_frame[(size_t)i * channels + parent];
                const uint32_t dv = (uint32_t)(uint16_t)out_frame[(size_t)i * channels + c];
                out_frame[(size_t)i * channels + c] = (int16_t)(uint16_t)(pv - dv);
            }
        }
    }
    flags = wave_or(flags);
    if (lane == 0 && flags) {
        if (frame_flags)
            frame_flags[(size_t)f * kDecMaxWaves + wave] = (uint8_t)flags;
        else {
            atomicOr(&status[0], flags);
            if (wave == 0 && (flags & SELA_HIP_FLAG_BAD_FRAME))
                atomicAdd(&status[1], 1u);
        }
    }
}

int decode_waves(uint32_t channels)
{
    return channels < (uint32_t)kDecMaxWaves ? (int)channels : kDecMaxWaves;
}

size_t decode_lds_bytes(uint32_t channels) { return decode_lds_bytes_for(channels, decode_waves(channels)); }

uint32_t decode_max_channels() { return (uint32_t)kDecMaxChannels; }

// Generic mode parks the residues of a subframe here (int32[2048]); the usual path never touches it.
size_t decode_workspace_bytes(uint32_t n_frames, uint32_t channels)
{
    return (size_t)n_frames * channels * kBlock * sizeof(int32_t) + 256;
}

// Which workgroups of a launch run their recurrence in the form for a wave that has its SIMD (nearly) to itself
// (synth_steps, kVecShift): all of them when the whole launch is at most kLonelyWaves waves, else those of the last, partial
// round of resident workgroups if that round is as small -- they start when the rounds before them are done.
constexpr uint32_t kLonelyWaves = 2560; // 2.5 per SIMD (tools/chain_ubench.py: the forms break even between 2 and 4)
struct DecodeResidency {
    std::atomic<uint32_t> frames[64][kDecMaxWaves + 1]; // [device][waves per workgroup]: workgroups the device holds at once, 0 = not asked yet
};
inline uint32_t resident_frames(const void* kernel, int n_waves, size_t lds)
{
    static DecodeResidency cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 0;
    uint32_t have = cache.frames[dev][n_waves].load(std::memory_order_relaxed);
    if (have == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, n_waves * 64, lds) != hipSuccess
            || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0)
            return 0;
        have = (uint32_t)per_cu * (uint32_t)cus;
        cache.frames[dev][n_waves].store(have, std::memory_order_relaxed);
    }
    return have;
}
inline uint32_t vec_shift_from_for(uint32_t n_frames, int n_waves, uint32_t resident)
{
    if ((uint64_t)n_frames * (uint32_t)n_waves <= kLonelyWaves)
        return 0;
    if (resident == 0)
        return n_frames;
    const uint32_t last_round = n_frames / resident * resident; // first frame of the last, partial round
    return (uint64_t)(n_frames - last_round) * (uint32_t)n_waves <= kLonelyWaves ? last_round : n_frames;
}

// ---- the decoder's stages on their own (the reference's public L1 classes) -------------------------------------------------
// lpc::SampleGenerator (src/include/lpc.hpp:106-117, src/lpc/sample_generator.cpp:11-39): order + quantised reflection
// coefficients + 2048 residues -> 2048 samples, one wave per block, through the very dequantisation, step-up, table and
// recurrence the frame kernels run (the residues come from memory as in the generic mode; the samples go out as the 32-bit
// values the reference's class returns -- a difference signal needs 17 bits).
__global__ __launch_bounds__(64) void k_stage_lpc_decode(const int32_t* __restrict__ order_in, const int32_t* __restrict__ q_in,
    const int32_t* __restrict__ residues, uint32_t n_blocks, int32_t* __restrict__ samples_out, int64_t* __restrict__ coefs_out /* [block][101] or null */,
    uint32_t* __restrict__ status)
{
    __shared__ DecWaveScratch scratch;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks)
        return;
    const int lane = threadIdx.x;
    uint32_t flags = 0;
    const int32_t o = order_in[b];
    if (o < 0 || o > kMaxOrder) {
        if (lane == 0)
            atomicOr(&status[0], (uint32_t)SELA_HIP_FLAG_BAD_FRAME);
        return;
    }
    const uint32_t order = (uint32_t)o;
    const int32_t q_lo = (uint32_t)lane < order ? q_in[(size_t)b * kMaxOrder + lane] : 0;
    const int32_t q_hi = (uint32_t)lane + 64 < order ? q_in[(size_t)b * kMaxOrder + lane + 64] : 0;
    const double k_lo = (uint32_t)lane < order ? (order <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
    const double k_hi = (uint32_t)lane + 64 < order ? dequant(lane + 64, q_hi, flags) : 0.0;
    SynthTables* const tables = &scratch.t;
    step_up_regs(k_lo, k_hi, tables->a, (int)order, lane, flags);
    if (coefs_out) // lpc::LinearPredictor::linearPredictionCoefficients (src/lpc/linear_predictor.cpp:57-60)
        for (uint32_t i = lane; i <= order; i += kWave)
            coefs_out[(size_t)b * (kMaxOrder + 1) + i] = tables->a[i];
    if (samples_out) {
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        synthesize_by_order<false, true>(order, nullptr, 0, 0, reinterpret_cast<uint16_t*>(samples_out + (size_t)b * kBlock), residues + (size_t)b * kBlock,
            tables->tab, fits24, lane);
    }
    flags = wave_or(flags);
    if (lane == 0 && flags)
        atomicOr(&status[0], flags);
}

// rice::RiceDecoder (src/include/rice.hpp:29-43, src/rice/rice_decoder.cpp:11-61): one wave per stream, the serial parse of the
// generic mode -- count the ones up to the first zero, read k bits MSB first -- for any number of values.
__global__ __launch_bounds__(64) void k_stage_rice_decode(const uint32_t* __restrict__ words, const uint64_t* __restrict__ word_offsets,
    const uint32_t* __restrict__ k_in, const uint64_t* __restrict__ value_offsets, uint32_t n_streams, int32_t* __restrict__ values_out,
    uint32_t* __restrict__ status)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_streams)
        return;
    const int lane = threadIdx.x;
    const uint64_t nw = word_offsets[s + 1] - word_offsets[s], count = value_offsets[s + 1] - value_offsets[s];
    const uint32_t k = k_in[s];
    uint32_t flags = 0;
    if (k >= 32 || nw >= (1ull << 26) || count >= (1ull << 31))
        flags = SELA_HIP_FLAG_BAD_FRAME;
    else
        flags = parse_stream_serial(words + word_offsets[s], 0, (uint32_t)(32 * nw), (uint32_t)nw, k, (uint32_t)count, values_out + value_offsets[s], lane);
    if (lane == 0 && flags)
        atomicOr(&status[0], flags);
}

hipError_t launch_stage_lpc_decode(const int32_t* d_order, const int32_t* d_q, const int32_t* d_residues, uint32_t n_blocks, int32_t* d_samples,
    int64_t* d_coefs, uint32_t* d_status, hipStream_t stream)
{
    if (n_blocks)
        hipLaunchKernelGGL(k_stage_lpc_decode, dim3(n_blocks), dim3(64), 0, stream, d_order, d_q, d_residues, n_blocks, d_samples, d_coefs, d_status);
    return hipGetLastError();
}

hipError_t launch_stage_rice_decode(const uint32_t* d_words, const uint64_t* d_word_offsets, const uint32_t* d_k, const uint64_t* d_value_offsets,
    uint32_t n_streams, int32_t* d_values, uint32_t* d_status, hipStream_t stream)
{
    if (n_streams)
        hipLaunchKernelGGL(k_stage_rice_decode, dim3(n_streams), dim3(64), 0, stream, d_words, d_word_offsets, d_k, d_value_offsets, n_streams, d_values, d_status);
    return hipGetLastError();
}

hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, hipStream_t stream, hipEvent_t* ev /* 2 events or nullptr */,
    uint64_t* d_phase_cycles, uint8_t* frame_flags /* null: flags go to d_status, which is zeroed here */,
    int recurrence_form /* -1: by launch size (vec_shift_from_for); 0 / 1: every frame in the scalar- / vector-shift form (tests) */,
    uint32_t synth_priorities /* s_setprio through the synthesis by order class (k_decode_frames); 0: none.  A launch that has the
                                 device to itself raises the subframes of orders above 60 -- two registers per lane, 1.6 x the
                                 synthesis work, 31 % of the bench track's subframes -- so that they do not finish last:
                                 k_decode_frames 0.247 -> 0.240 ms at 3875 frames, one lane +1.1 %; beside another stream's
                                 kernels the same costs 0.4 %, so the caller passes 0 there */)
{
    hipError_t err = frame_flags ? hipSuccess : hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
    if (err != hipSuccess || n_frames == 0)
        return err;
    const int n_waves = decode_waves(channels);
    if (channels > (uint32_t)kDecMaxChannels)
        return hipErrorInvalidValue;
    int32_t* ws = reinterpret_cast<int32_t*>(((uintptr_t)d_workspace + 255) & ~(uintptr_t)255);
    if (channels > (uint32_t)kDecMaxWaves) { // more channels than a workgroup has waves: rounds (k_decode_frames_wide)
        if (d_phase_cycles)
            return hipErrorInvalidValue; // (the phase counts are the one-wave-per-subframe kernel's)
        const size_t lds = (size_t)kDecMaxWaves * (sizeof(DecSubframeLds) + sizeof(DecWaveScratch)) + (size_t)channels * 4;
        if (ev)
            (void)hipEventRecord(ev[0], stream);
        const uint32_t from = recurrence_form >= 0 ? (recurrence_form ? 0u : n_frames)
                                                   : vec_shift_from_for(n_frames, kDecMaxWaves, resident_frames(reinterpret_cast<const void*>(k_decode_frames_wide), kDecMaxWaves, lds));
        hipLaunchKernelGGL(k_decode_frames_wide, dim3(n_frames), dim3(kDecMaxWaves * 64), lds, stream, d_frames, d_frame_offsets, n_frames, channels,
            d_pcm_out, d_status, ws, frame_flags, from);
        if (ev)
            (void)hipEventRecord(ev[1], stream);
        return hipGetLastError();
    }
    const size_t need = decode_lds_bytes(channels);
    // (Capping the occupancy so that a batch fills whole rounds of resident workgroups was tried -- extra dynamic
    // LDS -- and lost at every batch size: the recurrence is latency-bound per wave, more waves always overlap more.)
    const size_t lds = need;
    if (lds > 64 * 1024) { // above the default dynamic-LDS limit (more than four channels); per device, so every time
        err = d_phase_cycles
            ? hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    const uint32_t from = recurrence_form >= 0 ? (recurrence_form ? 0u : n_frames)
                                                : vec_shift_from_for(n_frames, n_waves, resident_frames(reinterpret_cast<const void*>(k_decode_frames<false>), n_waves, lds));
    const uint32_t synth_prio = synth_priorities;
    if (ev)
        (void)hipEventRecord(ev[0], stream);
    if (d_phase_cycles)
        hipLaunchKernelGGL(k_decode_frames<true>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames, channels,
            d_pcm_out, d_status, ws, d_phase_cycles, frame_flags, from, synth_prio);
    else
        hipLaunchKernelGGL(k_decode_frames<false>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames, channels,
            d_pcm_out, d_status, ws, d_phase_cycles, frame_flags, from, synth_prio);
    if (ev)
        (void)hipEventRecord(ev[1], stream);
    return hipGetLastError();
}

} // namespace sela
