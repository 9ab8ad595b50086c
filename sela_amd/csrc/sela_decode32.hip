// sela_decode32.hip -- k_decode_subframes32: subframes of ANY length decoded to 32-bit samples on the fast decoder's machinery
// (gfx950): the lane-parallel Rice parse and the tuned synthesis of sela_decode_core.inc with the length a run-time value.
// frame::FrameDecoder / sela_hip_decode_i32 and sela_hip_decode on streams whose frames are not 2048 samples long come here
// (sela_capi_generic.hip); k_generic_decode (sela_generic.hip, a serial walk) is only the judge of streams this kernel will not
// touch: misaligned or malformed frames, streams that run dry, coefficients outside the tables.
#include <hip/hip_runtime.h>

#include "sela_device.h"
#include "sela_generic.h"

namespace sela {

#include "sela_decode_core.inc"

// ---- one SEGMENT of a Rice stream, parsed across the lanes (any stream length, any number of values) --------------------------
// parse_subframe (sela_decode_core.inc) knows one shape: 2048 residues whose words fit one bitmap.  A stream of any length is
// cut into segments instead: up to kSegWords aligned words and up to kSegValues codewords, each parsed by the same three walks
// (phase A: every lane marks the codeword starts of its zone; phase B: on through the following zones until standing on a
// later lane's start; the true chain by pointer doubling; pass 2: the chain's lanes list the starts from their true entries)
// -- but with ONE stream per call, an entry anywhere in the segment's first word, a run-time number of codewords wanted, and
// a LIMIT: a codeword that starts at or behind it belongs to the next segment (the walks end there as they end at a stream's
// end).  A codeword that starts in front of the limit is this segment's however far it reaches.  Returns how many starts were
// listed (pos_out[0 .. found), relative to the segment's first word; found >= 1 whenever the entry lies in front of the limit),
// the bit behind the last of them -- the next segment's entry -- and whether one of them reaches beyond the stream's end.
constexpr int kSegWords = kStreamCap;                // words of one segment's start bitmap
constexpr uint32_t kSegValues = (uint32_t)kBlock;    // codewords listed per segment: their positions overwrite the bitmap (DecSubframeLds)
struct Segment {
    uint32_t found, next;
    bool overrun;
};

__device__ __attribute__((noinline)) Segment parse_segment(const uint32_t* seg_words /* the segment's first word */, uint32_t words_left /* of the subframe from there: reads beyond are zero */,
    uint32_t entry /* 0..31: the first codeword's bit */, uint32_t n_seg_words /* >= 1 */, uint32_t stream_end /* bit, relative to the segment's first word */, uint32_t k,
    uint32_t need /* 1 .. kSegValues */, uint32_t* marks /* zeroed: n_seg_words + kStreamMargin words */, uint16_t* pos_out, uint16_t* scratch16, int lane)
{
    const StreamWords sw = { seg_words, words_left };
    const __amdgpu_buffer_rsrc_t rs = stream_rsrc(sw);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    const uint32_t W = n_seg_words;
    const uint32_t zr = (W + (uint32_t)kWave - 1) / (uint32_t)kWave; // words per zone
    const uint32_t first_word = min((uint32_t)lane * zr, W), end_word = min((uint32_t)(lane + 1) * zr, W);
    const uint32_t limit = min(32 * W, stream_end);
    const uint32_t last_mark_word = W + kStreamMargin - 1;
    const uint32_t zone_end = min(32 * end_word, limit);

    // ---- phase A: own zone, marking every codeword start ----
    uint32_t pos = lane == 0 ? entry : 32 * first_word;
    bool in_run = false;
    while (__any(pos < zone_end)) {
        const bool act = pos < zone_end;
        uint32_t off[5];
        bool simple;
        analyse4(rs, pos, k, off, simple);
        if (!__any(act && (in_run || !simple))) {
            uint32_t adv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + off[j];
                const bool a = act && p < zone_end;
                atomicOr(&marks[a ? p >> 5 : 0u], a ? 1u << (p & 31) : 0u);
                adv = a ? off[j + 1] : adv;
            }
            pos += adv;
        } else {
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

    // ---- phase B: on through the following zones until standing on a later lane's start, or at the limit ----
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
                const bool ended = walking && p >= limit;
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
            const bool ended = at_start && pos >= limit;
            const bool met = at_start && !ended && ((mk >> (pos & 31)) & 1u);
            merged = ended ? kEndOfStream : (met ? pos : merged);
            walking = walking && !ended && !met;
            n_cont += (at_start && walking) ? 1u : 0u;
            pos += walking ? adv : 0u;
            in_run = walking ? full : in_run;
        }
    }

    // ---- the chain of lanes the true trajectory runs through (pointer doubling, as in parse_subframe) ----
    uint32_t succ = merged != kEndOfStream ? (merged >> 5) / zr : 64u;
    succ = (succ > (uint32_t)lane && succ < 64u) ? succ : 64u; // (always a zone further on; keeps the orbit finite whatever the stream holds)
    uint8_t* const flag = reinterpret_cast<uint8_t*>(scratch16) + 512;
    uint32_t* const entry_of = reinterpret_cast<uint32_t*>(scratch16) + 160;
    flag[lane] = 0;
    bool on_chain = lane == 0;
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
    if (on_chain && succ < 64)
        entry_of[succ] = merged;
    wave_sync();
    uint32_t e_true = kEndOfStream;
    if (on_chain)
        e_true = lane == 0 ? entry : entry_of[lane];
    wave_sync();
    uint32_t count = 0;
    {
        const uint32_t we = e_true >> 5;
        for (uint32_t j = 0; j < zr; j++) {
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
    const uint32_t idx = wave_exclusive_scan(count, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(idx + count), kWave - 1);
    Segment seg;
    seg.found = min(total, need);
    const uint32_t taken = idx < need ? min(count, need - idx) : 0u;
    uint32_t remaining = taken;
    wave_sync(); // every lane has read the bitmap: the positions may overwrite it

    // ---- pass 2: list the starts, every chain lane from its true entry ----
    uint16_t* out = pos_out + idx;
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
    // the lane that listed the segment's last codeword stands behind it
    const unsigned long long last = __ballot(taken != 0 && idx + taken == seg.found);
    seg.next = last ? (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)__builtin_ctzll(last)) : entry;
    seg.overrun = __any(overrun);
    wave_sync();
    return seg;
}

// rice::RiceDecoder::process (src/rice/rice_decoder.cpp:21-61) for `count` values of the stream that occupies bits
// [first_bit, stream_end) of the subframe's aligned words, segment by segment; value i goes to out[i] (LDS or global memory).
// Returns SELA_HIP_FLAG_RICE_OVERRUN when a codeword reaches beyond the stream's end (the rest is then not defined).
__device__ __forceinline__ uint32_t parse_stream_segments(const uint32_t* gw, uint32_t nw, uint32_t first_bit, uint32_t stream_end, uint32_t k, uint32_t count,
    uint32_t words_per_value_x256 /* the stream's own average, for the segments' sizes */, DecSubframeLds* sl, uint16_t* scratch16, int32_t* out, int lane)
{
    const uint32_t kmask = k ? (0xFFFFFFFFu >> (32 - k)) : 0u;
    const uint32_t end_word = (stream_end + 31) >> 5;
    uint32_t done = 0, entry = first_bit, boost = 0;
    bool overrun = false;
#pragma unroll 1
    while (done < count) {
        if (entry >= stream_end) { // the stream has run dry: what is missing reads as zero bits
            for (uint32_t i = done + lane; i < count; i += kWave)
                out[i] = 0;
            overrun = true;
            break;
        }
        const uint32_t need = min(count - done, kSegValues);
        const uint32_t w0 = entry >> 5;
        // as many words as `need` codewords of the stream's average length take (+ 1/16 + 2): a segment that finds fewer simply
        // hands the rest to the next one (and makes that one twice as long: a stream whose codewords grow must not be walked
        // a few codewords at a time), one that covers many more than it may list parses them for nothing
        const uint64_t guess = ((uint64_t)need * words_per_value_x256) >> 8;
        const uint64_t wanted = (guess + (guess >> 4) + 2) << boost; // (< 2^28 x 2^12: no wrap in 64 bits)
        const uint32_t W = max(1u, min(min((uint32_t)kSegWords, end_word - w0), (uint32_t)min(wanted, (uint64_t)kSegWords)));
        for (uint32_t w = lane; w < W + kStreamMargin; w += kWave)
            sl->marks[w] = 0;
        wave_sync();
        const Segment seg = parse_segment(gw + w0, nw - w0, entry & 31, W, stream_end - 32 * w0, k, need, sl->marks, sl->pos, scratch16, lane);
        const StreamWords sw = { gw + w0, nw - w0 };
        const __amdgpu_buffer_rsrc_t rs = stream_rsrc(sw);
        for (uint32_t i0 = 0; i0 < seg.found; i0 += kWave) {
            const bool valid = i0 + lane < seg.found;
            const int32_t v = decode_at(rs, valid ? sl->pos[i0 + lane] : 0u, k, kmask, valid);
            if (valid)
                out[done + i0 + lane] = v;
        }
        wave_sync();
        done += seg.found;
        entry = 32 * w0 + seg.next;
        overrun |= seg.overrun;
        boost = (seg.found < need && boost < 12) ? boost + 1 : boost;
    }
    return overrun ? (uint32_t)SELA_HIP_FLAG_RICE_OVERRUN : 0u;
}

// ---- subframe header walk for any length (layout of src/file/sela_file.cpp:58-91): what k_generic_decode accepts, of a frame
// of whole words at a word-aligned place --------------------------------------------------------------------------------------
struct AnyHeader {
    bool ok;
    uint32_t p, channel, type, parent, ck, cw, order, rk, rw, n;
};
__device__ inline AnyHeader walk_headers_any(const uint8_t* fb, uint64_t fbytes, uint32_t c)
{
    AnyHeader h;
    h.ok = fbytes >= 4 && fbytes < 0x7FFFFFFFull && (fbytes & 3) == 0 && reinterpret_cast<const uint32_t*>(fb)[0] == SELA_SYNC_WORD;
    uint64_t p = 4;
    h.channel = h.type = h.parent = h.ck = h.cw = h.order = h.rk = h.rw = h.n = 0;
    for (uint32_t i = 0; h.ok && i <= c; i++) {
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
        h.rk = h2 >> 24, h.rw = h3 & 0xFFFF, h.n = h3 >> 16;
        const uint64_t next = p + 12 + 4 * ((uint64_t)h.cw + h.rw);
        if (next > fbytes) {
            h.ok = false;
            break;
        }
        if (i < c)
            p = next;
    }
    h.ok = h.ok && h.order <= (uint32_t)kMaxOrder && h.ck < 32 && h.rk < 32;
    h.p = (uint32_t)p;
    return h;
}

// ---- subframes of any length, 32-bit samples out ---------------------------------------------------------------------------------
// frame::FrameDecoder returns what the synthesis produces, untruncated (src/frame/frame_decoder.cpp:24-25,64-71), at each
// subframe's own samplesPerChannel, so the class -- and sela_hip_decode_i32 behind it -- cannot use k_decode_frames, whose
// samples pass through int16 and whose plan is 2048 samples.  One wave per subframe here:
//   * a subframe of at most 2048 samples whose words fit the parser's plan (every subframe the reference's CLI writes, and every
//     shorter one) runs the very parse and synthesis of k_decode_frames -- positions in LDS, residues decoded just in time;
//   * a longer one, or a stream beyond the plan, is parsed segment by segment (parse_segment above), the residues parked
//     where the samples will lie (dec_ws), and synthesised in place by the same recurrence with the length a run-time value.
// The samples land where k_generic_decode would have left them (dec_ws, info: k_generic_combine follows either).  The kernel
// takes a subframe or leaves it alone: anything it would have to judge -- a frame that is not whole words at an aligned
// place, a header the walk refuses, a stream that runs dry, a coefficient outside the tables or beyond int64 -- is counted in
// status[2], and the caller then runs the whole chunk on k_generic_decode, which knows what the reference does with such streams.
template <bool kVecShift>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_decode_subframes32(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels, uint32_t stride,
    int32_t* __restrict__ dec_ws /* [n_frames][channels][stride] by subframe position */, GenericSubInfo* __restrict__ info, uint32_t* __restrict__ status,
    uint32_t standard_path /* 0: every subframe by segments (tests) */)
{
    __shared__ __attribute__((aligned(16))) DecSubframeLds sl;
    __shared__ DecWaveScratch scratch;
    const uint32_t sub = blockIdx.x;
    if (sub >= n_frames * channels)
        return;
    const int lane = threadIdx.x;
    const uint32_t f = sub / channels, c = sub % channels;
    const uint64_t at = frame_offsets[f] - base_bytes;
    const uint8_t* const fb = frames + at;
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    AnyHeader hd;
    hd.ok = false;
    hd.channel = hd.type = hd.parent = hd.n = 0;
    if ((at & 3) == 0)
        hd = walk_headers_any(fb, fbytes, c);
    const bool mine = hd.ok && hd.n <= stride && hd.n != 0 && hd.n > hd.order; // (a subframe not longer than its order: the reference writes past its vector -- the judge's)
    uint32_t flags = 0;
    if (mine) {
        const uint32_t nw = hd.cw + 2 + hd.rw;
        const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4); // the subframe's aligned words
        SynthTables* const tables = &scratch.t;
        const uint32_t order = hd.order;
        int32_t* const samples = dec_ws + (size_t)sub * stride;
        // (in one piece: up to 2048 samples whose words fit the parser's plan -- its positions are one 16-bit word per codeword)
        const bool standard = standard_path && hd.n <= (uint32_t)kBlock && nw <= (uint32_t)kStreamCap;
        if (standard) {
            for (uint32_t w = lane; w < nw + kStreamMargin; w += kWave) // the start bitmap
                sl.marks[w] = 0;
            wave_sync();
            ParseProfile pp;
            const StreamWords sw = { gw, nw };
            flags |= parse_subframe<false>(sw, sl.marks, sl.pos, reinterpret_cast<uint16_t*>(tables), coef_values(&scratch), hd.cw, hd.rw, hd.ck, hd.rk, order, lane, pp, hd.n);
        } else {
            // the coefficients (src/frame/frame_decoder.cpp:19-23): bits [24, 24 + 32 cw) of the aligned words
            if (order)
                flags |= parse_stream_segments(gw, nw, 24, 24 + 32 * hd.cw, hd.ck, order, (256 * hd.cw + order - 1) / order + 1, &sl, reinterpret_cast<uint16_t*>(tables),
                    coef_values(&scratch), lane);
            // the residues (:24-29): bits [32 (cw + 2), 32 (cw + 2 + rw)), parked where the samples will lie
            if (hd.n)
                flags |= parse_stream_segments(gw, nw, 32 * (hd.cw + 2), 32 * nw, hd.rk, hd.n, (256 * hd.rw + hd.n - 1) / hd.n + 1, &sl, reinterpret_cast<uint16_t*>(tables),
                    samples, lane);
            // the lanes read each other's residues back: the stores have left the CU, nothing older is served from its vector cache
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int32_t q_lo = (uint32_t)lane < order ? coef_values(&scratch)[lane] : 0, q_hi = (uint32_t)lane + 64 < order ? coef_values(&scratch)[lane + 64] : 0;
        wave_sync();
        const double k_lo = (uint32_t)lane < order ? (order <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
        const double k_hi = (uint32_t)lane + 64 < order ? dequant(lane + 64, q_hi, flags) : 0.0;
        step_up_regs(k_lo, k_hi, tables->a, (int)order, lane, flags);
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        SynthOut<true> out32;
        out32.samples = samples;
        out32.n = hd.n;
        flags = wave_or(flags);
        if (flags == 0) // (a stream in trouble is the other kernel's)
            synthesize_by_order<kVecShift, true>(order, gw, nw, hd.rk, sl.pos, standard ? nullptr : samples, tables->tab, fits24, lane, out32);
    }
    flags = wave_or(flags);
    if (lane == 0) {
        GenericSubInfo si;
        si.channel = (uint8_t)hd.channel, si.type = (uint8_t)hd.type, si.parent = (uint8_t)hd.parent, si.n = hd.n;
        si.ok = mine && flags == 0 ? 1 : 0;
        if (!si.ok) {
            si.channel = si.type = si.parent = 0, si.n = 0;
            atomicAdd(&status[2], 1u);
        } else if (!(hd.n <= (uint32_t)kBlock && hd.cw + 2 + hd.rw <= (uint32_t)kStreamCap && standard_path))
            atomicAdd(&status[3], 1u); // (subframes that went by segments: tests and the probes ask)
        info[sub] = si;
    }
}

// ---- lpc::SampleGenerator on its own for any length (src/lpc/sample_generator.cpp:11-39; src/include/lpc.hpp:106-117): order +
// quantised reflection coefficients + n residues -> n samples, one wave per block, through the very dequantisation, step-up,
// table and recurrence the frame kernels run; the samples go out as the 32-bit values the reference's class returns.
__global__ __launch_bounds__(64) void k_lpc_decode_any(const int32_t* __restrict__ order_in, const int32_t* __restrict__ q_in, const int32_t* __restrict__ residues,
    uint32_t n_blocks, uint32_t n, int32_t* __restrict__ samples_out, int64_t* __restrict__ coefs_out /* [block][101] or null */, uint32_t* __restrict__ status)
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
    if (samples_out && (n == 0 || n <= order)) // src/lpc/sample_generator.cpp:14-22 writes samples[0] and samples[1 .. order]: past its vector
        flags |= SELA_HIP_FLAG_SHORT_BLOCK;
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
        SynthOut<true> out32;
        out32.samples = samples_out + (size_t)b * n;
        out32.n = n;
        synthesize_by_order<false, true>(order, nullptr, 0, 0, nullptr, residues + (size_t)b * n, tables->tab, fits24, lane, out32);
    }
    flags = wave_or(flags);
    if (lane == 0 && flags)
        atomicOr(&status[0], flags);
}

hipError_t launch_decode_subframes32(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels,
    uint32_t stride, int32_t* d_dec, GenericSubInfo* d_info, uint32_t* d_status, bool standard_path, hipStream_t stream)
{
    const uint64_t subs = (uint64_t)n_frames * channels;
    if (subs == 0)
        return hipSuccess;
    if (subs <= kLonelyWaves) // (the recurrence's form for waves that have their SIMD nearly to themselves, vec_shift_from_for)
        hipLaunchKernelGGL(k_decode_subframes32<true>, dim3((uint32_t)subs), dim3(64), 0, stream, d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec,
            d_info, d_status, standard_path ? 1u : 0u);
    else
        hipLaunchKernelGGL(k_decode_subframes32<false>, dim3((uint32_t)subs), dim3(64), 0, stream, d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec,
            d_info, d_status, standard_path ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_lpc_decode_any(const int32_t* d_order, const int32_t* d_q, const int32_t* d_residues, uint32_t n_blocks, uint32_t n, int32_t* d_samples,
    int64_t* d_coefs, uint32_t* d_status, hipStream_t stream)
{
    if (n_blocks)
        hipLaunchKernelGGL(k_lpc_decode_any, dim3(n_blocks), dim3(64), 0, stream, d_order, d_q, d_residues, n_blocks, n, d_samples, d_coefs, d_status);
    return hipGetLastError();
}

} // namespace sela
