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

#include "sela_decode_core.inc" // the parser, the synthesis, the header walk: shared with sela_decode32.hip

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
        SynthOut<true> out32;
        out32.samples = samples_out + (size_t)b * kBlock;
        out32.n = (uint32_t)kBlock;
        synthesize_by_order<false, true>(order, nullptr, 0, 0, nullptr, residues + (size_t)b * kBlock, tables->tab, fits24, lane, out32);
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
