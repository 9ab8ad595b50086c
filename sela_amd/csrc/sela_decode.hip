// sela_decode.hip -- MI355X (gfx950) decoder kernel of the SELA frame path.
//
//   k_decode_frames   one WORKGROUP per frame, one WAVE per subframe (channel):
//       parse the subframe header               (layout of src/file/sela_file.cpp:58-91)
//       rice::RiceDecoder x2                    (src/rice/rice_decoder.cpp:11-61)
//       dequantise + step-up                    (src/lpc/linear_predictor.cpp:16-61)
//       lpc::SampleGenerator::generateSamples   (src/lpc/sample_generator.cpp:11-30)
//     then, after a workgroup barrier, frame::FrameDecoder's second pass
//       out[ch] = parent - difference           (src/frame/frame_decoder.cpp:40-69)
//     and the int16 interleave of               (src/file/wav_file.cpp:244-257)
//     written coalesced to HBM.
//
// The synthesis filter is a true serial recurrence (every sample is rounded before it feeds the
// next one), so the wave runs it as a transposed-form systolic array: lane L carries the partial
// sums of taps 2L+1 and 2L+2, the new sample is broadcast from lane 0, and the partial sums move one
// tap per step with a single DPP shift -- integer wrap-around arithmetic, any evaluation order exact.
#include "sela_device.h"

namespace sela {

constexpr int kDecMaxWaves = 8;
constexpr int kStageWords = kCoefWordsCap * 2 + kResWordsCap + 72; // + zero tail for the chunked parser // staged Rice words of one subframe

struct DecodeWaveLds {
    uint32_t words[kStageWords]; // [0,64) coefficient words, [64, ...) residue words
    double k[104];
    double t[104];
    int64_t a[104];
    int32_t q[256];
};

// ---- wave-parallel Golomb-Rice decoder ---------------------------------------------------------
// rice::RiceDecoder::generateDecodedUnsignedInts (src/rice/rice_decoder.cpp:21-44) is a serial bit
// parse: where codeword i+1 starts depends on codeword i.  The wave cuts the stream into 64
// word-aligned chunks (one per lane) and resolves the dependency with a scan over *parser states*:
//
//   state at a chunk boundary:  R_m (m = 0..k)  "m more remainder bits to skip, then a codeword
//                                                starts"  (R_0 = a codeword starts right here)
//                               M   (= k+1)     "inside a unary run: skip to the first zero, skip
//                                                k remainder bits, then a codeword starts"
//   phase 1  every lane parses its chunk once for EVERY entry state (k+2 parses in lockstep, all
//            independent -> the LDS latency of one hides behind the others) and records
//            map[entry] = (exit state, number of codewords that START in the chunk)
//   walk     the 64 maps are composed from lane 0 (entry R_0, value index 0): 64 dependent LDS
//            reads give every lane its true entry state and the index of its first codeword
//   phase 2  every lane decodes the codewords that start in its chunk (the last one may run past
//            the chunk end) straight to their final positions.
//
// Bit t of the stream is bit t%32 of word t/32 (LSB first); the k remainder bits are MSB first,
// i.e. bit-reversed in stream order.  `words` must be readable (zero) for 3 words past its end.
__device__ __forceinline__ uint64_t bit_window(const uint32_t* words, uint32_t pos)
{
    const uint32_t w = pos >> 5;
    return (((uint64_t)words[w + 1] << 32) | words[w]) >> (pos & 31); // >= 33 valid bits
}

// first zero bit at or after pos, capped at `limit` (a multiple of 32 inside the zero padding)
__device__ __forceinline__ uint32_t next_zero(const uint32_t* words, uint32_t pos, uint32_t limit)
{
    for (;;) {
        const uint32_t lo = (uint32_t)bit_window(words, pos);
        if (lo != 0xFFFFFFFFu)
            return pos + (uint32_t)__builtin_ctz(~lo);
        pos += 32;
        if (pos >= limit)
            return limit;
    }
}

constexpr int kMaxStates = SELA_MAX_RICE_PARAM + 2; // R_0..R_k, M with k <= 19 (header values >= 20 are rejected)

__device__ inline void rice_decode_wave(const uint32_t* words, uint32_t nwords, uint32_t n, uint32_t k, int32_t* out,
    uint32_t* maps /* 64 * kMaxStates words of LDS scratch */, int lane, uint32_t& flags, long long* tm = nullptr)
{
    if (n == 0)
        return;
    const uint32_t chunk_words = (nwords + 63) / 64 ? (nwords + 63) / 64 : 1;
    const uint32_t chunk_bits = chunk_words * 32;
    const uint32_t limit = 64 * chunk_bits; // every lane's chunk lies below this; words are zero beyond nwords
    const uint32_t c0 = (uint32_t)lane * chunk_bits, c1 = c0 + chunk_bits;
    const uint32_t n_states = k + 2, state_m = k + 1;

    // ---- phase 1: (exit state, starts) for every entry state ------------------------------------
    // An entry state is only possible if the bits in front of the chunk agree with it: R_m needs the
    // terminating zero at bit c0-k-1+m, M needs a one at bit c0-1 -- about half of the parses are
    // never started.  Each round first issues the LDS window reads of ALL live parses, then advances
    // every parse by one codeword (or by 32 bits of a long unary run), so the read latencies overlap.
    {
        uint32_t pos[kMaxStates], cnt[kMaxStates];
        uint32_t active = 0; // bit e: parse e still inside the chunk
        uint32_t fresh = 0;  // bit e: pos[e] is the first bit of a codeword (else: inside a unary run)
        {
            uint32_t feasible;
            if (c0 == 0) {
                feasible = 1u; // the stream starts with a codeword
            } else {
                const uint32_t pre = (uint32_t)bit_window(words, c0 - k - 1); // bit m = stream bit c0-k-1+m
                feasible = ~pre & ((1u << (k + 1)) - 1u);
                if ((pre >> k) & 1u)
                    feasible |= 1u << state_m;
            }
#pragma unroll
            for (int e = 0; e < kMaxStates; e++) {
                pos[e] = (uint32_t)e == state_m ? c0 : c0 + (uint32_t)e;
                cnt[e] = 0;
                if ((uint32_t)e < n_states)
                    maps[lane * kMaxStates + e] = 0;
            }
            active = feasible;
            fresh = feasible & ~(1u << state_m);
        }
        while (__any(active != 0)) {
            uint64_t win[kMaxStates];
#pragma unroll
            for (int e = 0; e < kMaxStates; e++)
                if ((uint32_t)e < n_states && (active & (1u << e)))
                    win[e] = bit_window(words, pos[e] < limit ? pos[e] : limit);
#pragma unroll
            for (int e = 0; e < kMaxStates; e++) {
                if ((uint32_t)e < n_states && (active & (1u << e))) {
                    const uint32_t p = pos[e];
                    const bool is_fresh = (fresh >> e) & 1u;
                    if (p >= c1) { // next codeword starts p - c1 bits into a later chunk / the run goes on
                        maps[lane * kMaxStates + e] = (is_fresh ? p - c1 : state_m) | (cnt[e] << 8);
                        active &= ~(1u << e);
                    } else {
                        if (is_fresh)
                            cnt[e]++;
                        const uint32_t lo = (uint32_t)win[e];
                        if (lo == 0xFFFFFFFFu) { // 32 more ones
                            pos[e] = p + 32;
                            fresh &= ~(1u << e);
                        } else {
                            const uint32_t z = p + (uint32_t)__builtin_ctz(~lo);
                            if (z >= c1) { // the terminating zero belongs to a later chunk
                                maps[lane * kMaxStates + e] = state_m | (cnt[e] << 8);
                                active &= ~(1u << e);
                            } else {
                                pos[e] = z + 1 + k;
                                fresh |= 1u << e;
                            }
                        }
                    }
                }
            }
        }
    }
    wave_sync();
    if (tm)
        tm[0] = clock64();

    // ---- walk: compose the maps from the stream start --------------------------------------------------
    uint32_t my_state = 0, my_first = 0;
    {
        uint32_t state = 0, first = 0; // R_0 at bit 0, value index 0
        for (int l = 0; l < 64; l++) {
            if (l == lane) {
                my_state = state;
                my_first = first;
            }
            const uint32_t m = maps[l * kMaxStates + state];
            state = m & 0xFFu;
            first += m >> 8;
        }
        if (first < n)
            flags |= SELA_HIP_FLAG_RICE_OVERRUN; // the words hold fewer than n codewords
    }
    wave_sync(); // maps may alias `out`
    if (tm)
        tm[1] = clock64();

    // ---- phase 2: decode the codewords that start in this lane's chunk -------------------------------
    {
        uint32_t p;
        if (my_state <= k) {
            p = c0 + my_state;
        } else {
            const uint32_t z = next_zero(words, c0, limit);
            p = z >= c1 ? c1 : z + 1 + k;
        }
        uint32_t idx = my_first;
        const uint32_t kmask = k ? (0xFFFFFFFFu >> (32 - k)) : 0u;
        while (p < c1 && idx < n) {
            uint64_t win = bit_window(words, p);
            uint32_t ones = 0;
            while ((uint32_t)win == 0xFFFFFFFFu) { // long unary run
                ones += 32;
                if (p + ones >= limit)
                    break;
                win = bit_window(words, p + ones);
            }
            const uint32_t t = (uint32_t)win == 0xFFFFFFFFu ? 0u : (uint32_t)__builtin_ctz(~(uint32_t)win);
            ones += t;
            uint32_t field; // the k bits after the zero, in stream order
            if (t + 1 + k <= 32)
                field = (uint32_t)(win >> (t + 1)) & kmask;
            else
                field = (uint32_t)bit_window(words, p + ones + 1) & kmask;
            const uint32_t rem = k ? (__brev(field) >> (32 - k)) : 0u; // MSB first (src/rice/rice_decoder.cpp:37-40)
            const uint64_t u = (uint64_t)(uint32_t)(ones << k) | rem;  // uint32 shift as src/rice/rice_decoder.cpp:35
            out[idx++] = unzigzag(u);
            p += ones + 1 + k;
        }
    }
    wave_sync();
}

// ---- synthesis filter ----------------------------------------------------------------------------------
// lpc::SampleGenerator::generateSamples (src/lpc/sample_generator.cpp:11-30), in place over the
// residues in LDS.  Transposed direct form: once sample s_i is known every tap position p adds
// a[p+1]*s_i to the partial sum that completes p+1 steps later and the partial sums move down one
// position:
//     z_p <- z_{p+1} + a[p+1] * s_i ,        P_{i+1} = z_0 ,
//     s_{i+1} = r_{i+1} - (int32)((2^34 - P_{i+1}) >> 35)
// which reproduces both reference loops (during warm-up the higher positions simply have not
// received anything yet).  Lane L owns positions P*L .. P*L+P-1 (P = 1 for order <= 64, else 2); the
// one-position move between lanes is a DPP wave shift.  The recurrence itself (z_0 -> s_i) runs on
// the scalar unit: s_i stays in an SGPR and feeds the multiply-adds as a scalar operand.
//
// 64x32-bit products in two instructions: a = ah*2^32 + al with al = (int32)a, so
//     a*s mod 2^64 = al*s (v_mad_i64_i32, exact) + ((ah*s mod 2^32) << 32) (v_mad_u64_u32, low half)
// and a position carries the pair (acc1, acc2) with z = acc1 + (acc2 << 32).
template <int P>
__device__ inline void synthesize(int32_t* rs, const int64_t* a, int order, int lane)
{
    int32_t al[P];
    uint32_t ah[P];
#pragma unroll
    for (int h = 0; h < P; h++) {
        const int idx = P * lane + h + 1;
        const int64_t av = idx <= order ? a[idx] : 0;
        al[h] = (int32_t)(uint32_t)(uint64_t)av;
        ah[h] = (uint32_t)((uint64_t)(av - (int64_t)al[h]) >> 32);
    }
    uint64_t acc1[P];
    uint32_t acc2[P];
#pragma unroll
    for (int h = 0; h < P; h++)
        acc1[h] = 0, acc2[h] = 0;
    const uint64_t half = (uint64_t)1 << (SELA_Q_SHIFT - 1);
#pragma unroll 1
    for (int base = 0; base < kBlock; base += 64) {
        const int32_t r_chunk = rs[base + lane];
        int32_t s_chunk = 0;
#pragma unroll
        for (int m = 0; m < 64; m++) {
            // scalar side: P_i sits in lane 0, position 0
            const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)acc1[0]);
            const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(acc1[0] >> 32))
                + (uint32_t)__builtin_amdgcn_readfirstlane((int)acc2[0]);
            const uint64_t pv = ((uint64_t)p_hi << 32) | p_lo;
            const int32_t pred = (int32_t)((int64_t)(half - pv) >> SELA_Q_SHIFT);
            const int32_t r_i = __builtin_amdgcn_readlane(r_chunk, m);
            const int32_t s_i = (int32_t)((uint32_t)r_i - (uint32_t)pred);
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(s_chunk) : "s"(s_i), "n"(m));
            // vector side: move every partial sum down one position and add this sample's products
            const uint64_t in1 = wave_shl1_zero(acc1[0]);
            const uint32_t in2 = wave_shl1_zero(acc2[0]);
#pragma unroll
            for (int h = 0; h < P; h++) {
                const uint64_t up1 = h + 1 < P ? acc1[h + 1 < P ? h + 1 : h] : in1;
                const uint32_t up2 = h + 1 < P ? acc2[h + 1 < P ? h + 1 : h] : in2;
                acc1[h] = (uint64_t)((int64_t)up1 + (int64_t)al[h] * (int64_t)s_i);
                acc2[h] = (uint32_t)((uint64_t)up2 + (uint64_t)ah[h] * (uint32_t)s_i);
            }
        }
        rs[base + lane] = s_chunk;
    }
    wave_sync();
}

// kProf: also write per-phase cycle counts (debug hook sela_hip_debug_phase_buffer; 16 uint64 per subframe).
template <bool kProf>
__global__ __launch_bounds__(kDecMaxWaves * 64) void k_decode_frames(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* __restrict__ pcm_out,
    uint32_t* __restrict__ status, uint64_t* __restrict__ phase_cycles)
{
    long long stamp[8], tm[2] = { 0, 0 };
    for (int i = 0; i < 8; i++)
        stamp[i] = 0;
    uint32_t prof_sub = 0xFFFFFFFFu;
    if (kProf)
        stamp[0] = clock64();
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    // [channels][2048] int32 samples, then one DecodeWaveLds per wave, then per-channel type/parent
    int32_t* const samples = reinterpret_cast<int32_t*>(dyn);
    const int n_waves = blockDim.x / 64;
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    DecodeWaveLds* const wl = reinterpret_cast<DecodeWaveLds*>(dyn + (size_t)channels * kBlock * 4) + wave;
    uint32_t* const sub_info = reinterpret_cast<uint32_t*>(dyn + (size_t)channels * kBlock * 4 + (size_t)n_waves * sizeof(DecodeWaveLds));

    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const uint8_t* fb = frames + frame_offsets[f];
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];
    uint32_t flags = 0;

    for (uint32_t c = threadIdx.x; c < channels; c += blockDim.x)
        sub_info[c] = 0xFFFFFFFFu; // "no subframe delivered this channel"
    for (uint32_t i = threadIdx.x; i < channels * kBlock; i += blockDim.x)
        samples[i] = 0;
    __syncthreads();

    const bool sync_ok = fbytes >= 4 && reinterpret_cast<const uint32_t*>(fb)[0] == SELA_SYNC_WORD;
    if (!sync_ok)
        flags |= SELA_HIP_FLAG_BAD_FRAME;

    // Every wave walks the subframe headers (wave-uniform scalar work) and decodes its share.
    uint64_t p = 4;
    for (uint32_t c = 0; sync_ok && c < channels; c++) {
        if (p + 12 > fbytes) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            break;
        }
        const uint8_t* h = fb + p;
        const uint32_t channel = h[0], type = h[1], parent = h[2], ck = h[3];
        const uint32_t cw = (uint32_t)h[4] | ((uint32_t)h[5] << 8), order = h[6];
        const uint8_t* h2 = h + 7 + 4 * (size_t)cw;
        if (p + 12 + 4 * (uint64_t)cw > fbytes) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            break;
        }
        const uint32_t rk = h2[0];
        const uint32_t rw = (uint32_t)h2[1] | ((uint32_t)h2[2] << 8), n = (uint32_t)h2[3] | ((uint32_t)h2[4] << 8);
        const uint64_t next = p + 12 + 4 * ((uint64_t)cw + rw);
        const bool ok = next <= fbytes && channel < channels && order <= (uint32_t)kMaxOrder && n == (uint32_t)kBlock
            && cw <= 2u * kCoefWordsCap && rw <= (uint32_t)kResWordsCap && ck < SELA_MAX_RICE_PARAM && rk < SELA_MAX_RICE_PARAM && type <= 1
            && (type == 0 || parent < channels);
        if (!ok) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            break;
        }
        if ((int)(c % (uint32_t)n_waves) == wave) {
            // stage the Rice words: coefficient words sit 3 bytes off alignment (funnel shift),
            // residue words are aligned again.
            const uint32_t* al = reinterpret_cast<const uint32_t*>(h + 4); // bytes 4..7 of the subframe
            for (uint32_t i = lane; i < 2u * kCoefWordsCap; i += 64)
                wl->words[i] = i < cw ? (al[i] >> 24) | (al[i + 1] << 8) : 0u;
            const uint32_t* rwp = reinterpret_cast<const uint32_t*>(h2 + 5);
            const uint32_t rw_pad = ((rw + 63) / 64) * 64 + 4; // zero tail: chunks are word-aligned, windows read 1 word ahead
            for (uint32_t i = lane; i < rw_pad; i += 64)
                wl->words[2 * kCoefWordsCap + i] = i < rw ? rwp[i] : 0u;
            wave_sync();
            if (kProf)
                stamp[1] = clock64(), prof_sub = c;

            int32_t* dst = samples + (size_t)channel * kBlock;
            // the parser-state maps of both Rice streams borrow the output buffer of this channel
            uint32_t* maps = reinterpret_cast<uint32_t*>(dst);
            rice_decode_wave(wl->words, cw, order, ck, wl->q, maps, lane, flags);
            if (kProf)
                stamp[2] = clock64();
            rice_decode_wave(wl->words + 2 * kCoefWordsCap, rw, n, rk, dst, maps, lane, flags, kProf ? tm : nullptr);
            if (kProf)
                stamp[3] = clock64();

            // dequantise (src/lpc/linear_predictor.cpp:16-28)
            for (uint32_t i = lane; i < order; i += 64)
                wl->k[i] = order <= 1 ? 0.0 : dequant((int)i, wl->q[i], flags);
            wave_sync();
            step_up(wl->k, wl->t, wl->a, (int)order, lane, flags);
            if (kProf)
                stamp[4] = clock64();
            if (order <= 64)
                synthesize<1>(dst, wl->a, (int)order, lane);
            else
                synthesize<2>(dst, wl->a, (int)order, lane);
            if (kProf)
                stamp[5] = clock64();
            if (lane == 0)
                sub_info[channel] = type | (parent << 8);
        }
        p = next;
    }
    __syncthreads();
    if (kProf)
        stamp[6] = clock64();

    // ---- second pass of frame::FrameDecoder + interleave to int16 ------------------------------------
    // dependent channels become parent - difference (parents are independent subframes).
    for (uint32_t i = threadIdx.x; i < (uint32_t)kBlock; i += blockDim.x) {
        for (uint32_t c = 0; c < channels; c++) {
            const uint32_t info = sub_info[c];
            int32_t v = samples[(size_t)c * kBlock + i];
            if (info != 0xFFFFFFFFu && (info & 0xFF) == 1) {
                const uint32_t par = info >> 8;
                v = (int32_t)((uint32_t)samples[(size_t)par * kBlock + i] - (uint32_t)v);
            }
            pcm_out[((size_t)f * kBlock + i) * channels + c] = (int16_t)(uint16_t)v;
        }
    }
    // a dependent subframe whose parent is itself dependent is outside what the reference defines
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < channels; c++) {
            const uint32_t info = sub_info[c];
            if (info == 0xFFFFFFFFu)
                flags |= SELA_HIP_FLAG_BAD_FRAME;
            else if ((info & 0xFF) == 1 && (sub_info[info >> 8] & 0xFF) != 0)
                flags |= SELA_HIP_FLAG_BAD_FRAME;
        }
    }
    for (int m = 32; m >= 1; m >>= 1)
        flags |= (uint32_t)__shfl_xor((int)flags, m, 64);
    if (lane == 0 && flags) {
        atomicOr(&status[0], flags);
        if (wave == 0 && (flags & SELA_HIP_FLAG_BAD_FRAME))
            atomicAdd(&status[1], 1u);
    }
    if (kProf && lane == 0 && prof_sub != 0xFFFFFFFFu) { // (one subframe per wave is reported)
        stamp[7] = clock64();
        for (int i = 0; i < 7; i++)
            phase_cycles[((size_t)f * channels + prof_sub) * 16 + i] = (uint64_t)(stamp[i + 1] - stamp[i]);
        // residue Rice parse split: phase 1 / walk / phase 2
        phase_cycles[((size_t)f * channels + prof_sub) * 16 + 8] = (uint64_t)(tm[0] - stamp[2]);
        phase_cycles[((size_t)f * channels + prof_sub) * 16 + 9] = (uint64_t)(tm[1] - tm[0]);
        phase_cycles[((size_t)f * channels + prof_sub) * 16 + 10] = (uint64_t)(stamp[3] - tm[1]);
    }
}

size_t decode_lds_bytes(uint32_t channels, int n_waves)
{
    return (size_t)channels * kBlock * 4 + (size_t)n_waves * sizeof(DecodeWaveLds) + (size_t)channels * 4 + 16;
}

int decode_waves(uint32_t channels)
{
    return channels < (uint32_t)kDecMaxWaves ? (int)channels : kDecMaxWaves;
}

hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, hipStream_t stream, hipEvent_t* ev /* 2 events or nullptr */, uint64_t* d_phase_cycles)
{
    hipError_t err = hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
    if (err != hipSuccess || n_frames == 0)
        return err;
    const int n_waves = decode_waves(channels);
    const size_t lds = decode_lds_bytes(channels, n_waves);
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess)
        return err;
    if (ev)
        (void)hipEventRecord(ev[0], stream);
    if (d_phase_cycles)
        hipLaunchKernelGGL(k_decode_frames<true>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames,
            channels, d_pcm_out, d_status, d_phase_cycles);
    else
        hipLaunchKernelGGL(k_decode_frames<false>, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames,
            channels, d_pcm_out, d_status, d_phase_cycles);
    if (ev)
        (void)hipEventRecord(ev[1], stream);
    return hipGetLastError();
}

} // namespace sela
