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
constexpr int kStageWords = kCoefWordsCap * 2 + kResWordsCap + 8; // staged Rice words of one subframe

struct DecodeWaveLds {
    uint32_t words[kStageWords]; // [0,64) coefficient words, [64, ...) residue words
    double k[104];
    double t[104];
    int64_t a[104];
    int32_t q[256];
};

// Serial Rice parse by lane 0 (src/rice/rice_decoder.cpp:21-44); `out` is in LDS.
__device__ inline void rice_decode_serial(const uint32_t* words, uint32_t nwords, uint32_t n, uint32_t k, int32_t* out,
    int lane, uint32_t& flags)
{
    if (lane == 0) {
        const uint64_t total = (uint64_t)nwords * 32;
        uint64_t pos = 0;
        bool overrun = false;
        for (uint32_t c = 0; c < n; c++) {
            uint32_t ones = 0;
            for (;;) { // count ones up to the terminating zero
                if (pos >= total) {
                    overrun = true;
                    break;
                }
                const uint32_t w = words[pos >> 5] >> (pos & 31);
                const uint32_t avail = 32 - (uint32_t)(pos & 31);
                const uint32_t run = (uint32_t)__builtin_ctzll((uint64_t)(~w) | (1ull << 32)); // trailing ones of w, <= 32
                const uint32_t take = run < avail ? run : avail;
                ones += take;
                pos += take;
                if (take < avail)
                    break;
            }
            pos++; // the zero
            uint64_t u = (uint64_t)(uint32_t)(ones << k); // uint32 shift, src/rice/rice_decoder.cpp:35
            for (uint32_t i = 1; i <= k; i++) {
                uint32_t bit = 0;
                if (pos < total)
                    bit = (words[pos >> 5] >> (pos & 31)) & 1u;
                else
                    overrun = true;
                u |= (uint64_t)bit << (k - i);
                pos++;
            }
            out[c] = unzigzag(u);
        }
        if (overrun)
            flags |= SELA_HIP_FLAG_RICE_OVERRUN;
    }
    wave_sync();
}

// lpc::SampleGenerator::generateSamples (src/lpc/sample_generator.cpp:11-30), in place over the
// residues in LDS.  Transposed direct form: after sample s_i is known every tap p adds a[p+1]*s_i to
// the partial sum that will be complete p+1 steps later:
//     z_p <- z_{p+1} + a[p+1] * s_i ,        P_{i+1} = z_0
//     s_{i+1} = r_{i+1} - (int32)((2^34 - P_{i+1}) >> 35)
// which reproduces both reference loops (warm-up: taps beyond the block start simply have not
// received anything yet).  Lane L holds taps p = 2L (even) and 2L+1 (odd); 50 lanes cover order 100.
__device__ inline void synthesize(int32_t* rs, const int64_t* a, int order, int lane)
{
    const uint64_t a_odd = (2 * lane + 1 <= order) ? (uint64_t)a[2 * lane + 1] : 0;  // multiplies into z_{2L}
    const uint64_t a_even = (2 * lane + 2 <= order) ? (uint64_t)a[2 * lane + 2] : 0; // multiplies into z_{2L+1}
    uint64_t z_even = 0, z_odd = 0;
    const uint64_t half = (uint64_t)1 << (SELA_Q_SHIFT - 1);
    for (int base = 0; base < kBlock; base += 64) {
        const int32_t r_chunk = rs[base + lane];
        int32_t s_chunk = 0;
#pragma unroll 8
        for (int m = 0; m < 64; m++) {
            const int32_t r_i = __builtin_amdgcn_readlane(r_chunk, m);
            // lane 0: z_even == P_i
            const int32_t pred = (int32_t)((int64_t)(half - z_even) >> SELA_Q_SHIFT);
            const int32_t s_i = __builtin_amdgcn_readfirstlane((int32_t)((uint32_t)r_i - (uint32_t)pred));
            s_chunk = lane == m ? s_i : s_chunk;
            const uint64_t sx = (uint64_t)(int64_t)s_i;
            const uint64_t from_next = wave_shl1((uint64_t)0, z_even); // z_{2L+2} (old)
            z_even = z_odd + a_odd * sx;
            z_odd = from_next + a_even * sx;
        }
        rs[base + lane] = s_chunk;
    }
    wave_sync();
}

__global__ __launch_bounds__(kDecMaxWaves * 64) void k_decode_frames(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* __restrict__ pcm_out,
    uint32_t* __restrict__ status)
{
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
            && cw <= 2u * kCoefWordsCap && rw <= (uint32_t)kResWordsCap && ck < 32 && rk < 32 && type <= 1
            && (type == 0 || parent < channels);
        if (!ok) {
            flags |= SELA_HIP_FLAG_BAD_FRAME;
            break;
        }
        if ((int)(c % (uint32_t)n_waves) == wave) {
            // stage the Rice words: coefficient words sit 3 bytes off alignment (funnel shift),
            // residue words are aligned again.
            const uint32_t* al = reinterpret_cast<const uint32_t*>(h + 4); // bytes 4..7 of the subframe
            for (uint32_t i = lane; i < cw; i += 64)
                wl->words[i] = (al[i] >> 24) | (al[i + 1] << 8);
            const uint32_t* rwp = reinterpret_cast<const uint32_t*>(h2 + 5);
            for (uint32_t i = lane; i < rw; i += 64)
                wl->words[2 * kCoefWordsCap + i] = rwp[i];
            wave_sync();

            int32_t* dst = samples + (size_t)channel * kBlock;
            rice_decode_serial(wl->words, cw, order, ck, wl->q, lane, flags);
            rice_decode_serial(wl->words + 2 * kCoefWordsCap, rw, n, rk, dst, lane, flags);

            // dequantise (src/lpc/linear_predictor.cpp:16-28)
            for (uint32_t i = lane; i < order; i += 64)
                wl->k[i] = order <= 1 ? 0.0 : dequant((int)i, wl->q[i], flags);
            wave_sync();
            step_up(wl->k, wl->t, wl->a, (int)order, lane, flags);
            synthesize(dst, wl->a, (int)order, lane);
            if (lane == 0)
                sub_info[channel] = type | (parent << 8);
        }
        p = next;
    }
    __syncthreads();

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
    int16_t* d_pcm_out, uint32_t* d_status, hipStream_t stream, hipEvent_t* ev /* 2 events or nullptr */)
{
    hipError_t err = hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
    if (err != hipSuccess || n_frames == 0)
        return err;
    const int n_waves = decode_waves(channels);
    const size_t lds = decode_lds_bytes(channels, n_waves);
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_frames), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess)
        return err;
    if (ev)
        (void)hipEventRecord(ev[0], stream);
    hipLaunchKernelGGL(k_decode_frames, dim3(n_frames), dim3(n_waves * 64), lds, stream, d_frames, d_frame_offsets, n_frames,
        channels, d_pcm_out, d_status);
    if (ev)
        (void)hipEventRecord(ev[1], stream);
    return hipGetLastError();
}

} // namespace sela
