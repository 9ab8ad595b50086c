// sela_decode32.hip -- k_decode_subframes32: standard subframes (2048 samples) decoded to 32-bit samples (gfx950).
// The any-length route's decoder (sela_capi_generic.hip, sela_generic.hip) offers every chunk to this kernel first.
#include <hip/hip_runtime.h>

#include "sela_device.h"
#include "sela_generic.h"

namespace sela {

#include "sela_decode_core.inc"

// ---- standard subframes, 32-bit samples out: the first try of the any-length route's decoder ------------------------------------
// frame::FrameDecoder returns what the synthesis produces, untruncated (src/frame/frame_decoder.cpp:24-25,64-71), so the
// class -- and sela_hip_decode_i32 behind it -- cannot use k_decode_frames, whose samples pass through int16.  But nearly
// every stream it is handed is an encoder's: subframes of 2048 samples that fit the parser's plan.  For those this kernel
// runs the very parse and synthesis of k_decode_frames, one wave per subframe, and leaves the 32-bit values where
// k_generic_decode would have left them (dec_ws, info: k_generic_combine follows either).  It takes a subframe or leaves it
// alone: anything it would have to judge -- a header walk_headers() refuses, a stream beyond the plan, any flag from the parse
// or the step-up -- is counted in status[2], and the caller then runs the whole chunk on k_generic_decode, which knows what
// the reference does with such streams.
template <bool kVecShift>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_decode_subframes32(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels, uint32_t stride,
    int32_t* __restrict__ dec_ws /* [n_frames][channels][stride] by subframe position */, GenericSubInfo* __restrict__ info, uint32_t* __restrict__ status)
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
    SubHeader hd;
    hd.ok = false;
    if ((at & 3) == 0 && stride >= (uint32_t)kBlock)
        hd = walk_headers(fb, fbytes, c, channels);
    const bool mine = hd.ok && hd.cw + 2 + hd.rw <= (uint32_t)kStreamCap;
    uint32_t flags = 0;
    if (mine) {
        const uint32_t nw = hd.cw + 2 + hd.rw;
        const uint32_t* const gw = reinterpret_cast<const uint32_t*>(fb + hd.p + 4); // the subframe's aligned words
        for (uint32_t w = lane; w < nw + kStreamMargin; w += kWave) // the start bitmap
            sl.marks[w] = 0;
        wave_sync();
        ParseProfile pp;
        const StreamWords sw = { gw, nw };
        flags |= parse_subframe<false>(sw, sl.marks, sl.pos, reinterpret_cast<uint16_t*>(&scratch.t), coef_values(&scratch), hd.cw, hd.rw, hd.ck, hd.rk, hd.order,
            lane, pp);
        SynthTables* const tables = &scratch.t;
        const uint32_t order = hd.order;
        const int32_t q_lo = (uint32_t)lane < order ? coef_values(&scratch)[lane] : 0, q_hi = (uint32_t)lane + 64 < order ? coef_values(&scratch)[lane + 64] : 0;
        wave_sync();
        const double k_lo = (uint32_t)lane < order ? (order <= 1 ? 0.0 : dequant(lane, q_lo, flags)) : 0.0;
        const double k_hi = (uint32_t)lane + 64 < order ? dequant(lane + 64, q_hi, flags) : 0.0;
        step_up_regs(k_lo, k_hi, tables->a, (int)order, lane, flags);
        const bool fits24 = build_synth_table(tables->a, tables->tab, (int)order, lane);
        SynthOut<true> out32;
        out32.samples = dec_ws + (size_t)sub * stride;
        synthesize_by_order<kVecShift, true>(order, gw, nw, hd.rk, sl.pos, nullptr, tables->tab, fits24, lane, out32);
    }
    flags = wave_or(flags);
    if (lane == 0) {
        GenericSubInfo si;
        si.channel = (uint8_t)hd.channel, si.type = (uint8_t)hd.type, si.parent = (uint8_t)hd.parent, si.n = (uint32_t)kBlock;
        si.ok = mine && flags == 0 ? 1 : 0;
        if (!si.ok) {
            si.channel = si.type = si.parent = 0, si.n = 0;
            atomicAdd(&status[2], 1u);
        }
        info[sub] = si;
    }
}

hipError_t launch_decode_subframes32(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels,
    uint32_t stride, int32_t* d_dec, GenericSubInfo* d_info, uint32_t* d_status, hipStream_t stream)
{
    const uint64_t subs = (uint64_t)n_frames * channels;
    if (subs == 0)
        return hipSuccess;
    if (subs <= kLonelyWaves) // (the recurrence's form for waves that have their SIMD nearly to themselves, vec_shift_from_for)
        hipLaunchKernelGGL(k_decode_subframes32<true>, dim3((uint32_t)subs), dim3(64), 0, stream, d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec,
            d_info, d_status);
    else
        hipLaunchKernelGGL(k_decode_subframes32<false>, dim3((uint32_t)subs), dim3(64), 0, stream, d_frames, d_frame_offsets, base_bytes, n_frames, channels, stride, d_dec,
            d_info, d_status);
    return hipGetLastError();
}

} // namespace sela
