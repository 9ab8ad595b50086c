// sela_decode.hip -- MI355X (gfx950) decoder kernels of the SELA frame path.
//
// Two launches on one stream:
//
//   k_parse_subframes      one LANE per subframe (64 independent bitstreams per wave).
//       rice::RiceDecoder x2 (src/rice/rice_decoder.cpp:11-61) is a serial bit parse -- where
//       codeword i+1 starts depends on codeword i -- so the parallelism used is ACROSS streams:
//       every lane walks to its subframe header (layout of src/file/sela_file.cpp:58-91), then
//       parses its coefficient stream and its 2048-value residue stream out of a 64-bit register
//       bit window.  Residues / quantised coefficients / a descriptor go to a per-subframe slot in
//       the workspace (HBM; L2/MALL-resident at these sizes).
//   k_synthesize_frames    one WORKGROUP per frame, one WAVE per subframe:
//       dequantise + step-up                    (src/lpc/linear_predictor.cpp:16-61)
//       lpc::SampleGenerator::generateSamples   (src/lpc/sample_generator.cpp:11-30)
//     then, after a workgroup barrier, frame::FrameDecoder's second pass
//       out[ch] = parent - difference           (src/frame/frame_decoder.cpp:40-69)
//     and the int16 interleave of               (src/file/wav_file.cpp:244-257), stored coalesced.
//
// The synthesis filter is a true serial recurrence (every sample is rounded before it feeds the
// next one), so a wave runs it as a transposed-form systolic array: the lanes carry the partial
// sums of the taps, the new sample is computed on the scalar unit and broadcast as an SGPR operand,
// and the partial sums move one tap per step with a DPP wave shift -- integer wrap-around
// arithmetic, exact under any evaluation order.
#include "sela_device.h"

namespace sela {

constexpr int kDecMaxWaves = 8;
constexpr int kQStride = 128; // int32 slots per subframe for the quantised coefficients
constexpr int kDecodeChunks = 4; // sample-axis pipeline depth of one decode call (2048 / 4 = 512 values per chunk)

// per-subframe record written by k_parse_subframes
struct SubDesc {
    uint32_t info;  // channel | type << 8 | parent << 16 | order << 24
    uint32_t flags; // SELA_HIP_FLAG_* bits; BAD_FRAME means "do not synthesise"
};

// ---- lane-per-stream Rice parser --------------------------------------------------------------------
// Each of the 64 lanes parses its own bitstream (stream bit t = bit t%32 of word t/32) and only keeps
// a bit position.  The words are staged through an LDS tile, one row of kTileWords words per lane,
// that the whole wave refills cooperatively with coalesced 16-byte loads whenever ANY lane gets close
// to the end of its row.  Control flow is wave-uniform throughout; only data is per lane.
//
// Codewords are decoded four at a time from a 128-bit register window read at the lane's bit
// position (one LDS round trip per four values).  A group falls back to the bit-by-window slow path
// when any lane meets a codeword longer than 31 bits (long unary run).
constexpr int kTileWords = 96;
constexpr int kTileStride = kTileWords + 1; // odd stride: lanes reading the same column hit different banks
constexpr int kTileMargin = 12;             // re-tile when a lane is within this many words of its row end
constexpr int kStageVals = 32;              // decoded values staged per lane before a coalesced store
constexpr int kStageStride = kStageVals + 1;

struct StreamReader {
    const uint32_t* base; // first aligned word of the stream (global)
    uint32_t n_words;     // words in the stream; reads beyond are zero
    uint32_t tile_first;  // stream index of tile column 0
    uint32_t bp;          // bit position of the next unread bit
};

// Refill every lane's row so that it starts at the word holding the lane's next unread bit.
__device__ inline void retile(StreamReader& r, uint32_t* tile, int lane)
{
    const uint32_t new_first = r.bp >> 5;
    const uint64_t my_base = reinterpret_cast<uint64_t>(r.base);
    wave_sync(); // earlier reads of the tile are done
#pragma unroll 4
    for (int i = 0; i < kTileWords / 4; i++) { // one wave-load = 16 bytes per lane; 64 rows x 24 loads in all
        // 24 lanes cover one row of 96 words; rows are dealt out 64 lanes at a time
        const int slot = i * 64 + lane; // 0 .. 64 * 24 - 1
        const int row = slot / (kTileWords / 4);
        const uint32_t c4 = (uint32_t)(slot % (kTileWords / 4)) * 4;
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)my_base, row, 64);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(my_base >> 32), row, 64);
        const uint32_t first = (uint32_t)__shfl((int)new_first, row, 64);
        const uint32_t nw = (uint32_t)__shfl((int)r.n_words, row, 64);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(((uint64_t)hi << 32) | lo);
        const uint32_t idx = first + c4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (idx + 4 <= nw) {
            v = *reinterpret_cast<const uint4*>(src + idx); // dword-aligned 16-byte load
        } else { // row reaches the end of its stream: never touch memory past it
            if (idx < nw)
                v.x = src[idx];
            if (idx + 1 < nw)
                v.y = src[idx + 1];
            if (idx + 2 < nw)
                v.z = src[idx + 2];
        }
        uint32_t* dst = tile + row * kTileStride + c4;
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
    }
    r.tile_first = new_first;
    wave_sync();
}

__device__ __forceinline__ bool reader_near_end(const StreamReader& r)
{
    return (r.bp >> 5) - r.tile_first >= (uint32_t)(kTileWords - kTileMargin);
}

// 64 stream bits starting at bit position bp (tile must cover them)
__device__ __forceinline__ uint64_t reader_window(const StreamReader& r, const uint32_t* tile, int lane, uint32_t bp)
{
    const uint32_t col = (bp >> 5) - r.tile_first;
    const uint32_t* w = tile + lane * kTileStride + (col < (uint32_t)kTileWords - 2 ? col : (uint32_t)kTileWords - 3);
    const uint32_t sh = bp & 31;
    const uint64_t lo = ((uint64_t)w[1] << 32) | w[0];
    return sh ? (lo >> sh) | ((uint64_t)w[2] << (64 - sh)) : lo;
}

__device__ __forceinline__ int32_t rice_value(uint32_t ones, uint32_t field, uint32_t k)
{
    const uint32_t rem = k ? (__brev(field) >> (32 - k)) : 0u; // remainder is MSB first in the stream
    const uint32_t u = (ones << k) | rem;                      // uint32 arithmetic as src/rice/rice_decoder.cpp:35
    return (int32_t)((u >> 1) ^ (0u - (u & 1u)));               // un-zig-zag, src/rice/rice_decoder.cpp:49-50
}

// Slow path: one codeword per lane with no length limit (src/rice/rice_decoder.cpp:27-42).
__device__ inline int32_t reader_codeword_slow(StreamReader& r, uint32_t* tile, int lane, uint32_t k, uint32_t kmask, bool live)
{
    uint32_t ones = 0;
    uint32_t lo;
    for (;;) {
        if (__any(reader_near_end(r)))
            retile(r, tile, lane);
        lo = (uint32_t)reader_window(r, tile, lane, r.bp);
        const bool in_run = live && lo == 0xFFFFFFFFu && (r.bp >> 5) <= r.n_words + 2; // zero padding ends any run
        if (!__any(in_run))
            break;
        ones += in_run ? 32u : 0u;
        r.bp += in_run ? 32u : 0u;
    }
    const uint32_t t = lo == 0xFFFFFFFFu ? 0u : (uint32_t)__builtin_ctz(~lo);
    ones += t;
    r.bp += live ? t + 1 : 0u;
    const uint32_t field = (uint32_t)reader_window(r, tile, lane, r.bp) & kmask;
    r.bp += live ? k : 0u;
    return rice_value(ones, field, k);
}

// Four codewords per lane.  live_mask bit j: value j of this group exists for this lane.
__device__ __forceinline__ void reader_codewords4(StreamReader& r, uint32_t* tile, int lane, uint32_t k, uint32_t kmask,
    uint32_t live_mask, int32_t (&out)[4])
{
    if (__any(reader_near_end(r)))
        retile(r, tile, lane);
    // 128-bit window at bp
    const uint32_t col = (r.bp >> 5) - r.tile_first;
    const uint32_t* w = tile + lane * kTileStride + col;
    const uint32_t sh = r.bp & 31;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
    uint32_t x2 = __builtin_amdgcn_alignbit(w3, w2, sh), x3 = __builtin_amdgcn_alignbit(w4, w3, sh);
    uint32_t used = 0;
    bool slow = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool live = (live_mask >> j) & 1u;
        const uint32_t t = (uint32_t)__builtin_ctz(~x0 | 0x80000000u); // <= 31
        const uint32_t len = t + 1 + k;
        slow |= live && (x0 == 0xFFFFFFFFu || len > 31);
        const uint32_t field = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (t + 1)) & kmask;
        out[j] = rice_value(t, field, k);
        const uint32_t adv = live && len <= 31 ? len : 0u;
        x0 = __builtin_amdgcn_alignbit(x1, x0, adv);
        x1 = __builtin_amdgcn_alignbit(x2, x1, adv);
        x2 = __builtin_amdgcn_alignbit(x3, x2, adv);
        x3 >>= adv;
        used += adv;
    }
    if (__any(slow)) { // some lane met a long codeword: redo the whole group bit-window by bit-window
#pragma unroll 1
        for (int j = 0; j < 4; j++)
            out[j] = reader_codeword_slow(r, tile, lane, k, kmask, (live_mask >> j) & 1u);
    } else {
        r.bp += used;
    }
}

__device__ __forceinline__ void reader_open(StreamReader& r, const uint32_t* base, uint32_t n_words, uint32_t start_bit, uint32_t* tile, int lane)
{
    r.base = base;
    r.n_words = n_words;
    r.bp = start_bit;
    retile(r, tile, lane);
}

// Values [v_begin, v_begin + v_count) of every residue stream (v_count a multiple of kStageVals); the
// call with v_begin == 0 also parses the coefficient streams and writes the descriptors.  The bit
// position each stream stopped at is kept in bit_pos[] for the next call, which lets the host pipeline
// parse(chunk j+1) against synthesize(chunk j).
__global__ __launch_bounds__(64) void k_parse_subframes(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, SubDesc* __restrict__ desc,
    int32_t* __restrict__ q_out, int32_t* __restrict__ residues, uint32_t* __restrict__ bit_pos, uint32_t* __restrict__ status,
    uint32_t v_begin, uint32_t v_count)
{
    __shared__ uint32_t tile[64 * kTileStride + 8];
    __shared__ int32_t stage[64 * kStageStride];
    const int lane = threadIdx.x;
    if (v_begin == 0 && blockIdx.x == 0 && lane < 4)
        status[lane] = 0; // ordered before every k_synthesize_frames of this call (stream / event order)
    const uint32_t n_subs = n_frames * channels;
    const uint32_t g_raw = blockIdx.x * 64 + lane; // subframe index = frame * channels + position
    const bool in_range = g_raw < n_subs;
    const uint32_t g = in_range ? g_raw : n_subs - 1; // idle lanes shadow a valid subframe, never store
    const uint32_t f = g / channels, c = g % channels;
    const uint8_t* fb = frames + frame_offsets[f];
    const uint64_t fbytes = frame_offsets[f + 1] - frame_offsets[f];

    bool ok = fbytes >= 4 && (fbytes & 3) == 0 && reinterpret_cast<const uint32_t*>(fb)[0] == SELA_SYNC_WORD;
    uint64_t p = 4;
    uint32_t channel = 0, type = 0, parent = 0, ck = 0, cw = 0, order = 0, rk = 0, rw = 0, n = 0;
    for (uint32_t i = 0; ok && i <= c; i++) { // walk the headers up to this subframe
        if (p + 12 > fbytes) {
            ok = false;
            break;
        }
        const uint32_t h0 = *reinterpret_cast<const uint32_t*>(fb + p);     // channel, type, parent, coefficient k
        const uint32_t h1 = *reinterpret_cast<const uint32_t*>(fb + p + 4); // word count (u16), order (u8), first coefficient byte
        channel = h0 & 0xFF, type = (h0 >> 8) & 0xFF, parent = (h0 >> 16) & 0xFF, ck = h0 >> 24;
        cw = h1 & 0xFFFF, order = (h1 >> 16) & 0xFF;
        const uint64_t p2 = p + 4 + 4 * (uint64_t)cw; // aligned word: last 3 coefficient bytes + residue k
        if (p2 + 8 > fbytes) {
            ok = false;
            break;
        }
        const uint32_t h2 = *reinterpret_cast<const uint32_t*>(fb + p2);
        const uint32_t h3 = *reinterpret_cast<const uint32_t*>(fb + p2 + 4);
        rk = h2 >> 24, rw = h3 & 0xFFFF, n = h3 >> 16;
        const uint64_t next = p + 12 + 4 * ((uint64_t)cw + rw);
        if (next > fbytes) {
            ok = false;
            break;
        }
        if (i < c)
            p = next;
    }
    ok = ok && channel < channels && order <= (uint32_t)kMaxOrder && n == (uint32_t)kBlock && ck < 32 && rk < 32 && type <= 1
        && (type == 0 || parent < channels);
    if (!ok) { // parse nothing: an empty stream of zeros keeps the lane in step with the wave
        p = 4, cw = 0, rw = 0, order = 0, ck = 0, rk = 0;
    }
    uint32_t flags = 0;
    const bool store = ok && in_range;

    // coefficient stream: starts 3 bytes into the aligned word at p + 4 (behind word count + order);
    // its last word shares an aligned word with the residue k, hence cw + 1 aligned words.
    if (v_begin == 0) {
        StreamReader r;
        reader_open(r, reinterpret_cast<const uint32_t*>(fb + p + 4), ok ? cw + 1 : 0, 24, tile, lane);
        const uint32_t kmask = ck ? (0xFFFFFFFFu >> (32 - ck)) : 0u;
        int32_t* qo = q_out + (size_t)g * kQStride;
        uint32_t max_order = order;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)max_order, m, 64);
            max_order = o > max_order ? o : max_order;
        }
        for (uint32_t i = 0; i < max_order; i += 4) {
            uint32_t live_mask = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                live_mask |= (i + j < order ? 1u : 0u) << j;
            int32_t v[4];
            reader_codewords4(r, tile, lane, ck, kmask, live_mask, v);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (((live_mask >> j) & 1u) && store)
                    qo[i + j] = v[j];
        }
        if (r.bp > 24 + 32 * cw)
            flags |= SELA_HIP_FLAG_RICE_OVERRUN;
    }
    // residue stream (aligned).  Values are staged in LDS, 32 per lane, and written out as full
    // 128-byte lines (8 lanes per subframe row) instead of 64 scattered stores per value.
    {
        StreamReader r;
        const uint32_t start_bit = v_begin == 0 ? 0u : bit_pos[g];
        reader_open(r, reinterpret_cast<const uint32_t*>(fb + p + 12 + 4 * (uint64_t)cw), rw, ok ? start_bit : 0u, tile, lane);
        const uint32_t kmask = rk ? (0xFFFFFFFFu >> (32 - rk)) : 0u;
        const unsigned long long store_mask = __ballot(store);
        const uint32_t live_mask = ok ? 0xFu : 0u;
#pragma unroll 1
        for (uint32_t blk = v_begin / kStageVals; blk < (v_begin + v_count) / kStageVals; blk++) {
#pragma unroll
            for (int j = 0; j < kStageVals; j += 4) {
                int32_t v[4];
                reader_codewords4(r, tile, lane, rk, kmask, live_mask, v);
                stage[lane * kStageStride + j] = v[0];
                stage[lane * kStageStride + j + 1] = v[1];
                stage[lane * kStageStride + j + 2] = v[2];
                stage[lane * kStageStride + j + 3] = v[3];
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; i++) { // 8 rows x 8 lanes x 16 bytes per wave-store
                const int row = 8 * i + (lane >> 3);
                const int c4 = (lane & 7) * 4;
                const int32_t* src = stage + row * kStageStride + c4;
                int4 v;
                v.x = src[0], v.y = src[1], v.z = src[2], v.w = src[3];
                const uint32_t g_row = blockIdx.x * 64 + (uint32_t)row;
                if ((store_mask >> row) & 1ull)
                    *reinterpret_cast<int4*>(residues + (size_t)g_row * kBlock + blk * kStageVals + c4) = v;
            }
            wave_sync();
        }
        if (v_begin + v_count == (uint32_t)kBlock && r.bp > 32 * rw)
            flags |= SELA_HIP_FLAG_RICE_OVERRUN;
        if (in_range)
            bit_pos[g] = r.bp;
    }
    if (in_range) {
        if (v_begin == 0) {
            SubDesc d;
            d.info = ok ? channel | (type << 8) | (parent << 16) | (order << 24) : 0u;
            d.flags = ok ? flags : (uint32_t)SELA_HIP_FLAG_BAD_FRAME;
            desc[g] = d;
        } else if (flags) {
            desc[g].flags |= flags;
        }
    }
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
// What is accumulated is N = 2^34 - sum(a_j s_(i-j)): the coefficients are negated once and every
// position receives the rounding constant 2^34 when it enters its last 64 samples, so the prediction
// (int32)((2^34 - P) >> 35) is the arithmetic shift (int32)N_hi >> 3 of the HIGH word alone (the
// reference's cast keeps exactly those 29 bits) -- one v_readfirstlane and two SALU ops per sample.
//
// 64x32-bit products: a' = ah*2^32 + al with al = (int32)a', so
//     z + a'*s mod 2^64 = (z + al*s)  [v_mad_i64_i32, exact]  +  ((ah*s mod 2^32) << 32)
// `rs` holds n_samples residues (a multiple of 64) of one chunk; zs is the subframe's partial-sum
// state in the workspace (position-major), carried from chunk to chunk.
//
// kFold: the residue is folded into the partial sum before the recurrence needs it,
//     N' = N - r * 2^35  (one add on the high word per 64 samples)   ==>   s = -(N' >> 35),
// which drops the per-sample v_readlane of r, and the high product is one v_mad_i32_i24.  Both need
// small operands: the shift keeps 29 bits and the multiplier 24, so this equals the reference's 32-bit
// r - (int32)((2^34 - P) >> 35) exactly while |s| < 2^23 and |a| < 2^55; the coefficients are checked
// up front and every 64 samples against 2^23, and the function returns false (state untouched) on a
// violation -- the caller then re-runs the chunk with kFold = false (v_readlane of r, v_mul_lo_u32 +
// v_add_u32).  16-bit audio never gets there; crafted streams do (tests).
template <int P, bool kFold>
__device__ inline bool synthesize(int32_t* rs, int n_samples, const int64_t* a, int order, uint64_t* zs, bool first, int lane)
{
    // The partial sums carry N = 2^34 - sum(a_j s_(i-j)) [- r_i 2^35 when folded], i.e. the NEGATED
    // coefficients are accumulated, so that the prediction is one arithmetic shift of the high word:
    // pred = N >> 35 = (int32)N_hi >> 3 -- the low word never leaves the vector unit.
    int32_t al[P];
    int32_t ah[P];
    bool mad24_ok = true;
#pragma unroll
    for (int h = 0; h < P; h++) {
        const int idx = P * lane + h + 1;
        const uint64_t nv = 0 - (uint64_t)(idx <= order ? a[idx] : 0);
        al[h] = (int32_t)(uint32_t)nv;
        ah[h] = (int32_t)(uint32_t)((nv - (uint64_t)(int64_t)al[h]) >> 32); // nv = ah 2^32 + al, al signed
        mad24_ok &= ah[h] >= -(1 << 23) && ah[h] < (1 << 23);
    }
    if (kFold && __any(!mad24_ok))
        return false; // |a| >= 2^55: the 24-bit multiply of the folded form does not apply
    uint64_t z[P];
#pragma unroll
    for (int h = 0; h < P; h++)
        z[h] = first ? 0 : zs[P * lane + h];
    bool in_range = true;
#pragma unroll 1
    for (int base = 0; base < n_samples; base += 64) {
        const int32_t r_chunk = rs[base + lane];
        // position p of this block completes at sample base + p: give it the rounding constant 2^34
        // (and, folded, its residue) now -- one add on the high word
        if (P == 1) {
            z[0] += (uint64_t)(4u - (kFold ? (uint32_t)r_chunk << 3 : 0u)) << 32;
        } else if (lane < 32) {
#pragma unroll
            for (int h = 0; h < P; h++)
                z[h] += (uint64_t)(4u - (kFold ? (uint32_t)rs[base + P * lane + h] << 3 : 0u)) << 32;
        }
        int32_t s_chunk = 0;
#pragma unroll
        for (int m = 0; m < 64; m++) {
            // scalar side: N_i sits in lane 0, position 0
            const int32_t n_hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(z[0] >> 32));
            const int32_t pred = n_hi >> 3;
            int32_t s_i;
            if (kFold)
                s_i = (int32_t)(0u - (uint32_t)pred);
            else
                s_i = (int32_t)((uint32_t)__builtin_amdgcn_readlane(r_chunk, m) - (uint32_t)pred);
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(s_chunk) : "s"(s_i), "n"(m));
            // vector side: move every partial sum down one position and add this sample's products
            const uint64_t in = wave_shl1_zero(z[0]);
#pragma unroll
            for (int h = 0; h < P; h++) {
                const uint64_t up = h + 1 < P ? z[h + 1 < P ? h + 1 : h] : in;
                const uint64_t lo = (uint64_t)((int64_t)up + (int64_t)al[h] * (int64_t)s_i);
                uint32_t hi;
                if (kFold) // high part of the product: both factors fit 24 bits in the folded form (checked)
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(hi) : "v"(ah[h]), "s"(s_i), "v"((uint32_t)(lo >> 32)));
                else
                    hi = (uint32_t)(lo >> 32) + (uint32_t)ah[h] * (uint32_t)s_i;
                z[h] = ((uint64_t)hi << 32) | (uint32_t)lo;
            }
        }
        if (kFold)
            in_range &= (uint32_t)(s_chunk + (1 << 23)) < (1u << 24);
        rs[base + lane] = s_chunk;
    }
    if (kFold && __any(!in_range))
        return false;
#pragma unroll
    for (int h = 0; h < P; h++)
        zs[P * lane + h] = z[h];
    wave_sync();
    return true;
}

// per-subframe state carried between the chunks of one decode call (workspace)
struct SynthState {
    int64_t a[104];   // Q35 predictor
    uint64_t z[128];  // partial sums by tap position
};

struct SynthWaveLds {
    double k[104];
    int64_t a[104];
};

// kProf: also write per-phase cycle counts (debug hook sela_hip_debug_phase_buffer; 16 uint64 per subframe).
template <bool kProf>
__global__ __launch_bounds__(kDecMaxWaves * 64) void k_synthesize_frames(const SubDesc* __restrict__ desc,
    const int32_t* __restrict__ q_in, const int32_t* __restrict__ residues, SynthState* __restrict__ state, uint32_t n_frames,
    uint32_t channels, uint32_t v_begin, uint32_t v_count, int16_t* __restrict__ pcm_out, uint32_t* __restrict__ status,
    uint64_t* __restrict__ phase_cycles)
{
    long long stamp[6];
    for (int i = 0; i < 6; i++)
        stamp[i] = 0;
    uint32_t prof_sub = 0xFFFFFFFFu;
    if (kProf)
        stamp[0] = clock64();
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    // [channels][v_count] int32 samples, then one SynthWaveLds per wave, then per-channel type/parent
    int32_t* const samples = reinterpret_cast<int32_t*>(dyn);
    const int n_waves = blockDim.x / 64;
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    SynthWaveLds* const wl = reinterpret_cast<SynthWaveLds*>(dyn + (size_t)channels * v_count * 4) + wave;
    uint32_t* const sub_info = reinterpret_cast<uint32_t*>(dyn + (size_t)channels * v_count * 4 + (size_t)n_waves * sizeof(SynthWaveLds));
    const bool first = v_begin == 0;

    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    uint32_t flags = 0;
    for (uint32_t c = threadIdx.x; c < channels; c += blockDim.x)
        sub_info[c] = 0xFFFFFFFFu; // "no subframe delivered this channel"
    __syncthreads();

    for (uint32_t c = wave; c < channels; c += n_waves) {
        const uint32_t g = f * channels + c;
        const SubDesc d = desc[g];
        flags |= d.flags;
        if (d.flags & SELA_HIP_FLAG_BAD_FRAME)
            continue;
        const uint32_t channel = d.info & 0xFF, type = (d.info >> 8) & 0xFF, parent = (d.info >> 16) & 0xFF, order = d.info >> 24;
        int32_t* dst = samples + (size_t)channel * v_count;
        // this chunk's residues -> LDS (coalesced 16-byte loads)
        const int4* rsrc = reinterpret_cast<const int4*>(residues + (size_t)g * kBlock + v_begin);
        int4* rdst = reinterpret_cast<int4*>(dst);
        for (uint32_t t4 = lane; t4 < v_count / 4; t4 += 64)
            rdst[t4] = rsrc[t4];
        SynthState* st = state + g;
        if (first) {
            // dequantise (src/lpc/linear_predictor.cpp:16-28) + step-up, kept for the later chunks
            for (uint32_t i = lane; i < order; i += 64)
                wl->k[i] = order <= 1 ? 0.0 : dequant((int)i, q_in[(size_t)g * kQStride + i], flags);
            wave_sync();
            if (kProf)
                stamp[1] = clock64(), prof_sub = c;
            step_up(wl->k, wl->a, (int)order, lane, flags);
            for (uint32_t i = lane; i <= order; i += 64)
                st->a[i] = wl->a[i];
        } else {
            for (uint32_t i = lane; i <= order; i += 64)
                wl->a[i] = st->a[i];
            wave_sync();
            if (kProf)
                stamp[1] = clock64(), prof_sub = c;
        }
        if (kProf)
            stamp[2] = clock64();
        const bool exact_needed = order <= 64 ? !synthesize<1, true>(dst, (int)v_count, wl->a, (int)order, st->z, first, lane)
                                              : !synthesize<2, true>(dst, (int)v_count, wl->a, (int)order, st->z, first, lane);
        if (exact_needed) { // a sample or coefficient left the range of the folded form: redo this chunk the long way
            wave_sync();
            for (uint32_t t4 = lane; t4 < v_count / 4; t4 += 64)
                rdst[t4] = rsrc[t4];
            wave_sync();
            if (order <= 64)
                synthesize<1, false>(dst, (int)v_count, wl->a, (int)order, st->z, first, lane);
            else
                synthesize<2, false>(dst, (int)v_count, wl->a, (int)order, st->z, first, lane);
        }
        if (kProf)
            stamp[3] = clock64();
        if (lane == 0)
            sub_info[channel] = type | (parent << 8);
    }
    __syncthreads();
    if (kProf)
        stamp[4] = clock64();

    // ---- second pass of frame::FrameDecoder + interleave to int16 ------------------------------------
    // dependent channels become parent - difference (parents are independent subframes); a channel
    // that no valid subframe delivered decodes to silence and raises BAD_FRAME.
    for (uint32_t i = threadIdx.x; i < v_count; i += blockDim.x) {
        for (uint32_t c = 0; c < channels; c++) {
            const uint32_t info = sub_info[c];
            int32_t v = info == 0xFFFFFFFFu ? 0 : samples[(size_t)c * v_count + i];
            if (info != 0xFFFFFFFFu && (info & 0xFF) == 1) {
                const uint32_t par = info >> 8;
                const int32_t pv = sub_info[par] == 0xFFFFFFFFu ? 0 : samples[(size_t)par * v_count + i];
                v = (int32_t)((uint32_t)pv - (uint32_t)v);
            }
            pcm_out[((size_t)f * kBlock + v_begin + i) * channels + c] = (int16_t)(uint16_t)v;
        }
    }
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < channels; c++) {
            const uint32_t info = sub_info[c];
            if (info == 0xFFFFFFFFu)
                flags |= SELA_HIP_FLAG_BAD_FRAME;
            else if ((info & 0xFF) == 1 && (sub_info[info >> 8] == 0xFFFFFFFFu || (sub_info[info >> 8] & 0xFF) != 0))
                flags |= SELA_HIP_FLAG_BAD_FRAME; // a parent that is itself dependent is outside what the reference defines
        }
    }
    for (int m = 32; m >= 1; m >>= 1)
        flags |= (uint32_t)__shfl_xor((int)flags, m, 64);
    if (lane == 0 && flags) {
        atomicOr(&status[0], flags);
        if (wave == 0 && first && (flags & SELA_HIP_FLAG_BAD_FRAME))
            atomicAdd(&status[1], 1u);
    }
    if (kProf && lane == 0 && prof_sub != 0xFFFFFFFFu) { // (one subframe per wave is reported)
        stamp[5] = clock64();
        for (int i = 0; i < 5; i++)
            phase_cycles[((size_t)f * channels + prof_sub) * 16 + i] = (uint64_t)(stamp[i + 1] - stamp[i]);
    }
}

size_t decode_lds_bytes(uint32_t channels, int n_waves, uint32_t v_count)
{
    return (size_t)channels * v_count * 4 + (size_t)n_waves * sizeof(SynthWaveLds) + (size_t)channels * 4 + 16;
}

int decode_waves(uint32_t channels)
{
    return channels < (uint32_t)kDecMaxWaves ? (int)channels : kDecMaxWaves;
}

static size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t decode_workspace_bytes(uint32_t n_frames, uint32_t channels)
{
    const size_t subs = (size_t)n_frames * channels;
    return round256(subs * sizeof(SubDesc)) + round256(subs * kQStride * 4) + round256(subs * kBlock * 4) + round256(subs * 4)
        + round256(subs * sizeof(SynthState)) + 256;
}

// Decode = parse + synthesise, pipelined along the sample axis: the parse of values chunk j+1 (a
// latency-bound kernel that occupies ~1 wave per 64 subframes) runs on a side stream while chunk j is
// synthesised on the caller's stream.  `side`/`parsed` are owned by the caller (sela_capi.hip); with
// side == nullptr the call degenerates to one chunk on the caller's stream.
hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, hipStream_t stream, hipEvent_t* ev /* 3 events or nullptr */,
    uint64_t* d_phase_cycles, hipStream_t side, hipEvent_t fork, hipEvent_t* parsed /* kDecodeChunks events */)
{
    hipError_t err = hipSuccess;
    if (n_frames == 0)
        return hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
    const size_t subs = (size_t)n_frames * channels;
    unsigned char* ws = reinterpret_cast<unsigned char*>(((uintptr_t)d_workspace + 255) & ~(uintptr_t)255);
    SubDesc* desc = reinterpret_cast<SubDesc*>(ws);
    ws += round256(subs * sizeof(SubDesc));
    int32_t* q = reinterpret_cast<int32_t*>(ws);
    ws += round256(subs * kQStride * 4);
    int32_t* residues = reinterpret_cast<int32_t*>(ws);
    ws += round256(subs * kBlock * 4);
    uint32_t* bit_pos = reinterpret_cast<uint32_t*>(ws);
    ws += round256(subs * 4);
    SynthState* state = reinterpret_cast<SynthState*>(ws);

    const bool pipelined = side != nullptr && ev == nullptr && d_phase_cycles == nullptr;
    const uint32_t chunks = pipelined ? (uint32_t)kDecodeChunks : 1u;
    const uint32_t v_count = (uint32_t)kBlock / chunks;
    const int n_waves = decode_waves(channels);
    const size_t lds = decode_lds_bytes(channels, n_waves, v_count);
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    if (lds > 64 * 1024) { // above the default dynamic-LDS limit (many channels, unpipelined call)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_synthesize_frames<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_synthesize_frames<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    const dim3 parse_grid((unsigned)((subs + 63) / 64));
    hipStream_t parse_stream = stream;
    if (pipelined) {
        if ((err = hipEventRecord(fork, stream)) != hipSuccess || (err = hipStreamWaitEvent(side, fork, 0)) != hipSuccess)
            return err;
        parse_stream = side;
    }
    for (uint32_t j = 0; j < chunks; j++) {
        const uint32_t v_begin = j * v_count;
        if (ev)
            (void)hipEventRecord(ev[0], stream);
        hipLaunchKernelGGL(k_parse_subframes, parse_grid, dim3(64), 0, parse_stream, d_frames, d_frame_offsets, n_frames, channels, desc, q,
            residues, bit_pos, d_status, v_begin, v_count);
        if (pipelined) {
            if ((err = hipEventRecord(parsed[j], side)) != hipSuccess || (err = hipStreamWaitEvent(stream, parsed[j], 0)) != hipSuccess)
                return err;
        }
        if (ev)
            (void)hipEventRecord(ev[1], stream);
        if (d_phase_cycles)
            hipLaunchKernelGGL(k_synthesize_frames<true>, dim3(n_frames), dim3(n_waves * 64), lds, stream, desc, q, residues, state, n_frames,
                channels, v_begin, v_count, d_pcm_out, d_status, d_phase_cycles);
        else
            hipLaunchKernelGGL(k_synthesize_frames<false>, dim3(n_frames), dim3(n_waves * 64), lds, stream, desc, q, residues, state, n_frames,
                channels, v_begin, v_count, d_pcm_out, d_status, d_phase_cycles);
        if (ev)
            (void)hipEventRecord(ev[2], stream);
    }
    return hipGetLastError();
}

} // namespace sela
