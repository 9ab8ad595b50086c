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
constexpr int kDecodeChunks = 4; // sample-axis pipeline depth of one decode call
// Values per chunk (multiples of 128).  Only the first parse is exposed -- and it also walks the headers
// and the coefficient streams -- so the first chunk is the short one.
constexpr uint32_t kChunkValues[kDecodeChunks] = { 256, 512, 640, 640 };
constexpr uint32_t kChunkMax = 640;
static_assert(kChunkValues[0] + kChunkValues[1] + kChunkValues[2] + kChunkValues[3] == (uint32_t)kBlock, "chunks cover the block");

// per-subframe record written by k_parse_subframes
struct SubDesc {
    uint32_t info;  // channel | type << 8 | parent << 16 | order << 24
    uint32_t flags; // SELA_HIP_FLAG_* bits; BAD_FRAME means "do not synthesise"
    uint32_t res_k; // Rice parameter of the residue stream (to unpack kPacked residue words)
    uint32_t pad;
};

// ---- lane-per-stream Rice parser --------------------------------------------------------------------
// Each of the 64 lanes parses its own bitstream (stream bit t = bit t%32 of word t/32) and only keeps
// a bit position.  The words are staged through an LDS tile, one row of kTileWords words per lane,
// that the whole wave refills cooperatively with coalesced 16-byte loads whenever ANY lane gets close
// to the end of its row.  Control flow is wave-uniform throughout; only data is per lane.
//
// Codewords are decoded four at a time from a 128-bit register window read at the lane's bit
// position (one LDS round trip per four values).  A group falls back to the bit-by-window slow path
// when any lane meets a codeword longer than 30 bits (long unary run).
//
// The tile holds the stream INVERTED (zero padding beyond a stream's end becomes ones): the unary run
// length is then one v_ffbl_b32, with no v_not_b32 on the per-value path.
//
// This kernel is the serial part of decoding -- 64 streams per wave, one wave per SIMD, every
// instruction of the per-value path costs its full issue latency -- so the residue path does the
// bare minimum per codeword: find its length, cut out the remainder bits, move the window.  It emits
// PACKED words, (ones + 1) << 24 | inverted remainder bits, and leaves bit reversal, un-zig-zag and
// friends to unpack_residue(), which runs in k_synthesize_frames across all lanes of 64x more waves.
// Blocks of 32 values in which some group took the slow path are stored as final values instead;
// one bit per block in res_raw[] says which (src/rice/rice_decoder.cpp:27-51 either way).
constexpr int kTileWords = 256;             // one wave-wide 16-byte load fills one row
constexpr int kTileStride = kTileWords + 1; // odd stride: lanes reading the same column hit different banks
constexpr int kTileMargin = 12;             // re-tile when a lane is within this many words of its row end
constexpr int kStageVals = 32;              // decoded values staged per lane before a coalesced store
constexpr int kStageStride = kStageVals + 1;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct StreamReader {
    const uint8_t* wg_frames; // this workgroup's part of the frame bytes and its size (wave-uniform)
    uint32_t wg_bytes;
    uint32_t base;        // byte offset of the stream's first aligned word in it
    uint32_t n_words;     // words in the stream; reads beyond are zero
    uint32_t tile_first;  // stream index of tile column 0
    uint32_t bp;          // bit position of the next unread bit
    long long t_retile;   // cycles spent refilling the tile, and how often (phase profile)
    uint32_t n_retile;
};

// Refill every lane's row so that it starts at the word holding the lane's next unread bit.  One
// bounds-checked buffer_load_dwordx4 per row (64 lanes x 16 bytes = the row): no branches, so all 64
// loads are in flight together; the row's start and length travel through SGPRs (v_readlane).
__device__ inline void retile(StreamReader& r, uint32_t* tile, int lane)
{
    const long long t_in = clock64();
    const uint32_t new_first = r.bp >> 5;
    const uint32_t row_start = r.base + 4 * new_first;                          // byte offset of column 0
    const uint32_t row_words = new_first < r.n_words ? r.n_words - new_first : 0; // stream words from there on
    const uint32_t c4 = 4 * (uint32_t)lane;
    // bounds-checked raw-dword resource; rebuilt from scalars here because a resource that travelled
    // through this function's arguments is no longer known to be wave-uniform
    const __amdgpu_buffer_rsrc_t frames = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(read_first_lane(reinterpret_cast<uint64_t>(r.wg_frames))), 0,
        (uint32_t)__builtin_amdgcn_readfirstlane((int)r.wg_bytes), 0x00020000);
    wave_sync(); // earlier reads of the tile are done
    constexpr int kBatch = 16; // rows in flight together
#pragma unroll 1
    for (int row0 = 0; row0 < 64; row0 += kBatch) {
        u32x4 v[kBatch];
        uint32_t words[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; i++) { // all loads first ...
            const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)row_start, row0 + i);
            words[i] = (uint32_t)__builtin_amdgcn_readlane((int)row_words, row0 + i);
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(frames, 16 * (uint32_t)lane, start, 0); // 0 past the buffer end
        }
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would otherwise pair each load with its use and serialise them)
#pragma unroll
        for (int i = 0; i < kBatch; i++) { // ... then into the tile: stream words inverted, zero padding behind the stream
            uint32_t* dst = tile + (row0 + i) * kTileStride + c4;
            dst[0] = c4 + 0 < words[i] ? ~v[i].x : 0xFFFFFFFFu;
            dst[1] = c4 + 1 < words[i] ? ~v[i].y : 0xFFFFFFFFu;
            dst[2] = c4 + 2 < words[i] ? ~v[i].z : 0xFFFFFFFFu;
            dst[3] = c4 + 3 < words[i] ? ~v[i].w : 0xFFFFFFFFu;
        }
    }
    r.tile_first = new_first;
    wave_sync();
    r.t_retile += clock64() - t_in;
    r.n_retile++;
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
    return ~(sh ? (lo >> sh) | ((uint64_t)w[2] << (64 - sh)) : lo); // the tile is inverted
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
    uint32_t x0 = ~__builtin_amdgcn_alignbit(w1, w0, sh), x1 = ~__builtin_amdgcn_alignbit(w2, w1, sh); // the tile is inverted
    uint32_t x2 = ~__builtin_amdgcn_alignbit(w3, w2, sh), x3 = ~__builtin_amdgcn_alignbit(w4, w3, sh);
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

// ---- residue fast path: four PACKED codewords per lane, every lane live --------------------------------
// packed word = (ones + 1) << 24 | the k remainder bits as they sit in the INVERTED stream (LSB first).
constexpr uint32_t kPackShift = 24;
constexpr uint32_t kPackMaxK = 24;   // remainder bits must fit below the run length
constexpr uint32_t kPackMaxLen = 30; // longest codeword the register window handles (ones + 1 + k)

__device__ __forceinline__ int32_t unpack_residue(uint32_t p, uint32_t k, uint32_t kmask)
{
    return rice_value((p >> kPackShift) - 1u, ~p & kmask, k);
}

// Returns true if the group was decoded (out = packed words, position advanced); false if some lane met
// a codeword the window cannot hold -- nothing is consumed then and the caller redoes the group slowly.
__device__ __forceinline__ bool reader_packed4(StreamReader& r, uint32_t* tile, int lane, uint32_t k, uint32_t kmask, bool k_fits,
    uint32_t (&out)[4])
{
    if (__any(reader_near_end(r)))
        retile(r, tile, lane);
    const uint32_t col = (r.bp >> 5) - r.tile_first;
    const uint32_t* w = tile + lane * kTileStride + col;
    const uint32_t sh = r.bp & 31;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    uint32_t n0 = __builtin_amdgcn_alignbit(w1, w0, sh), n1 = __builtin_amdgcn_alignbit(w2, w1, sh); // inverted bits
    uint32_t n2 = __builtin_amdgcn_alignbit(w3, w2, sh), n3 = __builtin_amdgcn_alignbit(w4, w3, sh);
    uint32_t used = 0;
    bool fast = k_fits;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // ones + 1 = index of the first set bit of (inverted window << 1); the sentinel caps it at 31
        const uint32_t t1 = (uint32_t)__builtin_ctz((n0 << 1) | 0x80000000u);
        const uint32_t len = t1 + k;
        fast &= len <= kPackMaxLen;
        const uint32_t field = __builtin_amdgcn_alignbit(n1, n0, t1) & kmask;
        out[j] = (t1 << kPackShift) | field;
        if (j < 3) { // (shift amounts are taken mod 32: a too-long codeword garbles a group that is redone anyway)
            n0 = __builtin_amdgcn_alignbit(n1, n0, len);
            n1 = __builtin_amdgcn_alignbit(n2, n1, len);
            if (j < 2)
                n2 = __builtin_amdgcn_alignbit(n3, n2, len);
            if (j < 1)
                n3 >>= len & 31;
        }
        used += len;
    }
    if (__any(!fast))
        return false;
    r.bp += used;
    return true;
}

__device__ __forceinline__ void reader_open(StreamReader& r, const uint8_t* wg_frames, uint32_t wg_bytes, uint32_t base, uint32_t n_words,
    uint32_t start_bit, uint32_t* tile, int lane)
{
    r.wg_frames = wg_frames;
    r.wg_bytes = wg_bytes;
    r.base = base;
    r.n_words = n_words;
    r.bp = start_bit;
    r.t_retile = 0;
    r.n_retile = 0;
    retile(r, tile, lane);
}

// Values [v_begin, v_begin + v_count) of every residue stream (v_count a multiple of kStageVals); the
// call with v_begin == 0 also parses the coefficient streams and writes the descriptors.  The bit
// position each stream stopped at is kept in bit_pos[] for the next call, which lets the host pipeline
// parse(chunk j+1) against synthesize(chunk j).
__global__ __launch_bounds__(64) void k_parse_subframes(const uint8_t* __restrict__ frames,
    const uint64_t* __restrict__ frame_offsets, uint32_t n_frames, uint32_t channels, SubDesc* __restrict__ desc,
    int32_t* __restrict__ q_out, int32_t* __restrict__ residues, uint32_t* __restrict__ bit_pos, uint64_t* __restrict__ res_raw,
    uint32_t* __restrict__ status, uint32_t v_begin, uint32_t v_count, uint64_t* __restrict__ phase_cycles)
{
    const long long t_start = clock64();
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
    // The streams are read through a bounds-checked buffer resource that starts at this workgroup's
    // first frame (64 subframes span far less than 4 GB) and ends with the frame bytes.
    const uint64_t wg_base = frame_offsets[(blockIdx.x * 64) / channels];
    const uint64_t wg_bytes = frame_offsets[n_frames] - wg_base;
    const uint32_t wg_size = wg_bytes < 0xFFFFFFFFull ? (uint32_t)wg_bytes : 0xFFFFFFFFu;

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

    // coefficient stream: starts 3 bytes into the aligned word at p + 4 (behind word count + order); cw
    // words later come the 5 bytes of the residue header and then, aligned again, the residue words at
    // aligned word cw + 2.  The first chunk opens ONE tile row over both streams (one refill less).
    StreamReader r;
    uint32_t res_origin = 0; // bit position of the residue stream within the row the reader was opened on
    if (v_begin == 0) {
        reader_open(r, frames + wg_base, wg_size, (uint32_t)(frame_offsets[f] - wg_base + p + 4), ok ? cw + 2 + rw : 0, 24, tile, lane);
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
        res_origin = ok ? 32 * (cw + 2) : 0u;
        r.bp = res_origin;
    } else {
        reader_open(r, frames + wg_base, wg_size, (uint32_t)(frame_offsets[f] - wg_base + p + 12 + 4 * (uint64_t)cw), rw, ok ? bit_pos[g] : 0u,
            tile, lane);
    }
    // residue stream (aligned).  Words are staged in LDS, 32 per lane, and written out as full
    // 128-byte lines (8 lanes per subframe row) instead of 64 scattered stores per value.
    {
        const uint32_t kmask = rk ? (0xFFFFFFFFu >> (32 - rk)) : 0u;
        const bool k_fits = rk <= kPackMaxK;
        const unsigned long long store_mask = __ballot(store);
        uint64_t raw_blocks = 0; // wave-uniform: blocks of 32 values stored as final values
        const long long t_residues = clock64();
        long long t_store = 0;
#pragma unroll 1
        for (uint32_t blk = v_begin / kStageVals; blk < (v_begin + v_count) / kStageVals; blk++) {
            uint32_t slow_groups = 0; // wave-uniform
#pragma unroll
            for (int j = 0; j < kStageVals; j += 4) {
                uint32_t pk[4];
                if (!reader_packed4(r, tile, lane, rk, kmask, k_fits, pk)) {
                    slow_groups |= 1u << (j / 4);
#pragma unroll 1
                    for (int jj = 0; jj < 4; jj++) // lanes of rejected subframes walk an empty stream
                        pk[jj] = (uint32_t)reader_codeword_slow(r, tile, lane, rk, kmask, true);
                }
                stage[lane * kStageStride + j] = (int32_t)pk[0];
                stage[lane * kStageStride + j + 1] = (int32_t)pk[1];
                stage[lane * kStageStride + j + 2] = (int32_t)pk[2];
                stage[lane * kStageStride + j + 3] = (int32_t)pk[3];
            }
            if (slow_groups)
                raw_blocks |= 1ull << blk;
            const long long t_s0 = clock64();
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; i++) { // 8 rows x 8 lanes x 16 bytes per wave-store; a lane's 4 words are one group
                const int row = 8 * i + (lane >> 3);
                const int c4 = (lane & 7) * 4;
                const int32_t* src = stage + row * kStageStride + c4;
                int4 v;
                v.x = src[0], v.y = src[1], v.z = src[2], v.w = src[3];
                if (slow_groups) { // raw block (wave-uniform, rare): finish its packed groups here
                    const uint32_t row_k = (uint32_t)__shfl((int)rk, row, 64), row_mask = (uint32_t)__shfl((int)kmask, row, 64);
                    if (!((slow_groups >> (lane & 7)) & 1u)) {
                        v.x = unpack_residue((uint32_t)v.x, row_k, row_mask);
                        v.y = unpack_residue((uint32_t)v.y, row_k, row_mask);
                        v.z = unpack_residue((uint32_t)v.z, row_k, row_mask);
                        v.w = unpack_residue((uint32_t)v.w, row_k, row_mask);
                    }
                }
                const uint32_t g_row = blockIdx.x * 64 + (uint32_t)row;
                if (store_mask == ~0ull || ((store_mask >> row) & 1ull)) // (the uniform test keeps the 8 stores branch-free)
                    *reinterpret_cast<int4*>(residues + (size_t)g_row * kBlock + blk * kStageVals + c4) = v;
            }
            wave_sync();
            t_store += clock64() - t_s0;
        }
        if (phase_cycles && lane == 0) { // slots 8.. of this wave's first subframe (tools/phase_profile.py)
            uint64_t* pc = phase_cycles + (size_t)blockIdx.x * 64 * 16 + 8;
            const long long t_end = clock64();
            pc[0] = (uint64_t)(t_residues - t_start);                       // headers + coefficient streams + open
            pc[1] = (uint64_t)(t_end - t_residues - t_store - r.t_retile);  // codeword groups
            pc[2] = (uint64_t)r.t_retile;                                   // tile refills ...
            pc[3] = r.n_retile;                                             // ... and their number
            pc[4] = (uint64_t)t_store;                                      // staged stores
        }
        if (v_begin + v_count == (uint32_t)kBlock && r.bp > res_origin + 32 * rw)
            flags |= SELA_HIP_FLAG_RICE_OVERRUN;
        if (in_range) {
            bit_pos[g] = r.bp - res_origin;
            res_raw[g] = v_begin == 0 ? raw_blocks : res_raw[g] | raw_blocks;
        }
    }
    if (in_range) {
        if (v_begin == 0) {
            SubDesc d;
            d.info = ok ? channel | (type << 8) | (parent << 16) | (order << 24) : 0u;
            d.flags = ok ? flags : (uint32_t)SELA_HIP_FLAG_BAD_FRAME;
            d.res_k = rk;
            d.pad = 0;
            desc[g] = d;
        } else if (flags) {
            desc[g].flags |= flags;
        }
    }
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
// `rs` holds n_samples residues (a multiple of the ring) of one chunk; zs is the subframe's partial-sum
// state in the workspace (register-major), carried from chunk to chunk.
//
// kFold: the residue is folded into its sum at the start of its block of 64,
//     N' = N - r * 2^35  (one subtract on the high word per 64 samples)   ==>   s = -(N' >> 35),
// which drops the per-sample v_readlane of r, and the high product is one v_mad_i32_i24.  Both need
// small operands: the shift keeps 29 bits and the multiplier 24, so this equals the reference's 32-bit
// r - (int32)((2^34 - P) >> 35) exactly while |s| < 2^23 and |a| < 2^55; the coefficients are checked
// when the table is built and every 64 samples against 2^23, and the function returns false (state
// untouched) on a violation -- the caller then re-runs the chunk with kFold = false (v_readlane of r,
// v_mul_lo_u32 + v_add_u32).  16-bit audio never gets there; crafted streams do (tests).
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
// barrier.
constexpr int kAhead = 4;
// volatile: keeps the two reads of a ring-of-128 step as ds_read_b64 (2 LDS cycles each); merged into
// one ds_read2_b64 they would take 8 (MI355X_MICROARCH.md, LDS table) and the kernel turns LDS-bound
typedef const volatile __attribute__((address_space(3))) uint64_t* LdsTable;

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

// One block of 64 samples at rs[0..63]; returns false if a sample left the range of the folded form.
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
    rs[lane] = s;
    return !kFold || (uint32_t)(s + (1 << 23)) < (1u << 24);
}

// R = ring / 64 (1: order <= 64 - G, 2: order <= 128 - G); G = recycling group (4 or 16).
template <int R, bool kFold, int G>
__device__ inline bool synthesize(int32_t* rs, int n_samples, const uint64_t* tab, uint64_t* zs, bool first, int lane)
{
    static_assert(G == 4 || G == 16, "groups are DPP banks or rows");
    uint32_t zl[2], zh[2];
#pragma unroll
    for (int h = 0; h < R; h++) {
        const uint64_t z = first ? (uint64_t)4 << 32 : zs[64 * h + lane]; // every sum starts at 2^34
        zl[h] = (uint32_t)z;
        zh[h] = (uint32_t)(z >> 32);
    }
    uint32_t four = 4, zero = 0;
    asm volatile("" : "+v"(four), "+v"(zero)); // DPP sources must be VGPRs
    const LdsTable tab_lane = (LdsTable)(tab + lane); // the table is in LDS: ds_read with immediate offsets
    bool in_range = true;
#pragma unroll 1
    for (int base = 0; base < n_samples; base += 64 * R) {
        in_range &= synth_block<R, kFold, G>(rs + base, zl[0], zh[0], zl[R - 1], zh[R - 1], tab_lane, lane, four, zero);
        if (R == 2)
            in_range &= synth_block<R, kFold, G>(rs + base + 64, zl[1], zh[1], zl[0], zh[0], tab_lane, lane, four, zero);
    }
    if (kFold && __any(!in_range))
        return false;
#pragma unroll
    for (int h = 0; h < R; h++)
        zs[64 * h + lane] = ((uint64_t)zh[h] << 32) | zl[h];
    wave_sync();
    return true;
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

// per-subframe state carried between the chunks of one decode call (workspace)
struct SynthState {
    int64_t a[104];   // Q35 predictor
    uint64_t z[128];  // partial sums: register h of lane l at [64 h + l]
};

struct SynthWaveLds {
    union {
        struct {
            double k[104];
            int64_t a[104];
        };
        uint64_t tab[256]; // synthesis coefficient table (build_synth_table), replaces k[] and a[]
    };
};

// kProf: also write per-phase cycle counts (debug hook sela_hip_debug_phase_buffer; 16 uint64 per subframe).
template <bool kProf>
__global__ __launch_bounds__(kDecMaxWaves * 64) void k_synthesize_frames(const SubDesc* __restrict__ desc,
    const int32_t* __restrict__ q_in, const int32_t* __restrict__ residues, const uint64_t* __restrict__ res_raw,
    SynthState* __restrict__ state, uint32_t n_frames, uint32_t channels, uint32_t v_begin, uint32_t v_count,
    int16_t* __restrict__ pcm_out, uint32_t* __restrict__ status, uint64_t* __restrict__ phase_cycles)
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
        // this chunk's residues -> LDS (coalesced 16-byte loads), finishing the parser's packed words
        // (every block of 32 values whose bit in res_raw is clear) on the way
        const int4* rsrc = reinterpret_cast<const int4*>(residues + (size_t)g * kBlock + v_begin);
        int4* rdst = reinterpret_cast<int4*>(dst);
        const uint64_t raw_blocks = res_raw[g];
        const uint32_t res_k = d.res_k, res_kmask = res_k ? (0xFFFFFFFFu >> (32 - res_k)) : 0u;
        auto load_residues = [&]() {
            for (uint32_t t4 = lane; t4 < v_count / 4; t4 += 64) {
                int4 v = rsrc[t4];
                if (!((raw_blocks >> ((v_begin + 4 * t4) / kStageVals)) & 1ull)) {
                    v.x = unpack_residue((uint32_t)v.x, res_k, res_kmask);
                    v.y = unpack_residue((uint32_t)v.y, res_k, res_kmask);
                    v.z = unpack_residue((uint32_t)v.z, res_k, res_kmask);
                    v.w = unpack_residue((uint32_t)v.w, res_k, res_kmask);
                }
                rdst[t4] = v;
            }
        };
        load_residues();
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
        const bool fits24 = build_synth_table(wl->a, wl->tab, (int)order, lane);
        // ring / recycling group by order: <= 48: 64 / 16, <= 60: 64 / 4, else 128 / 16
        bool done = false;
        if (fits24)
            done = order <= 48 ? synthesize<1, true, 16>(dst, (int)v_count, wl->tab, st->z, first, lane)
                 : order <= 60 ? synthesize<1, true, 4>(dst, (int)v_count, wl->tab, st->z, first, lane)
                               : synthesize<2, true, 16>(dst, (int)v_count, wl->tab, st->z, first, lane);
        if (!done) { // a sample or coefficient left the range of the folded form: redo this chunk the long way
            wave_sync();
            load_residues();
            wave_sync();
            if (order <= 48)
                synthesize<1, false, 16>(dst, (int)v_count, wl->tab, st->z, first, lane);
            else if (order <= 60)
                synthesize<1, false, 4>(dst, (int)v_count, wl->tab, st->z, first, lane);
            else
                synthesize<2, false, 16>(dst, (int)v_count, wl->tab, st->z, first, lane);
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
    if (channels == 2) {
        // stereo: four samples of both channels per thread, one 16-byte store (wave-uniform case analysis)
        const uint32_t i0 = sub_info[0], i1 = sub_info[1];
        const bool have0 = i0 != 0xFFFFFFFFu, have1 = i1 != 0xFFFFFFFFu;
        const bool dep0 = have0 && (i0 & 0xFF) == 1, dep1 = have1 && (i1 & 0xFF) == 1;
        const uint32_t par0 = i0 >> 8, par1 = i1 >> 8; // parent channel of a dependent subframe (0 or 1, checked by the parser)
        const int4* s0 = reinterpret_cast<const int4*>(samples);
        const int4* s1 = reinterpret_cast<const int4*>(samples + v_count);
        uint4* out = reinterpret_cast<uint4*>(pcm_out + ((size_t)f * kBlock + v_begin) * 2);
        for (uint32_t i4 = threadIdx.x; i4 < v_count / 4; i4 += blockDim.x) {
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
    } else
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
    flags = wave_or(flags);
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
        + round256(subs * 8) + round256(subs * sizeof(SynthState)) + 256;
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
    uint64_t* res_raw = reinterpret_cast<uint64_t*>(ws);
    ws += round256(subs * 8);
    SynthState* state = reinterpret_cast<SynthState*>(ws);

    const bool pipelined = side != nullptr && ev == nullptr && d_phase_cycles == nullptr;
    const uint32_t chunks = pipelined ? (uint32_t)kDecodeChunks : 1u;
    const int n_waves = decode_waves(channels);
    size_t lds = decode_lds_bytes(channels, n_waves, pipelined ? kChunkMax : (uint32_t)kBlock);
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
    uint32_t v_begin = 0;
    for (uint32_t j = 0; j < chunks; j++, v_begin += (pipelined ? kChunkValues[j - 1] : (uint32_t)kBlock)) {
        const uint32_t v_count = pipelined ? kChunkValues[j] : (uint32_t)kBlock;
        lds = decode_lds_bytes(channels, n_waves, v_count);
        if (ev)
            (void)hipEventRecord(ev[0], stream);
        hipLaunchKernelGGL(k_parse_subframes, parse_grid, dim3(64), 0, parse_stream, d_frames, d_frame_offsets, n_frames, channels, desc, q,
            residues, bit_pos, res_raw, d_status, v_begin, v_count, d_phase_cycles);
        if (pipelined) {
            if ((err = hipEventRecord(parsed[j], side)) != hipSuccess || (err = hipStreamWaitEvent(stream, parsed[j], 0)) != hipSuccess)
                return err;
        }
        if (ev)
            (void)hipEventRecord(ev[1], stream);
        if (d_phase_cycles)
            hipLaunchKernelGGL(k_synthesize_frames<true>, dim3(n_frames), dim3(n_waves * 64), lds, stream, desc, q, residues, res_raw, state, n_frames,
                channels, v_begin, v_count, d_pcm_out, d_status, d_phase_cycles);
        else
            hipLaunchKernelGGL(k_synthesize_frames<false>, dim3(n_frames), dim3(n_waves * 64), lds, stream, desc, q, residues, res_raw, state, n_frames,
                channels, v_begin, v_count, d_pcm_out, d_status, d_phase_cycles);
        if (ev)
            (void)hipEventRecord(ev[2], stream);
    }
    return hipGetLastError();
}

} // namespace sela
