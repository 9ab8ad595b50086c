// sela_capi_generic.hip -- the host side of the any-length / 32-bit route (kernels: sela_generic.hip; declarations:
// include/sela_hip.h "Blocks of any length" and "the frame classes' own value types").
//
// Plain synchronous calls on the calling thread's current device and a stream of the calling thread's own (GenericContext below:
// threads that code a frame at a time through the frame classes run side by side on the device instead of taking turns on
// the default stream): copy in, kernels, copy out, in chunks of frames that keep the device scratch bounded, ONE wait for the
// device per chunk (round 6: the encoder sizes its Rice words and frame bytes by an estimate instead of waiting for its plan,
// and what the host reads first lands in page-locked memory of the context).  The scratch is one grow-only device allocation
// per thread (a frame at a time through frame::FrameEncoder must not pay a hipMalloc / hipFree pair per call);
// sela_hip_thread_release() / sela_hip_shutdown() and the thread's end give it back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sela_device.h"
#include "sela_generic.h"

namespace sela {
int report_error(int code, const std::string& what);      // sela_capi.hip: sets the thread's last error, returns code
int report_hip_error(hipError_t e, const char* where);    // (ENOMEM for an allocation failure, ENODEV otherwise)
}

namespace {

using sela::GenericMeta;
using sela::GenericSubInfo;

struct Arena { // one device allocation, handed out in aligned pieces for the length of a call
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0;
    bool fits() const { return used <= cap; } // (asked after the pieces are taken: the callers' size estimates are checked, not trusted)
    void release()
    {
        if (base)
            (void)hipFree(base);
        base = nullptr, cap = used = 0;
    }
    hipError_t reserve(size_t bytes)
    {
        used = 0;
        if (base && bytes <= cap)
            return hipSuccess;
        release();
        // (grow-only, and from a size that the batches of frame-at-a-time callers do not outgrow call after call: a regrowth is a
        // hipFree + hipMalloc, hundreds of microseconds in the middle of a batch)
        const size_t want = std::max<size_t>(bytes + (bytes >> 1), (size_t)16 << 20);
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), want);
        if (e != hipSuccess) {
            base = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    template <typename T>
    T* take(size_t count)
    {
        const size_t at = (used + 255) & ~(size_t)255;
        used = at + std::max<size_t>(count, 1) * sizeof(T);
        return reinterpret_cast<T*>(base + at);
    }
};

// What a calling thread needs on a device: scratch and a stream of its own (threads that code a frame at a time through the
// frame classes run side by side on the device instead of taking turns on the default stream).  Leased like the fast path's
// contexts (sela_capi.hip): a thread that ends -- programs start threads per job -- parks its set for the next thread on
// that device instead of paying a stream and an allocation again; sela_hip_shutdown() frees the parked ones.
struct PinnedScratch { // page-locked host memory for what a call reads back first (status words, offsets, a small call's whole output)
    uint8_t* base = nullptr;
    size_t cap = 0;
    void release()
    {
        if (base)
            (void)hipHostFree(base);
        base = nullptr, cap = 0;
    }
    uint8_t* reserve(size_t bytes) // null when the runtime has none to give: the caller copies into its own (pageable) memory then
    {
        if (base && bytes <= cap)
            return base;
        release();
        const size_t want = std::max<size_t>(bytes + (bytes >> 1), (size_t)3 << 20); // (grow-only; a regrowth is a hipHostFree + hipHostMalloc: milliseconds)
        if (hipHostMalloc(reinterpret_cast<void**>(&base), want, hipHostMallocDefault) != hipSuccess) {
            base = nullptr;
            return nullptr;
        }
        cap = want;
        return base;
    }
};

struct GenericContext {
    int device = -1;
    hipStream_t stream = nullptr;
    Arena arena;
    PinnedScratch pinned;
    void destroy()
    {
        int before = -1;
        (void)hipGetDevice(&before);
        if (device >= 0 && before != device)
            (void)hipSetDevice(device);
        arena.release();
        pinned.release();
        if (stream)
            (void)hipStreamDestroy(stream);
        stream = nullptr;
        if (before >= 0 && before != device)
            (void)hipSetDevice(before);
    }
};
struct ContextPark {
    std::mutex mu;
    std::vector<GenericContext*> idle;
};
ContextPark& park()
{
    static ContextPark* p = new ContextPark; // (never destroyed: threads may end after the statics)
    return *p;
}
constexpr size_t kParked = 64;

// A context's stream.  The runtime serves all streams through a few hardware queues (four) and gives a new stream the queue with the
// fewest streams on it at that moment; two contexts whose streams share a queue take turns on the device instead of running side
// by side.  With three coalesced decode jobs in flight, runs of the reference's thread loop over the frame classes came in two kinds:
// 280-300 M samples/s at 64 threads, or 180-190 (6 of 16 runs; with GPU_MAX_HW_QUEUES=8 or 16: 0 of 12).  So streams are made four
// at a time, one after the other, and handed out in that order, and the park hands back the context parked last: the contexts in use
// together are the ones created in a row, on different queues (2 of 16 runs of the slow kind since; the runtime's placement is
// not ours to decide, only to make less likely to hurt).
hipError_t fresh_stream(int dev, hipStream_t* out)
{
    static std::mutex mu;
    static std::vector<hipStream_t> bank[64];
    const int d = dev >= 0 && dev < 64 ? dev : 0;
    std::lock_guard<std::mutex> lock(mu);
    if (bank[d].empty()) {
        for (int i = 0; i < 4; i++) {
            hipStream_t s = nullptr;
            const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            if (e != hipSuccess) {
                if (bank[d].empty())
                    return e;
                (void)hipGetLastError();
                break;
            }
            bank[d].insert(bank[d].begin(), s); // (handed out from the back: in the order of creation)
        }
    }
    *out = bank[d].back();
    bank[d].pop_back();
    return hipSuccess;
}

struct Lease {
    GenericContext* held = nullptr;
    ~Lease() { give_back(); }
    void give_back()
    {
        if (!held)
            return;
        GenericContext* c = held;
        held = nullptr;
        {
            std::lock_guard<std::mutex> lock(park().mu);
            if (park().idle.size() < kParked) {
                park().idle.push_back(c);
                return;
            }
        }
        c->destroy();
        delete c;
    }
    // the calling thread's context on its current device (null + an error code when the runtime refuses)
    GenericContext* get(hipError_t& err)
    {
        int dev = -1;
        err = hipGetDevice(&dev);
        if (err != hipSuccess)
            return nullptr;
        if (held && held->device == dev)
            return held;
        give_back();
        {
            // the context parked LAST: the few that are in use at a time stay the same few, and those are the ones whose streams were
            // created one after the other (see fresh_stream)
            std::lock_guard<std::mutex> lock(park().mu);
            for (size_t i = park().idle.size(); i-- > 0;)
                if (park().idle[i]->device == dev) {
                    held = park().idle[i];
                    park().idle.erase(park().idle.begin() + (ptrdiff_t)i);
                    return held;
                }
        }
        GenericContext* c = new GenericContext;
        c->device = dev;
        err = fresh_stream(dev, &c->stream);
        if (err != hipSuccess) {
            delete c;
            return nullptr;
        }
        held = c;
        return held;
    }
};
thread_local Lease g_lease;

std::atomic<int> g_standard_first_mode{-1}; // sela_hip_debug_standard_first
std::atomic<int> g_standard_chunks{0};
std::atomic<long long> g_segment_subframes{0}; // subframes k_decode_subframes32 parsed by segments (sela_hip_debug_segment_subframes)

constexpr size_t kPiece = 256; // what take() may add per piece
constexpr size_t kChunkBudget = (size_t)768 << 20; // device scratch per chunk of frames

int device_ready()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return sela::report_error(SELA_HIP_ENODEV, "no HIP device visible (the SELA MI355X path has no CPU fallback)");
    return SELA_HIP_OK;
}

int flags_error(uint32_t flags, const char* who)
{
    if (flags & SELA_HIP_FLAG_SHORT_BLOCK)
        return sela::report_error(SELA_HIP_ERANGE, std::string(who) + ": a block is not longer than its predictor order (the reference reads past its vector there, src/lpc/residue_generator.cpp:104-110)");
    if (flags & SELA_HIP_FLAG_RICE_RANGE)
        return sela::report_error(SELA_HIP_ERANGE, std::string(who) + ": a residue is beyond the reference's int32 zig-zag (|value| >= 2^30)");
    if (flags & SELA_HIP_FLAG_COEF_OVERFLOW)
        return sela::report_error(SELA_HIP_ERANGE, std::string(who) + ": a predictor coefficient left the int64 range");
    if (flags & SELA_HIP_FLAG_WORDS_CAP)
        return sela::report_error(SELA_HIP_ERANGE, std::string(who) + ": a Rice stream needs more words than a subframe's 16-bit count can say");
    return SELA_HIP_OK;
}

} // namespace

namespace sela {

void generic_release() { g_lease.give_back(); }
void generic_shutdown()
{
    g_lease.give_back();
    std::vector<GenericContext*> idle;
    {
        std::lock_guard<std::mutex> lock(park().mu);
        idle.swap(park().idle);
    }
    for (GenericContext* c : idle) {
        c->destroy();
        delete c;
    }
}

size_t generic_encode_bound_bytes(uint32_t n_frames, uint32_t channels, uint32_t n)
{
    // first-minimum k is no worse than k = 19: (u >> 19) + 20 bits per value, u < 2^31 (beyond: ERANGE); a subframe holds at most
    // 65535 words (beyond: ERANGE)
    const uint64_t res_words = std::min<uint64_t>(65535, ((uint64_t)n * (4095 + 20) + 31) / 32);
    return (size_t)n_frames * (4 + (size_t)channels * (SELA_SUBFRAME_HEADER_BYTES + 4 * ((size_t)kCoefWordsCap + res_words)));
}

// input: int16 [n_frames][n][channels] (in16) or int32 [n_frames][channels][n]
// One submission per chunk of frames: copy in, analyse, plan, pack, assemble, copy out -- the Rice words and the frame bytes go
// to scratch sized by an ESTIMATE (4.5 bytes per sample: what 32-bit noise costs; 16-bit audio needs half), the kernels stay
// inside it whatever the data, and only a chunk whose plan turns out larger is packed and assembled again at its exact size
// (round 5 waited for the plan before it sized anything: two trips to the device per call).  What the host reads back first --
// status, sizes, offsets, and a small call's frames -- lands in page-locked memory of the calling thread's context.
constexpr size_t kEagerBytes = (size_t)4 << 20;

int generic_encode(const void* input, bool in16, uint32_t n_frames, uint32_t channels, uint32_t n, uint8_t* frames_out, size_t frames_cap,
    uint64_t* frame_offsets_out)
{
    if (device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    hipError_t ctx_err = hipSuccess;
    GenericContext* const ctx = g_lease.get(ctx_err);
    if (!ctx)
        return report_hip_error(ctx_err, "the calling thread's scratch and stream");
    Arena& g_arena = ctx->arena;
    const uint32_t n_sig = channels == 2 ? 3u : channels;
    const size_t in_frame_bytes = (size_t)n * channels * (in16 ? 2 : 4);
    const size_t est_frame_bytes = ((size_t)n * channels * 9) / 2 + (size_t)channels * 192 + 64; // words and frame bytes, each (a subframe: 12 bytes of header, up to 32 coefficient words)
    const size_t per_frame = (size_t)n_sig * n * 8 + (size_t)n_sig * (kMaxOrder * 4 + sizeof(GenericMeta) + 2 * kPiece) + in_frame_bytes + (size_t)channels * 12 + 8
        + 2 * est_frame_bytes;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, kChunkBudget / per_frame));
    uint64_t base_bytes = 0;
    frame_offsets_out[0] = 0;
    const hipStream_t st = ctx->stream;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const uint32_t cf = std::min(chunk, n_frames - f0);
        const size_t blocks = (size_t)cf * n_sig, subs = (size_t)cf * channels;
        const size_t est_words = ((size_t)cf * est_frame_bytes + 3) / 4, est_bytes = est_words * 4;
        const size_t fixed = blocks * n * 8 + blocks * (kMaxOrder * 4 + sizeof(GenericMeta)) + cf * in_frame_bytes + (subs + 1) * 12 + ((size_t)cf + 1) * 8 + 64 + 16 * kPiece
            + est_words * 4 + 8 + est_bytes;
        hipError_t e = g_arena.reserve(fixed);
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode: scratch");
        void* d_in = g_arena.take<uint8_t>(cf * in_frame_bytes);
        int32_t* d_sig = g_arena.take<int32_t>(blocks * n);
        int32_t* d_res = g_arena.take<int32_t>(blocks * n);
        int32_t* d_q = g_arena.take<int32_t>(blocks * kMaxOrder);
        GenericMeta* d_meta = g_arena.take<GenericMeta>(blocks);
        // what the host reads back first, in one piece: status (4 x u32) | total words | frame offsets
        const size_t head_words = 3 + (size_t)cf + 1;
        uint64_t* d_head = g_arena.take<uint64_t>(head_words);
        uint32_t* d_status = reinterpret_cast<uint32_t*>(d_head);
        uint64_t* d_offsets = d_head + 3;
        uint64_t* d_word_base = g_arena.take<uint64_t>(subs + 1);
        uint32_t* d_chosen = g_arena.take<uint32_t>(subs);
        uint32_t* d_words = g_arena.take<uint32_t>(est_words + 2);
        uint8_t* d_frames = g_arena.take<uint8_t>(est_bytes);
        if (!g_arena.fits())
            return report_error(SELA_HIP_ENOMEM, "generic encode: internal scratch estimate too small");
        const bool eager = est_bytes <= kEagerBytes;
        uint8_t* const pin = ctx->pinned.reserve(head_words * 8 + (eager ? est_bytes : 0));
        std::vector<uint64_t> head_pageable;
        uint64_t* head = reinterpret_cast<uint64_t*>(pin);
        if (!pin) {
            head_pageable.resize(head_words);
            head = head_pageable.data();
        }
        uint8_t* const eager_frames = pin && eager ? pin + head_words * 8 : nullptr;
        e = hipMemcpyAsync(d_in, static_cast<const uint8_t*>(input) + (size_t)f0 * in_frame_bytes, cf * in_frame_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(d_head, 0, 24, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(d_words, 0, (est_words + 2) * 4, st);
        if (e == hipSuccess)
            e = launch_generic_analyse(d_in, in16, cf, channels, n_sig, n, d_sig, d_res, d_q, d_meta, st);
        if (e == hipSuccess)
            e = launch_generic_plan(d_meta, cf, channels, n_sig, base_bytes, d_offsets, d_word_base, d_chosen, d_status, d_head + 2, st);
        if (e == hipSuccess)
            e = launch_generic_emit(d_meta, cf, channels, n_sig, n, d_res, d_q, d_chosen, d_word_base, d_words, est_words, d_offsets, base_bytes, d_frames, est_bytes, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(head, d_head, head_words * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && eager_frames)
            e = hipMemcpyAsync(eager_frames, d_frames, est_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode");
        uint32_t status[4];
        std::memcpy(status, head, 16);
        const uint64_t total_words = head[2];
        std::memcpy(frame_offsets_out + f0, head + 3, ((size_t)cf + 1) * 8);
        const int rc = flags_error(status[0], "encode");
        if (rc != SELA_HIP_OK)
            return rc;
        const uint64_t chunk_bytes = frame_offsets_out[f0 + cf] - base_bytes;
        if (frame_offsets_out[f0 + cf] > frames_cap)
            return report_error(SELA_HIP_ECAPACITY, "frames_out too small (see sela_hip_encode_bound_bytes_n)");
        if (total_words <= est_words && chunk_bytes <= est_bytes) {
            if (eager_frames) {
                std::memcpy(frames_out + base_bytes, eager_frames, (size_t)chunk_bytes);
            } else {
                e = hipMemcpyAsync(frames_out + base_bytes, d_frames, chunk_bytes, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(st);
            }
        } else { // (rare: data that codes to more than 4.5 bytes per sample) pack and assemble again, at the plan's exact sizes
            uint8_t* extra = nullptr;
            const size_t words_bytes = (((size_t)total_words + 2) * 4 + 255) & ~(size_t)255;
            e = hipMalloc(reinterpret_cast<void**>(&extra), words_bytes + chunk_bytes + 256);
            if (e != hipSuccess)
                return report_hip_error(e, "generic encode: stream scratch");
            uint32_t* d_words2 = reinterpret_cast<uint32_t*>(extra);
            uint8_t* d_frames2 = extra + words_bytes;
            e = hipMemsetAsync(d_words2, 0, ((size_t)total_words + 2) * 4, st);
            if (e == hipSuccess)
                e = launch_generic_emit(d_meta, cf, channels, n_sig, n, d_res, d_q, d_chosen, d_word_base, d_words2, total_words, d_offsets, base_bytes, d_frames2, chunk_bytes, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(frames_out + base_bytes, d_frames2, chunk_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            (void)hipFree(extra);
        }
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode: emit");
        base_bytes += chunk_bytes;
    }
    return SELA_HIP_OK;
}

// Walk a stream's headers on the host: per frame the first subframe's samplesPerChannel (running total in sample_offsets, if
// given), the largest samplesPerChannel of any subframe (returned; 0 when the walk falls off a frame), and whether every
// subframe says exactly 2048.
uint32_t generic_index_samples(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, uint64_t* sample_offsets,
    bool* all_standard)
{
    uint32_t largest = 0;
    bool standard = true, broken = false;
    uint64_t total = 0;
    for (uint32_t f = 0; f < n_frames; f++) {
        if (sample_offsets)
            sample_offsets[f] = total;
        const uint8_t* fb = frames + frame_offsets[f];
        const uint64_t fbytes = frame_offsets[f + 1] >= frame_offsets[f] ? frame_offsets[f + 1] - frame_offsets[f] : 0;
        uint64_t p = 4;
        uint32_t first = 0;
        for (uint32_t c = 0; c < channels; c++) {
            if (p + 12 > fbytes) {
                broken = true;
                break;
            }
            const uint64_t cw = (uint64_t)fb[p + 4] | ((uint64_t)fb[p + 5] << 8);
            const uint64_t p2 = p + 7 + 4 * cw;
            if (p2 + 5 > fbytes) {
                broken = true;
                break;
            }
            const uint64_t rw = (uint64_t)fb[p2 + 1] | ((uint64_t)fb[p2 + 2] << 8);
            const uint32_t n = (uint32_t)fb[p2 + 3] | ((uint32_t)fb[p2 + 4] << 8);
            if (c == 0)
                first = n;
            largest = std::max(largest, n);
            standard = standard && n == SELA_HIP_SAMPLES_PER_FRAME;
            p = p2 + 5 + 4 * rw;
            if (p > fbytes) {
                broken = true;
                break;
            }
        }
        total += first;
    }
    if (sample_offsets)
        sample_offsets[n_frames] = total;
    if (all_standard)
        *all_standard = standard && !broken;
    return broken ? 0 : largest;
}

// One of: samples_out + counts_out (32-bit, planar, stride), or pcm_out + sample_offsets (16-bit interleaved).
// Every chunk's subframes are first offered to k_decode_subframes32 (the fast decoder's lane-parallel parse and tuned synthesis
// with 32-bit samples, any length); a chunk in which that kernel leaves anything alone -- a frame that is not whole words at
// an aligned place, a malformed header, a stream that runs dry, coefficients outside the tables -- is decoded again by
// k_generic_decode, which reproduces what the reference does with such streams (or reports them).
int generic_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int32_t* samples_out, uint32_t stride,
    uint32_t* counts_out, int16_t* pcm_out, const uint64_t* sample_offsets)
{
    if (device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    hipError_t ctx_err = hipSuccess;
    GenericContext* const ctx = g_lease.get(ctx_err);
    if (!ctx)
        return report_hip_error(ctx_err, "the calling thread's scratch and stream");
    Arena& g_arena = ctx->arena;
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f])
            return report_error(SELA_HIP_EFORMAT, "frame offsets must not decrease");
    if (stride == 0)
        stride = 1;
    const size_t per_frame = (size_t)channels * stride * 8 + (size_t)channels * (sizeof(GenericSubInfo) + 4) + 16 + (size_t)channels * stride * (pcm_out ? 2 : 0);
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, kChunkBudget / per_frame));
    const hipStream_t st = ctx->stream;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const uint32_t cf = std::min(chunk, n_frames - f0);
        const size_t subs = (size_t)cf * channels;
        const uint64_t base_bytes = frame_offsets[f0], in_bytes = frame_offsets[f0 + cf] - base_bytes;
        const uint64_t s0 = pcm_out ? sample_offsets[f0] : 0, chunk_samples = pcm_out ? sample_offsets[f0 + cf] - s0 : 0;
        const size_t need = in_bytes + 8 + ((size_t)cf + 1) * 16 + subs * stride * 8 + subs * (sizeof(GenericSubInfo) + 4) + (size_t)chunk_samples * channels * 2 + 64
            + 12 * kPiece;
        hipError_t e = g_arena.reserve(need);
        if (e != hipSuccess)
            return report_hip_error(e, "generic decode: scratch");
        uint8_t* d_frames = g_arena.take<uint8_t>(in_bytes + 8);
        uint64_t* d_offsets = g_arena.take<uint64_t>((size_t)cf + 1);
        int32_t* d_dec = g_arena.take<int32_t>(subs * stride);
        int32_t* d_all = g_arena.take<int32_t>(subs * stride);
        GenericSubInfo* d_info = g_arena.take<GenericSubInfo>(subs);
        // what the host reads back first, in one piece: status (4 x u32) | a count per (frame, channel)
        uint32_t* d_tail = g_arena.take<uint32_t>(4 + subs);
        uint32_t* d_status = d_tail;
        uint32_t* d_counts = d_tail + 4;
        uint64_t* d_sample_offsets = pcm_out ? g_arena.take<uint64_t>((size_t)cf + 1) : nullptr;
        int16_t* d_pcm = pcm_out ? g_arena.take<int16_t>((size_t)chunk_samples * channels) : nullptr;
        if (!g_arena.fits())
            return report_error(SELA_HIP_ENOMEM, "generic decode: internal scratch estimate too small");
        // A small chunk's samples travel with the status, one wait for the device instead of two (a call of one frame -- the
        // frame classes' kind -- is mostly waits); what they are worth is known when both have arrived.  Both land in
        // page-locked memory of the calling thread's context (a copy into pageable memory is staged by the runtime, and waits)
        // -- unless the caller's own buffer is page-locked (a coalesced batch's staging, sela_hip_host_alloc): then the samples
        // go there at once (through the context's buffer and a memcpy, a batch of 50 stereo frames cost its leader 0.8 MB of
        // copying more).  The frame offsets take the same way in: from page-locked memory the copy is queued, not staged.
        const size_t out_bytes = pcm_out ? (size_t)chunk_samples * channels * 2 : subs * stride * sizeof(int32_t);
        const bool eager = out_bytes <= kEagerBytes;
        const size_t tail_bytes = ((4 + subs) * 4 + 15) & ~(size_t)15, offsets_bytes = (((size_t)cf + 1) * 8 + 15) & ~(size_t)15;
        uint8_t* const user_out = pcm_out ? reinterpret_cast<uint8_t*>(pcm_out + s0 * channels) : reinterpret_cast<uint8_t*>(samples_out + (size_t)f0 * channels * stride);
        bool user_locked = false;
        if (eager) {
            hipPointerAttribute_t attr;
            user_locked = hipPointerGetAttributes(&attr, user_out) == hipSuccess && attr.type == hipMemoryTypeHost;
            (void)hipGetLastError(); // (ordinary memory is not an error)
        }
        uint8_t* const pin = ctx->pinned.reserve(tail_bytes + offsets_bytes + (eager && !user_locked ? out_bytes : 0));
        const uint64_t* offsets_src = frame_offsets + f0;
        if (pin) {
            std::memcpy(pin + tail_bytes, frame_offsets + f0, ((size_t)cf + 1) * 8);
            offsets_src = reinterpret_cast<const uint64_t*>(pin + tail_bytes);
        }
        e = hipMemcpyAsync(d_frames, frames + base_bytes, in_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_offsets, offsets_src, ((size_t)cf + 1) * 8, hipMemcpyHostToDevice, st);
        std::vector<uint64_t> local;
        if (e == hipSuccess && pcm_out) { // positions relative to the chunk's first sample
            local.resize((size_t)cf + 1);
            for (uint32_t i = 0; i <= cf; i++)
                local[i] = sample_offsets[f0 + i] - s0;
            e = hipMemcpyAsync(d_sample_offsets, local.data(), local.size() * 8, hipMemcpyHostToDevice, st);
        }
        const int mode = g_standard_first_mode.load(std::memory_order_relaxed); // -1: the product; 0: the serial kernel alone; 1: as -1; 2: offered, every subframe by segments
        const bool offer = mode != 0, standard_path = mode != 2;
        std::vector<uint32_t> tail_pageable;
        uint32_t* tail = reinterpret_cast<uint32_t*>(pin);
        if (!pin) {
            tail_pageable.resize(4 + subs);
            tail = tail_pageable.data();
        }
        uint8_t* const eager_out = pin && eager && !user_locked ? pin + tail_bytes + offsets_bytes : nullptr;
        const void* const device_out = pcm_out ? static_cast<const void*>(d_pcm) : static_cast<const void*>(d_all);
        // samples_out and the device's channel-major array have one layout ([frame][channel][stride]): ONE copy (a copy per
        // channel cost 13 us each: 7 ms for 256 stereo frames).  What lies behind a channel's count is not defined.
        for (int attempt = offer ? 0 : 1; attempt < 2; attempt++) {
            if (e == hipSuccess)
                e = hipMemsetAsync(d_tail, 0, (4 + subs) * 4, st);
            if (e == hipSuccess)
                e = launch_generic_decode(d_frames, d_offsets, base_bytes, cf, channels, stride, d_dec, d_info, d_all, d_counts, d_sample_offsets, d_pcm, d_status,
                    attempt == 0, standard_path, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(tail, d_tail, (4 + subs) * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && eager)
                e = hipMemcpyAsync(eager_out ? eager_out : user_out, device_out, out_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            if (e != hipSuccess)
                return report_hip_error(e, "generic decode");
            if (attempt == 0 && tail[2] == 0) { // (the fast kernel took every subframe and every one came out clean)
                g_standard_chunks.fetch_add(1, std::memory_order_relaxed);
                g_segment_subframes.fetch_add(tail[3], std::memory_order_relaxed);
                break;
            }
        }
        const uint32_t* const status = tail;
        if (status[0] & SELA_HIP_FLAG_BAD_FRAME)
            return report_error(SELA_HIP_EFORMAT, "malformed frame (sync word, sizes, an order above 100, a Rice parameter above 31, a channel or parent that does not exist, or channels of different lengths)");
        if (status[0] & SELA_HIP_FLAG_RICE_OVERRUN)
            return report_error(SELA_HIP_EFORMAT, "a Rice stream ended before all its values were read");
        // what the reference itself leaves undefined is reported, never decoded silently (one policy: here, in the streaming
        // jobs' sela_hip_decode_end and in sela_hip_lpc_decode[_n])
        if (status[0] & SELA_HIP_FLAG_COEF_OVERFLOW)
            return report_error(SELA_HIP_ERANGE, "decode: a predictor coefficient left the int64 range");
        if (status[0] & SELA_HIP_FLAG_Q_RANGE)
            return report_error(SELA_HIP_ERANGE, "decode: a quantised reflection coefficient outside [-64, 63] (the reference indexes past its tables, src/lpc/linear_predictor.cpp:23-26)");
        if (status[0] & SELA_HIP_FLAG_SHORT_BLOCK)
            return report_error(SELA_HIP_ERANGE, "decode: a subframe without samples or not longer than its predictor order (the reference writes past its vector, src/lpc/sample_generator.cpp:14-22)");
        if (!pcm_out)
            std::memcpy(counts_out + (size_t)f0 * channels, tail + 4, subs * 4);
        if (eager_out) {
            std::memcpy(user_out, eager_out, out_bytes);
        } else if (!eager) {
            e = hipMemcpyAsync(user_out, device_out, out_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess)
            return report_hip_error(e, "generic decode: copy out");
    }
    return SELA_HIP_OK;
}

int generic_lpc_encode(const int32_t* samples, uint32_t n_blocks, uint32_t n, int32_t* order_out, int32_t* q_out, int32_t* residues_out)
{
    if (device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    hipError_t ctx_err = hipSuccess;
    GenericContext* const ctx = g_lease.get(ctx_err);
    if (!ctx)
        return report_hip_error(ctx_err, "the calling thread's scratch and stream");
    Arena& g_arena = ctx->arena;
    const size_t per_block = (size_t)n * 12 + kMaxOrder * 4 + sizeof(GenericMeta) + 16;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_blocks, kChunkBudget / per_block));
    std::vector<GenericMeta> meta;
    const hipStream_t st = ctx->stream;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += chunk) {
        const uint32_t cb = std::min(chunk, n_blocks - b0);
        hipError_t e = g_arena.reserve((size_t)cb * per_block + 8 * kPiece);
        if (e != hipSuccess)
            return report_hip_error(e, "lpc_encode: scratch");
        int32_t* d_in = g_arena.take<int32_t>((size_t)cb * n);
        int32_t* d_sig = g_arena.take<int32_t>((size_t)cb * n);
        int32_t* d_res = g_arena.take<int32_t>((size_t)cb * n);
        int32_t* d_q = g_arena.take<int32_t>((size_t)cb * kMaxOrder);
        GenericMeta* d_meta = g_arena.take<GenericMeta>(cb);
        if (!g_arena.fits())
            return report_error(SELA_HIP_ENOMEM, "lpc_encode: internal scratch estimate too small");
        e = hipMemcpyAsync(d_in, samples + (size_t)b0 * n, (size_t)cb * n * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) // a "frame" of one channel per block
            e = launch_generic_analyse(d_in, false, cb, 1, 1, n, d_sig, d_res, d_q, d_meta, st);
        meta.resize(cb);
        if (e == hipSuccess)
            e = hipMemcpyAsync(meta.data(), d_meta, (size_t)cb * sizeof(GenericMeta), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(q_out + (size_t)b0 * kMaxOrder, d_q, (size_t)cb * kMaxOrder * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(residues_out + (size_t)b0 * n, d_res, (size_t)cb * n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return report_hip_error(e, "lpc_encode");
        uint32_t flags = 0;
        for (uint32_t i = 0; i < cb; i++) {
            order_out[b0 + i] = (int32_t)meta[i].order;
            flags |= meta[i].flags;
        }
        // (a residue beyond the zig-zag's range or a long Rice stream is the Rice stage's business, not this one's)
        const int rc = flags_error(flags & (SELA_HIP_FLAG_SHORT_BLOCK | SELA_HIP_FLAG_COEF_OVERFLOW), "lpc_encode");
        if (rc != SELA_HIP_OK)
            return rc;
    }
    return SELA_HIP_OK;
}

int generic_lpc_decode(const int32_t* order, const int32_t* q, const int32_t* residues, uint32_t n_blocks, uint32_t n, int32_t* samples_out, int64_t* coefs_out)
{
    if (device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    hipError_t ctx_err = hipSuccess;
    GenericContext* const ctx = g_lease.get(ctx_err);
    if (!ctx)
        return report_hip_error(ctx_err, "the calling thread's scratch and stream");
    Arena& g_arena = ctx->arena;
    constexpr size_t kCoefs = kMaxOrder + 1;
    const hipStream_t st = ctx->stream;
    const size_t per_block = (size_t)n * 8 + kMaxOrder * 4 + 4 + kCoefs * 8;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_blocks, kChunkBudget / per_block));
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += chunk) {
        const uint32_t cb = std::min(chunk, n_blocks - b0);
        hipError_t e = g_arena.reserve((size_t)cb * per_block + 64 + 8 * kPiece);
        if (e != hipSuccess)
            return report_hip_error(e, "lpc_decode: scratch");
        int32_t* d_order = g_arena.take<int32_t>(cb);
        int32_t* d_q = g_arena.take<int32_t>((size_t)cb * kMaxOrder);
        int32_t* d_res = samples_out ? g_arena.take<int32_t>((size_t)cb * n) : nullptr;
        int32_t* d_out = samples_out ? g_arena.take<int32_t>((size_t)cb * n) : nullptr;
        int64_t* d_coefs = coefs_out ? g_arena.take<int64_t>((size_t)cb * kCoefs) : nullptr;
        uint32_t* d_status = g_arena.take<uint32_t>(4);
        if (!g_arena.fits())
            return report_error(SELA_HIP_ENOMEM, "lpc_decode: internal scratch estimate too small");
        e = hipMemcpyAsync(d_order, order + b0, (size_t)cb * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_q, q + (size_t)b0 * kMaxOrder, (size_t)cb * kMaxOrder * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && d_res)
            e = hipMemcpyAsync(d_res, residues + (size_t)b0 * n, (size_t)cb * n * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(d_status, 0, 16, st);
        if (e == hipSuccess && d_coefs)
            e = hipMemsetAsync(d_coefs, 0, (size_t)cb * kCoefs * 8, st);
        if (e == hipSuccess)
            e = launch_lpc_decode_any(d_order, d_q, d_res, cb, n, d_out, d_coefs, d_status, st);
        uint32_t status[4] = {};
        if (e == hipSuccess)
            e = hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && samples_out)
            e = hipMemcpyAsync(samples_out + (size_t)b0 * n, d_out, (size_t)cb * n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && coefs_out)
            e = hipMemcpyAsync(coefs_out + (size_t)b0 * kCoefs, d_coefs, (size_t)cb * kCoefs * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return report_hip_error(e, "lpc_decode");
        if (status[0] & SELA_HIP_FLAG_BAD_FRAME)
            return report_error(SELA_HIP_EINVAL, "lpc_decode: order outside 0..100");
        if (status[0] & SELA_HIP_FLAG_COEF_OVERFLOW)
            return report_error(SELA_HIP_ERANGE, "lpc_decode: a predictor coefficient left the int64 range");
        if (status[0] & SELA_HIP_FLAG_Q_RANGE)
            return report_error(SELA_HIP_ERANGE, "lpc_decode: a quantised reflection coefficient outside [-64, 63] (the reference indexes past its tables, src/lpc/linear_predictor.cpp:23-26)");
        if (status[0] & SELA_HIP_FLAG_SHORT_BLOCK)
            return report_error(SELA_HIP_ERANGE, "lpc_decode: a block without samples or not longer than its predictor order (the reference writes past its vector, src/lpc/sample_generator.cpp:14-22)");
    }
    return SELA_HIP_OK;
}

} // namespace sela

extern "C" {
void sela_hip_debug_standard_first(int mode) { g_standard_first_mode.store(mode, std::memory_order_relaxed); }
int sela_hip_debug_standard_chunks(void) { return g_standard_chunks.load(std::memory_order_relaxed); }
long long sela_hip_debug_segment_subframes(void) { return g_segment_subframes.load(std::memory_order_relaxed); }
void sela_hip_debug_generic_wrap_taps(int on) { sela::set_generic_wrap_taps(on); }
}
