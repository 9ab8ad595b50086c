// sela_capi_generic.hip -- the host side of the any-length / 32-bit route (kernels: sela_generic.hip; declarations:
// include/sela_hip.h "Blocks of any length" and "the frame classes' own value types").
//
// Plain synchronous calls on the calling thread's current device and a stream of the calling thread's own (GenericContext below:
// threads that code a frame at a time through the frame classes run side by side on the device instead of taking turns on
// the default stream): copy in, kernels, copy out, in chunks of frames that keep the device scratch bounded, with as few
// waits for the device as the data flow allows (an encode: two; a decode: two).  The scratch is one grow-only device allocation per thread (a frame at a time through
// frame::FrameEncoder must not pay a hipMalloc / hipFree pair per call); sela_hip_thread_release() / sela_hip_shutdown() and
// the thread's end give it back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
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
        const size_t want = std::max<size_t>(bytes + (bytes >> 2), 1 << 20);
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
        const size_t want = std::max<size_t>(bytes + (bytes >> 2), 1 << 16);
        if (hipHostMalloc(reinterpret_cast<void**>(&base), want, hipHostMallocDefault) != hipSuccess) {
            base = nullptr;
            return nullptr;
        }
        cap = want;
        return base;
    }
};

// A large call is cut into pieces that alternate between two LANES -- a stream, device scratch and page-locked staging each --
// so that one piece's copies (host memory -> staging by the calling thread, staging <-> device by the copy engines) run beside
// the other piece's kernels.  A small call uses the first lane only.
struct Lane {
    hipStream_t stream = nullptr;
    Arena arena;
    PinnedScratch pinned;
};
struct GenericContext {
    int device = -1;
    Lane lane[2];
    hipStream_t& stream = lane[0].stream;
    Arena& arena = lane[0].arena;
    PinnedScratch& pinned = lane[0].pinned;
    Lane* second(hipError_t& err) // (its stream is made when a call first needs it)
    {
        err = hipSuccess;
        if (!lane[1].stream)
            err = hipStreamCreateWithFlags(&lane[1].stream, hipStreamNonBlocking);
        return err == hipSuccess ? &lane[1] : nullptr;
    }
    void destroy()
    {
        int before = -1;
        (void)hipGetDevice(&before);
        if (device >= 0 && before != device)
            (void)hipSetDevice(device);
        for (Lane& l : lane) {
            l.arena.release();
            l.pinned.release();
            if (l.stream)
                (void)hipStreamDestroy(l.stream);
            l.stream = nullptr;
        }
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
            std::lock_guard<std::mutex> lock(park().mu);
            for (size_t i = 0; i < park().idle.size(); i++)
                if (park().idle[i]->device == dev) {
                    held = park().idle[i];
                    park().idle.erase(park().idle.begin() + (ptrdiff_t)i);
                    return held;
                }
        }
        GenericContext* c = new GenericContext;
        c->device = dev;
        err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
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
constexpr size_t kPipelinePiece = (size_t)6 << 20; // host bytes (in + out) of one piece of a call that is cut up for the two lanes

// host memory to host memory, by up to four threads when it is worth their start (a core copies ~10 GB/s; the staging of a large
// call would otherwise be what the call waits for)
void copy_bytes(void* dst, const void* src, size_t bytes)
{
    constexpr size_t kPerThread = (size_t)2 << 20;
    const size_t parts = std::min<size_t>(4, bytes / kPerThread);
    if (parts < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> helpers;
    const size_t step = (bytes / parts + 63) & ~(size_t)63;
    for (size_t i = 1; i < parts; i++) {
        const size_t at = i * step, len = i + 1 == parts ? bytes - at : step;
        helpers.emplace_back([=] { std::memcpy(static_cast<uint8_t*>(dst) + at, static_cast<const uint8_t*>(src) + at, len); });
    }
    std::memcpy(dst, src, step);
    for (std::thread& t : helpers)
        t.join();
}

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
    const uint32_t n_sig = channels == 2 ? 3u : channels;
    const size_t in_frame_bytes = (size_t)n * channels * (in16 ? 2 : 4);
    const size_t est_frame_bytes = ((size_t)n * channels * 9) / 2 + (size_t)channels * 64 + 64; // words and frame bytes, each
    const size_t per_frame = (size_t)n_sig * n * 8 + (size_t)n_sig * (kMaxOrder * 4 + sizeof(GenericMeta) + 2 * kPiece) + in_frame_bytes + (size_t)channels * 12 + 8
        + 2 * est_frame_bytes;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, kChunkBudget / per_frame));
    frame_offsets_out[0] = 0;
    if (n_frames == 0)
        return SELA_HIP_OK;
    // pieces: the whole call when it is small; pieces of about kPipelinePiece bytes of input, alternating between two lanes, when
    // it is large -- a piece's frames land where the piece before it ended, which the host knows when that piece is finished
    // (its offsets are relative until then: the plan kernel starts every piece at 0)
    Lane* lanes[2] = { &ctx->lane[0], &ctx->lane[0] };
    if ((size_t)n_frames * in_frame_bytes > 2 * kPipelinePiece && n_frames > 1) {
        hipError_t e2 = hipSuccess;
        if (Lane* l = ctx->second(e2)) {
            lanes[1] = l;
            chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(chunk, kPipelinePiece / in_frame_bytes));
        }
    }
    const bool piped = lanes[1] != lanes[0];

    struct Piece {
        uint32_t f0 = 0, cf = 0;
        Lane* lane = nullptr;
        size_t blocks = 0, subs = 0, est_words = 0, est_bytes = 0, head_words = 0;
        int32_t *d_res = nullptr, *d_q = nullptr;
        GenericMeta* d_meta = nullptr;
        uint64_t *d_head = nullptr, *d_word_base = nullptr;
        uint32_t* d_chosen = nullptr;
        uint8_t* d_frames = nullptr;
        uint64_t* head = nullptr;        // host: status (4 x u32) | total words | frame offsets (relative to the piece)
        uint8_t* staged_frames = nullptr; // host, page-locked: the piece's frames at their estimated size (null: copied when the size is known)
        std::vector<uint64_t> head_pageable;
        bool live = false;
    } piece[2];
    uint64_t base_bytes = 0; // where the next piece to FINISH starts in frames_out

    auto begin = [&](Piece& p, uint32_t f0, uint32_t cf, Lane* lane) -> int {
        p.f0 = f0, p.cf = cf, p.lane = lane;
        p.blocks = (size_t)cf * n_sig, p.subs = (size_t)cf * channels;
        p.est_words = ((size_t)cf * est_frame_bytes + 3) / 4, p.est_bytes = p.est_words * 4;
        Arena& arena = lane->arena;
        const size_t fixed = p.blocks * n * 8 + p.blocks * (kMaxOrder * 4 + sizeof(GenericMeta)) + cf * in_frame_bytes + (p.subs + 1) * 12 + ((size_t)cf + 1) * 8 + 64 + 16 * kPiece
            + p.est_words * 4 + 8 + p.est_bytes;
        hipError_t e = arena.reserve(fixed);
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode: scratch");
        void* d_in = arena.take<uint8_t>(cf * in_frame_bytes);
        int32_t* d_sig = arena.take<int32_t>(p.blocks * n);
        p.d_res = arena.take<int32_t>(p.blocks * n);
        p.d_q = arena.take<int32_t>(p.blocks * kMaxOrder);
        p.d_meta = arena.take<GenericMeta>(p.blocks);
        p.head_words = 3 + (size_t)cf + 1;
        p.d_head = arena.take<uint64_t>(p.head_words);
        p.d_word_base = arena.take<uint64_t>(p.subs + 1);
        p.d_chosen = arena.take<uint32_t>(p.subs);
        uint32_t* d_words = arena.take<uint32_t>(p.est_words + 2);
        p.d_frames = arena.take<uint8_t>(p.est_bytes);
        if (!arena.fits())
            return report_error(SELA_HIP_ENOMEM, "generic encode: internal scratch estimate too small");
        // page-locked staging of this lane: head | frames out | (a piece of a large call:) samples in
        const bool stage_out = p.est_bytes <= kEagerBytes || piped;
        const size_t head_bytes = (p.head_words * 8 + 63) & ~(size_t)63;
        uint8_t* const pin = lane->pinned.reserve(head_bytes + (stage_out ? p.est_bytes : 0) + (piped ? cf * in_frame_bytes : 0) + 64);
        p.head = reinterpret_cast<uint64_t*>(pin);
        if (!pin) {
            p.head_pageable.resize(p.head_words);
            p.head = p.head_pageable.data();
        }
        p.staged_frames = pin && stage_out ? pin + head_bytes : nullptr;
        const hipStream_t st = lane->stream;
        const uint8_t* src = static_cast<const uint8_t*>(input) + (size_t)f0 * in_frame_bytes;
        if (pin && piped) { // (from staging the copy is asynchronous: the calling thread goes on to the other lane's piece)
            uint8_t* const staged_in = pin + head_bytes + (stage_out ? p.est_bytes : 0);
            copy_bytes(staged_in, src, cf * in_frame_bytes);
            src = staged_in;
        }
        e = hipMemcpyAsync(d_in, src, cf * in_frame_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(p.d_head, 0, 24, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(d_words, 0, (p.est_words + 2) * 4, st);
        if (e == hipSuccess)
            e = launch_generic_analyse(d_in, in16, cf, channels, n_sig, n, d_sig, p.d_res, p.d_q, p.d_meta, st);
        if (e == hipSuccess)
            e = launch_generic_plan(p.d_meta, cf, channels, n_sig, 0, p.d_head + 3, p.d_word_base, p.d_chosen, reinterpret_cast<uint32_t*>(p.d_head), p.d_head + 2, st);
        if (e == hipSuccess)
            e = launch_generic_emit(p.d_meta, cf, channels, n_sig, n, p.d_res, p.d_q, p.d_chosen, p.d_word_base, d_words, p.est_words, p.d_head + 3, 0, p.d_frames, p.est_bytes, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(p.head, p.d_head, p.head_words * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && p.staged_frames)
            e = hipMemcpyAsync(p.staged_frames, p.d_frames, p.est_bytes, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode");
        p.live = true;
        return SELA_HIP_OK;
    };
    auto finish = [&](Piece& p) -> int {
        if (!p.live)
            return SELA_HIP_OK;
        p.live = false;
        const hipStream_t st = p.lane->stream;
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode");
        uint32_t status[4];
        std::memcpy(status, p.head, 16);
        const uint64_t total_words = p.head[2];
        for (uint32_t i = 0; i <= p.cf; i++)
            frame_offsets_out[p.f0 + i] = base_bytes + p.head[3 + i];
        const int rc = flags_error(status[0], "encode");
        if (rc != SELA_HIP_OK)
            return rc;
        const uint64_t chunk_bytes = p.head[3 + p.cf];
        if (base_bytes + chunk_bytes > frames_cap)
            return report_error(SELA_HIP_ECAPACITY, "frames_out too small (see sela_hip_encode_bound_bytes_n)");
        if (total_words <= p.est_words && chunk_bytes <= p.est_bytes) {
            if (p.staged_frames) {
                copy_bytes(frames_out + base_bytes, p.staged_frames, (size_t)chunk_bytes);
            } else {
                e = hipMemcpyAsync(frames_out + base_bytes, p.d_frames, chunk_bytes, hipMemcpyDeviceToHost, st);
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
                e = launch_generic_emit(p.d_meta, p.cf, channels, n_sig, n, p.d_res, p.d_q, p.d_chosen, p.d_word_base, d_words2, total_words, p.d_head + 3, 0, d_frames2, chunk_bytes, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(frames_out + base_bytes, d_frames2, chunk_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            (void)hipFree(extra);
        }
        if (e != hipSuccess)
            return report_hip_error(e, "generic encode: emit");
        base_bytes += chunk_bytes;
        return SELA_HIP_OK;
    };
    auto drain = [&]() {
        for (Piece& p : piece)
            if (p.live)
                (void)hipStreamSynchronize(p.lane->stream), p.live = false;
    };
    int k = 0;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk, k++) {
        Piece& p = piece[k & 1];
        int rc = finish(p); // (the piece that used this lane before: pieces finish in the order they began)
        if (rc == SELA_HIP_OK)
            rc = begin(p, f0, std::min(chunk, n_frames - f0), lanes[k & 1]);
        if (rc != SELA_HIP_OK) {
            drain();
            return rc;
        }
        if (!piped) {
            rc = finish(p);
            if (rc != SELA_HIP_OK)
                return rc;
        }
    }
    for (int i = 0; i < 2; i++) {
        const int rc = finish(piece[(k + i) & 1]);
        if (rc != SELA_HIP_OK) {
            drain();
            return rc;
        }
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
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f])
            return report_error(SELA_HIP_EFORMAT, "frame offsets must not decrease");
    if (stride == 0)
        stride = 1;
    if (n_frames == 0)
        return SELA_HIP_OK;
    const int mode = g_standard_first_mode.load(std::memory_order_relaxed); // -1: the product; 0: the serial kernel alone; 1: as -1; 2: offered, every subframe by segments
    const bool offer = mode != 0, standard_path = mode != 2;
    // ---- pieces: the whole call when it is small (its samples then travel with the status: one wait for the device -- a call of
    //      one frame, the frame classes' kind, is mostly waits); pieces of about kPipelinePiece host bytes, alternating between two
    //      lanes, when it is large ----
    const size_t out_per_frame = pcm_out ? 0 : (size_t)channels * stride * sizeof(int32_t);
    const size_t total_out = pcm_out ? (size_t)(sample_offsets[n_frames] - sample_offsets[0]) * channels * 2 : (size_t)n_frames * out_per_frame;
    const size_t total_host = total_out + (size_t)(frame_offsets[n_frames] - frame_offsets[0]);
    const size_t per_frame_device = (size_t)channels * stride * 8 + (size_t)channels * (sizeof(GenericSubInfo) + 4) + 16 + (size_t)channels * stride * (pcm_out ? 2 : 0);
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, kChunkBudget / per_frame_device));
    Lane* lanes[2] = { &ctx->lane[0], &ctx->lane[0] };
    if (total_host > 2 * kPipelinePiece && n_frames > 1) {
        hipError_t e2 = hipSuccess;
        if (Lane* l = ctx->second(e2)) {
            lanes[1] = l;
            const size_t per_frame_host = std::max<size_t>(1, total_host / n_frames);
            chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(chunk, kPipelinePiece / per_frame_host));
        }
    }
    const bool piped = lanes[1] != lanes[0];

    struct Piece {
        uint32_t f0 = 0, cf = 0;
        Lane* lane = nullptr;
        size_t subs = 0, out_bytes = 0;
        uint64_t base_bytes = 0, s0 = 0;
        uint8_t* d_frames = nullptr;
        uint64_t* d_offsets = nullptr;
        int32_t *d_dec = nullptr, *d_all = nullptr;
        GenericSubInfo* d_info = nullptr;
        uint32_t* d_tail = nullptr;
        uint64_t* d_sample_offsets = nullptr;
        int16_t* d_pcm = nullptr;
        uint32_t* tail = nullptr;       // host: status (4 x u32) | a count per (frame, channel)
        uint8_t* staged_out = nullptr;  // host, page-locked: the piece's samples (null: they go straight to the caller's memory)
        std::vector<uint32_t> tail_pageable;
        std::vector<uint64_t> offsets_pageable; // (without page-locked staging: must outlive the copy that reads it)
        bool live = false;
    } piece[2];

    auto user_out_of = [&](const Piece& p) -> uint8_t* {
        return pcm_out ? reinterpret_cast<uint8_t*>(pcm_out + p.s0 * channels) : reinterpret_cast<uint8_t*>(samples_out + (size_t)p.f0 * channels * stride);
    };
    auto launch = [&](Piece& p, bool fast_first) -> hipError_t {
        const hipStream_t st = p.lane->stream;
        hipError_t e = hipMemsetAsync(p.d_tail, 0, (4 + p.subs) * 4, st);
        if (e == hipSuccess)
            e = launch_generic_decode(p.d_frames, p.d_offsets, p.base_bytes, p.cf, channels, stride, p.d_dec, p.d_info, p.d_all, p.d_tail + 4, p.d_sample_offsets, p.d_pcm,
                p.d_tail, fast_first, standard_path, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(p.tail, p.d_tail, (4 + p.subs) * 4, hipMemcpyDeviceToHost, st);
        const void* const device_out = pcm_out ? static_cast<const void*>(p.d_pcm) : static_cast<const void*>(p.d_all);
        // samples_out and the device's channel-major array have one layout ([frame][channel][stride]): ONE copy
        if (e == hipSuccess && (p.staged_out || p.out_bytes <= kEagerBytes))
            e = hipMemcpyAsync(p.staged_out ? p.staged_out : user_out_of(p), device_out, p.out_bytes, hipMemcpyDeviceToHost, st);
        return e;
    };
    auto begin = [&](Piece& p, uint32_t f0, uint32_t cf, Lane* lane) -> int {
        p.f0 = f0, p.cf = cf, p.lane = lane, p.subs = (size_t)cf * channels;
        p.base_bytes = frame_offsets[f0];
        const uint64_t in_bytes = frame_offsets[f0 + cf] - p.base_bytes;
        p.s0 = pcm_out ? sample_offsets[f0] : 0;
        const uint64_t chunk_samples = pcm_out ? sample_offsets[f0 + cf] - p.s0 : 0;
        p.out_bytes = pcm_out ? (size_t)chunk_samples * channels * 2 : p.subs * stride * sizeof(int32_t);
        Arena& arena = lane->arena;
        const size_t need = in_bytes + 8 + ((size_t)cf + 1) * 16 + p.subs * stride * 8 + p.subs * (sizeof(GenericSubInfo) + 4) + (size_t)chunk_samples * channels * 2 + 64 + 12 * kPiece;
        hipError_t e = arena.reserve(need);
        if (e != hipSuccess)
            return report_hip_error(e, "generic decode: scratch");
        p.d_frames = arena.take<uint8_t>(in_bytes + 8);
        p.d_offsets = arena.take<uint64_t>((size_t)cf + 1);
        p.d_dec = arena.take<int32_t>(p.subs * stride);
        p.d_all = arena.take<int32_t>(p.subs * stride);
        p.d_info = arena.take<GenericSubInfo>(p.subs);
        p.d_tail = arena.take<uint32_t>(4 + p.subs);
        p.d_sample_offsets = pcm_out ? arena.take<uint64_t>((size_t)cf + 1) : nullptr;
        p.d_pcm = pcm_out ? arena.take<int16_t>((size_t)chunk_samples * channels) : nullptr;
        if (!arena.fits())
            return report_error(SELA_HIP_ENOMEM, "generic decode: internal scratch estimate too small");
        // page-locked staging of this lane: tail | offsets | samples out | (a piece of a large call:) frames in
        const size_t tail_bytes = ((4 + p.subs) * 4 + 15) & ~(size_t)15, offs_bytes = 2 * ((size_t)cf + 1) * 8;
        const bool stage_out = p.out_bytes <= kEagerBytes || piped;
        uint8_t* const pin = lane->pinned.reserve(tail_bytes + offs_bytes + (stage_out ? p.out_bytes : 0) + (piped ? (size_t)in_bytes : 0) + 64);
        p.tail = reinterpret_cast<uint32_t*>(pin);
        p.staged_out = pin && stage_out ? pin + tail_bytes + offs_bytes : nullptr;
        if (!pin) {
            p.tail_pageable.resize(4 + p.subs);
            p.tail = p.tail_pageable.data();
        }
        const hipStream_t st = lane->stream;
        const uint8_t* src_frames = frames + p.base_bytes;
        if (pin && piped) { // (from staging the copy is asynchronous: the calling thread goes on to the other lane's piece)
            uint8_t* const staged_in = pin + tail_bytes + offs_bytes + (stage_out ? p.out_bytes : 0);
            copy_bytes(staged_in, src_frames, (size_t)in_bytes);
            src_frames = staged_in;
        }
        e = hipMemcpyAsync(p.d_frames, src_frames, in_bytes, hipMemcpyHostToDevice, st);
        uint64_t* const offs_host = pin ? reinterpret_cast<uint64_t*>(pin + tail_bytes) : nullptr;
        if (e == hipSuccess) {
            if (offs_host) {
                std::memcpy(offs_host, frame_offsets + f0, ((size_t)cf + 1) * 8);
                e = hipMemcpyAsync(p.d_offsets, offs_host, ((size_t)cf + 1) * 8, hipMemcpyHostToDevice, st);
            } else {
                e = hipMemcpyAsync(p.d_offsets, frame_offsets + f0, ((size_t)cf + 1) * 8, hipMemcpyHostToDevice, st);
            }
        }
        if (e == hipSuccess && pcm_out) { // positions relative to the piece's first sample
            uint64_t* local = offs_host ? offs_host + cf + 1 : nullptr;
            if (!local) {
                p.offsets_pageable.resize((size_t)cf + 1);
                local = p.offsets_pageable.data();
            }
            for (uint32_t i = 0; i <= cf; i++)
                local[i] = sample_offsets[f0 + i] - p.s0;
            e = hipMemcpyAsync(p.d_sample_offsets, local, ((size_t)cf + 1) * 8, hipMemcpyHostToDevice, st);
        }
        if (e == hipSuccess)
            e = launch(p, offer);
        if (e != hipSuccess)
            return report_hip_error(e, "generic decode");
        p.live = true;
        return SELA_HIP_OK;
    };
    auto finish = [&](Piece& p) -> int {
        if (!p.live)
            return SELA_HIP_OK;
        p.live = false;
        const hipStream_t st = p.lane->stream;
        hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess && offer) {
            if (p.tail[2] == 0) { // (the fast kernel took every subframe and every one came out clean)
                g_standard_chunks.fetch_add(1, std::memory_order_relaxed);
                g_segment_subframes.fetch_add(p.tail[3], std::memory_order_relaxed);
            } else { // something the fast kernel will not judge: the piece again on the serial kernel
                e = launch(p, false);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(st);
            }
        }
        if (e != hipSuccess)
            return report_hip_error(e, "generic decode");
        const uint32_t* const status = p.tail;
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
            std::memcpy(counts_out + (size_t)p.f0 * channels, p.tail + 4, p.subs * 4);
        if (p.staged_out) {
            copy_bytes(user_out_of(p), p.staged_out, p.out_bytes);
        } else if (p.out_bytes > kEagerBytes) { // (no staging: a large piece straight into the caller's memory, once its verdict is in)
            e = hipMemcpyAsync(user_out_of(p), pcm_out ? static_cast<const void*>(p.d_pcm) : static_cast<const void*>(p.d_all), p.out_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            if (e != hipSuccess)
                return report_hip_error(e, "generic decode: copy out");
        }
        return SELA_HIP_OK;
    };
    auto drain = [&]() { // after an error: nothing of this call may still be running when its buffers go back
        for (Piece& p : piece)
            if (p.live)
                (void)hipStreamSynchronize(p.lane->stream), p.live = false;
    };
    int k = 0;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk, k++) {
        Piece& p = piece[k & 1];
        int rc = finish(p); // (the piece that used this lane before)
        if (rc == SELA_HIP_OK)
            rc = begin(p, f0, std::min(chunk, n_frames - f0), lanes[k & 1]);
        if (rc != SELA_HIP_OK) {
            drain();
            return rc;
        }
        if (!piped) { // one lane: a piece is finished before the next begins
            rc = finish(p);
            if (rc != SELA_HIP_OK)
                return rc;
        }
    }
    for (int i = 0; i < 2; i++) { // in the order they were begun
        const int rc = finish(piece[(k + i) & 1]);
        if (rc != SELA_HIP_OK) {
            drain();
            return rc;
        }
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
