// sela_capi.hip -- the extern "C" boundary of libsela_hip.so (declared in include/sela_hip.h).
//
// Plain pointers and sizes only.  The *_device entry points enqueue kernels on the caller's stream;
// the host-pointer entry points (one-shot and streaming jobs) stage through a per-thread, grow-only set
// of device buffers.  There is no CPU fallback anywhere: without a HIP device every call fails with
// SELA_HIP_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "sela_coalescer.h"
#include "sela_device.h"

namespace sela {
size_t encode_workspace_bytes(uint32_t n_frames, uint32_t channels);
int encode_team_lanes(uint32_t n_frames, uint32_t channels, int forced);
void set_keep_both_candidates(int on);
void set_encode_hashes(int on);
hipError_t launch_encode(const int16_t* d_pcm, uint32_t n_frames, uint32_t channels, uint8_t* d_frames, size_t frames_cap,
    uint64_t* d_frame_offsets, uint32_t* d_status, void* d_workspace, sela_hip_trace* d_trace, hipStream_t stream,
    hipEvent_t* ev, uint64_t* d_phase_cycles, const EncodeHostLink* link, int force_plain_fir, int self_blocks_override, int team_lanes,
    int32_t* d_trace_residues = nullptr, uint32_t priorities = 0, int phase = 0, const uint64_t* plan_base = nullptr, bool plan_accumulate = false);
uint32_t encode_split_frames(uint32_t n_frames, uint32_t channels, int permille);
hipError_t launch_stage_rice_encode(const int32_t* d_values, const uint64_t* d_value_offsets, uint32_t n_streams, uint32_t* d_k, uint32_t* d_word_counts,
    uint32_t* d_words, const uint64_t* d_word_offsets, uint32_t* d_status, hipStream_t stream);
hipError_t launch_stage_lpc_decode(const int32_t* d_order, const int32_t* d_q, const int32_t* d_residues, uint32_t n_blocks, int32_t* d_samples,
    int64_t* d_coefs, uint32_t* d_status, hipStream_t stream);
hipError_t launch_stage_rice_decode(const uint32_t* d_words, const uint64_t* d_word_offsets, const uint32_t* d_k, const uint64_t* d_value_offsets,
    uint32_t n_streams, int32_t* d_values, uint32_t* d_status, hipStream_t stream);
hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, hipStream_t stream, hipEvent_t* ev, uint64_t* d_phase_cycles,
    uint8_t* frame_flags, int recurrence_form, uint32_t synth_priorities = 0);
int decode_waves(uint32_t channels);
// the any-length / 32-bit route (sela_capi_generic.hip)
void generic_release();
void generic_shutdown();
size_t generic_encode_bound_bytes(uint32_t n_frames, uint32_t channels, uint32_t n);
int generic_encode(const void* input, bool in16, uint32_t n_frames, uint32_t channels, uint32_t n, uint8_t* frames_out, size_t frames_cap, uint64_t* frame_offsets_out);
uint32_t generic_index_samples(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, uint64_t* sample_offsets, bool* all_standard);
int generic_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int32_t* samples_out, uint32_t stride,
    uint32_t* counts_out, int16_t* pcm_out, const uint64_t* sample_offsets);
int generic_lpc_encode(const int32_t* samples, uint32_t n_blocks, uint32_t n, int32_t* order_out, int32_t* q_out, int32_t* residues_out);
int generic_lpc_decode(const int32_t* order, const int32_t* q, const int32_t* residues, uint32_t n_blocks, uint32_t n, int32_t* samples_out, int64_t* coefs_out);
size_t decode_workspace_bytes(uint32_t n_frames, uint32_t channels);
uint32_t decode_max_channels();
} // namespace sela

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& what)
{
    g_error = what;
    return code;
}

int fail_hip(hipError_t e, const char* where)
{
    return fail(e == hipErrorOutOfMemory ? SELA_HIP_ENOMEM : SELA_HIP_ENODEV, std::string(where) + ": " + hipGetErrorString(e));
}

} // namespace
namespace sela {
int report_error(int code, const std::string& what) { return fail(code, what); }
int report_hip_error(hipError_t e, const char* where) { return fail_hip(e, where); }
}
namespace {

// ---- page-locked host memory (sela_hip_host_alloc) ---------------------------------------------------------
// Copies from and to pageable memory go through the runtime's own staging and block the calling thread;
// from page-locked memory hipMemcpyAsync really is asynchronous, which is what the chunk pipeline below
// needs.  Pinning is slow (page by page), so freed blocks are kept and handed out again.
struct PinnedPool {
    struct Block {
        void* ptr;
        size_t cap;
        bool pinned;
    };
    std::mutex mu;
    std::vector<Block> live, idle;
    size_t idle_bytes = 0;
    static constexpr size_t kIdleCap = (size_t)1 << 30; // blocks kept for reuse beyond this are unpinned and freed, oldest first

    void* take(size_t bytes)
    {
        std::lock_guard<std::mutex> lock(mu);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i].cap >= bytes && idle[i].cap <= 2 * bytes + 4096 && (best == idle.size() || idle[i].cap < idle[best].cap))
                best = i;
        Block b;
        if (best != idle.size()) {
            b = idle[best];
            idle.erase(idle.begin() + (std::ptrdiff_t)best);
            idle_bytes -= b.cap;
        } else {
            b.cap = (bytes + 4095) & ~(size_t)4095;
            b.ptr = nullptr;
            // (portable + mapped: one worker's kernels read and write these blocks from whichever device it is bound to)
            b.pinned = hipHostMalloc(&b.ptr, b.cap, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && b.ptr;
            if (!b.pinned) { // no device (CPU-only container code still runs): ordinary memory
                (void)hipGetLastError();
                b.ptr = std::aligned_alloc(4096, b.cap);
                if (!b.ptr)
                    return nullptr;
            }
        }
        live.push_back(b);
        return b.ptr;
    }
    void give(void* p)
    {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < live.size(); i++)
            if (live[i].ptr == p) {
                idle.push_back(live[i]);
                idle_bytes += live[i].cap;
                live.erase(live.begin() + (std::ptrdiff_t)i);
                while (idle_bytes > kIdleCap && idle.size() > 1) { // (the block just returned stays: it is the likeliest to be asked for again)
                    const Block old = idle.front();
                    idle.erase(idle.begin());
                    idle_bytes -= old.cap;
                    if (old.pinned)
                        (void)hipHostFree(old.ptr);
                    else
                        std::free(old.ptr);
                }
                return;
            }
    }
    void trim()
    {
        std::lock_guard<std::mutex> lock(mu);
        for (Block& b : idle) {
            if (b.pinned)
                (void)hipHostFree(b.ptr);
            else
                std::free(b.ptr);
        }
        idle.clear();
        idle_bytes = 0;
    }
};
PinnedPool& pool()
{
    static PinnedPool* p = new PinnedPool; // (never destroyed: the HIP runtime may already be gone at exit)
    return *p;
}

// Grow-only device scratch used by the host-pointer API (one set per calling thread).
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};
// page-locked host scratch owned by the library (frame offsets on their way in or out)
struct HostBuffer {
    void* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocPortable | hipHostMallocMapped);
        if (e == hipSuccess) {
            cap = want;
            std::memset(ptr, 0, want); // (tickets and flags the device leaves here are compared with what was there before)
        } else
            ptr = nullptr;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// Host-pointer decodes run as a chunked pipeline (encodes: see job_feed_encode_launch): the copy-in of every chunk is queued, in order
// on one stream, as soon as the chunk is fed (kSets chunks deep); the kernels of consecutive chunks alternate
// between streams (a chunk of this size leaves the device in its launch tail for a good part of its run time, so
// neighbours overlap); a chunk is copied out as soon as its kernels have finished.  Nothing in the issue path
// waits for the device until a buffer set comes round again.
constexpr uint32_t kHostChunkFrames = 1024; // (for up to eight channels, see chunk_frames())
constexpr uint32_t kSets = 8;
constexpr uint32_t kRunStreams = 2;

struct ChunkSet { // one chunk of a decode job in flight
    DeviceBuffer pcm, frames, workspace;
    hipEvent_t copied_in = nullptr, ran = nullptr, copied_out = nullptr;
};

struct HostContext {
    ChunkSet set[kSets];
    DeviceBuffer status;      // device-side status words (decode kernels of a job report through job_flags instead)
    // encode jobs: one launch per feed (see job_feed_encode)
    DeviceBuffer enc_pcm, enc_workspace, enc_ready, enc_words; // enc_words: the job's stream position (two cells in turn), 4 status words, the stagers' four words (EncodeHostLink::stage_started)
    hipEvent_t enc_prev = nullptr; // behind the newest work on the encode stream
    HostBuffer enc_mirror;                                     // page-locked: offsets + status per feed
    uint64_t* enc_mirror_mapped = nullptr;
    HostBuffer job_offsets_host; // decode: the frame offsets of every feed of the running job, where the kernels read them
    HostBuffer job_flags;     // decode: one byte per (frame, wave), zeroed by the host, written by the kernels (over the link) on errors only
    const uint64_t* job_offsets_mapped = nullptr; // device addresses of the two host buffers
    uint8_t* job_flags_mapped = nullptr;
    hipStream_t s_in = nullptr, s_out = nullptr, s_run[kRunStreams] = { nullptr, nullptr };
    int device = -1;
    bool job_open = false;
    int staged_device = -1; // the device whose staging path this context's open job holds (staged_path_acquire), or -1
    // the buffers belong to the device that was current when they were allocated
    bool bind_current_device()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess)
            return false;
        if (dev != device) {
            release();
            device = dev;
        }
        return true;
    }
    // Four streams, because the runtime multiplexes streams onto four hardware queues and two streams on one queue
    // serialise (an event record of one sits in front of the other's kernels).  Encode jobs use them as copy-in,
    // two kernel streams and copy-out; decode jobs as copy-in and three kernel + copy-out streams.
    hipError_t streams()
    {
        hipError_t e = hipSuccess;
        for (hipStream_t* s : { &s_in, &s_run[0], &s_run[1], &s_out })
            if (!*s && (e = hipStreamCreateWithFlags(s, hipStreamNonBlocking)) != hipSuccess)
                return e;
        for (ChunkSet& c : set)
            for (hipEvent_t* ev : { &c.copied_in, &c.ran, &c.copied_out })
                if (!*ev && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess)
                    return e;
        if (!enc_prev)
            e = hipEventCreateWithFlags(&enc_prev, hipEventDisableTiming);
        return e;
    }
    // decode: a chunk's kernel and copy-out share a stream, three streams in turn (the fourth copies in)
    hipStream_t decode_stream(uint32_t chunk) const
    {
        const hipStream_t run[3] = { s_run[0], s_run[1], s_out };
        return run[chunk % 3];
    }
    void sync_all()
    {
        for (hipStream_t s : { s_in, s_out, s_run[0], s_run[1] })
            if (s)
                (void)hipStreamSynchronize(s);
    }
    void release()
    {
        for (ChunkSet& c : set) {
            c.pcm.release();
            c.frames.release();
            c.workspace.release();
            for (hipEvent_t* ev : { &c.copied_in, &c.ran, &c.copied_out }) {
                if (*ev)
                    (void)hipEventDestroy(*ev);
                *ev = nullptr;
            }
        }
        status.release();
        if (enc_prev)
            (void)hipEventDestroy(enc_prev);
        enc_prev = nullptr;
        enc_pcm.release();
        enc_workspace.release();
        enc_ready.release();
        enc_words.release();
        enc_mirror.release();
        job_offsets_host.release();
        job_flags.release();
        for (hipStream_t* s : { &s_in, &s_out, &s_run[0], &s_run[1] }) {
            if (*s)
                (void)hipStreamDestroy(*s);
            *s = nullptr;
        }
    }
};
// A thread's context is LEASED: creating one (four streams, 25 events, a dozen device and page-locked buffers) takes the
// runtime about 10 ms -- three times what a 3-minute track takes file to file -- and host programs start threads per job
// (one worker per GPU and batch, the player's decoder).  A thread that ends, or calls sela_hip_thread_release(), parks its
// context; the next thread that needs one on that device takes it over warm.  sela_hip_shutdown() frees the parked ones.
struct ContextPark {
    std::mutex mu;
    std::vector<HostContext*> idle;
};
ContextPark& park()
{
    static ContextPark* p = new ContextPark; // (never destroyed: threads may end after the statics have)
    return *p;
}
std::atomic<int> g_contexts_created{ 0 }; // debug: sela_hip_debug_contexts_created
constexpr size_t kParkedContexts = 16; // more than this many idle ones are freed instead (a context holds up to a few hundred MB of HBM)

void staged_path_release(int device); // (below)

struct ContextLease {
    HostContext* held = nullptr;
    HostContext& get()
    {
        if (!held) {
            int dev = -1;
            (void)hipGetDevice(&dev);
            ContextPark& p = park();
            {
                std::lock_guard<std::mutex> lock(p.mu);
                for (size_t i = p.idle.size(); i-- > 0 && !held;)
                    if (p.idle[i]->device == dev) {
                        held = p.idle[i];
                        p.idle.erase(p.idle.begin() + (long)i);
                    }
            }
            if (!held) {
                held = new HostContext;
                g_contexts_created.fetch_add(1, std::memory_order_relaxed);
            }
        }
        return *held;
    }
    void give_back()
    {
        if (!held)
            return;
        HostContext* c = held;
        held = nullptr;
        if (c->job_open) { // a thread that ended in the middle of a job: nothing of it may be left in flight or marked
            c->sync_all();
            c->release();
            c->job_open = false;
            if (c->staged_device >= 0) // (its job held the device's staging path: nobody else will give it back)
                staged_path_release(c->staged_device);
            c->staged_device = -1;
        }
        ContextPark& p = park();
        {
            std::lock_guard<std::mutex> lock(p.mu);
            if (c->device >= 0 && p.idle.size() < kParkedContexts) {
                p.idle.push_back(c);
                return;
            }
        }
        c->release();
        delete c;
    }
    ~ContextLease() { give_back(); }
};
thread_local ContextLease g_lease;
inline HostContext& ctx() { return g_lease.get(); }

// Per-thread kernel timing (bench.py's roofline leg): events bracketing the kernels of the last call.
struct KernelTiming {
    bool enabled = false;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    int recorded = 0; // number of kernels bracketed by the last call
    hipEvent_t* events()
    {
        if (!enabled)
            return nullptr;
        for (auto& e : ev)
            if (!e && hipEventCreate(&e) != hipSuccess)
                return nullptr;
        return ev;
    }
};
thread_local KernelTiming g_timing;
thread_local uint64_t* g_phase_cycles = nullptr; // debug: per-block phase cycle counts (sela_hip_debug_phase_buffer)
thread_local int g_force_plain_fir = 0;          // debug: sela_hip_debug_force_plain_fir
thread_local int g_self_blocks = -1;             // debug: sela_hip_debug_mean_workers
thread_local int g_team_lanes = -1;              // debug: sela_hip_debug_encode_teams
thread_local int g_fused_device = 0;             // debug: sela_hip_debug_encode_fused
thread_local int g_stage_wait_naps = -1;         // debug: sela_hip_debug_stage_wait
thread_local int g_reissued_feeds = 0;           // debug: sela_hip_debug_reissued_feeds
thread_local int g_recurrence_form = -1;         // debug: sela_hip_debug_decode_recurrence

// The staging kernel asks for whole CUs and the blocks that wait for it never leave theirs: two jobs staging on one
// device at once (two host threads on one GPU) can keep each other's stagers off the device until the bounded waits run
// out.  So one job per device stages at a time; a job that finds the path taken uses the copy engine instead.
std::mutex g_staged_mu;
bool g_staged_busy[64] = {};
bool staged_path_acquire(int device)
{
    if (device < 0 || device >= 64)
        return false;
    std::lock_guard<std::mutex> lock(g_staged_mu);
    if (g_staged_busy[device])
        return false;
    g_staged_busy[device] = true;
    return true;
}
void staged_path_release(int device)
{
    if (device < 0 || device >= 64)
        return;
    std::lock_guard<std::mutex> lock(g_staged_mu);
    g_staged_busy[device] = false;
}

// Frames per decode chunk: kHostChunkFrames for up to eight channels; beyond that a chunk keeps about the bytes of
// 1024 eight-channel frames (a 255-channel frame is 1 MB of PCM: chunks are sized by what they move, and the chunk
// buffers -- PCM and the generic-mode workspace of every set -- by the chunk).
uint32_t chunk_frames(uint32_t channels)
{
    return channels <= 8 ? kHostChunkFrames : std::max(16u, kHostChunkFrames * 8 / channels);
}
// Size of chunk number `index` of a decode job that has `available` frames at hand.  It opens with two shorter
// chunks: its copy-outs run back to back from the moment the first chunk is done, so the job is as long as the way
// to that moment plus the bare copy of the PCM -- provided every later chunk is decoded by the time the copy-out
// before it ends, which is what keeps the first chunks from being shorter still.
uint32_t next_chunk_frames(uint32_t index, uint32_t available, uint32_t channels)
{
    const uint32_t full = chunk_frames(channels);
    uint32_t want = full;
    if (index < 2)
        want = std::min<uint32_t>(want, std::max(8u, (index ? 640u : 384u) * full / kHostChunkFrames));
    if (want < available && available - want < want / 4) // (no stub of a last chunk: split what is left in two)
        want = (available + 1) / 2;
    return want < available ? want : available;
}

uint32_t flags_to_error(uint32_t flags)
{
    return flags & (SELA_HIP_FLAG_WORDS_CAP | SELA_HIP_FLAG_RICE_RANGE | SELA_HIP_FLAG_COEF_OVERFLOW);
}

} // namespace

// ---- streaming jobs ----------------------------------------------------------------------------------------
struct EncodeFeed {      // one feed of an encode job = one launch
    uint32_t first = 0, n_frames = 0;
    size_t mirror_at = 0;        // where the launch reports (offsets, status) in ctx().enc_mirror
    hipEvent_t done = nullptr;   // recorded behind the launch
    void* bounce_in = nullptr;   // page-locked copy of a feed that came from ordinary memory
    const int16_t* host_src = nullptr;   // the feed's PCM in page-locked host memory, as the host sees it ...
    const int16_t* mapped_src = nullptr; // ... and as the device sees it
    bool reissued = false;       // went through the copy-engine path a second time (see job_reissue_encode)
};

struct sela_hip_job {
    bool encode = false;
    uint32_t channels = 0, total_frames = 0;
    uint32_t fed = 0;      // frames handed to feed() so far
    int error = SELA_HIP_OK;
    // ---- decode: chunks
    uint32_t issued = 0;       // chunks whose copies and kernel are enqueued
    uint32_t final_chunks = 0; // chunks whose results are complete in host memory
    std::vector<uint32_t> chunk_first, chunk_frames; // per issued chunk
    int16_t* pcm_out = nullptr;   // where the copy-outs land: the caller's buffer if it is page-locked, else a page-locked bounce buffer
    int16_t* pcm_caller = nullptr; // the caller's buffer
    bool pcm_bounce = false;
    uint32_t frames_moved = 0;     // frames copied from the bounce buffer to the caller's so far
    std::vector<void*> bounce_frames; // page-locked copies of feeds that came from ordinary memory
    size_t offsets_used = 0; // entries of ctx().job_offsets_host taken by the feeds so far
    // ---- encode: feeds
    uint8_t* frames_out = nullptr; // the caller's buffer
    size_t frames_cap = 0;
    uint64_t* offsets_out = nullptr;
    uint8_t* out_host = nullptr;   // where the kernels' stores land as the host sees it: frames_out, or a page-locked bounce buffer
    uint8_t* out_mapped = nullptr; // ... as the device sees it
    bool out_bounce = false;
    std::vector<EncodeFeed> feeds;
    size_t mirror_used = 0;
    bool staged = false;      // this job's feeds have their PCM fetched by the staging kernel (one such job per device at a time)
    bool holds_staged_path = false; // (taken at begin, given back at end -- also by a job that fell back to the copy engine on the way)
    uint32_t feeds_final = 0; // feeds whose launch has finished: their bytes and offsets are complete in host memory
    uint32_t frames_final = 0;
    uint64_t bytes_final = 0, bytes_copied = 0; // (bytes_copied: bounce buffer -> frames_out)
};

namespace {

int job_fail(sela_hip_job* job, int code)
{
    if (job->error == SELA_HIP_OK)
        job->error = code;
    return code;
}

// Decode into ordinary memory: the chunks that have arrived in the page-locked bounce buffer go on to the caller's.
void job_move_decoded(sela_hip_job* job)
{
    if (!job->pcm_bounce || job->final_chunks == 0)
        return;
    const uint32_t n = job->final_chunks;
    const uint32_t frames = job->chunk_first[n - 1] + job->chunk_frames[n - 1];
    if (frames > job->frames_moved) {
        const size_t frame_samples = (size_t)sela::kBlock * job->channels;
        std::memcpy(job->pcm_caller + job->frames_moved * frame_samples, job->pcm_out + job->frames_moved * frame_samples,
            (size_t)(frames - job->frames_moved) * frame_samples * sizeof(int16_t));
        job->frames_moved = frames;
    }
}

// Results of decode chunk `i` are in host memory once its copy-out has finished.
int job_finalize(sela_hip_job* job, uint32_t upto /* chunks */)
{
    while (job->final_chunks < upto) {
        ChunkSet& c = ctx().set[job->final_chunks % kSets];
        // (chunk i and chunk i + kSets share the event; waiting for the later record covers the earlier one)
        hipError_t e = hipEventSynchronize(c.copied_out);
        if (e != hipSuccess)
            return job_fail(job, fail_hip(e, "copy-out"));
        job->final_chunks++;
    }
    job_move_decoded(job);
    return SELA_HIP_OK;
}

hipError_t reserve_chunk_buffers(uint32_t channels)
{
    const size_t frame_pcm = (size_t)sela::kBlock * channels * sizeof(int16_t);
    const uint32_t frames = chunk_frames(channels);
    for (ChunkSet& c : ctx().set) {
        hipError_t e;
        if ((e = c.pcm.reserve(frames * frame_pcm + 16)) != hipSuccess
            || (e = c.workspace.reserve(sela::decode_workspace_bytes(frames, channels))) != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// The address the device uses for page-locked host memory, or null for ordinary (pageable) memory.
void* device_view(const void* host)
{
    hipPointerAttribute_t attr;
    void* dev = nullptr;
    if (host && hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost
        && hipHostGetDevicePointer(&dev, const_cast<void*>(host), 0) == hipSuccess)
        return dev;
    (void)hipGetLastError(); // (not an error: the caller falls back to a bounce buffer)
    return nullptr;
}

// ---- encode jobs ---------------------------------------------------------------------------------------------------
// One feed = one launch of k_encode_blocks (and, beside it on the copy-in stream, of k_stage_in, whose workgroups fetch
// the feed's PCM from page-locked host memory frame by frame): the blocks encode each frame as it lands, and the last
// block of every group of frames (kGroupFrames) writes the group's finished bytes straight to their place in frames_out
// and their offsets to page-locked memory (finish_group).  No copy engine, no hand-over: the host enqueues one kernel
// per feed and an event behind it; a feed's bytes and offsets are final when its event has fired.  Feeds run one
// after the other on one stream; the job's stream position passes from launch to launch in device memory
// (enc_words).  Buffers that are not page-locked go through page-locked bounce buffers.
constexpr uint32_t kStageWorkgroups = 8;         // of k_stage_in, four waves each with eight 16-byte loads in flight per lane (6: 0.95 ms, 16: 0.90, 24: 0.99)
constexpr uint32_t kEncodeLaunchFrames = 1u << 14; // a feed larger than this is cut (the device buffers -- 27 KB of workspace per stereo frame -- are sized for it)

// Enqueue the launch of feed number `index` (its PCM in page-locked memory, its place in the mirror and its event are
// in `feed`).  staged: the PCM is fetched by the staging kernel beside the launch; otherwise by the copy engine in
// front of it.
hipError_t issue_encode_feed(sela_hip_job* job, EncodeFeed& feed, size_t index, bool staged)
{
    const size_t frame_pcm = (size_t)sela::kBlock * job->channels * sizeof(int16_t);
    const hipStream_t s = ctx().s_run[0];
    hipError_t e;
    sela::EncodeHostLink link;
    link.mirror = ctx().enc_mirror_mapped + feed.mirror_at;
    // (two cells in turn: a launch reads where the one before left the stream and leaves its own end in the other)
    uint64_t* const pos = static_cast<uint64_t*>(ctx().enc_words.ptr);
    link.pos_in = pos + (index & 1);
    link.pos_out = pos + ((index + 1) & 1);
    link.host_pcm = feed.mapped_src;
    link.pcm_ready = static_cast<uint64_t*>(ctx().enc_ready.ptr);
    link.stage_workgroups = kStageWorkgroups;
    link.stage_stream = ctx().s_in;
    link.stage_started = pos + 4;
    link.wait_naps = g_stage_wait_naps;
    // what fills the device's copy of the PCM waits for the launch before this one, which reads it (and for the
    // allocation that may just have cleared the marks)
    if ((e = hipEventRecord(ctx().enc_prev, s)) != hipSuccess || (e = hipStreamWaitEvent(ctx().s_in, ctx().enc_prev, 0)) != hipSuccess)
        return e;
    if (!staged) { // (the stagers copy stereo frames; anything else goes in by the copy engine, ahead of the launch)
        if ((e = hipMemcpyAsync(ctx().enc_pcm.ptr, feed.host_src, feed.n_frames * frame_pcm, hipMemcpyHostToDevice, ctx().s_in)) != hipSuccess
            || (e = hipEventRecord(ctx().enc_prev, ctx().s_in)) != hipSuccess || (e = hipStreamWaitEvent(s, ctx().enc_prev, 0)) != hipSuccess)
            return e;
        link.host_pcm = nullptr;
    }
    uint32_t* d_status = reinterpret_cast<uint32_t*>(pos + 2);
    e = sela::launch_encode(static_cast<const int16_t*>(ctx().enc_pcm.ptr), feed.n_frames, job->channels, job->out_mapped, job->frames_cap, nullptr, d_status,
        ctx().enc_workspace.ptr, nullptr, s, nullptr, nullptr, &link, g_force_plain_fir, g_self_blocks, 0);
    if (e == hipSuccess && !feed.done)
        e = hipEventCreateWithFlags(&feed.done, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventRecord(feed.done, s);
    return e;
}

int job_feed_encode_launch(sela_hip_job* job, const int16_t* pcm, uint32_t nf)
{
    const size_t frame_pcm = (size_t)sela::kBlock * job->channels * sizeof(int16_t);
    const hipStream_t s = ctx().s_run[0];
    hipError_t e;
    EncodeFeed feed;
    feed.first = job->fed;
    feed.n_frames = nf;
    feed.mirror_at = job->mirror_used;
    feed.host_src = pcm;
    feed.mapped_src = static_cast<const int16_t*>(device_view(pcm));
    if (!feed.mapped_src) { // ordinary memory: through a page-locked copy (kept until the job ends)
        feed.bounce_in = pool().take(nf * frame_pcm);
        if (!feed.bounce_in)
            return job_fail(job, fail(SELA_HIP_ENOMEM, "page-locked bounce buffer"));
        std::memcpy(feed.bounce_in, pcm, nf * frame_pcm);
        feed.host_src = static_cast<const int16_t*>(feed.bounce_in);
        if (!(feed.mapped_src = static_cast<const int16_t*>(device_view(feed.bounce_in)))) {
            pool().give(feed.bounce_in);
            return job_fail(job, fail(SELA_HIP_ENODEV, "page-locked memory is not visible to the device"));
        }
    }
    // (grow-only; the launches of a job run in order on one stream, so growing waits for the last one)
    const size_t need_pcm = nf * frame_pcm + 16, need_ws = sela::encode_workspace_bytes(nf, job->channels), need_ready = (size_t)nf * 8;
    if (need_pcm > ctx().enc_pcm.cap || need_ws > ctx().enc_workspace.cap || need_ready > ctx().enc_ready.cap) {
        if ((e = hipStreamSynchronize(s)) != hipSuccess || (e = ctx().enc_pcm.reserve(need_pcm)) != hipSuccess
            || (e = ctx().enc_workspace.reserve(need_ws)) != hipSuccess || (e = ctx().enc_ready.reserve(need_ready)) != hipSuccess
            // (the "frame copied in" words carry the bare ticket -- and a checksum -- and another process's tickets count from the same start)
            || (e = hipMemsetAsync(ctx().enc_ready.ptr, 0, ctx().enc_ready.cap, s)) != hipSuccess) {
            if (feed.bounce_in)
                pool().give(feed.bounce_in);
            return job_fail(job, fail_hip(e, "hipMalloc"));
        }
    }
    e = issue_encode_feed(job, feed, job->feeds.size(), job->staged && job->channels == 2);
    if (e != hipSuccess) {
        if (feed.bounce_in)
            pool().give(feed.bounce_in);
        if (feed.done)
            (void)hipEventDestroy(feed.done);
        return job_fail(job, fail_hip(e, "encode launch"));
    }
    job->mirror_used += (size_t)nf + 2;
    job->fed += nf;
    job->feeds.push_back(feed);
    return SELA_HIP_OK;
}

// A launch whose bounded wait for the staging kernel ran out (SELA_HIP_FLAG_INTERNAL: the device was kept so busy by
// other work that the stagers got no compute units in time) has coded frames that were not there: its bytes, its
// sizes -- and with them the stream position every later feed started from -- are void.  Feed `from` and everything
// queued behind it are issued again, PCM by the copy engine this time (nothing waits for a kernel beside it there).
// A busy device costs time, never the result; only a feed that fails this way too is an error.
int job_reissue_encode(sela_hip_job* job, size_t from)
{
    hipError_t e;
    ctx().sync_all(); // (nothing of the void launches is left running)
    job->staged = false;
    uint64_t* const pos = static_cast<uint64_t*>(ctx().enc_words.ptr);
    const uint64_t start = job->bytes_final; // the stream behind the last good feed
    if ((e = hipMemcpyAsync(pos + (from & 1), &start, 8, hipMemcpyHostToDevice, ctx().s_run[0])) != hipSuccess
        || (e = hipStreamSynchronize(ctx().s_run[0])) != hipSuccess) // (`start` is a local)
        return job_fail(job, fail_hip(e, "encode re-issue"));
    for (size_t k = from; k < job->feeds.size(); k++) {
        job->feeds[k].reissued = true;
        g_reissued_feeds++;
        if ((e = issue_encode_feed(job, job->feeds[k], k, false)) != hipSuccess)
            return job_fail(job, fail_hip(e, "encode re-issue"));
    }
    return SELA_HIP_OK;
}

int job_feed_encode(sela_hip_job* job, const int16_t* pcm, uint32_t n_frames)
{
    const size_t frame_samples = (size_t)sela::kBlock * job->channels;
    for (uint32_t done = 0; done < n_frames;) {
        const uint32_t nf = std::min(kEncodeLaunchFrames, n_frames - done);
        const int rc = job_feed_encode_launch(job, pcm + done * frame_samples, nf);
        if (rc != SELA_HIP_OK)
            return rc;
        done += nf;
    }
    return SELA_HIP_OK;
}

// How far the stream has got in host memory: the feeds, in order, whose launch has finished.  Fills offsets_out for
// their frames, and moves their bytes out of the bounce buffer if there is one.  (Finer steps -- a word per group of
// frames written by the kernel behind the group's bytes -- were dropped with the release fence they need: it costs an
// L2 write-back per group.)
int job_encode_progress(sela_hip_job* job, bool wait)
{
    const uint64_t* mirror = static_cast<const uint64_t*>(ctx().enc_mirror.ptr);
    while (job->feeds_final < job->feeds.size()) {
        EncodeFeed& f = job->feeds[job->feeds_final];
        const hipError_t q = wait ? hipEventSynchronize(f.done) : hipEventQuery(f.done);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            break;
        }
        if (q != hipSuccess)
            return job_fail(job, fail_hip(q, "encode kernels"));
        const uint64_t* m = mirror + f.mirror_at;
        const uint64_t st = m[f.n_frames + 1];
        const uint32_t flags = (uint32_t)st, overflow = (uint32_t)(st >> 32);
        if (flags & SELA_HIP_FLAG_INTERNAL) {
            if (f.reissued)
                return job_fail(job, fail(SELA_HIP_ENODEV, "a wait inside the encode kernel ran out (internal error)"));
            const int rc = job_reissue_encode(job, job->feeds_final);
            if (rc != SELA_HIP_OK)
                return rc;
            continue; // (this feed again, now behind its new event)
        }
        if (flags_to_error(flags)) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "a block left the range the .sela format can carry (flags 0x%x)", flags);
            return job_fail(job, fail(SELA_HIP_ERANGE, msg));
        }
        for (uint32_t i = 0; i <= f.n_frames; i++)
            job->offsets_out[(size_t)f.first + i] = m[i];
        uint32_t fits = f.n_frames; // (a frame that ends beyond the capacity was not written, nor was any frame behind it)
        while (fits > 0 && m[fits] > job->frames_cap)
            fits--;
        if (m[fits] <= job->frames_cap) {
            job->frames_final = f.first + fits;
            job->bytes_final = m[fits];
        }
        if (job->out_bounce && job->bytes_final > job->bytes_copied) {
            std::memcpy(job->frames_out + job->bytes_copied, job->out_host + job->bytes_copied, (size_t)(job->bytes_final - job->bytes_copied));
            job->bytes_copied = job->bytes_final;
        }
        if (overflow)
            return job_fail(job, fail(SELA_HIP_ECAPACITY, "frames_out too small (see sela_hip_encode_bound_bytes)"));
        if (f.bounce_in) { // (a final feed is never issued again: its page-locked copy can go)
            pool().give(f.bounce_in);
            f.bounce_in = nullptr;
            f.host_src = f.mapped_src = nullptr;
        }
        job->feeds_final++;
    }
    return SELA_HIP_OK;
}

// A decode chunk needs nothing from the host once it is queued (its output size is known): copy-in on the copy
// stream; then, on one of three streams in turn, the kernel and right behind it -- no event in between -- the
// copy-out, so that the copy-outs of neighbouring chunks come from different streams (on ONE stream the copy engine
// idles ~20 us between two copies).  The kernels read the frame offsets where the host staged them (page-locked
// memory, two words per workgroup over the link): the first kernel starts as soon as the first chunk's frames are on
// the device.  (Measured and dropped: copy-in on the chunk's own stream as well -- the runtime keeps about three
// copies in flight and starts them in the order they were queued, so a copy-out that waits for its kernel holds up
// the copy-ins queued behind it: 0.90 -> 1.11 ms.)
// `k_offsets` = the chunk's first entry of the feed's staged offsets (absolute in `frames`), as the device sees it.
int job_issue_decode(sela_hip_job* job, const uint8_t* frames, const uint64_t* offsets, const uint64_t* k_offsets, uint32_t nf)
{
    const uint32_t i = job->issued;
    ChunkSet& c = ctx().set[i % kSets];
    const hipStream_t s = ctx().decode_stream(i);
    hipError_t e;
    if (i >= kSets) {
        // chunk i - kSets used this set.  The host goes no further ahead than that chunk's kernel (which also keeps
        // the queues short); the copy-in must not overwrite frames that kernel reads, the kernel not PCM on its way out.
        if ((e = hipEventSynchronize(c.ran)) != hipSuccess)
            return job_fail(job, fail_hip(e, "kernels"));
        if ((e = hipStreamWaitEvent(s, c.copied_out, 0)) != hipSuccess)
            return job_fail(job, fail_hip(e, "hipStreamWaitEvent"));
    }
    const uint64_t bytes = offsets[nf] - offsets[0];
    if ((size_t)bytes + 16 > c.frames.cap) {
        if (i >= kSets && (e = hipEventSynchronize(c.copied_out)) != hipSuccess) // (nothing of the set's last use is in flight)
            return job_fail(job, fail_hip(e, "copy-out"));
        if ((e = c.frames.reserve((size_t)bytes + 16)) != hipSuccess)
            return job_fail(job, fail_hip(e, "hipMalloc"));
    }
    if ((bytes && (e = hipMemcpyAsync(c.frames.ptr, frames + offsets[0], (size_t)bytes, hipMemcpyHostToDevice, ctx().s_in)) != hipSuccess)
        || (e = hipEventRecord(c.copied_in, ctx().s_in)) != hipSuccess || (e = hipStreamWaitEvent(s, c.copied_in, 0)) != hipSuccess)
        return job_fail(job, fail_hip(e, "H2D frames"));
    // the kernel adds the feed's absolute offsets to its base: bias the base so that offsets[0] lands on the chunk's copy
    const uint8_t* d_base = static_cast<const uint8_t*>(c.frames.ptr) - offsets[0];
    uint8_t* flags = ctx().job_flags_mapped + (size_t)job->fed * sela::decode_waves(job->channels);
    e = sela::launch_decode(d_base, k_offsets, nf, job->channels, static_cast<int16_t*>(c.pcm.ptr), static_cast<uint32_t*>(ctx().status.ptr),
        c.workspace.ptr, s, nullptr, nullptr, flags, g_recurrence_form);
    if (e != hipSuccess || (e = hipEventRecord(c.ran, s)) != hipSuccess)
        return job_fail(job, fail_hip(e, "decode launch"));
    const size_t frame_pcm = (size_t)sela::kBlock * job->channels * sizeof(int16_t);
    if ((e = hipMemcpyAsync(job->pcm_out + (size_t)job->fed * sela::kBlock * job->channels, c.pcm.ptr, nf * frame_pcm, hipMemcpyDeviceToHost, s)) != hipSuccess
        || (e = hipEventRecord(c.copied_out, s)) != hipSuccess)
        return job_fail(job, fail_hip(e, "D2H pcm"));
    job->chunk_first.push_back(job->fed);
    job->chunk_frames.push_back(nf);
    job->issued++;
    job->fed += nf;
    return SELA_HIP_OK;
}

// One feed of a decode job: stage its offsets, queue its chunks.
int job_feed_decode(sela_hip_job* job, const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames)
{
    if (n_frames == 0)
        return SELA_HIP_OK;
    // (sized at begin for the whole job: total_frames entries + one more per feed, and a feed has at least a frame)
    uint64_t* staged = static_cast<uint64_t*>(ctx().job_offsets_host.ptr) + job->offsets_used;
    const uint64_t* mapped = ctx().job_offsets_mapped + job->offsets_used;
    std::memcpy(staged, offsets, ((size_t)n_frames + 1) * 8);
    job->offsets_used += (size_t)n_frames + 1;
    if (!device_view(frames + offsets[0])) {
        // ordinary memory: through a page-locked copy (kept until the job ends).  Copies from pageable memory go through
        // the runtime's own staging, and with several host threads decoding at once their results were seen to arrive
        // behind the event that should cover them (tests/test_gpu_round3.py::test_four_threads_encode_on_one_gpu).
        const size_t bytes = (size_t)(offsets[n_frames] - offsets[0]);
        void* copy = pool().take(bytes ? bytes : 1);
        if (!copy)
            return job_fail(job, fail(SELA_HIP_ENOMEM, "page-locked bounce buffer"));
        std::memcpy(copy, frames + offsets[0], bytes);
        job->bounce_frames.push_back(copy);
        frames = static_cast<const uint8_t*>(copy) - offsets[0];
    }
    for (uint32_t done = 0; done < n_frames;) {
        const uint32_t nf = next_chunk_frames(job->issued, n_frames - done, job->channels);
        const int rc = job_issue_decode(job, frames, offsets + done, mapped + done, nf);
        if (rc != SELA_HIP_OK)
            return rc;
        done += nf;
    }
    return SELA_HIP_OK;
}

// Decode: which chunks have arrived in host memory by now (no waiting).
int job_drain_ready(sela_hip_job* job)
{
    if (job->encode)
        return job_encode_progress(job, false);
    while (job->final_chunks < job->issued && hipEventQuery(ctx().set[job->final_chunks % kSets].copied_out) == hipSuccess)
        job->final_chunks++;
    (void)hipGetLastError(); // (hipErrorNotReady from the query is not an error)
    job_move_decoded(job);
    return SELA_HIP_OK;
}

void job_progress(const sela_hip_job* job, uint32_t* frames_final, uint64_t* bytes_final)
{
    if (job->encode) {
        if (frames_final)
            *frames_final = job->frames_final;
        if (bytes_final)
            *bytes_final = job->bytes_final;
        return;
    }
    const uint32_t n = job->final_chunks;
    if (frames_final)
        *frames_final = n ? job->chunk_first[n - 1] + job->chunk_frames[n - 1] : 0;
    if (bytes_final)
        *bytes_final = 0;
}

int job_begin(sela_hip_job** out, bool encode, uint32_t channels, uint32_t total_frames)
{
    if (!out)
        return fail(SELA_HIP_EINVAL, "null job pointer");
    *out = nullptr;
    if (channels == 0 || channels > 255)
        return fail(SELA_HIP_EINVAL, "channels must be in 1..255");
    if (!encode && channels > sela::decode_max_channels())
        return fail(SELA_HIP_EINVAL, "too many channels for the on-chip decoder (sela_hip_decode_max_channels)");
    int rc = sela_hip_init(-1);
    if (rc != SELA_HIP_OK)
        return rc;
    if (!ctx().bind_current_device())
        return fail(SELA_HIP_ENODEV, "hipGetDevice failed");
    if (ctx().job_open)
        return fail(SELA_HIP_EINVAL, "this thread already has an open job");
    hipError_t e = ctx().streams();
    if (e == hipSuccess)
        e = ctx().status.reserve(16);
    if (e == hipSuccess && encode) {
        // (offsets + status per feed, a feed has at least a frame)
        if ((e = ctx().enc_mirror.reserve((3 * (size_t)total_frames + 4) * 8)) == hipSuccess
            && (e = ctx().enc_words.reserve(16 + 16 + 32)) == hipSuccess
            && (e = hipHostGetDevicePointer((void**)&ctx().enc_mirror_mapped, ctx().enc_mirror.ptr, 0)) == hipSuccess)
            e = hipMemsetAsync(ctx().enc_words.ptr, 0, 16 + 16 + 32, ctx().s_run[0]); // the job's stream position: 0
    }
    if (e == hipSuccess && !encode) {
        const size_t n_flags = (size_t)total_frames * sela::decode_waves(channels) + 1;
        if ((e = reserve_chunk_buffers(channels)) == hipSuccess
            && (e = ctx().job_offsets_host.reserve((2 * (size_t)total_frames + 2) * 8)) == hipSuccess
            && (e = ctx().job_flags.reserve(n_flags)) == hipSuccess
            && (e = hipHostGetDevicePointer((void**)&ctx().job_offsets_mapped, ctx().job_offsets_host.ptr, 0)) == hipSuccess
            && (e = hipHostGetDevicePointer((void**)&ctx().job_flags_mapped, ctx().job_flags.ptr, 0)) == hipSuccess)
            std::memset(ctx().job_flags.ptr, 0, n_flags);
    }
    if (e != hipSuccess)
        return fail_hip(e, "hipMalloc");
    sela_hip_job* job = new sela_hip_job;
    job->encode = encode;
    job->channels = channels;
    job->total_frames = total_frames;
    job->staged = job->holds_staged_path = encode && channels == 2 && staged_path_acquire(ctx().device);
    ctx().staged_device = job->holds_staged_path ? ctx().device : -1;
    ctx().job_open = true;
    *out = job;
    return SELA_HIP_OK;
}

int job_end(sela_hip_job* job, uint32_t* frames_final, uint64_t* bytes_final)
{
    int rc = job->error;
    hipError_t e = hipSuccess;
    if (job->encode) {
        // (also after an error: nothing of the job may still be running when its buffers go back)
        if ((e = hipStreamSynchronize(ctx().s_run[0])) != hipSuccess && rc == SELA_HIP_OK)
            rc = fail_hip(e, "encode kernels");
        if (rc == SELA_HIP_OK)
            rc = job_encode_progress(job, true);
        for (EncodeFeed& f : job->feeds) {
            if (f.bounce_in)
                pool().give(f.bounce_in);
            if (f.done)
                (void)hipEventDestroy(f.done);
        }
        if (job->out_bounce && job->out_host)
            pool().give(job->out_host);
    } else {
        if (rc == SELA_HIP_OK)
            rc = job_finalize(job, job->issued);
        if (rc != SELA_HIP_OK)
            ctx().sync_all(); // (nothing may still be reading or writing the bounce buffers)
        for (void* b : job->bounce_frames)
            pool().give(b);
        if (job->pcm_bounce && job->pcm_out)
            pool().give(job->pcm_out);
    }
    uint32_t seen_flags = 0;
    if (rc == SELA_HIP_OK && !job->encode && job->issued) {
        // every frame is decoded (bad ones to silence) before the verdict; the kernels left their flags in host memory
        const uint8_t* flags = static_cast<const uint8_t*>(ctx().job_flags.ptr);
        const size_t n = (size_t)job->fed * sela::decode_waves(job->channels);
        for (size_t i = 0; i < n; i++)
            seen_flags |= flags[i];
    }
    if (rc != SELA_HIP_OK) { // leave nothing in flight behind an error
        const std::string msg = sela_hip_last_error();
        ctx().sync_all();
        (void)fail(rc, msg);
    }
    job_progress(job, frames_final, bytes_final);
    ctx().job_open = false;
    if (job->holds_staged_path)
        staged_path_release(ctx().device);
    ctx().staged_device = -1;
    delete job;
    if (rc != SELA_HIP_OK)
        return rc;
    if (seen_flags & SELA_HIP_FLAG_BAD_FRAME)
        return fail(SELA_HIP_EFORMAT, "malformed frame stream (bad sync word or subframe header)");
    if (seen_flags & SELA_HIP_FLAG_RICE_OVERRUN)
        return fail(SELA_HIP_EFORMAT, "a Rice stream ended before all its values were read");
    // (the same policy as the any-length route, sela_capi_generic.hip: what the reference leaves undefined is reported)
    if (seen_flags & SELA_HIP_FLAG_COEF_OVERFLOW)
        return fail(SELA_HIP_ERANGE, "decode: a predictor coefficient left the int64 range");
    if (seen_flags & SELA_HIP_FLAG_Q_RANGE)
        return fail(SELA_HIP_ERANGE, "decode: a quantised reflection coefficient outside [-64, 63] (the reference indexes past its tables, src/lpc/linear_predictor.cpp:23-26)");
    return SELA_HIP_OK;
}


// ---- does a device-pointer launch have the device to itself? ------------------------------------------------------------------
// The encode kernels take a schedule of wave priorities that FALLS with a wave's progress (sela_encode.hip, the note on
// priorities): on its own a launch of one fill finishes 7 % sooner with it, beside another stream's kernels the same schedule
// starves the neighbour and costs 6 %.  Whether there is a neighbour the library can tell for its own launches: every
// device-pointer call leaves an event on its stream, and a call looks at the events of the OTHER streams -- one still pending
// means that stream's kernels will run beside this launch's.  (Decided when the launch is queued; work the library does not
// know of is not seen: then the schedule is merely not the best one.  Events, not the streams themselves, are kept: a caller
// may destroy its stream.)
struct Flights {
    struct Entry {
        hipStream_t stream;
        hipEvent_t done;
    };
    std::mutex mu;
    std::vector<Entry> entries[64]; // by device
    bool others_pending(int dev, hipStream_t stream)
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry& e : entries[dev])
            if (e.stream != stream && hipEventQuery(e.done) == hipErrorNotReady)
                return true;
        return false;
    }
    void note(int dev, hipStream_t stream)
    {
        std::lock_guard<std::mutex> lock(mu);
        std::vector<Entry>& v = entries[dev];
        Entry* mine = nullptr;
        for (Entry& e : v)
            if (e.stream == stream)
                mine = &e;
        if (!mine && v.size() >= 16) // streams come and go: an entry whose work is done is as good as a new one
            for (Entry& e : v)
                if (hipEventQuery(e.done) != hipErrorNotReady) {
                    mine = &e;
                    mine->stream = stream;
                    break;
                }
        if (!mine) {
            if (v.size() >= 64)
                return; // (not tracked: at worst a launch is taken to be alone)
            Entry e{ stream, nullptr };
            if (hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess)
                return;
            v.push_back(e);
            mine = &v.back();
        }
        (void)hipEventRecord(mine->done, stream);
    }
    void release_all() // sela_hip_shutdown: the events go back to the runtime, each on its device
    {
        std::lock_guard<std::mutex> lock(mu);
        int before = -1;
        (void)hipGetDevice(&before);
        for (int dev = 0; dev < 64; dev++) {
            if (entries[dev].empty())
                continue;
            (void)hipSetDevice(dev);
            for (Entry& e : entries[dev])
                (void)hipEventDestroy(e.done);
            entries[dev].clear();
        }
        if (before >= 0)
            (void)hipSetDevice(before);
    }
};
// A stream that is being captured into a graph must not be touched by the bookkeeping: the event record would be captured too,
// and a query of that event afterwards fails.  Such a launch counts as having neighbours (no schedule) and leaves no mark.
bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
Flights& flights()
{
    static Flights* f = new Flights; // (never destroyed: events outlive the statics' teardown)
    return *f;
}
std::atomic<int64_t> g_forced_priorities{ -1 }; // debug (sela_hip_debug_priorities): a fixed schedule for every launch; -1: by the neighbours
std::atomic<int> g_launches_alone{ 0 };         // debug: launches that were given the falling schedule
constexpr uint32_t kFallingPriorities = 0x00010203u; // 3, 2, 1, 0 by quarters of a wave's work (least significant byte first)
constexpr uint32_t kHeavySubframesFirst = 0x00010000u; // the decoder: orders above 60 at priority 1 through their synthesis (launch_decode)

uint32_t launch_priorities(hipStream_t stream, int& dev)
{
    dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        dev = -1;
        return 0;
    }
    const int64_t forced = g_forced_priorities.load(std::memory_order_relaxed);
    if (forced >= 0)
        return (uint32_t)forced;
    if (stream_is_capturing(stream) || flights().others_pending(dev, stream))
        return 0;
    g_launches_alone.fetch_add(1, std::memory_order_relaxed);
    return kFallingPriorities;
}

// ---- an encode launch cut in two (round 6: built, measured, OFF) ------------------------------------------------------------------
// One fill of team waves ends in a tail: the last waves walk alone while most SIMDs idle, and plan + assemble -- 30 us during which
// one CU works -- wait behind it.  Round 5 measured from outside that two encoders on two streams, each on its share of the track,
// finish 6 % sooner than one launch (tools/split_probe.py: 0.434 -> 0.408 ms per call).  Round 6 built it inside the call -- a launch
// the library gives to teams of 16 runs as two halves on the caller's stream and a side stream of the device, the first half's
// plan + assemble under the second half's tail, the second half's frames placed behind the first's by its plan kernel
// (k_plan_frames' base); same bytes, offsets and status words (tests) -- and it is SLOWER: one lane 12.16-12.20 G samples/s at every
// share from 40 to 70 % against 12.48 uncut (profiles/r06/split_sweep.txt): the two event hand-overs between the streams cost more
// than the 15 us of plan + assemble they hide.  So the library never cuts by itself; the form stays behind
// sela_hip_debug_encode_split (bench.py --encode-split N) for the comparison.  Never while kernel timing is on, with a trace or phase
// buffer, with a kernel forced by a debug hook, or into a stream that is being captured.
struct Splitter {
    std::mutex mu;
    hipStream_t side[64] = {};
    hipEvent_t begun[64] = {}, half_done[64] = {};
    std::map<const void*, std::pair<uint32_t, uint32_t>> last; // workspace -> (frames of the launch, frames of its first half; 0: whole)
    bool ready(int dev)
    {
        if (side[dev])
            return true;
        if (hipStreamCreateWithFlags(&side[dev], hipStreamNonBlocking) != hipSuccess) {
            side[dev] = nullptr;
            return false;
        }
        if (hipEventCreateWithFlags(&begun[dev], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&half_done[dev], hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(side[dev]);
            side[dev] = nullptr;
            return false;
        }
        return true;
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lock(mu);
        int before = -1;
        (void)hipGetDevice(&before);
        for (int dev = 0; dev < 64; dev++)
            if (side[dev]) {
                (void)hipSetDevice(dev);
                (void)hipStreamSynchronize(side[dev]);
                (void)hipEventDestroy(begun[dev]);
                (void)hipEventDestroy(half_done[dev]);
                (void)hipStreamDestroy(side[dev]);
                side[dev] = nullptr;
            }
        last.clear();
        if (before >= 0)
            (void)hipSetDevice(before);
    }
};
Splitter& splitter()
{
    static Splitter* s = new Splitter; // (never destroyed: streams outlive the statics' teardown)
    return *s;
}
std::atomic<int> g_encode_split{ 0 };    // debug (sela_hip_debug_encode_split): 0 never (the product), 1..999: every launch of teams of 16, that share to the first half
std::atomic<int> g_launches_split{ 0 };
size_t second_half_offset(uint32_t first, uint32_t channels) { return (sela::encode_workspace_bytes(first, channels) + 255) & ~(size_t)255; }
uint32_t split_of_last_launch(const void* d_workspace, uint32_t n_frames)
{
    std::lock_guard<std::mutex> lock(splitter().mu);
    const auto it = splitter().last.find(d_workspace);
    return it != splitter().last.end() && it->second.first == n_frames ? it->second.second : 0u;
}

} // namespace

// ---- the stages on their own (sela_hip.h): plain synchronous calls, device buffers of their own -----------------------------
namespace {

// device memory for the length of one call
struct Scratch {
    std::vector<void*> blocks;
    ~Scratch()
    {
        for (void* b : blocks)
            (void)hipFree(b);
    }
    template <typename T>
    T* take(size_t count, hipError_t& err)
    {
        void* p = nullptr;
        if (err == hipSuccess)
            err = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (p)
            blocks.push_back(p);
        return static_cast<T*>(p);
    }
    template <typename T>
    T* upload(const T* host, size_t count, hipError_t& err)
    {
        T* d = take<T>(count, err);
        if (err == hipSuccess && count)
            err = hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice);
        return d;
    }
};

int stage_device_ready()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SELA_HIP_ENODEV, "no HIP device visible (the SELA MI355X path has no CPU fallback)");
    return SELA_HIP_OK;
}

} // namespace

extern "C" {

const char* sela_hip_last_error(void) { return g_error.c_str(); }

int sela_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int sela_hip_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SELA_HIP_ENODEV, "no HIP device visible (the SELA MI355X path has no CPU fallback)");
    if (device >= n)
        return fail(SELA_HIP_EINVAL, "device index out of range");
    if (device >= 0) {
        e = hipSetDevice(device);
        if (e != hipSuccess)
            return fail_hip(e, "hipSetDevice");
    }
    return SELA_HIP_OK;
}

void sela_hip_thread_release(void)
{
    g_lease.give_back();
    sela::generic_release();
}

void sela_hip_shutdown(void)
{
    g_lease.give_back();
    sela::generic_shutdown();
    std::vector<HostContext*> idle;
    {
        std::lock_guard<std::mutex> lock(park().mu);
        idle.swap(park().idle);
    }
    int before = -1;
    (void)hipGetDevice(&before);
    for (HostContext* c : idle) { // (streams and buffers are freed on the device they belong to)
        if (c->device >= 0)
            (void)hipSetDevice(c->device);
        c->release();
        delete c;
    }
    if (before >= 0)
        (void)hipSetDevice(before);
    flights().release_all();
    splitter().release_all();
    pool().trim();
}

void* sela_hip_host_alloc(size_t bytes) { return pool().take(bytes ? bytes : 1); }

void sela_hip_host_free(void* p)
{
    if (p)
        pool().give(p);
}

void sela_hip_debug_phase_buffer(uint64_t* d_cycles) { g_phase_cycles = d_cycles; }

void sela_hip_debug_force_plain_fir(int enable) { g_force_plain_fir = enable & 3; }

void sela_hip_debug_mean_workers(int self_blocks) { g_self_blocks = self_blocks; }
void sela_hip_debug_encode_teams(int lanes) { g_team_lanes = lanes; }
void sela_hip_debug_encode_fused(int enable) { g_fused_device = enable != 0; }
void sela_hip_debug_priorities(uint32_t team_quarters) { g_forced_priorities.store((int64_t)team_quarters, std::memory_order_relaxed); }
void sela_hip_debug_priorities_adaptive(void) { g_forced_priorities.store(-1, std::memory_order_relaxed); }
int sela_hip_debug_launches_alone(void) { return g_launches_alone.load(std::memory_order_relaxed); }
int sela_hip_debug_block_forms(const void* d_workspace, uint32_t n_frames, uint32_t channels, uint32_t* counts_out, uint8_t* forms_out)
{
    if (!d_workspace || !counts_out || channels == 0)
        return fail(SELA_HIP_EINVAL, "bad argument");
    counts_out[0] = counts_out[1] = counts_out[2] = 0;
    const size_t blocks = (size_t)n_frames * sela_hip_signals_per_frame(channels);
    std::vector<sela::BlockMeta> meta(blocks);
    hipError_t e = hipDeviceSynchronize();
    // (the records open the workspace, launch_encode -- or, after a launch that was cut in two, each half's workspace)
    const uint32_t first = split_of_last_launch(d_workspace, n_frames);
    const size_t n_sig = sela_hip_signals_per_frame(channels);
    if (e == hipSuccess && blocks) {
        const size_t head = first ? (size_t)first * n_sig : blocks;
        e = hipMemcpy(meta.data(), reinterpret_cast<const void*>(((uintptr_t)d_workspace + 255) & ~(uintptr_t)255), head * sizeof(sela::BlockMeta), hipMemcpyDeviceToHost);
        if (e == hipSuccess && first) {
            const uint8_t* second = static_cast<const uint8_t*>(d_workspace) + second_half_offset(first, channels);
            e = hipMemcpy(meta.data() + head, reinterpret_cast<const void*>(((uintptr_t)second + 255) & ~(uintptr_t)255), (blocks - head) * sizeof(sela::BlockMeta), hipMemcpyDeviceToHost);
        }
    }
    if (e != hipSuccess)
        return fail_hip(e, "block forms");
    for (size_t b = 0; b < blocks; b++) {
        const uint32_t form = (meta[b].flags & sela::kBlockFormPlain) ? 2 : ((meta[b].flags & sela::kBlockFormTwoPass) ? 1 : 0);
        counts_out[form]++;
        if (forms_out)
            forms_out[b] = (uint8_t)form;
    }
    return SELA_HIP_OK;
}
void sela_hip_debug_keep_both_candidates(int on) { sela::set_keep_both_candidates(on); }
void sela_hip_debug_encode_split(int mode) { g_encode_split.store(mode < 0 || mode > 999 ? 0 : mode, std::memory_order_relaxed); }
int sela_hip_debug_launches_split(void) { return g_launches_split.load(std::memory_order_relaxed); }
void sela_hip_debug_encode_hashes(int on) { sela::set_encode_hashes(on); }
int sela_hip_debug_encode_kernel(uint32_t n_frames, uint32_t channels) { return sela::encode_team_lanes(n_frames, channels, g_team_lanes); }

void sela_hip_debug_stage_wait(int naps) { g_stage_wait_naps = naps; }

int sela_hip_debug_reissued_feeds(void) { return g_reissued_feeds; }
void sela_hip_debug_decode_recurrence(int form) { g_recurrence_form = form < 0 ? -1 : (form != 0); }
int sela_hip_debug_contexts_created(void) { return g_contexts_created.load(std::memory_order_relaxed); }

void sela_hip_enable_kernel_timing(int enable) { g_timing.enabled = enable != 0; }

int sela_hip_kernel_times(float* ms_out, int capacity)
{
    const int n = g_timing.recorded < capacity ? g_timing.recorded : capacity;
    for (int i = 0; i < n; i++) {
        if (hipEventSynchronize(g_timing.ev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms_out[i], g_timing.ev[i], g_timing.ev[i + 1]) != hipSuccess)
            return 0;
    }
    return n;
}

uint32_t sela_hip_signals_per_frame(uint32_t channels) { return channels == 2 ? 3u : channels; }

size_t sela_hip_encode_workspace_bytes(uint32_t n_frames, uint32_t channels)
{
    // (room for the two workspaces of a launch that the debug hook cuts in two, at any share: the fixed parts -- the rings -- twice)
    const size_t whole = sela::encode_workspace_bytes(n_frames, channels);
    return sela::encode_split_frames(n_frames, channels, 500) ? whole + sela::encode_workspace_bytes(0, channels) + 65536 : whole;
}

size_t sela_hip_decode_workspace_bytes(uint32_t n_frames, uint32_t channels) { return sela::decode_workspace_bytes(n_frames, channels); }

uint32_t sela_hip_decode_max_channels(void) { return sela::decode_max_channels(); }

size_t sela_hip_encode_bound_bytes(uint32_t n_frames, uint32_t channels)
{
    return (size_t)n_frames * sela_frame_bytes(channels, channels * (uint32_t)sela::kSlotWords);
}

int sela_hip_encode_device(const int16_t* d_pcm, uint32_t n_frames, uint32_t channels, uint8_t* d_frames, size_t frames_cap,
    uint64_t* d_frame_offsets, uint32_t* d_status, void* d_workspace, size_t workspace_bytes, sela_hip_trace* d_trace, void* stream)
{
    if (channels == 0 || channels > 255)
        return fail(SELA_HIP_EINVAL, "channels must be in 1..255");
    if (!d_frame_offsets || !d_status || (n_frames && (!d_pcm || !d_frames || !d_workspace)))
        return fail(SELA_HIP_EINVAL, "null device pointer");
    if (((uintptr_t)d_pcm & 3) || ((uintptr_t)d_frames & 3))
        return fail(SELA_HIP_EINVAL, "d_pcm and d_frames must be 4-byte aligned");
    if (workspace_bytes < sela::encode_workspace_bytes(n_frames, channels))
        return fail(SELA_HIP_ECAPACITY, "workspace smaller than sela_hip_encode_workspace_bytes()");
    hipEvent_t* ev = n_frames ? g_timing.events() : nullptr;
    g_timing.recorded = ev ? 3 : 0;
    sela::EncodeHostLink fused = {}; // (debug hook: the one-launch form on device pointers -- nothing mirrored, no stream position, no stagers)
    fused.wait_naps = -1;
    const bool use_fused = g_fused_device && !d_trace && !g_phase_cycles;
    if (use_fused)
        g_timing.recorded = ev ? 1 : 0;
    int dev = -1;
    const hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t priorities = launch_priorities(st, dev);
    // cut in two?  (see Splitter)
    uint32_t first = 0;
    const int split_mode = g_encode_split.load(std::memory_order_relaxed);
    if (dev >= 0 && n_frames && split_mode > 0 && !ev && !d_trace && !g_phase_cycles && !use_fused && g_team_lanes < 0 && !stream_is_capturing(st)) {
        first = sela::encode_split_frames(n_frames, channels, split_mode);
        if (first && second_half_offset(first, channels) + sela::encode_workspace_bytes(n_frames - first, channels) > workspace_bytes)
            first = 0;
    }
    hipError_t e = hipSuccess;
    bool split_done = false;
    if (first) {
        Splitter& sp = splitter();
        std::unique_lock<std::mutex> lock(sp.mu, std::try_to_lock); // (one split launch at a time per process: a second caller is not alone anyway)
        if (lock.owns_lock() && sp.ready(dev)) {
            const uint32_t rest = n_frames - first;
            void* const ws2 = static_cast<uint8_t*>(d_workspace) + second_half_offset(first, channels);
            const int16_t* const pcm2 = d_pcm + (size_t)first * SELA_HIP_SAMPLES_PER_FRAME * channels;
            const hipStream_t side = sp.side[dev];
            e = hipEventRecord(sp.begun[dev], st);
            if (e == hipSuccess)
                e = hipStreamWaitEvent(side, sp.begun[dev], 0);
            if (e == hipSuccess) // the first half's blocks, on the caller's stream
                e = sela::launch_encode(d_pcm, first, channels, d_frames, frames_cap, d_frame_offsets, d_status, d_workspace, nullptr, st, nullptr, nullptr, nullptr,
                    g_force_plain_fir, g_self_blocks, 16, nullptr, priorities, 1);
            if (e == hipSuccess) // the second half's, beside them
                e = sela::launch_encode(pcm2, rest, channels, d_frames, frames_cap, d_frame_offsets + first, d_status, ws2, nullptr, side, nullptr, nullptr, nullptr,
                    g_force_plain_fir, g_self_blocks, 16, nullptr, priorities, 1);
            if (e == hipSuccess)
                e = hipEventRecord(sp.half_done[dev], side);
            if (e == hipSuccess) // the first half's plan + assemble, under the second half's tail
                e = sela::launch_encode(d_pcm, first, channels, d_frames, frames_cap, d_frame_offsets, d_status, d_workspace, nullptr, st, nullptr, nullptr, nullptr,
                    g_force_plain_fir, g_self_blocks, 16, nullptr, priorities, 2);
            if (e == hipSuccess)
                e = hipStreamWaitEvent(st, sp.half_done[dev], 0);
            if (e == hipSuccess) // the second half's frames behind the first's
                e = sela::launch_encode(pcm2, rest, channels, d_frames, frames_cap, d_frame_offsets + first, d_status, ws2, nullptr, st, nullptr, nullptr, nullptr,
                    g_force_plain_fir, g_self_blocks, 16, nullptr, priorities, 2, d_frame_offsets + first, true);
            sp.last[d_workspace] = std::make_pair(n_frames, first);
            g_launches_split.fetch_add(1, std::memory_order_relaxed);
            split_done = true;
        }
    }
    if (!split_done) {
        e = sela::launch_encode(d_pcm, n_frames, channels, d_frames, frames_cap, d_frame_offsets, d_status, d_workspace,
            d_trace, st, ev, g_phase_cycles, use_fused ? &fused : nullptr, g_force_plain_fir, g_self_blocks, g_team_lanes, nullptr,
            priorities);
        if (n_frames) {
            std::lock_guard<std::mutex> lock(splitter().mu);
            const auto it = splitter().last.find(d_workspace);
            if (it != splitter().last.end())
                it->second = std::make_pair(n_frames, 0u);
        }
    }
    if (e != hipSuccess)
        return fail_hip(e, "encode launch");
    if (dev >= 0 && n_frames)
        if (!stream_is_capturing(static_cast<hipStream_t>(stream)))
            flights().note(dev, static_cast<hipStream_t>(stream));
    return SELA_HIP_OK;
}

int sela_hip_decode_device(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream)
{
    if (channels == 0 || channels > 255)
        return fail(SELA_HIP_EINVAL, "channels must be in 1..255");
    if (channels > sela::decode_max_channels())
        return fail(SELA_HIP_EINVAL, "too many channels for the on-chip decoder (sela_hip_decode_max_channels)");
    if (!d_status || (n_frames && (!d_frames || !d_frame_offsets || !d_pcm_out || !d_workspace)))
        return fail(SELA_HIP_EINVAL, "null device pointer");
    if ((uintptr_t)d_frames & 3)
        return fail(SELA_HIP_EINVAL, "d_frames must be 4-byte aligned");
    if (workspace_bytes < sela::decode_workspace_bytes(n_frames, channels))
        return fail(SELA_HIP_ECAPACITY, "workspace smaller than sela_hip_decode_workspace_bytes()");
    hipEvent_t* ev = n_frames ? g_timing.events() : nullptr;
    g_timing.recorded = ev ? 1 : 0;
    // (the decoder's counterpart of the encoder's schedule: a launch that has the device to itself raises its heavy subframes)
    int dev = -1;
    uint32_t synth_priorities = 0;
    if (n_frames && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        const int64_t forced = g_forced_priorities.load(std::memory_order_relaxed);
        synth_priorities = forced >= 0 ? (forced ? kHeavySubframesFirst : 0u) : (flights().others_pending(dev, static_cast<hipStream_t>(stream)) ? 0u : kHeavySubframesFirst);
    } else {
        dev = -1;
    }
    hipError_t e = sela::launch_decode(d_frames, d_frame_offsets, n_frames, channels, d_pcm_out, d_status, d_workspace,
        static_cast<hipStream_t>(stream), ev, g_phase_cycles, nullptr, g_recurrence_form, synth_priorities);
    if (e != hipSuccess)
        return fail_hip(e, "decode launch");
    if (dev >= 0)
        if (!stream_is_capturing(static_cast<hipStream_t>(stream)))
            flights().note(dev, static_cast<hipStream_t>(stream));
    return SELA_HIP_OK;
}

// ---- streaming jobs (host pointers) ----------------------------------------------------------------------------
int sela_hip_encode_begin(sela_hip_job** job, uint32_t channels, uint32_t total_frames, uint8_t* frames_out, size_t frames_cap,
    uint64_t* frame_offsets_out)
{
    if (!frame_offsets_out || (total_frames && !frames_out))
        return fail(SELA_HIP_EINVAL, "bad argument");
    int rc = job_begin(job, true, channels, total_frames);
    if (rc != SELA_HIP_OK)
        return rc;
    sela_hip_job* j = *job;
    j->frames_out = frames_out;
    j->frames_cap = frames_cap;
    j->offsets_out = frame_offsets_out;
    frame_offsets_out[0] = 0;
    // the kernels store the stream where the device can reach it: the caller's buffer if it is page-locked,
    // otherwise a page-locked bounce buffer (no larger than the job can need)
    j->out_host = frames_out;
    j->out_mapped = static_cast<uint8_t*>(device_view(frames_out));
    if (total_frames && !j->out_mapped) {
        j->frames_cap = std::min(frames_cap, sela_hip_encode_bound_bytes(total_frames, channels));
        j->out_host = static_cast<uint8_t*>(pool().take(j->frames_cap ? j->frames_cap : 1));
        j->out_bounce = true;
        j->out_mapped = j->out_host ? static_cast<uint8_t*>(device_view(j->out_host)) : nullptr;
        if (!j->out_mapped) {
            (void)job_end(j, nullptr, nullptr);
            *job = nullptr;
            return fail(SELA_HIP_ENOMEM, "page-locked bounce buffer");
        }
    }
    return SELA_HIP_OK;
}

int sela_hip_encode_feed(sela_hip_job* job, const int16_t* pcm, uint32_t n_frames, uint32_t* frames_final, uint64_t* bytes_final)
{
    if (!job || !job->encode || (n_frames && !pcm) || (uint64_t)job->fed + n_frames > job->total_frames)
        return fail(SELA_HIP_EINVAL, "bad argument");
    if (job->error != SELA_HIP_OK)
        return job->error;
    int rc = job_feed_encode(job, pcm, n_frames);
    if (rc != SELA_HIP_OK)
        return rc;
    rc = job_drain_ready(job);
    if (rc != SELA_HIP_OK)
        return rc;
    job_progress(job, frames_final, bytes_final);
    return SELA_HIP_OK;
}

int sela_hip_encode_end(sela_hip_job* job, uint32_t* frames_final, uint64_t* bytes_final)
{
    if (!job || !job->encode)
        return fail(SELA_HIP_EINVAL, "bad argument");
    return job_end(job, frames_final, bytes_final);
}

int sela_hip_decode_begin(sela_hip_job** job, uint32_t channels, uint32_t total_frames, int16_t* pcm_out)
{
    if (total_frames && !pcm_out)
        return fail(SELA_HIP_EINVAL, "bad argument");
    int rc = job_begin(job, false, channels, total_frames);
    if (rc != SELA_HIP_OK)
        return rc;
    sela_hip_job* j = *job;
    j->pcm_out = j->pcm_caller = pcm_out;
    if (total_frames && !device_view(pcm_out)) { // ordinary memory: the copy-outs land in a page-locked buffer first
        const size_t bytes = (size_t)total_frames * sela::kBlock * channels * sizeof(int16_t);
        j->pcm_out = static_cast<int16_t*>(pool().take(bytes));
        j->pcm_bounce = true;
        if (!j->pcm_out) {
            (void)job_end(j, nullptr, nullptr);
            *job = nullptr;
            return fail(SELA_HIP_ENOMEM, "page-locked bounce buffer");
        }
    }
    return SELA_HIP_OK;
}

int sela_hip_decode_feed(sela_hip_job* job, const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t* frames_final)
{
    if (!job || job->encode || (n_frames && (!frames || !frame_offsets)) || (uint64_t)job->fed + n_frames > job->total_frames)
        return fail(SELA_HIP_EINVAL, "bad argument");
    if (job->error != SELA_HIP_OK)
        return job->error;
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f] || (frame_offsets[f] & 3))
            return job_fail(job, fail(SELA_HIP_EFORMAT, "frame offsets must be ascending multiples of 4"));
    int rc = job_feed_decode(job, frames, frame_offsets, n_frames);
    if (rc != SELA_HIP_OK)
        return rc;
    rc = job_drain_ready(job);
    if (rc != SELA_HIP_OK)
        return rc;
    job_progress(job, frames_final, nullptr);
    return SELA_HIP_OK;
}

int sela_hip_decode_end(sela_hip_job* job, uint32_t* frames_final)
{
    if (!job || job->encode)
        return fail(SELA_HIP_EINVAL, "bad argument");
    return job_end(job, frames_final, nullptr);
}

// ---- one-shot host-pointer API -----------------------------------------------------------------------------------
namespace {

// One call, as it is: a job of one feed.
int encode_now(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint8_t* frames_out, size_t frames_cap, uint64_t* frame_offsets_out)
{
    sela_hip_job* job = nullptr;
    int rc = sela_hip_encode_begin(&job, channels, n_frames, frames_out, frames_cap, frame_offsets_out);
    if (rc != SELA_HIP_OK)
        return rc;
    rc = sela_hip_encode_feed(job, pcm, n_frames, nullptr, nullptr);
    const std::string msg = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
    const int rc_end = sela_hip_encode_end(job, nullptr, nullptr);
    if (rc != SELA_HIP_OK)
        return fail(rc, msg);
    return rc_end;
}

int decode_now(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out)
{
    sela_hip_job* job = nullptr;
    int rc = sela_hip_decode_begin(&job, channels, n_frames, pcm_out);
    if (rc != SELA_HIP_OK)
        return rc;
    rc = sela_hip_decode_feed(job, frames, frame_offsets, n_frames, nullptr);
    const std::string msg = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
    const int rc_end = sela_hip_decode_end(job, nullptr);
    if (rc != SELA_HIP_OK)
        return fail(rc, msg);
    return rc_end;
}

// ---- small calls from many threads are coalesced: sela_coalescer.h, on this backend ---------------------------------------
struct HipBackend {
    static int encode_now(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint8_t* frames_out, size_t frames_cap, uint64_t* offsets_out)
    {
        return ::encode_now(pcm, n_frames, channels, frames_out, frames_cap, offsets_out);
    }
    static int decode_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out)
    {
        return ::decode_now(frames, offsets, n_frames, channels, pcm_out);
    }
    static int decode_i32_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int32_t* samples_out, uint32_t stride,
        uint32_t* counts_out)
    {
        return sela::generic_decode(frames, offsets, n_frames, channels, samples_out, stride, counts_out, nullptr, nullptr);
    }
    static int encode_i32_now(const int32_t* samples, uint32_t n_frames, uint32_t channels, uint32_t n, uint8_t* frames_out, size_t frames_cap, uint64_t* offsets_out)
    {
        return sela::generic_encode(samples, false, n_frames, channels, n, frames_out, frames_cap, offsets_out);
    }
    static size_t encode_bound_bytes(uint32_t n_frames, uint32_t channels) { return sela_hip_encode_bound_bytes(n_frames, channels); }
    static size_t encode_i32_bound_bytes(uint32_t n_frames, uint32_t channels, uint32_t n) { return sela::generic_encode_bound_bytes(n_frames, channels, n); }
    static void* take(size_t bytes) { return pool().take(bytes); }
    static void give(void* p) { pool().give(p); }
    static std::string last_error() { return sela_hip_last_error(); }
    static void after_batch() // (both routes' leases: whoever leads next takes them over instead of creating a set of its own)
    {
        g_lease.give_back();
        sela::generic_release();
    }
};
using sela::kCoalesceFrames;
using sela::SmallCall;
typedef sela::CallCoalescer<HipBackend> Coalescer;

// One coalescer per device and direction: calls for different GPUs (one thread per GPU, each coding frame by frame) can never
// share a batch, so they do not wait for each other's leaders either.
Coalescer* coalescer(Coalescer::Kind kind, int device)
{
    static std::mutex mu;
    static Coalescer* table[4][64] = {};
    const int d = device >= 0 && device < 64 ? device : 0;
    std::lock_guard<std::mutex> lock(mu);
    Coalescer*& c = table[(int)kind][d];
    if (!c)
        c = new Coalescer(kind, kind == Coalescer::kDecode || kind == Coalescer::kDecode32 ? sela::kCoalesceLeadersDecode : sela::kCoalesceLeaders); // (never destroyed: calls may outlive the statics)
    return c;
}

int submit_small(Coalescer::Kind kind, SmallCall& call)
{
    const int rc = coalescer(kind, call.device)->submit(call);
    return rc == SELA_HIP_OK ? SELA_HIP_OK : fail(rc, call.error);
}
} // namespace

int sela_hip_encode(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel, uint8_t* frames_out,
    size_t frames_cap, uint64_t* frame_offsets_out)
{
    if (samples_per_channel == 0 || samples_per_channel > 65535)
        return fail(SELA_HIP_EINVAL, "samples_per_channel must be 1 .. 65535 (the subframe's field is 16 bits wide)");
    if (channels == 0 || channels > 255 || !frame_offsets_out || (n_frames && (!pcm || !frames_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    if (samples_per_channel != SELA_HIP_SAMPLES_PER_FRAME) // not the shape the fast kernels are built for: the any-length route
        return sela::generic_encode(pcm, true, n_frames, channels, samples_per_channel, frames_out, frames_cap, frame_offsets_out);
    if (n_frames == 0 || n_frames > kCoalesceFrames || (g_lease.held && g_lease.held->job_open)) // (a thread in the middle of a job of its own hears about that itself)
        return encode_now(pcm, n_frames, channels, frames_out, frames_cap, frame_offsets_out);
    SmallCall call;
    if (hipGetDevice(&call.device) != hipSuccess)
        return encode_now(pcm, n_frames, channels, frames_out, frames_cap, frame_offsets_out); // (reports the missing device)
    call.channels = channels, call.n_frames = n_frames;
    call.pcm = pcm, call.frames_out = frames_out, call.frames_cap = frames_cap, call.offsets_out = frame_offsets_out;
    return submit_small(Coalescer::kEncode, call);
}

namespace {
// the fast route: 2048 samples per channel and frame
int decode_standard(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out)
{
    if (n_frames == 0 || n_frames > kCoalesceFrames || (g_lease.held && g_lease.held->job_open))
        return decode_now(frames, frame_offsets, n_frames, channels, pcm_out);
    for (uint32_t f = 0; f < n_frames; f++) // (what the job would refuse is refused by the job, for this caller alone)
        if (frame_offsets[f + 1] < frame_offsets[f] || (frame_offsets[f] & 3))
            return decode_now(frames, frame_offsets, n_frames, channels, pcm_out);
    SmallCall call;
    if (hipGetDevice(&call.device) != hipSuccess)
        return decode_now(frames, frame_offsets, n_frames, channels, pcm_out);
    call.channels = channels, call.n_frames = n_frames;
    call.frames = frames, call.offsets_in = frame_offsets, call.pcm_out = pcm_out;
    return submit_small(Coalescer::kDecode, call);
}
} // namespace

int sela_hip_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out)
{
    if (channels == 0 || channels > 255 || (n_frames && (!frames || !frame_offsets || !pcm_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    // The fast kernels first, unasked: a stream of 2048-sample frames -- every stream an encoder writes -- pays nothing for the
    // other kind (a walk over the headers in host memory is a cache miss or two per frame: 0.1 ms and more for a 3-minute
    // track, a tenth of the whole call).  They refuse a subframe that does not say 2048 (SELA_HIP_EFORMAT, nothing written
    // beyond [n_frames][2048][channels], which is why pcm_out must hold that much, sela_hip.h); only then are the headers
    // walked, and a stream that turns out to be of the other kind goes, whole, down the any-length route.
    // (One look is free: the first subframe of the first frame.  A stream that says another length there is not offered to
    // the fast kernels at all -- they would write 2048-sample frames of silence into a pcm_out its caller sized from
    // sela_hip_index_samples().  A stream that turns odd LATER still needs the room the header asks for.)
    bool first_is_standard = true;
    if (n_frames && frame_offsets[1] >= frame_offsets[0] + 16) {
        const uint8_t* fb = frames + frame_offsets[0];
        const uint64_t fbytes = frame_offsets[1] - frame_offsets[0];
        const uint64_t cw = (uint64_t)fb[8] | ((uint64_t)fb[9] << 8), p2 = 4 + 7 + 4 * cw;
        if (p2 + 5 <= fbytes)
            first_is_standard = ((uint32_t)fb[p2 + 3] | ((uint32_t)fb[p2 + 4] << 8)) == SELA_HIP_SAMPLES_PER_FRAME;
    }
    const int rc = first_is_standard ? decode_standard(frames, frame_offsets, n_frames, channels, pcm_out) : SELA_HIP_EFORMAT;
    if (rc != SELA_HIP_EFORMAT || n_frames == 0)
        return rc;
    const std::string first_error = first_is_standard ? std::string(sela_hip_last_error()) : std::string("malformed frame stream");
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f])
            return rc;
    bool standard = true;
    std::vector<uint64_t> sample_offsets((size_t)n_frames + 1);
    const uint32_t largest = sela::generic_index_samples(frames, frame_offsets, n_frames, channels, sample_offsets.data(), &standard);
    if (standard || largest == 0)
        return fail(rc, first_error); // (malformed in the ordinary sense)
    return sela::generic_decode(frames, frame_offsets, n_frames, channels, nullptr, largest, nullptr, pcm_out, sample_offsets.data());
}

size_t sela_hip_encode_bound_bytes_n(uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel)
{
    if (samples_per_channel == SELA_HIP_SAMPLES_PER_FRAME)
        return sela_hip_encode_bound_bytes(n_frames, channels);
    return sela::generic_encode_bound_bytes(n_frames, channels, samples_per_channel);
}

uint32_t sela_hip_index_samples(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, uint64_t* sample_offsets)
{
    if (!frames || !frame_offsets || channels == 0) {
        if (sample_offsets)
            for (uint32_t f = 0; f <= n_frames; f++)
                sample_offsets[f] = 0;
        return 0;
    }
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f])
            return 0;
    return sela::generic_index_samples(frames, frame_offsets, n_frames, channels, sample_offsets, nullptr);
}

int sela_hip_encode_i32(const int32_t* samples, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel, uint8_t* frames_out, size_t frames_cap,
    uint64_t* frame_offsets_out)
{
    if (samples_per_channel == 0 || samples_per_channel > 65535)
        return fail(SELA_HIP_EINVAL, "samples_per_channel must be 1 .. 65535 (the subframe's field is 16 bits wide)");
    if (channels == 0 || channels > 255 || !frame_offsets_out || (n_frames && (!samples || !frames_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    // Small calls from many threads -- the reference's thread loop over frame::FrameEncoder (src/sela/encoder.cpp:58-73) on frames
    // that are not the CLI's shape -- go to the device together, like the one-shot calls of the fast path (sela_coalescer.h):
    // calls of one shape (channels, samples per channel) share a job, every call gets its own bytes and its own error.
    SmallCall call;
    if (n_frames == 0 || n_frames > kCoalesceFrames || (size_t)n_frames * channels * samples_per_channel > ((size_t)1 << 20) || hipGetDevice(&call.device) != hipSuccess)
        return sela::generic_encode(samples, false, n_frames, channels, samples_per_channel, frames_out, frames_cap, frame_offsets_out);
    call.channels = channels, call.n_frames = n_frames, call.shape = samples_per_channel;
    call.samples = samples, call.frames_out = frames_out, call.frames_cap = frames_cap, call.offsets_out = frame_offsets_out;
    return submit_small(Coalescer::kEncode32, call);
}

// One frame whose channels differ in length (src/frame/frame_encoder.cpp:11-102): every channel is a block of its own -- coded
// as a one-channel frame of its own length by the any-length kernels -- and the frame is their subframes, renamed, behind one
// sync word; the second channel of an exactly-stereo frame against the difference channel 0 - channel 1 over its own length.
int sela_hip_encode_ragged_i32(const int32_t* samples, const uint32_t* lengths, uint32_t channels, uint8_t* frame_out, size_t frame_cap, size_t* frame_bytes)
{
    if (channels == 0 || channels > 255 || !samples || !lengths || !frame_out || !frame_bytes)
        return fail(SELA_HIP_EINVAL, "bad argument");
    for (uint32_t c = 0; c < channels; c++)
        if (lengths[c] == 0 || lengths[c] > 65535)
            return fail(SELA_HIP_EINVAL, "every channel holds 1 .. 65535 samples (the subframe's field is 16 bits wide)");
    if (channels == 2 && lengths[0] < lengths[1])
        return fail(SELA_HIP_EINVAL, "an exactly-stereo frame whose first channel is the shorter one: the reference's difference signal reads it past its end (src/frame/frame_encoder.cpp:22-24)");
    if (frame_cap < 4)
        return fail(SELA_HIP_ECAPACITY, "frame_out too small");
    // a one-channel frame: sync word (4) | channel, type, parent | k, words (u16), order | words | k, words (u16), n (u16) | words
    auto mono = [&](const int32_t* x, uint32_t n, std::vector<uint8_t>& bytes, uint32_t& words) -> int {
        bytes.resize(sela::generic_encode_bound_bytes(1, 1, n));
        uint64_t offs[2] = { 0, 0 };
        const int rc = sela::generic_encode(x, false, 1, 1, n, bytes.data(), bytes.size(), offs);
        if (rc != SELA_HIP_OK)
            return rc;
        bytes.resize((size_t)offs[1]);
        const uint32_t cw = bytes[8] | ((uint32_t)bytes[9] << 8);
        const size_t p2 = 4 + 7 + 4 * (size_t)cw;
        words = cw + (bytes[p2 + 1] | ((uint32_t)bytes[p2 + 2] << 8));
        return SELA_HIP_OK;
    };
    const uint32_t sync = SELA_SYNC_WORD;
    std::memcpy(frame_out, &sync, 4);
    size_t at = 4;
    const int32_t* cur = samples;
    std::vector<uint8_t> own, dif;
    std::vector<int32_t> d;
    for (uint32_t c = 0; c < channels; c++) {
        const uint32_t n = lengths[c];
        uint32_t own_words = 0, dif_words = 0;
        int rc = mono(cur, n, own, own_words);
        if (rc != SELA_HIP_OK)
            return rc;
        const std::vector<uint8_t>* take = &own;
        uint8_t type = 0, parent = (uint8_t)c;
        if (channels == 2 && c == 1) { // :18-72
            d.resize(n);
            for (uint32_t j = 0; j < n; j++)
                d[j] = (int32_t)((uint32_t)samples[j] - (uint32_t)cur[j]);
            rc = mono(d.data(), n, dif, dif_words);
            if (rc != SELA_HIP_OK)
                return rc;
            if (dif_words < own_words) // :64-66: strictly fewer words
                take = &dif, type = 1, parent = 0;
        }
        const size_t sub = take->size() - 4;
        if (at + sub > frame_cap)
            return fail(SELA_HIP_ECAPACITY, "frame_out too small (4 + the sum over the channels of sela_hip_encode_bound_bytes_n(1, 1, lengths[c]) holds any frame of 17-bit samples)");
        std::memcpy(frame_out + at, take->data() + 4, sub);
        frame_out[at] = (uint8_t)c, frame_out[at + 1] = type, frame_out[at + 2] = parent;
        at += sub;
        cur += n;
    }
    *frame_bytes = at;
    return SELA_HIP_OK;
}

int sela_hip_decode_i32(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int32_t* samples_out, uint32_t stride,
    uint32_t* counts_out)
{
    if (channels == 0 || channels > 255 || (n_frames && (!frames || !frame_offsets || !samples_out || !counts_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    if (n_frames == 0)
        return SELA_HIP_OK;
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f])
            return fail(SELA_HIP_EFORMAT, "frame offsets must not decrease");
    const uint32_t largest = sela::generic_index_samples(frames, frame_offsets, n_frames, channels, nullptr, nullptr);
    if (largest > stride)
        return fail(SELA_HIP_ECAPACITY, "stride is smaller than the largest samplesPerChannel of the stream (see sela_hip_index_samples)");
    // (a stream the host walk cannot follow reports 0: the kernels find and report the malformed frame)
    // Small calls from many threads -- the reference's thread loop over frame::FrameDecoder -- go to the device together, like
    // the one-shot calls of the fast path (sela_coalescer.h): every call its own rows, its own error.
    SmallCall call;
    if (n_frames > kCoalesceFrames || hipGetDevice(&call.device) != hipSuccess)
        return sela::generic_decode(frames, frame_offsets, n_frames, channels, samples_out, stride, counts_out, nullptr, nullptr);
    call.channels = channels, call.n_frames = n_frames;
    call.frames = frames, call.offsets_in = frame_offsets, call.samples_out = samples_out, call.stride = stride, call.counts_out = counts_out;
    return submit_small(Coalescer::kDecode32, call);
}

int sela_hip_lpc_encode_n(const int32_t* samples, uint32_t n_blocks, uint32_t samples_per_block, int32_t* order_out, int32_t* q_out, int32_t* residues_out)
{
    if (n_blocks && (!samples || !order_out || !q_out || !residues_out))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (samples_per_block == 0 || samples_per_block > (1u << 24))
        return fail(SELA_HIP_EINVAL, "samples_per_block must be 1 .. 2^24");
    if (n_blocks == 0)
        return stage_device_ready();
    return sela::generic_lpc_encode(samples, n_blocks, samples_per_block, order_out, q_out, residues_out);
}

int sela_hip_lpc_decode_n(const int32_t* order, const int32_t* q, const int32_t* residues, uint32_t n_blocks, uint32_t samples_per_block, int32_t* samples_out,
    int64_t* coefs_out)
{
    if (n_blocks && (!order || !q || (samples_out && !residues) || (!samples_out && !coefs_out)))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (samples_per_block == 0 || samples_per_block > (1u << 24))
        return fail(SELA_HIP_EINVAL, "samples_per_block must be 1 .. 2^24");
    if (n_blocks == 0)
        return stage_device_ready();
    return sela::generic_lpc_decode(order, q, residues, n_blocks, samples_per_block, samples_out, coefs_out);
}

int sela_hip_lpc_encode(const int32_t* samples, uint32_t n_blocks, int32_t* order_out, int32_t* q_out, int32_t* residues_out)
{
    if (n_blocks && (!samples || !order_out || !q_out || !residues_out))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (stage_device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    if (n_blocks == 0)
        return SELA_HIP_OK;
    // A block is analysed as the DIFFERENCE signal of a stereo frame (left - right = the block's samples): that is how 17-bit
    // samples reach the kernels, whose input is the 16-bit PCM of a WAV file.  The trace build of k_encode_blocks leaves the
    // order and the quantised coefficients of every signal in its trace, and here the residues too.
    constexpr size_t kBlk = SELA_HIP_SAMPLES_PER_FRAME;
    std::vector<int16_t> pcm((size_t)n_blocks * kBlk * 2);
    for (size_t i = 0; i < (size_t)n_blocks * kBlk; i++) {
        const int32_t s = samples[i];
        if (s < -65535 || s > 65535) // beyond 16-bit channels and their difference: not the fast kernels' input
            return sela::generic_lpc_encode(samples, n_blocks, SELA_HIP_SAMPLES_PER_FRAME, order_out, q_out, residues_out);
        const int32_t left = s > 32767 ? 32767 : (s < -32768 ? -32768 : s);
        pcm[2 * i] = (int16_t)left;
        pcm[2 * i + 1] = (int16_t)(left - s);
    }
    hipError_t e = hipSuccess;
    Scratch mem;
    const size_t ws = sela::encode_workspace_bytes(n_blocks, 2), cap = sela_hip_encode_bound_bytes(n_blocks, 2);
    int16_t* d_pcm = mem.upload(pcm.data(), pcm.size(), e);
    uint8_t* d_frames = mem.take<uint8_t>(cap, e);
    uint64_t* d_offsets = mem.take<uint64_t>((size_t)n_blocks + 1, e);
    uint32_t* d_status = mem.take<uint32_t>(4, e);
    uint8_t* d_ws = mem.take<uint8_t>(ws, e);
    sela_hip_trace* d_trace = mem.take<sela_hip_trace>((size_t)n_blocks * 3, e);
    int32_t* d_res = mem.take<int32_t>((size_t)n_blocks * 3 * kBlk, e);
    if (e == hipSuccess)
        e = sela::launch_encode(d_pcm, n_blocks, 2, d_frames, cap, d_offsets, d_status, d_ws, d_trace, nullptr, nullptr, nullptr, nullptr, 0, -1, 0, d_res);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    std::vector<sela_hip_trace> trace((size_t)n_blocks * 3);
    if (e == hipSuccess)
        e = hipMemcpy(trace.data(), d_trace, trace.size() * sizeof(sela_hip_trace), hipMemcpyDeviceToHost);
    for (uint32_t b = 0; b < n_blocks && e == hipSuccess; b++) // signal 2 of frame b
        e = hipMemcpy(residues_out + (size_t)b * kBlk, d_res + ((size_t)b * 3 + 2) * kBlk, kBlk * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail_hip(e, "lpc_encode");
    uint32_t flags = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const sela_hip_trace& t = trace[(size_t)b * 3 + 2];
        order_out[b] = t.order;
        for (int i = 0; i < SELA_MAX_LPC_ORDER; i++)
            q_out[(size_t)b * SELA_MAX_LPC_ORDER + i] = i < t.order ? t.q[i] : 0;
        flags |= t.flags;
    }
    if (flags & (SELA_HIP_FLAG_RICE_RANGE | SELA_HIP_FLAG_COEF_OVERFLOW))
        return fail(SELA_HIP_ERANGE, "lpc_encode: a block left the range the format can carry");
    return SELA_HIP_OK;
}

int sela_hip_lpc_decode(const int32_t* order, const int32_t* q, const int32_t* residues, uint32_t n_blocks, int32_t* samples_out, int64_t* coefs_out)
{
    if (n_blocks && (!order || !q || (samples_out && !residues) || (!samples_out && !coefs_out)))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (stage_device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    if (n_blocks == 0)
        return SELA_HIP_OK;
    constexpr size_t kBlk = SELA_HIP_SAMPLES_PER_FRAME, kCoefs = SELA_MAX_LPC_ORDER + 1;
    hipError_t e = hipSuccess;
    Scratch mem;
    int32_t* d_order = mem.upload(order, n_blocks, e);
    int32_t* d_q = mem.upload(q, (size_t)n_blocks * SELA_MAX_LPC_ORDER, e);
    int32_t* d_res = samples_out ? mem.upload(residues, (size_t)n_blocks * kBlk, e) : nullptr;
    int32_t* d_out = samples_out ? mem.take<int32_t>((size_t)n_blocks * kBlk, e) : nullptr;
    int64_t* d_coefs = coefs_out ? mem.take<int64_t>((size_t)n_blocks * kCoefs, e) : nullptr;
    uint32_t* d_status = mem.take<uint32_t>(4, e);
    if (e == hipSuccess)
        e = hipMemset(d_status, 0, 16);
    if (e == hipSuccess && d_coefs)
        e = hipMemset(d_coefs, 0, (size_t)n_blocks * kCoefs * sizeof(int64_t));
    if (e == hipSuccess)
        e = sela::launch_stage_lpc_decode(d_order, d_q, d_res, n_blocks, d_out, d_coefs, d_status, nullptr);
    uint32_t status[4] = {};
    if (e == hipSuccess)
        e = hipMemcpy(status, d_status, 16, hipMemcpyDeviceToHost); // (synchronises)
    if (e == hipSuccess && samples_out)
        e = hipMemcpy(samples_out, d_out, (size_t)n_blocks * kBlk * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && coefs_out)
        e = hipMemcpy(coefs_out, d_coefs, (size_t)n_blocks * kCoefs * sizeof(int64_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail_hip(e, "lpc_decode");
    if (status[0] & SELA_HIP_FLAG_BAD_FRAME)
        return fail(SELA_HIP_EINVAL, "lpc_decode: order outside 0..100");
    if (status[0] & SELA_HIP_FLAG_COEF_OVERFLOW)
        return fail(SELA_HIP_ERANGE, "lpc_decode: a predictor coefficient left the int64 range");
    if (status[0] & SELA_HIP_FLAG_Q_RANGE)
        return fail(SELA_HIP_ERANGE, "lpc_decode: a quantised reflection coefficient outside [-64, 63] (the reference indexes past its tables, src/lpc/linear_predictor.cpp:23-26)");
    return SELA_HIP_OK;
}

int sela_hip_rice_encode(const int32_t* values, const uint64_t* value_offsets, uint32_t n_streams, uint32_t* k_out, uint32_t* word_counts_out,
    uint32_t* words_out, const uint64_t* word_offsets)
{
    if (n_streams && (!value_offsets || !k_out || !word_counts_out || !word_offsets))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (stage_device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    if (n_streams == 0)
        return SELA_HIP_OK;
    for (uint32_t i = 0; i < n_streams; i++)
        if (value_offsets[i + 1] < value_offsets[i] || word_offsets[i + 1] < word_offsets[i])
            return fail(SELA_HIP_EINVAL, "offsets must not decrease");
    const size_t n_values = (size_t)value_offsets[n_streams], n_words = (size_t)word_offsets[n_streams];
    if ((n_values && !values) || (n_words && !words_out))
        return fail(SELA_HIP_EINVAL, "null pointer");
    hipError_t e = hipSuccess;
    Scratch mem;
    int32_t* d_values = mem.upload(values, n_values, e);
    uint64_t* d_voff = mem.upload(value_offsets, (size_t)n_streams + 1, e);
    uint64_t* d_woff = mem.upload(word_offsets, (size_t)n_streams + 1, e);
    uint32_t* d_k = mem.take<uint32_t>(n_streams, e);
    uint32_t* d_counts = mem.take<uint32_t>(n_streams, e);
    uint32_t* d_words = mem.take<uint32_t>(n_words, e);
    uint32_t* d_status = mem.take<uint32_t>(4, e);
    if (e == hipSuccess)
        e = hipMemset(d_status, 0, 16);
    if (e == hipSuccess && n_words)
        e = hipMemset(d_words, 0, n_words * sizeof(uint32_t));
    if (e == hipSuccess)
        e = sela::launch_stage_rice_encode(d_values, d_voff, n_streams, d_k, d_counts, d_words, d_woff, d_status, nullptr);
    uint32_t status[4] = {};
    if (e == hipSuccess)
        e = hipMemcpy(status, d_status, 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(k_out, d_k, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(word_counts_out, d_counts, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_words)
        e = hipMemcpy(words_out, d_words, n_words * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail_hip(e, "rice_encode");
    if (status[0] & SELA_HIP_FLAG_RICE_RANGE)
        return fail(SELA_HIP_ERANGE, "rice_encode: a value is beyond the reference's int32 zig-zag (|value| >= 2^30)");
    if (status[0] & SELA_HIP_FLAG_WORDS_CAP)
        return fail(SELA_HIP_ECAPACITY, "rice_encode: a stream needs more words than its range of words_out (see word_counts_out)");
    return SELA_HIP_OK;
}

int sela_hip_rice_decode(const uint32_t* words, const uint64_t* word_offsets, const uint32_t* k, const uint64_t* value_offsets, uint32_t n_streams,
    int32_t* values_out)
{
    if (n_streams && (!word_offsets || !k || !value_offsets))
        return fail(SELA_HIP_EINVAL, "null pointer");
    if (stage_device_ready() != SELA_HIP_OK)
        return SELA_HIP_ENODEV;
    if (n_streams == 0)
        return SELA_HIP_OK;
    for (uint32_t i = 0; i < n_streams; i++)
        if (value_offsets[i + 1] < value_offsets[i] || word_offsets[i + 1] < word_offsets[i] || k[i] >= 32)
            return fail(SELA_HIP_EINVAL, "offsets must not decrease; Rice parameters are below 32");
    const size_t n_values = (size_t)value_offsets[n_streams], n_words = (size_t)word_offsets[n_streams];
    if ((n_values && !values_out) || (n_words && !words))
        return fail(SELA_HIP_EINVAL, "null pointer");
    hipError_t e = hipSuccess;
    Scratch mem;
    uint32_t* d_words = mem.upload(words, n_words, e);
    uint64_t* d_woff = mem.upload(word_offsets, (size_t)n_streams + 1, e);
    uint64_t* d_voff = mem.upload(value_offsets, (size_t)n_streams + 1, e);
    uint32_t* d_k = mem.upload(k, n_streams, e);
    int32_t* d_values = mem.take<int32_t>(n_values, e);
    uint32_t* d_status = mem.take<uint32_t>(4, e);
    if (e == hipSuccess)
        e = hipMemset(d_status, 0, 16);
    if (e == hipSuccess)
        e = sela::launch_stage_rice_decode(d_words, d_woff, d_k, d_voff, n_streams, d_values, d_status, nullptr);
    uint32_t status[4] = {};
    if (e == hipSuccess)
        e = hipMemcpy(status, d_status, 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_values)
        e = hipMemcpy(values_out, d_values, n_values * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail_hip(e, "rice_decode");
    if (status[0] & (SELA_HIP_FLAG_RICE_OVERRUN | SELA_HIP_FLAG_BAD_FRAME))
        return fail(SELA_HIP_EFORMAT, "rice_decode: a stream ended before all its values were read");
    return SELA_HIP_OK;
}

uint32_t sela_hip_index_frames(const uint8_t* frames, size_t frames_bytes, uint32_t n_frames, uint32_t channels, uint64_t* frame_offsets)
{
    size_t off = 0;
    uint32_t f = 0;
    for (; f < n_frames; f++) {
        frame_offsets[f] = off;
        if (off + 4 > frames_bytes)
            break;
        uint32_t sync;
        std::memcpy(&sync, frames + off, 4);
        if (sync != SELA_SYNC_WORD) // src/file/sela_file.cpp:54-56: stop silently
            break;
        size_t p = off + 4;
        bool ok = true;
        for (uint32_t c = 0; c < channels && ok; c++) {
            if (p + 7 > frames_bytes) {
                ok = false;
                break;
            }
            const size_t cw = (size_t)frames[p + 4] | ((size_t)frames[p + 5] << 8);
            p += 7 + 4 * cw;
            if (p + 5 > frames_bytes) {
                ok = false;
                break;
            }
            const size_t rw = (size_t)frames[p + 1] | ((size_t)frames[p + 2] << 8);
            p += 5 + 4 * rw;
            if (p > frames_bytes)
                ok = false;
        }
        if (!ok)
            break;
        off = p;
    }
    frame_offsets[f] = off;
    return f;
}

} // extern "C"
