// sela_capi.hip -- the extern "C" boundary of libsela_hip.so (declared in include/sela_hip.h).
//
// Plain pointers and sizes only.  The *_device entry points enqueue kernels on the caller's stream;
// the host-pointer entry points stage through a per-thread, grow-only set of device buffers.
// There is no CPU fallback anywhere: without a HIP device every call fails with SELA_HIP_ENODEV.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sela_device.h"

namespace sela {
size_t encode_workspace_bytes(uint32_t n_frames, uint32_t channels);
hipError_t launch_encode(const int16_t* d_pcm, uint32_t n_frames, uint32_t channels, uint8_t* d_frames, size_t frames_cap,
    uint64_t* d_frame_offsets, uint32_t* d_status, void* d_workspace, sela_hip_trace* d_trace, hipStream_t stream,
    hipEvent_t* ev, uint64_t* d_phase_cycles);
hipError_t launch_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, hipStream_t stream, hipEvent_t* ev, uint64_t* d_phase_cycles,
    hipStream_t side, hipEvent_t fork, hipEvent_t* parsed);
size_t decode_workspace_bytes(uint32_t n_frames, uint32_t channels);
size_t decode_lds_bytes(uint32_t channels, int n_waves, uint32_t v_count);
int decode_waves(uint32_t channels);
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

// Host-pointer calls run as a chunked pipeline: while chunk i is in the kernels,
// chunk i+1 is copied in and chunk i-1 is copied out, on three streams.  Two sets of device buffers
// alternate; the workspace is shared because the kernels of consecutive chunks run in stream order.
// Frames per chunk; batches below two chunks go in one piece.  Encoding 1024 frames takes 0.22 ms, about
// as long as copying them in and out; decoding is latency-bound below a few thousand frames (one lane
// per stream), so its chunks are larger.
constexpr uint32_t kHostChunkFramesEncode = 1024;
constexpr uint32_t kHostChunkFramesDecode = 4096;

struct HostContext {
    DeviceBuffer pcm[2], frames[2], offsets[2], status[2], workspace;
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t copied_in[2] = { nullptr, nullptr }, ran[2] = { nullptr, nullptr };
    std::vector<uint64_t> rebased[2]; // decode: a chunk's frame offsets relative to its first byte
    int device = -1;
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
    hipError_t streams()
    {
        hipError_t e = hipSuccess;
        for (hipStream_t* s : { &s_in, &s_run, &s_out })
            if (!*s && (e = hipStreamCreateWithFlags(s, hipStreamNonBlocking)) != hipSuccess)
                return e;
        for (hipEvent_t* ev : { &copied_in[0], &copied_in[1], &ran[0], &ran[1] })
            if (!*ev && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess)
                return e;
        return e;
    }
    void release()
    {
        for (int b = 0; b < 2; b++) {
            pcm[b].release();
            frames[b].release();
            offsets[b].release();
            status[b].release();
        }
        workspace.release();
        for (hipStream_t* s : { &s_in, &s_run, &s_out }) {
            if (*s)
                (void)hipStreamDestroy(*s);
            *s = nullptr;
        }
        for (hipEvent_t* ev : { &copied_in[0], &copied_in[1], &ran[0], &ran[1] }) {
            if (*ev)
                (void)hipEventDestroy(*ev);
            *ev = nullptr;
        }
    }
};
thread_local HostContext g_ctx;

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
// Side stream + events of the decode pipeline (parse chunk j+1 overlaps synthesis of chunk j), per
// calling thread and device.
struct DecodePipeline {
    int device = -1;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr;
    hipEvent_t parsed[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    bool ready()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess)
            return false;
        if (dev == device && side)
            return true;
        release();
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) {
            side = nullptr;
            return false;
        }
        bool ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
        for (auto& e : parsed)
            ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            release();
            return false;
        }
        device = dev;
        return true;
    }
    void release()
    {
        if (side)
            (void)hipStreamDestroy(side);
        if (fork)
            (void)hipEventDestroy(fork);
        for (auto& e : parsed) {
            if (e)
                (void)hipEventDestroy(e);
            e = nullptr;
        }
        side = nullptr;
        fork = nullptr;
        device = -1;
    }
};
thread_local DecodePipeline g_pipeline;
thread_local uint64_t* g_phase_cycles = nullptr; // debug: per-block phase cycle counts (sela_hip_debug_phase_buffer)

uint32_t flags_to_error(uint32_t flags)
{
    return flags & (SELA_HIP_FLAG_WORDS_CAP | SELA_HIP_FLAG_RICE_RANGE | SELA_HIP_FLAG_COEF_OVERFLOW);
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

void sela_hip_shutdown(void)
{
    g_ctx.release();
    g_pipeline.release();
}

void sela_hip_debug_phase_buffer(uint64_t* d_cycles) { g_phase_cycles = d_cycles; }

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

size_t sela_hip_encode_workspace_bytes(uint32_t n_frames, uint32_t channels) { return sela::encode_workspace_bytes(n_frames, channels); }

size_t sela_hip_decode_workspace_bytes(uint32_t n_frames, uint32_t channels) { return sela::decode_workspace_bytes(n_frames, channels); }

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
    hipError_t e = sela::launch_encode(d_pcm, n_frames, channels, d_frames, frames_cap, d_frame_offsets, d_status, d_workspace,
        d_trace, static_cast<hipStream_t>(stream), ev, g_phase_cycles);
    if (e != hipSuccess)
        return fail_hip(e, "encode launch");
    return SELA_HIP_OK;
}

int sela_hip_decode_device(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream)
{
    if (channels == 0 || channels > 255)
        return fail(SELA_HIP_EINVAL, "channels must be in 1..255");
    if (sela::decode_lds_bytes(channels, sela::decode_waves(channels), SELA_HIP_SAMPLES_PER_FRAME) > 160 * 1024)
        return fail(SELA_HIP_EINVAL, "too many channels for the on-chip decoder (LDS budget)");
    if (!d_status || (n_frames && (!d_frames || !d_frame_offsets || !d_pcm_out || !d_workspace)))
        return fail(SELA_HIP_EINVAL, "null device pointer");
    if ((uintptr_t)d_frames & 3)
        return fail(SELA_HIP_EINVAL, "d_frames must be 4-byte aligned");
    if (workspace_bytes < sela::decode_workspace_bytes(n_frames, channels))
        return fail(SELA_HIP_ECAPACITY, "workspace smaller than sela_hip_decode_workspace_bytes()");
    hipEvent_t* ev = n_frames ? g_timing.events() : nullptr;
    g_timing.recorded = ev ? 2 : 0;
    const bool piped = n_frames && !ev && !g_phase_cycles && g_pipeline.ready();
    hipError_t e = sela::launch_decode(d_frames, d_frame_offsets, n_frames, channels, d_pcm_out, d_status, d_workspace,
        static_cast<hipStream_t>(stream), ev, g_phase_cycles, piped ? g_pipeline.side : nullptr, g_pipeline.fork, g_pipeline.parsed);
    if (e != hipSuccess)
        return fail_hip(e, "decode launch");
    return SELA_HIP_OK;
}

int sela_hip_encode(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel, uint8_t* frames_out,
    size_t frames_cap, uint64_t* frame_offsets_out)
{
    if (samples_per_channel != SELA_HIP_SAMPLES_PER_FRAME)
        return fail(SELA_HIP_EINVAL, "samples_per_channel must be 2048 (reference frame size)");
    if (channels == 0 || channels > 255 || !frame_offsets_out || (n_frames && (!pcm || !frames_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    int rc = sela_hip_init(-1);
    if (rc != SELA_HIP_OK)
        return rc;
    if (!g_ctx.bind_current_device())
        return fail(SELA_HIP_ENODEV, "hipGetDevice failed");
    frame_offsets_out[0] = 0;
    if (n_frames == 0)
        return SELA_HIP_OK;
    const uint32_t chunk = n_frames >= 2 * kHostChunkFramesEncode ? kHostChunkFramesEncode : n_frames;
    const uint32_t n_chunks = (n_frames + chunk - 1) / chunk;
    const int n_sets = n_chunks > 1 ? 2 : 1;
    const size_t frame_pcm = (size_t)sela::kBlock * channels * sizeof(int16_t);
    const size_t chunk_bound = sela_hip_encode_bound_bytes(chunk, channels);
    hipError_t e = g_ctx.streams();
    if (e != hipSuccess)
        return fail_hip(e, "hipStreamCreate");
    for (int b = 0; b < n_sets; b++)
        if ((e = g_ctx.pcm[b].reserve(chunk * frame_pcm + 4)) != hipSuccess || (e = g_ctx.frames[b].reserve(chunk_bound + 4)) != hipSuccess
            || (e = g_ctx.offsets[b].reserve(((size_t)chunk + 1) * 8)) != hipSuccess || (e = g_ctx.status[b].reserve(16)) != hipSuccess)
            return fail_hip(e, "hipMalloc");
    if ((e = g_ctx.workspace.reserve(sela::encode_workspace_bytes(chunk, channels))) != hipSuccess)
        return fail_hip(e, "hipMalloc");

    auto frames_of = [&](uint32_t i) { return i + 1 < n_chunks ? chunk : n_frames - i * chunk; };
    auto copy_in = [&](uint32_t i) -> hipError_t {
        const int b = (int)(i & 1);
        hipError_t err;
        if (i >= 2 && (err = hipStreamWaitEvent(g_ctx.s_in, g_ctx.ran[b], 0)) != hipSuccess) // chunk i-2 has read this buffer
            return err;
        if ((err = hipMemcpyAsync(g_ctx.pcm[b].ptr, pcm + (size_t)i * chunk * sela::kBlock * channels, frames_of(i) * frame_pcm,
                 hipMemcpyHostToDevice, g_ctx.s_in)) != hipSuccess)
            return err;
        return hipEventRecord(g_ctx.copied_in[b], g_ctx.s_in);
    };
    size_t base = 0; // bytes of the chunks already copied out
    std::vector<uint64_t> chunk_offsets((size_t)chunk + 1);
    auto copy_out = [&](uint32_t i) -> int {
        const int b = (int)(i & 1);
        const uint32_t nf = frames_of(i);
        hipError_t err;
        uint32_t status[4];
        if ((err = hipEventSynchronize(g_ctx.ran[b])) != hipSuccess
            || (err = hipMemcpyAsync(status, g_ctx.status[b].ptr, sizeof status, hipMemcpyDeviceToHost, g_ctx.s_out)) != hipSuccess
            || (err = hipMemcpyAsync(chunk_offsets.data(), g_ctx.offsets[b].ptr, ((size_t)nf + 1) * 8, hipMemcpyDeviceToHost, g_ctx.s_out)) != hipSuccess
            || (err = hipStreamSynchronize(g_ctx.s_out)) != hipSuccess)
            return fail_hip(err, "D2H status/offsets");
        if (flags_to_error(status[0])) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "a block left the range the .sela format can carry (flags 0x%x)", status[0]);
            return fail(SELA_HIP_ERANGE, msg);
        }
        const size_t total = (size_t)chunk_offsets[nf];
        if (status[1] || base + total > frames_cap)
            return fail(SELA_HIP_ECAPACITY, "frames_out too small (see sela_hip_encode_bound_bytes)");
        for (uint32_t f = 0; f <= nf; f++)
            frame_offsets_out[(size_t)i * chunk + f] = base + chunk_offsets[f];
        if (total && ((err = hipMemcpyAsync(frames_out + base, g_ctx.frames[b].ptr, total, hipMemcpyDeviceToHost, g_ctx.s_out)) != hipSuccess
                         || (err = hipStreamSynchronize(g_ctx.s_out)) != hipSuccess))
            return fail_hip(err, "D2H frames");
        base += total;
        return SELA_HIP_OK;
    };

    rc = SELA_HIP_OK;
    if ((e = copy_in(0)) != hipSuccess)
        rc = fail_hip(e, "H2D pcm");
    for (uint32_t i = 0; rc == SELA_HIP_OK && i <= n_chunks; i++) {
        if (i < n_chunks) {
            const int b = (int)(i & 1);
            if ((e = hipStreamWaitEvent(g_ctx.s_run, g_ctx.copied_in[b], 0)) != hipSuccess) {
                rc = fail_hip(e, "hipStreamWaitEvent");
                break;
            }
            rc = sela_hip_encode_device(static_cast<const int16_t*>(g_ctx.pcm[b].ptr), frames_of(i), channels,
                static_cast<uint8_t*>(g_ctx.frames[b].ptr), chunk_bound, static_cast<uint64_t*>(g_ctx.offsets[b].ptr),
                static_cast<uint32_t*>(g_ctx.status[b].ptr), g_ctx.workspace.ptr, g_ctx.workspace.cap, nullptr, g_ctx.s_run);
            if (rc != SELA_HIP_OK)
                break;
            if ((e = hipEventRecord(g_ctx.ran[b], g_ctx.s_run)) != hipSuccess || (i + 1 < n_chunks && (e = copy_in(i + 1)) != hipSuccess)) {
                rc = fail_hip(e, "H2D pcm");
                break;
            }
        }
        if (i >= 1)
            rc = copy_out(i - 1);
    }
    if (rc != SELA_HIP_OK) { // leave nothing in flight behind an error
        const std::string msg = sela_hip_last_error();
        (void)hipStreamSynchronize(g_ctx.s_in);
        (void)hipStreamSynchronize(g_ctx.s_run);
        (void)hipStreamSynchronize(g_ctx.s_out);
        return fail(rc, msg.c_str());
    }
    return SELA_HIP_OK;
}

int sela_hip_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out)
{
    if (channels == 0 || channels > 255 || (n_frames && (!frames || !frame_offsets || !pcm_out)))
        return fail(SELA_HIP_EINVAL, "bad argument");
    int rc = sela_hip_init(-1);
    if (rc != SELA_HIP_OK)
        return rc;
    if (n_frames == 0)
        return SELA_HIP_OK;
    for (uint32_t f = 0; f < n_frames; f++)
        if (frame_offsets[f + 1] < frame_offsets[f] || (frame_offsets[f] & 3))
            return fail(SELA_HIP_EFORMAT, "frame offsets must be ascending multiples of 4");
    if (!g_ctx.bind_current_device())
        return fail(SELA_HIP_ENODEV, "hipGetDevice failed");
    const uint32_t chunk = n_frames >= 2 * kHostChunkFramesDecode ? kHostChunkFramesDecode : n_frames;
    const uint32_t n_chunks = (n_frames + chunk - 1) / chunk;
    const int n_sets = n_chunks > 1 ? 2 : 1;
    const size_t frame_pcm = (size_t)sela::kBlock * channels * sizeof(int16_t);
    auto frames_of = [&](uint32_t i) { return i + 1 < n_chunks ? chunk : n_frames - i * chunk; };
    size_t max_bytes = 0;
    for (uint32_t i = 0; i < n_chunks; i++) {
        const size_t bytes = (size_t)(frame_offsets[(size_t)i * chunk + frames_of(i)] - frame_offsets[(size_t)i * chunk]);
        max_bytes = bytes > max_bytes ? bytes : max_bytes;
    }
    hipError_t e = g_ctx.streams();
    if (e != hipSuccess)
        return fail_hip(e, "hipStreamCreate");
    for (int b = 0; b < n_sets; b++)
        if ((e = g_ctx.pcm[b].reserve(chunk * frame_pcm + 4)) != hipSuccess || (e = g_ctx.frames[b].reserve(max_bytes + 8)) != hipSuccess
            || (e = g_ctx.offsets[b].reserve(((size_t)chunk + 1) * 8)) != hipSuccess || (e = g_ctx.status[b].reserve(16)) != hipSuccess)
            return fail_hip(e, "hipMalloc");
    if ((e = g_ctx.workspace.reserve(sela::decode_workspace_bytes(chunk, channels))) != hipSuccess)
        return fail_hip(e, "hipMalloc");

    auto copy_in = [&](uint32_t i) -> hipError_t {
        const int b = (int)(i & 1);
        const uint32_t nf = frames_of(i);
        const uint64_t* off = frame_offsets + (size_t)i * chunk;
        hipError_t err;
        if (i >= 2 && (err = hipStreamWaitEvent(g_ctx.s_in, g_ctx.ran[b], 0)) != hipSuccess) // chunk i-2 has read these buffers
            return err;
        std::vector<uint64_t>& rel = g_ctx.rebased[b];
        rel.resize((size_t)nf + 1);
        for (uint32_t f = 0; f <= nf; f++)
            rel[f] = off[f] - off[0];
        if ((err = hipMemcpyAsync(g_ctx.frames[b].ptr, frames + off[0], (size_t)rel[nf], hipMemcpyHostToDevice, g_ctx.s_in)) != hipSuccess
            || (err = hipMemcpyAsync(g_ctx.offsets[b].ptr, rel.data(), ((size_t)nf + 1) * 8, hipMemcpyHostToDevice, g_ctx.s_in)) != hipSuccess)
            return err;
        return hipEventRecord(g_ctx.copied_in[b], g_ctx.s_in);
    };
    uint32_t seen_flags = 0;
    auto copy_out = [&](uint32_t i) -> int {
        const int b = (int)(i & 1);
        hipError_t err;
        uint32_t status[4];
        if ((err = hipEventSynchronize(g_ctx.ran[b])) != hipSuccess
            || (err = hipMemcpyAsync(status, g_ctx.status[b].ptr, sizeof status, hipMemcpyDeviceToHost, g_ctx.s_out)) != hipSuccess
            || (err = hipMemcpyAsync(pcm_out + (size_t)i * chunk * sela::kBlock * channels, g_ctx.pcm[b].ptr, frames_of(i) * frame_pcm,
                    hipMemcpyDeviceToHost, g_ctx.s_out)) != hipSuccess
            || (err = hipStreamSynchronize(g_ctx.s_out)) != hipSuccess)
            return fail_hip(err, "D2H pcm");
        seen_flags |= status[0];
        return SELA_HIP_OK;
    };

    rc = SELA_HIP_OK;
    if ((e = copy_in(0)) != hipSuccess)
        rc = fail_hip(e, "H2D frames");
    for (uint32_t i = 0; rc == SELA_HIP_OK && i <= n_chunks; i++) {
        if (i < n_chunks) {
            const int b = (int)(i & 1);
            if ((e = hipStreamWaitEvent(g_ctx.s_run, g_ctx.copied_in[b], 0)) != hipSuccess) {
                rc = fail_hip(e, "hipStreamWaitEvent");
                break;
            }
            rc = sela_hip_decode_device(static_cast<const uint8_t*>(g_ctx.frames[b].ptr), static_cast<const uint64_t*>(g_ctx.offsets[b].ptr),
                frames_of(i), channels, static_cast<int16_t*>(g_ctx.pcm[b].ptr), static_cast<uint32_t*>(g_ctx.status[b].ptr),
                g_ctx.workspace.ptr, g_ctx.workspace.cap, g_ctx.s_run);
            if (rc != SELA_HIP_OK)
                break;
            if ((e = hipEventRecord(g_ctx.ran[b], g_ctx.s_run)) != hipSuccess || (i + 1 < n_chunks && (e = copy_in(i + 1)) != hipSuccess)) {
                rc = fail_hip(e, "H2D frames");
                break;
            }
        }
        if (i >= 1)
            rc = copy_out(i - 1);
    }
    if (rc != SELA_HIP_OK) { // leave nothing in flight behind an error
        const std::string msg = sela_hip_last_error();
        (void)hipStreamSynchronize(g_ctx.s_in);
        (void)hipStreamSynchronize(g_ctx.s_run);
        (void)hipStreamSynchronize(g_ctx.s_out);
        return fail(rc, msg.c_str());
    }
    // every frame is decoded (bad ones to silence) before the verdict, as in the one-piece call
    if (seen_flags & SELA_HIP_FLAG_BAD_FRAME)
        return fail(SELA_HIP_EFORMAT, "malformed frame stream (bad sync word or subframe header)");
    if (seen_flags & SELA_HIP_FLAG_RICE_OVERRUN)
        return fail(SELA_HIP_EFORMAT, "a Rice stream ended before all its values were read");
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
