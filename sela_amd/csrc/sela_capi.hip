// sela_capi.hip -- the extern "C" boundary of libsela_hip.so (declared in include/sela_hip.h).
//
// Plain pointers and sizes only.  The *_device entry points enqueue kernels on the caller's stream;
// the host-pointer entry points stage through a per-thread, grow-only set of device buffers.
// There is no CPU fallback anywhere: without a HIP device every call fails with SELA_HIP_ENODEV.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

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

struct HostContext {
    DeviceBuffer pcm, frames, offsets, status, workspace;
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
    void release()
    {
        pcm.release();
        frames.release();
        offsets.release();
        status.release();
        workspace.release();
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
    const size_t pcm_bytes = (size_t)n_frames * sela::kBlock * channels * sizeof(int16_t);
    const size_t bound = sela_hip_encode_bound_bytes(n_frames, channels);
    const size_t dev_cap = ((frames_cap < bound ? frames_cap : bound) + 3) & ~(size_t)3;
    hipError_t e;
    if ((e = g_ctx.pcm.reserve(pcm_bytes + 4)) != hipSuccess || (e = g_ctx.frames.reserve(dev_cap + 4)) != hipSuccess
        || (e = g_ctx.offsets.reserve(((size_t)n_frames + 1) * 8)) != hipSuccess || (e = g_ctx.status.reserve(16)) != hipSuccess
        || (e = g_ctx.workspace.reserve(sela::encode_workspace_bytes(n_frames, channels))) != hipSuccess)
        return fail_hip(e, "hipMalloc");
    if (pcm_bytes && (e = hipMemcpyAsync(g_ctx.pcm.ptr, pcm, pcm_bytes, hipMemcpyHostToDevice, nullptr)) != hipSuccess)
        return fail_hip(e, "H2D pcm");
    rc = sela_hip_encode_device(static_cast<const int16_t*>(g_ctx.pcm.ptr), n_frames, channels, static_cast<uint8_t*>(g_ctx.frames.ptr),
        frames_cap < bound ? (frames_cap & ~(size_t)3) : bound, static_cast<uint64_t*>(g_ctx.offsets.ptr),
        static_cast<uint32_t*>(g_ctx.status.ptr), g_ctx.workspace.ptr, g_ctx.workspace.cap, nullptr, nullptr);
    if (rc != SELA_HIP_OK)
        return rc;
    uint32_t status[4];
    if ((e = hipMemcpy(status, g_ctx.status.ptr, sizeof status, hipMemcpyDeviceToHost)) != hipSuccess)
        return fail_hip(e, "D2H status");
    if ((e = hipMemcpy(frame_offsets_out, g_ctx.offsets.ptr, ((size_t)n_frames + 1) * 8, hipMemcpyDeviceToHost)) != hipSuccess)
        return fail_hip(e, "D2H offsets");
    if (flags_to_error(status[0])) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "a block left the range the .sela format can carry (flags 0x%x)", status[0]);
        return fail(SELA_HIP_ERANGE, msg);
    }
    if (status[1])
        return fail(SELA_HIP_ECAPACITY, "frames_out too small (see sela_hip_encode_bound_bytes)");
    const size_t total = (size_t)frame_offsets_out[n_frames];
    if (total && (e = hipMemcpy(frames_out, g_ctx.frames.ptr, total, hipMemcpyDeviceToHost)) != hipSuccess)
        return fail_hip(e, "D2H frames");
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
    const size_t total = (size_t)frame_offsets[n_frames];
    const size_t pcm_bytes = (size_t)n_frames * sela::kBlock * channels * sizeof(int16_t);
    hipError_t e;
    if ((e = g_ctx.pcm.reserve(pcm_bytes + 4)) != hipSuccess || (e = g_ctx.frames.reserve(total + 8)) != hipSuccess
        || (e = g_ctx.offsets.reserve(((size_t)n_frames + 1) * 8)) != hipSuccess || (e = g_ctx.status.reserve(16)) != hipSuccess
        || (e = g_ctx.workspace.reserve(sela::decode_workspace_bytes(n_frames, channels))) != hipSuccess)
        return fail_hip(e, "hipMalloc");
    if ((e = hipMemcpyAsync(g_ctx.frames.ptr, frames, total, hipMemcpyHostToDevice, nullptr)) != hipSuccess
        || (e = hipMemcpyAsync(g_ctx.offsets.ptr, frame_offsets, ((size_t)n_frames + 1) * 8, hipMemcpyHostToDevice, nullptr)) != hipSuccess)
        return fail_hip(e, "H2D frames");
    rc = sela_hip_decode_device(static_cast<const uint8_t*>(g_ctx.frames.ptr), static_cast<const uint64_t*>(g_ctx.offsets.ptr), n_frames,
        channels, static_cast<int16_t*>(g_ctx.pcm.ptr), static_cast<uint32_t*>(g_ctx.status.ptr), g_ctx.workspace.ptr, g_ctx.workspace.cap,
        nullptr);
    if (rc != SELA_HIP_OK)
        return rc;
    uint32_t status[4];
    if ((e = hipMemcpy(status, g_ctx.status.ptr, sizeof status, hipMemcpyDeviceToHost)) != hipSuccess)
        return fail_hip(e, "D2H status");
    if ((e = hipMemcpy(pcm_out, g_ctx.pcm.ptr, pcm_bytes, hipMemcpyDeviceToHost)) != hipSuccess)
        return fail_hip(e, "D2H pcm");
    if (status[0] & SELA_HIP_FLAG_BAD_FRAME)
        return fail(SELA_HIP_EFORMAT, "malformed frame stream (bad sync word or subframe header)");
    if (status[0] & SELA_HIP_FLAG_RICE_OVERRUN)
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
