// tools/pcie_probe.hip -- what the host link gives a kernel (loads from / stores to page-locked host memory) next to
// the copy engine, and whether stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) work here.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_probe tools/pcie_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        dst[i] = __builtin_nontemporal_load(src + i);
}
// the frame assembler's store shape: one aligned u32 per lane, a ~6 KB frame per workgroup
__global__ void k_copy_u32_frames(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n_frames, size_t frame_words)
{
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x)
        for (size_t w = threadIdx.x; w < frame_words; w += blockDim.x)
            dst[f * frame_words + w] = src[f * frame_words + w];
}
// in-order variant: the grid walks the buffer in slabs so that data arrives front to back
__global__ void k_copy_ordered(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, uint32_t* progress, size_t slab16)
{
    __shared__ uint32_t dummy;
    const size_t per_pass = (size_t)gridDim.x * blockDim.x;
    size_t done_slab = 0;
    for (size_t base = 0; base < n16; base += per_pass) {
        const size_t i = base + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n16)
            dst[i] = __builtin_nontemporal_load(src + i);
        (void)dummy; (void)done_slab; (void)progress; (void)slab16;
    }
}

static double ms_since(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }

int main()
{
    const size_t bytes = 31744000; // the bench track's PCM
    void *h = nullptr, *d = nullptr, *h2 = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h2, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d, bytes));
    void* d2 = nullptr;
    CK(hipMalloc(&d2, bytes));
    std::memset(h, 1, bytes);
    std::memset(h2, 0, bytes);
    void *hd = nullptr, *h2d = nullptr;
    CK(hipHostGetDevicePointer(&hd, h, 0));
    CK(hipHostGetDevicePointer(&h2d, h2, 0));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](auto&& fn, const char* what, double scale_bytes) {
        std::vector<double> t;
        for (int r = 0; r < 7; r++) {
            (void)hipDeviceSynchronize();
            const auto a = std::chrono::steady_clock::now();
            fn();
            (void)hipDeviceSynchronize();
            t.push_back(ms_since(a));
        }
        std::sort(t.begin(), t.end());
        std::printf("%-46s %8.3f ms  %6.1f GB/s\n", what, t[3], scale_bytes / t[3] / 1e6);
        return 0;
    };
    timed([&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); }, "copy engine H2D, one copy", (double)bytes);
    timed([&] { (void)hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, s); }, "copy engine D2H, one copy", (double)bytes);
    for (int parts : { 4, 8, 16 }) {
        char name[96];
        std::snprintf(name, sizeof name, "copy engine H2D, %d copies on one stream", parts);
        timed([&] { for (int p = 0; p < parts; p++) (void)hipMemcpyAsync((char*)d + bytes / parts * p, (char*)h + bytes / parts * p, bytes / parts, hipMemcpyHostToDevice, s); }, name, (double)bytes);
        std::snprintf(name, sizeof name, "copy engine H2D, %d copies alternating 2 streams", parts);
        timed([&] { for (int p = 0; p < parts; p++) (void)hipMemcpyAsync((char*)d + bytes / parts * p, (char*)h + bytes / parts * p, bytes / parts, hipMemcpyHostToDevice, (p & 1) ? s2 : s); }, name, (double)bytes);
        std::snprintf(name, sizeof name, "copy engine D2H, %d copies on one stream", parts);
        timed([&] { for (int p = 0; p < parts; p++) (void)hipMemcpyAsync((char*)h2 + bytes / parts * p, (char*)d + bytes / parts * p, bytes / parts, hipMemcpyDeviceToHost, s); }, name, (double)bytes);
    }
    for (int wgs : { 16, 32, 64, 128, 256, 512 }) {
        char name[96];
        std::snprintf(name, sizeof name, "kernel loads from host, %d x 256 threads", wgs);
        timed([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s, (const u32x4*)hd, (u32x4*)d, bytes / 16); }, name, (double)bytes);
    }
    for (int wgs : { 16, 32, 64, 128, 256, 512 }) {
        char name[96];
        std::snprintf(name, sizeof name, "kernel stores to host, %d x 256 threads", wgs);
        timed([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s, (const u32x4*)d, (u32x4*)h2d, bytes / 16); }, name, (double)bytes);
    }
    timed([&] {
        hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, (const u32x4*)hd, (u32x4*)d, bytes / 16);
        (void)hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, s2);
    }, "kernel loads (64 WG) + engine D2H together", 2.0 * bytes);
    timed([&] {
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, s2);
    }, "engine H2D + engine D2H together", 2.0 * bytes);
    timed([&] {
        hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, (const u32x4*)hd, (u32x4*)d, bytes / 16);
        hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, s2, (const u32x4*)d2, (u32x4*)h2d, bytes / 16);
    }, "kernel loads (64 WG) + kernel stores (128 WG)", 2.0 * bytes);
    timed([&] {
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, s2, (const u32x4*)d2, (u32x4*)h2d, bytes / 16);
    }, "engine H2D + kernel stores (128 WG)", 2.0 * bytes);
    timed([&] {
        hipLaunchKernelGGL(k_copy, dim3(32), dim3(256), 0, s, (const u32x4*)hd, (u32x4*)d, bytes / 16);
        (void)hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, s2);
    }, "kernel loads (32 WG) + engine D2H (other buffer)", 2.0 * bytes);
    timed([&] {
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, s2);
    }, "engine H2D + engine D2H (other buffer)", 2.0 * bytes);
    for (int threads : { 64, 256 }) {
        char name[96];
        const size_t fw = 1470, nfr = 22800000 / (fw * 4);
        std::snprintf(name, sizeof name, "u32 stores to host, 1 frame per WG of %d", threads);
        timed([&] { hipLaunchKernelGGL(k_copy_u32_frames, dim3((unsigned)nfr), dim3(threads), 0, s, (const uint32_t*)d2, (uint32_t*)h2d, nfr, fw); }, name, (double)nfr * fw * 4);
        std::snprintf(name, sizeof name, "u32 stores to device, 1 frame per WG of %d", threads);
        timed([&] { hipLaunchKernelGGL(k_copy_u32_frames, dim3((unsigned)nfr), dim3(threads), 0, s, (const uint32_t*)d2, (uint32_t*)d, nfr, fw); }, name, (double)nfr * fw * 4);
    }
    timed([&] {
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        const size_t fw = 1470, nfr = 22800000 / (fw * 4);
        hipLaunchKernelGGL(k_copy_u32_frames, dim3((unsigned)nfr), dim3(256), 0, s2, (const uint32_t*)d2, (uint32_t*)h2d, nfr, fw);
    }, "engine H2D 31.7 MB + u32 stores to host 22.8 MB", (double)bytes + 22.8e6);
    // ---- stream memory operations ----------------------------------------------------------------------------------
    uint32_t* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 256, hipMallocSignalMemory);
    std::printf("hipExtMallocWithFlags(signal): %s\n", hipGetErrorString(e));
    uint32_t* plain = nullptr;
    CK(hipMalloc((void**)&plain, 256));
    CK(hipMemset(plain, 0, 256));
    e = hipStreamWriteValue32(s, plain, 7, 0);
    std::printf("hipStreamWriteValue32(plain device memory): %s\n", hipGetErrorString(e));
    (void)hipStreamSynchronize(s);
    uint32_t v = 0;
    (void)hipMemcpy(&v, plain, 4, hipMemcpyDeviceToHost);
    std::printf("  value read back: %u\n", v);
    if (sig) {
        (void)hipMemset(sig, 0, 256);
        e = hipStreamWriteValue32(s, sig, 9, 0);
        std::printf("hipStreamWriteValue32(signal memory): %s\n", hipGetErrorString(e));
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&v, sig, 4, hipMemcpyDeviceToHost);
        std::printf("  value read back: %u\n", v);
        e = hipStreamWaitValue32(s2, sig, 9, hipStreamWaitValueGte, 0xFFFFFFFFu);
        std::printf("hipStreamWaitValue32(signal memory, already satisfied): %s\n", hipGetErrorString(e));
        e = hipStreamSynchronize(s2);
        std::printf("  sync: %s\n", hipGetErrorString(e));
    }
    e = hipStreamWaitValue32(s2, plain, 7, hipStreamWaitValueGte, 0xFFFFFFFFu);
    std::printf("hipStreamWaitValue32(plain memory, already satisfied): %s\n", hipGetErrorString(e));
    e = hipStreamSynchronize(s2);
    std::printf("  sync: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    // copies with a write-value between them: does the marker cost a gap?
    timed([&] { for (int p = 0; p < 8; p++) { (void)hipMemcpyAsync((char*)d + bytes / 8 * p, (char*)h + bytes / 8 * p, bytes / 8, hipMemcpyHostToDevice, s); (void)hipStreamWriteValue32(s, plain, (uint32_t)p + 1, 0); } },
        "engine H2D, 8 copies each + write-value", (double)bytes);
    return 0;
}
