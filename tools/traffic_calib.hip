// tools/traffic_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN number of bytes, per access width.
//
// MI355X_MICROARCH.md ("HBM"): on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B per lane) coalesced streaming
// read; other widths and WRITE_SIZE are uncalibrated -- "calibrate on a known byte count in your own access pattern".  The
// encoder's block kernels load 4 and 8 bytes per lane (k_encode_blocks: u32 per lane; k_encode_teams: 8 / 16), the workers
// and the assembler 16.  Each kernel below streams over a buffer far larger than the 256 MB Infinity Cache exactly once:
//     read_b32 / read_b64 / read_b128     every lane loads 4 / 8 / 16 bytes per instruction, a wave's lanes side by side
//     write_b32 / write_b64 / write_b128  the same for stores
// tools/traffic_calib.sh runs this under rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in passes of their own) and writes the
// factors (true bytes / reported bytes) to profiles/<round>/traffic_calibration.json; tools/collect_profiles.sh applies them.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/traffic_calib tools/traffic_calib.hip ; tools/traffic_calib [MiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, size_t n, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++)
            acc += w[k];
    }
    if (acc == 0x12345678u) // (never: keeps the loads)
        sink[0] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++)
            w[k] = seed + (uint32_t)i + k;
        dst[i] = v;
    }
}


int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)std::atoll(argv[1]) : 1024;
    const size_t bytes = mib << 20;
    void *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k_read<uint32_t>), grid, block, 0, 0, static_cast<const uint32_t*>(buf), bytes / 4, static_cast<uint32_t*>(sink));
        hipLaunchKernelGGL((k_read<uint2>), grid, block, 0, 0, static_cast<const uint2*>(buf), bytes / 8, static_cast<uint32_t*>(sink));
        hipLaunchKernelGGL((k_read<uint4>), grid, block, 0, 0, static_cast<const uint4*>(buf), bytes / 16, static_cast<uint32_t*>(sink));
        hipLaunchKernelGGL((k_write<uint32_t>), grid, block, 0, 0, static_cast<uint32_t*>(buf), bytes / 4, (uint32_t)rep);
        hipLaunchKernelGGL((k_write<uint2>), grid, block, 0, 0, static_cast<uint2*>(buf), bytes / 8, (uint32_t)rep);
        hipLaunchKernelGGL((k_write<uint4>), grid, block, 0, 0, static_cast<uint4*>(buf), bytes / 16, (uint32_t)rep);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("bytes_per_kernel %zu\n", bytes);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
