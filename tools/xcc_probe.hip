// tools/xcc_probe.hip -- which XCD does workgroup b of a 1-D grid run on?  Reads HW_REG_XCC_ID in 4096
// one-wave workgroups and compares with blockIdx.x % 8 (the mapping k_encode_blocks' XCD-aware block
// order and DESIGN.md 5.1 rely on for L2 sharing; the scalar-operand rings use the register itself).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip && tools/xcc_probe
// MI355X, ROCm 7.2: "blocks whose XCC_ID != blockIdx % 8: 0 of 4096".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k(unsigned* out)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0)
        out[blockIdx.x] = x;
}

int main()
{
    const int n = 4096;
    unsigned* d = nullptr;
    if (hipMalloc(&d, n * sizeof(unsigned)) != hipSuccess)
        return 1;
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    std::vector<unsigned> h(n);
    if (hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
        return 2;
    int mismatches = 0;
    for (int b = 0; b < n; b++)
        mismatches += (h[b] & 0xf) != (unsigned)(b % 8);
    std::printf("blocks whose XCC_ID != blockIdx %% 8: %d of %d\n", mismatches, n);
    return 0;
}
