// tools/mfma_overlap.hip -- does the FP64 matrix pipe of a gfx950 SIMD run beside its vector pipe?  (the gate
// for moving the encoder's residue filter onto v_mfma_f64_16x16x4_f64.)  One workgroup of 8 waves = two waves per SIMD (waves
// w and w + 4 share SIMD w % 4).  Every wave runs `iter` rounds of 16 instructions of its kind and reports its cycles:
//   VV  both waves of a SIMD issue v_fma_f64           MM  both issue v_mfma_f64_16x16x4_f64
//   VM  the first wave of each SIMD issues v_fma_f64, the second v_mfma_f64_16x16x4_f64
//   V-, M-  one wave per SIMD (the other idles)
// If the pipes overlap, VM's waves take what they take in V- / M-; if they share the issue, each takes longer.
// Not product code; the numbers are quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int iter, int kind_first, int kind_second /* 0 idle, 1 v_fma_f64, 2 mfma f64, 3 v_mul/v_add pairs */)
{
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const int kind = wave < 4 ? kind_first : kind_second;
    double x = 1.0 + lane, y = 1.0000001, z = 0.5, w = 2.0, m = 0.9999999;
    double x2 = x, y2 = y, z2 = z, w2 = w; // (eight independent chains: a dependent v_fma_f64 comes back after ~8 cycles)
    uint32_t ia = lane, ib = 3, ic = 5, id = 7;
    f64x4 acc0 = { 0, 0, 0, 0 }, acc1 = { 0, 0, 0, 0 }, acc2 = { 0, 0, 0, 0 }, acc3 = { 0, 0, 0, 0 };
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < iter; i++) {
        if (kind == 1)
            asm volatile(REP4("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n") REP4("v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(x2), "+v"(y2), "+v"(z2), "+v"(w2) : "v"(m));
        if (kind == 3)
            asm volatile(REP4("v_mul_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n") REP4("v_mul_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(x2), "+v"(y2), "+v"(z2), "+v"(w2) : "v"(m));
        if (kind == 4) // 32-bit integer work (what the Rice stages issue)
            asm volatile(REP4("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_lshrrev_b32 %2, 1, %2\n v_add_u32 %3, %3, %4\n") REP4("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_lshrrev_b32 %2, 1, %2\n v_add_u32 %3, %3, %4\n")
                         : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(lane));
        if (kind == 2)
            asm volatile(REP4("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(x), "v"(y));
    }
    const long long t1 = clock64();
    out[threadIdx.x] = x + y + z + w + x2 + y2 + z2 + w2 + acc0[0] + acc1[1] + acc2[2] + acc3[3] + (double)(ia + ib + ic + id);
    if (lane == 0)
        cyc[wave] = t1 - t0;
}

int main()
{
    double* out; long long* cyc;
    if (hipMalloc(&out, 512 * 8) != hipSuccess || hipMalloc(&cyc, 8 * 8) != hipSuccess) return 1;
    const int iter = 4000;
    struct { const char* name; int a, b; } modes[] = {
        {"V-  v_fma_f64 alone on its SIMD", 1, 0}, {"M-  mfma_f64_16x16x4 alone on its SIMD", 2, 0}, {"VV  v_fma_f64 beside v_fma_f64", 1, 1},
        {"MM  mfma beside mfma", 2, 2}, {"VM  v_fma_f64 beside mfma", 1, 2}, {"U-  v_mul_f64 / v_add_f64 alone", 3, 0}, {"UM  v_mul/v_add beside mfma", 3, 2},
        {"I-  32-bit integer VALU alone", 4, 0}, {"IM  32-bit integer VALU beside mfma", 4, 2},
    };
    printf("%-44s %16s %16s   (cycles per instruction of the wave's own kind; 16 matrix / 32 vector instructions per round, %d rounds)\n", "mode", "first wave", "second wave", iter);
    for (auto& m : modes) {
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, 10, m.a, m.b);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, iter, m.a, m.b);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        long long h[8];
        (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        long long a = 0, b = 0;
        for (int w = 0; w < 4; w++) a = h[w] > a ? h[w] : a, b = h[w + 4] > b ? h[w + 4] : b;
        const int per_a = m.a == 2 ? 16 : 32, per_b = m.b == 2 ? 16 : 32; // (the vector kinds issue 32 instructions per round)
        printf("%-44s %16.2f %16.2f\n", m.name, m.a ? (double)a / iter / per_a : 0.0, m.b ? (double)b / iter / per_b : 0.0);
    }
    return 0;
}
