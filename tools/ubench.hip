// tools/ubench.hip -- asm-level issue-cost micro-benchmarks (cycles per wave-instruction) on gfx950.
// Not product code; numbers are quoted in DESIGN.md.  Every body is 16 copies of one instruction
// (independent destinations unless noted) inside a counted loop; s_memtime brackets the loop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ void k(uint32_t* out, long long* cyc, int iter)
{
    __shared__ uint32_t lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = lane * 747796405u + 1, b = lane ^ 0x5bd1e995u, c = 12345u + lane, d = 99u;
    uint64_t p = lane, q = 7, r = 11, s = 13;
    double x = 1.0 + lane, y = 1.0000001, z = 0.5, w = 2.0;
    uint32_t addr = (uint32_t)(lane * 8);
    int sg = 3;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < iter; i++) {
        if (KIND == 0) asm volatile(REP4("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3\n") : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
        if (KIND == 1) asm volatile(REP16("v_mad_u64_u32 %0, vcc, %1, %2, %0\n") : "+v"(p) : "v"(a), "v"(b) : "vcc"); // dependent
        if (KIND == 2) asm volatile(REP4("v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %5, %6, %1\n v_mad_i32_i24 %2, %6, %7, %2\n v_mad_i32_i24 %3, %7, %4, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(d));
        if (KIND == 3) asm volatile(REP4("v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %5, %6\n v_mul_lo_u32 %2, %6, %7\n v_mul_lo_u32 %3, %7, %4\n") : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(d));
        if (KIND == 4) asm volatile(REP4("v_add_u32 %0, %4, %5\n v_add_u32 %1, %5, %6\n v_add_u32 %2, %6, %7\n v_add_u32 %3, %7, %4\n") : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(d));
        if (KIND == 5) asm volatile(REP4("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 6) asm volatile(REP16("v_readlane_b32 %0, %1, 5\n") : "=s"(sg) : "v"(a));
        if (KIND == 7) asm volatile(REP16("v_readfirstlane_b32 %0, %1\n") : "=s"(sg) : "v"(a));
        if (KIND == 8) asm volatile(REP16("v_writelane_b32 %0, %1, 7\n") : "+v"(a) : "s"(sg));
        if (KIND == 9) asm volatile(REP4("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(y));
        if (KIND == 10) asm volatile(REP4("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(y));
        if (KIND == 11) asm volatile(REP16("v_add_f64 %0, %0, %1\n") : "+v"(x) : "v"(y)); // dependent
        if (KIND == 12) asm volatile(REP4("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(y));
        if (KIND == 13) asm volatile(REP4("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(p), "=v"(q), "=v"(r), "=v"(s) : "v"(addr));
        if (KIND == 14) asm volatile(REP4("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:512\n ds_read_b32 %2, %4 offset:1024\n ds_read_b32 %3, %4 offset:1536\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr));
        if (KIND == 15) asm volatile(REP16("s_sub_u32 %0, %0, 3\n") : "+s"(sg) : : "scc");
        if (KIND == 16) // IIR inner step candidate: 2 readfirstlane, SALU pred, 2 mad with SGPR, 2 dpp
            asm volatile(REP4(
                "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n s_sub_u32 s20, 0, s20\n s_subb_u32 s21, 4, s21\n s_ashr_i32 s21, s21, 3\n s_sub_i32 s21, s22, s21\n"
                "v_mov_b32_dpp %4, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                "v_mad_u64_u32 %2, vcc, %6, s21, %2\n v_mad_u64_u32 %3, vcc, %7, s21, %3\n"
                "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n") : "+v"(a), "+v"(b), "+v"(p), "+v"(q), "+v"(c), "+v"(d) : "v"(lane), "v"(addr) : "vcc", "scc", "s20", "s21", "s22");
        if (KIND == 17) asm volatile(REP16("v_add_co_u32 %0, vcc, %0, %1\n") : "+v"(a) : "v"(b) : "vcc"); // dependent 32-bit add
        if (KIND == 18) asm volatile(REP4("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %5, %6, %1\n v_mad_u32_u24 %2, %6, %7, %2\n v_mad_u32_u24 %3, %7, %4, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(d));
        if (KIND == 19) asm volatile(REP4("v_mul_hi_u32 %0, %4, %5\n v_mul_hi_u32 %1, %5, %6\n v_mul_hi_u32 %2, %6, %7\n v_mul_hi_u32 %3, %7, %4\n") : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(d));
        if (KIND == 20) asm volatile(REP4("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr));
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)(p + q + r + s) + (uint32_t)(x + y + z + w) + sg;
    if (lane == 0)
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

struct Case { const char* name; void (*fn)(uint32_t*, long long*, int); int per_iter; };

int main(int argc, char** argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    uint32_t* out; long long* cyc;
    if (hipMalloc(&out, 4096 * 4) != hipSuccess || hipMalloc(&cyc, 64 * 8) != hipSuccess) return 1;
    const Case cases[] = {
        {"v_mad_u64_u32 independent", k<0>, 16}, {"v_mad_u64_u32 dependent", k<1>, 16}, {"v_mad_i32_i24", k<2>, 16},
        {"v_mul_lo_u32", k<3>, 16}, {"v_add_u32", k<4>, 16}, {"v_mov_b32_dpp wave_shr:1", k<5>, 16},
        {"v_readlane_b32", k<6>, 16}, {"v_readfirstlane_b32", k<7>, 16}, {"v_writelane_b32", k<8>, 16},
        {"v_mul_f64 independent", k<9>, 16}, {"v_add_f64 independent", k<10>, 16}, {"v_add_f64 dependent", k<11>, 16},
        {"v_fma_f64 independent", k<12>, 16}, {"ds_read_b64 (+wait per 16)", k<13>, 16}, {"ds_read_b32 (+wait per 16)", k<14>, 16},
        {"s_sub_u32 dependent", k<15>, 16}, {"IIR step candidate (12 instr/step)", k<16>, 4},
        {"v_add_co_u32 dependent", k<17>, 16}, {"v_mad_u32_u24", k<18>, 16}, {"v_mul_hi_u32", k<19>, 16},
        {"ds_bpermute_b32 (+wait per 16)", k<20>, 16},
    };
    printf("%-40s %12s %12s %12s   (cycles per instruction; waves on one SIMD share issue)\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    int idx = -1;
    for (const Case& c : cases) {
        idx++;
        if (only >= 0 && idx != only) continue;
        double res[3]; int wi = 0;
        for (int threads : {256, 512, 1024}) {
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, cyc, 10);
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, cyc, 2000);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            std::vector<long long> h(threads / 64);
            (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            long long mx = 0; for (long long v : h) mx = v > mx ? v : mx;
            res[wi++] = (double)mx / 2000 / c.per_iter / (threads / 256);
        }
        printf("%-40s %12.2f %12.2f %12.2f\n", c.name, res[0], res[1], res[2]);
        fflush(stdout);
    }
    return 0;
}
