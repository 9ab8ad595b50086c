// tools/ubench.hip -- instruction-rate micro-benchmarks that size the SELA kernels (not product code).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o /tmp/ubench && /tmp/ubench
//
// Each kernel runs an unrolled body of one instruction pattern for ITER iterations and reports
// shader cycles per wave-instruction (s_memtime), for 1 wave per SIMD and 4 waves per SIMD
// (one 256- or 1024-thread workgroup on one CU).  Numbers are quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));             \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

constexpr int ITER = 2000;

template <int KIND>
__global__ void k(double* out, long long* cyc, int iter)
{
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a0 = 1.0 + lane * 1e-9, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
    const double m = 1.0000001;
    uint64_t u0 = lane, u1 = 3, u2 = 5, u3 = 7;
    uint32_t w0 = lane * 2654435761u, w1 = 12345, w2 = 777, w3 = 99;
    int ops = 0;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < iter; i++) {
        if (KIND == 0) { // 8 independent v_mul_f64 + v_add_f64 pairs
            a0 += m * a0; a1 += m * a1; a2 += m * a2; a3 += m * a3; a4 += m * a4; a5 += m * a5; a6 += m * a6; a7 += m * a7;
        } else if (KIND == 1) { // dependent v_add_f64 chain
            a0 += m; a0 += m; a0 += m; a0 += m; a0 += m; a0 += m; a0 += m; a0 += m;
        } else if (KIND == 2) { // 4 independent v_mad_u64_u32
            u0 += (uint64_t)w0 * w1; u1 += (uint64_t)w1 * w2; u2 += (uint64_t)w2 * w3; u3 += (uint64_t)w3 * w0;
            u0 += (uint64_t)w0 * w2; u1 += (uint64_t)w1 * w3; u2 += (uint64_t)w2 * w0; u3 += (uint64_t)w3 * w1;
        } else if (KIND == 3) { // v_mul_lo_u32 (+ add)
            w0 = w0 * w1 + 1; w1 = w1 * w2 + 1; w2 = w2 * w3 + 1; w3 = w3 * w0 + 1;
            w0 = w0 * w2 + 1; w1 = w1 * w3 + 1; w2 = w2 * w0 + 1; w3 = w3 * w1 + 1;
        } else if (KIND == 4) { // v_mad_i32_i24
            w0 = __mul24((int)w0 >> 8, (int)w1 >> 8) + w2; w1 = __mul24((int)w1 >> 8, (int)w2 >> 8) + w3;
            w2 = __mul24((int)w2 >> 8, (int)w3 >> 8) + w0; w3 = __mul24((int)w3 >> 8, (int)w0 >> 8) + w1;
            w0 = __mul24((int)w0 >> 8, (int)w2 >> 8) + w1; w1 = __mul24((int)w1 >> 8, (int)w3 >> 8) + w2;
            w2 = __mul24((int)w2 >> 8, (int)w0 >> 8) + w3; w3 = __mul24((int)w3 >> 8, (int)w1 >> 8) + w0;
        } else if (KIND == 5) { // 8 x (2 v_mov_b32_dpp wave_shr:1)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                w0 = __builtin_amdgcn_update_dpp(w0, w1, 0x138, 0xf, 0xf, false);
                w1 = __builtin_amdgcn_update_dpp(w1, w0, 0x138, 0xf, 0xf, false);
            }
        } else if (KIND == 6) { // 8 x v_readfirstlane + dependent VALU use
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t s = __builtin_amdgcn_readfirstlane(w0);
                w0 = w0 + s + lane;
            }
        } else if (KIND == 7) { // 8 per-lane ds_read_b64 (conflict-free) feeding adds
            const double* p = lds + ((lane + i) & 1023);
            a0 += p[0]; a1 += p[64]; a2 += p[128]; a3 += p[192]; a4 += p[256]; a5 += p[320]; a6 += p[384]; a7 += p[448];
        } else if (KIND == 8) { // the autocorrelation step: readfirstlane x2, 2 mul, 2 add, 1 ds_read_b64
            const double* p = lds + (i & 1023);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double A = p[r * 64 + 63 - lane];
                const uint64_t x = __builtin_bit_cast(uint64_t, A);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
                const double cj = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
                a0 += cj * A;
                a1 += cj * a2;
                a2 = A;
            }
        } else if (KIND == 9) { // dependent chain: v_mad_u64_u32 -> v_readfirstlane -> ... (IIR critical path)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t s = __builtin_amdgcn_readfirstlane((uint32_t)(u0 >> 35));
                u0 = u1 + (uint64_t)w1 * s;
                u1 = u2 + (uint64_t)w2 * s;
            }
        } else if (KIND == 10) { // v_fma_f64, 8 independent
            a0 = __builtin_fma(a0, m, m); a1 = __builtin_fma(a1, m, m); a2 = __builtin_fma(a2, m, m); a3 = __builtin_fma(a3, m, m);
            a4 = __builtin_fma(a4, m, m); a5 = __builtin_fma(a5, m, m); a6 = __builtin_fma(a6, m, m); a7 = __builtin_fma(a7, m, m);
        } else if (KIND == 11) { // f64 division, 2 independent
            a0 = a1 / a0; a2 = a3 / a2;
        }
    }
    const long long t1 = clock64();
    (void)ops;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 + u1 + u2 + u3) + (double)(w0 + w1 + w2 + w3);
    if (lane == 0)
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

struct Case {
    const char* name;
    void (*fn)(double*, long long*, int);
    int instr_per_iter;
};

int main()
{
    double* out;
    long long* cyc;
    CHECK(hipMalloc(&out, 2048 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 64 * sizeof(long long)));
    const Case cases[] = {
        { "8x (v_mul_f64 + v_add_f64) independent        [16 instr]", k<0>, 16 },
        { "8x v_add_f64 dependent chain                  [ 8 instr]", k<1>, 8 },
        { "8x v_mad_u64_u32 independent                  [ 8 instr]", k<2>, 8 },
        { "8x (v_mul_lo_u32 + add)                       [16 instr]", k<3>, 16 },
        { "8x v_mad_i32_i24 (+shifts)                    [ 8 mads ]", k<4>, 8 },
        { "16x v_mov_b32_dpp wave_shr:1                  [16 instr]", k<5>, 16 },
        { "8x (v_readfirstlane + 2 add)                  [ 8 steps]", k<6>, 8 },
        { "8x ds_read_b64 per-lane + v_add_f64           [ 8 steps]", k<7>, 8 },
        { "8x autocorr step (ds_read,2 rfl,2 mul,2 add)  [ 8 steps]", k<8>, 8 },
        { "8x IIR step core (rfl, 2 mad_u64 dependent)   [ 8 steps]", k<9>, 8 },
        { "8x v_fma_f64 independent                      [ 8 instr]", k<10>, 8 },
        { "2x f64 division                               [ 2 divs ]", k<11>, 2 },
    };
    printf("%-62s %12s %12s %12s\n", "pattern (cycles per listed unit)", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (const Case& c : cases) {
        double r[3];
        int wi = 0;
        for (int threads : { 256, 512, 1024 }) {
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, cyc, 10); // warm-up
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, cyc, ITER);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> h(threads / 64);
            CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            long long mx = 0;
            for (long long v : h)
                mx = v > mx ? v : mx;
            // cycles per unit per WAVE SLOT: waves on one SIMD share it, so divide by waves per SIMD to get issue cost
            r[wi++] = (double)mx / ITER / c.instr_per_iter / (threads / 256);
        }
        printf("%-62s %12.2f %12.2f %12.2f\n", c.name, r[0], r[1], r[2]);
    }
    return 0;
}
