#!/usr/bin/env python3
"""How fast can one wave walk the decoder's synthesis recurrence?  (Not product code; DESIGN.md 9/10 quote it.)

Generates, compiles and runs a micro-benchmark whose variants are written instruction by instruction (one asm block per 64
samples, physical registers), 2048 dependent samples per wave like k_decode_frames' synthesis:
  A   the product's step: ds_read_b64 (table prefetch), v_readlane -> s_ashr -> v_mad_i64_i32 -> v_mad_i32_i24
  A2  the shift on the vector side: v_ashrrev -> v_readlane -> the two multiply-adds
  B   the recurrence on the scalar unit: m' = (Q + a1 * m) >> 35 from s_mul_i32 / s_mul_hi_i32 / s_add_u32 / s_addc_u32,
      Q (the sum of the sample after next, 64 bits) read with two v_readlanes one step ahead; the vector multiply-adds only
      feed later samples and sit in the shadow of the scalar chain
  B2  B without the vector work (the scalar chain alone)
  AR2 A on the ring of 128 (orders above 60): two accumulator registers per lane, 2 x (v_mad_i64_i32 + v_mad_i32_i24) per step
  FR2 the FP64 form of AR2: sums as doubles, one v_fma_f64 per register with the multiplier floor(N 2^-35)
      as a scalar operand pair -- v_mul_f64 + v_floor_f64 + two v_readlanes in the chain;  FR1 the same on the ring of 64
Run on the GPU box:  python tools/chain_ubench.py  (prints cycles per sample against waves per SIMD)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "tmp")

# registers inside the asm blocks: v[40:41] accumulator (lo, hi), v[42:43] / v[44:45] table values in flight, v46 temp
# s20 = m, s[22:23] = Q of the next sample, s[24:25] = Q of the one after, s26.. temporaries, s30/s31 = a1 (lo, hi)


def step_a(m):
    pf = "v[42:43]" if m % 2 == 0 else "v[44:45]"
    lo, hi = ("v42", "v43") if m % 2 == 0 else ("v44", "v45")
    return (f"v_readlane_b32 s14, v41, {m}\n s_ashr_i32 s20, s14, 3\n s_waitcnt lgkmcnt(1)\n"
            f"v_mad_i64_i32 v[40:41], s[14:15], {lo}, s20, v[40:41]\n v_mad_i32_i24 v41, {hi}, s20, v41\n"
            f"ds_read_b64 {pf}, v47 offset:{8 * ((m + 2) % 64)}\n")


def step_a2(m):
    pf = "v[42:43]" if m % 2 == 0 else "v[44:45]"
    lo, hi = ("v42", "v43") if m % 2 == 0 else ("v44", "v45")
    return (f"v_ashrrev_i32 v46, 3, v41\n v_readlane_b32 s20, v46, {m}\n s_waitcnt lgkmcnt(1)\n"
            f"v_mad_i64_i32 v[40:41], s[14:15], {lo}, s20, v[40:41]\n v_mad_i32_i24 v41, {hi}, s20, v41\n"
            f"ds_read_b64 {pf}, v47 offset:{8 * ((m + 2) % 64)}\n")


def step_b(m, vector=True):
    pf = "v[42:43]" if m % 2 == 0 else "v[44:45]"
    lo, hi = ("v42", "v43") if m % 2 == 0 else ("v44", "v45")
    q_now, q_next = ("s22", "s23"), ("s24", "s25")
    if m % 2:
        q_now, q_next = q_next, q_now
    lane = (m + 2) % 64
    v = vector
    return "".join([
        f"s_mul_i32 s26, s30, s20\n",
        f"s_waitcnt lgkmcnt(1)\n v_mad_i64_i32 v[40:41], s[14:15], {lo}, s20, v[40:41]\n" if v else "",
        f"s_mul_hi_i32 s27, s30, s20\n",
        f"s_mul_i32 s28, s31, s20\n",
        f"s_add_u32 s26, {q_now[0]}, s26\n",
        f"v_mad_i32_i24 v41, {hi}, s20, v41\n" if v else "",
        f"s_addc_u32 s27, {q_now[1]}, s27\n",
        f"s_add_u32 s27, s27, s28\n",
        f"v_readlane_b32 {q_next[0]}, v40, {lane}\n" if v else "",
        f"s_ashr_i32 s20, s27, 3\n",
        f"v_readlane_b32 {q_next[1]}, v41, {lane}\n" if v else "",
        f"ds_read_b64 {pf}, v47 offset:{8 * ((m + 2) % 64)}\n" if v else "",
    ])


def step_a_r2(m):
    """the product's step on the ring of 128 (orders above 60): two accumulator registers, two table reads"""
    pf, pf2 = ("v[42:43]", "v[50:51]") if m % 2 == 0 else ("v[44:45]", "v[52:53]")
    lo, hi, lo2, hi2 = ("v42", "v43", "v50", "v51") if m % 2 == 0 else ("v44", "v45", "v52", "v53")
    return (f"v_readlane_b32 s14, v41, {m}\n s_ashr_i32 s20, s14, 3\n s_waitcnt lgkmcnt(2)\n"
            f"v_mad_i64_i32 v[40:41], s[14:15], {lo}, s20, v[40:41]\n v_mad_i32_i24 v41, {hi}, s20, v41\n"
            f"v_mad_i64_i32 v[48:49], s[14:15], {lo2}, s20, v[48:49]\n v_mad_i32_i24 v49, {hi2}, s20, v49\n"
            f"ds_read_b64 {pf}, v47 offset:{8 * ((m + 2) % 64)}\n ds_read_b64 {pf2}, v47 offset:{256 + 8 * ((m + 2) % 64)}\n")


def step_f(m, regs):
    """the FP64 form VERDICT r4 asks to price: the sums as doubles (exact while below 2^53), one v_fma_f64 per register with the
    multiplier -- floor(N 2^-35), a double -- as a scalar operand pair; the floor costs a multiply and a v_floor_f64 on the vector
    side (the scalar unit has no FP64) and the multiplier crosses with two v_readlanes"""
    pf, pf2 = ("v[42:43]", "v[50:51]") if m % 2 == 0 else ("v[44:45]", "v[52:53]")
    text = (f"v_mul_f64 v[54:55], v[40:41], s[36:37]\n v_floor_f64 v[54:55], v[54:55]\n v_readlane_b32 s20, v54, {m}\n v_readlane_b32 s21, v55, {m}\n"
            f"s_waitcnt lgkmcnt({regs})\n v_fma_f64 v[40:41], {pf}, s[20:21], v[40:41]\n")
    if regs == 2:
        text += f"v_fma_f64 v[48:49], {pf2}, s[20:21], v[48:49]\n"
    text += f"ds_read_b64 {pf}, v47 offset:{8 * ((m + 2) % 64)}\n"
    if regs == 2:
        text += f"ds_read_b64 {pf2}, v47 offset:{256 + 8 * ((m + 2) % 64)}\n"
    return text


def block(kind):
    body = {"A": step_a, "A2": step_a2, "B": step_b, "B2": lambda m: step_b(m, False), "AR2": step_a_r2, "FR2": lambda m: step_f(m, 2),
            "FR1": lambda m: step_f(m, 1)}[kind]
    if kind in ("AR2", "FR2", "FR1"):
        text = ("v_mov_b32 v40, %[cl]\n v_mov_b32 v41, %[ch]\n v_mov_b32 v48, %[cl]\n v_mov_b32 v49, %[ch]\n v_mov_b32 v47, %[addr]\n s_mov_b32 s20, %[m]\n s_mov_b32 s21, 0x3f100000\n"
                "s_mov_b32 s36, 0\n s_mov_b32 s37, 0x3dc00000\n"
                "ds_read_b64 v[42:43], v47\n ds_read_b64 v[44:45], v47 offset:8\n ds_read_b64 v[50:51], v47 offset:256\n ds_read_b64 v[52:53], v47 offset:264\n")
        text += "".join(body(m) for m in range(64))
        text += "s_waitcnt lgkmcnt(0)\n v_add_u32 %[cl], v40, v48\n v_add_u32 %[ch], v41, v49\n s_mov_b32 %[m], s20\n"
        return text.replace("\n", "\\n")
    text = ("v_mov_b32 v40, %[cl]\n v_mov_b32 v41, %[ch]\n v_mov_b32 v47, %[addr]\n s_mov_b32 s20, %[m]\n s_mov_b32 s30, %[a1l]\n s_mov_b32 s31, %[a1h]\n"
            "s_mov_b32 s22, 17\n s_mov_b32 s23, 4\n s_mov_b32 s24, 19\n s_mov_b32 s25, 4\n"
            "ds_read_b64 v[42:43], v47\n ds_read_b64 v[44:45], v47 offset:8\n")
    text += "".join(body(m) for m in range(64))
    text += "s_waitcnt lgkmcnt(0)\n v_mov_b32 %[cl], v40\n v_mov_b32 %[ch], v41\n s_mov_b32 %[m], s20\n"
    return text.replace("\n", "\\n")


SOURCE = r'''
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s14", "s15", "s20", "s21", "s36", "s37", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s30", "s31", "scc", "vcc", "memory"
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, long long* cyc, int blocks64)
{
    __shared__ uint64_t tab[128];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128; i += 64)
        tab[i] = KIND >= 5 ? __builtin_bit_cast(uint64_t, 1e-3 * (double)((i * 37) % 13 - 6)) // (the FP64 forms read doubles)
                           : ((uint64_t)(uint32_t)((i * 37) % 13 - 6) << 32) | (uint32_t)(i * 2654435761u >> 9);
    __syncthreads();
    uint32_t cl = lane * 977u, ch = 4u + lane;
    const uint32_t addr = (uint32_t)(uintptr_t)tab + (lane & 31) * 8;
    int m = 5 + (int)blockIdx.x % 3, a1l = 123456789, a1h = -3;
    const long long t0 = clock64();
#pragma unroll 1
    for (int b = 0; b < blocks64; b++) {
        if (KIND == 0) asm volatile("%BLOCK_A%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 1) asm volatile("%BLOCK_A2%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 2) asm volatile("%BLOCK_B%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 3) asm volatile("%BLOCK_B2%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 4) asm volatile("%BLOCK_AR2%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 5) asm volatile("%BLOCK_FR2%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
        if (KIND == 6) asm volatile("%BLOCK_FR1%" : [cl] "+v"(cl), [ch] "+v"(ch), [m] "+s"(m) : [addr] "v"(addr), [a1l] "s"(a1l), [a1h] "s"(a1h) : CLOBBERS);
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = cl + ch + m;
    if (lane == 0)
        cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * 4 * waves_per_simd, blocks64 = 32;
    uint32_t* out;
    long long* cyc;
    hipMalloc(&out, blocks * 64 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<KIND><<<blocks, 64>>>(out, cyc, blocks64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(out, cyc, blocks64);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long c : h)
        mean += (double)c / blocks;
    std::printf("%-3s %d waves/SIMD: %7.1f clock64 ticks per sample per wave, kernel %.1f us = %.1f ns per sample per wave\n", name, waves_per_simd,
        mean / (blocks64 * 64.0), ms * 1e3, ms * 1e6 / (blocks64 * 64.0));
    hipFree(out), hipFree(cyc);
}
int main()
{
    for (int w : { 1, 2, 4, 7 }) {
        run<0>("A", w);
        run<1>("A2", w);
        run<2>("B", w);
        run<3>("B2", w);
        run<4>("AR2", w);
        run<5>("FR2", w);
        run<6>("FR1", w);
    }
    return 0;
}
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    src = SOURCE
    for kind in ("AR2", "FR2", "FR1", "A2", "B2", "A", "B"):
        src = src.replace(f"%BLOCK_{kind}%", block(kind))
    path = os.path.join(OUT, "chain_ubench.hip")
    with open(path, "w") as f:
        f.write(src)
    exe = os.path.join(OUT, "chain_ubench_bin")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", path, "-o", exe])
    if "--build-only" not in sys.argv:
        subprocess.check_call([exe])


if __name__ == "__main__":
    main()
