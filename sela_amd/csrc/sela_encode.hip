// sela_encode.hip -- MI355X (gfx950) encoder kernels of the SELA frame path.
//
// The blocks of a batch are analysed and coded by ONE of two kernels, picked by launch size (team_lanes_for):
// k_encode_teams (round 4: a wave takes four or eight blocks through mean, autocorrelation and Schur recursion side by
// side, see its own header further down) for launches that fill the device, k_encode_blocks -- a wave per block -- for
// small ones and for the host pipeline's one-launch form.  Both end in the same tail (sela_encode_tail.inc).  k_encode_blocks:
//
//   a block           one WAVE per (frame, signal): the whole of lpc::ResidueGenerator +
//                     rice::RiceEncoder x2 for that signal (reference src/lpc/residue_generator.cpp:
//                     121-134, src/rice/rice_encoder.cpp:73-81).  A stereo frame has three signals
//                     (ch0, ch1, ch0-ch1; src/frame/frame_encoder.cpp:18-60); all three are coded and
//                     the loser is simply not copied out (and, round 4, mostly not even written: the two
//                     candidates tell each other their sizes, see the tail).  Output: a fixed-stride slot of
//                     Rice words + an 8-byte BlockMeta per signal.
//   a group's last    the block that finishes LAST among those of 16 consecutive frames (finish_group):
//   block             per frame the stereo decision of src/frame/frame_encoder.cpp:64-72 (strict < on
//                     u32 word counts) and the frame's on-disk size; the group's place in the stream by a
//                     decoupled look-back over the groups before it; then the exact bytes that
//                     file::SelaFile::writeToFile emits for its frames (src/file/sela_file.cpp:115-135),
//                     written where they belong -- device memory, or the caller's page-locked buffer.
//   mean workers,     the first workgroups of the launch (mean_worker, stage_in).
//   stagers
//
// Arithmetic contract: this file must be compiled with -ffp-contract=off.  Every
// FP64 accumulator is updated in the reference's order: the lanes of a wave carry *independent*
// accumulators (autocorrelation lags, Schur columns, step-up elements), never a split of one sum.
#include <algorithm>
#include <atomic>
#include <random>
#include <vector>

#include "sela_device.h"

namespace sela {

// ---- LDS plan of k_encode_blocks (one 64-lane workgroup): 12.1 KB, twelve blocks per CU --------------
// Phase A (analysis):  samples as FP64 (x, later the centred c), split by index parity so that the
//                      per-lane operand of the lag-pair autocorrelation is a conflict-free LDS read:
//                        E[m] = c[2m], O[m] = c[2m+1].  Only HALF a block is resident at a time: each
//                      parity array holds entries [lo - 64, lo + 576) for lo = 0, then lo = 512 (the 64
//                      in front are the lags' reach back -- zeros before the block starts, x + (+-0)
//                      is exact -- the 64 behind are what is fetched ahead); the second half is
//                      recomputed from the samples in registers when the first is consumed.
// Phase B (after the autocorrelation c[] is dead): the same bytes hold
//                        [0, 8992)      samples s with 128 zero words ("no sample") in front
//                                       (FIR warm-up: "no sample" == 0), index i stored at i + i/32;
//                                       later the packed residue words
//                        [8992, 12000)  the small analysis arrays (ac, k, a, q)
constexpr int kPadC = 64;
constexpr int kHalfEntries = kBlock / 4;                 // 512 entries (1024 samples) of each parity per half
constexpr int kParityLen = kPadC + kHalfEntries + 64;    // 640 doubles
constexpr int kLastT0 = (kHalfEntries + 64) / 32 - 1;    // 17: first half holds samples lane + 64 t, t <= 17
constexpr int kFirstT1 = (kHalfEntries - kPadC) / 32;    // 14: second half holds t >= 14
// Scalar-operand scratch: each running block borrows a ring of 2 x 64 centred samples (+ a copy of the
// first values of half A behind half B, so that a fetch may run across the wrap) from a pool that is
// private to its XCD -- 512 rings per XCD -- so the ring stays in that XCD's L2 and the scratch never
// reaches HBM.  A ring is taken by swapping the launch's ticket (a number no earlier launch on this
// workspace used) into its owner word: whoever gets back anything else owns it; it is handed back by
// writing 0.  Stale or uninitialised owner words therefore read as "free" and nothing has to be
// cleared between launches.
constexpr int kRingHalf = 64;
constexpr int kRingLen = 2 * kRingHalf + 64;          // doubles (a multiple of 8, i.e. of 64 bytes)
constexpr int kRingsPerXcd = 512;                      // >= blocks resident on one XCD (12 per CU x 32 CUs)
constexpr int kXcds = 8;
constexpr int kPadS = 128;
constexpr int kSBufWords = (kPadS + kBlock) / 32 * 33; // samples, one pad word per 32: 2244 words
constexpr int kSmallBase = 8992;                      // >= kSBufWords * 4 = 8976, 16-byte aligned
constexpr int kBigBytes = kSmallBase + 3008;          // 12000
static_assert(kResWordsCap * 4 <= kSmallBase, "packed residue words must fit the dead sample buffer");
static_assert(kSBufWords * 4 <= kSmallBase && kSmallBase % 16 == 0, "sample buffer must end below the analysis arrays");
static_assert(kBigBytes >= 2 * kParityLen * 8, "the FP64 parity arrays must fit");

struct SmallArrays { // lives at kSmallBase during analysis; dead before the residue words are packed
    double ac[104];
    double k[104];   // reflection coefficients, later the dequantised ones
    int64_t a[104];  // Q35 predictor
    int32_t q[128];
};
static_assert(sizeof(SmallArrays) + kSmallBase <= kBigBytes, "analysis arrays overflow the LDS plan");

// OR `nbits` (1..32) low bits of v into the LDS bit buffer at bit position pos (buffer pre-zeroed;
// neighbouring lanes share boundary words, hence the atomic).
__device__ __forceinline__ void or_bits(uint32_t* buf, uint32_t pos, uint32_t v, uint32_t nbits)
{
    const uint32_t w = pos >> 5, sh = pos & 31;
    atomicOr(&buf[w], v << sh);
    if (sh + nbits > 32)
        atomicOr(&buf[w + 1], v >> (32 - sh));
}

// The same for a value of at most 32 bits without a branch: both words get an OR, the second one often of zero (the buffer
// has a word to spare behind the stream's last).
__device__ __forceinline__ void or_bits_both(uint32_t* buf, uint32_t pos, uint32_t v)
{
    const uint64_t wide = (uint64_t)v << (pos & 31);
    uint32_t* const w = buf + (pos >> 5);
    atomicOr(w, (uint32_t)wide);
    atomicOr(w + 1, (uint32_t)(wide >> 32));
}

// Append one Golomb-Rice codeword (src/rice/rice_encoder.cpp:41-53): u >> k ones, a zero, then the
// low k bits MSB first.  Stream bit t lives at bit t%32 of word t/32, so the MSB-first remainder is
// the bit-reversed remainder in stream order.
__device__ __attribute__((noinline)) uint32_t put_long_codeword(uint32_t* buf, uint32_t pos, uint32_t u, uint32_t k);

__device__ __forceinline__ uint32_t put_codeword(uint32_t* buf, uint32_t pos, uint32_t u, uint32_t k)
{
    uint32_t ones = u >> k;
    const uint32_t rem = k ? (__brev(u << (32 - k))) : 0u; // low k bits of u, reversed
    while (ones >= 32) {
        or_bits(buf, pos, 0xFFFFFFFFu, 32);
        pos += 32;
        ones -= 32;
    }
    if (ones + 1 + k <= 32) {
        or_bits(buf, pos, ((1u << ones) - 1u) | (rem << (ones + 1)), ones + 1 + k);
        return pos + ones + 1 + k;
    }
    or_bits(buf, pos, (1u << ones) - 1u, ones + 1);
    pos += ones + 1;
    if (k) {
        or_bits(buf, pos, rem, k);
        pos += k;
    }
    return pos;
}

__device__ __attribute__((noinline)) uint32_t put_long_codeword(uint32_t* buf, uint32_t pos, uint32_t u, uint32_t k)
{
    return put_codeword(buf, pos, u, k);
}

// rice::RiceEncoder::calculateOptimumRiceParam (src/rice/rice_encoder.cpp:20-33) for the values
// u[0..V) of every lane (invalid slots hold 0 and are not counted in n): the FIRST k in [0, 20) that
// minimises  bits(k) = sum(u >> k) + n * (1 + k).
//
// The reference evaluates all 20 candidates; the same answer needs only a few of them because
// bits() is convex in k:  T(k) = sum(u >> k) drops by d(k) = sum(ceil((u >> k) / 2)) from k to k+1 and
// ceil(floor(v/2)/2) <= ceil(v/2) makes d(k) non-increasing, so bits(k+1) - bits(k) = n - d(k) is
// non-decreasing.  Hence the first minimum is the smallest k with d(k) <= n (k = 19 if there is none),
// found by walking from a guess near log2(mean u).  Exact 64-bit sums throughout.
template <int V>
__device__ __forceinline__ uint64_t rice_shifted_sum(const uint32_t (&u)[V], uint32_t k, bool small, uint32_t& lane_part)
{
    if (small) { // every value below 2^20 (rice_plan): a lane's sum and the wave's fit 32 bits
        uint32_t part = 0;
#pragma unroll
        for (int t = 0; t < V; t++)
            part += u[t] >> k;
        lane_part = part;
        return wave_sum_small(part);
    }
    uint64_t part = 0;
#pragma unroll
    for (int t = 0; t < V; t++)
        part += u[t] >> k;
    lane_part = (uint32_t)part; // (a lane's V <= 32 quotients only count when they are short: a block with longer ones is not packed)
    return wave_sum_40(part); // V <= 32 values below 2^32 each
}

// lane_or = the OR of the lane's values (an upper bound of each); lane_quotients = sum(u >> best_k) of THIS lane's values
// (what the packer's scan needs).
template <int V>
__device__ __forceinline__ void rice_plan(const uint32_t (&u)[V], uint32_t n, uint32_t& best_k, uint64_t& best_bits, uint32_t& lane_or, uint32_t& lane_quotients)
{
    constexpr uint32_t kLast = SELA_MAX_RICE_PARAM - 1;
    lane_or = 0;
#pragma unroll
    for (int t = 0; t < V; t++)
        lane_or |= u[t];
    // 16-bit audio leaves residues below 2^19: then every sum below fits 32 bits (2048 values below 2^20), and the adds of
    // a lane go two to an instruction
    const bool small = !__any((lane_or >> 20) != 0);
    uint32_t p0, pa, pb, pc;
    const uint64_t t0 = rice_shifted_sum<V>(u, 0, small, p0);
    const uint64_t mean = n ? t0 / n : 0;
    uint32_t k = mean ? 63u - (uint32_t)__clzll(mean) : 0u; // floor(log2(mean))
    k = k > kLast - 1 ? kLast - 1 : k;
    uint64_t ta = t0; // T(k)
    pa = p0;
    if (k)
        ta = rice_shifted_sum<V>(u, k, small, pa);
    uint64_t tb = rice_shifted_sum<V>(u, k + 1, small, pb); // T(k + 1)
    if (ta - tb <= n) { // bits(k+1) >= bits(k): the first minimum is at or below k
        while (k > 0) {
            uint64_t tc = t0;
            pc = p0;
            if (k != 1)
                tc = rice_shifted_sum<V>(u, k - 1, small, pc);
            if (tc - ta > n)
                break;
            k--;
            tb = ta, pb = pa;
            ta = tc, pa = pc;
        }
    } else { // still descending: move up
        for (;;) {
            k++;
            ta = tb, pa = pb;
            if (k == kLast)
                break;
            tb = rice_shifted_sum<V>(u, k + 1, small, pb);
            if (ta - tb <= n)
                break;
        }
    }
    best_k = k;
    best_bits = ta + (uint64_t)n * (1 + k);
    lane_quotients = pa;
}

// x = s / 32767 (src/lpc/residue_generator.cpp:12-18) without the ~30-instruction IEEE division:
// q0 = s * RN(1/32767), one residual fma, one correction fma.  The result equals the correctly rounded
// quotient for EVERY |s| <= 70000 (exhaustive check: tests/test_host_logic.py::test_scale_division_is_exact);
// 16-bit channels and their difference stay within 65535.
__device__ __forceinline__ double scale_sample(int32_t s)
{
    constexpr double r = 1.0 / SELA_SAMPLE_SCALE;
    const double x = (double)s;
    const double q0 = x * r;
    const double e = __builtin_fma(-SELA_SAMPLE_SCALE, q0, x);
    return __builtin_fma(e, r, q0);
}

// ---- mean chain operand fetch (see k_encode_blocks): E[m..m+7] and O[m..m+7] by broadcast reads ------
typedef double f64x2 __attribute__((ext_vector_type(2)));
struct MeanFetch {
    f64x2 e0, e1, e2, e3, o0, o1, o2, o3;
};
#define SELA_MEAN_STR2(x) #x
#define SELA_MEAN_STR(x) SELA_MEAN_STR2(x)
#define SELA_MEAN_ISSUE(F, addr, OFF, sum)                                                    \
    asm volatile("ds_read_b128 %0, %9 offset:" #OFF "+0\n\t"                                    \
                 "ds_read_b128 %4, %9 offset:" #OFF "+5120\n\t"                                 \
                 "ds_read_b128 %1, %9 offset:" #OFF "+16\n\t"                                   \
                 "ds_read_b128 %5, %9 offset:" #OFF "+5136\n\t"                                 \
                 "ds_read_b128 %2, %9 offset:" #OFF "+32\n\t"                                   \
                 "ds_read_b128 %6, %9 offset:" #OFF "+5152\n\t"                                 \
                 "ds_read_b128 %3, %9 offset:" #OFF "+48\n\t"                                   \
                 "ds_read_b128 %7, %9 offset:" #OFF "+5168"                                      \
                 : "=&v"(F.e0), "=&v"(F.e1), "=&v"(F.e2), "=&v"(F.e3), "=&v"(F.o0), "=&v"(F.o1), "=&v"(F.o2), \
                 "=&v"(F.o3), "+v"(sum)                                                         \
                 : "v"(addr)                                                                    \
                 : "memory")
static_assert(kParityLen * 8 == 5120, "SELA_MEAN_ISSUE hard-codes the E -> O distance");

__device__ __forceinline__ void mean_wait(MeanFetch& f, double& sum)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.e0), "+v"(f.e1), "+v"(f.e2), "+v"(f.e3), "+v"(f.o0), "+v"(f.o1), "+v"(f.o2), "+v"(f.o3), "+v"(sum));
}

__device__ __forceinline__ void mean_steps(const MeanFetch& f, double& sum)
{
    sum += f.e0[0]; sum += f.o0[0]; sum += f.e0[1]; sum += f.o0[1];
    sum += f.e1[0]; sum += f.o1[0]; sum += f.e1[1]; sum += f.o1[1];
    sum += f.e2[0]; sum += f.o2[0]; sum += f.e2[1]; sum += f.o2[1];
    sum += f.e3[0]; sum += f.o3[0]; sum += f.e3[1]; sum += f.o3[1];
}

// ---- autocorrelation operand fetch (see k_encode_blocks) -------------------------------------------
// One fetch = the operands of a 16-step trip: the sixteen wave-uniform multipliers c[j] as 32 SGPRs
// (two s_load_dwordx16 from the block's scratch row, so that v_mul_f64 takes them as scalar operands)
// and this lane's sixteen new window values c[j + 1 - 2L] (eight ds_read2_b64, alternating between
// the odd and even parity arrays).  Issue and wait are separate asm statements and a fetch is in
// flight for a whole trip: the scalar cache answers in ~500 cycles, and LDS and scalar loads share
// one counter (lgkmcnt), so a wait for either is a wait for both -- hence ONE wait per trip, and the
// compiler must never see a pending load of its own in this loop (it would add more).
typedef int sgpr16 __attribute__((ext_vector_type(16)));
typedef double f64x8 __attribute__((ext_vector_type(8)));

struct AcFetch {
    sgpr16 c_lo, c_hi;          // c[j .. j+7], c[j+8 .. j+15]
    f64x2 o0, o1, o2, o3;       // O[q .. q+7]      (new values of the even steps)
    f64x2 e0, e1, e2, e3;       // E[q+1 .. q+8]    (new values of the odd steps)
};

// Issue the scalar half: S0, S1 = byte offsets into the scratch row.  The accumulators are named so
// that the statement keeps its place among the steps.
#define SELA_AC_ISSUE_S(F, c_ptr, S0, S1, acc_e, acc_o)                                   \
    asm volatile("s_load_dwordx16 %0, %4, " #S0 "\n\ts_load_dwordx16 %1, %4, " #S1          \
                 : "=&s"(F.c_lo), "=&s"(F.c_hi), "+v"(acc_e), "+v"(acc_o) : "s"(c_ptr) : "memory")

// Issue the vector half: O[q..q+7] and E[q+1..q+8] relative to addr_o / addr_e (offsets in doubles).
#define SELA_AC_ISSUE_V(F, addr_e, addr_o, Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7, Q8, acc_e, acc_o) \
    asm volatile("ds_read2_b64 %0, %11 offset0:" #Q0 " offset1:" #Q1 "\n\t"                   \
                 "ds_read2_b64 %4, %10 offset0:" #Q1 " offset1:" #Q2 "\n\t"                   \
                 "ds_read2_b64 %1, %11 offset0:" #Q2 " offset1:" #Q3 "\n\t"                   \
                 "ds_read2_b64 %5, %10 offset0:" #Q3 " offset1:" #Q4 "\n\t"                   \
                 "ds_read2_b64 %2, %11 offset0:" #Q4 " offset1:" #Q5 "\n\t"                   \
                 "ds_read2_b64 %6, %10 offset0:" #Q5 " offset1:" #Q6 "\n\t"                   \
                 "ds_read2_b64 %3, %11 offset0:" #Q6 " offset1:" #Q7 "\n\t"                   \
                 "ds_read2_b64 %7, %10 offset0:" #Q7 " offset1:" #Q8                           \
                 : "=&v"(F.o0), "=&v"(F.o1), "=&v"(F.o2), "=&v"(F.o3), "=&v"(F.e0), "=&v"(F.e1), \
                 "=&v"(F.e2), "=&v"(F.e3), "+v"(acc_e), "+v"(acc_o)                           \
                 : "v"(addr_e), "v"(addr_o)                                                   \
                 : "memory")

// Lands the fetch: the fetched registers become defined here, behind the steps issued so far.
__device__ __forceinline__ void ac_wait(AcFetch& f, double& acc_e, double& acc_o)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(f.c_lo), "+s"(f.c_hi), "+v"(f.o0), "+v"(f.o1), "+v"(f.o2), "+v"(f.o3), "+v"(f.e0), "+v"(f.e1),
                 "+v"(f.e2), "+v"(f.e3), "+v"(acc_e), "+v"(acc_o));
}

__device__ __forceinline__ void ac_step(double cj, double nw, double& A, double& B, double& acc_e, double& acc_o)
{
    acc_e += cj * A; // lag 2L   : c[j] * c[j - 2L]
    acc_o += cj * B; // lag 2L+1 : c[j] * c[j - 2L - 1]
    B = A;
    A = nw;          // c[j + 1 - 2L]
}

// steps 0 and 1 of a trip: the last to read the previous fetch's registers (through A and B)
__device__ __forceinline__ void ac_steps_head(const AcFetch& f, double& A, double& B, double& acc_e, double& acc_o)
{
    const f64x8 c = __builtin_bit_cast(f64x8, f.c_lo);
    ac_step(c[0], f.o0[0], A, B, acc_e, acc_o);
    ac_step(c[1], f.e0[0], A, B, acc_e, acc_o);
}

__device__ __forceinline__ void ac_steps_tail(const AcFetch& f, double& A, double& B, double& acc_e, double& acc_o)
{
    const f64x8 c = __builtin_bit_cast(f64x8, f.c_lo), d = __builtin_bit_cast(f64x8, f.c_hi);
    ac_step(c[2], f.o0[1], A, B, acc_e, acc_o);
    ac_step(c[3], f.e0[1], A, B, acc_e, acc_o);
    ac_step(c[4], f.o1[0], A, B, acc_e, acc_o);
    ac_step(c[5], f.e1[0], A, B, acc_e, acc_o);
    ac_step(c[6], f.o1[1], A, B, acc_e, acc_o);
    ac_step(c[7], f.e1[1], A, B, acc_e, acc_o);
    ac_step(d[0], f.o2[0], A, B, acc_e, acc_o);
    ac_step(d[1], f.e2[0], A, B, acc_e, acc_o);
    ac_step(d[2], f.o2[1], A, B, acc_e, acc_o);
    ac_step(d[3], f.e2[1], A, B, acc_e, acc_o);
    ac_step(d[4], f.o3[0], A, B, acc_e, acc_o);
    ac_step(d[5], f.e3[0], A, B, acc_e, acc_o);
    ac_step(d[6], f.o3[1], A, B, acc_e, acc_o);
    ac_step(d[7], f.e3[1], A, B, acc_e, acc_o);
}

// Taps j0 + JJ + 1 .. j0 + 32 of the residue FIR (see sela_encode_tail.inc), stopping at `order`.  JJ is a template
// parameter so that the 32-register sample window is addressed statically: tap j uses win[(t - j) mod 32] =
// s[32 lane + t - j] and loads the one new element s[32 lane - j].  In FP64 (round 4; rounds 1-3 ran the same window on
// 64-bit integers: a = a_hi 2^32 + a_lo, v_mad_i64_i32 -- which issues at half rate -- for the low part and a v_mad_i32_i24
// for the high part of the 15 % of the taps that have one): v_fma_f64 issues at full rate, and a fused multiply-add of
// integers is EXACT while every partial sum stays below 2^53 -- which the caller has checked for the block, for the whole
// coefficients (one pass) or for their halves (two passes).  coef[j] = the coefficient (or its half) as a double, the window
// holds the samples as doubles, acc the exact sum.
template <int JJ>
__device__ __forceinline__ void fir_taps_f64(int j0, int order, int lane, const int32_t* sT, const double* a_f,
    double (&win)[kPerLane], double (&acc)[kPerLane])
{
    const int j = j0 + JJ + 1;
    if (j > order)
        return;
    const double aj = read_first_lane(a_f[j]);
    const int e = kPadS + 32 * lane - j;
    win[(32 - JJ - 1) & 31] = (double)sT[e + (e >> 5)];
#pragma unroll
    for (int t = 0; t < kPerLane; t++)
        acc[t] = __builtin_fma(aj, win[(t - JJ - 1) & 31], acc[t]); // s[32 lane + t - j]
    __builtin_amdgcn_sched_barrier(0); // (taps are not interleaved: the window and the sums already fill the register file)
    if constexpr (JJ < 31)
        fir_taps_f64<JJ + 1>(j0, order, lane, sT, a_f, win, acc);
}

// The predictions for a predictor the FP64 taps cannot carry exactly (sela_encode_tail.inc): one multiply-add per (sample, tap) in
// full 64-bit wrap-around arithmetic, samples straight from LDS; (int32)((2^34 + sum) >> 35) of sample
// 32 lane + t goes to pred_out[t * 64 + lane] (global scratch: the block's own output slot, unused so
// far).  Slow, out of line, and never needed by 16-bit audio (its predictors stay below 2^37: the
// dequantisation tables cap every reflection coefficient); tests reach it through
// sela_hip_debug_force_plain_fir, which sends every block of the PRODUCT instantiation down this branch.
__device__ __attribute__((noinline)) void fir_plain(int order, int lane, const int32_t* sT, const int64_t* a, uint32_t* pred_out)
{
#pragma unroll 1
    for (int t = 0; t < kPerLane; t++) {
        uint64_t sum = (uint64_t)1 << (SELA_Q_SHIFT - 1);
#pragma unroll 1
        for (int j = 1; j <= order; j++) {
            const int e = kPadS + 32 * lane + t - j;
            sum += (uint64_t)a[j] * (uint64_t)(int64_t)sT[e + (e >> 5)];
        }
        pred_out[t * 64 + lane] = (uint32_t)(int32_t)((int64_t)sum >> SELA_Q_SHIFT);
    }
}

// ---- mean workers -------------------------------------------------------------------------------------------
// The mean of a block (src/lpc/residue_generator.cpp:27-30) is one strictly sequential 2048-term FP64 sum.  Inside
// a block's own wave it is a chain that all 64 lanes walk redundantly: 2048 vector adds (plus the FP64 samples
// written to LDS for it), 12 % of the block's instructions, for ONE number.  The same chain laid across lanes --
// lane = block, 64 blocks per wave -- costs 2048 adds per 64 blocks.  So the first workgroups of the launch are
// "mean workers": each sums the 64 blocks of encode indices [self_blocks + 64 w, +64) and publishes mean + a
// ready word (= the launch's ticket, release / acquire at agent scope: consumer and worker may sit on different
// XCDs).  Blocks below self_blocks -- the ones that start with the launch, before any worker could have finished --
// walk their own chain as before; so does any block whose ready word has not turned up after a bounded wait, so
// no block ever depends on another workgroup making progress.
constexpr uint32_t kFuseSpinLimit = 1u << 18; // bounded waits on other groups: x (s_sleep 64 + a trip to memory, ~4 us) = ~1 s, then the launch flags an error instead of hanging
constexpr uint32_t kFuseNapLimit = 600000;  // waits for the stagers: x 2048 cycles = ~0.5 s (EncodeHostLink::wait_naps overrides: tests)
constexpr uint32_t kMeanWaitSpins = 24;   // x s_sleep 16 (~1000 cycles each): ~10 us, then the block computes its own mean

__device__ __forceinline__ void block_of(uint32_t e, uint32_t n_sig, uint32_t& frame, uint32_t& sig)
{
    // XCD-aware, for speed only: as observed, workgroup b lands on XCD (b + where the launch started) % 8, so encode
    // indices that are equal mod 8 (the worker prefix is a multiple of 8) run on one XCD.  ALL blocks of a group of
    // eight frames (kGroupFrames) get such indices: the signals of a frame share that XCD's L2 copy of the PCM, and
    // what the group's blocks hand to its last block does not cross the chip.  Nothing is correct because of this
    // (see the note at store_through).  Eight groups, one per XCD, advance side by side: a "span" of 64 frames.
    const uint32_t per_group = 8 * n_sig, per_span = 8 * per_group;
    const uint32_t span = e / per_span, rem = e % per_span;
    const uint32_t xcd = rem % 8, j = rem / 8; // j: the block's number inside its group
    sig = j / 8;
    frame = (span * 8 + xcd) * 8 + (j % 8);
}

// ---- handing data to another workgroup -------------------------------------------------------------------------------
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility".)  The eight XCDs' L2 caches
// are not coherent with each other and a CU's vector cache is never refreshed by another CU's stores.  The textbook
// hand-over -- plain stores, a release fence at agent scope, a flag -- writes back EVERY dirty line of the writer's L2
// (buffer_wbl2), and this kernel keeps its scalar-operand rings dirty in L2 on purpose: with one such fence per block
// the launch took 2.36 ms instead of 0.41.  What the one-launch form hands from workgroup to workgroup therefore goes
//   producer: stores THROUGH the L2 (relaxed atomic stores at agent scope = sc1) -> asm s_waitcnt vmcnt(0) (written
//             out: a fence's own wait is dropped by the compiler when it believes nothing is outstanding -- the first
//             version, with a workgroup-scope fence here, handed over one stale word in some hundred runs and stalled
//             or not depending on an unrelated line of code) -> the mark, a relaxed atomic at agent scope;
//   consumer: sees the mark with a relaxed atomic -> ONE acquire fence at agent scope (invalidates this CU's vector
//             cache) -> plain loads: a group's last block, once per group.  Where every block would need one (the frame
//             from the stagers) the consumer loads past the L2 as well, which the producer's write-through stores allow;
// and the look-back cells are single 64-bit words that say themselves whether they are valid (launch mark | payload,
// relaxed atomics on both sides): such a word is either there or not yet.  Where workgroups land (block_of keeps a
// group's blocks on one XCD) is for speed only; nothing here is correct because of it.
template <typename T>
__device__ __forceinline__ void store_through(T* p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T load_through(const T* p)
{
    return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stores_done() // (every store of this wave so far has been acknowledged)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void drop_stale_lines() // (nothing older than now is served from this CU's vector cache)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// What marks a counter cell as this launch's (group_arrive): 40 bits of the hashed tag above a 24-bit count.
__device__ __forceinline__ uint64_t count_mark(uint64_t tag) { return (tag * 0x9E3779B97F4A7C15ull) & ~(uint64_t)0xFFFFFF; }

// ---- stagers: a second kernel fetches the PCM from page-locked host memory (host pipeline, stereo) ---------------
// A kernel reads host memory at the link's rate (55.9 GB/s measured with >= 128 waves of 16-byte loads,
// tools/pcie_probe.hip) without the ~15 us the copy engine idles between two copies and without an event per chunk.
// k_stage_in is launched before k_encode_blocks on a stream of its own; one wave copies one frame at a time.
//  * Every workgroup asks for (nearly) all the LDS of a CU and so has the CU to itself.  A load from host memory
//    takes microseconds and a CU returns its waves' loads in order: staged from the encode launch's own first
//    workgroups (the first version), the blocks that shared a CU with a stager took 500 us instead of 110 -- with
//    one stager per CU and 2 KB in flight each, all of them did.
//  * Nothing makes two kernels run at the same time (streams share hardware queues, and then the second launch
//    starts when the first has ended), so the stagers wait for nothing and the encode launch is correct either
//    way: behind a finished stager kernel it finds everything in place; beside a running one, a block waits for its
//    frame's word in pcm_ready.
//  * Beside each other the two kernels' workgroups sit on whatever XCDs the dispatcher's round robin has reached,
//    so a frame crosses from one L2 to another: the stager stores it through its L2 and then publishes ONE word, the
//    launch ticket | a checksum of the frame; the block reads the frame past its own L2 and takes it when the
//    checksum matches (await_frame) -- whatever the order in which the stores become visible (see store_through).
constexpr int kStageThreads = 256; // four waves per CU: with sixteen, the waves of a CU took turns so unevenly that single frames took 240 us
constexpr int kStageLdsBytes = 150 * 1024; // (with the 12.1 KB of a block that is more than a CU has)

__device__ __forceinline__ uint32_t frame_check_term(uint32_t word, uint32_t index) { return word * (2u * index + 1u); }

constexpr uint32_t kStageLingerSpins = 8000; // x (s_sleep 64 + a trip to memory, ~2.5 us) = ~20 ms

__global__ __launch_bounds__(kStageThreads) void k_stage_in(const int16_t* __restrict__ host_pcm, int16_t* __restrict__ pcm, uint32_t n_frames,
    uint64_t* __restrict__ pcm_ready, uint32_t ticket, uint64_t* __restrict__ started, uint64_t tag, uint32_t n_groups)
{
    // started[0]: roll call (launch ticket | workgroups that have a CU), [1]: the frame counter, [2]: k_stage_gate's mark,
    // [3]: groups of frames the encode launch has finished (count_mark | count)
    extern __shared__ unsigned char stage_lds[]; // (never touched: it keeps other workgroups off this CU)
    if (threadIdx.x == 0) { // one more stager workgroup has its CU (k_stage_gate)
        uint64_t old = __hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const uint64_t neu = (uint32_t)(old >> 32) == ticket ? old + 1 : (((uint64_t)ticket << 32) | 1u);
            if (__hip_atomic_compare_exchange_strong(started, &old, neu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                break;
        }
    }
    constexpr uint32_t n16 = kBlock * 2 * 2 / 16; // 16-byte pieces of a stereo frame: 512, eight per lane
    const uint32_t lane = threadIdx.x % 64;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // The frames are handed out one by one (a counter next to the roll call): the link does not serve the CUs evenly --
    // with frames dealt out in advance (wave w: frames w, w + 128, ...) some waves were at their fifteenth frame when
    // others were at their third, and the blocks, which start in frame order, sat waiting for the slow waves' frames
    // while frames far ahead lay ready.
    // Every lane adds 1 and the wave takes frame (lane 0's old value) / 64: the compiler folds the 64 additions into one
    // fetch-and-add of 64 (and 64 of them would be as correct).  The obvious form -- lane 0 adds 1, readfirstlane hands
    // the frame to the others -- is what the first version had, next to the flag store below, also lane 0's: the
    // compiler fused the two lane-0 regions across the loop's back edge and sent lanes 1..63 round again with the 0
    // they had been given "for now" -- a stager kernel that never ended, or not, depending on an unrelated line in the
    // loop.  No value leaves a one-lane region here any more.
    // (A compare-and-swap loop instead of the add: 128 waves took turns, a frame per 1.4 us.)
    uint64_t* const next = started + 1; // (zeroed by the launcher on this stream)
    for (;;) {
        const uint64_t drawn = __hip_atomic_fetch_add(next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(drawn >> 6));
        if (f >= n_frames)
            break;
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(host_pcm) + (size_t)f * n16;
        uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(pcm) + (size_t)f * n16 * 2;
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            v[u] = __builtin_nontemporal_load(src + 64 * u + lane);
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t q = 64 * u + lane;
            sum += frame_check_term(v[u].x, 4 * q) + frame_check_term(v[u].y, 4 * q + 1) + frame_check_term(v[u].z, 4 * q + 2)
                + frame_check_term(v[u].w, 4 * q + 3);
            store_through(dst + 2 * q, (uint64_t)v[u].x | ((uint64_t)v[u].y << 32));
            store_through(dst + 2 * q + 1, (uint64_t)v[u].z | ((uint64_t)v[u].w << 32));
        }
        sum = wave_sum_small(sum); // (mod 2^32)
        stores_done();
        if (lane == 0) // (nothing comes back out of this one-lane region)
            store_through(pcm_ready + f, ((uint64_t)ticket << 32) | sum);
    }
    // Do not END beside a running encode launch: when a kernel ends the device writes back its caches, and the blocks
    // that were then still waiting for the last frames saw them 30-45 us late (stamped: the frames of the last 8 groups
    // accepted 40 us after their publication instead of 8; the launch ended 50 us earlier with stagers that stayed).
    // So the waves stay -- asleep, on CUs nobody else needs at the tail -- until the encode launch has finished its
    // last group.  Only if that launch runs BESIDE this kernel, which k_stage_gate's mark says (the gate is on the
    // encode launch's stream, directly in front of it): behind a stager kernel that has to end first there is nothing
    // to wait for.  Bounded (~20 ms) like every wait here.
    if (load_through(started + 2) == tag) {
        const uint64_t mark = count_mark(tag);
        for (uint32_t spins = 0; spins < kStageLingerSpins; spins++) {
            const uint64_t c = load_through(started + 3);
            if ((c & ~(uint64_t)0xFFFFFF) == mark && (uint32_t)(c & 0xFFFFFF) >= n_groups)
                break;
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

// The stagers need whole CUs, and blocks that wait for their frames never leave theirs: an encode launch that got
// onto the device before the stagers would keep them off it.  So this one wave goes first on the encode launch's
// stream and ends when every stager workgroup has a CU (or, after ~20 ms, regardless: then the blocks' own waits
// decide).  If the two streams share a hardware queue the stager kernel, launched first, has ended by now.
__global__ __launch_bounds__(64) void k_stage_gate(uint64_t* __restrict__ started, uint32_t ticket, uint32_t n_workgroups, uint64_t tag)
{
    if (threadIdx.x == 0) // "the encode launch's stream runs beside the stagers" (see the end of k_stage_in)
        store_through(started + 2, tag);
    for (uint32_t spins = 0; spins < 20000; spins++) {
        const uint64_t c = load_through(started);
        if ((uint32_t)(c >> 32) == ticket && (uint32_t)c >= n_workgroups)
            return;
        __builtin_amdgcn_s_sleep(32);
    }
}

// The PCM pointer as the code behind a wait must see it: the kernel argument is const and restrict-qualified, which
// lets the compiler treat the samples as unchanging for the whole kernel and read them before the wait.  Passing the
// pointer through an opaque asm cuts that knowledge off (and nothing moves across the asm's memory clobber).
__device__ __forceinline__ const int16_t* pcm_after_wait(const int16_t* p)
{
    asm volatile("" : "+s"(p) : : "memory");
    return p;
}

// Wait until stereo frame `f` has been copied in and can be read from here (no-op without stagers): its word in
// pcm_ready carries this launch's ticket, and the frame, read past this XCD's L2, has the checksum that word gives.
// Bounded: a wait that runs out flags the launch.
__device__ __attribute__((noinline)) bool await_frame(const uint64_t* pcm_ready, const int16_t* pcm, uint32_t f, uint32_t ticket, uint32_t nap_limit)
{
    // Most of a launch's resident blocks wait here for most of the copy, and these loads go to memory: polled every
    // quarter microsecond by 3,000 waves, the ready words took the link's bandwidth from the stagers (4.7 ms for a
    // 0.6 ms copy).  The pause doubles from ~0.9 us to ~7 us.
    const uint32_t lane = threadIdx.x;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(pcm) + (size_t)f * kBlock;
    if (nap_limit == 0)
        return false; // (tests: "the stagers never showed up" -- whether or not they did)
    uint32_t naps = 1, slept = 0;
    for (;;) {
        const uint64_t c = load_through(pcm_ready + f);
        if ((uint32_t)(c >> 32) == ticket) {
            uint32_t sum = 0;
#pragma unroll 8
            for (uint32_t t = 0; t < (uint32_t)kPerLane; t++)
                sum += frame_check_term(load_through(words + lane + 64 * t), lane + 64 * t);
            if (wave_sum_small(sum) == (uint32_t)c) {
                return true;
            }
        }
        if (slept > nap_limit)
            return false;
        for (uint32_t i = 0; i < naps; i++)
            __builtin_amdgcn_s_sleep(32); // 2048 cycles
        slept += naps;
        naps = naps < 8 ? naps * 2 : 8;
    }
}

// Which lane of which worker sums which block: lane -> (frame, signal) with the signals of a frame in NEIGHBOURING lanes
// (64 / n_sig frames per wave).  The lanes of one frame then load the same addresses, which the memory pipeline serves with
// one request: a frame is fetched once per worker instead of once per signal (lane = encode index, the first version,
// fetched every stereo frame three times -- 26 MB of the launch's 58 MB of reads, PMC FETCH_SIZE).
__device__ __forceinline__ uint32_t worker_frames_per_wave(uint32_t n_sig) { return n_sig < 64u ? 64u / n_sig : 1u; }
__device__ __forceinline__ uint32_t worker_first_frame(uint32_t self_blocks, uint32_t n_sig) { return self_blocks / (64u * n_sig) * 64u; } // the span of 64 frames that holds encode index self_blocks
__device__ __forceinline__ uint32_t encode_index_of(uint32_t frame, uint32_t sig, uint32_t n_sig) // (block_of, the other way round)
{
    const uint32_t span = frame / 64, within = frame % 64;
    return span * 64 * n_sig + (sig * 8 + within % 8) * 8 + within / 8;
}

__device__ __attribute__((noinline)) void mean_worker(const int16_t* __restrict__ pcm, uint32_t n_frames, uint32_t channels, uint32_t n_sig,
    uint32_t first_frame, double* __restrict__ mean_out, uint64_t* __restrict__ mean_ready, uint64_t tag)
{
    const uint32_t per_wave = worker_frames_per_wave(n_sig);
    const uint32_t slot = n_sig <= 64u ? threadIdx.x / n_sig : 0u;
    uint32_t frame = first_frame + slot, sig = n_sig <= 64u ? threadIdx.x % n_sig : threadIdx.x; // (more than 64 signals: one frame, lane = signal, the first 64)
    const bool live = slot < per_wave && frame < n_frames && sig < n_sig;
    if (!live)
        frame = 0, sig = 0; // idle lanes shadow a valid block, never publish
    const uint32_t e = encode_index_of(frame, sig, n_sig);
    const int16_t* fp = pcm + (size_t)frame * kBlock * channels;
    // the blocks that wait for these means hold CU slots: take the issue slots the co-resident block waves would
    // otherwise win (their phases are throughput-bound, this one is a chain)
    __builtin_amdgcn_s_setprio(3);
    double sum = 0.0;
    if (channels == 2) {
        const uint4* p = reinterpret_cast<const uint4*>(fp); // 4 stereo pairs per load (dword-aligned: unaligned vector loads are fine)
        const uint32_t first_shift = sig == 1 ? 16u : 0u;     // signal 0: l, 1: r, 2: l - r  as  a - (b & mask)
        const uint32_t second_mask = sig == 2 ? 0xFFFFFFFFu : 0u;
#pragma unroll 4
        for (int j4 = 0; j4 < kBlock / 4; j4++) {
            const uint4 v = p[j4];
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t a = (int32_t)(w[i] << (16 - first_shift)) >> 16, b = ((int32_t)w[i] >> 16) & (int32_t)second_mask;
                sum += scale_sample(a - b);
            }
        }
    } else if (channels == 1) {
        const uint4* p = reinterpret_cast<const uint4*>(fp); // 8 samples per load
#pragma unroll 2
        for (int j8 = 0; j8 < kBlock / 8; j8++) {
            const uint4 v = p[j8];
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                sum += scale_sample((int16_t)(w[i] & 0xFFFFu));
                sum += scale_sample((int16_t)(w[i] >> 16));
            }
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < kBlock; j++)
            sum += scale_sample(fp[(size_t)j * channels + sig]);
    }
    const double mean = sum / (double)kBlock;
    if (live) {
        // (two words for blocks on any XCD, both past the L2 on both sides)
        store_through(reinterpret_cast<uint64_t*>(mean_out) + e, __builtin_bit_cast(uint64_t, mean));
        stores_done();
        store_through(mean_ready + e, tag);
    }
}

// ---- a group's last block: sizes, place in the stream, on-disk bytes ---------------------------------------------
// Round 2 had two more kernels behind k_encode_blocks (a one-workgroup plan: stereo decision, sizes, scan; an
// assembler, one workgroup per frame).  Between two batches they cost two launch hand-overs; in the host pipeline
// they waited behind the next chunk's blocks for a free CU (80-95 us per chunk, traced) and forced a host
// hand-over (sizes) before every copy-out.  Now the blocks do it: frames are grouped by kGroupFrames; every block, when its
// slot and BlockMeta are written, counts itself in; the one that completes the count finishes the group.

// Decoupled look-back cells of one group.  Groups finish on different XCDs, so every cell is ONE 64-bit word that says
// itself whether it is valid: the launch's 32-bit mark (a hash of its tag) in the high half, the payload in the low half.
struct GroupState {
    uint64_t agg_bytes;  // bytes of this group's frames                              (known as soon as the group is complete)
    uint64_t agg_flags;  // flag bits of its blocks
    uint64_t prefix_lo;  // stream offset behind this group's last frame, bits 0..31  (known once every group before it is)
    uint64_t prefix_hi;  // ... bits 32..63
    uint64_t prefix_info; // flags | frames that do not fit frames_cap << 8, of all groups up to and including this one
};
static_assert(sizeof(GroupState) == 40, "GroupState");

constexpr int kGroupCountStride = 16; // uint64 words: one counter per 128-byte line

struct FuseArgs {
    const BlockMeta* meta;
    const uint32_t* slots;
    uint64_t* group_count;   // [n_groups]: 40 bits of the launch's tag (hashed) << 24 | blocks that have arrived
    GroupState* group_state; // [n_groups]
    uint8_t* frames;         // the stream (device memory, or page-locked host memory as the device sees it)
    size_t frames_cap;
    uint64_t* frame_offsets; // [n_frames + 1] or null
    uint64_t* mirror;        // host copy of frame_offsets, + one word status[0] | status[1] << 32; or null
    uint32_t* status;
    const uint64_t* pos_in;  // offset of this launch's first frame in the stream (null: 0) ...
    uint64_t* pos_out;       // ... and where the launch leaves the offset behind its last frame (null: nowhere).  Two cells: groups
                             // read the start long after the last group -- which only needs the others' sizes -- has written the end
    const uint64_t* pcm_ready; // [n_frames]: launch ticket | checksum of every frame k_stage_in has copied in; or null (the PCM is there)
    uint64_t* groups_done;   // count_mark | groups finished, for the stagers (see the end of k_stage_in); or null
    uint32_t n_frames, channels, n_sig, ticket;
    int32_t* trace_residues; // the trace build also stores every block's residues here ([block][2048]); or null (sela_hip_lpc_encode)
    uint32_t nap_limit;      // bound of a block's wait for its frame (await_frame)
    uint64_t tag;            // process nonce << 32 | ticket (see launch_encode): what marks a cell as written by THIS launch
    uint64_t* sizes_pub;     // [blocks]: the stereo candidates' sizes (sela_encode_tail.inc), sizes_tag | words; or null
    uint64_t sizes_tag;      // 20 bits of the nonce << 44 | ticket << 12
    uint32_t priorities;     // wave priorities by quarters of a wave's work (0: none), see k_encode_teams' note on priorities
};

__device__ __forceinline__ uint32_t frame_words(const BlockMeta* m, uint32_t channels, uint32_t& choice, uint32_t& flags)
{
    uint32_t words = 0;
    choice = 0;
    for (uint32_t c = 0; c < channels; c++) {
        BlockMeta b = m[c];
        if (c == 1 && channels == 2) { // exactly-stereo only, src/frame/frame_encoder.cpp:18
            const BlockMeta d = m[2];
            const uint32_t dsz = (uint32_t)d.coef_words + d.res_words, asz = (uint32_t)b.coef_words + b.res_words;
            flags |= d.flags & ~kBlockFormBits;
            if (dsz < asz) { // strict <, src/frame/frame_encoder.cpp:64
                choice = 1;
                b = d;
            }
        }
        flags |= b.flags & ~kBlockFormBits;
        words += (uint32_t)b.coef_words + b.res_words;
    }
    return words;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
    for (int d = 32; d >= 1; d >>= 1)
        v += (uint64_t)__shfl_xor((unsigned long long)v, d, 64);
    return v;
}

// ---- kMode 3: the analysis' FP64 intermediates folded into two 64-bit words per block --------------------------------------
// The trace builds (kMode 1) write mean, ac[0..100] and k[0..99] of every block and cost registers the product kernels do not
// have to spare, so they are instantiations of their own; this mode is the product kernel plus a handful of instructions behind
// the normalisation and the Schur recursion: hash(ac) = XOR_i mix(bits(ac[i]), i), hash(k) likewise, written to
// hashes[2 block], hashes[2 block + 1] (the launch's d_trace pointer, reinterpreted).  The tests fold the oracle's trace the
// same way: one wrong bit in one of the 201 doubles changes the word.  NaNs (degenerate blocks) count as one canonical NaN.
__device__ __forceinline__ uint64_t hash_term(double v, int i)
{
    uint64_t x = v != v ? 0x7FF8000000000000ull : __builtin_bit_cast(uint64_t, v);
    x ^= 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <int kLanes> // XOR over aligned groups of kLanes lanes (every lane of a group gets the result)
__device__ __forceinline__ uint64_t lanes_xor(uint64_t v)
{
#pragma unroll
    for (int d = kLanes / 2; d >= 1; d >>= 1)
        v ^= (uint64_t)__shfl_xor((unsigned long long)v, d, 64);
    return v;
}

// The on-disk bytes of one frame (src/file/sela_file.cpp:115-135), by one wave.  Frame sizes are multiples of 4
// and the stream base is 4-byte aligned, so everything is written as aligned u32.  Within a subframe the 7 header
// bytes push the coefficient words 3 bytes off word alignment (funnel shift below); the 5 bytes of the residue
// header realign the residue words.
__device__ __forceinline__ void assemble_frame(const FuseArgs& fa, uint32_t f, uint32_t choice, uint32_t* __restrict__ out, int lane)
{
    if (lane == 0)
        out[0] = SELA_SYNC_WORD;
    uint32_t p = 1; // word cursor inside the frame
    for (uint32_t c = 0; c < fa.channels; c++) {
        uint32_t sig = c, type = 0, parent = c;
        if (c == 1 && fa.channels == 2 && choice) {
            sig = 2;
            type = 1;
            parent = 0;
        }
        const BlockMeta b = fa.meta[(size_t)f * fa.n_sig + sig];
        const uint32_t* __restrict__ slot = fa.slots + ((size_t)f * fa.n_sig + sig) * kSlotWords;
        const uint32_t cw = b.coef_words, rw = b.res_words;
        if (lane == 0)
            out[p] = c | (type << 8) | (parent << 16) | ((uint32_t)b.coef_k << 24);
        // words p+1 .. p+1+cw: [cw:16 | order:8] then the coefficient words shifted by 3 bytes, then res_k
        if ((uint32_t)lane <= cw) { // (cw <= kCoefWordsCap = 32)
            const uint32_t low = lane == 0 ? (cw | ((uint32_t)b.order << 16)) : (slot[lane - 1] >> 8);
            const uint32_t top = (uint32_t)lane < cw ? slot[lane] : (uint32_t)b.res_k;
            out[p + 1 + lane] = (low & 0x00FFFFFFu) | (top << 24);
        }
        if (lane == 0)
            out[p + 2 + cw] = rw | ((uint32_t)kBlock << 16);
        // the residue words: 16-byte loads (the slot is 16-byte aligned), four loads in flight, word stores
        const uint4* __restrict__ rs = reinterpret_cast<const uint4*>(slot + kCoefWordsCap);
        uint32_t* __restrict__ ro = out + p + 3 + cw;
        for (uint32_t base = 0; base < rw; base += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t w = base + 256 * u + 4 * lane;
                v[u] = w < rw ? rs[w / 4] : make_uint4(0, 0, 0, 0); // (reads inside the slot: rw <= kResWordsCap, a multiple of 4 words)
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t w = base + 256 * u + 4 * lane;
                if (w < rw)
                    ro[w] = v[u].x;
                if (w + 1 < rw)
                    ro[w + 1] = v[u].y;
                if (w + 2 < rw)
                    ro[w + 2] = v[u].z;
                if (w + 3 < rw)
                    ro[w + 3] = v[u].w;
            }
        }
        p += 3 + cw + rw;
    }
}

// The on-disk bytes of a whole group, when its subframes fit one per lane (<= 64: the usual case).  assemble_frame
// walks a frame's subframes one behind the other, every load waiting for the one before it: ~6 us per subframe, 190 us
// for 16 stereo frames, and the launch ended that much after its last block.  Here every subframe has a lane that
// writes its headers and coefficient words (all of a subframe's coefficient words are fetched at once), and the
// residue words are copied by all lanes, sixteen 16-byte loads in flight each.

__device__ __forceinline__ void assemble_group(const FuseArgs& fa, uint32_t f0, uint32_t nfg, uint32_t choice, uint64_t begin, bool fits,
    int lane)
{
    const uint32_t ch = fa.channels, n_sub = nfg * ch;
    const uint32_t fi = (uint32_t)lane / ch, c = (uint32_t)lane % ch; // this lane's subframe: frame f0 + fi, channel c
    const bool live = (uint32_t)lane < n_sub;
    const int src_lane = live ? (int)fi : 0;
    const uint32_t my_choice = (uint32_t)__shfl((int)choice, src_lane, 64);
    const uint64_t my_begin = (uint64_t)__shfl((unsigned long long)begin, src_lane, 64);
    const bool my_fits = __shfl((int)fits, src_lane, 64) != 0;
    uint32_t sig = c, type = 0, parent = c;
    if (c == 1 && ch == 2 && my_choice) {
        sig = 2;
        type = 1;
        parent = 0;
    }
    BlockMeta b = {};
    const uint32_t* slot = fa.slots;
    if (live) {
        const size_t block_id = (size_t)(f0 + fi) * fa.n_sig + sig;
        b = fa.meta[block_id];
        slot = fa.slots + block_id * kSlotWords;
    }
    const uint32_t cw = live ? b.coef_words : 0u, rw = live ? b.res_words : 0u;
    const uint32_t sub_words = live ? 3u + cw + rw : 0u;
    const uint32_t before = wave_exclusive_scan(sub_words, lane);
    const uint32_t frame_start = (uint32_t)__shfl((int)before, live ? (int)(fi * ch) : 0, 64);
    const uint32_t p = 1u + before - frame_start; // word cursor inside the frame
    uint32_t* out = reinterpret_cast<uint32_t*>(fa.frames + my_begin);
    const bool write = live && my_fits;
    // ---- headers and coefficient words ----
    uint4 cwv[kCoefWordsCap / 4];
#pragma unroll
    for (int i = 0; i < kCoefWordsCap / 4; i++)
        cwv[i] = (write && 4u * i < cw) ? reinterpret_cast<const uint4*>(slot)[i] : make_uint4(0, 0, 0, 0);
    if (write) {
        if (c == 0)
            out[0] = SELA_SYNC_WORD;
        out[p] = c | (type << 8) | (parent << 16) | ((uint32_t)b.coef_k << 24);
        // words p+1 .. p+1+cw: [cw:16 | order:8] then the coefficient words shifted by 3 bytes, then res_k
        uint32_t prev = cw | ((uint32_t)b.order << 16);
#pragma unroll
        for (int i = 0; i <= kCoefWordsCap; i++) {
            const uint4 q = cwv[(i < kCoefWordsCap ? i : 0) / 4];
            const uint32_t word = i >= kCoefWordsCap ? 0u : ((i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w);
            if ((uint32_t)i <= cw) {
                const uint32_t top = (uint32_t)i < cw ? word : (uint32_t)b.res_k;
                out[p + 1 + i] = (prev & 0x00FFFFFFu) | (top << 24);
            }
            prev = word >> 8;
        }
        out[p + 2 + cw] = rw | ((uint32_t)kBlock << 16);
    }
    // ---- residue words ----
    // The destination of a subframe's residue words is 4-byte aligned, no more.  Its own lane writes the 0..3 words up to
    // the first 16-byte boundary and the 0..3 behind the last one; everything between is 16-byte pieces, stored aligned
    // (one instruction per piece, fully coalesced) and loaded from wherever that puts them in the slot (4-byte aligned
    // 16-byte loads are fine).  The pieces of all subframes form one flat list, sixteen loads in flight per lane.
    const uint32_t* res_src = slot + kCoefWordsCap;
    uint32_t* res_dst = out + p + 3 + cw;
    const uint32_t head = write ? min(rw, (uint32_t)((16u - ((uintptr_t)res_dst & 15u)) & 15u) / 4u) : 0u;
    const uint32_t pieces = write ? (rw - head) / 4 : 0u;
    const uint32_t tail = write ? rw - head - 4 * pieces : 0u;
    if (write) {
        uint32_t hw[3], tw[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            hw[j] = (uint32_t)j < head ? res_src[j] : 0u;
            tw[j] = (uint32_t)j < tail ? res_src[head + 4 * pieces + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if ((uint32_t)j < head)
                res_dst[j] = hw[j];
            if ((uint32_t)j < tail)
                res_dst[head + 4 * pieces + j] = tw[j];
        }
    }
    // The lanes split into teams, one per subframe (four lanes each for the sixteen subframes of eight stereo frames);
    // a team copies its subframe's pieces in turn, every lane with sixteen loads in flight.  (First version: one flat
    // list of pieces over all subframes, looked up in an LDS table -- the lookups, two dependent LDS reads per piece,
    // took longer than the copies: 18 us per group.)
    uint32_t team = 64;
    while (team > 1 && 64u / team < n_sub)
        team >>= 1; // lanes per subframe: the largest power of two with 64 / team >= n_sub
    const int owner = (int)((uint32_t)lane / team); // the lane that did this subframe's headers above
    const uint32_t l = (uint32_t)lane % team;
    const uint64_t src0 = (uint64_t)__shfl((unsigned long long)(uintptr_t)(res_src + head), owner, 64);
    const uint64_t dst0 = (uint64_t)__shfl((unsigned long long)(uintptr_t)(res_dst + head), owner, 64);
    const uint32_t mine = (uint32_t)__shfl((int)pieces, owner, 64); // (0 beyond the last subframe: those lanes' `pieces` are 0)
    uint32_t most = mine;
    for (int d = 32; d >= 1; d >>= 1)
        most = max(most, (uint32_t)__shfl_xor((int)most, d, 64));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    typedef u32x4 __attribute__((address_space(1))) * GlobalPieces;
    constexpr int kFlight = 16;
    for (uint32_t k0 = 0; k0 < most; k0 += team * kFlight) { // (wave-uniform trip count)
        u32x4 v[kFlight];
#pragma unroll
        for (int u = 0; u < kFlight; u++) {
            const uint32_t k = k0 + l + team * u;
            v[u] = u32x4{ 0, 0, 0, 0 };
            if (k < mine) {
                const GlobalWords from = (GlobalWords)(uintptr_t)src0 + 4 * (size_t)k;
                v[u] = u32x4{ from[0], from[1], from[2], from[3] };
            }
        }
#pragma unroll
        for (int u = 0; u < kFlight; u++) {
            const uint32_t k = k0 + l + team * u;
            if (k < mine)
                *((GlobalPieces)(uintptr_t)dst0 + k) = v[u];
        }
    }
}

// Count one finished block into its group; true for the block that completes the group.  The counter carries 40
// bits of the launch's tag, so nothing is cleared between launches (a stale or uninitialised counter reads as
// "nobody yet"; whatever lay in the workspace before -- another process's cells, Rice words of a launch with another
// layout -- matches a tag with probability 2^-40 here and 2^-62 .. 2^-64 in the other cells).
__device__ __forceinline__ bool group_arrive(uint64_t* cell, uint64_t tag, uint32_t blocks_in_group)
{
    // (agent scope: where the group's blocks run is not ours to rely on.  One counter per 128-byte line -- sixteen to
    // a line, the 11,625 updates of a 3,875-frame launch added 77 us to it.)
    const uint64_t mark = count_mark(tag); // (blocks_in_group <= 8 * 256 < 2^24)
    uint64_t old = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const uint64_t neu = (old & ~(uint64_t)0xFFFFFF) == mark ? old + 1 : (mark | 1u);
        if (__hip_atomic_compare_exchange_strong(cell, &old, neu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return (uint32_t)(neu & 0xFFFFFF) == blocks_in_group;
    }
}

__device__ __attribute__((noinline)) void finish_group(const FuseArgs& fa, uint32_t g, int lane)
{
    // (a chain of memory round trips that later groups and the end of the launch wait for: ahead of the co-resident
    // blocks' throughput-bound phases)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t f0 = g * kGroupFrames;
    const uint32_t nfg = min((uint32_t)kGroupFrames, fa.n_frames - f0);
    const uint32_t n_groups = (fa.n_frames + kGroupFrames - 1) / kGroupFrames;
    // ---- sizes ----
    uint32_t size = 0, choice = 0, flags = 0;
    if ((uint32_t)lane < nfg) {
        const uint32_t words = frame_words(fa.meta + (size_t)(f0 + lane) * fa.n_sig, fa.channels, choice, flags);
        size = sela_frame_bytes(fa.channels, words);
    }
    const uint32_t excl = wave_exclusive_scan(size, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(excl + size), 63);
    flags = wave_or(flags);
    GroupState* const st = fa.group_state;
    const uint64_t mark = (fa.tag * 0x9E3779B97F4A7C15ull) & 0xFFFFFFFF00000000ull; // (what group_arrive uses, 32 bits of it)
    if (lane == 0) {
        store_through(&st[g].agg_bytes, mark | total);
        store_through(&st[g].agg_flags, mark | flags);
    }
    // ---- decoupled look-back: sum the groups before this one, nearest first, until one knows its prefix ----
    uint64_t base = 0;
    uint32_t over_before = 0, timed_out = 0;
    bool reached_start = true;
    for (int64_t p = (int64_t)g - 1; p >= 0; p -= 64) {
        const int64_t mine = p - lane;
        uint32_t kind = 0, info = 0;
        uint64_t bytes = 0;
        if (mine >= 0) {
            for (uint32_t spins = 0;;) {
                // (all five at once: one trip to memory per poll)
                const uint64_t lo = load_through(&st[mine].prefix_lo), hi = load_through(&st[mine].prefix_hi),
                               pi = load_through(&st[mine].prefix_info);
                const uint64_t ab = load_through(&st[mine].agg_bytes), af = load_through(&st[mine].agg_flags);
                if ((lo & 0xFFFFFFFF00000000ull) == mark && (hi & 0xFFFFFFFF00000000ull) == mark && (pi & 0xFFFFFFFF00000000ull) == mark) {
                    kind = 2;
                    bytes = (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
                    info = (uint32_t)pi;
                    break;
                }
                if ((ab & 0xFFFFFFFF00000000ull) == mark && (af & 0xFFFFFFFF00000000ull) == mark) {
                    kind = 1;
                    bytes = (uint32_t)ab;
                    info = (uint32_t)af & 0xFFu;
                    break;
                }
                if (++spins > kFuseSpinLimit) {
                    timed_out = 1;
                    break;
                }
                // (each poll is five loads from memory per lane; up to a hundred groups wait at a time)
                __builtin_amdgcn_s_sleep(64);
            }
        }
        const uint64_t knows = __ballot(kind == 2u);
        const int first = knows ? __builtin_ctzll(knows) : 64;
        const bool take = mine >= 0 && lane <= first;
        base += wave_sum_u64(take ? bytes : 0);
        flags |= wave_or(take ? (info & 0xFFu) : 0u);
        over_before += wave_sum_small(take ? (info >> 8) : 0u);
        if (knows) {
            reached_start = false;
            break;
        }
    }
    if (reached_start && fa.pos_in) // (the launch before this one on the stream left its end here)
        base += *fa.pos_in;
    timed_out = wave_or(timed_out);
    if (timed_out)
        flags |= SELA_HIP_FLAG_INTERNAL;
    // ---- this group's frames in the stream ----
    const uint64_t begin = base + excl, end = begin + size;
    const bool fits = end <= fa.frames_cap;
    const uint32_t over = over_before + (uint32_t)__popcll(__ballot((uint32_t)lane < nfg && !fits));
    if (lane == 0) {
        const uint64_t behind = base + total;
        store_through(&st[g].prefix_lo, mark | (uint32_t)behind);
        store_through(&st[g].prefix_hi, mark | (uint32_t)(behind >> 32));
        store_through(&st[g].prefix_info, mark | ((flags & 0xFFu) | (over << 8)));
    }
    if ((uint32_t)lane < nfg) {
        if (fa.frame_offsets) {
            fa.frame_offsets[f0 + lane + 1] = end;
            if (f0 + lane == 0)
                fa.frame_offsets[0] = begin;
        }
        if (fa.mirror) {
            fa.mirror[f0 + lane + 1] = end;
            if (f0 + lane == 0)
                fa.mirror[0] = begin;
        }
    }
    if (g + 1 == n_groups && lane == 0) { // (every group before this one has been counted in: the launch's verdict)
        fa.status[0] = flags & 0xFFu;
        fa.status[1] = over;
        fa.status[2] = fa.status[3] = 0;
        if (fa.mirror)
            fa.mirror[(size_t)fa.n_frames + 1] = (uint64_t)(flags & 0xFFu) | ((uint64_t)over << 32);
        if (fa.pos_out)
            *fa.pos_out = base + total;
    }
    // ---- the bytes ----
    if (nfg * fa.channels <= 64u) {
        if (!timed_out)
            assemble_group(fa, f0, nfg, choice, begin, fits, lane);
    } else
    for (uint32_t i = 0; i < nfg; i++) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)begin, (int)i);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(begin >> 32), (int)i);
        const bool ok = __builtin_amdgcn_readlane((int)fits, (int)i) != 0;
        const uint32_t ch = (uint32_t)__builtin_amdgcn_readlane((int)choice, (int)i);
        if (ok && !timed_out)
            assemble_frame(fa, f0 + i, ch, reinterpret_cast<uint32_t*>(fa.frames + (((uint64_t)hi << 32) | lo)), lane);
    }
    if (fa.groups_done && lane == 0)
        (void)group_arrive(fa.groups_done, fa.tag, n_groups);
}

// kMode: 0 = product, 1 = also write the analysis trace, 2 = also write per-phase cycle counts, 3 = product + two hash words per block (hash_term)
// (debug hook sela_hip_debug_phase_buffer; 16 uint64 per block).
#define SELA_STAMP(n)                 \
    do {                              \
        if (kMode == 2)               \
            stamp[n] = clock64();     \
    } while (0)

// The priorities (s_setprio) of a wave of k_encode_teams in the four quarters of its work, one byte each from the low end;
// sela_hip_debug_priorities sets them (device-wide, measurements).  Default: none -- see the note at the autocorrelation's loop.

// kFused: the host pipeline's one-launch form (await_frame, finish_group); compiled out of the device-pointer path's kernel
template <int kMode, bool kFused>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_encode_blocks(const int16_t* __restrict__ pcm, uint32_t n_frames,
    uint32_t channels, uint32_t n_sig, BlockMeta* __restrict__ meta, uint32_t* __restrict__ slots,
    double* __restrict__ rings, uint32_t* __restrict__ ring_owner, uint32_t ticket, sela_hip_trace* __restrict__ trace,
    uint64_t* __restrict__ phase_cycles, int force_plain_fir, double* __restrict__ mean_out, uint64_t* __restrict__ mean_ready,
    uint32_t n_workers, uint32_t self_blocks, uint32_t total_e, const FuseArgs fa)
{
    constexpr bool kTrace = kMode == 1;
    if (blockIdx.x < n_workers) { // (the first workgroups of the launch: see mean_worker)
        mean_worker(pcm, n_frames, channels, n_sig, worker_first_frame(self_blocks, n_sig) + blockIdx.x * worker_frames_per_wave(n_sig), mean_out, mean_ready, fa.tag);
        return;
    }
    long long stamp[14];
    SELA_STAMP(0);
    __shared__ __attribute__((aligned(16))) unsigned char big[kBigBytes];
    __shared__ uint32_t cw_buf[kCoefWordsCap];

    const int lane = threadIdx.x;

    const uint32_t e = blockIdx.x - n_workers; // encode index
    uint32_t frame, sig;
    block_of(e, n_sig, frame, sig);
    if (frame >= n_frames)
        return;
    const uint32_t block_id = frame * n_sig + sig;
    uint32_t flags = 0;
    if (kFused && fa.pcm_ready && !await_frame(fa.pcm_ready, pcm, frame, ticket, fa.nap_limit))
        flags |= SELA_HIP_FLAG_INTERNAL;

    double* const E = reinterpret_cast<double*>(big) + kPadC;              // first half: E[-64 .. 575]
    double* const O = reinterpret_cast<double*>(big) + kParityLen + kPadC; // first half: O[-64 .. 575]
    double* const E1 = E - kHalfEntries;                                   // second half: E1[448 .. 1087]
    double* const O1 = O - kHalfEntries;
    SmallArrays* const sm = reinterpret_cast<SmallArrays*>(big + kSmallBase);

    // ---- load this signal: s[t] = sample lane + 64 t  (coalesced) ---------------------------------
    int32_t s[kPerLane];
    {
        const int16_t* fp = pcm_after_wait(pcm) + (size_t)frame * kBlock * channels;
        if (channels == 2) {
            const uint32_t* fp2 = reinterpret_cast<const uint32_t*>(fp);
#pragma unroll
            for (int t = 0; t < kPerLane; t++) {
                // (behind the stagers: past the L2 like await_frame's check -- an acquire fence here instead, one per block,
                // cost the 3,875-frame call 0.24 ms)
                const uint32_t w = (kFused && fa.pcm_ready) ? load_through(fp2 + lane + 64 * t) : fp2[lane + 64 * t];
                const int32_t l = (int16_t)(w & 0xFFFFu), r = (int16_t)(w >> 16);
                s[t] = sig == 0 ? l : (sig == 1 ? r : l - r); // src/frame/frame_encoder.cpp:22-24
            }
        } else {
#pragma unroll
            for (int t = 0; t < kPerLane; t++)
                s[t] = fp[(size_t)(lane + 64 * t) * channels + sig];
        }
    }

    // ---- quantizeSamples (src/lpc/residue_generator.cpp:12-18): x = s / 32767 ----------------------
    // sample i = lane + 64 t has parity lane & 1 and half-index (lane >> 1) + 32 t.
    double* const mine = (lane & 1) ? O : E;    // first half
    double* const mine1 = (lane & 1) ? O1 : E1; // second half
    const int half = lane >> 1;
    // The mean: from a mean worker if this block starts late enough for one to have run (bounded wait) ...
    double mean = 0.0;
    bool have_mean = false;
    if (e >= self_blocks) {
        for (uint32_t spin = 0; spin < kMeanWaitSpins; spin++) {
            // (polled relaxed: an acquire per poll invalidates the vector cache under every co-resident wave)
            if (__hip_atomic_load(mean_ready + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fa.tag) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // (the mean is loaded past the L2 as well: order only)
                have_mean = true;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        if (have_mean)
            mean = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<uint64_t*>(mean_out) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        have_mean = __builtin_amdgcn_readfirstlane((int)have_mean) != 0;
        mean = read_first_lane(mean);
    }
    SELA_STAMP(1);
    if (!have_mean) { // ... else the chain, walked here
    for (int m = lane; m < kPadC; m += 64) { // zero pads in front of both parity arrays
        E[m - kPadC] = 0.0;
        O[m - kPadC] = 0.0;
    }
#pragma unroll
    for (int t = 0; t <= kLastT0; t++)
        mine[half + 32 * t] = scale_sample(s[t]);
    wave_sync();

    // ---- mean (src/lpc/residue_generator.cpp:27-30): one strictly sequential sum -----------------
    // Every lane walks the same chain from broadcast LDS reads, so the result is wave-uniform.
    // (a pure dependency chain: run it at raised wave priority so that its adds issue the moment
    // they are ready instead of queueing behind the co-resident wave's throughput-bound phases)
    // The operands are fetched one half-trip (16 values) ahead by hand, like the autocorrelation's:
    // left to the compiler each batch of reads is issued only after the previous batch's last add.
    __builtin_amdgcn_s_setprio(3);
    double sum = 0.0;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        if (h == 1) { // the chain has consumed entries 0 .. 511: bring in the second half
            wave_sync();
#pragma unroll
            for (int t = kFirstT1; t < kPerLane; t++)
                mine1[half + 32 * t] = scale_sample(s[t]);
            wave_sync();
        }
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)E; // entry lo of either half
        MeanFetch f0, f1;
        SELA_MEAN_ISSUE(f0, addr, 0, sum);
#pragma unroll 1
        for (int m0 = 0; m0 < kHalfEntries; m0 += 16) {
            mean_wait(f0, sum);
            SELA_MEAN_ISSUE(f1, addr, 64, sum);
            mean_steps(f0, sum);
            mean_wait(f1, sum);
            SELA_MEAN_ISSUE(f0, addr, 128, sum); // (the last one reads 8 values past the half: in bounds, unused)
            mean_steps(f1, sum);
            addr += 128;
        }
        mean_wait(f0, sum);
    }
    mean = sum / (double)kBlock;
    __builtin_amdgcn_s_setprio(0);
    }

    SELA_STAMP(2);
    // c[j] = x[j] - mean (same value at every use): the first half again
    wave_sync();
    for (int m = lane; m < kPadC; m += 64) { // (the second half of x overwrote the zero pads)
        E[m - kPadC] = 0.0;
        O[m - kPadC] = 0.0;
    }
#pragma unroll
    for (int t = 0; t <= kLastT0; t++)
        mine[half + 32 * t] = scale_sample(s[t]) - mean;
    wave_sync();

    SELA_STAMP(3);
    // ---- autocorrelation (src/lpc/residue_generator.cpp:33-38) -------------------------------------
    // Lane L owns lags 2L and 2L+1 (lanes 0..50 matter).  At step j it needs
    //     A = c[j - 2L]      (lag 2L)       B = c[j - 2L - 1]   (lag 2L+1)
    // and the wave-uniform c[j].  Going to j+1: B' = A, A' = c[j+1-2L], whose parity is that of j+1
    // for every lane -> one conflict-free LDS read per step.  c[j] reaches the multiplies as a SCALAR
    // operand: the wave copies its centred samples, one ring half (kRingHalf) at a time, into a ring in global memory that
    // it alone uses (L2-resident, see kRingLen) and reads them back with s_load_dwordx16.  Each
    // accumulator sees its products in ascending j exactly like the reference loop; the extra
    // leading terms c[j]*0 (j < lag) leave an accumulator at +0.0.
    double acc_e = 0.0, acc_o = 0.0;
    {
        // borrow a ring from this XCD's pool (only workgroups on the same XCD, i.e. behind the same L2,
        // ever touch a pool: no cross-L2 coherence is needed for the rings or for their owner words)
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= kXcds - 1;
        uint32_t* const pool = ring_owner + xcc * kRingsPerXcd;
        uint32_t slot = 0;
        if (lane == 0) {
            // workgroups that run together have nearby indices within their XCD: each starts at its own
            // ring (index mod 512), so that in the common case one atomic takes a ring without a retry
            slot = (e >> 3) & (kRingsPerXcd - 1);
            while (atomicExch(pool + slot, ticket) == ticket)
                slot = (slot + 1) & (kRingsPerXcd - 1);
        }
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        double* const ring = reinterpret_cast<double*>(
            read_first_lane((uint64_t)(rings + ((size_t)xcc * kRingsPerXcd + slot) * kRingLen)));
        __attribute__((address_space(1))) double* const ring_g = (__attribute__((address_space(1))) double*)ring;
        // samples [kRingHalf k, kRingHalf (k + 1)) -> half k & 1 (natural order; lane's element i = lane + 64 t)
        auto store_chunk = [&](int k, const double* resident) { // resident = mine or mine1, whichever holds chunk k
#pragma unroll
            for (int t = 0; t < kRingHalf / 64; t++) {
                const double c = resident[half + 32 * (t + (kRingHalf / 64) * k)];
                ring_g[(k & 1) * kRingHalf + lane + 64 * t] = c;
                if (t == 0 && !(k & 1) && lane < 16)
                    ring_g[2 * kRingHalf + lane] = c; // the 16 values a fetch running off the end of half B must find
            }
        };
        store_chunk(0, mine);
        store_chunk(1, mine);
        // the stores are in L2 once vmcnt drains (the scalar loads are glc: they bypass the scalar
        // cache, which may hold older contents of the ring)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv" ::: "memory");

        double A = E[-lane]; // c[0 - 2L]
        double B = 0.0;      // c[-2L - 1]
        // LDS byte addresses of E[m0 - L] and O[m0 - L]; ring pointer at c[2 m0]
        uint32_t addr_e = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)(E - lane);
        uint32_t addr_o = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)(O - lane);
        const double* c_ptr = ring;
        AcFetch f0, f1;
        asm volatile("" : "+v"(A)); // land A here: a compiler-placed wait inside the loop would drain the prefetch
        SELA_AC_ISSUE_S(f0, c_ptr, 0, 64, acc_e, acc_o);
        SELA_AC_ISSUE_V(f0, addr_e, addr_o, 0, 1, 2, 3, 4, 5, 6, 7, 8, acc_e, acc_o);
        // Two 16-step trips per iteration, kRingHalf / 32 iterations per chunk.  A fetch's scalar half is
        // issued a whole trip ahead, its vector half after the first two steps of the running trip,
        // once the registers it overwrites are dead.  (The last fetch of the block lands behind the
        // ring's half B and in LDS behind the parity arrays; it is never used.)
        constexpr int kSwitch = kBlock / 2 / kRingHalf; // chunk at which the second half of c[] takes over
        // (the hook of k_encode_teams' note on priorities, for a launch of at most one fill of this kernel: 2 and 1 through the
        // autocorrelation, 1 through the Schur recursion, 0 behind it.  1000 frames on their own: 0.158 -> 0.148 ms; off by default
        // for the same reason -- beside another stream's kernels it costs what it gains here)
        const bool falling = !kFused && n_workers == 0 && fa.priorities != 0; // (the host pipeline runs beside its stagers and copies: never alone)
#pragma unroll 1
        for (int k = 0; k < kBlock / kRingHalf; k++) {
            if (falling && k == 0)
                __builtin_amdgcn_s_setprio(2);
            if (falling && k == kSwitch)
                __builtin_amdgcn_s_setprio(1);
            if (k == kSwitch) {
                // steps 0 .. 1023 are done, the window fetch in flight was issued from the first half
                // (LDS serves a wave's reads and writes in order): recompute c[896 ..] over it
                wave_sync();
#pragma unroll
                for (int t = kFirstT1; t < kPerLane; t++)
                    mine1[half + 32 * t] = scale_sample(s[t]) - mean;
                wave_sync();
                addr_e -= kHalfEntries * 8;
                addr_o -= kHalfEntries * 8;
            }
            if (k >= 1 && k + 1 < kBlock / kRingHalf)
                store_chunk(k + 1, k >= kSwitch ? mine1 : mine); // over chunk k - 1, which is consumed
#pragma unroll 1
            for (int it = 0; it < kRingHalf / 32; it++) {
                if (it == kRingHalf / 32 - 1) // this iteration's prefetch crosses into chunk k + 1
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv" ::: "memory");
                ac_wait(f0, acc_e, acc_o);
                SELA_AC_ISSUE_S(f1, c_ptr, 128, 192, acc_e, acc_o);
                ac_steps_head(f0, A, B, acc_e, acc_o);
                SELA_AC_ISSUE_V(f1, addr_e, addr_o, 8, 9, 10, 11, 12, 13, 14, 15, 16, acc_e, acc_o);
                ac_steps_tail(f0, A, B, acc_e, acc_o);
                ac_wait(f1, acc_e, acc_o);
                SELA_AC_ISSUE_S(f0, c_ptr, 256, 320, acc_e, acc_o);
                ac_steps_head(f1, A, B, acc_e, acc_o);
                SELA_AC_ISSUE_V(f0, addr_e, addr_o, 16, 17, 18, 19, 20, 21, 22, 23, 24, acc_e, acc_o);
                ac_steps_tail(f1, A, B, acc_e, acc_o);
                c_ptr += 32;
                addr_e += 128;
                addr_o += 128;
            }
            if (k & 1)
                c_ptr = ring; // half B is followed by half A (the fetch in flight came from the copy behind B)
        }
        ac_wait(f0, acc_e, acc_o); // drain the fetch past the end
        if (lane == 0)
            atomicExch(pool + slot, 0u); // hand the ring back (0 is never a ticket)
    }
    wave_sync(); // c[] is dead from here on

    SELA_STAMP(4);
    // normalise (src/lpc/residue_generator.cpp:41-44)
    {
        const double ac0 = read_first_lane(acc_e);
        if (lane <= 50) {
            sm->ac[2 * lane] = lane == 0 ? 1.0 : acc_e / ac0;
            sm->ac[2 * lane + 1] = acc_o / ac0;
        }
    }
    wave_sync();

    // ---- Schur recursion (src/lpc/residue_generator.cpp:47-68), always 100 stages -------------------
    // Column j of gen[0]/gen[1] lives in lane j (j < 64) and lane j - 64 of a second register.
    // Stage i reads gen1[j+1] (old) -> a one-lane shift; all columns update from old values.
    double k_lo = 0.0, k_hi = 0.0; // k[lane], k[lane + 64]
    if (!kFused && n_workers == 0 && fa.priorities != 0)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(2); // latency-bound (100 dependent stages, one division each)
    {
        double g0a = sm->ac[lane + 1], g1a = g0a;
        double g0b = lane < 36 ? sm->ac[lane + 65] : 0.0, g1b = g0b;
        double err = 1.0; // ac[0]
        // k_i is wave-uniform: every lane stores the same value to sm->k[i] (one LDS write) instead of
        // selecting it into the lane that keeps it (two compares + four selects per stage)
        double g = read_first_lane(g1a);
        double ki = -g / err;
        err += g * ki;
        sm->k[0] = ki;
        // stage i only needs columns j < 100 - i: the second register (columns 64..99) is dead from
        // stage 37 on, and with it the value shifted into lane 63
        constexpr int kTwoRegs = kMaxOrder - 64 + 1; // 37
#pragma unroll 1
        for (int i = 1; i < kTwoRegs; i++) {
            const double sa = wave_shl1(read_first_lane(g1b), g1a); // gen1[j+1], j = lane
            const double sb = wave_shl1_zero(g1b);                  // gen1[j+1], j = lane + 64
            g1a = sa + ki * g0a;
            g0a = sa * ki + g0a;
            g1b = sb + ki * g0b;
            g0b = sb * ki + g0b;
            g = read_first_lane(g1a);
            ki = -g / err;
            err += g * ki;
            sm->k[i] = ki;
        }
#pragma unroll 1
        for (int i = kTwoRegs; i < kMaxOrder; i++) {
            const double sa = wave_shl1_zero(g1a);
            g1a = sa + ki * g0a;
            g0a = sa * ki + g0a;
            g = read_first_lane(g1a);
            ki = -g / err;
            err += g * ki;
            sm->k[i] = ki;
        }
        wave_sync();
        k_lo = sm->k[lane];
        k_hi = lane < kMaxOrder - 64 ? sm->k[lane + 64] : 0.0;
        wave_sync(); // (sm->k is overwritten with the dequantised coefficients below)
    }
    if (kMode == 3) { // (hash_term: the product kernel + these lines)
        uint64_t h_ac = hash_term(sm->ac[lane], lane) ^ (lane + 64 <= kMaxOrder ? hash_term(sm->ac[lane + 64], lane + 64) : 0ull);
        uint64_t h_k = hash_term(k_lo, lane) ^ (lane < kMaxOrder - 64 ? hash_term(k_hi, lane + 64) : 0ull);
        h_ac = lanes_xor<64>(h_ac);
        h_k = lanes_xor<64>(h_k);
        if (lane == 0) {
            uint64_t* const hp = reinterpret_cast<uint64_t*>(trace) + 2 * (size_t)block_id;
            hp[0] = h_ac;
            hp[1] = h_k;
        }
    }

    {
        constexpr bool kAcInLds = true;
        uint64_t* const sizes_pub = fa.sizes_pub;
        const uint64_t sizes_tag = fa.sizes_tag;
#define SELA_TAIL_TRACE_RESIDUES fa.trace_residues
#include "sela_encode_tail.inc"
#undef SELA_TAIL_TRACE_RESIDUES
    }
    SELA_STAMP(12);
    // ---- host pipeline: count this block into its group; the group's last block places and writes the group's frames
    if (kFused) {
        const uint32_t g = frame / kGroupFrames;
        const uint32_t blocks_in_group = min((uint32_t)kGroupFrames, n_frames - g * kGroupFrames) * n_sig;
        stores_done(); // the slot and the BlockMeta of every lane, before the count
        uint32_t last = 0;
        if (lane == 0)
            last = group_arrive(fa.group_count + (size_t)g * kGroupCountStride, fa.tag, blocks_in_group) ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane((int)last)) {
            drop_stale_lines(); // (the other blocks' slots and metas are in memory: nothing older may come from this CU's cache)
            finish_group(fa, g, lane);
        }
    }
    SELA_STAMP(13);
    if (kMode == 2 && lane == 0)
        for (int i = 0; i < 13; i++)
            phase_cycles[(size_t)block_id * 16 + i] = (uint64_t)(stamp[i + 1] - stamp[i]);
}

// ---- k_encode_teams: the analysis of SEVERAL blocks side by side in one wave (round 4) ------------------------------
// k_encode_blocks gives a block a whole wave and keeps 51 of its 64 lanes busy through the autocorrelation (lane = a pair
// of lags), all 64 through 100 divisions that serve one block, and the sequential mean of a block costs it 2048 adds on
// every lane.  At the issue rate the kernel runs at, only fewer instructions help -- so here a wave takes B = 64 / P
// blocks at once, P lanes (a "team") per block:
//   autocorrelation   lane p of a team owns the G = ceil(101 / P) CONSECUTIVE lags G p .. G p + G - 1: G accumulators, and a
//                     window of the block's centred samples c[j - G p - g] that slides through 16 registers (static
//                     names: the loop is unrolled by 16 steps).  Per step and lane: one new window value and the team's
//                     multiplier c[j], both read from LDS three steps ahead, and 2 G multiply / add instructions -- for B
//                     blocks.  P = 8: 26 instructions per step for 8 blocks (3.25 per block and step, against 4); 104 of
//                     104 lag slots used (k_encode_blocks: 102 of 128).  Every accumulator still sees its products in
//                     ascending j, each product rounded before it is added (src/lpc/residue_generator.cpp:33-38).
//                     The samples of B blocks do not fit LDS as FP64, and need not: a ring of 160 per block (the lags
//                     reach 103 back, a chunk of 16 is being consumed, the next one is being written; P = 16: 256 and
//                     64) is refilled from the PCM chunk by chunk -- L2 hits, no scalar-operand scratch in global
//                     memory at all.
//   mean              (:27-30) one chain per block as before, but the chains of the wave's B blocks advance together:
//                     2048 adds per WAVE, on samples the lanes converted side by side.  No mean workers.
//   Schur recursion   (:47-68) lane p holds columns G p .. G p + G - 1 of gen0 / gen1: the shift gen1[j+1] is the next
//                     register (one DPP move per stage for the column that crosses to the next lane), and the division of a
//                     stage serves B blocks.
// Then the wave takes its blocks one after the other through the same tail as k_encode_blocks (sela_encode_tail.inc):
// all 64 lanes on one block's order / quantisation / step-up / residues / Rice streams.
// What this costs is granularity: a wave is B blocks' worth of work (P = 8: ~100 k instructions, ~0.2 ms), so launches
// that do not fill the device several times over keep k_encode_blocks (launch_encode picks).
constexpr int kTeamMirror = 24;                             // the ring's first entries again behind its last: a run of <= 19 reads may start at any entry
constexpr int kTeamWin = 16;                                // window registers (a circular file: >= G + kTeamAhead)
constexpr int kTeamAhead = 3;                               // steps a fetch runs ahead of its use
constexpr int kTeamMeanChunk = 64;                          // the mean's chunks (two of them in a ring)

template <int P>
struct TeamPlan {
    static constexpr int B = kWave / P;                        // blocks per wave
    static constexpr int G = (kMaxOrder + 1 + P - 1) / P;      // lags (Schur: columns) per lane
    // samples of a block staged at a time / resident in LDS (positions = sample index mod kRing).  The ring must hold the
    // lags' reach, the chunk in use and the chunk being written; eight rings at once (P = 8) only fit a third of a CU's LDS
    // share with the smallest chunk, one 16-step trip
    static constexpr int kChunk = P == 8 ? 16 : 64;
    static constexpr int kRing = P == 8 ? 160 : 256;
    static constexpr int kStride = P == 8 ? kRing + kTeamMirror : kRing + 48; // doubles; 184 is 24 mod 32, 304 is 16 mod 32: consecutive rings start 48 / 32 banks apart (team_ac_steps)
    static constexpr int kPer = kChunk / P;                    // samples a lane stages per chunk
    static constexpr int kMeanPer = kTeamMeanChunk / P;
    static constexpr int kRingBytes = B * kStride * 8;
    static constexpr int kCwBase = kBigBytes;                  // the tail's coefficient words behind its LDS plan (k_encode_blocks: a second array)
    static constexpr int kKStride = 104;                       // doubles per block: its reflection coefficients, in the dead rings until they are quantised
    static constexpr int kQBase = kSmallBase;                  // the quantised coefficients of the wave's blocks wait for their tails where the tail's plan keeps
    static constexpr int kQStride = 104;                       //   ac[] for k_encode_blocks (832 bytes it does not use here): 100 x int8, the order, an escape mark
    static constexpr int kLdsBytes = kRingBytes > kCwBase + kCoefWordsCap * 4 ? kRingBytes : kCwBase + kCoefWordsCap * 4;
    static_assert(G + kTeamAhead <= kTeamWin && kChunk % kTeamWin == 0 && kRing % kChunk == 0 && kStride % 32 == (P == 8 ? 24 : 16) && kStride >= kRing + kTeamMirror, "team plan");
    static_assert((G * P - 1) + kChunk + kChunk <= kRing && 2 * kTeamMeanChunk <= kRing, "the ring must hold the lags' reach, the chunk in use and the chunk being written");
    static_assert(kLdsBytes * 12 <= 160 * 1024, "twelve waves per CU");
    static_assert(B * kKStride * 8 <= kQBase && B * kKStride * 8 <= kRingBytes && B * kQStride <= 104 * 8, "k[] in the dead rings below the q store; the q store inside SmallArrays::ac");
};

// every lane receives the value of lane 0 of its team
template <int P>
__device__ __forceinline__ uint32_t team_first(uint32_t v)
{
    if constexpr (P == 16) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 /* row_newbcast:0 */, 0xf, 0xf, true);
    } else {
        static_assert(P == 8, "teams of 8 or 16 lanes");
        const int q = __builtin_amdgcn_mov_dpp((int)v, 0x00 /* quad_perm:[0,0,0,0] */, 0xf, 0xf, true);       // lanes 4 k .. 4 k + 3 <- lane 4 k
        return (uint32_t)__builtin_amdgcn_update_dpp(q, q, 0x114 /* row_shr:4 */, 0xf, 0xa /* banks 1, 3 */, false); // lanes 4..7 <- 0..3, 12..15 <- 8..11
    }
}
template <int P>
__device__ __forceinline__ double team_first(double v)
{
    const uint64_t x = __builtin_bit_cast(uint64_t, v);
    return __builtin_bit_cast(double, ((uint64_t)team_first<P>((uint32_t)(x >> 32)) << 32) | team_first<P>((uint32_t)x));
}

// the samples first, first + kStep, .. (kPer of them) of signal `sig` of the frame at fp, as the integers the reference analyses
// (src/frame/frame_encoder.cpp:22-24 for the difference signal).  kStep = P (teams of 16): the lanes of a team take neighbouring samples, so
// their ds_write_b64 into the ring fall on 16 different bank pairs (with kPer consecutive samples per lane, a stride of 2 kPer
// dwords, a 16-lane store group is 4-way conflicted: SQ_LDS_BANK_CONFLICT, 3072 of the 7168 extra LDS cycles per wave of P = 16).
// Teams of 8 keep kStep = 1 and their vector loads (two teams to a store group, 2-way at worst; the strided form costs that
// kernel three more spilled registers).
template <int kPer, int kStep>
__device__ __forceinline__ void team_load_raw(const int16_t* __restrict__ fp, uint32_t channels, uint32_t sig, int first, int32_t (&raw)[kPer])
{
    if (channels == 2) {
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(fp) + first;
        uint32_t w[kPer];
        if constexpr (kStep == 1 && kPer % 4 == 0) { // (dword-aligned: unaligned vector loads are fine)
#pragma unroll
            for (int u = 0; u < kPer / 4; u++) {
                const uint4 v = reinterpret_cast<const uint4*>(pw)[u];
                w[4 * u] = v.x, w[4 * u + 1] = v.y, w[4 * u + 2] = v.z, w[4 * u + 3] = v.w;
            }
        } else if constexpr (kStep == 1 && kPer % 2 == 0) {
#pragma unroll
            for (int u = 0; u < kPer / 2; u++) {
                const uint2 v = reinterpret_cast<const uint2*>(pw)[u];
                w[2 * u] = v.x, w[2 * u + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < kPer; u++)
                w[u] = pw[u * kStep];
        }
        const uint32_t first_shift = sig == 1 ? 16u : 0u;     // signal 0: l, 1: r, 2: l - r  as  a - (b & mask)
        const uint32_t second_mask = sig == 2 ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int32_t a = (int32_t)(w[i] << (16 - first_shift)) >> 16, b = ((int32_t)w[i] >> 16) & (int32_t)second_mask;
            raw[i] = a - b;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kPer; i++)
            raw[i] = fp[(size_t)(first + i * kStep) * channels + sig];
    }
}

// x = s / 32767 (kCentre: minus the block's mean) of a lane's kPer samples to ring positions pos, pos + kStep, ..; kMirror: the
// ring's first entries also behind its end
template <int kPer, int kStep, bool kCentre, bool kMirror, int kRing>
__device__ __forceinline__ void team_stage(double* ring_b, int pos, const int32_t (&raw)[kPer], double mean)
{
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const double x = scale_sample(raw[i]);
        const double c = kCentre ? x - mean : x;
        ring_b[pos + i * kStep] = c;
        if (kMirror && pos + i * kStep < kTeamMirror)
            ring_b[kRing + pos + i * kStep] = c;
    }
}

// sum + x[0] + x[1] + ... in that order (the mean's chain); the values come eight at a time, each batch fetched while the
// batch before it is being added (a chain that waits for its own reads walks at the LDS's latency, not at the adder's)
template <int kN>
__device__ __forceinline__ double team_chain(const double* x, double sum)
{
    constexpr int kBatch = 8;
    static_assert(kN % (2 * kBatch) == 0, "two batches per round");
    double u[kBatch], v[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; i++)
        u[i] = x[i];
#pragma unroll
    for (int i0 = 0; i0 < kN; i0 += 2 * kBatch) { // (straight-line code: as a loop the compiler copies one batch into the other's registers)
#pragma unroll
        for (int i = 0; i < kBatch; i++)
            v[i] = x[i0 + kBatch + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kBatch; i++)
            sum += u[i];
        __builtin_amdgcn_sched_barrier(0);
        if (i0 + 2 * kBatch < kN) {
#pragma unroll
            for (int i = 0; i < kBatch; i++)
                u[i] = x[i0 + 2 * kBatch + i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kBatch; i++)
            sum += v[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    return sum;
}

// Steps R0 .. 15 of a 16-step trip of the autocorrelation.  At step j = j0 + r a lane holds c[j - G p - g] in W[(r - g) & 15]
// and adds c[j] * c[j - G p - g] to acc[g], g = 0 .. G - 1; the fetch for step r + 3 (its new window value and its
// multiplier: addr_w / addr_m are the LDS addresses of c[j0 - G p] and c[j0]) is issued first and lands -- the LDS
// returns a wave's reads in order, so "at most six newer ones outstanding" is "the pair of three steps ago is there" --
// right before its first use.  The compiler sees no load of its own in this loop.
// Banks: a ds_read_b64 is served in two halves of 32 lanes = 4 teams of 8 (2 of 16), bank = dword address mod 64.  Within a
// team the lanes read G entries apart: -26 p dwords mod 64 for G = 13 and -14 p for G = 7, all different bank pairs.  Between
// the teams of a half, the rings' distance decides: 184 doubles = 48 dwords mod 64 puts the four teams of 8 on 32 different bank
// pairs; for two teams of 16 the same 48 (a ring of 280) put 8 of the second team's 16 lanes on the first team's pairs -- every
// window read took 4 LDS cycles instead of 2 (SQ_LDS_BANK_CONFLICT 21.0 M of SQ_LDS_IDX_ACTIVE 70.0 M per 3875-frame launch,
// round 6) -- while 32 dwords (a ring of 304: {-14 p} and {32 - 14 p} are the two halves of the even residues) leave none:
// 2.6 M of 51.7 M.  The launch itself went from 362 to 360 us: the LDS was not what it waited for (SQ_WAIT_INST_LDS 3 % of
// the wave cycles), the cycles are simply no longer spent.
template <int G, int R0>
__device__ __forceinline__ void team_ac_steps(double (&W)[kTeamWin], double (&M)[4], double (&acc)[G], uint32_t addr_w, uint32_t addr_m)
{
    asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4"
                 : "=&v"(W[(R0 + kTeamAhead) & (kTeamWin - 1)]), "=&v"(M[(R0 + kTeamAhead) & 3])
                 : "v"(addr_w), "v"(addr_m), "n"(8 * (R0 + kTeamAhead)), "v"(acc[0]), "v"(acc[G - 1])
                 : "memory");
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(W[R0 & (kTeamWin - 1)]), "+v"(M[R0 & 3]));
    const double m = M[R0 & 3];
    // (products a few at a time ahead of their additions: a product directly in front of the addition that needs it stalls a
    // wave that has its SIMD to itself)
    constexpr int kAheadOps = 5;
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += kAheadOps) {
        double t[kAheadOps];
#pragma unroll
        for (int u = 0; u < kAheadOps; u++)
            if (g0 + u < G)
                t[u] = m * W[(R0 - g0 - u) & (kTeamWin - 1)];
#pragma unroll
        for (int u = 0; u < kAheadOps; u++)
            if (g0 + u < G)
                acc[g0 + u] += t[u];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (R0 < kTeamWin - 1)
        team_ac_steps<G, R0 + 1>(W, M, acc, addr_w, addr_m);
}

template <int kMode, int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_encode_teams(
    const int16_t* __restrict__ pcm, uint32_t n_frames, uint32_t channels, uint32_t n_sig, BlockMeta* __restrict__ meta,
    uint32_t* __restrict__ slots, sela_hip_trace* __restrict__ trace, int force_plain_fir, uint64_t* __restrict__ phase_cycles,
    uint64_t* __restrict__ sizes_pub, uint64_t sizes_tag, uint32_t team_priorities)
{
    using Plan = TeamPlan<P>;
    constexpr int B = Plan::B, G = Plan::G, kPer = Plan::kPer, kMeanPer = Plan::kMeanPer, kChunk = Plan::kChunk, kRing = Plan::kRing;
    constexpr bool kTrace = kMode == 1;
    constexpr bool kFused = false;
    constexpr int kStep = P == 16 ? P : 1; // team_load_raw
    __shared__ __attribute__((aligned(32))) unsigned char lds[Plan::kLdsBytes];
    unsigned char* const big = lds; // the tail's LDS plan (k_encode_blocks), over the dead rings
    uint32_t* const cw_buf = reinterpret_cast<uint32_t*>(lds + Plan::kCwBase);

    const int lane0 = threadIdx.x;
    const int b = lane0 / P, p = lane0 % P;
    // Which blocks: signal `sig` of B consecutive frames.  Workgroups that are equal mod 8 share an XCD (as observed; for
    // speed only): the n_sig waves that take the signals of the same frames are 8 workgroups apart, so a frame's PCM comes
    // into one L2.  (A wave's frames a B-th of the batch apart instead -- so that no wave gets B blocks of one passage, all
    // with long predictors -- changed nothing: the waves of a launch do differ by a factor of two in duration, 0.51 .. 0.97 M
    // cycles at 3875 frames, but that is the arbiter's doing, see the note on priorities below.)
    const uint32_t wq = blockIdx.x % 8, wv = blockIdx.x / 8;
    const uint32_t sig = wv % n_sig;
    const uint32_t frame0 = ((wv / n_sig) * 8 + wq) * B;
    constexpr uint32_t frame_stride = 1;
    if (frame0 >= n_frames)
        return;
    const bool team_live = frame0 + (uint32_t)b * frame_stride < n_frames;
    const uint32_t my_frame = team_live ? frame0 + (uint32_t)b * frame_stride : frame0; // (a team beyond the last frame shadows the wave's first and writes nothing)
    const int16_t* const fp = pcm + (size_t)my_frame * kBlock * channels;
    double* const ring_b = reinterpret_cast<double*>(lds) + b * Plan::kStride;
    // The team's reflection coefficients go to the dead rings (LDS), are quantised there block by block right behind the
    // Schur recursion, and 100 bytes per block wait for the block's tail (TeamPlan::kQBase).  (First version: the 800 bytes
    // of k[] waited in the block's own output slot in global memory -- 9.3 MB more written and read back per 3875-frame
    // launch, PMC.)
    double* const k_b = reinterpret_cast<double*>(lds) + b * Plan::kKStride;
    long long wave_stamp[5];
    if (kMode == 2)
        wave_stamp[0] = clock64();

    // ---- the mean (src/lpc/residue_generator.cpp:27-30): x = s / 32767 summed in order, the B chains side by side ----
    // Chunks of 64 samples alternate between two places in the ring; the PCM of a chunk is fetched two chunks ahead.
    double mean;
    {
        constexpr int kChunks = kBlock / kTeamMeanChunk;
        const int mine = kStep == 1 ? p * kMeanPer : p; // this lane's samples of a chunk: mine + kStep i
        int32_t raw_a[kMeanPer], raw_b[kMeanPer];
        team_load_raw<kMeanPer, kStep>(fp, channels, sig, mine, raw_a);
        team_stage<kMeanPer, kStep, false, false, kRing>(ring_b, mine, raw_a, 0.0);
        team_load_raw<kMeanPer, kStep>(fp, channels, sig, kTeamMeanChunk + mine, raw_a);
        team_load_raw<kMeanPer, kStep>(fp, channels, sig, 2 * kTeamMeanChunk + mine, raw_b);
        double sum = 0.0;
        const uint32_t quarter_priorities = team_priorities;
        set_wave_priority((int)(quarter_priorities & 0xFF)); // (see the note on priorities at the autocorrelation's loop)
#pragma unroll 1
        for (int c = 0; c < kChunks; c++) {
            // chunk c + 1 -> the place chunk c - 1 was summed from, chunk c is summed from the other one
            const int here = (c & 1) * kTeamMeanChunk, there = kTeamMeanChunk - here;
            if (c + 1 < kChunks)
                team_stage<kMeanPer, kStep, false, false, kRing>(ring_b, there + mine, raw_a, 0.0);
#pragma unroll
            for (int i = 0; i < kMeanPer; i++)
                raw_a[i] = raw_b[i];
            if (c + 3 < kChunks)
                team_load_raw<kMeanPer, kStep>(fp, channels, sig, (c + 3) * kTeamMeanChunk + mine, raw_b);
            wave_sync();
            sum = team_chain<kTeamMeanChunk>(ring_b + here, sum);
            wave_sync();
        }
        mean = sum / (double)kBlock;
    }
    if (kMode == 2)
        wave_stamp[1] = clock64();

    // ---- autocorrelation (src/lpc/residue_generator.cpp:33-38) ------------------------------------------------------------
    double acc[G];
    {
#pragma unroll
        for (int g = 0; g < G; g++)
            acc[g] = 0.0;
        // "no sample" before the block: every position but chunk 0's stands for a negative index until its chunk arrives
        for (int i = kChunk + p; i < Plan::kStride; i += P)
            ring_b[i] = 0.0;
        const int mine = kStep == 1 ? p * kPer : p; // this lane's samples of a chunk: mine + kStep i
        int32_t raw_a[kPer], raw_b[kPer];
        team_load_raw<kPer, kStep>(fp, channels, sig, mine, raw_a);
        team_stage<kPer, kStep, true, true, kRing>(ring_b, mine, raw_a, mean);
        team_load_raw<kPer, kStep>(fp, channels, sig, kChunk + mine, raw_a);
        team_load_raw<kPer, kStep>(fp, channels, sig, 2 * kChunk + mine, raw_b);
        wave_sync();
        const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)ring_b;
        double W[kTeamWin], M[4];
#pragma unroll
        for (int i = 0; i < kTeamWin; i++)
            W[i] = 0.0; // c[negative]
        uint32_t idx_w = (uint32_t)((kRing - G * p) % kRing); // position of index -G p
        uint32_t idx_m = 0, pos_stage = kChunk;               // positions of index j0 / of the chunk staged next
        uint32_t addr_w = ring_addr + 8 * idx_w, addr_m = ring_addr;
        // the fetches of steps 0, 1, 2
        asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %3, %7\n\t"
                     "ds_read_b64 %1, %6 offset:8\n\tds_read_b64 %4, %7 offset:8\n\t"
                     "ds_read_b64 %2, %6 offset:16\n\tds_read_b64 %5, %7 offset:16"
                     : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(M[0]), "=&v"(M[1]), "=&v"(M[2])
                     : "v"(addr_w), "v"(addr_m)
                     : "memory");
        M[3] = 0.0;
        constexpr int kChunks = kBlock / kChunk;
        auto trips = [&]() { // the kChunk steps of a chunk
#pragma unroll 1
            for (int it = 0; it < kChunk / kTeamWin; it++) {
                team_ac_steps<G, 0>(W, M, acc, addr_w, addr_m);
                idx_w += kTeamWin;
                idx_w = idx_w >= (uint32_t)kRing ? idx_w - kRing : idx_w;
                idx_m += kTeamWin;
                idx_m = idx_m >= (uint32_t)kRing ? idx_m - kRing : idx_m;
                addr_w = ring_addr + 8 * idx_w;
                addr_m = ring_addr + 8 * idx_m;
            }
        };
        auto stage_next = [&](const int32_t (&raw)[kPer]) { // the next chunk over the oldest one, which no lag reaches any more
            team_stage<kPer, kStep, true, true, kRing>(ring_b, (int)pos_stage + mine, raw, mean);
            pos_stage += kChunk;
            pos_stage = pos_stage >= (uint32_t)kRing ? 0u : pos_stage;
        };
        // Priorities (team_priorities, a launch argument; 0 = none).  The SIMD's arbiter serves the OLDEST of its waves first, at equal
        // priority: of three waves that start together, the first keeps nearly the whole SIMD to itself, finishes after 0.51 M
        // cycles and leaves the last one to walk its second half alone, at a lone wave's issue rate, until 0.96 M
        // (tools/ramp_profile.py: a launch that fills the device once takes as long as that last wave).  With a priority that
        // FALLS with a wave's progress (3, 2, 1, 0 by quarters: whoever is ahead yields) the waves of a SIMD finish together,
        // 0.65 .. 0.86 M cycles, and a 3875-frame launch ON ITS OWN takes 0.396 ms instead of 0.427 (12.25 G samples/s
        // strictly serial instead of 11.65) -- but beside another stream's kernels, which is how throughput is had, the same
        // priorities starve the other kernel's waves: 14.97 G samples/s with two batches in flight against 15.71 without
        // (raising the decoder's priority as well: 15.26; milder schedules: 15.3 .. 15.6).  So the schedule is decided per launch
        // (round 5): sela_hip_encode_device passes it when no other stream of the library has work pending, 0 otherwise
        // (sela_capi.hip, "does a device-pointer launch have the device to itself?").
        constexpr int kPrio2From = (int)(0.27 * kChunks), kPrio1From = (int)(0.71 * kChunks);
#pragma unroll 1
        for (int c = 0; c < kChunks; c++) { // (the PCM of a chunk is fetched two chunks ahead)
            if (c == kPrio2From)
                set_wave_priority((int)((team_priorities >> 8) & 0xFF));
            if (c == kPrio1From)
                set_wave_priority((int)((team_priorities >> 16) & 0xFF));
            if (c + 1 < kChunks)
                stage_next(raw_a); // chunk c + 1
#pragma unroll
            for (int i = 0; i < kPer; i++)
                raw_a[i] = raw_b[i];
            if (c + 3 < kChunks)
                team_load_raw<kPer, kStep>(fp, channels, sig, (c + 3) * kChunk + mine, raw_b);
            trips();
        }
        // (three fetches past the end are in flight: land them before the rings are reused)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(M[0]), "+v"(M[1]), "+v"(M[2]));
    }
    wave_sync(); // the rings are dead from here on
    if (kMode == 2)
        wave_stamp[2] = clock64();

    // ---- normalise (src/lpc/residue_generator.cpp:41-44): lane p holds ac[G p + g] ---------------------------------------
    {
        const double ac0 = team_first<P>(acc[0]);
#pragma unroll
        for (int g = 0; g < G; g++)
            acc[g] = (p == 0 && g == 0) ? 1.0 : acc[g] / ac0;
    }
    if (kTrace && team_live) {
        sela_hip_trace* tr = trace + (size_t)my_frame * n_sig + sig;
        if (p == 0)
            tr->mean = mean;
#pragma unroll
        for (int g = 0; g < G; g++)
            if (G * p + g <= kMaxOrder)
                tr->ac[G * p + g] = acc[g];
    }
    if (kMode == 3) { // (hash_term: the product kernel + these lines and their like behind the Schur recursion)
        uint64_t h = 0;
#pragma unroll
        for (int g = 0; g < G; g++)
            h ^= G * p + g <= kMaxOrder ? hash_term(acc[g], G * p + g) : 0ull;
        h = lanes_xor<P>(h);
        if (team_live && p == 0)
            reinterpret_cast<uint64_t*>(trace)[2 * ((size_t)my_frame * n_sig + sig)] = h;
    }

    // ---- Schur recursion (src/lpc/residue_generator.cpp:47-68), always 100 stages ------------------------------------------
    // Lane p holds columns j = G p + r of gen0 / gen1.  Stage i reads gen1[j + 1] (old): the next register, and for the
    // lane's last column the first register of the next lane.  Columns beyond 99 hold zeros (and what leaks in from the
    // next team's first column); stage i only uses columns below 100 - i, and what is wrong moves down one column per
    // stage from column G P - 1 >= 103: it never gets there.
    {
        double g0[G], g1[G];
        const double next_first = wave_shl1_zero(acc[0]); // ac[G (p + 1)]
#pragma unroll
        for (int r = 0; r < G; r++) {
            const double v = r + 1 < G ? acc[r + 1] : next_first; // ac[j + 1]
            g0[r] = g1[r] = (G * p + r < kMaxOrder) ? v : 0.0;
        }
        double err = 1.0; // ac[0]
        const bool keeper = p == 0;
        double g = team_first<P>(g1[0]);
        double ki = -g / err;
        err += g * ki;
        if (keeper)
            k_b[0] = ki;
#pragma unroll 1
        for (int i = 1; i < kMaxOrder; i++) {
            const double crossing = wave_shl1_zero(g1[0]);
#pragma unroll
            for (int r = 0; r < G; r++) {
                const double sa = r + 1 < G ? g1[r + 1] : crossing; // gen1[j + 1], old
                g1[r] = sa + ki * g0[r];
                g0[r] = sa * ki + g0[r];
            }
            g = team_first<P>(g1[0]);
            ki = -g / err;
            err += g * ki;
            if (keeper)
                k_b[i] = ki;
        }
    }
    wave_sync();
    if (kMode == 2)
        wave_stamp[3] = clock64();

    // ---- order and quantisation of every block (src/lpc/residue_generator.cpp:70-96), lane = coefficient ------------------
    // What a block's tail needs of its analysis: the order and <= 100 quantised coefficients.  They fit int8 (q = floor(64 k),
    // |k| <= 1, and the two square-root forms stay within [-64, 64]) unless the recursion went astray on a degenerate block;
    // such a block keeps its 32-bit values in its own output slot (global memory; the tail reads them back before it
    // writes the slot) and says so in its escape mark.
    int8_t* const q_all = reinterpret_cast<int8_t*>(lds + Plan::kQBase);
#pragma unroll 1
    for (int bb = 0; bb < B; bb++) {
        const uint32_t frame = frame0 + (uint32_t)bb * frame_stride;
        if (frame >= n_frames)
            break; // (the frames of a wave ascend)
        int lane_now = lane0;
        asm volatile("" : "+v"(lane_now));
        const int lane = lane_now;
        const double* const k_mine = reinterpret_cast<const double*>(lds) + bb * Plan::kKStride;
        const double k_lo = k_mine[lane];
        const double k_hi = lane < kMaxOrder - 64 ? k_mine[64 + lane] : 0.0;
        int order;
        {
            const unsigned long long b_lo = __ballot(fabs(k_lo) > SELA_ORDER_THRESHOLD);
            const unsigned long long b_hi = __ballot(lane < 36 && fabs(k_hi) > SELA_ORDER_THRESHOLD);
            order = b_hi ? 128 - __clzll(b_hi) : (b_lo ? 64 - __clzll(b_lo) : 1);
        }
        const double sqrt2 = SELA_SQRT2;
        double v_lo;
        if (lane == 0)
            v_lo = floor(64 * (-1 + (sqrt2 * sqrt(k_lo + 1))));
        else if (lane == 1)
            v_lo = floor(64 * (-1 + (sqrt2 * sqrt(-k_lo + 1))));
        else
            v_lo = floor(64 * k_lo);
        const double v_hi = floor(64 * k_hi);
        const int32_t q_lo = isnan(v_lo) ? 0 : (int32_t)v_lo;
        const int32_t q_hi = isnan(v_hi) ? 0 : (int32_t)v_hi;
        const bool wide = (lane < order && (q_lo < -128 || q_lo > 127)) || (lane + 64 < order && (q_hi < -128 || q_hi > 127));
        const bool escape = __any(wide);
        int8_t* const q_mine = q_all + bb * Plan::kQStride;
        q_mine[lane] = (int8_t)q_lo;
        if (lane < kMaxOrder - 64)
            q_mine[64 + lane] = (int8_t)q_hi;
        if (lane == 0) {
            q_mine[100] = (int8_t)order;
            q_mine[101] = escape ? 1 : 0;
        }
        if (escape) {
            int32_t* const q_wide = reinterpret_cast<int32_t*>(slots + ((size_t)frame * n_sig + sig) * kSlotWords);
            q_wide[lane] = q_lo;
            if (lane < kMaxOrder - 64)
                q_wide[64 + lane] = q_hi;
        }
        if (kTrace) {
            sela_hip_trace* tr = trace + (size_t)frame * n_sig + sig;
            tr->k[lane] = k_lo;
            if (lane < 36)
                tr->k[lane + 64] = k_hi;
        }
        if (kMode == 3) {
            uint64_t h = hash_term(k_lo, lane) ^ (lane < kMaxOrder - 64 ? hash_term(k_hi, lane + 64) : 0ull);
            h = lanes_xor<64>(h);
            if (lane == 0)
                reinterpret_cast<uint64_t*>(trace)[2 * ((size_t)frame * n_sig + sig) + 1] = h;
        }
    }
    wave_sync();

    // ---- the wave's blocks, one after the other: sela_encode_tail.inc ------------------------------------------------------
    SmallArrays* const sm = reinterpret_cast<SmallArrays*>(big + kSmallBase);
    long long stamp[14];
#pragma unroll 1
    for (int bb = 0; bb < B; bb++) {
        const uint32_t frame = frame0 + (uint32_t)bb * frame_stride;
        if (frame >= n_frames)
            break; // (the frames of a wave ascend)
        // (the lane number as this round of the loop must see it: left visibly loop-invariant, the compiler computes every
        // address the tail derives from it -- some three hundred values -- once in front of the loop and keeps or spills
        // them through the analysis above: 256 VGPRs and 22 spilled instead of fitting the budget)
        int lane_now = lane0;
        asm volatile("" : "+v"(lane_now));
        const int lane = lane_now;
        const uint32_t block_id = frame * n_sig + sig;
        if (4 * (bb + 1) > B) // (the last three quarters of the tails: the wave is through most of its work -- lowest priority)
            set_wave_priority((int)((team_priorities >> 24) & 0xFF));
        const int8_t* const q_mine = q_all + bb * Plan::kQStride;
        const int order = (uint8_t)q_mine[100];
        int32_t q_lo = q_mine[lane], q_hi = lane < kMaxOrder - 64 ? q_mine[64 + lane] : 0;
        if (__builtin_amdgcn_readfirstlane((int)q_mine[101])) { // (a degenerate block: its 32-bit values, from its own slot)
            const int32_t* const q_wide = reinterpret_cast<const int32_t*>(slots + (size_t)block_id * kSlotWords);
            q_lo = q_wide[lane];
            q_hi = lane < kMaxOrder - 64 ? q_wide[64 + lane] : 0;
        }
        const double mean = 0.0; // (the trace's mean has been written above; the tail only names it)
        int32_t s[kPerLane];
        {
            const int16_t* fb = pcm + (size_t)frame * kBlock * channels;
            if (channels == 2) {
                const uint32_t* fp2 = reinterpret_cast<const uint32_t*>(fb);
#pragma unroll
                for (int t = 0; t < kPerLane; t++) {
                    const uint32_t w = fp2[lane + 64 * t];
                    const int32_t l = (int16_t)(w & 0xFFFFu), r = (int16_t)(w >> 16);
                    s[t] = sig == 0 ? l : (sig == 1 ? r : l - r); // src/frame/frame_encoder.cpp:22-24
                }
            } else {
#pragma unroll
                for (int t = 0; t < kPerLane; t++)
                    s[t] = fb[(size_t)(lane + 64 * t) * channels + sig];
            }
        }
        uint32_t flags = 0;
        wave_sync(); // (the block before this one is through with the LDS plan)
        {
            constexpr bool kAcInLds = false;
#define SELA_TAIL_HAVE_Q
#include "sela_encode_tail.inc"
#undef SELA_TAIL_HAVE_Q
        }
        SELA_STAMP(12);
        if (kMode == 2 && lane == 0) { // the wave's analysis phases (shared by its blocks) and this block's own tail
            uint64_t* row = phase_cycles + (size_t)block_id * 16;
            const long long now = clock64();
            row[0] = 0;
            row[1] = (uint64_t)(wave_stamp[1] - wave_stamp[0]);
            row[2] = 0;
            row[3] = (uint64_t)(wave_stamp[2] - wave_stamp[1]);
            row[4] = (uint64_t)(wave_stamp[3] - wave_stamp[2]);
            for (int i = 5; i < 12; i++)
                row[i] = (uint64_t)(stamp[i + 1] - stamp[i]);
            row[12] = (uint64_t)(now - wave_stamp[3]); // since the end of the analysis: the tails of the blocks before this one included
            row[13] = (uint64_t)wave_stamp[0];         // absolute: when the wave started ...
            row[14] = (uint64_t)now;                   // ... and when this block was done (tools/phase_profile.py: the launch's ramp)
        }
    }
}

// ---- the device-pointer path's two small kernels behind k_encode_blocks -------------------------------------------
// (batches that are resident in HBM: a kernel boundary costs a few microseconds there, less than the groups' last blocks
// take to do the same work inside the launch -- the one-launch form is the host pipeline's, see finish_group.)
// plan: stereo decision + frame sizes + exclusive scan (one workgroup of 1024)
// choice[f] = 1 when the second channel of an exactly-stereo frame is stored as the difference
// signal (src/frame/frame_encoder.cpp:64-72).
constexpr int kPlanThreadsBig = 1024, kPlanFramesBig = 12288;  // large batches: 57 KB of LDS, 10 us per tile of 12,288 frames
constexpr int kPlanThreadsSmall = 256, kPlanFramesSmall = 4096; // up to 4096 frames: four waves and 18 KB -- a workgroup that finds a
                                                                // place between another stream's resident blocks at once (the big one
                                                                // waited 80-160 us for sixteen waves and 57 KB on ONE CU, timeline_two_lanes.txt)

template <int kPlanThreads, int kPlanLdsFrames>
__global__ __launch_bounds__(kPlanThreads) void k_plan_frames(const BlockMeta* __restrict__ meta, uint32_t n_frames,
    uint32_t channels, uint32_t n_sig, size_t frames_cap, uint64_t* frame_offsets,
    uint8_t* __restrict__ choice_out, uint32_t* __restrict__ status,
    const uint64_t* base_in /* null: the stream starts at 0; else where the frames before these end (may BE frame_offsets: the second half of a split launch) */,
    uint32_t accumulate /* the status words already hold the first half's */)
{
    __shared__ uint64_t part[kPlanThreads / kWave]; // the waves' totals of one tile
    __shared__ uint32_t frame_size[kPlanLdsFrames]; // bytes of the frames of one tile
    __shared__ uint32_t acc[2];                      // flags, frames that do not fit frames_cap
    __shared__ uint64_t start;
    const uint32_t tid = threadIdx.x;
    if (tid < 2)
        acc[tid] = 0;
    if (tid == 0)
        start = base_in ? *base_in : 0;
    __syncthreads(); // (read before any thread writes frame_offsets[0], which may be the same word)
    uint64_t base = start; // bytes of the frames and tiles before this one (the same value in every thread)
    uint32_t flags = 0, overflow = 0;
    // Tiles of kPlanLdsFrames frames.  Within a tile, frame f is sized by thread f mod 1024: the metadata loads of
    // one pass are independent and coalesced, and the passes do not depend on each other (a thread that walks
    // consecutive frames waits for memory once per frame: 0.45 ms for 61 k frames, against 10 us per tile).
    for (uint32_t tile0 = 0; tile0 < n_frames; tile0 += kPlanLdsFrames) {
        const uint32_t tile_n = min((uint32_t)kPlanLdsFrames, n_frames - tile0);
        __syncthreads(); // the previous tile's sizes have been read
#pragma unroll 8
        for (uint32_t i = tid; i < tile_n; i += kPlanThreads) {
            uint32_t choice;
            const uint32_t words = frame_words(meta + (size_t)(tile0 + i) * n_sig, channels, choice, flags);
            choice_out[tile0 + i] = (uint8_t)choice;
            frame_size[i] = (uint32_t)sela_frame_bytes(channels, words);
        }
        __syncthreads();
        const uint32_t per = (tile_n + kPlanThreads - 1) / kPlanThreads;
        const uint32_t begin = min(tid * per, tile_n), end = min(begin + per, tile_n);
        uint64_t bytes = 0;
        for (uint32_t i = begin; i < end; i++)
            bytes += frame_size[i];
        // inclusive scan of the threads' sums: inside every wave by shuffles, the waves' totals by every thread itself
        // (at most 16 of them) -- two barriers, where the Hillis-Steele scan over 256 threads took sixteen
        uint64_t incl = bytes;
        const int lane = (int)(tid % kWave);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d, kWave), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d, kWave);
            if (lane >= d)
                incl += ((uint64_t)hi << 32) | lo;
        }
        if (lane == kWave - 1)
            part[tid / kWave] = incl; // this wave's total
        __syncthreads();
        uint64_t before = 0, tile_total = 0;
#pragma unroll
        for (int w = 0; w < kPlanThreads / kWave; w++) {
            const uint64_t t = part[w];
            before += w < (int)(tid / kWave) ? t : 0;
            tile_total += t;
        }
        __syncthreads(); // (part[] is written again for the next tile)
        uint64_t off = base + before + incl - bytes;
        for (uint32_t i = begin; i < end; i++) {
            frame_offsets[tile0 + i] = off;
            off += frame_size[i];
            if (off > frames_cap)
                overflow++;
        }
        base += tile_total;
    }
    if (tid == 0)
        frame_offsets[n_frames] = base;
    if (flags)
        atomicOr(&acc[0], flags);
    if (overflow)
        atomicAdd(&acc[1], overflow);
    __syncthreads();
    if (tid == 0) { // this single workgroup is the only writer of the encode status words
        if (accumulate) {
            status[0] |= acc[0];
            status[1] += acc[1];
        } else {
            status[0] = acc[0];
            status[1] = acc[1];
            status[2] = status[3] = 0;
        }
    }
}

// ---- assemble: the on-disk bytes of each frame (src/file/sela_file.cpp:115-135) ------------------------
// Frame sizes are multiples of 4 and the stream base is 4-byte aligned, so everything is written as
// aligned u32.  Within a subframe the 7 header bytes push the coefficient words 3 bytes off word
// alignment (funnel shift below); the 5 bytes of the residue header realign the residue words.
constexpr int kAsmThreads = 256;

__global__ __launch_bounds__(kAsmThreads) void k_assemble_frames(const BlockMeta* __restrict__ meta,
    const uint32_t* __restrict__ slots, const uint8_t* __restrict__ choice, const uint64_t* __restrict__ frame_offsets,
    uint32_t n_frames, uint32_t channels, uint32_t n_sig, size_t frames_cap, uint8_t* __restrict__ frames)
{
    const uint32_t f = blockIdx.x;
    if (f >= n_frames)
        return;
    const uint64_t begin = frame_offsets[f], end = frame_offsets[f + 1];
    if (end > frames_cap)
        return; // reported through status[1] by k_plan_frames
    uint32_t* out = reinterpret_cast<uint32_t*>(frames + begin);
    const uint32_t tid = threadIdx.x;
    if (tid == 0)
        out[0] = SELA_SYNC_WORD;
    uint32_t p = 1; // word cursor inside the frame
    for (uint32_t c = 0; c < channels; c++) {
        uint32_t sig = c, type = 0, parent = c;
        if (c == 1 && channels == 2 && choice[f]) {
            sig = 2;
            type = 1;
            parent = 0;
        }
        const BlockMeta b = meta[(size_t)f * n_sig + sig];
        const uint32_t* slot = slots + ((size_t)f * n_sig + sig) * kSlotWords;
        const uint32_t cw = b.coef_words, rw = b.res_words;
        if (tid == 0)
            out[p] = c | (type << 8) | (parent << 16) | ((uint32_t)b.coef_k << 24);
        // words p+1 .. p+1+cw: [cw:16 | order:8] then the coefficient words shifted by 3 bytes, then res_k
        for (uint32_t i = tid; i <= cw; i += kAsmThreads) {
            const uint32_t low = i == 0 ? (cw | ((uint32_t)b.order << 16)) : (slot[i - 1] >> 8);
            const uint32_t top = i < cw ? slot[i] : (uint32_t)b.res_k;
            out[p + 1 + i] = (low & 0x00FFFFFFu) | (top << 24);
        }
        if (tid == 0)
            out[p + 2 + cw] = rw | ((uint32_t)kBlock << 16);
        const uint32_t* rs = slot + kCoefWordsCap;
        uint32_t* ro = out + p + 3 + cw;
        for (uint32_t i = tid; i < rw; i += kAsmThreads)
            ro[i] = rs[i];
        p += 3 + cw + rw;
    }
}

// ---- rice::RiceEncoder on its own (src/rice/rice_encoder.cpp:12-81): any number of int32 values per stream ------------------
// The reference's public L1 class (src/include/rice.hpp:9-27); here one wave per stream, for callers and tests of the stage
// itself (sela_hip_rice_encode) -- the frame path packs its residues inside k_encode_blocks / k_encode_teams.  All 20
// candidates are summed (no walk: a stream may be a single value), the first minimum taken, requiredInts =
// ceil((float)bits / 32) as the reference computes it, and the codewords are OR-ed into the caller's zeroed words.
__global__ __launch_bounds__(64) void k_stage_rice_encode(const int32_t* __restrict__ values, const uint64_t* __restrict__ value_offsets,
    uint32_t n_streams, uint32_t* __restrict__ k_out, uint32_t* __restrict__ word_counts, uint32_t* __restrict__ words_out,
    const uint64_t* __restrict__ word_offsets, uint32_t* __restrict__ status)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_streams)
        return;
    const int lane = threadIdx.x;
    const int32_t* v = values + value_offsets[s];
    const uint64_t n = value_offsets[s + 1] - value_offsets[s];
    uint32_t* out = words_out + word_offsets[s];
    const uint64_t cap = word_offsets[s + 1] - word_offsets[s];
    uint64_t sum[SELA_MAX_RICE_PARAM];
#pragma unroll
    for (int k = 0; k < SELA_MAX_RICE_PARAM; k++)
        sum[k] = 0;
    bool wide = false;
    for (uint64_t i = lane; i < n; i += 64) {
        const int32_t x = v[i];
        wide |= (x >= (1 << 30)) || (x < -(1 << 30)); // (the reference's int32 zig-zag overflows: undefined there)
        const uint32_t u = zigzag32(x);
#pragma unroll
        for (int k = 0; k < SELA_MAX_RICE_PARAM; k++)
            sum[k] += u >> k;
    }
    uint32_t best_k = 0;
    uint64_t best_bits = ~0ull;
#pragma unroll
    for (int k = 0; k < SELA_MAX_RICE_PARAM; k++) {
        const uint64_t bits = wave_sum_u64(sum[k]) + n * (uint64_t)(1 + k);
        if (bits < best_bits) // strict: the FIRST minimum (src/rice/rice_encoder.cpp:26-31)
            best_bits = bits, best_k = (uint32_t)k;
    }
    uint32_t flags = __any(wide) ? (uint32_t)SELA_HIP_FLAG_RICE_RANGE : 0u;
    // requiredInts = ceil((float)bits / 32) is what the reference reports AND writes (rice_encoder.cpp:37,63-70): above 2^24
    // bits the float drops low bits of the count, and a count that rounds DOWN across a multiple of 32 leaves the stream's
    // last bits unwritten there.  Same here: `words` is the reference's number, and no codeword bit lands beyond it.
    const uint32_t words = best_bits < (1ull << 31) ? words_for_bits(best_bits) : 0xFFFFFFFFu;
    if (words > cap)
        flags |= SELA_HIP_FLAG_WORDS_CAP;
    if (lane == 0) {
        k_out[s] = best_k;
        word_counts[s] = words;
        if (flags)
            atomicOr(&status[0], flags);
    }
    if (flags)
        return;
    const bool clipped = ((best_bits + 31) >> 5) > (uint64_t)words; // (only ever beyond 2^24 bits)
    uint32_t base = 0;
    for (uint64_t i0 = 0; i0 < n; i0 += 64) {
        const bool valid = i0 + lane < n;
        const uint32_t u = valid ? zigzag32(v[i0 + lane]) : 0u;
        const uint32_t len = valid ? (u >> best_k) + 1 + best_k : 0u;
        const uint32_t before = wave_exclusive_scan(len, lane);
        if (valid) {
            const uint32_t at = base + before;
            if (!clipped || (uint64_t)at + len <= 32ull * words) {
                (void)put_codeword(out, at, u, best_k);
            } else { // the codeword that straddles (or lies beyond) the reference's last word: bit by bit, the rest dropped
                const uint32_t ones = u >> best_k;
                for (uint32_t b = 0; b < len && (uint64_t)at + b < 32ull * words; b++) {
                    const uint32_t bit = b < ones ? 1u : (b == ones ? 0u : (u >> (best_k - 1 - (b - ones - 1))) & 1u);
                    if (bit)
                        atomicOr(&out[(at + b) >> 5], 1u << ((at + b) & 31));
                }
            }
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)(before + len), 63);
    }
}

hipError_t launch_stage_rice_encode(const int32_t* d_values, const uint64_t* d_value_offsets, uint32_t n_streams, uint32_t* d_k, uint32_t* d_word_counts,
    uint32_t* d_words, const uint64_t* d_word_offsets, uint32_t* d_status, hipStream_t stream)
{
    if (n_streams)
        hipLaunchKernelGGL(k_stage_rice_encode, dim3(n_streams), dim3(64), 0, stream, d_values, d_value_offsets, n_streams, d_k, d_word_counts, d_words,
            d_word_offsets, d_status);
    return hipGetLastError();
}

} // namespace sela

// ---- host-side launchers (called from sela_capi.hip) ---------------------------------------------------
namespace sela {

size_t encode_workspace_bytes(uint32_t n_frames, uint32_t channels)
{
    const uint32_t n_sig = sela_hip_signals_per_frame(channels);
    const size_t blocks = (size_t)n_frames * n_sig;
    size_t bytes = 0;
    bytes += (blocks * sizeof(BlockMeta) + 255) & ~(size_t)255;
    bytes += (blocks * kSlotWords * 4 + 255) & ~(size_t)255;
    const size_t n_groups = ((size_t)n_frames + kGroupFrames - 1) / kGroupFrames;
    bytes += ((n_groups * kGroupCountStride * sizeof(uint64_t) + 255) & ~(size_t)255) + ((n_groups * sizeof(GroupState) + 255) & ~(size_t)255); // finish_group
    bytes += (size_t)kXcds * kRingsPerXcd * kRingLen * sizeof(double) + 256; // scalar-operand rings (L2-resident) ...
    bytes += (size_t)kXcds * kRingsPerXcd * 4 + 256;                          // ... and their owner words
    const size_t padded = (((size_t)n_frames + 63) / 64) * 64 * n_sig;        // encode indices (frames rounded up to a span of 64)
    bytes += 2 * ((padded * sizeof(double) + 255) & ~(size_t)255); // worker means + ready words (64-bit launch tags)
    bytes += (blocks * sizeof(uint64_t) + 255) & ~(size_t)255;     // the stereo candidates' sizes (sela_encode_tail.inc)
    return bytes + 256;
}

// Workgroups of k_encode_blocks the current device holds at once: 12 per CU (LDS), cached per device.
static uint32_t resident_encode_blocks()
{
    static std::atomic<uint32_t> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256 * 12;
    uint32_t v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        v = (uint32_t)cus * 12;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}


// Lanes per team (k_encode_teams) for a launch of `blocks` blocks; 0: k_encode_blocks.
//
// A wave of k_encode_teams is 4 / 8 blocks' worth of work, a SIMD holds three, and a launch ends with its last wave: the
// kernel time of the team kernels is a staircase over the batch size -- one step per wave that the fullest SIMD holds in the
// launch's last round (tools/teams_sweep.py on a fine grid, profiles/r04/teams_fine_sweep.txt; stereo, kernel alone, ms):
//   frames   k_encode_blocks   teams of 16   teams of 8        frames   k_encode_blocks   teams of 16   teams of 8
//     3300        0.389           0.421         0.502           8192        0.805           0.708         0.691
//     3875        0.434           0.405         0.514           8300        0.801           0.777         0.854
//     4096        0.461           0.409         0.514          12288        1.252           1.049         1.055
//     4200        0.464           0.492         0.646          12500        1.281           1.120         1.059
//     4700        0.520           0.511         0.669          16384        1.628           1.366         1.255
//     6000        0.621           0.602         0.678          16600        1.668           1.440         1.422
// (one fill of teams of 16 = 3072 waves = 4096 stereo frames, of teams of 8 = 8192), while k_encode_blocks -- a block per
// wave -- grows evenly.  So the choice is made by a model of the three kernels' times, in waves per SIMD of the last round,
// with the constants of that sweep (they scale with the device's clock and CU count together, and only their ratios
// matter): it reproduces the sweep's times to within 5 % and its choice everywhere but at near-ties.  With two batches in
// flight (bench.py) teams of 16 give 16.3 G samples/s at 3875 frames against 14.4 for k_encode_blocks.
struct TeamCost {
    double first[3], next[3], per_fill, first_fill; // ms: a last round with <= 1, 2, 3 waves on the fullest SIMD, alone / behind full rounds
};
static double team_kernel_ms(size_t waves, size_t slots, const TeamCost& c)
{
    const size_t full = waves / slots, rem = waves % slots, per_simd = slots / 3;
    const int step = rem == 0 ? -1 : (rem <= per_simd ? 0 : (rem <= 2 * per_simd ? 1 : 2));
    if (full == 0)
        return step < 0 ? 0.0 : c.first[step];
    return c.first_fill + c.per_fill * (double)(full - 1) + (step < 0 ? 0.0 : c.next[step]);
}
static int team_lanes_for(size_t blocks)
{
    const size_t slots = resident_encode_blocks(); // 12 waves per CU = 3 per SIMD
    const double scale = 3072.0 / (double)slots;   // (the constants are those of 256 CUs)
    const double kb = (double)blocks / 1000.0 * scale;
    const double t_blocks = kb <= 27.0 ? 0.10 + 0.0292 * kb : 0.0332 * kb;
    static const TeamCost c16 = {{0.215, 0.305, 0.405}, {0.10, 0.19, 0.30}, 0.32, 0.41};
    static const TeamCost c8 = {{0.36, 0.58, 0.69}, {0.175, 0.365, 0.56}, 0.564, 0.69};
    const double t16 = team_kernel_ms((blocks + 3) / 4, slots, c16);
    const double t8 = team_kernel_ms((blocks + 7) / 8, slots, c8);
    // (near ties go to the kernel with fewer instructions per block -- 16.2 k / 14.5 k / 13.3 k -- which is what counts once
    // another stream's kernels fill the launch's idle slots)
    const double s16 = 0.97 * t16, s8 = 0.94 * t8;
    if (t_blocks < s16 && t_blocks < s8)
        return 0;
    return s8 <= s16 ? 8 : 16;
}

// Where a launch that has the device to itself is cut in two (sela_capi.hip: the halves run on two streams, the first half's plan +
// assemble under the second half's tail): only launches the library gives to teams of 16 as a whole -- both halves then run
// teams of 16 as well -- and at a multiple of 32 frames (eight waves of four frames: a half's grid stays whole).  0: not split.
uint32_t encode_split_frames(uint32_t n_frames, uint32_t channels, int permille /* of the frames to the first half; < 0: the default */)
{
    const size_t blocks = (size_t)n_frames * sela_hip_signals_per_frame(channels);
    if (n_frames < 2048 || team_lanes_for(blocks) != 16)
        return 0;
    const uint32_t share = permille < 0 ? 520u : (uint32_t)permille;
    const uint32_t first = (uint32_t)((uint64_t)n_frames * share / 1000 + 16) / 32 * 32;
    return first >= 32 && first + 32 <= n_frames ? first : 0;
}

// debug (sela_hip_debug_keep_both_candidates): write both stereo candidates' slots as round 3 did -- for the comparison of
// the traffic and for the tests, which check that the bytes do not depend on it
static std::atomic<int> g_keep_both_candidates{0};
void set_keep_both_candidates(int on) { g_keep_both_candidates.store(on, std::memory_order_relaxed); }
// debug (sela_hip_debug_encode_hashes): a launch with a d_trace pointer takes the kMode 3 instantiations and leaves two 64-bit
// words per block there (hash_term) instead of a trace
static std::atomic<int> g_encode_hashes{0};
void set_encode_hashes(int on) { g_encode_hashes.store(on, std::memory_order_relaxed); }


int encode_team_lanes(uint32_t n_frames, uint32_t channels, int forced)
{
    return forced >= 0 ? forced : team_lanes_for((size_t)n_frames * sela_hip_signals_per_frame(channels));
}

hipError_t launch_encode(const int16_t* d_pcm, uint32_t n_frames, uint32_t channels, uint8_t* d_frames,
    size_t frames_cap, uint64_t* d_frame_offsets, uint32_t* d_status, void* d_workspace, sela_hip_trace* d_trace,
    hipStream_t stream, hipEvent_t* ev /* 4 events or nullptr */, uint64_t* d_phase_cycles,
    const EncodeHostLink* link /* the host pipeline's one-launch form; nullptr: three kernels */,
    int force_plain_fir, int self_blocks_override, int team_lanes /* -1: by launch size; 0: k_encode_blocks; 8, 16: k_encode_teams<P> */,
    int32_t* d_trace_residues /* with d_trace and team_lanes 0: every block's residues, [block][2048]; or nullptr */,
    uint32_t priorities /* wave priorities by quarters of a wave's work, e.g. 0x00010203 falling; 0: none (the caller knows whether the launch has the device to itself) */,
    int phase /* 0: everything; 1: the blocks only; 2: plan + assemble only (a launch split over two streams, sela_capi.hip) */,
    const uint64_t* plan_base /* phase 2: where the frames before these end (device), or null */, bool plan_accumulate)
{
    const uint32_t n_sig = sela_hip_signals_per_frame(channels);
    const size_t blocks = (size_t)n_frames * n_sig;
    const size_t n_groups = ((size_t)n_frames + kGroupFrames - 1) / kGroupFrames;
    unsigned char* ws = static_cast<unsigned char*>(d_workspace);
    ws = reinterpret_cast<unsigned char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    BlockMeta* meta = reinterpret_cast<BlockMeta*>(ws);
    ws += (blocks * sizeof(BlockMeta) + 255) & ~(size_t)255;
    uint32_t* slots = reinterpret_cast<uint32_t*>(ws);
    ws += (blocks * kSlotWords * 4 + 255) & ~(size_t)255;
    uint64_t* group_count = reinterpret_cast<uint64_t*>(ws);
    ws += (n_groups * kGroupCountStride * sizeof(uint64_t) + 255) & ~(size_t)255;
    GroupState* group_state = reinterpret_cast<GroupState*>(ws);
    ws += (n_groups * sizeof(GroupState) + 255) & ~(size_t)255;
    double* rings = reinterpret_cast<double*>(ws);
    ws += ((size_t)kXcds * kRingsPerXcd * kRingLen * sizeof(double) + 255) & ~(size_t)255;
    uint32_t* ring_owner = reinterpret_cast<uint32_t*>(ws);
    ws += ((size_t)kXcds * kRingsPerXcd * 4 + 255) & ~(size_t)255;
    const size_t padded = (((size_t)n_frames + 63) / 64) * 64 * n_sig;
    double* mean_out = reinterpret_cast<double*>(ws);
    ws += (padded * sizeof(double) + 255) & ~(size_t)255;
    uint64_t* mean_ready = reinterpret_cast<uint64_t*>(ws);
    ws += (padded * sizeof(uint64_t) + 255) & ~(size_t)255;
    uint64_t* sizes_pub = reinterpret_cast<uint64_t*>(ws);

    if (link && (d_trace || d_phase_cycles))
        return hipErrorInvalidValue; // (the analysis trace and the phase counts are the device-pointer path's)
    if (phase != 0 && (link || n_frames == 0))
        return hipErrorInvalidValue;
    if (phase != 2) {
    if (n_frames == 0) { // (nothing to launch; a job's stream position stays where it is)
        hipError_t err = hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream);
        if (err == hipSuccess && d_frame_offsets)
            err = hipMemsetAsync(d_frame_offsets, 0, sizeof(uint64_t), stream);
        return err;
    }
    const uint32_t total_e = (n_frames + 63) / 64 * 64 * n_sig; // (whole spans of 64 frames: block_of)
    // mean workers (see mean_worker): blocks that cannot start before the first ones retire get their mean from
    // a worker.  self_blocks = what the device holds at once (12 workgroups per CU) less the workers themselves.
    const bool staged = link && link->host_pcm && channels == 2;
    uint32_t self_blocks = self_blocks_override >= 0 ? (uint32_t)self_blocks_override : resident_encode_blocks();
    uint32_t n_workers = 0;
    // a worker wave sums the blocks of 64 / n_sig frames (mean_worker); the workers cover the frames from the span of 64
    // that holds encode index self_blocks to the end
    const uint32_t frames_padded = (n_frames + 63) / 64 * 64, per_wave = n_sig < 64u ? 64u / n_sig : 1u;
    auto workers_for = [&](uint32_t self) { return (frames_padded - std::min(frames_padded, self / (64u * n_sig) * 64u) + per_wave - 1) / per_wave; };
    if (total_e > self_blocks && !staged) { // (with stagers the link sets the pace, and a worker has no await_frame)
        n_workers = workers_for(self_blocks);
        if (self_blocks_override < 0) { // the workers take slots of the first fill too
            const uint32_t resident = self_blocks;
            self_blocks = resident > n_workers + 64 ? resident - n_workers : 64;
            n_workers = workers_for(self_blocks);
        }
        n_workers = (n_workers + 7) & ~7u; // keeps encode index == workgroup index mod 8 (XCD placement)
    }
    if (n_workers == 0)
        self_blocks = total_e; // (nobody to wait for)
    const dim3 grid(n_workers + total_e), wg(64);
    // launch ticket: unique per launch in this process, never 0.  Everything a launch leaves in the workspace for its
    // own workgroups (ring owners, worker means, group counters, look-back cells) carries it, so nothing is cleared
    // between launches; the cells whose stale contents could be mistaken for this launch's carry a 64-bit tag, the
    // ticket under a nonce drawn once per process (another process's launches count from the same start).
    static std::atomic<uint32_t> next_ticket{ 0x5E1A0001u };
    static const uint32_t nonce = [] {
        std::random_device rd;
        return (uint32_t)rd() ^ ((uint32_t)rd() << 16);
    }();
    uint32_t ticket = next_ticket.fetch_add(1, std::memory_order_relaxed);
    if (ticket == 0)
        ticket = next_ticket.fetch_add(1, std::memory_order_relaxed);
    FuseArgs fa;
    fa.meta = meta;
    fa.slots = slots;
    fa.group_count = link ? group_count : nullptr; // (null: the blocks leave placing and writing to the two kernels behind them)
    fa.group_state = group_state;
    fa.frames = d_frames;
    fa.frames_cap = frames_cap;
    fa.frame_offsets = d_frame_offsets;
    fa.mirror = link ? link->mirror : nullptr;
    fa.status = d_status;
    fa.pos_in = link ? link->pos_in : nullptr;
    fa.pos_out = link ? link->pos_out : nullptr;
    fa.pcm_ready = staged ? link->pcm_ready : nullptr;
    fa.groups_done = staged ? link->stage_started + 3 : nullptr;
    if (staged) { // the stagers first, on their own stream
        // (above the default dynamic-LDS limit; the attribute is per device, so every time)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage_in), hipFuncAttributeMaxDynamicSharedMemorySize, kStageLdsBytes);
        (void)hipMemsetAsync(link->stage_started + 1, 0, 8, link->stage_stream); // the stagers' frame counter
        hipLaunchKernelGGL(k_stage_in, dim3(link->stage_workgroups), dim3(kStageThreads), kStageLdsBytes, link->stage_stream, link->host_pcm,
            const_cast<int16_t*>(d_pcm), n_frames, link->pcm_ready, ticket, link->stage_started, ((uint64_t)nonce << 32) | ticket, (uint32_t)n_groups);
        hipLaunchKernelGGL(k_stage_gate, dim3(1), dim3(64), 0, stream, link->stage_started, ticket, link->stage_workgroups, ((uint64_t)nonce << 32) | ticket);
    }
    fa.trace_residues = d_trace_residues;
    fa.nap_limit = link && link->wait_naps >= 0 ? (uint32_t)link->wait_naps : kFuseNapLimit;
    fa.n_frames = n_frames;
    fa.channels = channels;
    fa.n_sig = n_sig;
    fa.ticket = ticket;
    fa.tag = ((uint64_t)nonce << 32) | ticket;
    // (the stereo candidates' sizes: 20 bits of the nonce and the ticket above the 12 bits of a size, sela_encode_tail.inc)
    const uint64_t sizes_tag = ((uint64_t)(nonce & 0xFFFFFu) << 44) | ((uint64_t)ticket << 12);
    if (g_keep_both_candidates.load(std::memory_order_relaxed))
        sizes_pub = nullptr;
    fa.sizes_pub = sizes_pub;
    fa.sizes_tag = sizes_tag;
    fa.priorities = priorities;
    if (ev)
        (void)hipEventRecord(ev[0], stream);
    // Which kernel analyses the blocks: k_encode_teams (several blocks side by side in a wave: fewer instructions per block,
    // but a wave is B blocks' worth of work) for launches that fill the device several times over, k_encode_blocks otherwise
    // and for the host pipeline's one-launch form.
    if (d_trace_residues && (!d_trace || team_lanes != 0))
        return hipErrorInvalidValue;
    const bool hashes = d_trace && !d_trace_residues && !d_phase_cycles && !link && g_encode_hashes.load(std::memory_order_relaxed);
    int teams = 0;
    if (!link) {
        teams = team_lanes >= 0 ? team_lanes : (d_phase_cycles ? 0 : team_lanes_for(blocks));
        if (teams != 0 && teams != 8 && teams != 16)
            return hipErrorInvalidValue;
    }
    if (teams) {
        const uint32_t per_wave = 64u / (uint32_t)teams;
        const uint32_t waves = ((n_frames + per_wave - 1) / per_wave + 7) / 8 * 8 * n_sig;
#define SELA_LAUNCH_TEAMS(MODE, LANES) \
    hipLaunchKernelGGL((k_encode_teams<MODE, LANES>), dim3(waves), wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, d_trace, force_plain_fir, d_phase_cycles, sizes_pub, sizes_tag, priorities)
        if (teams == 8) {
            if (d_phase_cycles)
                SELA_LAUNCH_TEAMS(2, 8);
            else if (hashes)
                SELA_LAUNCH_TEAMS(3, 8);
            else if (d_trace)
                SELA_LAUNCH_TEAMS(1, 8);
            else
                SELA_LAUNCH_TEAMS(0, 8);
        } else {
            if (d_phase_cycles)
                SELA_LAUNCH_TEAMS(2, 16);
            else if (hashes)
                SELA_LAUNCH_TEAMS(3, 16);
            else if (d_trace)
                SELA_LAUNCH_TEAMS(1, 16);
            else
                SELA_LAUNCH_TEAMS(0, 16);
        }
#undef SELA_LAUNCH_TEAMS
    } else
    if (d_phase_cycles)
        hipLaunchKernelGGL((k_encode_blocks<2, false>), grid, wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, rings, ring_owner, ticket, d_trace, d_phase_cycles, force_plain_fir, mean_out, mean_ready, n_workers, self_blocks, total_e, fa);
    else if (hashes)
        hipLaunchKernelGGL((k_encode_blocks<3, false>), grid, wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, rings, ring_owner, ticket, d_trace, d_phase_cycles, force_plain_fir, mean_out, mean_ready, n_workers, self_blocks, total_e, fa);
    else if (d_trace)
        hipLaunchKernelGGL((k_encode_blocks<1, false>), grid, wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, rings, ring_owner, ticket, d_trace, d_phase_cycles, force_plain_fir, mean_out, mean_ready, n_workers, self_blocks, total_e, fa);
    else if (link)
        hipLaunchKernelGGL((k_encode_blocks<0, true>), grid, wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, rings, ring_owner, ticket, d_trace, d_phase_cycles, force_plain_fir, mean_out, mean_ready, n_workers, self_blocks, total_e, fa);
    else
        hipLaunchKernelGGL((k_encode_blocks<0, false>), grid, wg, 0, stream, d_pcm, n_frames, channels, n_sig, meta, slots, rings, ring_owner, ticket, d_trace, d_phase_cycles, force_plain_fir, mean_out, mean_ready, n_workers, self_blocks, total_e, fa);
    if (ev)
        (void)hipEventRecord(ev[1], stream);
    } // (phase != 2)
    if (!link && phase != 1) {
        uint8_t* const choice = reinterpret_cast<uint8_t*>(group_state); // (the look-back cells' space: five bytes per frame, unused on this path)
        // (the small plan exists for launches with a neighbour, whose resident workgroups leave no CU sixteen free wave slots
        // and 57 KB; a launch that is alone -- the caller says so with its priorities -- finds them at once, and sixteen waves
        // size and scan 2048+ frames in half the time of four: 10.8 against 19.2 us at 3875 frames, one lane 12.39 -> 12.58 G
        // samples/s, two lanes unchanged, A/B on one box)
        if (n_frames <= (uint32_t)kPlanFramesSmall && !(priorities != 0 && n_frames >= 2048u))
            hipLaunchKernelGGL((k_plan_frames<kPlanThreadsSmall, kPlanFramesSmall>), dim3(1), dim3(kPlanThreadsSmall), 0, stream, meta, n_frames, channels, n_sig,
                frames_cap, d_frame_offsets, choice, d_status, plan_base, plan_accumulate ? 1u : 0u);
        else
            hipLaunchKernelGGL((k_plan_frames<kPlanThreadsBig, kPlanFramesBig>), dim3(1), dim3(kPlanThreadsBig), 0, stream, meta, n_frames, channels, n_sig,
                frames_cap, d_frame_offsets, choice, d_status, plan_base, plan_accumulate ? 1u : 0u);
        if (ev)
            (void)hipEventRecord(ev[2], stream);
        hipLaunchKernelGGL(k_assemble_frames, dim3(n_frames), dim3(kAsmThreads), 0, stream, meta, slots, choice, d_frame_offsets, n_frames, channels,
            n_sig, frames_cap, d_frames);
        if (ev)
            (void)hipEventRecord(ev[3], stream);
    }
    return hipGetLastError();
}

} // namespace sela
