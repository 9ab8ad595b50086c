/*
 * sela_oracle.c -- CPU restatement of the SELA frame encode/decode path (see sela_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: parity oracle + "port"-kind CPU baseline.  Never linked into,
 * imported by, or used as a fallback for the product path.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * checkout).  The arithmetic contract: IEEE binary64, round-to-nearest,
 * no FMA contraction (built with -ffp-contract=off), each FP64 accumulator updated in the
 * reference's loop order; integer side is two's-complement wrap-around with arithmetic
 * right shifts.  Where the reference has undefined behaviour we do what x86-64/g++ -O2 does
 * and raise a flag bit instead of crashing.
 */
#define _POSIX_C_SOURCE 200809L
#include "sela_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sela_format.h"

static void raise_flag(uint32_t* flags, uint32_t bit)
{
    if (flags)
        *flags |= bit;
}

/* (int64_t)double with the x86 cvttsd2si result for NaN / out-of-range inputs. */
static int64_t trunc_to_i64(double v, uint32_t* flags)
{
    if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0)) {
        raise_flag(flags, SELA_ORACLE_FLAG_COEF_OVERFLOW);
        return INT64_MIN;
    }
    return (int64_t)v;
}

/* (int32_t)double for the quantiser; NaN is handled by the caller (-> 0). */
static int32_t trunc_to_i32(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT32_MIN;
    return (int32_t)v;
}

static int clamp_q(int32_t q, uint32_t* flags)
{
    int idx = (int)q + 64;
    if (idx < 0 || idx > 127) {
        raise_flag(flags, SELA_ORACLE_FLAG_Q_RANGE);
        idx = idx < 0 ? 0 : 127;
    }
    return idx;
}

/* ---- LPC: dequantise + step-up -------------------------------------------------------
 * src/lpc/linear_predictor.cpp:16-28 (dequantizeReflectionCoefficients) and :30-61
 * (generatelinearPredictionCoefficients).  a[0] = 0, a[m+1] = (int64)(2^35 * -t[m]). */
void sela_oracle_lpc_coeffs(int order, const int32_t* q, int64_t* a, uint32_t* flags)
{
    double k[SELA_MAX_LPC_ORDER];
    double t[SELA_MAX_LPC_ORDER];
    int nk;
    if (order <= 1) { /* :19-22 */
        k[0] = 0.0;
        nk = 1;
    } else { /* :23-27 */
        k[0] = SELA_DEQUANT_FIRST[clamp_q(q[0], flags)];
        k[1] = SELA_DEQUANT_SECOND[clamp_q(q[1], flags)];
        for (int i = 2; i < order; i++)
            k[i] = SELA_DEQUANT_HIGHER[clamp_q(q[i], flags)];
        nk = order;
    }
    (void)nk;
    for (int i = 0; i < order; i++) { /* :39-54 */
        t[i] = k[i];
        const int i2 = i >> 1;
        int j;
        for (j = 0; j < i2; j++) {
            const double tmp = t[j];
            t[j] += k[i] * t[i - 1 - j];
            t[i - 1 - j] += k[i] * tmp;
        }
        if (i % 2 == 1)
            t[j] += t[j] * k[i];
    }
    a[0] = 0; /* :57 */
    for (int i = 0; i < order; i++) /* :58-60, correction = 2^35 as a double */
        a[i + 1] = trunc_to_i64(34359738368.0 * (-t[i]), flags);
}

/* ---- LPC analysis ---------------------------------------------------------------------
 * src/lpc/residue_generator.cpp:12-134. */
int sela_oracle_lpc_analyze(const int32_t* s, int n, int32_t* q, int64_t* a, int32_t* r,
    sela_oracle_lpc_trace* trace, uint32_t* flags)
{
    double* x = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double ac[SELA_MAX_LPC_ORDER + 1];
    double k[SELA_MAX_LPC_ORDER];
    double gen0[SELA_MAX_LPC_ORDER], gen1[SELA_MAX_LPC_ORDER];

    /* quantizeSamples :12-18 */
    for (int j = 0; j < n; j++)
        x[j] = (double)s[j] / SELA_SAMPLE_SCALE;

    /* generateAutoCorrelation :20-45 -- sequential sum, then sequential per-lag sums */
    double sum = 0.0;
    for (int j = 0; j < n; j++)
        sum += x[j];
    const double mean = sum / (double)n;
    for (int i = 0; i <= SELA_MAX_LPC_ORDER; i++) {
        double acc = 0.0;
        for (int j = i; j < n; j++)
            acc += (x[j] - mean) * (x[j - i] - mean);
        ac[i] = acc;
    }
    for (int i = 1; i <= SELA_MAX_LPC_ORDER; i++)
        ac[i] /= ac[0];
    ac[0] = 1.0;

    /* generateReflectionCoefficients :47-68 (Schur recursion, always 100 stages) */
    for (int i = 0; i < SELA_MAX_LPC_ORDER; i++)
        gen0[i] = gen1[i] = ac[i + 1];
    double err = ac[0];
    k[0] = -gen1[0] / err;
    err += gen1[0] * k[0];
    for (int i = 1; i < SELA_MAX_LPC_ORDER; i++) {
        for (int j = 0; j < SELA_MAX_LPC_ORDER - i; j++) {
            gen1[j] = gen1[j + 1] + k[i - 1] * gen0[j];
            gen0[j] = gen1[j + 1] * k[i - 1] + gen0[j];
        }
        k[i] = -gen1[0] / err;
        err += gen1[0] * k[i];
    }

    /* generateoptimalLpcOrder :70-78 (default 1, src/include/lpc.hpp:76) */
    int order = 1;
    for (int i = SELA_MAX_LPC_ORDER - 1; i >= 0; i--) {
        if (fabs(k[i]) > SELA_ORDER_THRESHOLD) {
            order = i + 1;
            break;
        }
    }

    /* quantizeReflectionCoefficients :80-96 */
    {
        const double sqrt2 = SELA_SQRT2;
        double val = floor(64 * (-1 + (sqrt2 * sqrt(k[0] + 1))));
        q[0] = isnan(val) ? 0 : trunc_to_i32(val);
        if (order > 1) {
            val = floor(64 * (-1 + (sqrt2 * sqrt(-k[1] + 1))));
            q[1] = isnan(val) ? 0 : trunc_to_i32(val);
        }
        for (int i = 2; i < order; i++) {
            val = floor(64 * k[i]);
            q[i] = isnan(val) ? 0 : trunc_to_i32(val);
        }
    }

    if (trace) {
        trace->mean = mean;
        memcpy(trace->ac, ac, sizeof ac);
        memcpy(trace->k, k, sizeof k);
    }

    /* dequantise + step-up :129-130 */
    sela_oracle_lpc_coeffs(order, q, a, flags);

    /* generateResidues :98-119 (wrap-around int64 MACs, arithmetic >> 35, int32 truncation) */
    const uint64_t corr = (uint64_t)1 << (SELA_Q_SHIFT - 1);
    if (n > 0)
        r[0] = s[0];
    for (int i = 1; i <= order && i < n; i++) {
        uint64_t temp = corr;
        for (int j = 1; j <= i; j++)
            temp += (uint64_t)a[j] * (uint64_t)(int64_t)s[i - j];
        r[i] = (int32_t)((uint32_t)s[i] - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
    }
    for (int i = order + 1; i < n; i++) {
        uint64_t temp = corr;
        for (int j = 0; j <= order; j++)
            temp += (uint64_t)a[j] * (uint64_t)(int64_t)s[i - j];
        r[i] = (int32_t)((uint32_t)s[i] - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
    }
    free(x);
    return order;
}

/* ---- LPC synthesis ----------------------------------------------------------------------
 * src/lpc/sample_generator.cpp:11-39.  Note the decoder rounds half-DOWN (2^34 - sum) where
 * the encoder rounds half-up (2^34 + sum): replicated as is. */
void sela_oracle_lpc_synth(int order, const int32_t* q, const int32_t* r, int n, int32_t* s, uint32_t* flags)
{
    int64_t a[SELA_MAX_LPC_ORDER + 1];
    sela_oracle_lpc_coeffs(order, q, a, flags);
    const uint64_t corr = (uint64_t)1 << (SELA_Q_SHIFT - 1);
    memset(s, 0, sizeof(int32_t) * (size_t)n);
    if (n > 0)
        s[0] = r[0];
    for (int i = 1; i <= order && i < n; i++) {
        uint64_t temp = corr;
        for (int j = 1; j <= i; j++)
            temp -= (uint64_t)a[j] * (uint64_t)(int64_t)s[i - j];
        s[i] = (int32_t)((uint32_t)r[i] - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
    }
    for (int i = order + 1; i < n; i++) {
        uint64_t temp = corr;
        for (int j = 0; j <= order; j++) /* s[i] is still 0 and a[0] is 0 */
            temp -= (uint64_t)a[j] * (uint64_t)(int64_t)s[i - j];
        s[i] = (int32_t)((uint32_t)r[i] - (uint32_t)(int32_t)((int64_t)temp >> SELA_Q_SHIFT));
    }
}

/* ---- Golomb-Rice encode -------------------------------------------------------------------
 * src/rice/rice_encoder.cpp:12-81.  Bit t of the stream is bit (t % 32) of word t / 32. */
static uint64_t zigzag(int32_t x)
{
    /* :15  x < 0 ? (-(x << 1)) - 1 : (x << 1), evaluated in int32 then widened to uint64 */
    const int32_t t = (int32_t)((uint32_t)x << 1);
    const int32_t z = x < 0 ? (int32_t)(0u - (uint32_t)t - 1u) : t;
    return (uint64_t)(int64_t)z;
}

static void put_bits(uint32_t* words, uint64_t* pos, uint64_t value, unsigned nbits)
{
    /* append nbits (<= 32) of value, least significant bit first */
    while (nbits) {
        const uint64_t w = *pos >> 5;
        const unsigned off = (unsigned)(*pos & 31);
        const unsigned take = (32 - off) < nbits ? (32 - off) : nbits;
        const uint32_t mask = take == 32 ? 0xFFFFFFFFu : ((1u << take) - 1u);
        words[w] |= ((uint32_t)value & mask) << off;
        value >>= take;
        *pos += take;
        nbits -= take;
    }
}

static uint32_t bitrev(uint32_t v, unsigned nbits)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < nbits; i++)
        r |= ((v >> i) & 1u) << (nbits - 1 - i);
    return r;
}

int sela_oracle_rice_encode(const int32_t* in, int n, uint32_t* k_out, uint32_t* words, int cap, uint32_t* flags)
{
    uint64_t best_bits = 0;
    uint32_t best_k = 0;
    for (uint32_t k = 0; k < SELA_MAX_RICE_PARAM; k++) { /* :20-33, first minimum wins */
        uint64_t bits = 0;
        for (int i = 0; i < n; i++)
            bits += (zigzag(in[i]) >> k) + 1 + k;
        if (k == 0 || bits < best_bits) {
            best_bits = bits;
            best_k = k;
        }
    }
    *k_out = best_k;
    /* :37,:63  requiredInts = ceil((float)requiredBits / 32) */
    const uint64_t nwords = (uint64_t)ceilf((float)best_bits / 32);
    if (nwords > (uint64_t)(cap > 0 ? cap : 0))
        return -1;
    memset(words, 0, 4 * (size_t)nwords);
    uint64_t pos = 0;
    for (int i = 0; i < n; i++) { /* :41-53 */
        const uint64_t u = zigzag(in[i]);
        if (u >> 32) {
            raise_flag(flags, SELA_ORACLE_FLAG_RICE_RANGE);
            return -1;
        }
        uint64_t ones = u >> best_k;
        while (ones >= 32) {
            put_bits(words, &pos, 0xFFFFFFFFu, 32);
            ones -= 32;
        }
        put_bits(words, &pos, ((uint64_t)1 << ones) - 1, (unsigned)ones + 1); /* ones then a zero */
        put_bits(words, &pos, bitrev((uint32_t)u & ((1u << best_k) - 1u), best_k), best_k); /* MSB first */
    }
    return (int)nwords;
}

/* ---- Golomb-Rice decode -------------------------------------------------------------------
 * src/rice/rice_decoder.cpp:11-61. */
void sela_oracle_rice_decode(const uint32_t* words, int nwords, int n, uint32_t k, int32_t* out, uint32_t* flags)
{
    const uint64_t total = (uint64_t)nwords * 32;
    uint64_t pos = 0;
#define BIT_AT(p) ((p) < total ? (words[(p) >> 5] >> ((p)&31)) & 1u : (raise_flag(flags, SELA_ORACLE_FLAG_RICE_OVERRUN), 0u))
    for (int c = 0; c < n; c++) {
        uint32_t ones = 0; /* :29-33 */
        while (BIT_AT(pos) == 1) {
            ones++;
            pos++;
        }
        pos++;
        uint64_t u = (uint64_t)(uint32_t)(k < 32 ? ones << k : 0); /* :35, uint32 shift */
        for (uint32_t i = 1; i < k + 1; i++) { /* :37-40 */
            u |= (uint64_t)BIT_AT(pos) << (k - i);
            pos++;
        }
        /* :46-52 */
        out[c] = (int32_t)((u & 1) ? -(int64_t)((u + 1) >> 1) : (int64_t)(u >> 1));
    }
#undef BIT_AT
}

/* ---- frame encode ---------------------------------------------------------------------------
 * src/frame/frame_encoder.cpp:11-102, serialised like src/file/sela_file.cpp:115-135. */
typedef struct {
    int order;
    uint32_t ck, rk;
    int cwords, rwords;
    uint32_t* cw; /* coefficient words */
    uint32_t* rw; /* residue words */
} coded_block;

static void code_block(const int32_t* s, int n, coded_block* b, uint32_t* flags)
{
    int32_t q[SELA_MAX_LPC_ORDER];
    int64_t a[SELA_MAX_LPC_ORDER + 1];
    int32_t* r = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    b->order = sela_oracle_lpc_analyze(s, n, q, a, r, NULL, flags);
    b->cw = (uint32_t*)calloc(128, 4);
    b->cwords = sela_oracle_rice_encode(q, b->order, &b->ck, b->cw, 128, flags);
    int cap = n * 2 + 64;
    for (;;) {
        b->rw = (uint32_t*)calloc((size_t)cap, 4);
        b->rwords = sela_oracle_rice_encode(r, n, &b->rk, b->rw, cap, flags);
        if (b->rwords >= 0 || cap > (1 << 26))
            break;
        free(b->rw);
        cap *= 8;
    }
    free(r);
}

static uint8_t* put_subframe(uint8_t* p, uint8_t channel, uint8_t type, uint8_t parent, const coded_block* b, int n)
{
    const uint16_t cwords = (uint16_t)b->cwords, rwords = (uint16_t)b->rwords, nn = (uint16_t)n;
    *p++ = channel;
    *p++ = type;
    *p++ = parent;
    *p++ = (uint8_t)b->ck;
    memcpy(p, &cwords, 2), p += 2;
    *p++ = (uint8_t)b->order;
    memcpy(p, b->cw, 4 * (size_t)b->cwords), p += 4 * (size_t)b->cwords;
    *p++ = (uint8_t)b->rk;
    memcpy(p, &rwords, 2), p += 2;
    memcpy(p, &nn, 2), p += 2;
    memcpy(p, b->rw, 4 * (size_t)b->rwords), p += 4 * (size_t)b->rwords;
    return p;
}

size_t sela_oracle_frame_bound(uint32_t channels, uint32_t n)
{
    /* generous: 64 bits per sample + coefficient words + headers */
    return 4 + (size_t)channels * (SELA_SUBFRAME_HEADER_BYTES + 4 * 128 + 8 * (size_t)n + 256);
}

size_t sela_oracle_frame_encode_i32(const int32_t* planar, uint32_t channels, uint32_t n, uint8_t* out, uint32_t* flags)
{
    /* data::WavFrame carries int32 samples per channel (src/include/data/wav_frame.hpp:8-16) and nothing in
     * src/frame/frame_encoder.cpp:11-102 depends on their number or range: any n, any 32-bit values. */
    uint8_t* p = out;
    const uint32_t sync = SELA_SYNC_WORD;
    memcpy(p, &sync, 4), p += 4;
    int32_t* dif = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    for (uint32_t c = 0; c < channels; c++) {
        const int32_t* cur = planar + (size_t)c * n;
        coded_block act;
        code_block(cur, (int)n, &act, flags);
        if (c == 1 && channels == 2) { /* frame_encoder.cpp:18 -- exactly-stereo second channel */
            for (uint32_t j = 0; j < n; j++) /* :22-24, int32 subtraction (wraps like the x86 build) */
                dif[j] = (int32_t)((uint32_t)planar[j] - (uint32_t)planar[(size_t)n + j]);
            coded_block dc;
            code_block(dif, (int)n, &dc, flags);
            if ((size_t)dc.cwords + (size_t)dc.rwords < (size_t)act.cwords + (size_t)act.rwords) /* :64-66 */
                p = put_subframe(p, (uint8_t)c, 1, (uint8_t)(c - 1), &dc, (int)n);
            else
                p = put_subframe(p, (uint8_t)c, 0, (uint8_t)c, &act, (int)n);
            free(dc.cw);
            free(dc.rw);
        } else {
            p = put_subframe(p, (uint8_t)c, 0, (uint8_t)c, &act, (int)n); /* :96 */
        }
        free(act.cw);
        free(act.rw);
    }
    free(dif);
    return (size_t)(p - out);
}

size_t sela_oracle_frame_encode_ragged(const int32_t* samples, const uint32_t* lengths, uint32_t channels, uint8_t* out, uint32_t* flags)
{
    /* src/frame/frame_encoder.cpp:11-102 on a data::WavFrame whose channels differ in length: every channel is analysed at
     * its own samples[i].size() (:73-98); the second channel of an exactly-stereo frame takes the difference over ITS OWN
     * length, reading channel 0 up to there (:20-24) -- so channel 0 must be at least as long (the reference indexes past
     * its vector otherwise: the caller's business).  Every subframe carries its own samplesPerChannel (residueData.dataCount,
     * src/include/data/sela_sub_frame.hpp:41).  samples = the channels back to back. */
    uint8_t* p = out;
    const uint32_t sync = SELA_SYNC_WORD;
    memcpy(p, &sync, 4), p += 4;
    const int32_t* cur = samples;
    const int32_t* first = samples;
    for (uint32_t c = 0; c < channels; c++) {
        const uint32_t n = lengths[c];
        coded_block act;
        code_block(cur, (int)n, &act, flags);
        if (c == 1 && channels == 2) {
            int32_t* dif = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
            for (uint32_t j = 0; j < n; j++)
                dif[j] = (int32_t)((uint32_t)first[j] - (uint32_t)cur[j]);
            coded_block dc;
            code_block(dif, (int)n, &dc, flags);
            if ((size_t)dc.cwords + (size_t)dc.rwords < (size_t)act.cwords + (size_t)act.rwords)
                p = put_subframe(p, (uint8_t)c, 1, (uint8_t)(c - 1), &dc, (int)n);
            else
                p = put_subframe(p, (uint8_t)c, 0, (uint8_t)c, &act, (int)n);
            free(dc.cw);
            free(dc.rw);
            free(dif);
        } else {
            p = put_subframe(p, (uint8_t)c, 0, (uint8_t)c, &act, (int)n);
        }
        free(act.cw);
        free(act.rw);
        cur += n;
    }
    return (size_t)(p - out);
}

size_t sela_oracle_frame_encode(const int16_t* pcm, uint32_t channels, uint32_t n, uint8_t* out, uint32_t* flags)
{
    /* the demux of src/file/wav_file.cpp:194-200, then the frame encoder */
    int32_t* planar = (int32_t*)malloc(sizeof(int32_t) * ((size_t)channels * n + 1));
    for (uint32_t c = 0; c < channels; c++)
        for (uint32_t j = 0; j < n; j++)
            planar[(size_t)c * n + j] = pcm[(size_t)j * channels + c];
    const size_t used = sela_oracle_frame_encode_i32(planar, channels, n, out, flags);
    free(planar);
    return used;
}

/* ---- frame decode ---------------------------------------------------------------------------
 * src/frame/frame_decoder.cpp:11-72 reading the layout of src/file/sela_file.cpp:58-91. */
typedef struct {
    uint8_t channel, type, parent, ck, order, rk;
    uint16_t cwords, rwords, n;
    const uint8_t* cw;
    const uint8_t* rw;
} sub_view;

static const uint8_t* view_subframe(const uint8_t* p, sub_view* v)
{
    v->channel = p[0];
    v->type = p[1];
    v->parent = p[2];
    v->ck = p[3];
    memcpy(&v->cwords, p + 4, 2);
    v->order = p[6];
    v->cw = p + 7;
    p += 7 + 4 * (size_t)v->cwords;
    v->rk = p[0];
    memcpy(&v->rwords, p + 1, 2);
    memcpy(&v->n, p + 3, 2);
    v->rw = p + 5;
    return p + 5 + 4 * (size_t)v->rwords;
}

static void decode_sub(const sub_view* v, int32_t* s, uint32_t* flags)
{
    int32_t q[256];
    uint32_t* cw = (uint32_t*)malloc(4 * (size_t)v->cwords + 4);
    uint32_t* rw = (uint32_t*)malloc(4 * (size_t)v->rwords + 4);
    int32_t* r = (int32_t*)malloc(4 * (size_t)v->n + 4);
    memcpy(cw, v->cw, 4 * (size_t)v->cwords);
    memcpy(rw, v->rw, 4 * (size_t)v->rwords);
    if (v->n == 0 || v->n <= v->order) /* samples[0] = residues[0]; samples[1 .. order] (src/lpc/sample_generator.cpp:14-22): past the vectors */
        raise_flag(flags, SELA_ORACLE_FLAG_SHORT_BLOCK);
    sela_oracle_rice_decode(cw, v->cwords, v->order, v->ck, q, flags);
    sela_oracle_rice_decode(rw, v->rwords, v->n, v->rk, r, flags);
    int order = v->order > SELA_MAX_LPC_ORDER ? SELA_MAX_LPC_ORDER : v->order;
    sela_oracle_lpc_synth(order, q, r, v->n, s, flags);
    free(cw);
    free(rw);
    free(r);
}

size_t sela_oracle_frame_decode_i32(const uint8_t* in, uint32_t channels, int32_t* out, uint32_t stride, uint32_t* counts, uint32_t* flags)
{
    /* out[c][0 .. counts[c]) = allSamples[c] of src/frame/frame_decoder.cpp:13-71: every subframe brings its own
     * samplesPerChannel (:24-25, :48-49), the values stay 32-bit (:64-71).  A channel no subframe wrote keeps count 0.
     * Where the reference indexes out of bounds (a channel or parent number >= the number of subframes, a parent shorter
     * than the difference, more samples than `stride`) the subframe is skipped / cut and BAD_FRAME raised. */
    sub_view* v = (sub_view*)malloc(sizeof(sub_view) * (channels ? channels : 1));
    const uint8_t* p = in + 4;
    for (uint32_t c = 0; c < channels; c++) {
        p = view_subframe(p, &v[c]);
        counts[c] = 0;
    }
    int32_t* tmp = (int32_t*)malloc(4 * (size_t)65536);
    for (uint32_t c = 0; c < channels; c++) /* :17-37 independent subframes first */
        if (v[c].type == 0) {
            if (v[c].channel >= channels || v[c].n > stride) {
                raise_flag(flags, SELA_ORACLE_FLAG_BAD_FRAME);
                continue;
            }
            decode_sub(&v[c], out + (size_t)v[c].channel * stride, flags);
            counts[v[c].channel] = v[c].n;
        }
    for (uint32_t c = 0; c < channels; c++) /* :40-69 then dependent ones, in subframe order */
        if (v[c].type == 1) {
            if (v[c].channel >= channels || v[c].parent >= channels || v[c].n > stride || counts[v[c].parent] < v[c].n) {
                raise_flag(flags, SELA_ORACLE_FLAG_BAD_FRAME);
                continue;
            }
            decode_sub(&v[c], tmp, flags);
            int32_t* dst = out + (size_t)v[c].channel * stride;
            const int32_t* par = out + (size_t)v[c].parent * stride;
            for (uint32_t i = 0; i < v[c].n; i++) /* :64-66 */
                dst[i] = (int32_t)((uint32_t)par[i] - (uint32_t)tmp[i]);
            counts[v[c].channel] = v[c].n;
        }
    free(tmp);
    free(v);
    return (size_t)(p - in);
}

size_t sela_oracle_frame_decode(const uint8_t* in, uint32_t channels, int16_t* pcm, uint32_t* flags)
{
    uint16_t n16;
    memcpy(&n16, in + 4 + 7 + 4 * (size_t)(in[4 + 4] | (in[4 + 5] << 8)) + 3, 2); /* the first subframe's samplesPerChannel */
    const uint32_t n = n16;
    int32_t* all = (int32_t*)calloc((size_t)channels * (n ? n : 1), 4);
    uint32_t* counts = (uint32_t*)calloc(channels ? channels : 1, 4);
    const size_t used = sela_oracle_frame_decode_i32(in, channels, all, n, counts, flags);
    for (uint32_t i = 0; i < n; i++) /* truncation to 16 bits as in src/file/wav_file.cpp:248-251 */
        for (uint32_t c = 0; c < channels; c++)
            pcm[(size_t)i * channels + c] = (int16_t)(uint16_t)all[(size_t)c * n + i];
    free(all);
    free(counts);
    return used;
}

/* ---- batch drivers: the reference's static contiguous partition over T threads -------------
 * src/sela/encoder.cpp:58-73 / src/sela/decoder.cpp:58-73: framesPerThread = N / T, thread
 * i < T-1 takes [i*fpt, (i+1)*fpt), the last thread takes the rest. */
typedef struct {
    const int16_t* pcm;
    int16_t* pcm_out;
    const uint8_t* blob;
    const uint64_t* offsets;
    uint32_t begin, end, channels, n;
    uint8_t** frames; /* per-frame heap buffers (encode) */
    size_t* sizes;
} job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* encode_worker(void* arg)
{
    job* j = (job*)arg;
    const size_t bound = sela_oracle_frame_bound(j->channels, j->n);
    for (uint32_t f = j->begin; f < j->end; f++) {
        j->frames[f] = (uint8_t*)malloc(bound);
        j->sizes[f] = sela_oracle_frame_encode(j->pcm + (size_t)f * j->n * j->channels, j->channels, j->n, j->frames[f], NULL);
    }
    return NULL;
}

static void* decode_worker(void* arg)
{
    job* j = (job*)arg;
    for (uint32_t f = j->begin; f < j->end; f++)
        sela_oracle_frame_decode(j->blob + j->offsets[f], j->channels, j->pcm_out + (size_t)f * j->n * j->channels, NULL);
    return NULL;
}

static double fan_out(void* (*fn)(void*), job* proto, uint32_t n_frames, uint32_t threads)
{
    if (threads == 0)
        threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    job* jobs = (job*)malloc(sizeof(job) * threads);
    const uint32_t per = n_frames / threads;
    const double t0 = now_s();
    for (uint32_t t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].begin = t * per;
        jobs[t].end = (t == threads - 1) ? n_frames : (t + 1) * per;
        pthread_create(&th[t], NULL, fn, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; t++)
        pthread_join(th[t], NULL);
    const double t1 = now_s();
    free(th);
    free(jobs);
    return t1 - t0;
}

double sela_oracle_encode_frames_mt(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t n,
    uint32_t threads, uint8_t* out, uint64_t* offsets)
{
    job proto;
    memset(&proto, 0, sizeof proto);
    proto.pcm = pcm;
    proto.channels = channels;
    proto.n = n;
    proto.frames = (uint8_t**)calloc(n_frames ? n_frames : 1, sizeof(uint8_t*));
    proto.sizes = (size_t*)calloc(n_frames ? n_frames : 1, sizeof(size_t));
    const double secs = fan_out(encode_worker, &proto, n_frames, threads);
    uint64_t off = 0;
    for (uint32_t f = 0; f < n_frames; f++) { /* ordered concatenation, encoder.cpp:75-84 */
        offsets[f] = off;
        memcpy(out + off, proto.frames[f], proto.sizes[f]);
        off += proto.sizes[f];
        free(proto.frames[f]);
    }
    offsets[n_frames] = off;
    free(proto.frames);
    free(proto.sizes);
    return secs;
}

double sela_oracle_decode_frames_mt(const uint8_t* in, const uint64_t* offsets, uint32_t n_frames,
    uint32_t channels, uint32_t n, uint32_t threads, int16_t* pcm)
{
    job proto;
    memset(&proto, 0, sizeof proto);
    proto.blob = in;
    proto.offsets = offsets;
    proto.pcm_out = pcm;
    proto.channels = channels;
    proto.n = n;
    return fan_out(decode_worker, &proto, n_frames, threads);
}
