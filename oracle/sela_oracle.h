/*
 * sela_oracle.h -- CPU restatement of the SELA frame encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a from-scratch, flat-buffer, plain-C
 * restatement of the reference's algorithm (sahaRatul/sela v2.0.2, src/lpc, src/rice,
 * src/frame).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product path (sela_amd/csrc, host/) never links,
 * imports or falls back to anything in oracle/.
 *
 * Pinning: tests/test_oracle_vs_reference.py checks every function below bit-for-bit
 * against the unmodified reference (oracle/_ref/libsela_ref.so, built by `make ref`) where
 * that library exists, and tests/test_oracle_golden.py checks it against the committed
 * fixtures in tests/golden/ (generated from the reference by tests/golden/make_golden.py)
 * everywhere else.
 */
#ifndef SELA_ORACLE_H_
#define SELA_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status bits accumulated by the functions below when the input drives the reference into
 * undefined behaviour (SURVEY.md App. E "(G)" items).  Parity is undefined when any is set. */
#define SELA_ORACLE_FLAG_Q_RANGE 1u      /* quantised coefficient outside [-64,63]: table index clamped */
#define SELA_ORACLE_FLAG_COEF_OVERFLOW 2u /* |2^35*coef| >= 2^63 in the step-up output */
#define SELA_ORACLE_FLAG_RICE_RANGE 4u   /* zig-zag value does not fit 32 bits */
#define SELA_ORACLE_FLAG_RICE_OVERRUN 8u /* Rice decoder ran past the end of its words */
#define SELA_ORACLE_FLAG_SHORT_BLOCK 128u /* decoder: a subframe with no samples, or not longer than its order (sample_generator.cpp:14-22 writes past its vector) */
#define SELA_ORACLE_FLAG_BAD_FRAME 32u   /* a subframe names a channel / parent that does not exist, or a parent shorter than itself */

/* FP64 intermediates of one analysis, exposed so that GPU kernels can be compared stage by
 * stage (the reference keeps these private). */
typedef struct {
    double mean;
    double ac[101]; /* normalised autocorrelation, ac[0] == 1.0 */
    double k[100];  /* reflection coefficients */
} sela_oracle_lpc_trace;

/* lpc::ResidueGenerator::process.  q[100], a[101], r[n]; returns the order (1..100). */
int sela_oracle_lpc_analyze(const int32_t* s, int n, int32_t* q, int64_t* a, int32_t* r,
    sela_oracle_lpc_trace* trace, uint32_t* flags);

/* lpc::LinearPredictor::{dequantizeReflectionCoefficients,generatelinearPredictionCoefficients}. */
void sela_oracle_lpc_coeffs(int order, const int32_t* q, int64_t* a, uint32_t* flags);

/* lpc::SampleGenerator::process. */
void sela_oracle_lpc_synth(int order, const int32_t* q, const int32_t* r, int n, int32_t* s, uint32_t* flags);

/* rice::RiceEncoder::process.  Returns the word count (or -1 if cap is too small). */
int sela_oracle_rice_encode(const int32_t* in, int n, uint32_t* k, uint32_t* words, int cap, uint32_t* flags);

/* rice::RiceDecoder::process. */
void sela_oracle_rice_decode(const uint32_t* words, int nwords, int n, uint32_t k, int32_t* out, uint32_t* flags);

/* frame::FrameEncoder::process on interleaved int16 PCM -> on-disk frame bytes. Returns bytes. */
size_t sela_oracle_frame_encode(const int16_t* pcm, uint32_t channels, uint32_t n, uint8_t* out, uint32_t* flags);

/* The same on the reference's own value type: data::WavFrame = int32 samples per channel, any number of them
 * (src/include/data/wav_frame.hpp:8-16).  planar = [channels][n]. */
size_t sela_oracle_frame_encode_i32(const int32_t* planar, uint32_t channels, uint32_t n, uint8_t* out, uint32_t* flags);

/* The same on a frame whose channels differ in length (src/frame/frame_encoder.cpp:20-24,73-98): samples = the channels back to
 * back, lengths[c] samples each; an exactly-stereo frame needs lengths[0] >= lengths[1] (the reference reads channel 0 up to
 * channel 1's length). */
size_t sela_oracle_frame_encode_ragged(const int32_t* samples, const uint32_t* lengths, uint32_t channels, uint8_t* out, uint32_t* flags);

/* frame::FrameDecoder::process as the reference returns it: out[c][0 .. counts[c]) = WavFrame.samples[c], 32-bit,
 * every subframe with its own samplesPerChannel; out is [channels][stride].  Returns bytes consumed. */
size_t sela_oracle_frame_decode_i32(const uint8_t* in, uint32_t channels, int32_t* out, uint32_t stride, uint32_t* counts, uint32_t* flags);

/* frame::FrameDecoder::process from on-disk frame bytes -> interleaved int16 (the first subframe's samplesPerChannel
 * of them per channel). Returns bytes consumed. */
size_t sela_oracle_frame_decode(const uint8_t* in, uint32_t channels, int16_t* pcm, uint32_t* flags);

/* Batch drivers with the reference's static contiguous thread partition
 * (src/sela/encoder.cpp:58-73, src/sela/decoder.cpp:58-73).  Return seconds spent in the
 * fan-out.  offsets has n_frames+1 entries (byte offset of each frame in the blob). */
double sela_oracle_encode_frames_mt(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t n,
    uint32_t threads, uint8_t* out, uint64_t* offsets);
double sela_oracle_decode_frames_mt(const uint8_t* in, const uint64_t* offsets, uint32_t n_frames,
    uint32_t channels, uint32_t n, uint32_t threads, int16_t* pcm);

/* Upper bound on the on-disk size of one frame that these functions can produce. */
size_t sela_oracle_frame_bound(uint32_t channels, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* SELA_ORACLE_H_ */
