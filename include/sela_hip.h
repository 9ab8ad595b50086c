/*
 * sela_hip.h -- C ABI of the MI355X-native SELA frame encode/decode path (libsela_hip.so).
 *
 * This is the drop-in boundary for the hot path.  The reference (sahaRatul/sela v2.0.2) has no
 * FFI of its own; the work these entry points replace is the worker fan-out plus everything
 * below it:
 *
 *   sela_hip_encode*  replaces  sela::Encoder::processFrames        src/sela/encoder.cpp:40-92
 *                     i.e. per frame frame::FrameEncoder::process   src/frame/frame_encoder.cpp:11-102
 *                          -> lpc::ResidueGenerator::process        src/lpc/residue_generator.cpp:121-134
 *                          -> rice::RiceEncoder::process            src/rice/rice_encoder.cpp:73-81
 *                     and the per-frame serialisation of            src/file/sela_file.cpp:115-135
 *   sela_hip_decode*  replaces  sela::Decoder::processFrames        src/sela/decoder.cpp:41-92
 *                     i.e. per frame frame::FrameDecoder::process   src/frame/frame_decoder.cpp:11-72
 *                          -> rice::RiceDecoder::process            src/rice/rice_decoder.cpp:54-61
 *                          -> lpc::SampleGenerator::process         src/lpc/sample_generator.cpp:32-39
 *                     and the interleave of                         src/file/wav_file.cpp:244-257
 *
 * Data formats at the boundary are the reference's own on-disk formats, so no conversion is
 * needed on either side:
 *   PCM     : interleaved little-endian int16, [n_frames][2048][channels] -- the WAV data chunk
 *             (src/file/wav_file.cpp:193-199), whole 2048-sample frames only (tail dropped by the
 *             caller exactly as src/file/wav_file.cpp:184,203 does).
 *   frames  : the byte stream that follows the 15-byte .sela file header: per frame the sync word
 *             0xAA55FF00 and `channels` subframes (src/file/sela_file.cpp:115-135), frames
 *             back to back.  frame_offsets[f] is the byte offset of frame f in that stream,
 *             frame_offsets[n_frames] its total size.  Every frame size is a multiple of 4.
 *
 * All functions return 0 on success or a negative SELA_HIP_E* code; they never throw and never
 * fall back to a CPU implementation.  sela_hip_last_error() gives a thread-local message.
 *
 * Two flavours:
 *   *_device : pointers are DEVICE pointers on the current HIP device, work is enqueued on
 *              `stream` (a hipStream_t passed as void*; NULL = the default stream) and the call
 *              returns without synchronising unless stated.  This is what bench.py times
 *              (inputs resident in HBM).
 *   host     : pointers are HOST pointers; the library stages through its own device buffers
 *              (H2D, kernels, D2H) and returns when the result is in host memory.  This is what
 *              the C++ host (host/) calls from sela::Encoder/Decoder and frame::Frame{En,De}coder.
 */
#ifndef SELA_HIP_H_
#define SELA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELA_HIP_OK 0
#define SELA_HIP_ENODEV (-1)   /* no usable HIP device / HIP runtime error (see last_error) */
#define SELA_HIP_EINVAL (-2)   /* bad argument (channels == 0, samples_per_channel != 2048, ...) */
#define SELA_HIP_ENOMEM (-3)   /* device or host allocation failed */
#define SELA_HIP_ECAPACITY (-4) /* caller-provided output or workspace too small */
#define SELA_HIP_EFORMAT (-5)  /* malformed frame stream (bad sync word, inconsistent sizes) */
#define SELA_HIP_ERANGE (-6)   /* a block left the range the format can carry (SURVEY.md App. E "(G)") */

#define SELA_HIP_SAMPLES_PER_FRAME 2048u

/* Per-(frame, signal) analysis record, optional debug output of the encoder (FP64 intermediates
 * that the reference keeps private).  signal 0..channels-1 are the channels, signal `channels`
 * is the difference channel0-channel1 of an exactly-stereo frame. */
typedef struct sela_hip_trace {
    double mean;
    double ac[101];
    double k[100];
    int64_t a[101];
    int32_t q[100];
    int32_t order;
    uint32_t coef_k, coef_words, res_k, res_words, flags;
} sela_hip_trace;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Select/initialise `device` (>= 0) for the calling thread, or -1 to keep the current device. */
int sela_hip_init(int device);
void sela_hip_shutdown(void);
const char* sela_hip_last_error(void);
int sela_hip_device_count(void);

/* ---- sizing ------------------------------------------------------------------------------------ */
/* Number of signals analysed per frame: channels, +1 for exactly-stereo input. */
uint32_t sela_hip_signals_per_frame(uint32_t channels);
/* Bytes of device workspace the *_device calls need for a batch of n_frames.  The workspace needs no
 * initialisation and may be reused by later calls; one call at a time may use it. */
size_t sela_hip_encode_workspace_bytes(uint32_t n_frames, uint32_t channels);
size_t sela_hip_decode_workspace_bytes(uint32_t n_frames, uint32_t channels);
/* Upper bound of the frame byte stream produced by encoding n_frames (what `frames_cap` must be
 * to be certain never to get SELA_HIP_ECAPACITY). */
size_t sela_hip_encode_bound_bytes(uint32_t n_frames, uint32_t channels);

/* ---- device-pointer API (asynchronous on `stream`) --------------------------------------------- */
/*
 * Encode n_frames frames.  d_pcm: int16 [n_frames][2048][channels].  Outputs: d_frames (byte stream,
 * capacity frames_cap bytes, 4-byte aligned), d_frame_offsets (uint64 [n_frames + 1]),
 * d_status (uint32[4]: [0] = OR of per-block flag bits, [1] = number of frames that did not fit
 * frames_cap, [2..3] reserved; zeroed by the call).  d_trace may be NULL.
 * Launches 3 kernels on `stream` and returns immediately.
 */
int sela_hip_encode_device(const int16_t* d_pcm, uint32_t n_frames, uint32_t channels,
    uint8_t* d_frames, size_t frames_cap, uint64_t* d_frame_offsets, uint32_t* d_status,
    void* d_workspace, size_t workspace_bytes, sela_hip_trace* d_trace, void* stream);

/*
 * Decode n_frames frames.  d_frames / d_frame_offsets as produced above (or by parsing a .sela
 * file).  d_pcm_out: int16 [n_frames][2048][channels].  d_status: uint32[4], [0] = OR of flag
 * bits, [1] = number of malformed frames.  d_workspace holds the parsed residues and the filter state
 * between the kernels (sela_hip_decode_workspace_bytes()).  The call is cut into four chunks along the
 * sample axis: the parse kernel of chunk j+1 runs on a library-owned side stream beside the synthesis
 * kernel of chunk j on `stream`; all of it is ordered after earlier work on `stream`, and later work on
 * `stream` is ordered after all of it.
 */
int sela_hip_decode_device(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames,
    uint32_t channels, int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, size_t workspace_bytes,
    void* stream);

/* ---- host-pointer API (synchronous) -------------------------------------------------------------- */
/* frames_out must hold sela_hip_encode_bound_bytes() or the call may return SELA_HIP_ECAPACITY.
 * Large batches run as a chunked pipeline on three library-owned streams (copy in / kernels / copy out
 * overlapped); results are identical to one call on the whole batch. */
int sela_hip_encode(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel,
    uint8_t* frames_out, size_t frames_cap, uint64_t* frame_offsets_out /* [n_frames+1] */);
int sela_hip_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* pcm_out);

/* Walk a frame byte stream on the host and fill frame_offsets[0..n_frames]; stops at the first bad
 * sync word like src/file/sela_file.cpp:54-56.  Returns the number of frames found (<= n_frames). */
uint32_t sela_hip_index_frames(const uint8_t* frames, size_t frames_bytes, uint32_t n_frames, uint32_t channels,
    uint64_t* frame_offsets);

/* ---- per-kernel timing (measurement hook used by bench.py) ------------------------------------------
 * When enabled, the *_device calls of the calling thread bracket each kernel launch with HIP events
 * recorded on the caller's stream.  sela_hip_kernel_times() waits for the events of the most recent
 * encode (3 kernels: blocks, plan, assemble) or decode (2 kernels: parse, synthesize) call and returns their durations
 * in milliseconds; it returns the number of kernels reported (0 if timing was off). */
void sela_hip_enable_kernel_timing(int enable);
int sela_hip_kernel_times(float* ms_out, int capacity);

/* Debug hook: while a non-NULL device buffer is set, the *_device calls of the calling thread run an
 * instrumented build of the kernels that stores s_memtime deltas per phase: 16 uint64 per
 * (frame, signal) for the encoder and per (frame, subframe) for the decoder.  Slower; never set
 * in the timed path. */
void sela_hip_debug_phase_buffer(uint64_t* d_cycles);

/* ---- flag bits reported through d_status[0] / sela_hip_trace.flags --------------------------------- */
#define SELA_HIP_FLAG_Q_RANGE 1u       /* quantised reflection coefficient outside [-64,63] (clamped) */
#define SELA_HIP_FLAG_COEF_OVERFLOW 2u /* |2^35 * coefficient| >= 2^63 */
#define SELA_HIP_FLAG_RICE_RANGE 4u    /* zig-zag residue does not fit 32 bits */
#define SELA_HIP_FLAG_RICE_OVERRUN 8u  /* decoder ran past the end of a Rice stream */
#define SELA_HIP_FLAG_WORDS_CAP 16u    /* a Rice stream exceeded the per-block slot (encoder) */
#define SELA_HIP_FLAG_BAD_FRAME 32u    /* bad sync word / inconsistent subframe header (decoder) */

#ifdef __cplusplus
}
#endif
#endif /* SELA_HIP_H_ */
