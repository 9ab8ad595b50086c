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
 *              One-shot calls (whole batch in memory) and streaming jobs (begin / feed / end: the
 *              caller keeps reading its file while earlier pieces are already on the device).
 *              Buffers from sela_hip_host_alloc() are page-locked: copies from and to them are truly
 *              asynchronous; ordinary (pageable) memory works too, only slower.
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
#define SELA_HIP_EINVAL (-2)   /* bad argument (channels == 0, samples_per_channel == 0 or > 65535, ...) */
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
/* The host-pointer API keeps staging buffers, streams and events per calling thread; creating them costs the runtime
 * about 10 ms.  sela_hip_thread_release() PARKS the calling thread's set for the next thread that uses the same device
 * (a thread that simply ends parks it too), so short-lived worker threads start warm.  sela_hip_shutdown() frees the
 * calling thread's set and every parked one, and returns the idle page-locked blocks to the system. */
void sela_hip_thread_release(void);
void sela_hip_shutdown(void);
const char* sela_hip_last_error(void);
int sela_hip_device_count(void);
/* Page-locked host memory for the buffers handed to the host-pointer API (PCM in, frames out, ...).
 * Freed blocks are pooled and handed out again (pinning is slow); sela_hip_shutdown() returns them to the
 * system.  Without a usable HIP device these fall back to ordinary aligned memory, so container code
 * (WAV / .sela parsing) that holds its data in such buffers still runs on a CPU-only box. */
void* sela_hip_host_alloc(size_t bytes);
void sela_hip_host_free(void* p);

/* ---- sizing ------------------------------------------------------------------------------------ */
/* Number of signals analysed per frame: channels, +1 for exactly-stereo input. */
uint32_t sela_hip_signals_per_frame(uint32_t channels);
/* Bytes of device workspace the *_device calls need for a batch of n_frames.  The workspace needs no
 * initialisation and may be reused by later calls; one call at a time may use it. */
size_t sela_hip_encode_workspace_bytes(uint32_t n_frames, uint32_t channels);
/* (The decoder keeps positions and samples on chip; the workspace is only written for subframes that take the
 * kernels' generic mode -- Rice streams beyond what 16-bit audio produces.) */
size_t sela_hip_decode_workspace_bytes(uint32_t n_frames, uint32_t channels);
/* Most channels the decoder takes: 255, what the 8-bit channel field of the .sela header can say (up to eight channels
 * take one wave per subframe, more take the same waves in rounds; src/frame/frame_decoder.cpp:11-72). */
uint32_t sela_hip_decode_max_channels(void);
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
 * bits, [1] = number of malformed frames (zeroed by the call).  channels <= sela_hip_decode_max_channels().
 * One kernel on `stream` (one workgroup per frame, one wave per subframe); d_workspace as sized by
 * sela_hip_decode_workspace_bytes().  Returns immediately.
 */
int sela_hip_decode_device(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint32_t n_frames,
    uint32_t channels, int16_t* d_pcm_out, uint32_t* d_status, void* d_workspace, size_t workspace_bytes,
    void* stream);

/* ---- host-pointer API (synchronous) -------------------------------------------------------------- */
/* frames_out must hold sela_hip_encode_bound_bytes() or the call may return SELA_HIP_ECAPACITY.
 * These are begin + feed(everything) + end of the streaming jobs below, on library-owned streams: an encode is ONE
 * kernel launch per feed (it fetches the PCM from page-locked memory itself and stores the finished frames into
 * frames_out), a decode a pipeline of 1024-frame chunks (copy in / kernel / copy out overlapped).  Results are
 * identical to one device-pointer call on the whole batch.
 * Calls of at most 32 frames made from several threads at once -- a binding that keeps the reference's per-frame thread
 * loop (src/sela/encoder.cpp:58-73) -- are coalesced: calls that arrive while another one is on the device go there
 * together, as one job, when it returns.  Every call still gets exactly its own result and its own error (a buffer that
 * is too small, a malformed frame); a lone caller is not delayed. */
int sela_hip_encode(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel,
    uint8_t* frames_out, size_t frames_cap, uint64_t* frame_offsets_out /* [n_frames+1] */);
int sela_hip_decode(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels,
    int16_t* pcm_out);
/* Blocks of any length.  The reference's frame path does not know the number 2048: lpc::ResidueGenerator loops over
 * samples.size() (src/lpc/residue_generator.cpp:12-45,98-119) and a subframe carries its own samplesPerChannel, a u16
 * (src/include/data/sela_sub_frame.hpp:27, src/frame/frame_decoder.cpp:24-25,48-49); only its WAV reader cuts 2048-sample
 * frames.  So:
 *   sela_hip_encode() takes any samples_per_channel in 1 .. 65535 (pcm = [n_frames][samples_per_channel][channels]); other
 *     than 2048 goes through the any-length kernels (sela_generic.hip: the same arithmetic with a run-time length, one wave
 *     per block).  frames_out needs sela_hip_encode_bound_bytes_n().  A block that is not longer than the
 *     predictor order its own analysis picks makes the reference read past its vector (residue_generator.cpp:104-110):
 *     SELA_HIP_ERANGE.
 *   sela_hip_decode() takes a stream whose frames say anything in 0 .. 65535: when one says something other than 2048 the
 *     whole call goes through the any-length kernels, frame f's samples land at pcm_out + sample_offsets[f] * channels with
 *     sample_offsets[] as sela_hip_index_samples() reports them (for 2048 everywhere that is f * 2048, the layout above), and
 *     a frame whose channels disagree about the length is malformed (SELA_HIP_EFORMAT; the reference's WAV writer indexes
 *     past the shorter ones, src/file/wav_file.cpp:248-262).  The fast kernels are tried first, unasked (a stream of
 *     2048-sample frames pays nothing for the other kind) -- unless the stream's FIRST frame already says another length --
 *     and they may write up to [n_frames][2048][channels] before they find an odd frame further on: pcm_out must hold
 *     max(n_frames * 2048, sample_offsets[n_frames]) * channels samples.
 * The streaming jobs and the device-pointer calls below stay what they are: the fast path for what the reference's CLI writes
 * (2048 everywhere); a stream with another length gets SELA_HIP_EFORMAT from them, and the caller comes here. */
size_t sela_hip_encode_bound_bytes_n(uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel);
/* sample_offsets[f] = samples per channel before frame f (its first subframe's samplesPerChannel counts for the frame),
 * [n_frames + 1] entries; returns the largest samplesPerChannel any subframe of the stream names (0 for a stream the walk
 * cannot follow: the decode calls report that as SELA_HIP_EFORMAT). */
uint32_t sela_hip_index_samples(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels,
    uint64_t* sample_offsets);

/* ---- the frame classes' own value types: 32-bit samples, any length ---------------------------------------------------------
 * frame::FrameEncoder(const data::WavFrame&).process() and frame::FrameDecoder(const data::SelaFrame&).process()
 * (src/include/frame.hpp:8-24) on what they really take and return: data::WavFrame = int32 samples per channel
 * (src/include/data/wav_frame.hpp:8-16), nothing narrowed (src/frame/frame_decoder.cpp:64-71; only file::WavFile::writeToFile
 * truncates to 16 bits).  The encoder always runs the any-length kernels (sela_generic.hip: the fast kernels' loops with a
 * run-time length); the decoder runs the fast decoder's lane-parallel parse and tuned synthesis with 32-bit samples and a
 * run-time length (k_decode_subframes32: one piece for subframes of at most 2048 samples that fit the parser's plan, segments for
 * everything else) and leaves to a serial kernel only the streams it will not judge (frames that are not whole words, malformed
 * headers, Rice streams that run dry, coefficients outside the tables).  Results identical to the calls above wherever both apply.
 *   samples      [n_frames][channels][samples_per_channel] (planar per frame: WavFrame.samples[c][i]), 1 .. 65535 per channel.
 *   samples_out  [n_frames][channels][stride]: channel c of frame f at ((f * channels) + c) * stride, counts_out[f * channels + c]
 *                of them valid (0 for a channel no subframe of the frame names; what lies behind a channel's count is not
 *                defined); stride >= the largest samplesPerChannel in the stream (sela_hip_index_samples() returns it) or
 *                SELA_HIP_ECAPACITY.  The calls run on a stream of the calling thread's own (leased, not the default stream)
 *                and return when the result is in host memory; after an error the outputs' contents are not defined.
 * Errors as above; values whose int32 zig-zag overflows in the reference (|residue| >= 2^30) and Rice streams beyond the u16
 * word count of a subframe are SELA_HIP_ERANGE. */
int sela_hip_encode_i32(const int32_t* samples, uint32_t n_frames, uint32_t channels, uint32_t samples_per_channel,
    uint8_t* frames_out, size_t frames_cap, uint64_t* frame_offsets_out /* [n_frames+1] */);
int sela_hip_decode_i32(const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t channels,
    int32_t* samples_out, uint32_t stride, uint32_t* counts_out /* [n_frames * channels] */);
/* ONE frame whose channels differ in length, as frame::FrameEncoder::process codes it (src/frame/frame_encoder.cpp:11-102):
 * every channel is analysed at its own samples[i].size() (:73-98) and its subframe carries that as samplesPerChannel
 * (src/include/data/sela_sub_frame.hpp:41); the second channel of an exactly-stereo frame is also tried as the difference
 * channel 0 - channel 1 over ITS OWN length (:20-24), so channel 0 must be at least as long -- where it is shorter the
 * reference indexes past its vector, and this call returns SELA_HIP_EINVAL.  The decoder has always taken such frames
 * (sela_hip_decode_i32).
 *   samples      the channels back to back: lengths[0] samples of channel 0, then lengths[1] of channel 1, ...
 *   lengths      [channels], each 1 .. 65535
 *   frame_out    the frame's bytes (sync word + subframes); 4 + the sum over the channels of
 *                sela_hip_encode_bound_bytes_n(1, 1, lengths[c]) holds any frame of samples within 17 bits (SELA_HIP_ECAPACITY
 *                when a frame does not fit); *frame_bytes receives the size.
 * Errors as sela_hip_encode_i32. */
int sela_hip_encode_ragged_i32(const int32_t* samples, const uint32_t* lengths, uint32_t channels, uint8_t* frame_out, size_t frame_cap,
    size_t* frame_bytes);

/* ---- streaming jobs (host pointers) -------------------------------------------------------------------
 * For callers that produce their input piece by piece (a file being read): feed() enqueues a piece and
 * returns at once -- from page-locked buffers nothing in it waits for the device (an encode feed is one kernel
 * launch; a decode feed waits when one of its eight chunk buffer sets comes round again) -- so the caller's next
 * read runs beside the device work.  A piece's buffer must stay valid and unchanged until the job reports its
 * frames final (or ends).  Pieces -- and a decode job's pcm_out -- in ordinary (pageable) memory go through page-locked
 * bounce buffers of the library (one more host copy: slower, same results).  One open job per calling thread; a job is
 * used from the thread that began it.
 * An encode feed normally has its PCM fetched by a staging kernel beside the encode launch.  If the device is so busy
 * with other work that the two cannot run side by side within the launch's bounded wait (or another thread's job on
 * this device is using that path), the feed -- and any queued behind it -- is issued again with the copy engine in
 * place of the staging kernel: a busy device costs time, never the result.
 *
 * encode: the job appends to frames_out (capacity frames_cap) and fills frame_offsets_out[0 .. total_frames];
 * *frames_final / *bytes_final (optional) report how much of both is complete in host memory, so a writer can
 * drain finished bytes to disk while later pieces are still being encoded (src/file/sela_file.cpp:105-137 is
 * what it replaces).  end() waits for everything, reports the totals, and frees the job -- also after an error.
 * decode: pieces are whole frames (frame_offsets[0 .. n_frames] index into `frames`); pcm_out fills in order.
 * Errors are those of the one-shot calls. */
typedef struct sela_hip_job sela_hip_job;
int sela_hip_encode_begin(sela_hip_job** job, uint32_t channels, uint32_t total_frames, uint8_t* frames_out, size_t frames_cap,
    uint64_t* frame_offsets_out /* [total_frames + 1] */);
int sela_hip_encode_feed(sela_hip_job* job, const int16_t* pcm, uint32_t n_frames, uint32_t* frames_final, uint64_t* bytes_final);
int sela_hip_encode_end(sela_hip_job* job, uint32_t* frames_final, uint64_t* bytes_final);
int sela_hip_decode_begin(sela_hip_job** job, uint32_t channels, uint32_t total_frames, int16_t* pcm_out);
int sela_hip_decode_feed(sela_hip_job* job, const uint8_t* frames, const uint64_t* frame_offsets, uint32_t n_frames, uint32_t* frames_final);
int sela_hip_decode_end(sela_hip_job* job, uint32_t* frames_final);

/* Walk a frame byte stream on the host and fill frame_offsets[0..n_frames]; stops at the first bad
 * sync word like src/file/sela_file.cpp:54-56.  Returns the number of frames found (<= n_frames). */
uint32_t sela_hip_index_frames(const uint8_t* frames, size_t frames_bytes, uint32_t n_frames, uint32_t channels,
    uint64_t* frame_offsets);

/* ---- the stages on their own -------------------------------------------------------------------------------
 * The reference's public L1 classes (src/include/lpc.hpp:73-117, src/include/rice.hpp:9-43; its tests call them directly:
 * test/lpctests.cpp:10-32, test/ricetests.cpp:7-25), batched, on HOST pointers.  Not the fast path -- a frame goes through
 * all of them inside one kernel (sela_hip_encode / sela_hip_decode) -- but the same device code, for callers and tests of
 * a stage by itself.  Every call synchronises.
 *
 * lpc::ResidueGenerator::process (src/lpc/residue_generator.cpp:121-134): n_blocks blocks of 2048 samples (int32) -> per block the order, the quantised reflection
 * coefficients (q_out[block][0 .. order), the rest zeroed) and 2048 residues.  Samples beyond 17 bits are taken too (through the
 * any-length kernels, like sela_hip_lpc_encode_n).  A Rice stream too long for a frame's slot is not this stage's business:
 * the residues are returned whatever their size. */
int sela_hip_lpc_encode(const int32_t* samples, uint32_t n_blocks, int32_t* order_out, int32_t* q_out /* [n_blocks][100] */, int32_t* residues_out);
/* The same for blocks of samples_per_block samples (1 .. 2^24) of any 32-bit value -- the class takes any vector
 * (src/lpc/residue_generator.cpp:6-10).  q_out[block][order .. 100) is zeroed.  A block not longer than the order its analysis
 * picks: SELA_HIP_ERANGE (the reference reads past its vector, residue_generator.cpp:104-110). */
int sela_hip_lpc_encode_n(const int32_t* samples, uint32_t n_blocks, uint32_t samples_per_block, int32_t* order_out, int32_t* q_out /* [n_blocks][100] */,
    int32_t* residues_out);
/* lpc::SampleGenerator::process (src/lpc/sample_generator.cpp:11-39): the inverse.  q[block][0 .. order[block]); samples_out
 * [n_blocks][2048] as the 32-bit values the reference returns.  coefs_out (or NULL): [n_blocks][101], the Q35 predictor
 * a[0 .. order] of lpc::LinearPredictor::generatelinearPredictionCoefficients (src/lpc/linear_predictor.cpp:30-61);
 * samples_out may be NULL when only the predictor is wanted. */
int sela_hip_lpc_decode(const int32_t* order, const int32_t* q /* [n_blocks][100] */, const int32_t* residues, uint32_t n_blocks, int32_t* samples_out,
    int64_t* coefs_out);
/* The same for blocks of samples_per_block residues (1 .. 2^24). */
int sela_hip_lpc_decode_n(const int32_t* order, const int32_t* q /* [n_blocks][100] */, const int32_t* residues, uint32_t n_blocks, uint32_t samples_per_block,
    int32_t* samples_out, int64_t* coefs_out);
/* rice::RiceEncoder::process (src/rice/rice_encoder.cpp:73-81): n_streams streams of int32 values, stream i =
 * values[value_offsets[i] .. value_offsets[i + 1]) -> its Rice parameter k_out[i] (the first minimum over 0..19), its
 * word count word_counts_out[i] (ceil((float)bits / 32) as the reference computes it) and its words at
 * words_out[word_offsets[i] ..] (word_offsets[i + 1] - word_offsets[i] words of room: SELA_HIP_ECAPACITY if a stream needs
 * more -- the counts are valid then).  |value| >= 2^30 (the reference's int32 zig-zag overflows there): SELA_HIP_ERANGE. */
int sela_hip_rice_encode(const int32_t* values, const uint64_t* value_offsets, uint32_t n_streams, uint32_t* k_out, uint32_t* word_counts_out,
    uint32_t* words_out, const uint64_t* word_offsets);
/* rice::RiceDecoder::process (src/rice/rice_decoder.cpp:54-61): stream i = words[word_offsets[i] .. word_offsets[i + 1]) with
 * parameter k[i] (< 32) -> value_offsets[i + 1] - value_offsets[i] values at values_out[value_offsets[i] ..].  A stream that
 * ends before its values do decodes the missing bits as zeros and the call returns SELA_HIP_EFORMAT. */
int sela_hip_rice_decode(const uint32_t* words, const uint64_t* word_offsets, const uint32_t* k, const uint64_t* value_offsets, uint32_t n_streams,
    int32_t* values_out);

/* ---- per-kernel timing (measurement hook used by bench.py) ------------------------------------------
 * When enabled, the *_device calls of the calling thread bracket each kernel launch with HIP events
 * recorded on the caller's stream.  sela_hip_kernel_times() waits for the events of the most recent
 * encode (3 kernels: blocks, plan, assemble) or decode (1 kernel) call and returns their durations
 * in milliseconds; it returns the number of kernels reported (0 if timing was off). */
void sela_hip_enable_kernel_timing(int enable);
int sela_hip_kernel_times(float* ms_out, int capacity);

/* ---- flag bits reported through d_status[0] / sela_hip_trace.flags --------------------------------- */
#define SELA_HIP_FLAG_Q_RANGE 1u       /* quantised reflection coefficient outside [-64,63] (clamped) */
#define SELA_HIP_FLAG_COEF_OVERFLOW 2u /* |2^35 * coefficient| >= 2^63 */
#define SELA_HIP_FLAG_RICE_RANGE 4u    /* zig-zag residue does not fit 32 bits */
#define SELA_HIP_FLAG_RICE_OVERRUN 8u  /* decoder ran past the end of a Rice stream */
#define SELA_HIP_FLAG_WORDS_CAP 16u    /* a Rice stream exceeded the per-block slot (encoder) */
#define SELA_HIP_FLAG_BAD_FRAME 32u    /* bad sync word / inconsistent subframe header (decoder) */
#define SELA_HIP_FLAG_INTERNAL 64u     /* a bounded wait inside a kernel ran out (never expected; reported as SELA_HIP_ENODEV) */
#define SELA_HIP_FLAG_SHORT_BLOCK 128u /* a block no longer than its own predictor order: the reference's warm-up loop reads past
                                        * its vector there (src/lpc/residue_generator.cpp:104-110); reported as SELA_HIP_ERANGE */

#ifdef __cplusplus
}
#endif
#endif /* SELA_HIP_H_ */
