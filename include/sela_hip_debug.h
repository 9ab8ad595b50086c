/*
 * sela_hip_debug.h -- TEST HOOKS of libsela_hip.so.  NOT part of the drop-in boundary (include/sela_hip.h) and not
 * stable: tests/ and tools/ use them to send the PRODUCT kernels down branches real audio never takes and to read
 * instrumentation; nothing of host/ or of an integration calls them.  All but the last are per calling thread.
 */
#ifndef SELA_HIP_DEBUG_H_
#define SELA_HIP_DEBUG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Debug hook: while a non-NULL device buffer is set, the *_device calls of the calling thread run an
 * instrumented build of the kernels that stores s_memtime deltas per phase: 16 uint64 per
 * (frame, signal) for the encoder and per (frame, subframe) for the decoder.  Slower; never set
 * in the timed path. */
void sela_hip_debug_phase_buffer(uint64_t* d_cycles);
/* Debug hook: the three forms of the encoder's residue filter (sela_encode_tail.inc).  0: by the block -- one pass of FP64
 * taps where that is exact (2^34 + sum |a[j]| x max |s| below 2^53), else two passes (the coefficients' low 20 bits, then the
 * rest), else the plain 64-bit wrap-around loop (predictors beyond 2^39: never reached by 16-bit audio).  1: every
 * block of the calling thread's encodes takes the plain loop.  2: two passes wherever one would do.  Results are identical by
 * construction, which is what the tests check. */
void sela_hip_debug_force_plain_fir(int enable);
/* Debug hook: the encoder hands the sequential mean of every block beyond the first `self_blocks` of a launch
 * to "mean worker" workgroups (normally self_blocks = what the device holds at once, so small batches never use
 * workers).  Setting a small value makes small test batches take the worker path; -1 restores the default.
 * Results are identical by construction, which is what the tests check. */
void sela_hip_debug_mean_workers(int self_blocks);
/* Debug hook: which kernel analyses the blocks of the calling thread's *_device encodes: 0 = k_encode_blocks (one block per
 * wave), 8 / 16 = k_encode_teams with teams of that many lanes (eight / four blocks side by side in a wave), -1 restores
 * the choice by launch size.  Results are identical by construction, which is what the tests check. */
void sela_hip_debug_encode_teams(int lanes);
/* Which of them a *_device encode of n_frames frames of `channels` channels by the calling thread runs right now (the hook
 * above included): 0 = k_encode_blocks, 8 / 16 = k_encode_teams<.., 8 / 16>.  (bench.py names the kernel it reports on.) */
int sela_hip_debug_encode_kernel(uint32_t n_frames, uint32_t channels);
/* Debug hook (measurements only): while set, the calling thread's *_device encodes take the ONE-LAUNCH form of the host
 * pipeline (k_encode_blocks<0, true>: the blocks count themselves into groups of frames, a group's last block places and
 * writes the group's bytes -- no plan / assemble kernels) on device pointers.  Same bytes; bench.py --encode-fused times it
 * beside the default. */
void sela_hip_debug_encode_fused(int enable);
/* Debug hooks (measurements and tests; process-wide): the wave priorities (s_setprio 0..3) of the encode kernels' waves in the four
 * quarters of their work, one byte each from the low end (0x00010203: falling from 3 to 0; 0: none).  By default the library
 * decides per device-pointer launch: the falling schedule when no other stream has library work pending on the device, none
 * otherwise (sela_capi.hip, "does a device-pointer launch have the device to itself?"); a decode launch that is alone raises
 * the subframes of orders above 60 through their synthesis likewise (any non-zero forced schedule switches that on, 0 off).  sela_hip_debug_priorities() fixes the
 * schedule for every launch, sela_hip_debug_priorities_adaptive() goes back to the default; sela_hip_debug_launches_alone()
 * counts the launches that were given the falling schedule by that default. */
void sela_hip_debug_priorities(uint32_t team_quarters);
void sela_hip_debug_priorities_adaptive(void);
int sela_hip_debug_launches_alone(void);
/* Debug hook (process-wide): 1 = write the slots of BOTH candidates for an exactly-stereo frame's second channel (the channel
 * itself and the difference signal), as rounds 1-3 did; 0 (default) = the candidate that knows it has lost does not write
 * its slot (sela_encode_tail.inc).  Same bytes either way, which is what the tests check; the difference is HBM traffic. */
void sela_hip_debug_keep_both_candidates(int on);
/* Debug hook: bound of an encode block's wait for its frame from the staging kernel, in naps of 2048 cycles; 0: every
 * block gives up without looking ("the stagers never showed up"), which flags the launch and sends the feed through
 * the copy-engine path again; -1 restores the default (~0.5 s). */
void sela_hip_debug_stage_wait(int naps);
/* Debug hook: how many encode feeds of the calling thread were issued again through the copy-engine path so far. */
int sela_hip_debug_reissued_feeds(void);
/* Debug hook: the decoder runs a subframe's recurrence in one of two forms (same bits): the >> 3 of the prediction on the
 * scalar unit (for workgroups that share their SIMDs with many others) or on the vector unit (for those that run nearly
 * alone: small launches, the last workgroups of a launch).  0 / 1 force the first / second on every frame of the calling
 * thread's decodes, -1 restores the choice by launch size. */
void sela_hip_debug_decode_recurrence(int form);
/* Debug hook: how many per-thread contexts (streams, events, staging buffers) this process has CREATED so far -- threads
 * that take over a parked one (sela_hip.h, sela_hip_thread_release) do not count. */
int sela_hip_debug_contexts_created(void);

/* Debug hook (tests; process-wide): while on, a device-pointer encode with a d_trace pointer runs the PRODUCT kernels plus a few
 * instructions (their kMode 3 instantiations, not the trace builds) and leaves, instead of traces, two 64-bit words per block at
 * d_trace -- uint64 [n_frames * signals][2]: a position-keyed hash of the block's normalised autocorrelation ac[0..100] and
 * one of its reflection coefficients k[0..99], as bit patterns (sela_encode.hip, hash_term; tests/test_gpu_round5.py folds
 * the oracle's trace the same way).  The FP64 contract checked on the kernels that are timed. */
void sela_hip_debug_encode_hashes(int on);
/* Debug hook (tests): which form of the residue filter (sela_encode_tail.inc: one pass of FP64 taps / two passes / the plain
 * 64-bit loop) the blocks of the LAST device-pointer encode that used this workspace took -- counts_out[0..2], over all
 * n_frames * signals blocks (the losing stereo candidate included); forms_out (or NULL): the form of every block, 0 / 1 / 2,
 * [n_frames * signals].  The forms are chosen per block from its predictor and its
 * loudest sample; the product kernels leave the choice in two spare bits of their per-block records, which this reads back.
 * Synchronises the device.  Returns SELA_HIP_OK or an error code. */
int sela_hip_debug_block_forms(const void* d_workspace, uint32_t n_frames, uint32_t channels, uint32_t* counts_out, uint8_t* forms_out);
/* Debug hook (tests, bench.py --encode-split; process-wide): an encode launch on device pointers that the library gives to teams of
 * 16 is cut in two -- the halves on two streams, the first half's plan + assemble under the second half's tail (sela_capi.hip,
 * Splitter).  Built and measured in round 6: slower than the whole launch, so the library never does it by itself.  0: never
 * (the product); 1 .. 999: every such launch, with that share (per mille) of its frames in the first half.
 * sela_hip_debug_launches_split(): how many launches were. */
void sela_hip_debug_encode_split(int mode);
int sela_hip_debug_launches_split(void);
/* Debug hook (tests; process-wide): the any-length decoder (sela_hip_decode_i32 -- frame::FrameDecoder behind it -- and
 * sela_hip_decode on streams that are not 2048 samples per frame) offers its subframes to k_decode_subframes32 first (the fast
 * decoder's lane-parallel parse and tuned synthesis with 32-bit samples: one piece for subframes of at most 2048 samples that
 * fit the parser's plan, segments for everything else) and decodes again on the serial kernel k_generic_decode when that kernel left
 * anything alone.  -1 / 1: the product; 0: never offer (the serial kernel alone); 2: offer, but every subframe by segments.
 * sela_hip_debug_standard_chunks(): how many chunks of frames the fast kernel has decoded, alone, so far;
 * sela_hip_debug_segment_subframes(): how many subframes of those chunks it parsed by segments. */
void sela_hip_debug_standard_first(int mode);
int sela_hip_debug_standard_chunks(void);
long long sela_hip_debug_segment_subframes(void);
/* Debug hook (tests; process-wide): the any-length encoder's residue filter runs in FP64 wherever that is exact for the block
 * (2^34 + sum |a[j]| x max |sample| < 2^53) and on 64-bit wrap-around taps otherwise; on != 0 sends every block down the
 * wrap-around taps. */
void sela_hip_debug_generic_wrap_taps(int on);

#ifdef __cplusplus
}
#endif
#endif /* SELA_HIP_DEBUG_H_ */
