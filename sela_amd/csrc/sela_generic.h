// sela_generic.h -- what sela_generic.hip (the any-length / 32-bit route) shares with the C ABI (sela_capi.hip).
#ifndef SELA_GENERIC_H_
#define SELA_GENERIC_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sela {

struct GenericMeta { // one per (frame, signal), written by k_generic_analyse
    uint32_t order, coef_k, coef_words, res_k, res_words, flags;
    uint32_t form; // which form of the residue filter the block took: 0 FP64 taps (exact by its bound), 1 the 64-bit wrap-around taps
};
void set_generic_wrap_taps(int on); // tests: every block on the wrap-around taps
struct GenericSubInfo { // one per (frame, subframe position), written by k_generic_decode
    uint8_t channel, type, parent, ok;
    uint32_t n;
};

size_t generic_encode_workspace_bytes(uint32_t n_frames, uint32_t channels, uint32_t n);
hipError_t launch_generic_analyse(const void* d_input, bool in16, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n, int32_t* d_sig,
    int32_t* d_res, int32_t* d_q, GenericMeta* d_meta, hipStream_t stream);
hipError_t launch_generic_plan(const GenericMeta* d_meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint64_t base_bytes, uint64_t* d_frame_offsets,
    uint64_t* d_word_base, uint32_t* d_chosen, uint32_t* d_status, uint64_t* d_total_words, hipStream_t stream);
hipError_t launch_generic_emit(const GenericMeta* d_meta, uint32_t n_frames, uint32_t channels, uint32_t n_sig, uint32_t n, const int32_t* d_res, const int32_t* d_q,
    const uint32_t* d_chosen, const uint64_t* d_word_base, uint32_t* d_words /* zeroed */, uint64_t words_cap /* a subframe whose words would reach beyond is left out */,
    const uint64_t* d_frame_offsets, uint64_t base_bytes, uint8_t* d_frames, uint64_t frames_cap, hipStream_t stream);
// fast_first: the subframes go to k_decode_subframes32 (sela_decode32.hip: the fast decoder's lane-parallel parse and tuned
// synthesis, any length) instead of k_generic_decode (the serial walk); d_status[2] then counts the subframes that kernel left
// alone -- not zero: run the launch again without fast_first -- and d_status[3] those it parsed by segments (standard_path:
// 2048-sample subframes that fit the parser's plan take the frame kernel's own one-piece parse; off only in tests).
hipError_t launch_generic_decode(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels, uint32_t stride,
    int32_t* d_dec, GenericSubInfo* d_info, int32_t* d_all, uint32_t* d_counts, const uint64_t* d_sample_offsets, int16_t* d_pcm_out, uint32_t* d_status,
    bool fast_first, bool standard_path, hipStream_t stream);
hipError_t launch_decode_subframes32(const uint8_t* d_frames, const uint64_t* d_frame_offsets, uint64_t base_bytes, uint32_t n_frames, uint32_t channels,
    uint32_t stride, int32_t* d_dec, GenericSubInfo* d_info, uint32_t* d_status, bool standard_path, hipStream_t stream);
hipError_t launch_lpc_decode_any(const int32_t* d_order, const int32_t* d_q, const int32_t* d_residues, uint32_t n_blocks, uint32_t n, int32_t* d_samples,
    int64_t* d_coefs, uint32_t* d_status, hipStream_t stream);

} // namespace sela
#endif
