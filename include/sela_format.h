/*
 * sela_format.h -- format-defining constants of the SELA frame codec.
 *
 * These numbers ARE the bitstream contract; every implementation in this repo
 * (HIP kernels, CPU oracle, C++ host) includes this one header so they cannot drift.
 * Each constant cites the reference line that fixes it (paths relative to the
 * reference checkout, sahaRatul/sela v2.0.2).
 */
#ifndef SELA_FORMAT_H_
#define SELA_FORMAT_H_

#include <stdint.h>

#define SELA_MAX_LPC_ORDER 100          /* src/include/lpc.hpp:7  */
#define SELA_Q_SHIFT 35                 /* CORRECTION_FACTOR, src/include/lpc.hpp:8 */
#define SELA_SQRT2 1.4142135623730950488016887242096 /* src/include/lpc.hpp:9 (literal, rounds to 0x1.6a09e667f3bcdp+0) */
#define SELA_MAX_RICE_PARAM 20          /* k in [0,20), src/include/rice.hpp:7 */
#define SELA_BLOCK 2048                 /* samplesPerChannelPerFrame, src/include/file/wav_file.hpp:12 */
#define SELA_SAMPLE_SCALE 32767.0       /* quantizationFactor = INT16_MAX, src/include/lpc.hpp:93 */
#define SELA_ORDER_THRESHOLD 0.05       /* src/lpc/residue_generator.cpp:73 */
#define SELA_SYNC_WORD 0xAA55FF00u      /* src/include/data/sela_frame.hpp:9 */
#define SELA_FILE_HEADER_BYTES 15       /* 'SeLa' u32 rate u16 bps u8 ch u32 frames, src/file/sela_file.cpp:108-112 */
#define SELA_SUBFRAME_HEADER_BYTES 12   /* 3 + 4 + 5 bytes of fields, src/file/sela_file.cpp:121-133 */

/* Dequantisation tables, src/include/lpc.hpp:10-71 (verbatim data, see tools/gen_tables.py). */
#ifndef SELA_TABLE_QUAL
#define SELA_TABLE_QUAL static const
#endif
#include "sela_tables.inc"

/* On-disk size of one frame given the per-subframe word counts. */
#if defined(__HIPCC__)
#define SELA_HOST_DEVICE __host__ __device__
#else
#define SELA_HOST_DEVICE
#endif
SELA_HOST_DEVICE static inline uint32_t sela_frame_bytes(uint32_t channels, uint32_t total_words)
{
    return 4u + channels * SELA_SUBFRAME_HEADER_BYTES + 4u * total_words;
}

#endif /* SELA_FORMAT_H_ */
