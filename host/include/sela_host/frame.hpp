// frame.hpp -- frame::FrameEncoder / frame::FrameDecoder with the reference's signatures
// (src/include/frame.hpp:8-24), running on the MI355X through libsela_hip.so.  A call codes one frame, but calls made
// from several threads at once -- which is how the reference uses these classes, src/sela/encoder.cpp:58-73 -- are
// coalesced into one job per trip to the device (inside libsela_hip.so), so keeping the reference's thread loop costs a factor,
// not the GPU: the fast way in is still sela::Encoder / sela::Decoder (codec.hpp), which see the whole file.
//
// Also the flat <-> object conversions between the .sela frame byte stream (what the GPU path reads
// and writes) and data::SelaFrame objects.
#pragma once

#include <cstddef>

#include "sela_host/data.hpp"

namespace frame {

class FrameEncoder {
    const data::WavFrame& wavFrame;

public:
    explicit FrameEncoder(const data::WavFrame& frame) : wavFrame(frame) {}
    data::SelaFrame process(); // throws data::Exception if the GPU path fails
};

class FrameDecoder {
    const data::SelaFrame& selaFrame;

public:
    explicit FrameDecoder(const data::SelaFrame& frame) : selaFrame(frame) {}
    // The reference's answer: every channel as long as its subframe says, the samples as the 32-bit values the synthesis
    // produces (src/frame/frame_decoder.cpp:24-25,64-71) -- sela_hip_decode_i32: the fast kernels' parse and synthesis for the
    // 2048-sample subframes an encoder writes, the any-length kernel for the rest; calls from many threads are coalesced into
    // device batches like the encoder's.
    data::WavFrame process();
};

// On-disk bytes of one frame (layout of the reference's src/file/sela_file.cpp:115-135) -> object.
// Returns the number of bytes consumed; throws data::Exception on a truncated or malformed frame.
size_t parseFrame(const uint8_t* bytes, size_t available, uint8_t channels, uint8_t bitsPerSample, data::SelaFrame& out);
// Object -> on-disk bytes, appended to `out`.
void appendFrame(const data::SelaFrame& frame, std::vector<uint8_t>& out);

} // namespace frame
