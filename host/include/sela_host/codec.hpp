// codec.hpp -- sela::Encoder / sela::Decoder with the reference's signatures
// (src/include/sela/encoder.hpp:9-22, src/include/sela/decoder.hpp:9-22).
//
// Where the reference fans frames out to hardware_concurrency() threads (src/sela/encoder.cpp:40-92),
// these hand the whole file to the MI355X as one batch: sela_hip_encode / sela_hip_decode.
// There is no CPU fallback: if the GPU path is unavailable process() throws data::Exception.
#pragma once

#include <fstream>
#include <vector>

#include "sela_host/files.hpp"

namespace sela {

class Encoder {
    std::ifstream& ifStream;
    file::WavFile wavFile;

public:
    // Also build the data::SelaFrame objects in SelaFile::selaFrames (API fidelity).  Tools that only
    // write the file can switch this off: the byte stream is complete without them.
    static bool materializeFrames;
    explicit Encoder(std::ifstream& in) : ifStream(in) {}
    file::SelaFile process();
};

class Decoder {
    std::ifstream& ifStream;
    file::SelaFile selaFile;

public:
    static bool demuxFrames; // also fill WavFile::wavFrames (per-frame de-interleaved copies)
    explicit Decoder(std::ifstream& in) : ifStream(in) {}
    file::WavFile process();
};

// Many files, one GPU batch per channel count (BASELINE.json configs[3]: an album is a few thousand
// frames per track -- batching the tracks fills the device where one track would leave it in its
// launch tail).  Results are what Encoder / Decoder give file by file.
std::vector<file::SelaFile> encodeBatch(const std::vector<file::WavFile>& wavs);
std::vector<file::WavFile> decodeBatch(const std::vector<file::SelaFile>& selas);

} // namespace sela
