// files.hpp -- file::WavFile and file::SelaFile (reference: src/include/file/wav_file.hpp:10-19,
// src/include/file/sela_file.hpp:10-18), re-designed around flat buffers:
//
//   * WavFile keeps the data chunk as the interleaved little-endian int16 array it already is -- that
//     array IS the input format of sela_hip_encode, so encoding is zero-copy on the host -- in page-locked
//     memory, read from the file piece by piece (readHeader + the caller's reads) so that the GPU can start
//     before the file has been read.  The reference's per-frame de-interleaved `wavFrames` are still
//     available (demuxSamples()).
//   * SelaFile keeps the frame byte stream exactly as it goes to disk (one write()) plus the frame
//     offsets; `selaFrames` objects are materialised for callers that want them.
//
// Deviations from the reference's public members are listed in INTEGRATION.md section 3.
#pragma once

#include <fstream>
#include <string>

#include "sela_host/buffer.hpp"
#include "sela_host/data.hpp"

namespace file {

class WavFile {
public:
    size_t samplesPerChannelPerFrame = 2048;
    uint32_t sampleRate = 0;
    uint16_t bitsPerSample = 16;
    uint16_t numChannels = 0;
    sela_host::PinnedBuffer<int16_t> pcm;   // interleaved, whole data chunk
    // The reference's view of the same file (src/include/file/wav_file.hpp:14): format fields, sizes, and the data
    // chunk as a pointer into `pcm` (no copy).  Refreshed by the readers, the constructors and syncChunk().
    data::WavChunk wavChunk;
    // filled by demuxSamples(); the same vector the reference keeps at wavChunk.dataSubChunk.wavFrames (one copy, two names)
    std::vector<data::WavFrame>& wavFrames = wavChunk.dataSubChunk.wavFrames;
    void syncChunk();

    WavFile() {}
    // (wavChunk points into pcm: copies and moves re-point it)
    WavFile(const WavFile& o) { *this = o; }
    WavFile(WavFile&& o) noexcept { *this = std::move(o); }
    WavFile& operator=(const WavFile& o);
    WavFile& operator=(WavFile&& o) noexcept;
    WavFile(uint32_t rate, uint16_t bps, uint16_t channels, std::vector<data::WavFrame>&& frames);
    WavFile(uint32_t rate, uint16_t channels, std::vector<int16_t>&& interleaved);
    WavFile(uint32_t rate, uint16_t channels, sela_host::PinnedBuffer<int16_t>&& interleaved);

    // Parse up to the data chunk and leave `in` at its first byte; returns the chunk's size in bytes
    // (clipped to the file).  Throws data::Exception with the reference's messages.
    size_t readHeader(std::ifstream& in);
    void readFromFile(std::ifstream& in);   // readHeader + the whole data chunk (+ demuxSamples(), like the reference's, while demuxOnRead)
    // The reference's readFromFile ends by de-interleaving the samples into wavFrames (src/file/wav_file.cpp:178); on by default
    // for that fidelity.  The codec classes never need the copies (the GPU reads `pcm` as it lies in the file): tools that read
    // a file only to code it switch this off, or use Encoder / encodeFile, which do not go through here.
    static bool demuxOnRead;
    void writeToFile(std::ofstream& out);   // canonical 44-byte header + data
    static void writeHeader(std::ofstream& out, uint32_t rate, uint16_t channels, uint16_t bps, uint32_t dataBytes);
    void demuxSamples();                    // pcm -> wavFrames (whole frames only, tail dropped)
    size_t frameCount() const { return numChannels ? pcm.size() / numChannels / samplesPerChannelPerFrame : 0; }
};

class SelaFile {
public:
    data::SelaHeader selaHeader;
    std::vector<data::SelaFrame> selaFrames; // filled by readFromFile() and materializeFrames()
    sela_host::PinnedBuffer<uint8_t> frameBytes; // the stream behind the 15-byte header
    std::vector<uint64_t> frameOffsets;       // [numFrames + 1] byte offsets into frameBytes

    SelaFile() {}
    SelaFile(uint32_t rate, uint16_t bps, uint8_t channels, std::vector<data::SelaFrame>&& frames);
    SelaFile(uint32_t rate, uint16_t bps, uint8_t channels, sela_host::PinnedBuffer<uint8_t>&& bytes, std::vector<uint64_t>&& offsets);

    // Read the 15-byte header; returns the bytes that follow it in the file.
    size_t readHeader(std::ifstream& in);
    void readFromFile(std::ifstream& in);
    void writeToFile(std::ofstream& out);
    void writeHeader(std::ofstream& out) const;
    void materializeFrames(); // frameBytes -> selaFrames
    size_t frameCount() const { return frameOffsets.empty() ? 0 : frameOffsets.size() - 1; }
};

} // namespace file
