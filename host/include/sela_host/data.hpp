// data.hpp -- value types that cross the frame-codec boundary of the C++ host.
//
// Same namespaces, class names and public members as the reference's src/include/data/*.hpp
// (wav_frame.hpp:8-16, sela_frame.hpp:7-18, sela_sub_frame.hpp:7-45, rice_encoded_data.hpp:8-21,
// sela_header.hpp:7-14, exception.hpp:7-14), so code written against the reference compiles against
// this host.  Everything is a plain value type; on the fast path (sela::Encoder / sela::Decoder) these
// objects are only materialised for callers that ask for them -- the bytes travel as flat buffers.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace data {

// Thrown (by value, not derived from std::exception -- like the reference) by the file parsers and by
// the codec classes when the GPU path reports an error.
class Exception {
public:
    const std::string exceptionMessage;
    explicit Exception(std::string message) noexcept : exceptionMessage(std::move(message)) {}
};

// One block of de-interleaved PCM: samples[channel][i].
class WavFrame {
public:
    uint8_t bitsPerSample;
    std::vector<std::vector<int32_t>> samples;
    WavFrame(uint8_t bps, std::vector<std::vector<int32_t>> s) : bitsPerSample(bps), samples(std::move(s)) {}
};

// A Golomb-Rice coded integer sequence: parameter, value count, LSB-first packed words.
class RiceEncodedData {
public:
    uint32_t optimumRiceParam;
    uint32_t dataCount;
    std::vector<uint32_t> encodedData;
    RiceEncodedData(uint32_t param, uint32_t count, std::vector<uint32_t> words)
        : optimumRiceParam(param), dataCount(count), encodedData(std::move(words))
    {
    }
};

// What the stages hand each other (src/include/data/rice_decoded_data.hpp:8-17, lpc_decoded_data.hpp:8-19,
// lpc_encoded_data.hpp:8-21): names, members and constructor shapes as in the reference.
class RiceDecodedData {
public:
    std::vector<int32_t> decodedData;
    explicit RiceDecodedData(std::vector<int32_t> values) : decodedData(std::move(values)) {}
};

class LpcDecodedData {
public:
    uint8_t bitsPerSample;
    std::vector<int32_t> samples;
    LpcDecodedData(uint8_t bitsPerSample, std::vector<int32_t> samples) : bitsPerSample(bitsPerSample), samples(std::move(samples)) {}
};

class LpcEncodedData {
public:
    uint8_t optimalLpcOrder;
    uint8_t bitsPerSample;
    std::vector<int32_t> quantizedReflectionCoefficients;
    std::vector<int32_t> residues;
    LpcEncodedData(uint8_t optimalLpcOrder, uint8_t bitsPerSample, std::vector<int32_t> quantizedReflectionCoefficients, std::vector<int32_t> residues)
        : optimalLpcOrder(optimalLpcOrder), bitsPerSample(bitsPerSample), quantizedReflectionCoefficients(std::move(quantizedReflectionCoefficients)),
          residues(std::move(residues))
    {
    }
};

// One channel of a frame as stored on disk (field widths are the on-disk widths).
class SelaSubFrame {
public:
    uint8_t channel;
    uint8_t subFrameType;        // 0 independent, 1 difference against parentChannelNumber
    uint8_t parentChannelNumber;
    uint8_t reflectionCoefficientRiceParam;
    uint16_t reflectionCoefficientRequiredInts;
    uint8_t optimumLpcOrder;
    std::vector<uint32_t> encodedReflectionCoefficients;
    uint8_t residueRiceParam;
    uint16_t residueRequiredInts;
    uint16_t samplesPerChannel;
    std::vector<uint32_t> encodedResidues;

    SelaSubFrame(uint8_t ch, uint8_t type, uint8_t parent, const RiceEncodedData& refl, const RiceEncodedData& resid)
        : channel(ch), subFrameType(type), parentChannelNumber(parent),
          reflectionCoefficientRiceParam((uint8_t)refl.optimumRiceParam),
          reflectionCoefficientRequiredInts((uint16_t)refl.encodedData.size()), optimumLpcOrder((uint8_t)refl.dataCount),
          encodedReflectionCoefficients(refl.encodedData), residueRiceParam((uint8_t)resid.optimumRiceParam),
          residueRequiredInts((uint16_t)resid.encodedData.size()), samplesPerChannel((uint16_t)resid.dataCount),
          encodedResidues(resid.encodedData)
    {
    }
};

class SelaFrame {
public:
    int32_t syncWord = (int32_t)0xAA55FF00;
    std::vector<SelaSubFrame> subFrames;
    uint8_t bitsPerSample; // not written to the stream
    explicit SelaFrame(uint8_t bps) : bitsPerSample(bps) {}
};

// The RIFF view of a WAV file (reference: src/include/data/wav_chunk.hpp:7-16, wav_sub_chunk.hpp:9-33), kept so that
// code written against file::WavFile::wavChunk compiles: the same classes and member names.  What differs is where
// the audio lives: the data chunk's bytes are NOT copied into dataSubChunk.subChunkData (that copy is what the flat
// design exists to avoid) -- they are file::WavFile::pcm, and dataSubChunk.samples / sampleCount point at them.
class WavSubChunk {
public:
    std::string subChunkId;
    uint32_t subChunkSize = 0;
    std::vector<int8_t> subChunkData; // (other chunks: not kept; 'fmt ': its 16 bytes; 'data': empty, see samples)
};

class WavFormatSubChunk : public WavSubChunk {
public:
    int16_t audioFormat = 1;
    uint16_t numChannels = 0;
    uint32_t sampleRate = 0;
    uint32_t byteRate = 0;
    uint16_t blockAlign = 0;
    uint16_t bitsPerSample = 16;
};

class WavDataSubChunk : public WavSubChunk {
public:
    uint8_t bitsPerSample = 16;
    uint8_t channels = 0;
    std::vector<WavFrame> wavFrames;   // filled by file::WavFile::demuxSamples()
    const int16_t* samples = nullptr;  // the interleaved samples where they lie (file::WavFile::pcm)
    size_t sampleCount = 0;
};

class WavChunk {
public:
    std::string chunkId = "RIFF";
    uint32_t chunkSize = 0;
    std::string format = "WAVE";
    WavFormatSubChunk formatSubChunk;
    WavDataSubChunk dataSubChunk;
    std::vector<WavSubChunk> wavSubChunks;
};

// One frame's worth of interleaved int16 samples on its way to the audio device (reference:
// src/include/data/audio_packet.hpp:6-12).  Here `audio` points INTO the decoded samples (file::WavFile::pcm or the
// player's page-locked buffer): nothing is allocated per packet and nothing is to be freed.
class AudioPacket {
public:
    char* audio;
    const size_t bufferSize;
    AudioPacket(char* audio, const size_t bufferSize) : audio(audio), bufferSize(bufferSize) {}
};

class SelaHeader {
public:
    uint8_t magicNumber[4] = { 'S', 'e', 'L', 'a' };
    uint32_t sampleRate = 0;
    uint16_t bitsPerSample = 0;
    uint8_t channels = 0;
    uint32_t numFrames = 0;
};

} // namespace data
