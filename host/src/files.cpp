// files.cpp -- WAV and .sela containers on flat buffers (see files.hpp).
//
// Behaviour kept from the reference (src/file/wav_file.cpp, src/file/sela_file.cpp): the accepted
// WAV subset (RIFF/WAVE, 16-bit PCM, any chunk order after 'fmt '), the error conditions and their
// messages, the dropped tail (only whole 2048-sample frames are coded), the canonical 44-byte header
// on output, the 15-byte .sela header and the silent stop at the first frame without a sync word.
#include "sela_host/files.hpp"

#include <cstring>
#include <iterator>

#include "sela_hip.h"
#include "sela_host/frame.hpp"

namespace {

uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

std::vector<uint8_t> slurp(std::ifstream& in)
{
    in.seekg(0, std::ios::end);
    const std::streamoff size = in.tellg();
    in.seekg(0, std::ios::beg);
    std::vector<uint8_t> bytes(size > 0 ? (size_t)size : 0);
    if (!bytes.empty())
        in.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)bytes.size());
    return bytes;
}

template <typename T>
void put(std::ofstream& out, T v)
{
    out.write(reinterpret_cast<const char*>(&v), sizeof v); // little-endian host, like the reference
}

} // namespace

namespace file {

WavFile::WavFile(uint32_t rate, uint16_t bps, uint16_t channels, std::vector<data::WavFrame>&& frames)
    : sampleRate(rate), bitsPerSample(bps), numChannels(channels), wavFrames(std::move(frames))
{
    for (const data::WavFrame& f : wavFrames) {
        const size_t n = f.samples.empty() ? 0 : f.samples[0].size();
        for (size_t i = 0; i < n; i++)
            for (size_t c = 0; c < f.samples.size(); c++)
                pcm.push_back((int16_t)(uint16_t)f.samples[c][i]);
    }
}

WavFile::WavFile(uint32_t rate, uint16_t channels, std::vector<int16_t>&& interleaved)
    : sampleRate(rate), bitsPerSample(16), numChannels(channels), pcm(std::move(interleaved))
{
}

void WavFile::readFromFile(std::ifstream& in)
{
    const std::vector<uint8_t> bytes = slurp(in);
    if (bytes.size() < 44)
        throw data::Exception("File is too small, probably not a wav file.");
    if (std::memcmp(bytes.data(), "RIFF", 4) != 0)
        throw data::Exception("chunkId is not RIFF, probably not a wav file.");
    if ((size_t)le32(bytes.data() + 4) > bytes.size())
        throw data::Exception("chunkSize exceeds file size, probably a corrupted file");
    if (std::memcmp(bytes.data() + 8, "WAVE", 4) != 0)
        throw data::Exception("format is not WAVE, probably not a wav file.");

    bool haveFmt = false, haveData = false;
    size_t pos = 12;
    while (pos + 8 <= bytes.size()) {
        const uint8_t* id = bytes.data() + pos;
        size_t size = le32(bytes.data() + pos + 4);
        const uint8_t* body = bytes.data() + pos + 8;
        if (pos + 8 + size > bytes.size())
            size = bytes.size() - pos - 8; // tolerate a short last chunk
        if (std::memcmp(id, "fmt ", 4) == 0 && size >= 16) {
            haveFmt = true;
            numChannels = le16(body + 2);
            sampleRate = le32(body + 4);
            bitsPerSample = le16(body + 14);
            if (bitsPerSample != 16)
                throw data::Exception("Only 16bits per sample wav is supported.");
        } else if (std::memcmp(id, "data", 4) == 0) {
            if (!haveFmt)
                throw data::Exception("Probably corrupt wav, data subChunk present without fmt subChunk.");
            haveData = true;
            pcm.resize(size / 2);
            std::memcpy(pcm.data(), body, pcm.size() * 2); // already interleaved little-endian int16
        }
        pos += 8 + size;
    }
    if (!haveFmt)
        throw data::Exception("fmt subChunk is missing from file");
    if (!haveData)
        throw data::Exception("data subChunk is missing from file");
}

void WavFile::demuxSamples()
{
    wavFrames.clear();
    const size_t frames = frameCount(), n = samplesPerChannelPerFrame;
    wavFrames.reserve(frames);
    for (size_t f = 0; f < frames; f++) {
        std::vector<std::vector<int32_t>> s(numChannels, std::vector<int32_t>(n));
        const int16_t* src = pcm.data() + f * n * numChannels;
        for (size_t i = 0; i < n; i++)
            for (size_t c = 0; c < numChannels; c++)
                s[c][i] = src[i * numChannels + c];
        wavFrames.emplace_back((uint8_t)bitsPerSample, std::move(s));
    }
}

void WavFile::writeToFile(std::ofstream& out)
{
    const uint32_t dataBytes = (uint32_t)(pcm.size() * 2);
    out.write("RIFF", 4);
    put<uint32_t>(out, 36 + dataBytes);
    out.write("WAVE", 4);
    out.write("fmt ", 4);
    put<uint32_t>(out, 16);
    put<int16_t>(out, 1); // PCM
    put<uint16_t>(out, numChannels);
    put<uint32_t>(out, sampleRate);
    put<uint32_t>(out, sampleRate * numChannels * bitsPerSample / 8);
    put<uint16_t>(out, (uint16_t)(numChannels * bitsPerSample / 8));
    put<uint16_t>(out, bitsPerSample);
    out.write("data", 4);
    put<uint32_t>(out, dataBytes);
    out.write(reinterpret_cast<const char*>(pcm.data()), dataBytes);
}

SelaFile::SelaFile(uint32_t rate, uint16_t bps, uint8_t channels, std::vector<data::SelaFrame>&& frames)
    : selaFrames(std::move(frames))
{
    selaHeader.sampleRate = rate;
    selaHeader.bitsPerSample = bps;
    selaHeader.channels = channels;
    selaHeader.numFrames = (uint32_t)selaFrames.size();
    frameOffsets.push_back(0);
    for (const data::SelaFrame& f : selaFrames) {
        frame::appendFrame(f, frameBytes);
        frameOffsets.push_back(frameBytes.size());
    }
}

SelaFile::SelaFile(uint32_t rate, uint16_t bps, uint8_t channels, std::vector<uint8_t>&& bytes, std::vector<uint64_t>&& offsets)
    : frameBytes(std::move(bytes)), frameOffsets(std::move(offsets))
{
    selaHeader.sampleRate = rate;
    selaHeader.bitsPerSample = bps;
    selaHeader.channels = channels;
    selaHeader.numFrames = frameOffsets.empty() ? 0 : (uint32_t)(frameOffsets.size() - 1);
}

void SelaFile::readFromFile(std::ifstream& in)
{
    std::vector<uint8_t> bytes = slurp(in);
    if (bytes.size() < 15)
        throw data::Exception("File is too small, probably not a sela file.");
    if (std::memcmp(bytes.data(), "SeLa", 4) != 0)
        throw data::Exception("Magic number is incorrect, probably not a sela file.");
    selaHeader.sampleRate = le32(bytes.data() + 4);
    selaHeader.bitsPerSample = le16(bytes.data() + 8);
    selaHeader.channels = bytes[10];
    selaHeader.numFrames = le32(bytes.data() + 11);
    frameBytes.assign(bytes.begin() + 15, bytes.end());
    // index the frames; like the reference, stop silently at the first one without a sync word
    frameOffsets.assign((size_t)selaHeader.numFrames + 1, 0);
    const uint32_t found = sela_hip_index_frames(frameBytes.data(), frameBytes.size(), selaHeader.numFrames, selaHeader.channels,
        frameOffsets.data());
    frameOffsets.resize((size_t)found + 1);
    frameBytes.resize((size_t)frameOffsets.back());
    materializeFrames();
}

void SelaFile::materializeFrames()
{
    selaFrames.clear();
    const size_t n = frameOffsets.empty() ? 0 : frameOffsets.size() - 1;
    selaFrames.reserve(n);
    for (size_t f = 0; f < n; f++) {
        data::SelaFrame frame((uint8_t)selaHeader.bitsPerSample);
        frame::parseFrame(frameBytes.data() + frameOffsets[f], (size_t)(frameOffsets[f + 1] - frameOffsets[f]), selaHeader.channels,
            (uint8_t)selaHeader.bitsPerSample, frame);
        selaFrames.push_back(std::move(frame));
    }
}

void SelaFile::writeToFile(std::ofstream& out)
{
    out.write(reinterpret_cast<const char*>(selaHeader.magicNumber), 4);
    put<uint32_t>(out, selaHeader.sampleRate);
    put<uint16_t>(out, selaHeader.bitsPerSample);
    put<uint8_t>(out, selaHeader.channels);
    put<uint32_t>(out, selaHeader.numFrames);
    out.write(reinterpret_cast<const char*>(frameBytes.data()), (std::streamsize)frameBytes.size()); // one write
}

} // namespace file
