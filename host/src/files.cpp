// files.cpp -- WAV and .sela containers on flat buffers (see files.hpp).
//
// Behaviour kept from the reference (src/file/wav_file.cpp, src/file/sela_file.cpp): the accepted
// WAV subset (RIFF/WAVE, 16-bit PCM, chunks in any order up to 'data'), the error conditions and their
// messages, the dropped tail (only whole 2048-sample frames are coded), the canonical 44-byte header
// on output, the 15-byte .sela header and the silent stop at the first frame without a sync word.
// The files are read sequentially (headers first, then the payload straight into page-locked memory)
// instead of being slurped and copied.
#include "sela_host/files.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <iterator>

#include "sela_hip.h"
#include "sela_host/fileio.hpp"
#include "sela_host/frame.hpp"

namespace {

uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

size_t fileSize(std::ifstream& in)
{
    in.seekg(0, std::ios::end);
    const std::streamoff size = in.tellg();
    in.seekg(0, std::ios::beg);
    return size > 0 ? (size_t)size : 0;
}

bool readExact(std::ifstream& in, void* dst, size_t n)
{
    in.read(static_cast<char*>(dst), (std::streamsize)n);
    return (size_t)in.gcount() == n;
}

template <typename T>
void put(std::ofstream& out, T v)
{
    out.write(reinterpret_cast<const char*>(&v), sizeof v); // little-endian host, like the reference
}

// body(f) for every f in [0, n), in runs on the I/O pool's threads (the calling thread takes a share); the first
// data::Exception of a body is rethrown here.
template <typename Body>
void forFrames(size_t n, Body body)
{
    const size_t workers = std::min<size_t>(sela_host::IoPool::instance().threads() + 1, 8), run = 64;
    if (n < 4 * run || workers < 2) {
        for (size_t f = 0; f < n; f++)
            body(f);
        return;
    }
    std::atomic<size_t> next{ 0 };
    auto work = [&] {
        for (;;) {
            const size_t begin = next.fetch_add(run, std::memory_order_relaxed);
            if (begin >= n)
                return;
            for (size_t f = begin; f < std::min(n, begin + run); f++)
                body(f);
        }
    };
    sela_host::IoGroup group;
    for (size_t w = 1; w < workers; w++)
        group.run(work);
    std::string mine;
    try {
        work();
    } catch (const data::Exception& e) {
        mine = e.exceptionMessage.empty() ? "failed" : e.exceptionMessage;
        next.store(n, std::memory_order_relaxed);
    }
    group.wait(); // (throws what a pool task threw)
    if (!mine.empty())
        throw data::Exception(mine);
}

} // namespace

namespace file {

WavFile::WavFile(uint32_t rate, uint16_t bps, uint16_t channels, std::vector<data::WavFrame>&& frames)
    : sampleRate(rate), bitsPerSample(bps), numChannels(channels)
{
    wavFrames = std::move(frames);
    size_t total = 0;
    for (const data::WavFrame& f : wavFrames)
        total += (f.samples.empty() ? 0 : f.samples[0].size()) * f.samples.size();
    pcm.resize(total);
    size_t at = 0;
    for (const data::WavFrame& f : wavFrames) {
        const size_t n = f.samples.empty() ? 0 : f.samples[0].size();
        for (size_t i = 0; i < n; i++)
            for (size_t c = 0; c < f.samples.size(); c++)
                pcm[at++] = (int16_t)(uint16_t)f.samples[c][i];
    }
    syncChunk();
}

WavFile::WavFile(uint32_t rate, uint16_t channels, std::vector<int16_t>&& interleaved)
    : sampleRate(rate), bitsPerSample(16), numChannels(channels), pcm(interleaved)
{
    syncChunk();
}

WavFile::WavFile(uint32_t rate, uint16_t channels, sela_host::PinnedBuffer<int16_t>&& interleaved)
    : sampleRate(rate), bitsPerSample(16), numChannels(channels), pcm(std::move(interleaved))
{
    syncChunk();
}

WavFile& WavFile::operator=(const WavFile& o)
{
    if (this != &o) {
        samplesPerChannelPerFrame = o.samplesPerChannelPerFrame;
        sampleRate = o.sampleRate, bitsPerSample = o.bitsPerSample, numChannels = o.numChannels;
        pcm = o.pcm;
        wavChunk = o.wavChunk; // (with it the frames: wavFrames is a name for wavChunk.dataSubChunk.wavFrames)
        wavChunk.dataSubChunk.samples = pcm.data();
    }
    return *this;
}

WavFile& WavFile::operator=(WavFile&& o) noexcept
{
    if (this != &o) {
        samplesPerChannelPerFrame = o.samplesPerChannelPerFrame;
        sampleRate = o.sampleRate, bitsPerSample = o.bitsPerSample, numChannels = o.numChannels;
        pcm = std::move(o.pcm);
        wavChunk = std::move(o.wavChunk);
        wavChunk.dataSubChunk.samples = pcm.data();
        o.wavChunk.dataSubChunk.samples = nullptr, o.wavChunk.dataSubChunk.sampleCount = 0;
    }
    return *this;
}

// wavChunk as the reference's constructor and readFromFile leave it (src/file/wav_file.cpp:9-36, 80-162), minus the
// byte copy of the data chunk.
void WavFile::syncChunk()
{
    const uint32_t dataBytes = (uint32_t)(pcm.size() * 2);
    wavChunk.chunkId = "RIFF";
    wavChunk.format = "WAVE";
    wavChunk.chunkSize = 36 + dataBytes;
    data::WavFormatSubChunk& f = wavChunk.formatSubChunk;
    f.subChunkId = "fmt ";
    f.subChunkSize = 16;
    f.audioFormat = 1;
    f.numChannels = numChannels;
    f.sampleRate = sampleRate;
    f.bitsPerSample = bitsPerSample;
    f.blockAlign = (uint16_t)(numChannels * bitsPerSample / 8);
    f.byteRate = sampleRate * f.blockAlign;
    data::WavDataSubChunk& d = wavChunk.dataSubChunk;
    d.subChunkId = "data";
    d.subChunkSize = dataBytes;
    d.bitsPerSample = (uint8_t)bitsPerSample;
    d.channels = (uint8_t)numChannels;
    d.samples = pcm.data();
    d.sampleCount = pcm.size();
}

size_t WavFile::readHeader(std::ifstream& in)
{
    const size_t size = fileSize(in);
    if (size < 44)
        throw data::Exception("File is too small, probably not a wav file.");
    uint8_t riff[12];
    if (!readExact(in, riff, 12) || std::memcmp(riff, "RIFF", 4) != 0)
        throw data::Exception("chunkId is not RIFF, probably not a wav file.");
    if ((size_t)le32(riff + 4) > size)
        throw data::Exception("chunkSize exceeds file size, probably a corrupted file");
    if (std::memcmp(riff + 8, "WAVE", 4) != 0)
        throw data::Exception("format is not WAVE, probably not a wav file.");

    // Every chunk header of the file is walked, as the reference does (src/file/wav_file.cpp:80-162): a 'fmt ' chunk
    // counts wherever it stands, and when there are several 'data' chunks the LAST one is the audio.  Only the
    // headers are touched here; the payload is read later, piece by piece.
    bool haveFmt = false, haveData = false;
    size_t pos = 12, dataAt = 0, dataBytes = 0;
    while (pos + 8 <= size) {
        uint8_t head[8];
        in.seekg((std::streamoff)pos, std::ios::beg);
        if (!readExact(in, head, 8))
            break;
        size_t chunk = le32(head + 4);
        if (pos + 8 + chunk > size)
            chunk = size - pos - 8; // tolerate a short last chunk
        if (std::memcmp(head, "fmt ", 4) == 0 && chunk >= 16) {
            uint8_t body[16];
            if (!readExact(in, body, 16))
                break;
            haveFmt = true;
            numChannels = le16(body + 2);
            sampleRate = le32(body + 4);
            bitsPerSample = le16(body + 14);
            if (bitsPerSample != 16)
                throw data::Exception("Only 16bits per sample wav is supported.");
        } else if (std::memcmp(head, "data", 4) == 0) {
            if (!haveFmt)
                throw data::Exception("Probably corrupt wav, data subChunk present without fmt subChunk.");
            haveData = true;
            dataAt = pos + 8;
            dataBytes = chunk;
        }
        pos += 8 + chunk;
    }
    in.clear();
    if (!haveFmt)
        throw data::Exception("fmt subChunk is missing from file");
    if (!haveData)
        throw data::Exception("data subChunk is missing from file");
    in.seekg((std::streamoff)dataAt, std::ios::beg); // the stream stands at the first PCM byte
    syncChunk();                                       // (format fields now; the data view when the samples are in)
    wavChunk.chunkSize = le32(riff + 4);
    wavChunk.dataSubChunk.subChunkSize = (uint32_t)dataBytes;
    return dataBytes;
}

void WavFile::readFromFile(std::ifstream& in)
{
    const size_t bytes = readHeader(in);
    pcm.resize(bytes / 2);
    if (!pcm.empty() && !readExact(in, pcm.data(), pcm.size() * 2)) // already interleaved little-endian int16
        throw data::Exception("data subChunk is shorter than its header says");
    const uint32_t chunkSize = wavChunk.chunkSize;
    syncChunk();
    if (demuxOnRead)
        demuxSamples();
    wavChunk.chunkSize = chunkSize; // (what the file's header says)
}

bool WavFile::demuxOnRead = true;

void WavFile::demuxSamples()
{
    const size_t frames = frameCount(), n = samplesPerChannelPerFrame, channels = numChannels;
    const uint8_t bps = (uint8_t)bitsPerSample;
    wavFrames.assign(frames, data::WavFrame(bps, {}));
    // 2048 x channels int32 per frame, each frame on its own: spread over the I/O pool's threads (a 3-minute track is
    // 64 MB of them -- the objects cost more than the decode)
    forFrames(frames, [&](size_t f) {
        std::vector<std::vector<int32_t>> s(channels, std::vector<int32_t>(n));
        const int16_t* src = pcm.data() + f * n * channels;
        for (size_t i = 0; i < n; i++)
            for (size_t c = 0; c < channels; c++)
                s[c][i] = src[i * channels + c];
        wavFrames[f].samples = std::move(s);
    });
    syncChunk();
}

void WavFile::writeHeader(std::ofstream& out, uint32_t rate, uint16_t channels, uint16_t bps, uint32_t dataBytes)
{
    out.write("RIFF", 4);
    put<uint32_t>(out, 36 + dataBytes);
    out.write("WAVE", 4);
    out.write("fmt ", 4);
    put<uint32_t>(out, 16);
    put<int16_t>(out, 1); // PCM
    put<uint16_t>(out, channels);
    put<uint32_t>(out, rate);
    put<uint32_t>(out, rate * channels * bps / 8);
    put<uint16_t>(out, (uint16_t)(channels * bps / 8));
    put<uint16_t>(out, bps);
    out.write("data", 4);
    put<uint32_t>(out, dataBytes);
}

void WavFile::writeToFile(std::ofstream& out)
{
    const uint32_t dataBytes = (uint32_t)(pcm.size() * 2);
    writeHeader(out, sampleRate, numChannels, bitsPerSample, dataBytes);
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
    std::vector<uint8_t> bytes;
    for (const data::SelaFrame& f : selaFrames) {
        frame::appendFrame(f, bytes);
        frameOffsets.push_back(bytes.size());
    }
    frameBytes.assign(bytes.data(), bytes.size());
}

SelaFile::SelaFile(uint32_t rate, uint16_t bps, uint8_t channels, sela_host::PinnedBuffer<uint8_t>&& bytes, std::vector<uint64_t>&& offsets)
    : frameBytes(std::move(bytes)), frameOffsets(std::move(offsets))
{
    selaHeader.sampleRate = rate;
    selaHeader.bitsPerSample = bps;
    selaHeader.channels = channels;
    selaHeader.numFrames = frameOffsets.empty() ? 0 : (uint32_t)(frameOffsets.size() - 1);
}

size_t SelaFile::readHeader(std::ifstream& in)
{
    const size_t size = fileSize(in);
    uint8_t head[15];
    if (size < 15 || !readExact(in, head, 15))
        throw data::Exception("File is too small, probably not a sela file.");
    if (std::memcmp(head, "SeLa", 4) != 0)
        throw data::Exception("Magic number is incorrect, probably not a sela file.");
    selaHeader.sampleRate = le32(head + 4);
    selaHeader.bitsPerSample = le16(head + 8);
    selaHeader.channels = head[10];
    selaHeader.numFrames = le32(head + 11);
    return size - 15;
}

void SelaFile::readFromFile(std::ifstream& in)
{
    const size_t bytes = readHeader(in);
    frameBytes.resize(bytes);
    if (bytes && !readExact(in, frameBytes.data(), bytes))
        throw data::Exception("File is too small, probably not a sela file.");
    // index the frames; like the reference, stop silently at the first one without a sync word.  The frame
    // count of the header is not trusted for sizing: a frame has at least 4 + 12 bytes per channel.
    const size_t least = 4 + 12 * (size_t)selaHeader.channels;
    const uint32_t plausible = (uint32_t)std::min<size_t>(selaHeader.numFrames, bytes / least);
    frameOffsets.assign((size_t)plausible + 1, 0);
    const uint32_t found = sela_hip_index_frames(frameBytes.data(), frameBytes.size(), plausible, selaHeader.channels, frameOffsets.data());
    frameOffsets.resize((size_t)found + 1);
    frameBytes.resize((size_t)frameOffsets.back());
    materializeFrames();
}

void SelaFile::materializeFrames()
{
    const size_t n = frameCount();
    selaFrames.assign(n, data::SelaFrame((uint8_t)selaHeader.bitsPerSample));
    forFrames(n, [&](size_t f) {
        frame::parseFrame(frameBytes.data() + frameOffsets[f], (size_t)(frameOffsets[f + 1] - frameOffsets[f]), selaHeader.channels,
            (uint8_t)selaHeader.bitsPerSample, selaFrames[f]);
    });
}

void SelaFile::writeHeader(std::ofstream& out) const
{
    out.write(reinterpret_cast<const char*>(selaHeader.magicNumber), 4);
    put<uint32_t>(out, selaHeader.sampleRate);
    put<uint16_t>(out, selaHeader.bitsPerSample);
    put<uint8_t>(out, selaHeader.channels);
    put<uint32_t>(out, selaHeader.numFrames);
}

void SelaFile::writeToFile(std::ofstream& out)
{
    writeHeader(out);
    out.write(reinterpret_cast<const char*>(frameBytes.data()), (std::streamsize)frameBytes.size()); // one write
}

} // namespace file
