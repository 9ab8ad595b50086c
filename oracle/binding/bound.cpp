// oracle/binding/bound.cpp -- TEST INFRASTRUCTURE (like ref_shim.cpp): the binding of INTEGRATION.md section 2, compiled.
//
// One translation unit that defines sela::Encoder::{readFrames, processFrames, process} and the sela::Decoder
// equivalents AGAINST THE REFERENCE'S OWN HEADERS (-I/root/reference/src/include: include/sela/encoder.hpp:9-22,
// include/sela/decoder.hpp:9-22) in place of the reference's src/sela/encoder.cpp:40-100 and src/sela/decoder.cpp:41-100:
// the thread fan-out over frames becomes one call into libsela_hip.so.  `make -C oracle bound` links it with the
// reference's UNMODIFIED file classes (src/file/wav_file.cpp, src/file/sela_file.cpp, compiled where they lie) and
// bound_main.cpp into oracle/_ref/sela_ref_bound; src/lpc, src/rice and src/frame are NOT linked -- they have dropped out
// of the path.  tests/test_gpu_round4.py runs the binary's -e / -d on the WAV files of tests/golden/file_digests.json and
// compares whole-file SHA-256s with what the unmodified reference wrote.
//
// Nothing of the product (sela_amd/, host/, bench.py's timed region) refers to this file or to the binary.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "data/exception.hpp"
#include "sela/decoder.hpp"
#include "sela/encoder.hpp"

#include "sela_hip.h" // -I<repo>/include; link -lsela_hip

namespace {

const size_t kSamplesPerFrame = 2048;

uint32_t load32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t load16(const uint8_t* p) { return (uint16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8)); }
void store32(std::vector<uint8_t>& out, uint32_t v)
{
    for (int i = 0; i < 4; i++)
        out.push_back((uint8_t)(v >> (8 * i)));
}
void store16(std::vector<uint8_t>& out, uint16_t v)
{
    out.push_back((uint8_t)v);
    out.push_back((uint8_t)(v >> 8));
}

// the frame layout file::SelaFile reads and writes (src/file/sela_file.cpp:58-91, :115-135) -> the reference's objects
data::SelaFrame parseFrame(const uint8_t* bytes, size_t available, uint8_t channels, uint8_t bitsPerSample)
{
    if (available < 4 || load32(bytes) != 0xAA55FF00u)
        throw data::Exception(std::string("frame does not start with the sync word"));
    size_t pos = 4;
    data::SelaFrame frame(bitsPerSample);
    frame.subFrames.reserve(channels);
    for (unsigned c = 0; c < channels; c++) {
        if (pos + 7 > available)
            throw data::Exception(std::string("truncated subframe header"));
        const uint8_t channel = bytes[pos], type = bytes[pos + 1], parent = bytes[pos + 2], coefK = bytes[pos + 3];
        const uint16_t coefWords = load16(bytes + pos + 4);
        const uint8_t order = bytes[pos + 6];
        pos += 7;
        if (pos + 4 * (size_t)coefWords + 5 > available)
            throw data::Exception(std::string("truncated reflection coefficient words"));
        std::vector<uint32_t> cw(coefWords);
        for (size_t i = 0; i < cw.size(); i++, pos += 4)
            cw[i] = load32(bytes + pos);
        const uint8_t resK = bytes[pos];
        const uint16_t resWords = load16(bytes + pos + 1), n = load16(bytes + pos + 3);
        pos += 5;
        if (pos + 4 * (size_t)resWords > available)
            throw data::Exception(std::string("truncated residue words"));
        std::vector<uint32_t> rw(resWords);
        for (size_t i = 0; i < rw.size(); i++, pos += 4)
            rw[i] = load32(bytes + pos);
        const data::RiceEncodedData reflectionData(coefK, order, std::move(cw));
        const data::RiceEncodedData residueData(resK, n, std::move(rw));
        frame.subFrames.push_back(data::SelaSubFrame(channel, type, parent, reflectionData, residueData));
    }
    return frame;
}

void appendFrame(const data::SelaFrame& f, std::vector<uint8_t>& out)
{
    store32(out, (uint32_t)f.syncWord);
    for (size_t i = 0; i < f.subFrames.size(); i++) {
        const data::SelaSubFrame& s = f.subFrames[i];
        out.push_back(s.channel);
        out.push_back(s.subFrameType);
        out.push_back(s.parentChannelNumber);
        out.push_back(s.reflectionCoefficientRiceParam);
        store16(out, s.reflectionCoefficientRequiredInts);
        out.push_back(s.optimumLpcOrder);
        for (size_t k = 0; k < s.encodedReflectionCoefficients.size(); k++)
            store32(out, s.encodedReflectionCoefficients[k]);
        out.push_back(s.residueRiceParam);
        store16(out, s.residueRequiredInts);
        store16(out, s.samplesPerChannel);
        for (size_t k = 0; k < s.encodedResidues.size(); k++)
            store32(out, s.encodedResidues[k]);
    }
}

} // namespace

namespace sela {

void Encoder::readFrames()
{
    wavFile.readFromFile(ifStream);
}

// replaces the thread pool of src/sela/encoder.cpp:40-92
void Encoder::processFrames(std::vector<data::SelaFrame>& encodedSelaFrames)
{
    const data::WavFormatSubChunk& fmt = wavFile.wavChunk.formatSubChunk;
    const std::vector<int8_t>& raw = wavFile.wavChunk.dataSubChunk.subChunkData; // interleaved little-endian int16
    const uint32_t channels = fmt.numChannels;
    const uint32_t frames = (uint32_t)wavFile.wavChunk.dataSubChunk.wavFrames.size(); // (= raw.size() / 2 / channels / 2048, tail dropped)
    if (frames == 0)
        return;
    // (the vector's storage is only byte-aligned by type; the library wants int16 alignment: a copy keeps this honest)
    std::vector<int16_t> pcm((size_t)frames * kSamplesPerFrame * channels);
    std::memcpy(pcm.data(), raw.data(), pcm.size() * sizeof(int16_t));
    std::vector<uint8_t> bytes(sela_hip_encode_bound_bytes(frames, channels));
    std::vector<uint64_t> offsets(frames + 1);
    if (sela_hip_encode(pcm.data(), frames, channels, (uint32_t)kSamplesPerFrame, bytes.data(), bytes.size(), offsets.data()) != SELA_HIP_OK)
        throw data::Exception(std::string(sela_hip_last_error()));
    encodedSelaFrames.reserve(frames);
    for (uint32_t f = 0; f < frames; f++)
        encodedSelaFrames.push_back(parseFrame(bytes.data() + offsets[f], (size_t)(offsets[f + 1] - offsets[f]), (uint8_t)channels, (uint8_t)fmt.bitsPerSample));
}

file::SelaFile Encoder::process()
{
    std::vector<data::SelaFrame> selaFrames;
    readFrames();
    processFrames(selaFrames);
    return file::SelaFile(wavFile.wavChunk.formatSubChunk.sampleRate, wavFile.wavChunk.formatSubChunk.bitsPerSample,
        (uint8_t)wavFile.wavChunk.formatSubChunk.numChannels, std::move(selaFrames));
}

void Decoder::readFrames()
{
    selaFile.readFromFile(ifStream);
}

// replaces the thread pool of src/sela/decoder.cpp:41-92
void Decoder::processFrames(std::vector<data::WavFrame>& decodedWavFrames)
{
    const uint32_t channels = selaFile.selaHeader.channels;
    const uint32_t frames = (uint32_t)selaFile.selaFrames.size();
    if (frames == 0)
        return;
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offsets(frames + 1);
    for (uint32_t f = 0; f < frames; f++) {
        offsets[f] = bytes.size();
        appendFrame(selaFile.selaFrames[f], bytes);
    }
    offsets[frames] = bytes.size();
    // frames say their own length (2048 from the reference's encoder; anything in a hand-made stream, src/frame/frame_decoder.cpp:24-25)
    std::vector<uint64_t> at(frames + 1);
    (void)sela_hip_index_samples(bytes.data(), offsets.data(), frames, channels, at.data());
    std::vector<int16_t> pcm(std::max<size_t>((size_t)at[frames], (size_t)frames * kSamplesPerFrame) * channels + 1); // (sela_hip.h: the fast kernels are tried first)
    if (sela_hip_decode(bytes.data(), offsets.data(), frames, channels, pcm.data()) != SELA_HIP_OK)
        throw data::Exception(std::string(sela_hip_last_error()));
    decodedWavFrames.reserve(frames);
    for (uint32_t f = 0; f < frames; f++) {
        const size_t n = (size_t)(at[f + 1] - at[f]);
        std::vector<std::vector<int32_t> > samples(channels, std::vector<int32_t>(n));
        const int16_t* p = pcm.data() + (size_t)at[f] * channels;
        for (size_t i = 0; i < n; i++)
            for (uint32_t c = 0; c < channels; c++)
                samples[c][i] = p[i * channels + c];
        decodedWavFrames.push_back(data::WavFrame((uint8_t)selaFile.selaHeader.bitsPerSample, std::move(samples)));
    }
}

file::WavFile Decoder::process()
{
    std::vector<data::WavFrame> wavFrames;
    readFrames();
    processFrames(wavFrames);
    return file::WavFile(selaFile.selaHeader.sampleRate, selaFile.selaHeader.bitsPerSample, selaFile.selaHeader.channels, std::move(wavFrames));
}

} // namespace sela
