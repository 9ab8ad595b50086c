// frame.cpp -- frame-level host glue: object <-> byte-stream conversion and the single-frame
// FrameEncoder / FrameDecoder entry points on top of libsela_hip.so.
#include "sela_host/frame.hpp"

#include <cstring>
#include <string>
#include <vector>

#include "sela_hip.h"

namespace {

constexpr uint32_t kSyncWord = 0xAA55FF00u;
constexpr size_t kBlock = SELA_HIP_SAMPLES_PER_FRAME;

uint16_t load16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t load32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void store16(std::vector<uint8_t>& out, uint16_t v)
{
    out.push_back((uint8_t)v);
    out.push_back((uint8_t)(v >> 8));
}
void store32(std::vector<uint8_t>& out, uint32_t v)
{
    store16(out, (uint16_t)v);
    store16(out, (uint16_t)(v >> 16));
}

} // namespace

namespace frame {

size_t parseFrame(const uint8_t* bytes, size_t available, uint8_t channels, uint8_t bitsPerSample, data::SelaFrame& out)
{
    if (available < 4 || load32(bytes) != kSyncWord)
        throw data::Exception("frame does not start with the sync word");
    size_t pos = 4;
    out = data::SelaFrame(bitsPerSample);
    out.subFrames.reserve(channels);
    for (unsigned c = 0; c < channels; c++) {
        if (pos + 7 > available)
            throw data::Exception("truncated subframe header");
        const uint8_t ch = bytes[pos], type = bytes[pos + 1], parent = bytes[pos + 2], coefK = bytes[pos + 3];
        const uint16_t coefWords = load16(bytes + pos + 4);
        const uint8_t order = bytes[pos + 6];
        pos += 7;
        if (pos + 4 * (size_t)coefWords + 5 > available)
            throw data::Exception("truncated reflection coefficient words");
        std::vector<uint32_t> cw(coefWords);
        for (auto& w : cw) {
            w = load32(bytes + pos);
            pos += 4;
        }
        const uint8_t resK = bytes[pos];
        const uint16_t resWords = load16(bytes + pos + 1), n = load16(bytes + pos + 3);
        pos += 5;
        if (pos + 4 * (size_t)resWords > available)
            throw data::Exception("truncated residue words");
        std::vector<uint32_t> rw(resWords);
        for (auto& w : rw) {
            w = load32(bytes + pos);
            pos += 4;
        }
        out.subFrames.emplace_back(ch, type, parent, data::RiceEncodedData(coefK, order, std::move(cw)),
            data::RiceEncodedData(resK, n, std::move(rw)));
    }
    return pos;
}

void appendFrame(const data::SelaFrame& f, std::vector<uint8_t>& out)
{
    store32(out, (uint32_t)f.syncWord);
    for (const data::SelaSubFrame& s : f.subFrames) {
        out.push_back(s.channel);
        out.push_back(s.subFrameType);
        out.push_back(s.parentChannelNumber);
        out.push_back(s.reflectionCoefficientRiceParam);
        store16(out, s.reflectionCoefficientRequiredInts);
        out.push_back(s.optimumLpcOrder);
        for (uint32_t w : s.encodedReflectionCoefficients)
            store32(out, w);
        out.push_back(s.residueRiceParam);
        store16(out, s.residueRequiredInts);
        store16(out, s.samplesPerChannel);
        for (uint32_t w : s.encodedResidues)
            store32(out, w);
    }
}

// The reference calls these classes from hardware_concurrency() threads at once, one frame per call
// (src/sela/encoder.cpp:58-73, src/sela/decoder.cpp:58-73).  One frame is a poor launch for a GPU; concurrent small calls
// are coalesced into one device job per trip inside libsela_hip.so (sela_hip_encode / sela_hip_decode, sela_capi.hip), so
// nothing of that needs to be done here.
data::SelaFrame FrameEncoder::process()
{
    const size_t channels = wavFrame.samples.size();
    if (channels == 0 || channels > 255)
        throw data::Exception("FrameEncoder: a frame needs 1..255 channels");
    std::vector<int16_t> pcm(kBlock * channels);
    for (size_t c = 0; c < channels; c++) {
        if (wavFrame.samples[c].size() != kBlock)
            throw data::Exception("FrameEncoder: the MI355X path codes whole 2048-sample frames only");
        for (size_t i = 0; i < kBlock; i++) {
            const int32_t v = wavFrame.samples[c][i];
            if (v < INT16_MIN || v > INT16_MAX)
                throw data::Exception("FrameEncoder: sample outside the 16-bit range");
            pcm[i * channels + c] = (int16_t)v;
        }
    }
    std::vector<uint8_t> bytes(sela_hip_encode_bound_bytes(1, (uint32_t)channels));
    uint64_t offsets[2] = { 0, 0 };
    if (sela_hip_encode(pcm.data(), 1, (uint32_t)channels, (uint32_t)kBlock, bytes.data(), bytes.size(), offsets) != SELA_HIP_OK)
        throw data::Exception(std::string("FrameEncoder: ") + sela_hip_last_error());
    data::SelaFrame frame(wavFrame.bitsPerSample);
    parseFrame(bytes.data(), (size_t)offsets[1], (uint8_t)channels, wavFrame.bitsPerSample, frame);
    return frame;
}

data::WavFrame FrameDecoder::process()
{
    const size_t channels = selaFrame.subFrames.size();
    if (channels == 0 || channels > 255)
        throw data::Exception("FrameDecoder: a frame needs 1..255 subframes");
    std::vector<uint8_t> bytes;
    appendFrame(selaFrame, bytes);
    const uint64_t offsets[2] = { 0, bytes.size() };
    std::vector<int16_t> pcm(kBlock * channels);
    if (sela_hip_decode(bytes.data(), offsets, 1, (uint32_t)channels, pcm.data()) != SELA_HIP_OK)
        throw data::Exception(std::string("FrameDecoder: ") + sela_hip_last_error());
    std::vector<std::vector<int32_t>> samples(channels, std::vector<int32_t>(kBlock));
    for (size_t i = 0; i < kBlock; i++)
        for (size_t c = 0; c < channels; c++)
            samples[c][i] = pcm[i * channels + c];
    return data::WavFrame(selaFrame.bitsPerSample, std::move(samples));
}

} // namespace frame
