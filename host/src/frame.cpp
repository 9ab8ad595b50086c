// frame.cpp -- frame-level host glue: object <-> byte-stream conversion and the single-frame
// FrameEncoder / FrameDecoder entry points on top of libsela_hip.so.
#include "sela_host/frame.hpp"

#include <algorithm>
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
    // The reference takes a data::WavFrame as it is: int32 samples, as many per channel as the vectors hold
    // (src/frame/frame_encoder.cpp:11-102 never looks at a length; only its WAV reader cuts 2048-sample frames).  The shape that
    // reader produces -- 2048 samples within 16 bits -- goes to the fast kernels (and is coalesced with other threads' frames);
    // anything else to the any-length kernels (sela_hip_encode_i32), channels of different lengths included: every channel is
    // analysed at its own length (:73-98), the second channel of an exactly-stereo frame against channel 0 - channel 1 over
    // its own length (:20-24; sela_hip_encode_ragged_i32).  What stays refused is what the reference itself cannot answer: an
    // exactly-stereo frame whose first channel is the shorter one (its difference signal indexes that channel out of bounds,
    // :22-24), more than 65535 samples (the subframe's u16 field) and a block not longer than its own predictor order (the
    // reference reads past its vector; SELA_HIP_ERANGE).
    const size_t n = wavFrame.samples[0].size();
    bool narrow = n == kBlock, ragged = false;
    for (size_t c = 0; c < channels; c++) {
        ragged = ragged || wavFrame.samples[c].size() != n;
        if (wavFrame.samples[c].size() == 0 || wavFrame.samples[c].size() > 65535)
            throw data::Exception("FrameEncoder: a channel holds 1..65535 samples (the subframe's count is 16 bits wide)");
    }
    if (ragged) {
        if (channels == 2 && wavFrame.samples[0].size() < wavFrame.samples[1].size())
            throw data::Exception("FrameEncoder: the first channel of a stereo frame must not be the shorter one (the reference's difference signal reads it past its end)");
        std::vector<int32_t> flat;
        std::vector<uint32_t> lengths(channels);
        size_t cap = 4;
        for (size_t c = 0; c < channels; c++) {
            lengths[c] = (uint32_t)wavFrame.samples[c].size();
            flat.insert(flat.end(), wavFrame.samples[c].begin(), wavFrame.samples[c].end());
            cap += sela_hip_encode_bound_bytes_n(1, 1, lengths[c]);
        }
        std::vector<uint8_t> bytes(cap);
        size_t used = 0;
        if (sela_hip_encode_ragged_i32(flat.data(), lengths.data(), (uint32_t)channels, bytes.data(), bytes.size(), &used) != SELA_HIP_OK)
            throw data::Exception(std::string("FrameEncoder: ") + sela_hip_last_error());
        data::SelaFrame frame(wavFrame.bitsPerSample);
        parseFrame(bytes.data(), used, (uint8_t)channels, wavFrame.bitsPerSample, frame);
        return frame;
    }
    for (size_t c = 0; c < channels; c++)
        for (size_t i = 0; narrow && i < n; i++)
            narrow = wavFrame.samples[c][i] >= INT16_MIN && wavFrame.samples[c][i] <= INT16_MAX;
    if (n == 0 || n > 65535)
        throw data::Exception("FrameEncoder: a channel holds 1..65535 samples (the subframe's count is 16 bits wide)");
    uint64_t offsets[2] = { 0, 0 };
    std::vector<uint8_t> bytes;
    int rc;
    if (narrow) {
        std::vector<int16_t> pcm(kBlock * channels);
        for (size_t c = 0; c < channels; c++)
            for (size_t i = 0; i < kBlock; i++)
                pcm[i * channels + c] = (int16_t)wavFrame.samples[c][i];
        bytes.resize(sela_hip_encode_bound_bytes(1, (uint32_t)channels));
        rc = sela_hip_encode(pcm.data(), 1, (uint32_t)channels, (uint32_t)kBlock, bytes.data(), bytes.size(), offsets);
    } else {
        std::vector<int32_t> planar(n * channels);
        for (size_t c = 0; c < channels; c++)
            std::memcpy(planar.data() + c * n, wavFrame.samples[c].data(), n * sizeof(int32_t));
        // room for six bytes per sample first (24-bit audio needs three): the format's own bound -- 65535 words per subframe, a
        // quarter of a megabyte per channel -- only for the frame that asks for it
        const size_t bound = sela_hip_encode_bound_bytes_n(1, (uint32_t)channels, (uint32_t)n);
        bytes.resize(std::min(bound, 4 + channels * (12 + 128 + 6 * n)));
        rc = sela_hip_encode_i32(planar.data(), 1, (uint32_t)channels, (uint32_t)n, bytes.data(), bytes.size(), offsets);
        if (rc == SELA_HIP_ECAPACITY && bytes.size() < bound) {
            bytes.resize(bound);
            rc = sela_hip_encode_i32(planar.data(), 1, (uint32_t)channels, (uint32_t)n, bytes.data(), bytes.size(), offsets);
        }
    }
    if (rc != SELA_HIP_OK)
        throw data::Exception(std::string("FrameEncoder: ") + sela_hip_last_error());
    data::SelaFrame frame(wavFrame.bitsPerSample);
    parseFrame(bytes.data(), (size_t)offsets[1], (uint8_t)channels, wavFrame.bitsPerSample, frame);
    return frame;
}

// frame::FrameDecoder::process returns what the subframes hold: every channel as long as its subframe says and the samples as
// the 32-bit values the synthesis produces (src/frame/frame_decoder.cpp:24-25,48-49,64-71; only file::WavFile::writeToFile
// narrows to 16 bits).  That is sela_hip_decode_i32: a frame of 2048-sample subframes -- every frame an encoder writes -- runs
// the fast kernels' parse and synthesis with the samples kept in 32 bits (k_decode_subframes32), any other frame the
// any-length kernel; calls from many threads are coalesced into device batches like the encoder's.
data::WavFrame FrameDecoder::process()
{
    const size_t channels = selaFrame.subFrames.size();
    if (channels == 0 || channels > 255)
        throw data::Exception("FrameDecoder: a frame needs 1..255 subframes");
    std::vector<uint8_t> bytes;
    appendFrame(selaFrame, bytes);
    const uint64_t offsets[2] = { 0, bytes.size() };
    uint32_t stride = 1;
    for (const data::SelaSubFrame& s : selaFrame.subFrames)
        stride = std::max<uint32_t>(stride, s.samplesPerChannel);
    std::vector<int32_t> planar((size_t)stride * channels);
    std::vector<uint32_t> counts(channels, 0);
    if (sela_hip_decode_i32(bytes.data(), offsets, 1, (uint32_t)channels, planar.data(), stride, counts.data()) != SELA_HIP_OK)
        throw data::Exception(std::string("FrameDecoder: ") + sela_hip_last_error());
    std::vector<std::vector<int32_t>> samples(channels);
    for (size_t c = 0; c < channels; c++)
        samples[c].assign(planar.begin() + (ptrdiff_t)(c * stride), planar.begin() + (ptrdiff_t)(c * stride + counts[c]));
    return data::WavFrame(selaFrame.bitsPerSample, std::move(samples));
}

} // namespace frame
