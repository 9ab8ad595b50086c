// codec.cpp -- sela::Encoder / sela::Decoder: whole-file batches through libsela_hip.so.
#include "sela_host/codec.hpp"

#include "sela_hip.h"

namespace sela {

file::SelaFile Encoder::process()
{
    wavFile.readFromFile(ifStream);
    const uint32_t channels = wavFile.numChannels;
    const size_t frames = wavFile.frameCount(); // tail samples beyond the last whole frame are dropped
    if (channels == 0 || channels > 255)
        throw data::Exception("Encoder: unsupported channel count");
    std::vector<uint8_t> bytes(sela_hip_encode_bound_bytes((uint32_t)frames, channels));
    std::vector<uint64_t> offsets(frames + 1, 0);
    // the WAV data chunk is already the interleaved int16 layout the GPU path reads
    if (sela_hip_encode(wavFile.pcm.data(), (uint32_t)frames, channels, SELA_HIP_SAMPLES_PER_FRAME, bytes.data(), bytes.size(),
            offsets.data()) != SELA_HIP_OK)
        throw data::Exception(std::string("Encoder: ") + sela_hip_last_error());
    bytes.resize((size_t)offsets[frames]);
    file::SelaFile out(wavFile.sampleRate, wavFile.bitsPerSample, (uint8_t)channels, std::move(bytes), std::move(offsets));
    if (materializeFrames)
        out.materializeFrames();
    return out;
}

file::WavFile Decoder::process()
{
    selaFile.readFromFile(ifStream);
    const uint32_t channels = selaFile.selaHeader.channels;
    const size_t frames = selaFile.frameOffsets.size() - 1;
    if (channels == 0)
        throw data::Exception("Decoder: unsupported channel count");
    std::vector<int16_t> pcm(frames * SELA_HIP_SAMPLES_PER_FRAME * channels);
    if (sela_hip_decode(selaFile.frameBytes.data(), selaFile.frameOffsets.data(), (uint32_t)frames, channels, pcm.data()) != SELA_HIP_OK)
        throw data::Exception(std::string("Decoder: ") + sela_hip_last_error());
    file::WavFile out(selaFile.selaHeader.sampleRate, (uint16_t)channels, std::move(pcm));
    if (demuxFrames)
        out.demuxSamples();
    return out;
}

std::vector<file::SelaFile> encodeBatch(const std::vector<file::WavFile>& wavs)
{
    std::vector<file::SelaFile> out(wavs.size());
    std::vector<bool> done(wavs.size(), false);
    for (size_t first = 0; first < wavs.size(); first++) {
        if (done[first])
            continue;
        const uint32_t channels = wavs[first].numChannels;
        if (channels == 0 || channels > 255)
            throw data::Exception("encodeBatch: unsupported channel count");
        // every not yet encoded file with this channel count joins the batch: whole frames only,
        // tail samples beyond a file's last whole frame are dropped exactly as for a single file
        std::vector<size_t> members;
        std::vector<int16_t> pcm;
        size_t frames = 0;
        for (size_t i = first; i < wavs.size(); i++) {
            if (done[i] || wavs[i].numChannels != channels)
                continue;
            members.push_back(i);
            const size_t n = wavs[i].frameCount();
            pcm.insert(pcm.end(), wavs[i].pcm.begin(), wavs[i].pcm.begin() + (std::ptrdiff_t)(n * SELA_HIP_SAMPLES_PER_FRAME * channels));
            frames += n;
            done[i] = true;
        }
        std::vector<uint8_t> bytes(sela_hip_encode_bound_bytes((uint32_t)frames, channels));
        std::vector<uint64_t> offsets(frames + 1, 0);
        if (sela_hip_encode(pcm.data(), (uint32_t)frames, channels, SELA_HIP_SAMPLES_PER_FRAME, bytes.data(), bytes.size(), offsets.data())
            != SELA_HIP_OK)
            throw data::Exception(std::string("encodeBatch: ") + sela_hip_last_error());
        size_t f0 = 0;
        for (size_t i : members) { // cut the batch's frame stream back into files
            const size_t n = wavs[i].frameCount();
            std::vector<uint64_t> offs(n + 1);
            for (size_t f = 0; f <= n; f++)
                offs[f] = offsets[f0 + f] - offsets[f0];
            std::vector<uint8_t> part(bytes.begin() + (std::ptrdiff_t)offsets[f0], bytes.begin() + (std::ptrdiff_t)offsets[f0 + n]);
            out[i] = file::SelaFile(wavs[i].sampleRate, wavs[i].bitsPerSample, (uint8_t)channels, std::move(part), std::move(offs));
            if (Encoder::materializeFrames)
                out[i].materializeFrames();
            f0 += n;
        }
    }
    return out;
}

std::vector<file::WavFile> decodeBatch(const std::vector<file::SelaFile>& selas)
{
    std::vector<file::WavFile> out(selas.size());
    std::vector<bool> done(selas.size(), false);
    for (size_t first = 0; first < selas.size(); first++) {
        if (done[first])
            continue;
        const uint32_t channels = selas[first].selaHeader.channels;
        if (channels == 0)
            throw data::Exception("decodeBatch: unsupported channel count");
        std::vector<size_t> members;
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offsets(1, 0);
        for (size_t i = first; i < selas.size(); i++) {
            if (done[i] || selas[i].selaHeader.channels != channels)
                continue;
            members.push_back(i);
            const uint64_t base = bytes.size();
            bytes.insert(bytes.end(), selas[i].frameBytes.begin(), selas[i].frameBytes.end());
            for (size_t f = 1; f < selas[i].frameOffsets.size(); f++)
                offsets.push_back(base + selas[i].frameOffsets[f]);
            done[i] = true;
        }
        const size_t frames = offsets.size() - 1;
        std::vector<int16_t> pcm(frames * SELA_HIP_SAMPLES_PER_FRAME * channels);
        if (frames && sela_hip_decode(bytes.data(), offsets.data(), (uint32_t)frames, channels, pcm.data()) != SELA_HIP_OK)
            throw data::Exception(std::string("decodeBatch: ") + sela_hip_last_error());
        size_t f0 = 0;
        for (size_t i : members) {
            const size_t n = selas[i].frameOffsets.size() - 1;
            const size_t per_frame = (size_t)SELA_HIP_SAMPLES_PER_FRAME * channels;
            std::vector<int16_t> part(pcm.begin() + (std::ptrdiff_t)(f0 * per_frame), pcm.begin() + (std::ptrdiff_t)((f0 + n) * per_frame));
            out[i] = file::WavFile(selas[i].selaHeader.sampleRate, (uint16_t)channels, std::move(part));
            if (Decoder::demuxFrames)
                out[i].demuxSamples();
            f0 += n;
        }
    }
    return out;
}

bool Encoder::materializeFrames = true;
bool Decoder::demuxFrames = true;

} // namespace sela
