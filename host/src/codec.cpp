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

bool Encoder::materializeFrames = true;
bool Decoder::demuxFrames = true;

} // namespace sela
