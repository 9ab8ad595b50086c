// codec.cpp -- sela::Encoder / sela::Decoder, file-to-file streaming, and the multi-GPU batch dispatcher,
// all on top of libsela_hip.so's host-pointer API.
#include "sela_host/codec.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>

#include "sela_hip.h"

namespace {

constexpr size_t kBlock = SELA_HIP_SAMPLES_PER_FRAME;
constexpr uint32_t kPieceFrames = 1024;        // file read granularity while encoding = one pipeline chunk
constexpr size_t kPieceBytes = (size_t)8 << 20; // ... and while decoding

[[noreturn]] void gpuFailure(const char* what)
{
    throw data::Exception(std::string(what) + ": " + sela_hip_last_error());
}

bool readExact(std::ifstream& in, void* dst, size_t n)
{
    in.read(static_cast<char*>(dst), (std::streamsize)n);
    return (size_t)in.gcount() == n;
}

// Output capacity tried first: audio does not grow under the codec (the certain bound,
// sela_hip_encode_bound_bytes, is 2.2x the PCM and would be pinned for nothing); SELA_HIP_ECAPACITY falls
// back to the bound.
size_t optimisticBytes(size_t frames, uint32_t channels)
{
    const size_t pcm = frames * kBlock * channels * 2;
    return std::min(sela_hip_encode_bound_bytes((uint32_t)frames, channels), pcm + pcm / 8 + 64 * frames + 4096);
}

// Encode `frames` frames that are (or, with `in`, are being read piece by piece) at pcm.  drain(bytes, n) is
// told whenever more of the output is final.
template <typename Drain>
void streamEncode(std::ifstream* in, int16_t* pcm, size_t frames, uint32_t channels, sela_host::PinnedBuffer<uint8_t>& bytes,
    std::vector<uint64_t>& offsets, Drain drain)
{
    const size_t frameSamples = kBlock * channels;
    offsets.assign(frames + 1, 0);
    size_t readFrames = in ? 0 : frames; // frames of pcm that are in memory
    for (int attempt = 0; attempt < 2; attempt++) {
        bytes.resize(attempt == 0 ? optimisticBytes(frames, channels) : sela_hip_encode_bound_bytes((uint32_t)frames, channels));
        sela_hip_job* job = nullptr;
        if (sela_hip_encode_begin(&job, channels, (uint32_t)frames, bytes.data(), bytes.size(), offsets.data()) != SELA_HIP_OK)
            gpuFailure("Encoder");
        int rc = SELA_HIP_OK;
        // (a feed is one kernel launch: samples that are in memory already go in one piece)
        const size_t piece = in ? kPieceFrames : std::max<size_t>(frames, 1);
        for (size_t f0 = 0; f0 < frames && rc == SELA_HIP_OK; f0 += piece) {
            const size_t nf = std::min<size_t>(piece, frames - f0);
            if (in && f0 + nf > readFrames) { // the device works on the earlier pieces while this one is read
                if (!readExact(*in, pcm + f0 * frameSamples, nf * frameSamples * 2)) {
                    (void)sela_hip_encode_end(job, nullptr, nullptr);
                    throw data::Exception("data subChunk is shorter than its header says");
                }
                readFrames = f0 + nf;
            }
            uint64_t done = 0;
            rc = sela_hip_encode_feed(job, pcm + f0 * frameSamples, (uint32_t)nf, nullptr, &done);
            if (rc == SELA_HIP_OK)
                drain(bytes.data(), (size_t)done);
        }
        const std::string feedError = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
        uint64_t total = 0;
        const int rcEnd = sela_hip_encode_end(job, nullptr, &total);
        if (rc == SELA_HIP_OK)
            rc = rcEnd;
        if (rc == SELA_HIP_OK) {
            drain(bytes.data(), (size_t)total);
            bytes.resize((size_t)total);
            return;
        }
        if (rc != SELA_HIP_ECAPACITY || attempt == 1)
            throw data::Exception("Encoder: " + (feedError.empty() ? std::string(sela_hip_last_error()) : feedError));
        // (what was drained so far stays valid: the bytes do not depend on the capacity)
    }
}

// Decode the frame stream that is being read from `in` (payload bytes behind the 15-byte header) into pcm;
// drain(samples, n) is told whenever more samples are final.  Fills sela.frameBytes / frameOffsets.
template <typename Drain>
void streamDecode(std::ifstream& in, file::SelaFile& sela, size_t payload, sela_host::PinnedBuffer<int16_t>& pcm, Drain drain)
{
    const uint32_t channels = sela.selaHeader.channels;
    if (channels == 0)
        throw data::Exception("Decoder: unsupported channel count");
    const size_t frameSamples = kBlock * channels;
    // the header's frame count is not trusted for sizing: a frame has at least 4 + 12 bytes per channel
    const size_t plausible = std::min<size_t>(sela.selaHeader.numFrames, payload / (4 + 12 * (size_t)channels));
    sela.frameBytes.resize(payload);
    sela.frameOffsets.assign(plausible + 1, 0);
    pcm.resize(plausible * frameSamples);
    sela_hip_job* job = nullptr;
    if (sela_hip_decode_begin(&job, channels, (uint32_t)plausible, pcm.data()) != SELA_HIP_OK)
        gpuFailure("Decoder");
    size_t have = 0, indexed = 0, fed = 0;
    int rc = SELA_HIP_OK;
    std::vector<uint64_t> local(kPieceFrames + 1);
    while (rc == SELA_HIP_OK && (have < payload || fed < indexed)) {
        if (have < payload) {
            const size_t n = std::min(kPieceBytes, payload - have);
            if (!readExact(in, sela.frameBytes.data() + have, n)) {
                (void)sela_hip_decode_end(job, nullptr);
                throw data::Exception("File is too small, probably not a sela file.");
            }
            have += n;
        }
        // index the frames that are complete in what has been read; like the reference, stop for good at the
        // first one without a sync word (which this cannot tell from "not yet read" until the file is in)
        for (;;) {
            const size_t want = std::min<size_t>(kPieceFrames, plausible - indexed);
            if (want == 0)
                break;
            const uint64_t base = sela.frameOffsets[indexed];
            const uint32_t found = sela_hip_index_frames(sela.frameBytes.data() + base, have - (size_t)base, (uint32_t)want, channels, local.data());
            for (uint32_t f = 1; f <= found; f++)
                sela.frameOffsets[indexed + f] = base + local[f];
            indexed += found;
            if (found < want)
                break;
        }
        const bool last = have == payload;
        if (indexed - fed >= kPieceFrames || (last && indexed > fed)) {
            uint32_t done = 0;
            rc = sela_hip_decode_feed(job, sela.frameBytes.data(), sela.frameOffsets.data() + fed, (uint32_t)(indexed - fed), &done);
            fed = indexed;
            if (rc == SELA_HIP_OK)
                drain(pcm.data(), (size_t)done * frameSamples);
        }
    }
    const std::string feedError = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
    uint32_t done = 0;
    const int rcEnd = sela_hip_decode_end(job, &done);
    if (rc == SELA_HIP_OK)
        rc = rcEnd;
    if (rc != SELA_HIP_OK)
        throw data::Exception("Decoder: " + (feedError.empty() ? std::string(sela_hip_last_error()) : feedError));
    sela.frameOffsets.resize(indexed + 1);
    sela.frameBytes.resize((size_t)sela.frameOffsets.back());
    pcm.resize(indexed * frameSamples);
    drain(pcm.data(), pcm.size());
}

// ---- multi-GPU dispatcher ---------------------------------------------------------------------------------------
std::mutex g_devicesMutex;
std::vector<int> g_devices; // empty = every visible device

struct Piece {
    size_t track, first, n; // frames [first, first + n) of `track`
};

// Contiguous balanced ranges of the tracks' flattened frame space, the first (total % workers) one longer,
// cut at track boundaries.
std::vector<std::vector<Piece>> partitionPieces(const std::vector<size_t>& trackFrames, size_t workers)
{
    size_t total = 0;
    for (size_t n : trackFrames)
        total += n;
    std::vector<std::vector<Piece>> out(workers);
    size_t track = 0, inTrack = 0;
    for (size_t w = 0; w < workers; w++) {
        size_t left = total / workers + (w < total % workers ? 1 : 0);
        while (left) {
            while (track < trackFrames.size() && inTrack == trackFrames[track])
                track++, inTrack = 0;
            const size_t n = std::min(left, trackFrames[track] - inTrack);
            out[w].push_back({ track, inTrack, n });
            inTrack += n;
            left -= n;
        }
    }
    return out;
}

// Run work(worker, pieces) on one host thread per device; the first exception is rethrown in the caller.
template <typename Work>
void runOnDevices(const std::vector<std::vector<Piece>>& pieces, const std::vector<int>& devs, Work work)
{
    std::vector<std::thread> pool;
    std::mutex errorMutex;
    std::string error;
    for (size_t w = 0; w < devs.size(); w++) {
        pool.emplace_back([&, w]() {
            try {
                if (sela_hip_init(devs[w]) != SELA_HIP_OK)
                    gpuFailure("device");
                work(w, pieces[w]);
            } catch (const data::Exception& e) {
                std::lock_guard<std::mutex> lock(errorMutex);
                if (error.empty())
                    error = e.exceptionMessage;
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lock(errorMutex);
                if (error.empty())
                    error = e.what();
            }
            sela_hip_thread_release(); // this thread's device buffers and streams
        });
    }
    for (std::thread& t : pool)
        t.join();
    if (!error.empty())
        throw data::Exception(error);
}

std::vector<int> workerDevices()
{
    std::vector<int> devs = sela::devices();
    if (devs.empty())
        throw data::Exception("no HIP device visible (the SELA MI355X path has no CPU fallback)");
    return devs;
}

} // namespace

namespace sela {

file::SelaFile Encoder::process()
{
    const size_t dataBytes = wavFile.readHeader(ifStream);
    const uint32_t channels = wavFile.numChannels;
    if (channels == 0 || channels > 255)
        throw data::Exception("Encoder: unsupported channel count");
    wavFile.pcm.resize(dataBytes / 2);
    const size_t frames = wavFile.frameCount(); // tail samples beyond the last whole frame are dropped
    sela_host::PinnedBuffer<uint8_t> bytes;
    std::vector<uint64_t> offsets;
    streamEncode(&ifStream, wavFile.pcm.data(), frames, channels, bytes, offsets, [](const uint8_t*, size_t) {});
    const size_t coded = frames * kBlock * channels;
    if (wavFile.pcm.size() > coded && !readExact(ifStream, wavFile.pcm.data() + coded, (wavFile.pcm.size() - coded) * 2))
        throw data::Exception("data subChunk is shorter than its header says");
    file::SelaFile out(wavFile.sampleRate, wavFile.bitsPerSample, (uint8_t)channels, std::move(bytes), std::move(offsets));
    if (materializeFrames)
        out.materializeFrames();
    return out;
}

file::WavFile Decoder::process()
{
    const size_t payload = selaFile.readHeader(ifStream);
    sela_host::PinnedBuffer<int16_t> pcm;
    streamDecode(ifStream, selaFile, payload, pcm, [](const int16_t*, size_t) {});
    file::WavFile out(selaFile.selaHeader.sampleRate, (uint16_t)selaFile.selaHeader.channels, std::move(pcm));
    if (demuxFrames)
        out.demuxSamples();
    return out;
}

size_t encodeFile(std::ifstream& in, std::ofstream& out)
{
    file::WavFile wav;
    const size_t dataBytes = wav.readHeader(in);
    const uint32_t channels = wav.numChannels;
    if (channels == 0 || channels > 255)
        throw data::Exception("Encoder: unsupported channel count");
    const size_t frames = dataBytes / 2 / channels / kBlock;
    wav.pcm.resize(frames * kBlock * channels);
    file::SelaFile header;
    header.selaHeader.sampleRate = wav.sampleRate;
    header.selaHeader.bitsPerSample = wav.bitsPerSample;
    header.selaHeader.channels = (uint8_t)channels;
    header.selaHeader.numFrames = (uint32_t)frames;
    header.writeHeader(out);
    sela_host::PinnedBuffer<uint8_t> bytes;
    std::vector<uint64_t> offsets;
    size_t written = 0;
    streamEncode(&in, wav.pcm.data(), frames, channels, bytes, offsets, [&](const uint8_t* p, size_t done) {
        if (done > written) { // finished frames go to disk while later pieces are on the device
            out.write(reinterpret_cast<const char*>(p + written), (std::streamsize)(done - written));
            written = done;
        }
    });
    return frames;
}

size_t decodeFile(std::ifstream& in, std::ofstream& out)
{
    file::SelaFile sela;
    const size_t payload = sela.readHeader(in);
    const uint32_t channels = sela.selaHeader.channels;
    if (channels == 0)
        throw data::Exception("Decoder: unsupported channel count");
    const size_t announced = std::min<size_t>(sela.selaHeader.numFrames, payload / (4 + 12 * (size_t)channels));
    file::WavFile::writeHeader(out, sela.selaHeader.sampleRate, (uint16_t)channels, 16, (uint32_t)(announced * kBlock * channels * 2));
    sela_host::PinnedBuffer<int16_t> pcm;
    size_t written = 0;
    streamDecode(in, sela, payload, pcm, [&](const int16_t* p, size_t done) {
        if (done > written) {
            out.write(reinterpret_cast<const char*>(p + written), (std::streamsize)((done - written) * 2));
            written = done;
        }
    });
    const size_t frames = sela.frameCount();
    if (frames != announced) { // the stream ended early (bad sync word): the header sizes follow what was decoded
        out.seekp(0, std::ios::beg);
        file::WavFile::writeHeader(out, sela.selaHeader.sampleRate, (uint16_t)channels, 16, (uint32_t)(frames * kBlock * channels * 2));
        out.seekp(0, std::ios::end);
    }
    return frames;
}

void setDevices(const std::vector<int>& devs)
{
    std::lock_guard<std::mutex> lock(g_devicesMutex);
    g_devices = devs;
}

std::vector<int> devices()
{
    std::lock_guard<std::mutex> lock(g_devicesMutex);
    if (!g_devices.empty())
        return g_devices;
    std::vector<int> all;
    for (int d = 0; d < sela_hip_device_count(); d++)
        all.push_back(d);
    return all;
}

std::vector<file::SelaFile> encodeBatch(const std::vector<file::WavFile>& wavs)
{
    std::vector<file::SelaFile> out(wavs.size());
    std::vector<bool> done(wavs.size(), false);
    const std::vector<int> devs = workerDevices();
    for (size_t first = 0; first < wavs.size(); first++) {
        if (done[first])
            continue;
        const uint32_t channels = wavs[first].numChannels;
        if (channels == 0 || channels > 255)
            throw data::Exception("encodeBatch: unsupported channel count");
        // every not yet encoded file with this channel count joins the job: whole frames only,
        // tail samples beyond a file's last whole frame are dropped exactly as for a single file
        std::vector<size_t> members, trackFrames;
        for (size_t i = first; i < wavs.size(); i++) {
            if (done[i] || wavs[i].numChannels != channels)
                continue;
            members.push_back(i);
            trackFrames.push_back(wavs[i].frameCount());
            done[i] = true;
        }
        const std::vector<std::vector<Piece>> pieces = partitionPieces(trackFrames, devs.size());
        struct Coded {
            sela_host::PinnedBuffer<uint8_t> bytes;
            std::vector<uint64_t> offsets;
        };
        std::vector<std::vector<Coded>> coded(devs.size());
        for (size_t w = 0; w < devs.size(); w++)
            coded[w].resize(pieces[w].size());
        runOnDevices(pieces, devs, [&](size_t w, const std::vector<Piece>& mine) {
            for (size_t p = 0; p < mine.size(); p++) {
                const file::WavFile& wav = wavs[members[mine[p].track]];
                int16_t* pcm = const_cast<int16_t*>(wav.pcm.data()) + mine[p].first * kBlock * channels;
                streamEncode(nullptr, pcm, mine[p].n, channels, coded[w][p].bytes, coded[w][p].offsets, [](const uint8_t*, size_t) {});
            }
        });
        // the pieces' sizes meet here: every track's stream is its pieces back to back
        std::vector<std::vector<std::pair<size_t, size_t>>> ofTrack(members.size()); // (worker, piece) in frame order
        for (size_t w = 0; w < devs.size(); w++)
            for (size_t p = 0; p < pieces[w].size(); p++)
                ofTrack[pieces[w][p].track].push_back({ w, p });
        for (size_t t = 0; t < members.size(); t++) {
            const file::WavFile& wav = wavs[members[t]];
            sela_host::PinnedBuffer<uint8_t> bytes;
            std::vector<uint64_t> offsets(1, 0);
            if (ofTrack[t].size() == 1) { // the usual case: the whole track was one piece
                Coded& c = coded[ofTrack[t][0].first][ofTrack[t][0].second];
                bytes = std::move(c.bytes);
                offsets = std::move(c.offsets);
            } else {
                size_t total = 0;
                for (auto& wp : ofTrack[t])
                    total += coded[wp.first][wp.second].bytes.size();
                bytes.resize(total);
                size_t at = 0;
                for (auto& wp : ofTrack[t]) {
                    const Coded& c = coded[wp.first][wp.second];
                    if (c.bytes.size())
                        std::memcpy(bytes.data() + at, c.bytes.data(), c.bytes.size());
                    for (size_t f = 1; f < c.offsets.size(); f++)
                        offsets.push_back(at + c.offsets[f]);
                    at += c.bytes.size();
                }
            }
            out[members[t]] = file::SelaFile(wav.sampleRate, wav.bitsPerSample, (uint8_t)channels, std::move(bytes), std::move(offsets));
            if (Encoder::materializeFrames)
                out[members[t]].materializeFrames();
        }
    }
    return out;
}

std::vector<file::WavFile> decodeBatch(const std::vector<file::SelaFile>& selas)
{
    std::vector<file::WavFile> out(selas.size());
    std::vector<bool> done(selas.size(), false);
    const std::vector<int> devs = workerDevices();
    for (size_t first = 0; first < selas.size(); first++) {
        if (done[first])
            continue;
        const uint32_t channels = selas[first].selaHeader.channels;
        if (channels == 0)
            throw data::Exception("decodeBatch: unsupported channel count");
        std::vector<size_t> members, trackFrames;
        for (size_t i = first; i < selas.size(); i++) {
            if (done[i] || selas[i].selaHeader.channels != channels)
                continue;
            members.push_back(i);
            trackFrames.push_back(selas[i].frameCount()); // (0 for an empty or default-constructed file)
            done[i] = true;
        }
        std::vector<sela_host::PinnedBuffer<int16_t>> pcm(members.size());
        for (size_t t = 0; t < members.size(); t++)
            pcm[t].resize(trackFrames[t] * kBlock * channels);
        const std::vector<std::vector<Piece>> pieces = partitionPieces(trackFrames, devs.size());
        runOnDevices(pieces, devs, [&](size_t, const std::vector<Piece>& mine) {
            for (const Piece& p : mine) { // decoded samples land in their track's buffer: nothing to stitch
                const file::SelaFile& sela = selas[members[p.track]];
                if (sela_hip_decode(sela.frameBytes.data(), sela.frameOffsets.data() + p.first, (uint32_t)p.n, channels,
                        pcm[p.track].data() + p.first * kBlock * channels)
                    != SELA_HIP_OK)
                    gpuFailure("decodeBatch");
            }
        });
        for (size_t t = 0; t < members.size(); t++) {
            out[members[t]] = file::WavFile(selas[members[t]].selaHeader.sampleRate, (uint16_t)channels, std::move(pcm[t]));
            if (Decoder::demuxFrames)
                out[members[t]].demuxSamples();
        }
    }
    return out;
}

bool Encoder::materializeFrames = true;
bool Decoder::demuxFrames = true;

} // namespace sela
