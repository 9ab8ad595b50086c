// codec.cpp -- sela::Encoder / sela::Decoder, file-to-file streaming, and the multi-GPU batch dispatcher,
// all on top of libsela_hip.so's host-pointer API.
#include "sela_host/codec.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "sela_format.h"
#include "sela_hip.h"
#include "sela_host/fileio.hpp"

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

// Encode `frames` frames at pcm that are arriving in memory front to back: need(n) returns once the first n frames are
// there (an ifstream read, or a wait for read-ahead tasks; a no-op for samples that are in memory already).  Every
// `piece` frames are one feed = one kernel launch.  drain(bytes, n) is told whenever more of the output is final.
// beforeRealloc(): called before `bytes` is moved to a larger block (the retry with the certain bound) -- a sink that
// still READS the old block on another thread (WriteBehind) must be through with it by then.
template <typename Need, typename Drain>
void streamEncode(Need need, size_t piece, int16_t* pcm, size_t frames, uint32_t channels, sela_host::PinnedBuffer<uint8_t>& bytes,
    std::vector<uint64_t>& offsets, Drain drain, const std::function<void()>& beforeRealloc = {})
{
    const size_t frameSamples = kBlock * channels;
    offsets.assign(frames + 1, 0);
    piece = std::max<size_t>(piece, 1);
    for (int attempt = 0; attempt < 2; attempt++) {
        if (attempt == 1 && beforeRealloc)
            beforeRealloc(); // (resize frees the block the sink was draining from)
        bytes.resize(attempt == 0 ? optimisticBytes(frames, channels) : sela_hip_encode_bound_bytes((uint32_t)frames, channels));
        sela_hip_job* job = nullptr;
        if (sela_hip_encode_begin(&job, channels, (uint32_t)frames, bytes.data(), bytes.size(), offsets.data()) != SELA_HIP_OK)
            gpuFailure("Encoder");
        int rc = SELA_HIP_OK;
        for (size_t f0 = 0; f0 < frames && rc == SELA_HIP_OK; f0 += piece) {
            const size_t nf = std::min<size_t>(piece, frames - f0);
            try {
                need(f0 + nf); // the device works on the earlier pieces while this one arrives
            } catch (...) {
                (void)sela_hip_encode_end(job, nullptr, nullptr);
                throw;
            }
            uint64_t done = 0;
            const long long t0 = sela_host::ioTrace ? sela_host::ioNow() : 0;
            rc = sela_hip_encode_feed(job, pcm + f0 * frameSamples, (uint32_t)nf, nullptr, &done);
            if (sela_host::ioTrace)
                sela_host::ioTrace("encode feed", t0, sela_host::ioNow(), nf * frameSamples * 2);
            if (rc == SELA_HIP_OK)
                drain(bytes.data(), (size_t)done);
        }
        // everything is queued: hand finished bytes on while the last launches run (a feed of zero frames only reports)
        uint64_t reported = 0;
        const long long tPoll = sela_host::ioTrace ? sela_host::ioNow() : 0;
        for (uint32_t finalFrames = 0; rc == SELA_HIP_OK && frames && finalFrames < frames;) {
            uint64_t done = 0;
            rc = sela_hip_encode_feed(job, pcm, 0, &finalFrames, &done);
            if (sela_host::ioTrace && done > reported)
                sela_host::ioTrace("encoded bytes final", tPoll, sela_host::ioNow(), (size_t)done), reported = done;
            if (rc == SELA_HIP_OK)
                drain(bytes.data(), (size_t)done);
            if (finalFrames < frames)
                std::this_thread::sleep_for(std::chrono::microseconds(20));
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

// need() of a file that is read with the calling thread's own ifstream reads
struct StreamReader {
    std::ifstream& in;
    int16_t* pcm;
    size_t frameSamples, readFrames = 0;
    void operator()(size_t upTo)
    {
        if (upTo > readFrames) {
            if (!readExact(in, pcm + readFrames * frameSamples, (upTo - readFrames) * frameSamples * 2))
                throw data::Exception("data subChunk is shorter than its header says");
            readFrames = upTo;
        }
    }
};

// Decode the frame stream that is arriving in sela.frameBytes front to back (payload bytes behind the 15-byte header;
// fetch(have, payload) brings more of it in and returns the new number of bytes there) into pcm; drain(samples, n) is
// told whenever more samples are final.  Fills sela.frameOffsets.  Only frames [firstFrame, firstFrame + maxFrames) are
// decoded (the batch dispatcher cuts a file between workers; frames before the range are only indexed); pcm holds
// those frames from its start.
template <typename Fetch, typename Drain>
void streamDecode(Fetch fetch, file::SelaFile& sela, size_t payload, sela_host::PinnedBuffer<int16_t>& pcm, Drain drain, size_t firstFrame = 0,
    size_t maxFrames = (size_t)-1, size_t firstFeedFrames = kPieceFrames /* the first feed does not wait for a whole piece (a player wants its first samples early) */)
{
    const uint32_t channels = sela.selaHeader.channels;
    if (channels == 0)
        throw data::Exception("Decoder: unsupported channel count");
    const size_t frameSamples = kBlock * channels;
    // the header's frame count is not trusted for sizing: a frame has at least 4 + 12 bytes per channel
    const size_t plausible = std::min<size_t>(sela.selaHeader.numFrames, payload / (4 + 12 * (size_t)channels));
    const size_t stop = std::min(plausible, maxFrames == (size_t)-1 ? plausible : firstFrame + maxFrames); // frames indexed at most
    const size_t mine = stop > firstFrame ? stop - firstFrame : 0;
    if (sela.frameBytes.size() < payload)
        sela.frameBytes.resize(payload);
    sela.frameOffsets.assign(stop + 1, 0);
    pcm.resize(mine * frameSamples);
    sela_hip_job* job = nullptr;
    if (sela_hip_decode_begin(&job, channels, (uint32_t)mine, pcm.data()) != SELA_HIP_OK)
        gpuFailure("Decoder");
    size_t have = 0, indexed = 0, fed = std::min(firstFrame, stop);
    bool ended = false; // a frame without a sync word: the stream stops there for good
    int rc = SELA_HIP_OK;
    std::vector<uint64_t> local(kPieceFrames + 1);
    while (rc == SELA_HIP_OK && ((have < payload && indexed < stop && !ended) || fed < indexed)) {
        if (have < payload && indexed < stop && !ended) {
            try {
                have = fetch(have, payload);
            } catch (...) {
                (void)sela_hip_decode_end(job, nullptr);
                throw;
            }
        }
        // index the frames that are complete in what has been read; like the reference, stop for good at the
        // first one without a sync word (which this cannot tell from "not yet read" until the file is in)
        for (;;) {
            const size_t want = std::min<size_t>(kPieceFrames, stop - indexed);
            if (want == 0)
                break;
            const uint64_t base = sela.frameOffsets[indexed];
            const uint32_t found = sela_hip_index_frames(sela.frameBytes.data() + base, have - (size_t)base, (uint32_t)want, channels, local.data());
            for (uint32_t f = 1; f <= found; f++)
                sela.frameOffsets[indexed + f] = base + local[f];
            indexed += found;
            if (found < want) {
                ended = have == payload;
                break;
            }
        }
        const bool last = have == payload || indexed == stop || ended;
        if (indexed > fed && (indexed - fed >= (fed == std::min(firstFrame, stop) ? std::min<size_t>(firstFeedFrames, kPieceFrames) : kPieceFrames) || last)) {
            uint32_t done = 0;
            const long long t0 = sela_host::ioTrace ? sela_host::ioNow() : 0;
            rc = sela_hip_decode_feed(job, sela.frameBytes.data(), sela.frameOffsets.data() + fed, (uint32_t)(indexed - fed), &done);
            if (sela_host::ioTrace)
                sela_host::ioTrace("decode feed", t0, sela_host::ioNow(), indexed - fed);
            fed = indexed;
            if (rc == SELA_HIP_OK)
                drain(pcm.data(), (size_t)done * frameSamples);
        } else if (firstFeedFrames < kPieceFrames && fed > std::min(firstFrame, stop)) {
            // an eager consumer also hears of finished frames between the feeds (a feed of no frames only looks)
            uint32_t done = 0;
            rc = sela_hip_decode_feed(job, sela.frameBytes.data(), sela.frameOffsets.data(), 0, &done);
            if (rc == SELA_HIP_OK)
                drain(pcm.data(), (size_t)done * frameSamples);
        }
        if (have == payload && fed >= indexed)
            break;
    }
    // everything is queued: hand finished samples on while the last chunks are decoded and copied out
    uint32_t reportedFrames = 0;
    const long long tPoll = sela_host::ioTrace ? sela_host::ioNow() : 0;
    for (uint32_t finalFrames = 0; rc == SELA_HIP_OK && fed > firstFrame && finalFrames < fed - std::min(firstFrame, stop);) {
        rc = sela_hip_decode_feed(job, sela.frameBytes.data(), sela.frameOffsets.data(), 0, &finalFrames);
        if (sela_host::ioTrace && finalFrames > reportedFrames)
            sela_host::ioTrace("decoded frames final", tPoll, sela_host::ioNow(), finalFrames), reportedFrames = finalFrames;
        if (rc == SELA_HIP_OK)
            drain(pcm.data(), (size_t)finalFrames * frameSamples);
        if (finalFrames < fed - std::min(firstFrame, stop))
            std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    const std::string feedError = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
    uint32_t done = 0;
    const int rcEnd = sela_hip_decode_end(job, &done);
    if (rc == SELA_HIP_OK)
        rc = rcEnd;
    if (rc != SELA_HIP_OK)
        throw data::Exception("Decoder: " + (feedError.empty() ? std::string(sela_hip_last_error()) : feedError));
    sela.frameOffsets.resize(indexed + 1);
    if (firstFrame == 0 && maxFrames == (size_t)-1)
        sela.frameBytes.resize((size_t)sela.frameOffsets.back());
    const size_t decoded = indexed > firstFrame ? indexed - firstFrame : 0;
    pcm.resize(decoded * frameSamples);
    drain(pcm.data(), pcm.size());
}

// A stream whose frames do not all say 2048 samples: the reference's encoder never writes one, its decoder takes one
// (every subframe brings its own samplesPerChannel, src/frame/frame_decoder.cpp:24-25; file::WavFile sizes its header by the
// frames it gets, src/file/wav_file.cpp:9-14).  The streaming jobs are the fast path for 2048 everywhere; such a stream goes,
// whole, through sela_hip_decode's any-length route instead.  sela.frameBytes holds the whole payload.  Returns false for a
// stream that is an ordinary one (the caller's first error stands).
bool decodeOddStream(file::SelaFile& sela, size_t payload, sela_host::PinnedBuffer<int16_t>& pcm)
{
    const uint32_t channels = sela.selaHeader.channels;
    const size_t plausible = std::min<size_t>(sela.selaHeader.numFrames, payload / (4 + 12 * (size_t)channels));
    sela.frameOffsets.assign(plausible + 1, 0);
    const uint32_t found = sela_hip_index_frames(sela.frameBytes.data(), payload, (uint32_t)plausible, channels, sela.frameOffsets.data());
    sela.frameOffsets.resize((size_t)found + 1);
    std::vector<uint64_t> sampleOffsets((size_t)found + 1);
    const uint32_t largest = sela_hip_index_samples(sela.frameBytes.data(), sela.frameOffsets.data(), found, channels, sampleOffsets.data());
    bool ordinary = largest == kBlock;
    for (uint32_t f = 0; ordinary && f <= found; f++)
        ordinary = sampleOffsets[f] == (uint64_t)f * kBlock;
    if (ordinary || largest == 0)
        return false;
    pcm.resize(std::max<size_t>((size_t)sampleOffsets[found], (size_t)found * kBlock) * channels); // (sela_hip.h: the fast kernels are tried first)
    if (sela_hip_decode(sela.frameBytes.data(), sela.frameOffsets.data(), found, channels, pcm.data()) != SELA_HIP_OK)
        gpuFailure("Decoder");
    pcm.resize((size_t)sampleOffsets[found] * channels);
    sela.frameBytes.resize((size_t)sela.frameOffsets.back());
    return true;
}

// fetch() of a file that is read with the calling thread's own ifstream reads, kPieceBytes at a time
struct StreamFetcher {
    std::ifstream& in;
    file::SelaFile& sela;
    size_t operator()(size_t have, size_t payload)
    {
        const size_t n = std::min(kPieceBytes, payload - have);
        if (!readExact(in, sela.frameBytes.data() + have, n))
            throw data::Exception("File is too small, probably not a sela file.");
        return have + n;
    }
};

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

// Run work(worker, pieces) on one host thread per device; the first exception is rethrown in the caller.  onFailure() runs
// in the failing worker's thread for ANY failure of that worker -- its device not coming up included -- before the others
// are joined: workers that wait for each other (a track cut across GPUs) must hear of it, or join() never returns.
template <typename Work>
void runOnDevices(const std::vector<std::vector<Piece>>& pieces, const std::vector<int>& devs, Work work, const std::function<void()>& onFailure = {})
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
                if (onFailure)
                    onFailure();
                std::lock_guard<std::mutex> lock(errorMutex);
                if (error.empty())
                    error = e.exceptionMessage;
            } catch (const std::exception& e) {
                if (onFailure)
                    onFailure();
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
    streamEncode(StreamReader{ ifStream, wavFile.pcm.data(), kBlock * channels }, kPieceFrames, wavFile.pcm.data(), frames, channels, bytes, offsets,
        [](const uint8_t*, size_t) {});
    const size_t coded = frames * kBlock * channels;
    if (wavFile.pcm.size() > coded && !readExact(ifStream, wavFile.pcm.data() + coded, (wavFile.pcm.size() - coded) * 2))
        throw data::Exception("data subChunk is shorter than its header says");
    wavFile.syncChunk();
    file::SelaFile out(wavFile.sampleRate, wavFile.bitsPerSample, (uint8_t)channels, std::move(bytes), std::move(offsets));
    if (materializeFrames)
        out.materializeFrames();
    return out;
}

file::WavFile Decoder::process()
{
    const size_t payload = selaFile.readHeader(ifStream);
    sela_host::PinnedBuffer<int16_t> pcm;
    try {
        streamDecode(StreamFetcher{ ifStream, selaFile }, selaFile, payload, pcm, [](const int16_t*, size_t) {});
    } catch (const data::Exception&) { // frames that do not say 2048?  (decodeOddStream)
        ifStream.clear();
        ifStream.seekg(SELA_FILE_HEADER_BYTES, std::ios::beg);
        selaFile.frameBytes.resize(payload);
        if (!readExact(ifStream, selaFile.frameBytes.data(), payload) || !decodeOddStream(selaFile, payload, pcm))
            throw;
    }
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
    streamEncode(StreamReader{ in, wav.pcm.data(), kBlock * channels }, kPieceFrames, wav.pcm.data(), frames, channels, bytes, offsets, [&](const uint8_t* p, size_t done) {
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
    // Nothing goes to `out` before the stream's verdict is in: the streaming job reports a frame that does not say 2048 samples
    // only at its end (such frames decode to silence until then), and a std::ofstream cannot be cut back -- a fallback that
    // finds fewer samples than were already written would leave the stale ones behind the data chunk.  (The variant by path,
    // which can truncate its file, writes while it decodes.)
    sela_host::PinnedBuffer<int16_t> pcm;
    try {
        streamDecode(StreamFetcher{ in, sela }, sela, payload, pcm, [](const int16_t*, size_t) {});
    } catch (const data::Exception&) { // frames that do not say 2048?  (decodeOddStream)
        in.clear();
        in.seekg(SELA_FILE_HEADER_BYTES, std::ios::beg);
        sela.frameBytes.resize(payload);
        if (!readExact(in, sela.frameBytes.data(), payload) || !decodeOddStream(sela, payload, pcm))
            throw;
        file::WavFile::writeHeader(out, sela.selaHeader.sampleRate, (uint16_t)channels, 16, (uint32_t)(pcm.size() * 2));
        out.write(reinterpret_cast<const char*>(pcm.data()), (std::streamsize)(pcm.size() * 2));
        return sela.frameCount();
    }
    const size_t frames = sela.frameCount(); // (fewer than announced when the stream ended early: a bad sync word)
    (void)announced;
    file::WavFile::writeHeader(out, sela.selaHeader.sampleRate, (uint16_t)channels, 16, (uint32_t)(frames * kBlock * channels * 2));
    out.write(reinterpret_cast<const char*>(pcm.data()), (std::streamsize)(frames * kBlock * channels * 2));
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
                // (a feed is one kernel launch: samples that are in memory already go in one piece)
                streamEncode([](size_t) {}, mine[p].n, pcm, mine[p].n, channels, coded[w][p].bytes, coded[w][p].offsets, [](const uint8_t*, size_t) {});
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
                // (a job, not sela_hip_decode: the samples of a piece have their place in the track's buffer, 2048 per frame --
                // the one-shot call would take a stream with frames of another length down the any-length route and lay them
                // out by their own lengths; the batch verbs serve the shape files have, and a job refuses anything else)
                sela_hip_job* job = nullptr;
                int rc = sela_hip_decode_begin(&job, channels, (uint32_t)p.n, pcm[p.track].data() + p.first * kBlock * channels);
                if (rc == SELA_HIP_OK) {
                    rc = sela_hip_decode_feed(job, sela.frameBytes.data(), sela.frameOffsets.data() + p.first, (uint32_t)p.n, nullptr);
                    const std::string why = rc != SELA_HIP_OK ? sela_hip_last_error() : "";
                    const int rcEnd = sela_hip_decode_end(job, nullptr);
                    if (rc != SELA_HIP_OK)
                        throw data::Exception("decodeBatch: " + why);
                    rc = rcEnd;
                }
                if (rc != SELA_HIP_OK)
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

// ---- path-based verbs: positioned I/O on the pool's threads beside the feeding thread ------------------------------
namespace {

constexpr size_t kIoSubBytes = (size_t)1 << 20;  // one pread / pwrite task
constexpr size_t kFeedFrames = 2048;             // frames per encode feed while a file is being read (one launch each; a 3-minute track is two)

void put16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8); }
void put32(uint8_t* p, uint32_t v) { put16(p, v), put16(p + 2, v >> 16); }

// the 15 bytes of src/file/sela_file.cpp:108-112
void selaHeaderBytes(uint8_t (&h)[15], uint32_t rate, uint16_t bps, uint8_t channels, uint32_t frames)
{
    std::memcpy(h, "SeLa", 4);
    put32(h + 4, rate);
    put16(h + 8, bps);
    h[10] = channels;
    put32(h + 11, frames);
}

// the canonical 44 bytes of src/file/wav_file.cpp:227-243
void wavHeaderBytes(uint8_t (&h)[44], uint32_t rate, uint16_t channels, uint16_t bps, uint32_t dataBytes)
{
    std::memcpy(h, "RIFF", 4);
    put32(h + 4, 36 + dataBytes);
    std::memcpy(h + 8, "WAVEfmt ", 8);
    put32(h + 16, 16);
    put16(h + 20, 1);
    put16(h + 22, channels);
    put32(h + 24, rate);
    put32(h + 28, rate * channels * bps / 8);
    put16(h + 32, (uint16_t)(channels * bps / 8));
    put16(h + 34, bps);
    std::memcpy(h + 36, "data", 4);
    put32(h + 40, dataBytes);
}

struct WavInfo {
    uint32_t rate = 0;
    uint16_t channels = 0, bps = 16;
    size_t dataOffset = 0, dataBytes = 0, frames = 0;
};

// The header walk of file::WavFile::readHeader (same acceptance, same messages), for a file that is then read by offset.
WavInfo probeWav(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw data::Exception("cannot open " + path);
    file::WavFile wav;
    WavInfo info;
    info.dataBytes = wav.readHeader(in);
    info.dataOffset = (size_t)in.tellg();
    info.rate = wav.sampleRate;
    info.channels = wav.numChannels;
    info.bps = wav.bitsPerSample;
    if (info.channels == 0 || info.channels > 255)
        throw data::Exception("Encoder: unsupported channel count");
    info.frames = info.dataBytes / 2 / info.channels / kBlock; // tail samples beyond the last whole frame are dropped
    return info;
}

struct SelaInfo {
    data::SelaHeader header;
    size_t payload = 0, announced = 0; // bytes behind the header; frames the header promises (as far as the file can hold them)
};

SelaInfo probeSela(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw data::Exception("cannot open " + path);
    file::SelaFile sela;
    SelaInfo info;
    info.payload = sela.readHeader(in);
    info.header = sela.selaHeader;
    if (info.header.channels == 0)
        throw data::Exception("Decoder: unsupported channel count");
    info.announced = std::min<size_t>(info.header.numFrames, info.payload / (4 + 12 * (size_t)info.header.channels));
    return info;
}

// Frames [first, first + n) of a probed WAV -> .sela frame bytes, streamed: reads ahead on the pool, one feed per
// kFeedFrames, sink(bytes, final) is told whenever more of the output is final.  Returns the total.
template <typename Sink>
size_t encodeRange(const sela_host::PosixFile& in, const WavInfo& info, size_t first, size_t n, sela_host::PinnedBuffer<int16_t>& pcm,
    sela_host::PinnedBuffer<uint8_t>& bytes, std::vector<uint64_t>& offsets, Sink sink, const std::function<void()>& sinkQuiesce = {})
{
    const size_t frameBytes = kBlock * info.channels * 2;
    pcm.resize(n * kBlock * info.channels);
    sela_host::ReadAhead ahead(in, pcm.data(), info.dataOffset + first * frameBytes, n * frameBytes, kFeedFrames * frameBytes, kIoSubBytes);
    streamEncode([&](size_t upTo) { ahead.need(upTo * frameBytes); }, kFeedFrames, pcm.data(), n, info.channels, bytes, offsets, sink, sinkQuiesce);
    ahead.finish();
    return bytes.size();
}

size_t expectedSelaBytes(const WavInfo& info) { return info.frames * kBlock * info.channels * 2 / 4 * 3; }

} // namespace

void setIoThreads(unsigned n) { sela_host::IoPool::configure(n); }

size_t encodeFile(const std::string& inPath, const std::string& outPath)
{
    const WavInfo info = probeWav(inPath);
    const sela_host::PosixFile in = sela_host::PosixFile::openForRead(inPath);
    const sela_host::PosixFile out = sela_host::PosixFile::create(outPath);
    uint8_t header[15];
    selaHeaderBytes(header, info.rate, info.bps, (uint8_t)info.channels, (uint32_t)info.frames);
    out.writeAt(header, 15, 0);
    sela_host::PinnedBuffer<int16_t> pcm;
    sela_host::PinnedBuffer<uint8_t> bytes;
    std::vector<uint64_t> offsets;
    // (the pages of the output are allocated in one go while the input is read and coded: audio codes to about 3/4)
    sela_host::WriteBehind behind(out, 15, kIoSubBytes, expectedSelaBytes(info));
    const size_t total = encodeRange(in, info, 0, info.frames, pcm, bytes, offsets, [&](const uint8_t* p, size_t done) { behind.drain(p, done); }, [&] { behind.quiesce(); });
    behind.finish(&total);
    return info.frames;
}

size_t decodeFile(const std::string& inPath, const std::string& outPath)
{
    const SelaInfo info = probeSela(inPath);
    const uint32_t channels = info.header.channels;
    const size_t frameBytes = kBlock * channels * 2;
    const sela_host::PosixFile in = sela_host::PosixFile::openForRead(inPath);
    const sela_host::PosixFile out = sela_host::PosixFile::create(outPath);
    uint8_t header[44];
    wavHeaderBytes(header, info.header.sampleRate, (uint16_t)channels, 16, (uint32_t)(info.announced * frameBytes));
    out.writeAt(header, 44, 0);
    file::SelaFile sela;
    sela.selaHeader = info.header;
    sela.frameBytes.resize(info.payload);
    sela_host::PinnedBuffer<int16_t> pcm;
    sela_host::ReadAhead ahead(in, sela.frameBytes.data(), 15, info.payload, kIoSubBytes, kIoSubBytes);
    sela_host::WriteBehind behind(out, 44, kIoSubBytes, info.announced * frameBytes);
    try {
        streamDecode(
            [&](size_t have, size_t payload) {
                const size_t upTo = std::min(payload, have + kIoSubBytes);
                ahead.need(upTo);
                return upTo;
            },
            sela, info.payload, pcm, [&](const int16_t* p, size_t done) { behind.drain(p, done * 2); });
    } catch (const data::Exception&) { // frames that do not say 2048?  (decodeOddStream)
        ahead.need(info.payload);
        ahead.finish();
        behind.quiesce();
        if (!decodeOddStream(sela, info.payload, pcm))
            throw;
        const size_t bytes = pcm.size() * 2;
        wavHeaderBytes(header, info.header.sampleRate, (uint16_t)channels, 16, (uint32_t)bytes);
        out.writeAt(header, 44, 0);
        if (bytes)
            out.writeAt(pcm.data(), bytes, 44);
        out.truncate(44 + bytes);
        return sela.frameCount();
    }
    ahead.finish();
    const size_t frames = sela.frameCount(), decodedBytes = frames * frameBytes;
    behind.finish(&decodedBytes);
    if (frames != info.announced) { // the stream ended early (bad sync word): the header sizes follow what was decoded
        wavHeaderBytes(header, info.header.sampleRate, (uint16_t)channels, 16, (uint32_t)decodedBytes);
        out.writeAt(header, 44, 0);
        out.truncate(44 + decodedBytes);
    }
    return frames;
}

size_t decodeFileTo(const std::string& inPath, DecodedStream& to, sela_host::PinnedBuffer<int16_t>& pcm)
{
    const SelaInfo info = probeSela(inPath);
    const sela_host::PosixFile in = sela_host::PosixFile::openForRead(inPath);
    file::SelaFile sela;
    sela.selaHeader = info.header;
    sela.frameBytes.resize(info.payload);
    to.begin(info.header, info.announced);
    sela_host::ReadAhead ahead(in, sela.frameBytes.data(), 15, info.payload, kIoSubBytes, kIoSubBytes);
    streamDecode(
        [&](size_t have, size_t payload) {
            const size_t upTo = std::min(payload, have + kIoSubBytes);
            ahead.need(upTo);
            return upTo;
        },
        sela, info.payload, pcm, [&](const int16_t* p, size_t done) { to.ready(p, done); }, 0, (size_t)-1, 64);
    ahead.finish();
    return sela.frameCount();
}

// ---- many files by path: every GPU worker reads, codes and writes its own pieces ----------------------------------
namespace {

// What workers that share a track tell each other: the byte count of every piece of the track, in frame order
// (piece k of a track goes behind the header and the k pieces before it).
struct SharedTracks {
    std::mutex mu;
    std::condition_variable cv;
    bool aborted = false;
    std::vector<std::vector<int64_t>> pieceBytes; // [track][piece of the track] or -1
    void publish(size_t track, size_t piece, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            pieceBytes[track][piece] = (int64_t)bytes;
        }
        cv.notify_all();
    }
    // bytes of the pieces before `piece`; blocks until they are all known
    size_t before(size_t track, size_t piece)
    {
        std::unique_lock<std::mutex> lock(mu);
        size_t sum = 0;
        for (size_t k = 0; k < piece; k++) {
            cv.wait(lock, [&] { return aborted || pieceBytes[track][k] >= 0; });
            if (aborted)
                throw data::Exception("another worker failed");
            sum += (size_t)pieceBytes[track][k];
        }
        return sum;
    }
    void abort()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            aborted = true;
        }
        cv.notify_all();
    }
};

// An output file whose last ranges are still being written behind a worker's back while the worker reads and codes its next
// piece: the file, the page-locked buffer the bytes come from, and the strand that writes them.  A worker keeps two of
// these in flight (writes to DIFFERENT files run side by side, section 5.5 of DESIGN.md) and waits for the oldest.
template <typename T>
struct PendingOutput {
    sela_host::PosixFile out;
    sela_host::PinnedBuffer<T> data;
    std::unique_ptr<sela_host::WriteBehind> behind;
    bool cut = false;   // the file ends behind these bytes (allocated from an estimate)
    size_t cutTo = 0;
    void finish()
    {
        if (cut)
            behind->finish(&cutTo);
        else
            behind->finish();
    }
};
constexpr size_t kOutputsInFlight = 2;

template <typename T>
void settle(std::deque<std::unique_ptr<PendingOutput<T>>>& pending, size_t keep)
{
    while (pending.size() > keep) {
        pending.front()->finish();
        pending.pop_front();
    }
}

// the index of every piece within its track, in (worker, piece) order
std::vector<std::vector<size_t>> pieceIndexInTrack(const std::vector<std::vector<Piece>>& pieces, size_t tracks, std::vector<size_t>& piecesOfTrack)
{
    piecesOfTrack.assign(tracks, 0);
    std::vector<std::vector<size_t>> index(pieces.size());
    for (size_t w = 0; w < pieces.size(); w++)
        for (const Piece& p : pieces[w])
            index[w].push_back(piecesOfTrack[p.track]++);
    return index;
}

} // namespace

void encodeFiles(const std::vector<std::string>& inputs, const std::vector<std::string>& outputs)
{
    if (inputs.size() != outputs.size())
        throw data::Exception("encodeFiles: one output path per input");
    const std::vector<int> devs = workerDevices();
    sela_host::IoPool::instance().grow((unsigned)(3 * devs.size())); // (a write strand and two reads per GPU worker)
    std::vector<WavInfo> info(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) {
        info[i] = probeWav(inputs[i]);
        // every output exists with its header before the first worker starts: a track shorter than a frame has no piece
        const sela_host::PosixFile out = sela_host::PosixFile::create(outputs[i]);
        uint8_t header[15];
        selaHeaderBytes(header, info[i].rate, info[i].bps, (uint8_t)info[i].channels, (uint32_t)info[i].frames);
        out.writeAt(header, 15, 0);
    }
    std::vector<bool> done(inputs.size(), false);
    for (size_t first = 0; first < inputs.size(); first++) {
        if (done[first])
            continue;
        // every file with this channel count joins the job (the frames of one job have one layout)
        const uint16_t channels = info[first].channels;
        std::vector<size_t> members, trackFrames;
        for (size_t i = first; i < inputs.size(); i++) {
            if (done[i] || info[i].channels != channels)
                continue;
            members.push_back(i);
            trackFrames.push_back(info[i].frames);
            done[i] = true;
        }
        const std::vector<std::vector<Piece>> pieces = partitionPieces(trackFrames, devs.size());
        SharedTracks shared;
        std::vector<size_t> piecesOfTrack;
        const std::vector<std::vector<size_t>> indexInTrack = pieceIndexInTrack(pieces, members.size(), piecesOfTrack);
        for (size_t t = 0; t < members.size(); t++)
            shared.pieceBytes.emplace_back(piecesOfTrack[t], -1);
        const size_t frameBytes = kBlock * channels * 2;
        try {
            runOnDevices(pieces, devs, [&](size_t w, const std::vector<Piece>& mine) {
                try {
                    // A RUN = consecutive pieces read into one page-locked buffer back to back and coded as ONE job, so that many
                    // small tracks cost one set of launches instead of one each (a feed may span tracks); a piece larger than
                    // the run size is a run of its own and is written while it is still being coded.
                    constexpr size_t kRunFrames = 4096;
                    struct Deferred { // a piece that starts inside a track: its place is known when the pieces before it are
                        size_t track, piece;
                        sela_host::PinnedBuffer<uint8_t> bytes;
                    };
                    std::vector<Deferred> deferred;
                    std::deque<std::unique_ptr<PendingOutput<uint8_t>>> pending;
                    sela_host::PinnedBuffer<int16_t> pcm;
                    for (size_t p0 = 0; p0 < mine.size();) {
                        size_t p1 = p0 + 1, runFrames = mine[p0].n;
                        while (p1 < mine.size() && runFrames + mine[p1].n <= kRunFrames)
                            runFrames += mine[p1++].n;
                        sela_host::PinnedBuffer<uint8_t> bytes;
                        std::vector<uint64_t> offsets;
                        if (p1 == p0 + 1) { // one piece: streamed in and -- if it starts its track -- streamed out
                            const Piece& pc = mine[p0];
                            const size_t member = members[pc.track];
                            const sela_host::PosixFile in = sela_host::PosixFile::openForRead(inputs[member]);
                            if (pc.first == 0) {
                                // (the last writes of this file go on while the next piece is read and coded)
                                std::unique_ptr<PendingOutput<uint8_t>> po(new PendingOutput<uint8_t>);
                                po->out = sela_host::PosixFile::openForWrite(outputs[member]);
                                po->behind.reset(new sela_host::WriteBehind(po->out, 15, kIoSubBytes, expectedSelaBytes(info[member])));
                                sela_host::WriteBehind* const behind = po->behind.get();
                                encodeRange(in, info[member], 0, pc.n, pcm, po->data, offsets, [behind](const uint8_t* b, size_t n) { behind->drain(b, n); }, [behind] { behind->quiesce(); });
                                po->cut = pc.n == info[member].frames; // (the whole track: the file ends here)
                                po->cutTo = po->data.size();
                                shared.publish(pc.track, indexInTrack[w][p0], po->data.size());
                                pending.push_back(std::move(po));
                                settle(pending, kOutputsInFlight);
                            } else {
                                encodeRange(in, info[member], pc.first, pc.n, pcm, bytes, offsets, [](const uint8_t*, size_t) {});
                                shared.publish(pc.track, indexInTrack[w][p0], bytes.size());
                                deferred.push_back({ pc.track, indexInTrack[w][p0], std::move(bytes) });
                            }
                        } else { // several small pieces: one buffer, one job
                            pcm.resize(runFrames * kBlock * channels);
                            std::vector<sela_host::PosixFile> files;
                            std::vector<std::unique_ptr<sela_host::ReadAhead>> reads;
                            std::vector<size_t> startFrame; // of every piece in the run
                            size_t at = 0;
                            for (size_t p = p0; p < p1; p++) {
                                const size_t member = members[mine[p].track];
                                files.push_back(sela_host::PosixFile::openForRead(inputs[member]));
                                startFrame.push_back(at);
                                at += mine[p].n;
                            }
                            startFrame.push_back(at);
                            for (size_t p = p0; p < p1; p++) {
                                const size_t member = members[mine[p].track], k = p - p0;
                                reads.emplace_back(new sela_host::ReadAhead(files[k], pcm.data() + startFrame[k] * kBlock * channels,
                                    info[member].dataOffset + mine[p].first * frameBytes, mine[p].n * frameBytes, kFeedFrames * frameBytes, kIoSubBytes));
                            }
                            streamEncode(
                                [&](size_t upTo) {
                                    for (size_t k = 0; k < reads.size() && startFrame[k] < upTo; k++)
                                        reads[k]->need((std::min(upTo, startFrame[k + 1]) - startFrame[k]) * frameBytes);
                                },
                                kFeedFrames * 2, pcm.data(), runFrames, channels, bytes, offsets, [](const uint8_t*, size_t) {});
                            for (auto& r : reads)
                                r->finish();
                            for (size_t p = p0; p < p1; p++) {
                                const size_t k = p - p0;
                                const size_t b0 = (size_t)offsets[startFrame[k]], b1 = (size_t)offsets[startFrame[k + 1]];
                                shared.publish(mine[p].track, indexInTrack[w][p], b1 - b0);
                                if (mine[p].first == 0) {
                                    const sela_host::PosixFile out = sela_host::PosixFile::openForWrite(outputs[members[mine[p].track]]);
                                    out.writeAt(bytes.data() + b0, b1 - b0, 15);
                                } else {
                                    sela_host::PinnedBuffer<uint8_t> part;
                                    part.assign(bytes.data() + b0, b1 - b0);
                                    deferred.push_back({ mine[p].track, indexInTrack[w][p], std::move(part) });
                                }
                            }
                        }
                        p0 = p1;
                    }
                    settle(pending, 0);
                    for (Deferred& d : deferred) { // (the pieces before these belong to workers that do not wait for anybody)
                        const size_t at = shared.before(d.track, d.piece);
                        const sela_host::PosixFile out = sela_host::PosixFile::openForWrite(outputs[members[d.track]]);
                        sela_host::WriteBehind behind(out, 15 + at, kIoSubBytes);
                        behind.drain(d.bytes.data(), d.bytes.size());
                        behind.finish();
                        if (d.piece + 1 == piecesOfTrack[d.track]) // (the track's first piece allocated the file's pages from an estimate)
                            out.truncate(15 + at + d.bytes.size());
                    }
                } catch (...) {
                    shared.abort();
                    throw;
                }
            }, [&] { shared.abort(); }); // (also when a worker's device does not come up: the others may be waiting for its pieces' sizes)
        } catch (...) {
            shared.abort();
            throw;
        }
    }
}

void decodeFiles(const std::vector<std::string>& inputs, const std::vector<std::string>& outputs)
{
    if (inputs.size() != outputs.size())
        throw data::Exception("decodeFiles: one output path per input");
    const std::vector<int> devs = workerDevices();
    sela_host::IoPool::instance().grow((unsigned)(3 * devs.size()));
    std::vector<SelaInfo> info(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) {
        info[i] = probeSela(inputs[i]);
        const sela_host::PosixFile out = sela_host::PosixFile::create(outputs[i]);
        uint8_t header[44];
        wavHeaderBytes(header, info[i].header.sampleRate, info[i].header.channels, 16,
            (uint32_t)(info[i].announced * kBlock * info[i].header.channels * 2));
        out.writeAt(header, 44, 0);
    }
    std::vector<bool> done(inputs.size(), false);
    for (size_t first = 0; first < inputs.size(); first++) {
        if (done[first])
            continue;
        const uint32_t channels = info[first].header.channels;
        std::vector<size_t> members, trackFrames;
        for (size_t i = first; i < inputs.size(); i++) {
            if (done[i] || info[i].header.channels != channels)
                continue;
            members.push_back(i);
            trackFrames.push_back(info[i].announced);
            done[i] = true;
        }
        const std::vector<std::vector<Piece>> pieces = partitionPieces(trackFrames, devs.size());
        const size_t frameBytes = kBlock * channels * 2;
        std::mutex foundMutex;
        std::vector<size_t> found(members.size(), (size_t)-1); // frames a track really holds, where a worker saw its stream end early
        runOnDevices(pieces, devs, [&](size_t, const std::vector<Piece>& mine) {
            std::deque<std::unique_ptr<PendingOutput<int16_t>>> pending;
            for (const Piece& pc : mine) {
                // output offsets are frame x 2048 x channels x 2: nothing to agree on with the other workers.  A piece that
                // starts inside a file walks the frame headers before it (the bytes in front are read, not decoded).
                const size_t member = members[pc.track];
                const sela_host::PosixFile in = sela_host::PosixFile::openForRead(inputs[member]);
                std::unique_ptr<PendingOutput<int16_t>> po(new PendingOutput<int16_t>);
                po->out = sela_host::PosixFile::openForWrite(outputs[member]);
                file::SelaFile sela;
                sela.selaHeader = info[member].header;
                sela.frameBytes.resize(info[member].payload);
                sela_host::ReadAhead ahead(in, sela.frameBytes.data(), 15, info[member].payload, kIoSubBytes, kIoSubBytes);
                // (a piece that starts its file allocates the whole file's pages, for the pieces behind it too)
                po->behind.reset(new sela_host::WriteBehind(po->out, 44 + pc.first * frameBytes, kIoSubBytes,
                    pc.first == 0 ? info[member].announced * frameBytes : 0));
                sela_host::WriteBehind* const behind = po->behind.get();
                streamDecode(
                    [&](size_t have, size_t payload) {
                        const size_t upTo = std::min(payload, have + kIoSubBytes);
                        ahead.need(upTo);
                        return upTo;
                    },
                    sela, info[member].payload, po->data, [behind](const int16_t* p, size_t n) { behind->drain(p, n * 2); }, pc.first, pc.n);
                ahead.finish();
                pending.push_back(std::move(po)); // (its last samples are written while the next piece is read and decoded)
                settle(pending, kOutputsInFlight);
                if (sela.frameCount() < pc.first + pc.n) {
                    std::lock_guard<std::mutex> lock(foundMutex);
                    found[pc.track] = std::min(found[pc.track], sela.frameCount());
                }
            }
            settle(pending, 0);
        });
        for (size_t t = 0; t < members.size(); t++) {
            if (found[t] == (size_t)-1)
                continue; // every announced frame was there
            const size_t member = members[t];
            const sela_host::PosixFile out = sela_host::PosixFile::openForWrite(outputs[member]);
            uint8_t header[44];
            wavHeaderBytes(header, info[member].header.sampleRate, (uint16_t)channels, 16, (uint32_t)(found[t] * frameBytes));
            out.writeAt(header, 44, 0);
            out.truncate(44 + found[t] * frameBytes);
        }
    }
}

bool Encoder::materializeFrames = true;
bool Decoder::demuxFrames = true;

} // namespace sela
