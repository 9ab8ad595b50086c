// frame.cpp -- frame-level host glue: object <-> byte-stream conversion and the single-frame
// FrameEncoder / FrameDecoder entry points on top of libsela_hip.so.
#include "sela_host/frame.hpp"

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sela_hip.h"
#include "sela_host/buffer.hpp"

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

// ---- coalescing --------------------------------------------------------------------------------------------------
// The reference calls these classes from hardware_concurrency() threads at once, one frame per call
// (src/sela/encoder.cpp:58-73, src/sela/decoder.cpp:58-73).  One frame is a poor launch for a GPU: 3 of its 3072 slots.
// So concurrent calls are coalesced the way databases group commits: a call finds nobody ahead of it and runs at once;
// the calls that arrive while it is on the device queue up, and when it comes back ONE of them takes everything that
// is waiting (with its channel count) to the device as a single batch -- and so on.  Nobody waits for a timer, a lone
// caller pays nothing, and T busy threads end up in batches of about T frames.
namespace {

struct FrameJob {
    uint32_t channels = 0;
    const uint8_t* in = nullptr; // encode: interleaved int16 PCM of one frame; decode: the frame's on-disk bytes
    size_t inBytes = 0;
    std::vector<uint8_t> out;    // encode: the frame's on-disk bytes; decode: interleaved int16 PCM
    bool done = false, lead = false;
    std::string error;
};

class Coalescer {
    const bool encode;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<FrameJob*> queue;
    bool busy = false;
    size_t lastBatch = 0; // calls in the batch before this one
    // the leader's staging (one leader at a time): page-locked, reused from batch to batch
    sela_host::PinnedBuffer<uint8_t> in, out;
    std::vector<uint64_t> offsets;

    static constexpr size_t kMaxBatch = 4096;

    void runOne(FrameJob& j) // a batch of one, straight from and to the caller's memory
    {
        if (encode) {
            j.out.resize(sela_hip_encode_bound_bytes(1, j.channels));
            uint64_t offs[2] = { 0, 0 };
            if (sela_hip_encode(reinterpret_cast<const int16_t*>(j.in), 1, j.channels, (uint32_t)kBlock, j.out.data(), j.out.size(), offs) != SELA_HIP_OK)
                j.error = std::string("FrameEncoder: ") + sela_hip_last_error();
            else
                j.out.resize((size_t)offs[1]);
        } else {
            const uint64_t offs[2] = { 0, j.inBytes };
            j.out.resize(kBlock * j.channels * 2);
            if (sela_hip_decode(j.in, offs, 1, j.channels, reinterpret_cast<int16_t*>(j.out.data())) != SELA_HIP_OK)
                j.error = std::string("FrameDecoder: ") + sela_hip_last_error();
        }
    }

    void runBatch(const std::vector<FrameJob*>& batch)
    {
        const uint32_t channels = batch[0]->channels;
        const size_t n = batch.size(), framePcm = kBlock * channels * 2;
        if (n == 1)
            return runOne(*batch[0]);
        offsets.assign(n + 1, 0);
        if (encode) {
            in.resize(n * framePcm);
            for (size_t i = 0; i < n; i++)
                std::memcpy(in.data() + i * framePcm, batch[i]->in, framePcm);
            out.resize(sela_hip_encode_bound_bytes((uint32_t)n, channels));
            if (sela_hip_encode(reinterpret_cast<const int16_t*>(in.data()), (uint32_t)n, channels, (uint32_t)kBlock, out.data(), out.size(), offsets.data())
                != SELA_HIP_OK) {
                const std::string what = std::string("FrameEncoder: ") + sela_hip_last_error();
                for (FrameJob* j : batch)
                    j->error = what;
                return;
            }
            for (size_t i = 0; i < n; i++)
                batch[i]->out.assign(out.data() + offsets[i], out.data() + offsets[i + 1]);
        } else {
            size_t total = 0;
            for (size_t i = 0; i < n; i++)
                offsets[i] = total, total += (batch[i]->inBytes + 3) & ~(size_t)3; // (frames are whole words; keep every start aligned)
            offsets[n] = total;
            in.resize(total + 4);
            for (size_t i = 0; i < n; i++)
                std::memcpy(in.data() + offsets[i], batch[i]->in, batch[i]->inBytes);
            out.resize(n * framePcm);
            if (sela_hip_decode(in.data(), offsets.data(), (uint32_t)n, channels, reinterpret_cast<int16_t*>(out.data())) != SELA_HIP_OK) {
                // one caller's malformed frame must not fail its neighbours': everyone on their own
                for (FrameJob* j : batch)
                    runOne(*j);
                return;
            }
            for (size_t i = 0; i < n; i++)
                batch[i]->out.assign(out.data() + i * framePcm, out.data() + (i + 1) * framePcm);
        }
    }

public:
    explicit Coalescer(bool enc) : encode(enc) {}

    void submit(FrameJob& job)
    {
        std::unique_lock<std::mutex> lock(mu);
        queue.push_back(&job);
        if (!busy)
            busy = job.lead = true;
        cv.wait(lock, [&] { return job.done || job.lead; });
        if (job.done)
            return;
        // this call leads.  If the batch before held several calls, their threads are on their way back with their next
        // frames right now: give them until the queue has stopped growing for a moment (bounded), a trip costs more than that
        if (lastBatch > 1) {
            const auto t0 = std::chrono::steady_clock::now();
            size_t seen = queue.size();
            auto lastGrowth = t0;
            for (;;) {
                lock.unlock();
                std::this_thread::yield();
                lock.lock();
                const auto now = std::chrono::steady_clock::now();
                if (queue.size() != seen)
                    seen = queue.size(), lastGrowth = now;
                if (seen >= lastBatch || now - lastGrowth > std::chrono::microseconds(20) || now - t0 > std::chrono::microseconds(150))
                    break;
            }
        }
        // everything that is waiting with its channel count, itself included
        std::vector<FrameJob*> batch;
        for (auto it = queue.begin(); it != queue.end() && batch.size() < kMaxBatch;) {
            if ((*it)->channels == job.channels) {
                batch.push_back(*it);
                it = queue.erase(it);
            } else {
                ++it;
            }
        }
        lock.unlock();
        try {
            runBatch(batch);
        } catch (...) { // (out of page-locked memory, ...: the callers hear of it, nobody is left waiting)
            for (FrameJob* j : batch)
                j->error = encode ? "FrameEncoder: the batch could not be staged" : "FrameDecoder: the batch could not be staged";
        }
        // (the device buffers and streams this thread used go to whoever leads next: any of the callers may)
        sela_hip_thread_release();
        lock.lock();
        for (FrameJob* j : batch)
            j->done = true;
        lastBatch = batch.size();
        if (queue.empty())
            busy = false;
        else
            queue.front()->lead = true;
        lock.unlock();
        cv.notify_all();
    }
};

Coalescer& encodeCoalescer()
{
    static Coalescer* c = new Coalescer(true); // (never destroyed: calls may outlive the statics)
    return *c;
}
Coalescer& decodeCoalescer()
{
    static Coalescer* c = new Coalescer(false);
    return *c;
}

} // namespace

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
    FrameJob job;
    job.channels = (uint32_t)channels;
    job.in = reinterpret_cast<const uint8_t*>(pcm.data());
    job.inBytes = pcm.size() * 2;
    encodeCoalescer().submit(job);
    if (!job.error.empty())
        throw data::Exception(job.error);
    data::SelaFrame frame(wavFrame.bitsPerSample);
    parseFrame(job.out.data(), job.out.size(), (uint8_t)channels, wavFrame.bitsPerSample, frame);
    return frame;
}

data::WavFrame FrameDecoder::process()
{
    const size_t channels = selaFrame.subFrames.size();
    if (channels == 0 || channels > 255)
        throw data::Exception("FrameDecoder: a frame needs 1..255 subframes");
    std::vector<uint8_t> bytes;
    appendFrame(selaFrame, bytes);
    FrameJob job;
    job.channels = (uint32_t)channels;
    job.in = bytes.data();
    job.inBytes = bytes.size();
    decodeCoalescer().submit(job);
    if (!job.error.empty())
        throw data::Exception(job.error);
    const int16_t* pcm = reinterpret_cast<const int16_t*>(job.out.data());
    std::vector<std::vector<int32_t>> samples(channels, std::vector<int32_t>(kBlock));
    for (size_t i = 0; i < kBlock; i++)
        for (size_t c = 0; c < channels; c++)
            samples[c][i] = pcm[i * channels + c];
    return data::WavFrame(selaFrame.bitsPerSample, std::move(samples));
}

} // namespace frame
