// player.cpp -- see player.hpp.
#include "sela_host/player.hpp"

#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>

#include <unistd.h>

#include "sela_host/codec.hpp"

namespace sela {

namespace {

constexpr size_t kBlock = SELA_HIP_SAMPLES_PER_FRAME;

data::WavFormatSubChunk formatOf(uint32_t rate, uint16_t channels, uint16_t bps)
{
    data::WavFormatSubChunk f;
    f.subChunkId = "fmt ";
    f.subChunkSize = 16;
    f.audioFormat = 1;
    f.numChannels = channels;
    f.sampleRate = rate;
    f.bitsPerSample = bps;
    f.blockAlign = (uint16_t)(channels * bps / 8);
    f.byteRate = rate * f.blockAlign;
    return f;
}

RawPcmSink& standardOutputSink()
{
    static RawPcmSink sink(1);
    return sink;
}

// What the decoding thread tells the playing thread.
struct Arrivals : DecodedStream {
    std::mutex mu;
    std::condition_variable cv;
    data::SelaHeader header;
    size_t announced = 0, samples = 0;
    const int16_t* pcm = nullptr;
    bool begun = false, over = false;
    std::string error;

    void begin(const data::SelaHeader& h, size_t frames) override
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            header = h, announced = frames, begun = true;
        }
        cv.notify_all();
    }
    void ready(const int16_t* p, size_t n) override
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            pcm = p, samples = n;
        }
        cv.notify_all();
    }
    void end(const std::string& what)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            over = true, error = what;
        }
        cv.notify_all();
    }
};

} // namespace

Player::Player() : sink(standardOutputSink()) {}

void RawPcmSink::play(const data::AudioPacket& packet)
{
    const char* p = packet.audio;
    size_t n = packet.bufferSize;
    while (n) {
        const ssize_t put = ::write(fd, p, n);
        if (put < 0) {
            if (errno == EINTR)
                continue;
            throw data::Exception(std::string("cannot write the samples: ") + std::strerror(errno));
        }
        p += put, n -= (size_t)put;
    }
}

void Player::printProgress(size_t current, size_t total, bool last) const
{
    // the reference's bar (src/sela/player.cpp:106-131): "[=====>      ] 42% (n/total)"
    const size_t barWidth = 40;
    const double progress = total ? (double)current / (double)total : 1.0;
    const size_t pos = (size_t)(barWidth * progress);
    std::string output = "[";
    for (size_t i = 0; i < barWidth; ++i)
        output += i < pos ? "=" : i == pos ? ">" : " ";
    output += "] " + std::to_string((unsigned)(progress * 100)) + "% (" + std::to_string(current) + "/" + std::to_string(total) + ")\r";
    std::cerr << output << std::flush;
    if (last)
        std::cerr << std::endl;
}

void Player::play(const file::WavFile& wavFile)
{
    packetsPlayed = 0;
    const size_t frames = wavFile.frameCount(), packetBytes = kBlock * wavFile.numChannels * sizeof(int16_t);
    sink.open(formatOf(wavFile.sampleRate, wavFile.numChannels, wavFile.bitsPerSample));
    auto lastBar = std::chrono::steady_clock::now();
    for (size_t f = 0; f < frames; f++) {
        // (ao_play takes a non-const char*, and so does the reference's packet: the bytes are not written to)
        char* at = const_cast<char*>(reinterpret_cast<const char*>(wavFile.pcm.data())) + f * packetBytes;
        sink.play(data::AudioPacket(at, packetBytes));
        packetsPlayed++;
        if (showProgress && std::chrono::steady_clock::now() - lastBar >= std::chrono::milliseconds(100)) {
            printProgress(packetsPlayed, frames, false);
            lastBar = std::chrono::steady_clock::now();
        }
    }
    if (showProgress)
        printProgress(packetsPlayed, frames, true);
    sink.close();
}

size_t Player::playFile(const std::string& selaPath)
{
    const auto t0 = std::chrono::steady_clock::now();
    packetsPlayed = 0;
    firstPacketSeconds = 0;
    Arrivals arrivals;
    sela_host::PinnedBuffer<int16_t> pcm; // the decoded samples: the packets point into it (decoding ends long before playing does)
    std::thread decoder([&] {
        std::string what;
        try {
            (void)decodeFileTo(selaPath, arrivals, pcm);
        } catch (const data::Exception& e) {
            what = e.exceptionMessage.empty() ? "decoding failed" : e.exceptionMessage;
        } catch (const std::exception& e) {
            what = e.what();
        } catch (...) {
            what = "decoding failed";
        }
        arrivals.end(what);
    });
    std::string failure;
    bool opened = false;
    try {
        std::unique_lock<std::mutex> lock(arrivals.mu);
        arrivals.cv.wait(lock, [&] { return arrivals.begun || arrivals.over; });
        if (arrivals.begun) {
            const uint32_t channels = arrivals.header.channels;
            const size_t packetSamples = kBlock * channels, packetBytes = packetSamples * sizeof(int16_t);
            const data::WavFormatSubChunk format = formatOf(arrivals.header.sampleRate, channels, arrivals.header.bitsPerSample);
            lock.unlock();
            sink.open(format);
            opened = true;
            lock.lock();
            auto lastBar = std::chrono::steady_clock::now();
            for (;;) {
                const size_t next = packetsPlayed;
                arrivals.cv.wait(lock, [&] { return arrivals.over || arrivals.samples >= (next + 1) * packetSamples; });
                if (arrivals.samples < (next + 1) * packetSamples)
                    break; // the stream is over (or failed) and holds no further whole frame
                // every frame that is there, outside the lock: the decoder keeps reporting meanwhile
                const size_t there = arrivals.samples / packetSamples;
                const int16_t* const pcm = arrivals.pcm;
                const size_t total = arrivals.announced;
                lock.unlock();
                for (size_t f = next; f < there; f++) {
                    char* at = const_cast<char*>(reinterpret_cast<const char*>(pcm)) + f * packetBytes;
                    sink.play(data::AudioPacket(at, packetBytes));
                    if (packetsPlayed++ == 0)
                        firstPacketSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (showProgress && std::chrono::steady_clock::now() - lastBar >= std::chrono::milliseconds(100)) {
                        printProgress(packetsPlayed, total, false);
                        lastBar = std::chrono::steady_clock::now();
                    }
                }
                lock.lock();
            }
        }
    } catch (const data::Exception& e) {
        failure = e.exceptionMessage.empty() ? "the audio sink failed" : e.exceptionMessage;
    } catch (const std::exception& e) { // (whatever a sink throws: the decoder is joined before it travels on)
        failure = e.what();
    } catch (...) {
        failure = "the audio sink failed";
    }
    decoder.join();
    if (showProgress && failure.empty() && arrivals.error.empty())
        printProgress(packetsPlayed, packetsPlayed, true);
    if (opened)
        sink.close();
    if (!failure.empty())
        throw data::Exception(failure);
    if (!arrivals.error.empty())
        throw data::Exception(arrivals.error);
    return packetsPlayed;
}

} // namespace sela
