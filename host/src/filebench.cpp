// filebench.cpp -- end-to-end timing of the C++ host on one file (bench.py's `e2e` and `file_to_file` legs).
//
//   sela_filebench in.wav scratch_dir [repeats]
//
// Measures, with steady_clock, after one untimed warm-up of each (HIP initialisation, buffer pinning):
//   e2e         sela_hip_encode / sela_hip_decode on page-locked host buffers: H2D + kernels + D2H
//   file        sela::encodeFile / sela::decodeFile (by path): file read, H2D, kernels, D2H and file write overlapped
//               (what the reference's `sela -e` / `sela -d` do, src/main.cpp:29-41)
// and prints one JSON object: medians in ms and stereo Msamples/s.  Results are checked against each other
// (file-to-file .sela == header + e2e frames; decoded .wav == e2e decode) -- parity against the reference is
// what tests/ do.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "sela_hip.h"
#include "sela_host/codec.hpp"
#include "sela_host/frame.hpp"
#include "sela_host/player.hpp"

#include <mutex>

#include "sela_host/fileio.hpp"

namespace {

// `trace` mode: the pool's tasks and the feeding thread's waits of ONE file-to-file encode and decode on one time axis
struct TraceRow {
    std::string what;
    long long t0, t1;
    size_t bytes;
};
std::mutex g_traceMutex;
std::vector<TraceRow> g_trace;
void traceHook(const char* what, long long t0, long long t1, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_traceMutex);
    g_trace.push_back({ what, t0, t1, bytes });
}
void printTrace(const char* title, long long origin)
{
    std::sort(g_trace.begin(), g_trace.end(), [](const TraceRow& a, const TraceRow& b) { return a.t0 < b.t0; });
    std::printf("---- %s (us from the call; what, start, end, bytes or frames)\n", title);
    for (const TraceRow& r : g_trace)
        std::printf("%-22s %9.1f %9.1f %10zu\n", r.what.c_str(), (r.t0 - origin) / 1e3, (r.t1 - origin) / 1e3, r.bytes);
    g_trace.clear();
}

double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

std::vector<uint8_t> slurp(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s in.wav scratch_dir [repeats] [e2e|all] [io threads]   (e2e: only the host-pointer leg, for traces)\n", argv[0]);
        return 2;
    }
    if (std::string(argv[1]) == "frames") {
        // sela_filebench frames <threads> <frames per thread> [samples per channel = 2048] [bits = 16]: the reference's own fan-out
        // (src/sela/encoder.cpp:58-73: T threads, each constructing a frame::FrameEncoder per frame of its share) on the host
        // classes, then the decoders the same way.  Another length, or 17 bits, is the any-length route's business.
        const int threads = std::max(1, std::atoi(argv[2])), per = argc > 3 ? std::max(1, std::atoi(argv[3])) : 16;
        const int len = argc > 4 ? std::min(65535, std::max(128, std::atoi(argv[4]))) : 2048, bits = argc > 5 ? std::atoi(argv[5]) : 16;
        const int top = bits > 16 ? 60000 : 30000, step = bits > 16 ? 1201 : 601;
        using clock = std::chrono::steady_clock;
        std::vector<data::WavFrame> in;
        uint32_t x = 2463534242u;
        for (int i = 0; i < threads * per; i++) {
            std::vector<std::vector<int32_t>> ch(2, std::vector<int32_t>((size_t)len));
            int v[2] = { 0, 0 };
            for (int j = 0; j < len; j++)
                for (int c = 0; c < 2; c++) {
                    x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                    v[c] += (int)(x % (uint32_t)step) - step / 2;
                    v[c] = std::min(top, std::max(-top, v[c]));
                    ch[c][j] = v[c];
                }
            in.emplace_back(16, std::move(ch));
        }
        std::vector<data::SelaFrame> coded(in.size(), data::SelaFrame(16));
        std::vector<data::WavFrame> back(in.size(), data::WavFrame(16, {}));
        std::vector<std::string> errors(threads);
        auto fanOut = [&](bool encode) {
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++)
                pool.emplace_back([&, t] {
                    try {
                        for (int i = t * per; i < (t + 1) * per; i++) {
                            if (encode)
                                coded[i] = frame::FrameEncoder(in[i]).process();
                            else
                                back[i] = frame::FrameDecoder(coded[i]).process();
                        }
                    } catch (const data::Exception& e) {
                        errors[t] = e.exceptionMessage;
                    }
                });
            for (std::thread& th : pool)
                th.join();
        };
        fanOut(true), fanOut(false); // (untimed: contexts, pinned memory)
        const auto t0 = clock::now();
        fanOut(true);
        const auto t1 = clock::now();
        fanOut(false);
        const auto t2 = clock::now();
        size_t differing = 0;
        for (size_t i = 0; i < in.size(); i++)
            differing += back[i].samples != in[i].samples;
        for (const std::string& e : errors)
            if (!e.empty()) {
                std::fprintf(stderr, "%s\n", e.c_str());
                return 1;
            }
        const double encMs = std::chrono::duration<double, std::milli>(t1 - t0).count(), decMs = std::chrono::duration<double, std::milli>(t2 - t1).count();
        std::printf("{\"threads\": %d, \"frames\": %zu, \"samples_per_channel\": %d, \"bits\": %d, \"encode_ms\": %.3f, \"decode_ms\": %.3f, \"encode_msps\": %.1f, \"decode_msps\": %.1f, \"frames_not_lossless\": %zu}\n",
            threads, in.size(), len, bits, encMs, decMs, in.size() * (double)len / encMs / 1e3, in.size() * (double)len / decMs / 1e3, differing);
        return 0;
    }
    if (std::string(argv[1]) == "batch") {
        // sela_filebench batch <scratch dir> <devices, e.g. 0 or 0,0> <repeats> a.wav b.wav ...: the batch verbs in-process (HIP
        // initialised, buffers pinned by an untimed first pass), outputs removed before every timed pass
        if (argc < 6) {
            std::fprintf(stderr, "usage: %s batch scratch_dir devices repeats a.wav ...\n", argv[0]);
            return 2;
        }
        using clock = std::chrono::steady_clock;
        const std::string dir = argv[2], devs = argv[3];
        const int repeats = std::max(1, std::atoi(argv[4]));
        std::vector<int> devices;
        for (size_t at = 0; at <= devs.size();) {
            const size_t comma = std::min(devs.find(',', at), devs.size());
            devices.push_back(std::atoi(devs.substr(at, comma - at).c_str()));
            at = comma + 1;
        }
        sela::setDevices(devices);
        std::vector<std::string> wavs, selas, backs;
        for (int i = 5; i < argc; i++) {
            wavs.push_back(argv[i]);
            selas.push_back(dir + "/b" + std::to_string(i) + ".sela");
            backs.push_back(dir + "/b" + std::to_string(i) + ".wav");
        }
        try {
            std::vector<double> enc, dec;
            size_t inBytes = 0, outBytes = 0;
            for (int r = 0; r <= repeats; r++) {
                for (const std::string& p : selas)
                    std::remove(p.c_str());
                for (const std::string& p : backs)
                    std::remove(p.c_str());
                const auto t0 = clock::now();
                sela::encodeFiles(wavs, selas);
                const auto t1 = clock::now();
                sela::decodeFiles(selas, backs);
                const auto t2 = clock::now();
                if (r) {
                    enc.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
                    dec.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
                }
            }
            for (size_t i = 0; i < wavs.size(); i++) {
                inBytes += slurp(wavs[i]).size();
                outBytes += slurp(selas[i]).size();
            }
            std::printf("{\"files\": %zu, \"workers\": %zu, \"wav_bytes\": %zu, \"sela_bytes\": %zu, \"encode_ms\": %.3f, \"decode_ms\": %.3f}\n", wavs.size(),
                devices.size(), inBytes, outBytes, median(enc), median(dec));
            return 0;
        } catch (const data::Exception& e) {
            std::fprintf(stderr, "%s\n", e.exceptionMessage.c_str());
            return 1;
        }
    }
    const std::string wavPath = argv[1], dir = argv[2];
    const int repeats = argc > 3 ? std::max(1, std::atoi(argv[3])) : 9;
    const bool onlyE2e = argc > 4 && std::string(argv[4]) == "e2e";
    if (argc > 5)
        sela::setIoThreads((unsigned)std::max(0, std::atoi(argv[5]))); // (experiments: sela_filebench in.wav dir repeats all N)
    const std::string selaPath = dir + "/filebench.sela", backPath = dir + "/filebench.wav";
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    try {
        file::WavFile wav;
        {
            std::ifstream in(wavPath, std::ios::binary);
            if (!in)
                throw data::Exception("cannot open " + wavPath);
            file::WavFile::demuxOnRead = false; // (the samples are all this tool needs)
            wav.readFromFile(in);
        }
        const uint32_t ch = wav.numChannels;
        const uint32_t frames = (uint32_t)wav.frameCount();
        const double samples = (double)frames * 2048;

        // ---- host-pointer API on page-locked buffers ----------------------------------------------------
        sela_host::PinnedBuffer<uint8_t> bytes(sela_hip_encode_bound_bytes(frames, ch));
        std::vector<uint64_t> offs(frames + 1);
        sela_host::PinnedBuffer<int16_t> back((size_t)frames * 2048 * ch);
        std::vector<double> enc, dec;
        for (int r = 0; r <= repeats; r++) {
            const auto t0 = clock::now();
            if (sela_hip_encode(wav.pcm.data(), frames, ch, 2048, bytes.data(), bytes.size(), offs.data()) != SELA_HIP_OK)
                throw data::Exception(sela_hip_last_error());
            const auto t1 = clock::now();
            if (!std::getenv("SELA_FILEBENCH_ENCODE_ONLY") && sela_hip_decode(bytes.data(), offs.data(), frames, ch, back.data()) != SELA_HIP_OK)
                throw data::Exception(sela_hip_last_error());
            const auto t2 = clock::now();
            if (r) { // (the first pass pins and allocates)
                enc.push_back(ms(t0, t1));
                dec.push_back(ms(t1, t2));
            }
        }
        if (onlyE2e) {
            std::printf("{\"frames\": %u, \"e2e_encode_ms\": %.4f, \"e2e_decode_ms\": %.4f}\n", frames, median(enc), median(dec));
            return 0;
        }
        // ---- the PCIe transfers alone (page-locked <-> device), for scale: PCM one way, frames the other -------------
        double h2d_pcm = 0, d2h_pcm = 0, h2d_sela = 0, d2h_sela = 0;
        {
            void* dev = nullptr;
            const size_t pcmBytes = (size_t)frames * 2048 * ch * 2, selaBytes = (size_t)offs[frames];
            if (hipMalloc(&dev, pcmBytes) == hipSuccess) {
                std::vector<double> a, b, c, d;
                for (int r = 0; r <= repeats; r++) {
                    const auto t0 = clock::now();
                    (void)hipMemcpy(dev, wav.pcm.data(), pcmBytes, hipMemcpyHostToDevice);
                    const auto t1 = clock::now();
                    (void)hipMemcpy(back.data(), dev, pcmBytes, hipMemcpyDeviceToHost);
                    const auto t2 = clock::now();
                    (void)hipMemcpy(dev, bytes.data(), selaBytes, hipMemcpyHostToDevice);
                    const auto t3 = clock::now();
                    (void)hipMemcpy(bytes.data(), dev, selaBytes, hipMemcpyDeviceToHost);
                    const auto t4 = clock::now();
                    if (r)
                        a.push_back(ms(t0, t1)), b.push_back(ms(t1, t2)), c.push_back(ms(t2, t3)), d.push_back(ms(t3, t4));
                }
                h2d_pcm = median(a), d2h_pcm = median(b), h2d_sela = median(c), d2h_sela = median(d);
                (void)hipFree(dev);
                // (back and bytes were overwritten with the same contents they held)
            }
        }
        if (argc > 4 && std::string(argv[4]) == "trace") { // (after the warm-up above: buffers pinned, pool started)
            for (int r = 0; r < 2; r++) {
                std::remove(selaPath.c_str());
                std::remove(backPath.c_str());
                sela_host::ioTrace = r ? traceHook : nullptr;
                const long long a = sela_host::ioNow();
                sela::encodeFile(wavPath, selaPath);
                const long long b = sela_host::ioNow();
                if (r)
                    printTrace("sela::encodeFile", a), std::printf("encodeFile: %.1f us\n", (b - a) / 1e3);
                const long long c = sela_host::ioNow();
                sela::decodeFile(selaPath, backPath);
                const long long d = sela_host::ioNow();
                if (r)
                    printTrace("sela::decodeFile", c), std::printf("decodeFile: %.1f us\n", (d - c) / 1e3);
            }
            return 0;
        }
        // ---- file to file -------------------------------------------------------------------------------------
        std::vector<double> fenc, fdec;
        for (int r = 0; r <= repeats; r++) {
            // Into NEW files: opening an existing 32 MB file with O_TRUNC takes the kernel 4 ms to free its pages
            // (tools/io_probe.cpp), which is not the codec's time -- the outputs of the last round are removed first.
            std::remove(selaPath.c_str());
            std::remove(backPath.c_str());
            const auto t0 = clock::now();
            sela::encodeFile(wavPath, selaPath);
            const auto t1 = clock::now();
            sela::decodeFile(selaPath, backPath);
            const auto t2 = clock::now();
            if (r) {
                fenc.push_back(ms(t0, t1));
                fdec.push_back(ms(t1, t2));
            }
        }
        // ---- the player's feed (sela::Player::playFile into a sink that only compares): time to the first packet ---
        struct Comparing : sela::AudioSink {
            const int16_t* expect;
            size_t at = 0;
            bool same = true;
            void open(const data::WavFormatSubChunk&) override {}
            void play(const data::AudioPacket& p) override
            {
                same = same && std::memcmp(p.audio, reinterpret_cast<const char*>(expect) + at, p.bufferSize) == 0;
                at += p.bufferSize;
            }
        };
        std::vector<double> pfirst, pall;
        bool samePlayed = true;
        for (int r = 0; r <= repeats; r++) {
            Comparing sink;
            sink.expect = back.data();
            sela::Player player(sink);
            const auto t0 = clock::now();
            player.playFile(selaPath);
            const auto t1 = clock::now();
            samePlayed = samePlayed && sink.same && sink.at == back.size() * 2;
            if (r) {
                pfirst.push_back(player.firstPacketSeconds * 1e3);
                pall.push_back(ms(t0, t1));
            }
        }
        // the two paths agree with each other
        const std::vector<uint8_t> selaFile = slurp(selaPath), backFile = slurp(backPath);
        const bool sameSela = selaFile.size() == 15 + (size_t)offs[frames] && std::memcmp(selaFile.data() + 15, bytes.data(), (size_t)offs[frames]) == 0;
        const bool sameWav = backFile.size() == 44 + back.size() * 2 && std::memcmp(backFile.data() + 44, back.data(), back.size() * 2) == 0;
        std::printf("{\"frames\": %u, \"channels\": %u, \"repeats\": %d, \"sela_bytes\": %zu, "
                    "\"e2e_encode_ms\": %.4f, \"e2e_decode_ms\": %.4f, \"e2e_encode_msps\": %.1f, \"e2e_decode_msps\": %.1f, "
                    "\"file_encode_ms\": %.4f, \"file_decode_ms\": %.4f, \"file_encode_msps\": %.1f, \"file_decode_msps\": %.1f, "
                    "\"pcie_h2d_pcm_ms\": %.4f, \"pcie_d2h_pcm_ms\": %.4f, \"pcie_h2d_sela_ms\": %.4f, \"pcie_d2h_sela_ms\": %.4f, "
                    "\"play_first_packet_ms\": %.4f, \"play_all_ms\": %.4f, \"file_equals_e2e\": %s}\n",
            frames, ch, repeats, (size_t)offs[frames], median(enc), median(dec), samples / median(enc) / 1e3, samples / median(dec) / 1e3,
            median(fenc), median(fdec), samples / median(fenc) / 1e3, samples / median(fdec) / 1e3, h2d_pcm, d2h_pcm, h2d_sela, d2h_sela,
            median(pfirst), median(pall), (sameSela && sameWav && samePlayed) ? "true" : "false");
        return (sameSela && sameWav && samePlayed) ? 0 : 1;
    } catch (const data::Exception& e) {
        std::fprintf(stderr, "%s\n", e.exceptionMessage.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
    }
    return 1;
}
