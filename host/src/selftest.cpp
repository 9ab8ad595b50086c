// selftest.cpp -- CPU-only checks of the host's container code (no GPU needed).
// Exit code 0 = all passed.  Driven by tests/test_host_cpp.py.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>

#include "sela_hip.h"
#include "sela_host/fileio.hpp"
#include "sela_host/codec.hpp"
#include "sela_host/player.hpp"
#include "sela_host/stages.hpp"
#include "sela_host/files.hpp"
#include "sela_host/frame.hpp"

namespace {

int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                    \
        }                                                                  \
    } while (0)

std::string expectError(const std::string& path, bool asWav)
{
    try {
        std::ifstream in(path, std::ios::binary);
        if (asWav) {
            file::WavFile w;
            w.readFromFile(in);
        } else {
            file::SelaFile s;
            s.readFromFile(in);
        }
    } catch (const data::Exception& e) {
        return e.exceptionMessage;
    }
    return "";
}

void writeBytes(const std::string& path, const std::string& bytes)
{
    std::ofstream out(path, std::ios::binary);
    out.write(bytes.data(), (std::streamsize)bytes.size());
}

} // namespace

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";

    // ---- WAV round trip, tail handling, de-interleave ------------------------------------------------
    {
        std::vector<int16_t> pcm(2 * (2048 * 2 + 100));
        for (size_t i = 0; i < pcm.size(); i++)
            pcm[i] = (int16_t)((i * 7919u) ^ (i >> 3));
        file::WavFile w(44100, 2, std::vector<int16_t>(pcm));
        {
            std::ofstream out(dir + "/t.wav", std::ios::binary);
            w.writeToFile(out);
        }
        std::ifstream in(dir + "/t.wav", std::ios::binary);
        file::WavFile r;
        r.readFromFile(in);
        CHECK(r.numChannels == 2 && r.sampleRate == 44100 && r.bitsPerSample == 16);
        CHECK(r.pcm == pcm);
        CHECK(r.frameCount() == 2); // 100 tail samples per channel are dropped
        // the reference's view of the file (src/include/file/wav_file.hpp:14): same member names, no copy of the samples
        CHECK(r.wavChunk.chunkId == "RIFF" && r.wavChunk.format == "WAVE" && r.wavChunk.chunkSize == 36 + pcm.size() * 2);
        CHECK(r.wavChunk.formatSubChunk.numChannels == 2 && r.wavChunk.formatSubChunk.sampleRate == 44100 && r.wavChunk.formatSubChunk.bitsPerSample == 16);
        CHECK(r.wavChunk.formatSubChunk.blockAlign == 4 && r.wavChunk.formatSubChunk.byteRate == 44100 * 4 && r.wavChunk.formatSubChunk.audioFormat == 1);
        CHECK(r.wavChunk.dataSubChunk.subChunkSize == pcm.size() * 2 && r.wavChunk.dataSubChunk.channels == 2);
        CHECK(r.wavChunk.dataSubChunk.samples == r.pcm.data() && r.wavChunk.dataSubChunk.sampleCount == pcm.size());
        CHECK(r.wavChunk.dataSubChunk.wavFrames.size() == 2); // readFromFile de-interleaves like the reference's (demuxOnRead)
        r.demuxSamples(); // (again: idempotent)
        CHECK(r.wavChunk.dataSubChunk.wavFrames.size() == 2);
        CHECK(r.wavFrames.size() == 2 && r.wavFrames[1].samples[1][5] == pcm[(2048 + 5) * 2 + 1]);
    }
    // ---- WAV error paths (messages as the reference's src/file/wav_file.cpp) ---------------------------
    {
        writeBytes(dir + "/small.wav", "RIFF");
        CHECK(expectError(dir + "/small.wav", true) == "File is too small, probably not a wav file.");
        std::string bad(64, '\0');
        std::memcpy(&bad[0], "RIFX", 4);
        writeBytes(dir + "/bad.wav", bad);
        CHECK(expectError(dir + "/bad.wav", true) == "chunkId is not RIFF, probably not a wav file.");
        std::string w24(64, '\0');
        std::memcpy(&w24[0], "RIFF", 4);
        w24[4] = 56;
        std::memcpy(&w24[8], "WAVEfmt ", 8);
        w24[16] = 16;
        w24[20] = 1;
        w24[22] = 2;
        w24[34] = 24; // 24 bits per sample
        writeBytes(dir + "/w24.wav", w24);
        CHECK(expectError(dir + "/w24.wav", true) == "Only 16bits per sample wav is supported.");
    }
    // ---- several 'data' chunks: the last one is the audio, a 'fmt ' chunk counts wherever it stands (the reference walks
    // every chunk, src/file/wav_file.cpp:80-162) ------------------------------------------------------------------------
    {
        auto u32 = [](uint32_t v) { return std::string({ (char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24) }); };
        auto u16 = [](uint32_t v) { return std::string({ (char)v, (char)(v >> 8) }); };
        const std::string fmt = std::string("fmt ") + u32(16) + u16(1) + u16(1) + u32(8000) + u32(16000) + u16(2) + u16(16);
        const std::string first = std::string("data") + u32(8) + std::string("\x01\x00\x02\x00\x03\x00\x04\x00", 8);
        const std::string junk = std::string("LIST") + u32(4) + "abcd";
        const std::string second = std::string("data") + u32(6) + std::string("\x11\x00\x12\x00\x13\x00", 6);
        const std::string body = std::string("WAVE") + fmt + first + junk + second;
        writeBytes(dir + "/two.wav", std::string("RIFF") + u32((uint32_t)body.size()) + body);
        std::ifstream in(dir + "/two.wav", std::ios::binary);
        file::WavFile w;
        w.readFromFile(in);
        CHECK(w.numChannels == 1 && w.sampleRate == 8000);
        CHECK(w.pcm.size() == 3 && w.pcm[0] == 0x11 && w.pcm[2] == 0x13);
    }
    // ---- positioned I/O on the pool: a file read ahead in pieces, written behind in pieces -----------------------------
    {
        std::string blob(5 * 1000 * 1000 + 123, '\0');
        uint32_t x = 99;
        for (char& c : blob)
            c = (char)((x = x * 1664525u + 1013904223u) >> 24);
        writeBytes(dir + "/blob.bin", blob);
        try {
            const sela_host::PosixFile in = sela_host::PosixFile::openForRead(dir + "/blob.bin");
            CHECK(in.size() == blob.size());
            std::vector<char> got(blob.size() - 100);
            {
                sela_host::ReadAhead ahead(in, got.data(), 100, got.size(), 1 << 20, 1 << 18);
                ahead.need(10);
                CHECK(std::memcmp(got.data(), blob.data() + 100, 10) == 0);
                ahead.need(got.size());
                ahead.finish();
            }
            CHECK(std::memcmp(got.data(), blob.data() + 100, got.size()) == 0);
            {
                const sela_host::PosixFile out = sela_host::PosixFile::create(dir + "/copy.bin");
                out.writeAt("0123456", 7, 0);
                sela_host::WriteBehind behind(out, 7, 1 << 18);
                behind.drain(got.data(), 1000);
                behind.drain(got.data(), 3000000);
                behind.drain(got.data(), got.size());
                behind.finish();
                CHECK(out.size() == 7 + got.size());
            }
            std::ifstream back(dir + "/copy.bin", std::ios::binary);
            std::string copy((std::istreambuf_iterator<char>(back)), std::istreambuf_iterator<char>());
            CHECK(copy.size() == 7 + got.size() && copy.compare(0, 7, "0123456") == 0 && std::memcmp(copy.data() + 7, got.data(), got.size()) == 0);
            // a file that ends before the range does
            bool threw = false;
            try {
                std::vector<char> more(blob.size() + 4096);
                sela_host::ReadAhead ahead(in, more.data(), 0, more.size(), 1 << 20, 1 << 18);
                ahead.need(more.size());
            } catch (const data::Exception&) {
                threw = true;
            }
            CHECK(threw);
        } catch (const data::Exception& e) {
            std::fprintf(stderr, "FAIL exception: %s\n", e.exceptionMessage.c_str());
            failures++;
        }
    }
    // ---- .sela container: objects -> bytes -> objects ---------------------------------------------------
    {
        std::vector<data::SelaFrame> frames;
        for (int f = 0; f < 3; f++) {
            data::SelaFrame fr(16);
            for (int c = 0; c < 2; c++) {
                std::vector<uint32_t> cw = { 0x4914dd7fu + (uint32_t)f, 0x4a519ce4u, 0x6318c108u };
                std::vector<uint32_t> rw(10 + 5 * f + c, 0xA5A50000u + (uint32_t)c);
                fr.subFrames.emplace_back((uint8_t)c, (uint8_t)c, (uint8_t)0, data::RiceEncodedData(4, 17, cw), data::RiceEncodedData(7, 2048, rw));
            }
            frames.push_back(fr);
        }
        file::SelaFile s(48000, 16, 2, std::move(frames));
        CHECK(s.frameOffsets.size() == 4 && s.frameOffsets[1] == 4 + 2 * 12 + 4 * (3 + 10 + 3 + 11));
        {
            std::ofstream out(dir + "/t.sela", std::ios::binary);
            s.writeToFile(out);
        }
        std::ifstream in(dir + "/t.sela", std::ios::binary);
        file::SelaFile r;
        r.readFromFile(in);
        CHECK(r.selaHeader.sampleRate == 48000 && r.selaHeader.channels == 2 && r.selaHeader.numFrames == 3);
        CHECK(r.frameBytes == s.frameBytes && r.frameOffsets == s.frameOffsets);
        CHECK(r.selaFrames.size() == 3 && r.selaFrames[2].subFrames[1].encodedResidues.size() == 21);
        CHECK(r.selaFrames[1].subFrames[1].subFrameType == 1 && r.selaFrames[1].subFrames[0].optimumLpcOrder == 17);
        CHECK(r.selaFrames[0].subFrames[0].encodedReflectionCoefficients[0] == 0x4914dd7fu);
        // a broken sync word silently ends the file (src/file/sela_file.cpp:54-56)
        std::ifstream again(dir + "/t.sela", std::ios::binary);
        std::string raw((std::istreambuf_iterator<char>(again)), std::istreambuf_iterator<char>());
        raw[15 + (size_t)s.frameOffsets[1]] ^= 0x01;
        writeBytes(dir + "/cut.sela", raw);
        std::ifstream cut(dir + "/cut.sela", std::ios::binary);
        file::SelaFile c;
        c.readFromFile(cut);
        CHECK(c.selaFrames.size() == 1 && c.selaHeader.numFrames == 3);
        writeBytes(dir + "/magic.sela", "NoPe00000000000000");
        CHECK(expectError(dir + "/magic.sela", false) == "Magic number is incorrect, probably not a sela file.");
    }
    // ---- the player's feed: one packet per whole frame, views of the interleaved samples (src/sela/player.cpp:30-62) ----
    {
        struct Counting : sela::AudioSink {
            data::WavFormatSubChunk format;
            std::vector<std::pair<const char*, size_t>> packets;
            int opened = 0, closed = 0;
            void open(const data::WavFormatSubChunk& f) override { format = f, opened++; }
            void play(const data::AudioPacket& p) override { packets.emplace_back(p.audio, p.bufferSize); }
            void close() override { closed++; }
        } sink;
        std::vector<int16_t> pcm(3 * (2048 * 3 + 77)); // three channels, three frames and a tail
        for (size_t i = 0; i < pcm.size(); i++)
            pcm[i] = (int16_t)(i * 31u + (i >> 5));
        file::WavFile w(48000, 3, std::vector<int16_t>(pcm));
        sela::Player player(sink);
        player.play(w);
        CHECK(sink.opened == 1 && sink.closed == 1 && player.packetsPlayed == 3 && sink.packets.size() == 3);
        CHECK(sink.format.numChannels == 3 && sink.format.sampleRate == 48000 && sink.format.bitsPerSample == 16 && sink.format.blockAlign == 6
            && sink.format.byteRate == 48000 * 6);
        for (size_t f = 0; f < sink.packets.size(); f++) {
            CHECK(sink.packets[f].second == 2048 * 3 * 2);
            CHECK(sink.packets[f].first == reinterpret_cast<const char*>(w.pcm.data()) + f * 2048 * 3 * 2); // a view, in order
        }
        // the raw sink: the packets' bytes as they are
        const std::string path = dir + "/played.pcm";
        std::FILE* fp = std::fopen(path.c_str(), "wb");
        CHECK(fp != nullptr);
        if (fp) {
            sela::RawPcmSink raw(fileno(fp));
            sela::Player p2(raw);
            p2.play(w);
            std::fclose(fp);
            std::ifstream in(path, std::ios::binary);
            std::string got((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            CHECK(got.size() == 3 * 2048 * 3 * 2 && std::memcmp(got.data(), pcm.data(), got.size()) == 0);
        }
        // a file that is not there: the reference's kind of exception, nothing opened
        bool threw = false;
        try {
            player.playFile(dir + "/no_such_file.sela");
        } catch (const data::Exception&) {
            threw = true;
        }
        CHECK(threw && sink.opened == 1);
    }
    // ---- GPU mode: the reference's own frame tests (test/frametests.cpp:8-70) through the host classes --
    if (argc > 2 && std::string(argv[2]) == "gpu") {
        try {
            std::vector<int32_t> sine(2048);
            for (int i = 0; i < 2048; i++)
                sine[i] = (int32_t)(32767 * std::sin((double)i * (3.141592653589793238462643383279502884 / 180)));
            data::WavFrame input(16, { sine, sine });
            data::SelaFrame coded = frame::FrameEncoder(input).process();
            CHECK(coded.subFrames.size() == 2);
            // channel 0 independent (order 17, 552 residue words), channel 1 a silent difference
            CHECK(coded.subFrames[0].subFrameType == 0 && coded.subFrames[0].optimumLpcOrder == 17);
            CHECK(coded.subFrames[0].reflectionCoefficientRiceParam == 4 && coded.subFrames[0].residueRiceParam == 7);
            CHECK(coded.subFrames[0].encodedResidues.size() == 552 && coded.subFrames[0].encodedReflectionCoefficients.size() == 3);
            CHECK(coded.subFrames[0].encodedReflectionCoefficients[0] == 0x4914dd7fu);
            CHECK(coded.subFrames[1].subFrameType == 1 && coded.subFrames[1].parentChannelNumber == 0);
            CHECK(coded.subFrames[1].optimumLpcOrder == 1 && coded.subFrames[1].encodedResidues.size() == 64);
            data::WavFrame output = frame::FrameDecoder(coded).process();
            CHECK(output.samples.size() == 2 && output.samples[0] == sine && output.samples[1] == sine);
            // the reference's stage tests (test/ricetests.cpp:7-25, test/lpctests.cpp:10-32) through the L1 classes
            {
                std::vector<int32_t> values;
                for (size_t i = 0; i < 100; i++)
                    values.push_back(200 + (std::rand() % (201)));
                data::RiceDecodedData plain = data::RiceDecodedData(std::vector<int32_t>(values));
                rice::RiceEncoder enc = rice::RiceEncoder(plain);
                data::RiceEncodedData encodedData = enc.process();
                CHECK(encodedData.dataCount == 100 && encodedData.optimumRiceParam < MAX_RICE_PARAM && !encodedData.encodedData.empty());
                rice::RiceDecoder dec = rice::RiceDecoder(encodedData);
                data::RiceDecodedData decodedData = dec.process();
                CHECK(plain.decodedData.size() == decodedData.decodedData.size() && plain.decodedData == decodedData.decodedData);

                data::LpcDecodedData block = data::LpcDecodedData((uint8_t)16, std::vector<int32_t>(sine));
                lpc::ResidueGenerator resGen = lpc::ResidueGenerator(block);
                data::LpcEncodedData encoded = resGen.process();
                CHECK(encoded.optimalLpcOrder == 17 && encoded.quantizedReflectionCoefficients.size() == 17 && encoded.residues.size() == 2048);
                lpc::SampleGenerator sampleGen = lpc::SampleGenerator(encoded);
                data::LpcDecodedData decoded = sampleGen.process();
                CHECK(block.samples.size() == decoded.samples.size() && block.samples == decoded.samples);
                lpc::LinearPredictor predictor(encoded.quantizedReflectionCoefficients, encoded.optimalLpcOrder);
                predictor.dequantizeReflectionCoefficients();
                predictor.generatelinearPredictionCoefficients();
                CHECK(predictor.linearPredictionCoefficients.size() == 18 && predictor.linearPredictionCoefficients[0] == 0);
                // the Rice coder on what the LPC stage left, against the frame coder's subframe of the same signal
                data::RiceEncodedData residueWords = rice::RiceEncoder(data::RiceDecodedData(std::vector<int32_t>(encoded.residues))).process();
                CHECK(residueWords.optimumRiceParam == coded.subFrames[0].residueRiceParam && residueWords.encodedData == coded.subFrames[0].encodedResidues);
                data::RiceEncodedData coefWords = rice::RiceEncoder(data::RiceDecodedData(std::vector<int32_t>(encoded.quantizedReflectionCoefficients))).process();
                CHECK(coefWords.optimumRiceParam == coded.subFrames[0].reflectionCoefficientRiceParam && coefWords.encodedData == coded.subFrames[0].encodedReflectionCoefficients);
            }
            // what the reference's classes take beyond its CLI's shape: samples beyond 16 bits, any length (round 5: the any-length
            // kernels behind the same classes).  A 17-bit mono frame, a 1000-sample stereo frame whose difference wins, a
            // 5000-sample three-channel frame: each comes back exactly, channel lengths and all.
            {
                std::vector<int32_t> wide(2048);
                for (int i = 0; i < 2048; i++)
                    wide[i] = 40000 + (int32_t)(20000 * std::sin(i * 0.01)) + (i * 7919) % 13;
                data::SelaFrame f = frame::FrameEncoder(data::WavFrame(16, { wide })).process();
                CHECK(f.subFrames.size() == 1 && f.subFrames[0].samplesPerChannel == 2048);
                data::WavFrame back = frame::FrameDecoder(f).process();
                CHECK(back.samples.size() == 1 && back.samples[0] == wide);

                std::vector<int32_t> left(1000), right(1000);
                for (int i = 0; i < 1000; i++) {
                    left[i] = (int32_t)(12000 * std::sin(i * 0.05) + 3000 * std::sin(i * 0.31)) + (i * 104729) % 7;
                    right[i] = left[i] - (i % 3);
                }
                f = frame::FrameEncoder(data::WavFrame(16, { left, right })).process();
                CHECK(f.subFrames.size() == 2 && f.subFrames[1].subFrameType == 1 && f.subFrames[1].samplesPerChannel == 1000);
                back = frame::FrameDecoder(f).process();
                CHECK(back.samples.size() == 2 && back.samples[0] == left && back.samples[1] == right);

                std::vector<std::vector<int32_t>> three(3, std::vector<int32_t>(5000));
                for (int c = 0; c < 3; c++)
                    for (int i = 0; i < 5000; i++)
                        three[c][i] = (int32_t)((60000 - 9000 * c) * std::sin(i * (0.003 + 0.002 * c))) + (i * (31 + c)) % 17;
                f = frame::FrameEncoder(data::WavFrame(24, three)).process();
                CHECK(f.subFrames.size() == 3 && f.subFrames[2].samplesPerChannel == 5000 && f.subFrames[2].subFrameType == 0);
                back = frame::FrameDecoder(f).process();
                CHECK(back.samples == three && back.bitsPerSample == 24);

                // the stages on a vector of another length
                data::LpcDecodedData block((uint8_t)16, std::vector<int32_t>(three[1]));
                data::LpcEncodedData enc = lpc::ResidueGenerator(block).process();
                CHECK(enc.residues.size() == 5000 && enc.optimalLpcOrder >= 1 && enc.quantizedReflectionCoefficients.size() == enc.optimalLpcOrder);
                CHECK(lpc::SampleGenerator(enc).process().samples == three[1]);
                lpc::LinearPredictor predictor(enc.quantizedReflectionCoefficients, enc.optimalLpcOrder);
                predictor.dequantizeReflectionCoefficients();
                CHECK(predictor.reflectionCoefficients.size() == std::max<size_t>(1, enc.optimalLpcOrder > 1 ? enc.optimalLpcOrder : 1));
                CHECK(enc.optimalLpcOrder <= 1 || (predictor.reflectionCoefficients[0] >= -1.0 && predictor.reflectionCoefficients[0] <= 1.0));

                // channels of different lengths (round 6; src/frame/frame_encoder.cpp:20-24,73-98): every channel at its own length,
                // the second channel of a stereo frame against the difference over ITS length
                f = frame::FrameEncoder(data::WavFrame(16, { wide, right })).process();
                CHECK(f.subFrames.size() == 2 && f.subFrames[0].samplesPerChannel == 2048 && f.subFrames[1].samplesPerChannel == 1000);
                back = frame::FrameDecoder(f).process();
                CHECK(back.samples.size() == 2 && back.samples[0] == wide && back.samples[1] == right);
                std::vector<int32_t> near(left.begin(), left.begin() + 700);
                for (int i = 0; i < 700; i++)
                    near[i] -= i % 3;
                f = frame::FrameEncoder(data::WavFrame(16, { left, near })).process();
                CHECK(f.subFrames.size() == 2 && f.subFrames[1].subFrameType == 1 && f.subFrames[1].parentChannelNumber == 0 && f.subFrames[1].samplesPerChannel == 700);
                back = frame::FrameDecoder(f).process();
                CHECK(back.samples.size() == 2 && back.samples[0] == left && back.samples[1] == near);
                f = frame::FrameEncoder(data::WavFrame(16, { right, three[0], wide })).process();
                CHECK(f.subFrames.size() == 3 && f.subFrames[0].samplesPerChannel == 1000 && f.subFrames[1].samplesPerChannel == 5000 && f.subFrames[2].samplesPerChannel == 2048);
                back = frame::FrameDecoder(f).process();
                CHECK(back.samples.size() == 3 && back.samples[0] == right && back.samples[1] == three[0] && back.samples[2] == wide);
                // a stereo frame whose FIRST channel is the shorter one, and a block shorter than its order, stay refused (the
                // reference reads past its vectors there)
                bool threw = false;
                try {
                    frame::FrameEncoder(data::WavFrame(16, { left, wide })).process();
                } catch (const data::Exception&) {
                    threw = true;
                }
                CHECK(threw);
                std::vector<int32_t> noise(30);
                uint32_t x = 99u;
                for (int32_t& v : noise) {
                    x = x * 1664525u + 1013904223u;
                    v = (int32_t)(x >> 17) - 16384;
                }
                threw = false;
                try {
                    frame::FrameEncoder(data::WavFrame(16, { noise })).process();
                } catch (const data::Exception&) {
                    threw = true;
                }
                CHECK(threw);
            }
        } catch (const data::Exception& e) {
            std::fprintf(stderr, "FAIL exception: %s\n", e.exceptionMessage.c_str());
            failures++;
        }
        // ---- the reference's file-level classes (src/include/sela/encoder.hpp:9-22, decoder.hpp:9-22) and the three forms of
        // the file verbs -- objects, streams, paths -- write the same bytes
        try {
            const uint32_t ch = 2;
            std::vector<int16_t> pcm((size_t)ch * (2048 * 2300 + 777)); // two and a bit read pieces, a tail that is dropped
            uint32_t x = 777u;
            int v[2] = { 0, 0 };
            for (size_t i = 0; i < pcm.size(); i++) {
                x = x * 1664525u + 1013904223u;
                int& smp = v[i & 1];
                smp += (int)((x >> 21) & 511) - 256;
                smp = smp > 30000 ? 30000 : (smp < -30000 ? -30000 : smp);
                pcm[i] = (int16_t)smp;
            }
            const std::string wav = dir + "/forms.wav";
            {
                file::WavFile w(48000, (uint16_t)ch, std::vector<int16_t>(pcm));
                std::ofstream out(wav, std::ios::binary);
                w.writeToFile(out);
            }
            auto slurp = [](const std::string& path) {
                std::ifstream in(path, std::ios::binary);
                return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            };
            // objects: sela::Encoder(ifstream).process() + SelaFile::writeToFile, as src/main.cpp:29-41 does
            {
                std::ifstream in(wav, std::ios::binary);
                sela::Encoder enc(in);
                file::SelaFile sf = enc.process();
                CHECK(sf.selaHeader.numFrames == 2300 && sf.selaFrames.size() == 2300 && sf.selaHeader.sampleRate == 48000);
                std::ofstream out(dir + "/forms_obj.sela", std::ios::binary);
                sf.writeToFile(out);
            }
            { // streams
                std::ifstream in(wav, std::ios::binary);
                std::ofstream out(dir + "/forms_stream.sela", std::ios::binary);
                CHECK(sela::encodeFile(in, out) == 2300);
            }
            CHECK(sela::encodeFile(wav, dir + "/forms_path.sela") == 2300); // paths
            const std::string a = slurp(dir + "/forms_obj.sela"), b = slurp(dir + "/forms_stream.sela"), c = slurp(dir + "/forms_path.sela");
            CHECK(a.size() > 15 && a == b && a == c);
            {
                std::ifstream in(dir + "/forms_obj.sela", std::ios::binary);
                sela::Decoder dec(in);
                file::WavFile wf = dec.process();
                CHECK(wf.numChannels == ch && wf.sampleRate == 48000 && wf.frameCount() == 2300 && wf.wavFrames.size() == 2300);
                CHECK(wf.wavChunk.formatSubChunk.numChannels == ch && wf.wavChunk.dataSubChunk.samples == wf.pcm.data());
                std::ofstream out(dir + "/forms_obj.wav", std::ios::binary);
                wf.writeToFile(out);
            }
            {
                std::ifstream in(dir + "/forms_obj.sela", std::ios::binary);
                std::ofstream out(dir + "/forms_stream.wav", std::ios::binary);
                CHECK(sela::decodeFile(in, out) == 2300);
            }
            CHECK(sela::decodeFile(dir + "/forms_obj.sela", dir + "/forms_path.wav") == 2300);
            const std::string wa = slurp(dir + "/forms_obj.wav"), wb = slurp(dir + "/forms_stream.wav"), wc = slurp(dir + "/forms_path.wav");
            CHECK(wa.size() == 44 + (size_t)2300 * 2048 * ch * 2 && wa == wb && wa == wc);
            // a hand-made file whose frames say 2048, 700, 3000 and 700 samples (fewer per frame than 2048 on average: a decoder
            // that wrote 2048-sample frames before it found out has written too much): the stream variant of decodeFile must
            // leave exactly what the variant by path and the object path leave
            {
                std::vector<data::SelaFrame> odd;
                const size_t lens[4] = { 2048, 700, 3000, 700 };
                std::vector<int16_t> want;
                for (size_t f = 0; f < 4; f++) {
                    std::vector<std::vector<int32_t>> chans(2, std::vector<int32_t>(lens[f]));
                    for (size_t i = 0; i < lens[f]; i++) {
                        chans[0][i] = (int32_t)(9000 * std::sin((double)(i + 100 * f) * 0.02)) + (int32_t)((i * 7919 + f) % 11);
                        chans[1][i] = chans[0][i] / 2 + (int32_t)(i % 5);
                        want.push_back((int16_t)chans[0][i]);
                        want.push_back((int16_t)chans[1][i]);
                    }
                    odd.push_back(frame::FrameEncoder(data::WavFrame(16, chans)).process());
                }
                file::SelaFile of(44100, 16, 2, std::move(odd));
                {
                    std::ofstream out(dir + "/odd.sela", std::ios::binary);
                    of.writeToFile(out);
                }
                {
                    std::ifstream in(dir + "/odd.sela", std::ios::binary);
                    std::ofstream out(dir + "/odd_stream.wav", std::ios::binary);
                    CHECK(sela::decodeFile(in, out) == 4);
                }
                CHECK(sela::decodeFile(dir + "/odd.sela", dir + "/odd_path.wav") == 4);
                const std::string os = slurp(dir + "/odd_stream.wav"), op = slurp(dir + "/odd_path.wav");
                CHECK(os.size() == 44 + want.size() * 2 && os == op);
                CHECK(os.size() >= 44 && std::memcmp(os.data() + 44, want.data(), std::min(os.size() - 44, want.size() * 2)) == 0);
            }
            size_t differing = 0; // (the codec is the reference's: off by one in a handful of frames at most)
            for (size_t i = 0; i < (size_t)2300 * 2048 * ch; i++)
                differing += std::memcmp(&wa[44 + 2 * i], &pcm[i], 2) != 0;
            CHECK(differing < 4096 * 4);
            // frame::FrameEncoder / FrameDecoder from many threads at once (the reference's own fan-out,
            // src/sela/encoder.cpp:58-73): concurrent calls are coalesced into batches; every result is the bytes the same
            // call gives alone, mono and stereo callers mixed, and a caller with a broken frame fails alone
            {
                const int threads = 24, per = 6;
                std::vector<data::WavFrame> in;
                uint32_t y = 88172645u;
                for (int i = 0; i < threads * per; i++) {
                    const size_t chn = (i / per) % 3 == 2 ? 1 : 2; // every third thread codes mono frames
                    std::vector<std::vector<int32_t>> smp(chn, std::vector<int32_t>(2048));
                    int v[2] = { 0, 0 };
                    for (int j = 0; j < 2048; j++)
                        for (size_t c = 0; c < chn; c++) {
                            y ^= y << 13, y ^= y >> 17, y ^= y << 5;
                            v[c] = std::min(32000, std::max(-32000, v[c] + (int)(y % 2001) - 1000));
                            smp[c][j] = v[c];
                        }
                    in.emplace_back(16, std::move(smp));
                }
                std::vector<std::vector<uint8_t>> alone(in.size()), together(in.size());
                std::vector<data::SelaFrame> coded(in.size(), data::SelaFrame(16));
                for (size_t i = 0; i < in.size(); i++) {
                    coded[i] = frame::FrameEncoder(in[i]).process();
                    frame::appendFrame(coded[i], alone[i]);
                }
                for (int pass = 0; pass < 2; pass++) { // (twice: the second pass meets parked contexts and warm coalescers)
                for (auto& bytes : together)
                    bytes.clear();
                std::vector<int> bad(threads, 0), threw(threads, 0);
                std::vector<std::thread> pool;
                for (int t = 0; t < threads; t++)
                    pool.emplace_back([&, t] {
                        for (int i = t * per; i < (t + 1) * per; i++) {
                            try {
                                const data::SelaFrame f = frame::FrameEncoder(in[i]).process();
                                frame::appendFrame(f, together[i]);
                                data::SelaFrame toDecode = f;
                                if (t == 5 && i == t * per + 2)
                                    toDecode.subFrames[0].channel = 9; // a subframe for a channel the frame does not have
                                const data::WavFrame w = frame::FrameDecoder(toDecode).process();
                                bad[t] += w.samples != in[i].samples;
                            } catch (const data::Exception&) {
                                threw[t]++;
                            }
                        }
                    });
                for (std::thread& th : pool)
                    th.join();
                size_t differing = 0, notLossless = 0, exceptions = 0;
                for (size_t i = 0; i < in.size(); i++)
                    differing += together[i] != alone[i];
                for (int t = 0; t < threads; t++)
                    notLossless += (size_t)bad[t], exceptions += (size_t)threw[t];
                CHECK(differing == 0);
                CHECK(notLossless == 0);
                CHECK(exceptions == 1 && threw[5] == 1); // the broken frame's caller, nobody else
                }
            }
            // the player on the same file: every frame of the decoded file, in order, as packets of one frame; a sink that
            // fails in the middle surfaces as the reference's kind of exception, and the next job runs as if nothing had happened
            struct Checking : sela::AudioSink {
                const std::string* wav;
                size_t at = 44, packets = 0, failAt = (size_t)-1;
                bool same = true;
                int opened = 0, closed = 0;
                uint32_t rate = 0;
                void open(const data::WavFormatSubChunk& f) override { opened++, rate = f.sampleRate; }
                void play(const data::AudioPacket& p) override
                {
                    if (packets == failAt)
                        throw data::Exception("the device went away");
                    same = same && at + p.bufferSize <= wav->size() && std::memcmp(p.audio, wav->data() + at, p.bufferSize) == 0;
                    at += p.bufferSize, packets++;
                }
                void close() override { closed++; }
            };
            {
                Checking sink;
                sink.wav = &wa;
                sela::Player player(sink);
                CHECK(player.playFile(dir + "/forms_obj.sela") == 2300);
                CHECK(sink.same && sink.packets == 2300 && sink.at == wa.size() && sink.opened == 1 && sink.closed == 1 && sink.rate == 48000);
                CHECK(player.firstPacketSeconds > 0 && player.packetsPlayed == 2300);
            }
            {
                Checking sink;
                sink.wav = &wa;
                sink.failAt = 700;
                sela::Player player(sink);
                std::string what;
                try {
                    player.playFile(dir + "/forms_obj.sela");
                } catch (const data::Exception& e) {
                    what = e.exceptionMessage;
                }
                CHECK(what == "the device went away" && sink.packets == 700 && sink.same && sink.closed == 1);
                sink.failAt = (size_t)-1, sink.at = 44, sink.packets = 0;
                CHECK(player.playFile(dir + "/forms_obj.sela") == 2300 && sink.same && sink.at == wa.size());
            }
        } catch (const data::Exception& e) {
            std::fprintf(stderr, "FAIL exception: %s\n", e.exceptionMessage.c_str());
            failures++;
        }
        // host-pointer API timing on a 3-minute stereo track of pseudo-random-walk PCM (informational)
        {
            const uint32_t frames = 3875, ch = 2;
            std::vector<int16_t> pcm((size_t)frames * 2048 * ch);
            uint32_t x = 12345u;
            int v[2] = { 0, 0 };
            for (size_t i = 0; i < pcm.size(); i++) {
                x = x * 1664525u + 1013904223u;
                int& s = v[i & 1];
                s += (int)((x >> 20) & 1023) - 512;
                s = s > 30000 ? 30000 : (s < -30000 ? -30000 : s);
                pcm[i] = (int16_t)s;
            }
            std::vector<uint8_t> bytes(sela_hip_encode_bound_bytes(frames, ch));
            std::vector<uint64_t> offs(frames + 1);
            std::vector<int16_t> back(pcm.size());
            double enc_ms = 0, dec_ms = 0;
            for (int rep = 0; rep < 4; rep++) {
                auto t0 = std::chrono::steady_clock::now();
                CHECK(sela_hip_encode(pcm.data(), frames, ch, 2048, bytes.data(), bytes.size(), offs.data()) == SELA_HIP_OK);
                auto t1 = std::chrono::steady_clock::now();
                CHECK(sela_hip_decode(bytes.data(), offs.data(), frames, ch, back.data()) == SELA_HIP_OK);
                auto t2 = std::chrono::steady_clock::now();
                if (rep) {
                    enc_ms += std::chrono::duration<double, std::milli>(t1 - t0).count() / 3;
                    dec_ms += std::chrono::duration<double, std::milli>(t2 - t1).count() / 3;
                }
            }
            size_t diff = 0;
            for (size_t i = 0; i < pcm.size(); i++)
                diff += pcm[i] != back[i];
            CHECK(diff < pcm.size() / 100);
            std::printf("host-pointer API, 3875 stereo frames: encode %.2f ms, decode %.2f ms (PCIe inclusive), %zu bytes\n", enc_ms, dec_ms,
                (size_t)offs[frames]);
        }
    }
    std::printf(failures ? "selftest: %d failure(s)\n" : "selftest: ok\n", failures);
    return failures ? 1 : 0;
}
