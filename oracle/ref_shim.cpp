// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper (our code) around the *unmodified* reference classes, compiled
// by oracle/Makefile from the sources where they lie under /root/reference into
// oracle/_ref/libsela_ref.so.  Nothing from the reference is copied into this repo: this
// file only #includes the reference's public headers through -I$(REF)/src/include.
//
// It exists to (1) pin the CPU restatement in sela_oracle.c against the real thing,
// (2) generate the golden fixtures under tests/golden/, and (3) serve as the
// "reference"-kind CPU baseline in bench.py (threads fan out exactly like
// src/sela/encoder.cpp:58-73).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include <fstream>

#include "data/exception.hpp"
#include "frame.hpp"
#include "lpc.hpp"
#include "rice.hpp"
#include "sela/decoder.hpp"
#include "sela/encoder.hpp"

namespace {

// Serialise one data::SelaFrame exactly like file::SelaFile::writeToFile does for a frame
// (src/file/sela_file.cpp:115-135); returns bytes written.
size_t put_frame(const data::SelaFrame& f, uint8_t* out)
{
    uint8_t* p = out;
    const uint32_t sync = (uint32_t)f.syncWord;
    std::memcpy(p, &sync, 4), p += 4;
    for (const data::SelaSubFrame& s : f.subFrames) {
        *p++ = s.channel;
        *p++ = s.subFrameType;
        *p++ = s.parentChannelNumber;
        *p++ = s.reflectionCoefficientRiceParam;
        std::memcpy(p, &s.reflectionCoefficientRequiredInts, 2), p += 2;
        *p++ = s.optimumLpcOrder;
        std::memcpy(p, s.encodedReflectionCoefficients.data(), 4 * s.encodedReflectionCoefficients.size());
        p += 4 * s.encodedReflectionCoefficients.size();
        *p++ = s.residueRiceParam;
        std::memcpy(p, &s.residueRequiredInts, 2), p += 2;
        std::memcpy(p, &s.samplesPerChannel, 2), p += 2;
        std::memcpy(p, s.encodedResidues.data(), 4 * s.encodedResidues.size());
        p += 4 * s.encodedResidues.size();
    }
    return (size_t)(p - out);
}

// Parse one frame the way file::SelaFile::readFromFile does (src/file/sela_file.cpp:44-99).
data::SelaFrame get_frame(const uint8_t* in, uint32_t channels, size_t* used)
{
    const uint8_t* p = in + 4;
    data::SelaFrame frame((uint8_t)16);
    for (uint32_t c = 0; c < channels; c++) {
        uint8_t ch = p[0], type = p[1], parent = p[2];
        uint8_t rk = p[3];
        uint16_t rw;
        std::memcpy(&rw, p + 4, 2);
        uint8_t order = p[6];
        p += 7;
        std::vector<uint32_t> cw(rw);
        std::memcpy(cw.data(), p, 4 * (size_t)rw), p += 4 * (size_t)rw;
        uint8_t ek = p[0];
        uint16_t ew, n;
        std::memcpy(&ew, p + 1, 2);
        std::memcpy(&n, p + 3, 2);
        p += 5;
        std::vector<uint32_t> rwv(ew);
        std::memcpy(rwv.data(), p, 4 * (size_t)ew), p += 4 * (size_t)ew;
        data::RiceEncodedData refl(rk, order, std::move(cw));
        data::RiceEncodedData resi(ek, n, std::move(rwv));
        frame.subFrames.push_back(data::SelaSubFrame(ch, type, parent, refl, resi));
    }
    *used = (size_t)(p - in);
    return frame;
}

data::WavFrame make_wav_frame(const int16_t* pcm, uint32_t channels, uint32_t n)
{
    std::vector<std::vector<int32_t>> s(channels, std::vector<int32_t>(n));
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < channels; c++)
            s[c][i] = pcm[(size_t)i * channels + c];
    return data::WavFrame((uint8_t)16, std::move(s));
}

} // namespace

extern "C" {

// lpc::ResidueGenerator (src/lpc/residue_generator.cpp:121-134).  q must hold 100 ints.
int ref_lpc_analyze(const int32_t* samples, int n, int32_t* q, int32_t* residues)
{
    data::LpcDecodedData in((uint8_t)16, std::vector<int32_t>(samples, samples + n));
    data::LpcEncodedData enc = lpc::ResidueGenerator(in).process();
    std::memcpy(q, enc.quantizedReflectionCoefficients.data(), 4 * enc.quantizedReflectionCoefficients.size());
    std::memcpy(residues, enc.residues.data(), 4 * enc.residues.size());
    return enc.optimalLpcOrder;
}

// lpc::LinearPredictor dequantise + step-up (src/lpc/linear_predictor.cpp:16-61). a holds order+1.
void ref_lpc_coeffs(int order, const int32_t* q, int64_t* a)
{
    lpc::LinearPredictor lp(std::vector<int32_t>(q, q + order), (uint8_t)order);
    lp.dequantizeReflectionCoefficients();
    lp.generatelinearPredictionCoefficients();
    std::memcpy(a, lp.linearPredictionCoefficients.data(), 8 * lp.linearPredictionCoefficients.size());
}

// lpc::SampleGenerator (src/lpc/sample_generator.cpp:32-39).
void ref_lpc_synth(int order, const int32_t* q, const int32_t* residues, int n, int32_t* samples)
{
    data::LpcEncodedData enc((uint8_t)order, (uint8_t)16, std::vector<int32_t>(q, q + order),
        std::vector<int32_t>(residues, residues + n));
    data::LpcDecodedData dec = lpc::SampleGenerator(enc).process();
    std::memcpy(samples, dec.samples.data(), 4 * dec.samples.size());
}

// rice::RiceEncoder (src/rice/rice_encoder.cpp:73-81). Returns word count, *k = parameter.
int ref_rice_encode(const int32_t* in, int n, uint32_t* k, uint32_t* words, int cap)
{
    data::RiceDecodedData d(std::vector<int32_t>(in, in + n));
    data::RiceEncodedData e = rice::RiceEncoder(d).process();
    *k = e.optimumRiceParam;
    if ((int)e.encodedData.size() > cap)
        return -1;
    std::memcpy(words, e.encodedData.data(), 4 * e.encodedData.size());
    return (int)e.encodedData.size();
}

// rice::RiceDecoder (src/rice/rice_decoder.cpp:54-61).
void ref_rice_decode(const uint32_t* words, int nwords, int n, uint32_t k, int32_t* out)
{
    data::RiceEncodedData e((int32_t)k, n, std::vector<uint32_t>(words, words + nwords));
    data::RiceDecodedData d = rice::RiceDecoder(e).process();
    std::memcpy(out, d.decodedData.data(), 4 * d.decodedData.size());
}

// frame::FrameEncoder over interleaved int16 PCM (src/frame/frame_encoder.cpp:11-102);
// output = on-disk frame bytes.  Returns bytes written.
size_t ref_frame_encode(const int16_t* pcm, uint32_t channels, uint32_t n, uint8_t* out)
{
    data::WavFrame wf = make_wav_frame(pcm, channels, n);
    data::SelaFrame sf = frame::FrameEncoder(wf).process();
    return put_frame(sf, out);
}

// frame::FrameDecoder (src/frame/frame_decoder.cpp:11-72) from on-disk frame bytes to
// interleaved int16 (truncating like wav_file.cpp:248-251).  Returns bytes consumed.
size_t ref_frame_decode(const uint8_t* in, uint32_t channels, int16_t* pcm)
{
    size_t used = 0;
    data::SelaFrame sf = get_frame(in, channels, &used);
    data::WavFrame wf = frame::FrameDecoder(sf).process();
    const size_t n = wf.samples[0].size();
    for (size_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < channels; c++)
            pcm[i * channels + c] = (int16_t)(uint16_t)wf.samples[c][i];
    return used;
}

// The same two on the reference's own value type (data::WavFrame: int32 samples per channel, any number of them,
// src/include/data/wav_frame.hpp:8-16).  planar = [channels][n].
size_t ref_frame_encode_i32(const int32_t* planar, uint32_t channels, uint32_t n, uint8_t* out)
{
    std::vector<std::vector<int32_t>> s(channels);
    for (uint32_t c = 0; c < channels; c++)
        s[c].assign(planar + (size_t)c * n, planar + (size_t)(c + 1) * n);
    data::WavFrame wf((uint8_t)16, std::move(s));
    data::SelaFrame sf = frame::FrameEncoder(wf).process();
    return put_frame(sf, out);
}

// ... and on a data::WavFrame whose channels differ in length (samples = the channels back to back).  The caller keeps to frames
// on which the reference is defined: an exactly-stereo frame with channel 0 the shorter indexes past its vector
// (src/frame/frame_encoder.cpp:22-24).
size_t ref_frame_encode_ragged(const int32_t* samples, const uint32_t* lengths, uint32_t channels, uint8_t* out)
{
    std::vector<std::vector<int32_t>> s(channels);
    const int32_t* cur = samples;
    for (uint32_t c = 0; c < channels; c++) {
        s[c].assign(cur, cur + lengths[c]);
        cur += lengths[c];
    }
    data::WavFrame wf((uint8_t)16, std::move(s));
    data::SelaFrame sf = frame::FrameEncoder(wf).process();
    return put_frame(sf, out);
}

// out[c][0 .. counts[c]) = WavFrame.samples[c] exactly as frame::FrameDecoder::process returns them (32-bit, every
// channel with its own length); out is [channels][stride].  Returns bytes consumed.
size_t ref_frame_decode_i32(const uint8_t* in, uint32_t channels, int32_t* out, uint32_t stride, uint32_t* counts)
{
    size_t used = 0;
    data::SelaFrame sf = get_frame(in, channels, &used);
    data::WavFrame wf = frame::FrameDecoder(sf).process();
    for (uint32_t c = 0; c < channels; c++) {
        const size_t n = c < wf.samples.size() ? wf.samples[c].size() : 0;
        counts[c] = (uint32_t)n;
        std::memcpy(out + (size_t)c * stride, wf.samples[c].data(), 4 * (n < stride ? n : stride));
    }
    return used;
}

// Batch encode with the reference's thread fan-out (src/sela/encoder.cpp:40-92): T threads,
// static contiguous ranges, last thread takes the remainder.  offsets[n_frames+1] receives
// byte offsets of each frame in out.  Returns seconds spent in the fan-out (frames are
// converted to data::WavFrame before the clock starts, serialisation happens after it stops).
double ref_encode_frames_mt(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint32_t n,
    uint32_t threads, uint8_t* out, uint64_t* offsets)
{
    std::vector<data::WavFrame> wav;
    wav.reserve(n_frames);
    for (uint32_t f = 0; f < n_frames; f++)
        wav.push_back(make_wav_frame(pcm + (size_t)f * n * channels, channels, n));
    if (threads == 0)
        threads = 1;
    std::vector<std::vector<data::SelaFrame>> seg(threads);
    const size_t per = n_frames / threads;
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < threads; t++) {
            const size_t b = t * per;
            const size_t e = (t == threads - 1) ? n_frames : b + per;
            pool.emplace_back([&wav, &seg, t, b, e]() {
                seg[t].reserve(e - b);
                for (size_t i = b; i < e; i++)
                    seg[t].push_back(frame::FrameEncoder(wav[i]).process());
            });
        }
        for (auto& th : pool)
            th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    uint64_t off = 0;
    uint32_t f = 0;
    for (auto& s : seg)
        for (auto& fr : s) {
            offsets[f++] = off;
            off += put_frame(fr, out + off);
        }
    offsets[f] = off;
    return std::chrono::duration<double>(t1 - t0).count();
}

// Batch decode, same fan-out (src/sela/decoder.cpp:41-92).  Returns seconds in the fan-out.
double ref_decode_frames_mt(const uint8_t* in, const uint64_t* offsets, uint32_t n_frames, uint32_t channels,
    uint32_t n, uint32_t threads, int16_t* pcm)
{
    std::vector<data::SelaFrame> sela;
    sela.reserve(n_frames);
    for (uint32_t f = 0; f < n_frames; f++) {
        size_t used;
        sela.push_back(get_frame(in + offsets[f], channels, &used));
    }
    if (threads == 0)
        threads = 1;
    std::vector<std::vector<data::WavFrame>> seg(threads);
    const size_t per = n_frames / threads;
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < threads; t++) {
            const size_t b = t * per;
            const size_t e = (t == threads - 1) ? n_frames : b + per;
            pool.emplace_back([&sela, &seg, t, b, e]() {
                seg[t].reserve(e - b);
                for (size_t i = b; i < e; i++)
                    seg[t].push_back(frame::FrameDecoder(sela[i]).process());
            });
        }
        for (auto& th : pool)
            th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    size_t f = 0;
    for (auto& s : seg)
        for (auto& wf : s) {
            int16_t* o = pcm + f * (size_t)n * channels;
            for (size_t i = 0; i < n; i++)
                for (uint32_t c = 0; c < channels; c++)
                    o[i * channels + c] = (int16_t)(uint16_t)wf.samples[c][i];
            f++;
        }
    return std::chrono::duration<double>(t1 - t0).count();
}

// The reference's own file-to-file verbs (src/main.cpp:29-41): sela::Encoder + SelaFile::writeToFile
// (src/sela/encoder.cpp:94-100, src/file/sela_file.cpp:105-137) and sela::Decoder + WavFile::writeToFile
// (src/sela/decoder.cpp, src/file/wav_file.cpp:222-267).  Return 0, or 1 when the reference throws.
int ref_encode_file(const char* wav_path, const char* sela_path)
{
    try {
        std::ifstream in(wav_path, std::ios::binary);
        std::ofstream out(sela_path, std::ios::binary);
        sela::Encoder encoder(in);
        file::SelaFile sela_file = encoder.process();
        sela_file.writeToFile(out);
    } catch (data::Exception&) {
        return 1;
    }
    return 0;
}

int ref_decode_file(const char* sela_path, const char* wav_path)
{
    try {
        std::ifstream in(sela_path, std::ios::binary);
        std::ofstream out(wav_path, std::ios::binary);
        sela::Decoder decoder(in);
        file::WavFile wav_file = decoder.process();
        wav_file.writeToFile(out);
    } catch (data::Exception&) {
        return 1;
    }
    return 0;
}

uint32_t ref_hardware_concurrency(void) { return std::thread::hardware_concurrency(); }

} // extern "C"
