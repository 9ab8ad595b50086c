// tests/c/coalescer_stress.cpp -- the call coalescer of libsela_hip.so (sela_amd/csrc/sela_coalescer.h) on a CPU stub
// backend, built with -fsanitize=thread by tests/test_sanitizers.py: many threads submit one- to four-frame calls of two
// channel counts, every call must get back ITS OWN result and error (a too-small output buffer, a "malformed" frame that
// fails a whole batch and is then retried call by call), and the thread sanitizer must see no race in the hand-overs
// (queue, leader election, results written by the leader and read by the callers).
//
// TEST INFRASTRUCTURE: the stub stands in for the device; nothing here is compiled into the library.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "sela_coalescer.h"

namespace {

constexpr uint32_t kFrame = SELA_HIP_SAMPLES_PER_FRAME;
std::atomic<int> g_jobs{ 0 }, g_frames{ 0 }, g_leaders_done{ 0 };
thread_local std::string t_error;

// "encoding" = 8 bytes per frame: the frame's first sample and a checksum of the rest; "decoding" = the inverse filling
// the frame with the first sample.  A frame whose first sample is -32768 is "malformed" (decode only).
struct StubBackend {
    static size_t encode_bound_bytes(uint32_t n_frames, uint32_t) { return (size_t)n_frames * 8; }
    static int encode_now(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint8_t* out, size_t cap, uint64_t* offsets)
    {
        g_jobs++, g_frames += (int)n_frames;
        if (cap < (size_t)n_frames * 8) {
            t_error = "frames_out too small";
            return SELA_HIP_ECAPACITY;
        }
        for (uint32_t f = 0; f < n_frames; f++) {
            const int16_t* p = pcm + (size_t)f * kFrame * channels;
            uint32_t sum = 0;
            for (size_t i = 0; i < (size_t)kFrame * channels; i++)
                sum = sum * 31u + (uint16_t)p[i];
            const uint32_t first = (uint16_t)p[0] | (channels << 16);
            std::memcpy(out + 8 * f, &first, 4);
            std::memcpy(out + 8 * f + 4, &sum, 4);
            offsets[f] = 8 * (uint64_t)f;
        }
        offsets[n_frames] = 8 * (uint64_t)n_frames;
        std::this_thread::sleep_for(std::chrono::microseconds(30)); // (a trip to the device)
        return SELA_HIP_OK;
    }
    static int decode_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm)
    {
        g_jobs++, g_frames += (int)n_frames;
        for (uint32_t f = 0; f < n_frames; f++) {
            uint32_t first;
            std::memcpy(&first, frames + offsets[f], 4);
            if ((int16_t)(uint16_t)first == -32768) {
                t_error = "malformed frame";
                return SELA_HIP_EFORMAT;
            }
            for (size_t i = 0; i < (size_t)kFrame * channels; i++)
                pcm[(size_t)f * kFrame * channels + i] = (int16_t)(uint16_t)first;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(30));
        return SELA_HIP_OK;
    }
    // "decoding to 32-bit channels": channel c of a frame gets 100 + 10 c samples, all the frame's first sample + c
    static int decode_i32_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int32_t* samples, uint32_t stride,
        uint32_t* counts)
    {
        g_jobs++, g_frames += (int)n_frames;
        for (uint32_t f = 0; f < n_frames; f++) {
            uint32_t first;
            std::memcpy(&first, frames + offsets[f], 4);
            if ((int16_t)(uint16_t)first == -32768) {
                t_error = "malformed frame";
                return SELA_HIP_EFORMAT;
            }
            for (uint32_t c = 0; c < channels; c++) {
                const uint32_t cnt = 100 + 10 * c;
                if (cnt > stride) {
                    t_error = "stride too small";
                    return SELA_HIP_ECAPACITY;
                }
                counts[f * channels + c] = cnt;
                for (uint32_t i = 0; i < cnt; i++)
                    samples[((size_t)f * channels + c) * stride + i] = (int32_t)(int16_t)(uint16_t)first + (int32_t)c;
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(30));
        return SELA_HIP_OK;
    }
    // "encoding 32-bit channels": 8 bytes per frame -- its first sample | channels << 16 | n << 24, and a checksum
    static size_t encode_i32_bound_bytes(uint32_t n_frames, uint32_t, uint32_t) { return (size_t)n_frames * 8; }
    static int encode_i32_now(const int32_t* samples, uint32_t n_frames, uint32_t channels, uint32_t n, uint8_t* out, size_t cap, uint64_t* offsets)
    {
        g_jobs++, g_frames += (int)n_frames;
        if (cap < (size_t)n_frames * 8) {
            t_error = "frames_out too small";
            return SELA_HIP_ECAPACITY;
        }
        for (uint32_t f = 0; f < n_frames; f++) {
            const int32_t* p = samples + (size_t)f * n * channels;
            if (p[0] == -777) {
                t_error = "a block the reference cannot answer";
                return SELA_HIP_ERANGE;
            }
            uint32_t sum = 0;
            for (size_t i = 0; i < (size_t)n * channels; i++)
                sum = sum * 31u + (uint32_t)p[i];
            const uint32_t first = ((uint32_t)p[0] & 0xFFFFu) | (channels << 16) | (n << 24);
            std::memcpy(out + 8 * f, &first, 4);
            std::memcpy(out + 8 * f + 4, &sum, 4);
            offsets[f] = 8 * (uint64_t)f;
        }
        offsets[n_frames] = 8 * (uint64_t)n_frames;
        std::this_thread::sleep_for(std::chrono::microseconds(30));
        return SELA_HIP_OK;
    }
    static void* take(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
    static void give(void* p) { std::free(p); }
    static std::string last_error() { return t_error; }
    static void after_batch() { g_leaders_done++; }
};

typedef sela::CallCoalescer<StubBackend> Coalescer;

int worker(Coalescer& enc, Coalescer& dec, Coalescer& dec32, Coalescer& enc32, int id, int rounds, std::atomic<int>& failures)
{
    std::mt19937 rng(1000 + id);
    for (int r = 0; r < rounds; r++) {
        const uint32_t channels = (id & 1) ? 2 : 1, n = 1 + rng() % 4;
        std::vector<int16_t> pcm((size_t)n * kFrame * channels);
        for (uint32_t f = 0; f < n; f++) {
            const int16_t v = (int16_t)(id * 100 + r * 7 + (int)f);
            for (size_t i = 0; i < (size_t)kFrame * channels; i++)
                pcm[(size_t)f * kFrame * channels + i] = v;
        }
        // encode; every 11th call with a buffer that is too small: that caller's error, nobody else's
        const bool small = r % 11 == 5;
        std::vector<uint8_t> bytes(small ? 4 : (size_t)n * 8);
        std::vector<uint64_t> offsets(n + 1, ~0ull);
        sela::SmallCall e;
        e.device = id % 3 == 0 ? 1 : 0, e.channels = channels, e.n_frames = n;
        e.pcm = pcm.data(), e.frames_out = bytes.data(), e.frames_cap = bytes.size(), e.offsets_out = offsets.data();
        const int rc = enc.submit(e);
        if (small) {
            if (rc != SELA_HIP_ECAPACITY || e.error.empty())
                failures++, std::fprintf(stderr, "thread %d round %d: a too-small buffer gave rc %d\n", id, r, rc);
            continue;
        }
        if (rc != SELA_HIP_OK || offsets[n] != 8ull * n) {
            failures++, std::fprintf(stderr, "thread %d round %d: encode rc %d\n", id, r, rc);
            continue;
        }
        for (uint32_t f = 0; f < n; f++) {
            uint32_t first;
            std::memcpy(&first, bytes.data() + offsets[f], 4);
            if ((int16_t)(uint16_t)first != pcm[(size_t)f * kFrame * channels] || (first >> 16) != channels)
                failures++, std::fprintf(stderr, "thread %d round %d: frame %u came back as somebody else's\n", id, r, f);
        }
        // decode; every 13th call carries a "malformed" frame: the batch fails, is retried call by call, only this caller hears of it
        const bool bad = r % 13 == 7;
        if (bad) {
            const uint32_t poison = 0x8000u | (channels << 16);
            std::memcpy(bytes.data() + offsets[n - 1], &poison, 4);
        }
        std::vector<int16_t> back(pcm.size(), 12345);
        sela::SmallCall d;
        d.device = e.device, d.channels = channels, d.n_frames = n;
        d.frames = bytes.data(), d.offsets_in = offsets.data(), d.pcm_out = back.data();
        const int rd = dec.submit(d);
        if (bad ? rd != SELA_HIP_EFORMAT : (rd != SELA_HIP_OK || back != pcm))
            failures++, std::fprintf(stderr, "thread %d round %d: decode rc %d (bad frame: %d)\n", id, r, rd, (int)bad);
        // the same frames to 32-bit channels, every caller with a stride of its own; every 17th call with one that is too small
        const bool narrow = !bad && r % 17 == 3;
        const uint32_t stride = narrow ? 50u : 128u + (uint32_t)(id % 5) * 7u;
        std::vector<int32_t> wide((size_t)n * channels * stride, -7);
        std::vector<uint32_t> counts((size_t)n * channels, 999);
        sela::SmallCall w;
        w.device = e.device, w.channels = channels, w.n_frames = n;
        w.frames = bytes.data(), w.offsets_in = offsets.data(), w.samples_out = wide.data(), w.stride = stride, w.counts_out = counts.data();
        const int rw = dec32.submit(w);
        if (bad ? rw != SELA_HIP_EFORMAT : (narrow ? rw != SELA_HIP_ECAPACITY : rw != SELA_HIP_OK)) {
            failures++, std::fprintf(stderr, "thread %d round %d: decode32 rc %d (bad %d narrow %d)\n", id, r, rw, (int)bad, (int)narrow);
        } else if (!bad && !narrow) {
            for (uint32_t f = 0; f < n; f++)
                for (uint32_t c = 0; c < channels; c++) {
                    const size_t row = (size_t)f * channels + c;
                    bool ok = counts[row] == 100 + 10 * c;
                    for (uint32_t i = 0; ok && i < counts[row]; i++)
                        ok = wide[row * stride + i] == (int32_t)pcm[(size_t)f * kFrame * channels] + (int32_t)c;
                    if (!ok)
                        failures++, std::fprintf(stderr, "thread %d round %d: decode32 row %zu came back as somebody else's\n", id, r, row);
                }
        }
        // 32-bit channels of a shape of the thread's own (three shapes among the threads: only equal shapes may share a job); every
        // 19th call holds a block "the reference cannot answer": the batch is retried call by call, only this caller hears of it
        const uint32_t shape = 40u + 10u * (uint32_t)(id % 3);
        const bool refused = r % 19 == 4;
        std::vector<int32_t> planar((size_t)n * channels * shape);
        for (uint32_t f = 0; f < n; f++)
            for (size_t i = 0; i < (size_t)shape * channels; i++)
                planar[(size_t)f * shape * channels + i] = (refused && f == n - 1) ? -777 : (int32_t)(id * 1000 + r * 3 + (int)f);
        std::vector<uint8_t> bytes32((size_t)n * 8);
        std::vector<uint64_t> offsets32(n + 1, ~0ull);
        sela::SmallCall x;
        x.device = e.device, x.channels = channels, x.n_frames = n, x.shape = shape;
        x.samples = planar.data(), x.frames_out = bytes32.data(), x.frames_cap = bytes32.size(), x.offsets_out = offsets32.data();
        const int rx = enc32.submit(x);
        if (refused ? rx != SELA_HIP_ERANGE : (rx != SELA_HIP_OK || offsets32[n] != 8ull * n)) {
            failures++, std::fprintf(stderr, "thread %d round %d: encode32 rc %d (refused: %d)\n", id, r, rx, (int)refused);
        } else if (!refused) {
            for (uint32_t f = 0; f < n; f++) {
                uint32_t first;
                std::memcpy(&first, bytes32.data() + offsets32[f], 4);
                if ((first & 0xFFFFu) != ((uint32_t)planar[(size_t)f * shape * channels] & 0xFFFFu) || ((first >> 16) & 0xFF) != channels || (first >> 24) != shape)
                    failures++, std::fprintf(stderr, "thread %d round %d: encode32 frame %u came back as somebody else's\n", id, r, f);
            }
        }
    }
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    const int threads = argc > 1 ? std::atoi(argv[1]) : 16, rounds = argc > 2 ? std::atoi(argv[2]) : 120;
    std::atomic<int> failures{ 0 };
    for (int seats : { sela::kCoalesceLeaders, 1, 3 }) { // the shipped number of batches in flight, the strictly serial form, one more
        Coalescer enc(true, seats), dec(false, seats), dec32(Coalescer::kDecode32, seats), enc32(Coalescer::kEncode32, seats);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back(worker, std::ref(enc), std::ref(dec), std::ref(dec32), std::ref(enc32), t, rounds, std::ref(failures));
        for (std::thread& t : pool)
            t.join();
        std::printf("%d seats: %d threads x %d rounds: %d device jobs for %d frames so far, %d batches led, %d failures\n", seats, threads, rounds, g_jobs.load(),
            g_frames.load(), g_leaders_done.load(), failures.load());
    }
    return failures.load() ? 1 : 0;
}
