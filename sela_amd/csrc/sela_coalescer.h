// sela_coalescer.h -- one-frame calls from many threads become device jobs (the host-pointer API of libsela_hip.so).
//
// Header-only and written against a BACKEND (what runs a job, where staging memory comes from), so that the locking can be
// built on its own: libsela_hip.so instantiates it with the HIP backend (sela_capi.hip), tests/c/coalescer_stress.cpp with a
// CPU stub under -fsanitize=thread (tests/test_sanitizers.py).  Nothing of the stub is compiled into the library.
//
//   struct Backend {
//       static int encode_now(const int16_t* pcm, uint32_t n_frames, uint32_t channels, uint8_t* frames_out, size_t frames_cap, uint64_t* offsets_out);
//       static int decode_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int16_t* pcm_out);
//       static int decode_i32_now(const uint8_t* frames, const uint64_t* offsets, uint32_t n_frames, uint32_t channels, int32_t* samples_out,
//                                 uint32_t stride, uint32_t* counts_out);   // sela_hip_decode_i32: [frame][channel][stride] + a count per row
//       static int encode_i32_now(const int32_t* samples, uint32_t n_frames, uint32_t channels, uint32_t n, uint8_t* frames_out, size_t frames_cap,
//                                 uint64_t* offsets_out);                      // sela_hip_encode_i32: [frame][channel][n]
//       static size_t encode_bound_bytes(uint32_t n_frames, uint32_t channels);
//       static size_t encode_i32_bound_bytes(uint32_t n_frames, uint32_t channels, uint32_t n);
//       static void* take(size_t bytes);  static void give(void* p);      // staging memory of a batch
//       static std::string last_error();                                  // of the calling thread's last *_now
//       static void after_batch();                                        // the leader is through with the device
//   };
#ifndef SELA_COALESCER_H_
#define SELA_COALESCER_H_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sela_hip.h"

namespace sela {

// The reference hands its frames to hardware_concurrency() threads, one frame per call (src/sela/encoder.cpp:58-73,
// src/sela/decoder.cpp:58-73), and a binding that keeps that loop calls sela_hip_encode / sela_hip_decode the same way.
// One frame is a poor launch (3 of the device's 3072 block slots), and every calling thread would want streams and
// buffers of its own.  So one-shot calls of at most kCoalesceFrames frames group the way databases group commits: a call
// that finds nobody ahead of it runs at once, as it is; calls that arrive while it is on the device queue up, and when
// it returns ONE of them takes everything that is waiting for the same device and channel count to the device as a
// single job, hands every call its part of the result, and parks the streams it used for the next leader.  A lone caller
// pays nothing; T busy threads end up in batches of about T / kCoalesceLeaders calls.  A call's own failure (output buffer
// too small, a malformed frame) stays its own.
//
// Up to kCoalesceLeaders batches are on the device at a time (round 6; one until then): a batch of a few dozen frames is a
// trip of ~0.2 ms that leaves the device and the link nearly idle, and with one batch in flight a call waits out the batch
// in flight, then rides the next.  Calls still pile up only while every leader's seat is taken, so the batches stay batches:
// T threads settle into kCoalesceLeaders batches of T / kCoalesceLeaders calls that overlap on the device, each on the
// streams and buffers its leader leases.  Measured with the reference's thread loop over the frame classes (host/sela_filebench
// frames T 256, 2048-sample 16-bit stereo frames; encode / decode M samples/s), together with the per-call wake-up
// (SmallCall::tell): T = 64: 376 / 214 -> 437-496 / 246-265; T = 256 (the GPU box's hardware_concurrency(), what the
// reference starts): 111 / 111 -> 650-930 / 550-565 with two seats for either direction.  By direction (four runs each,
// T = 256):
// encode 650-930 / 590-840 / 470-550 M samples/s with 2 / 3 / 4 seats (its jobs read the caller's samples from host memory
// inside one launch, sela_capi.hip: more of them at once get in each other's way), decode 550-565 / 640-670 / 630-670 -- two
// seats for the encoders, three for the decoders (1000-sample 17-bit frames, the any-length kinds: 320-430 / 300-440 encode,
// 420-425 / 435-460 decode with 2 / 3).
constexpr uint32_t kCoalesceFrames = 32;
constexpr int kCoalesceLeaders = 2, kCoalesceLeadersDecode = 3;

struct SmallCall {
    int device = 0;
    uint32_t channels = 0, n_frames = 0;
    const int16_t* pcm = nullptr; // encode
    uint8_t* frames_out = nullptr;
    size_t frames_cap = 0;
    uint64_t* offsets_out = nullptr;
    const uint8_t* frames = nullptr; // decode
    const uint64_t* offsets_in = nullptr;
    int16_t* pcm_out = nullptr;
    int32_t* samples_out = nullptr; // decode to 32-bit channels (sela_hip_decode_i32): [n_frames][channels][stride], counts_out[n_frames * channels]
    uint32_t stride = 0;
    uint32_t* counts_out = nullptr;
    const int32_t* samples = nullptr; // encode from 32-bit channels (sela_hip_encode_i32): [n_frames][channels][shape], shape = samples per channel
    uint32_t shape = 0;               // (calls only share a batch with calls of the same shape; 0 for the other kinds)
    int rc = SELA_HIP_OK;
    std::string error;
    // What a decoded batch's leader leaves to the calling threads (round 6): the copies out of the batch's staging buffer into the
    // call's own memory -- 16 KB per stereo frame for the 32-bit kinds, a third of the leader's time on the device's behalf when
    // it made them all itself, one call after the other, before it gave up its seat.  The buffer lives until the last call of the
    // batch has let go of it.
    struct Piece {
        void* to;
        const void* from;
        size_t bytes;
    };
    std::vector<Piece> pieces;
    std::shared_ptr<void> pieces_from;
    void collect()
    {
        for (const Piece& p : pieces)
            std::memcpy(p.to, p.from, p.bytes);
        pieces.clear();
        pieces_from.reset();
    }
    // How the calling thread hears that it leads, or that its results are there: through its OWN mutex and condition variable
    // (round 6).  With one condition variable for everybody, the end of a batch of T calls woke T threads that each had to
    // take the coalescer's mutex to look at their flag, one after the other, while the first ones back were already queueing
    // for the same mutex with their next frames: at the reference's T = hardware_concurrency() = 256 on the GPU box a batch
    // took 4.7 ms from end to end, 0.3 of them on the device.
    std::mutex own;
    std::condition_variable told;
    std::atomic<bool> done{false}, lead{false};

    void tell(std::atomic<bool>& what) // (the notify under the lock: the waiter may destroy this object as soon as it has seen the flag)
    {
        std::lock_guard<std::mutex> hold(own);
        what.store(true, std::memory_order_release);
        told.notify_one();
    }
    void wait_to_be_told() // (no looking before going to sleep: 200 / 2000 sched_yield rounds first cost 256 threads a third / nine tenths of their rate)
    {
        std::unique_lock<std::mutex> hold(own);
        told.wait(hold, [&] { return done.load(std::memory_order_acquire) || lead.load(std::memory_order_acquire); });
    }
};

template <class Backend>
class CallCoalescer {
public:
    enum Kind { kDecode = 0, kEncode = 1, kDecode32 = 2, kEncode32 = 3 };

private:
    const Kind kind;
    const bool encode; // (kind == kEncode)
    std::mutex mu;
    std::deque<SmallCall*> queue;
    std::atomic<size_t> waiting{0}; // queue.size(), for a lingering leader to watch without the mutex
    const int max_leaders;
    int leaders = 0;    // batches between "a call was told to lead" and "its callers have their results"
    int designated = 0; // of those, the ones still in the queue (lingering for company): arrivals join them instead of leading
    size_t last_batch = 0;

    // a free seat and calls nobody leads: the oldest of them leads (mu held; the caller notifies)
    bool promote_locked()
    {
        if (leaders >= max_leaders || designated != 0 || queue.empty())
            return false;
        leaders++, designated++;
        queue.front()->tell(queue.front()->lead);
        return true;
    }
    static constexpr size_t kMaxCalls = 4096;

    void run_one(SmallCall& c)
    {
        c.pieces.clear(), c.pieces_from.reset(), c.rc = SELA_HIP_OK, c.error.clear(); // (a call retried on its own after its batch failed)
        c.rc = encode ? Backend::encode_now(c.pcm, c.n_frames, c.channels, c.frames_out, c.frames_cap, c.offsets_out)
            : kind == kEncode32 ? Backend::encode_i32_now(c.samples, c.n_frames, c.channels, c.shape, c.frames_out, c.frames_cap, c.offsets_out)
            : kind == kDecode32 ? Backend::decode_i32_now(c.frames, c.offsets_in, c.n_frames, c.channels, c.samples_out, c.stride, c.counts_out)
                                : Backend::decode_now(c.frames, c.offsets_in, c.n_frames, c.channels, c.pcm_out);
        if (c.rc != SELA_HIP_OK)
            c.error = Backend::last_error();
    }

    static std::shared_ptr<void> shared_staging(size_t bytes) // a page-locked buffer that goes back to the pool when its last holder lets go
    {
        void* p = Backend::take(bytes);
        return p ? std::shared_ptr<void>(p, [](void* q) { Backend::give(q); }) : std::shared_ptr<void>();
    }

    void run_batch(const std::vector<SmallCall*>& batch)
    {
        if (batch.size() == 1)
            return run_one(*batch[0]);
        const uint32_t channels = batch[0]->channels;
        const size_t frame_pcm = (size_t)SELA_HIP_SAMPLES_PER_FRAME * channels * sizeof(int16_t);
        size_t total = 0;
        for (const SmallCall* c : batch)
            total += c->n_frames;
        std::vector<uint64_t> offsets(total + 1, 0);
        struct Staging { // the page-locked buffers go back to the pool however this function is left (a std::string or an error
            void* p = nullptr; // assignment below may throw: submit() catches, the buffers must not leak)
            ~Staging()
            {
                if (p)
                    Backend::give(p);
            }
            operator void*() const { return p; }
            Staging& operator=(void* q)
            {
                p = q;
                return *this;
            }
        } in, out;
        int rc = SELA_HIP_OK;
        bool oom = false;
        if (encode) {
            const size_t cap = Backend::encode_bound_bytes((uint32_t)total, channels);
            in = Backend::take(total * frame_pcm);
            out = Backend::take(cap);
            if (!in.p || !out.p) {
                rc = SELA_HIP_ENOMEM, oom = true;
            } else {
                size_t at = 0;
                for (const SmallCall* c : batch) {
                    std::memcpy(static_cast<uint8_t*>(in.p) + at * frame_pcm, c->pcm, c->n_frames * frame_pcm);
                    at += c->n_frames;
                }
                rc = Backend::encode_now(static_cast<const int16_t*>(in.p), (uint32_t)total, channels, static_cast<uint8_t*>(out.p), cap, offsets.data());
            }
            const std::string msg = rc == SELA_HIP_OK ? std::string() : (oom ? std::string("no page-locked memory for a coalesced batch") : Backend::last_error());
            size_t at = 0;
            for (SmallCall* c : batch) {
                const uint64_t base = offsets[at], bytes = offsets[at + c->n_frames] - base;
                if (rc != SELA_HIP_OK) {
                    c->rc = rc, c->error = msg;
                } else if (bytes > c->frames_cap) {
                    c->rc = SELA_HIP_ECAPACITY, c->error = "frames_out too small (see sela_hip_encode_bound_bytes)";
                } else {
                    std::memcpy(c->frames_out, static_cast<const uint8_t*>(out.p) + base, (size_t)bytes);
                    for (uint32_t f = 0; f <= c->n_frames; f++)
                        c->offsets_out[f] = offsets[at + f] - base;
                }
                at += c->n_frames;
            }
        } else if (kind == kEncode32) {
            // data::WavFrame values of ONE shape (submit() batches by channels and samples per channel): one job, every call its own bytes
            const uint32_t n = batch[0]->shape;
            const size_t frame_in = (size_t)n * channels * sizeof(int32_t);
            // room for what the callers have room for (the bound of a frame is the format's worst case, 65535 words per subframe:
            // half a megabyte per stereo frame; callers size their buffers for their data and are told when that was too little)
            size_t cap = 0;
            for (const SmallCall* c : batch)
                cap += std::min(c->frames_cap, Backend::encode_i32_bound_bytes(c->n_frames, channels, n));
            in = Backend::take(total * frame_in);
            out = Backend::take(cap);
            if (!in.p || !out.p) {
                rc = SELA_HIP_ENOMEM, oom = true;
            } else {
                size_t at = 0;
                for (const SmallCall* c : batch) {
                    std::memcpy(static_cast<uint8_t*>(in.p) + at * frame_in, c->samples, c->n_frames * frame_in);
                    at += c->n_frames;
                }
                rc = Backend::encode_i32_now(static_cast<const int32_t*>(in.p), (uint32_t)total, channels, n, static_cast<uint8_t*>(out.p), cap, offsets.data());
            }
            if (rc == SELA_HIP_ERANGE || rc == SELA_HIP_ECAPACITY) {
                for (SmallCall* c : batch) // a block the reference cannot answer must not fail its neighbours' calls: everyone on their own
                    run_one(*c);
            } else {
                const std::string msg = rc == SELA_HIP_OK ? std::string() : (oom ? std::string("no page-locked memory for a coalesced batch") : Backend::last_error());
                size_t at = 0;
                for (SmallCall* c : batch) {
                    const uint64_t base = offsets[at], bytes = offsets[at + c->n_frames] - base;
                    if (rc != SELA_HIP_OK) {
                        c->rc = rc, c->error = msg;
                    } else if (bytes > c->frames_cap) {
                        c->rc = SELA_HIP_ECAPACITY, c->error = "frames_out too small (see sela_hip_encode_bound_bytes_n)";
                    } else {
                        std::memcpy(c->frames_out, static_cast<const uint8_t*>(out.p) + base, (size_t)bytes);
                        for (uint32_t f = 0; f <= c->n_frames; f++)
                            c->offsets_out[f] = offsets[at + f] - base;
                    }
                    at += c->n_frames;
                }
            }
        } else if (kind == kDecode32) {
            // frames of any shape to 32-bit channels: one job with the widest caller's stride, every call handed its own rows
            size_t bytes = 0;
            uint32_t stride = 1;
            for (const SmallCall* c : batch) {
                bytes += (size_t)(c->offsets_in[c->n_frames] - c->offsets_in[0] + 3) & ~(size_t)3;
                stride = c->stride > stride ? c->stride : stride;
            }
            const size_t rows = total * channels;
            in = Backend::take(bytes + 4);
            const std::shared_ptr<void> shared_out = shared_staging(rows * stride * sizeof(int32_t) + rows * sizeof(uint32_t));
            if (!in.p || !shared_out) {
                rc = SELA_HIP_ENOMEM, oom = true;
            } else {
                int32_t* const samples = static_cast<int32_t*>(shared_out.get());
                uint32_t* const counts = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(shared_out.get()) + rows * stride * sizeof(int32_t));
                size_t at = 0, pos = 0;
                for (const SmallCall* c : batch) {
                    const uint64_t first = c->offsets_in[0], len = c->offsets_in[c->n_frames] - first;
                    std::memcpy(static_cast<uint8_t*>(in.p) + pos, c->frames + first, (size_t)len);
                    for (uint32_t f = 0; f < c->n_frames; f++)
                        offsets[at + f] = pos + (c->offsets_in[f] - first);
                    at += c->n_frames;
                    pos += ((size_t)len + 3) & ~(size_t)3;
                    offsets[at] = pos;
                }
                rc = Backend::decode_i32_now(static_cast<const uint8_t*>(in.p), offsets.data(), (uint32_t)total, channels, samples, stride, counts);
                if (rc == SELA_HIP_OK) {
                    at = 0;
                    for (SmallCall* c : batch) {
                        for (size_t r = 0; r < (size_t)c->n_frames * channels && c->rc == SELA_HIP_OK; r++) {
                            const uint32_t cnt = counts[at * channels + r];
                            if (cnt > c->stride) {
                                c->rc = SELA_HIP_ECAPACITY, c->error = "stride is smaller than a channel of the frame";
                                c->pieces.clear();
                                break;
                            }
                            c->counts_out[r] = cnt;
                            c->pieces.push_back({ c->samples_out + r * c->stride, samples + (at * channels + r) * stride, (size_t)cnt * sizeof(int32_t) });
                        }
                        if (!c->pieces.empty())
                            c->pieces_from = shared_out;
                        at += c->n_frames;
                    }
                }
            }
            if (rc == SELA_HIP_EFORMAT || rc == SELA_HIP_ERANGE || rc == SELA_HIP_ECAPACITY) {
                for (SmallCall* c : batch) // somebody's frame must not fail its neighbours' calls: everyone on their own
                    run_one(*c);
            } else if (rc != SELA_HIP_OK) {
                const std::string msg = oom ? std::string("no page-locked memory for a coalesced batch") : Backend::last_error();
                for (SmallCall* c : batch)
                    c->rc = rc, c->error = msg;
            }
        } else {
            size_t bytes = 0;
            for (const SmallCall* c : batch)
                bytes += (size_t)(c->offsets_in[c->n_frames] - c->offsets_in[0] + 3) & ~(size_t)3;
            in = Backend::take(bytes + 4);
            const std::shared_ptr<void> shared_out = shared_staging(total * frame_pcm);
            if (!in.p || !shared_out) {
                rc = SELA_HIP_ENOMEM, oom = true;
            } else {
                size_t at = 0, pos = 0;
                for (const SmallCall* c : batch) {
                    const uint64_t first = c->offsets_in[0], len = c->offsets_in[c->n_frames] - first;
                    std::memcpy(static_cast<uint8_t*>(in.p) + pos, c->frames + first, (size_t)len);
                    for (uint32_t f = 0; f < c->n_frames; f++)
                        offsets[at + f] = pos + (c->offsets_in[f] - first);
                    at += c->n_frames;
                    pos += ((size_t)len + 3) & ~(size_t)3; // (frames are whole words: every call's first frame stays aligned)
                    offsets[at] = pos; // (the padding, if a malformed frame left any, belongs to the call's last frame)
                }
                rc = Backend::decode_now(static_cast<const uint8_t*>(in.p), offsets.data(), (uint32_t)total, channels, static_cast<int16_t*>(shared_out.get()));
            }
            if (rc == SELA_HIP_EFORMAT) {
                // somebody's malformed frame must not fail its neighbours' calls: everyone on their own
                for (SmallCall* c : batch)
                    run_one(*c);
            } else {
                const std::string msg = rc == SELA_HIP_OK ? std::string() : (oom ? std::string("no page-locked memory for a coalesced batch") : Backend::last_error());
                size_t at = 0;
                for (SmallCall* c : batch) {
                    if (rc != SELA_HIP_OK)
                        c->rc = rc, c->error = msg;
                    else
                        c->pieces.push_back({ c->pcm_out, static_cast<const uint8_t*>(shared_out.get()) + at * frame_pcm, c->n_frames * frame_pcm }), c->pieces_from = shared_out;
                    at += c->n_frames;
                }
            }
        }
    }

public:
    explicit CallCoalescer(bool enc, int seats = kCoalesceLeaders) : kind(enc ? kEncode : kDecode), encode(enc), max_leaders(seats < 1 ? 1 : seats) {}
    explicit CallCoalescer(Kind k, int seats = kCoalesceLeaders) : kind(k), encode(k == kEncode), max_leaders(seats < 1 ? 1 : seats) {}

    int submit(SmallCall& call)
    {
        std::unique_lock<std::mutex> lock(mu);
        queue.push_back(&call);
        waiting.store(queue.size(), std::memory_order_relaxed);
        promote_locked(); // (this call itself, if a seat is free and nobody is gathering; nobody else waits to be told)
        lock.unlock();
        call.wait_to_be_told();
        if (!call.done.load(std::memory_order_acquire)) {
            lock.lock();
            // this call leads.  If the batch before held several calls, their threads are on their way back with their
            // next frames right now: give them until the queue has stopped growing for a moment (bounded) -- a trip to
            // the device costs more than that
            if (last_batch > 1) {
                // (the queue's length is watched without the mutex, which the arriving calls need: polling under it, 256 threads kept
                // the leader waiting 370 us for its "at most 150")
                const size_t want = last_batch;
                lock.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                size_t seen = waiting.load(std::memory_order_relaxed);
                auto last_growth = t0;
                for (;;) {
                    std::this_thread::yield();
                    const auto now = std::chrono::steady_clock::now();
                    const size_t is = waiting.load(std::memory_order_relaxed);
                    if (is != seen)
                        seen = is, last_growth = now;
                    if (seen >= want || now - last_growth > std::chrono::microseconds(20) || now - t0 > std::chrono::microseconds(150))
                        break;
                }
                lock.lock();
            }
            std::vector<SmallCall*> batch; // everything that waits for this device with this channel count (and shape), this call included
            for (auto it = queue.begin(); it != queue.end() && batch.size() < kMaxCalls;) {
                if ((*it)->channels == call.channels && (*it)->device == call.device && (*it)->shape == call.shape) {
                    batch.push_back(*it);
                    it = queue.erase(it);
                } else {
                    ++it;
                }
            }
            waiting.store(queue.size(), std::memory_order_relaxed);
            designated--;
            promote_locked(); // (calls of another shape that stay behind, if a seat is free)
            lock.unlock();
            try {
                run_batch(batch);
            } catch (...) { // (std::bad_alloc: the callers hear of it, nobody is left waiting)
                for (SmallCall* c : batch)
                    if (c->rc == SELA_HIP_OK)
                        c->rc = SELA_HIP_ENOMEM, c->error = "out of memory while staging a coalesced batch";
            }
            Backend::after_batch(); // the streams and buffers this thread used go to whoever leads next: any caller may
            lock.lock();
            last_batch = batch.size();
            leaders--;
            promote_locked();
            lock.unlock();
            for (SmallCall* c : batch) // (this call's own flag last: nothing of it is touched by anybody afterwards)
                if (c != &call)
                    c->tell(c->done);
            call.done.store(true, std::memory_order_release);
        }
        call.collect(); // (this call's share of a decoded batch, by its own thread)
        return call.rc; // (call.error says why; the caller turns it into its thread's last error)
    }
};

} // namespace sela
#endif // SELA_COALESCER_H_
