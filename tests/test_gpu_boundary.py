"""GPU tests, by subject: the C-ABI boundary (include/sela_hip.h) -- host-pointer calls, the streaming jobs, calls from many threads and their
coalescing, leased contexts, and the reference's L1 classes as stages (sela_hip_lpc_*, sela_hip_rice_*) -- every result against the oracle."""
import ctypes as C
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
import generic_cases as gc
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import _decode, _encode, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


def test_host_pointer_api(gpu):
    """The synchronous host-pointer entry points used by the C++ host."""
    from sela_amd import codec

    o = oracle()
    pcm = synth_frames(9, 2, 30)
    frames, offsets = codec.encode_host(pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=2)
    assert np.array_equal(frames, ref_frames) and np.array_equal(offsets, ref_offsets)
    assert np.array_equal(codec.index_frames(frames, 9, 2), offsets)
    assert np.array_equal(codec.decode_host(frames, offsets, 2), pcm)
    # empty batch
    f0, o0 = codec.encode_host(np.zeros((0, 2048, 2), np.int16))
    assert len(f0) == 0 and o0.tolist() == [0]


def test_host_pointer_api_chunked_pipeline(gpu):
    """Host pointers (sela_capi.hip): an encode is one launch that fetches its PCM and writes its frames itself
    (stereo; other channel counts are copied in first), a decode goes through the copy-in / kernel / copy-out
    pipeline in chunks of 384 / 640 / 1024 frames: same bytes as the device-pointer call on the whole batch,
    including a short last chunk, and a corrupt frame in a later chunk is still reported."""
    from sela_amd import capi, codec

    n = 2 * 1024 + 300
    pcm = synth_frames(n, 2, 31)
    frames, offsets = codec.encode_host(pcm)
    dev_frames, dev_offsets, _, _ = _encode(gpu, pcm)
    assert np.array_equal(offsets, dev_offsets) and np.array_equal(frames, dev_frames)
    assert np.array_equal(codec.index_frames(frames, n, 2), offsets)
    assert np.array_equal(codec.decode_host(frames, offsets, 2), _decode(gpu, frames, offsets, 2))
    # mono, nine decode chunks, the last of a single frame
    n1 = 2 * 4096 + 1
    pcm1 = np.tile(synth_frames(683, 1, 32), (13, 1, 1))[:n1]
    f1, o1 = codec.encode_host(pcm1)
    d1, do1, _, _ = _encode(gpu, pcm1)
    assert np.array_equal(o1, do1) and np.array_equal(f1, d1)
    assert np.array_equal(codec.decode_host(f1, o1, 1), _decode(gpu, f1, o1, 1))
    bad = f1.copy()
    bad[int(o1[5000])] ^= 0xFF  # sync word of a frame in the second decode chunk
    with pytest.raises(capi.SelaHipError):
        codec.decode_host(bad, o1, 1)


def test_streaming_jobs_equal_one_shot(gpu):
    """begin / feed / end with pieces of awkward sizes, on page-locked buffers from sela_hip_host_alloc: the
    same bytes, offsets and samples as the one-shot calls, and the progress counters only ever report
    data that is final."""
    from sela_amd import capi, codec

    lib = capi.lib()
    n, ch = 3 * 1024 + 517, 2
    pcm = synth_frames(n, ch, 71)
    want_frames, want_offsets = codec.encode_host(pcm)

    def pinned(nbytes, dtype):
        p = lib.sela_hip_host_alloc(nbytes)
        assert p
        return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dtype)

    p_pcm, a_pcm = pinned(pcm.nbytes, np.int16)
    a_pcm[:] = pcm.reshape(-1)
    cap = int(lib.sela_hip_encode_bound_bytes(n, ch))
    p_out, a_out = pinned(cap, np.uint8)
    offs = np.zeros(n + 1, np.uint64)
    job = C.c_void_p()
    capi.check(lib.sela_hip_encode_begin(C.byref(job), ch, n, p_out, cap, offs.ctypes.data))
    fed, ff, bf = 0, C.c_uint32(0), C.c_uint64(0)
    frame_bytes = 2048 * ch * 2
    for piece in (1, 700, 1024, 1500, 10**6):
        nf = min(piece, n - fed)
        capi.check(lib.sela_hip_encode_feed(job, p_pcm + fed * frame_bytes, nf, C.byref(ff), C.byref(bf)))
        fed += nf
        assert ff.value <= fed and bf.value == (int(want_offsets[ff.value]) if ff.value else 0)
        assert np.array_equal(a_out[: bf.value], want_frames[: bf.value])  # what is reported final is final
        if fed == n:
            break
    capi.check(lib.sela_hip_encode_end(job, C.byref(ff), C.byref(bf)))
    assert ff.value == n and bf.value == len(want_frames)
    assert np.array_equal(offs, want_offsets) and np.array_equal(a_out[: bf.value], want_frames)

    # decode job: pieces of whole frames
    want_pcm = codec.decode_host(want_frames, want_offsets, ch)
    p_back, a_back = pinned(pcm.nbytes, np.int16)
    job = C.c_void_p()
    capi.check(lib.sela_hip_decode_begin(C.byref(job), ch, n, p_back))
    fed = 0
    for piece in (3, 1024, 2000, 10**6):
        nf = min(piece, n - fed)
        o = np.ascontiguousarray(offs[fed: fed + nf + 1])
        capi.check(lib.sela_hip_decode_feed(job, p_out, o.ctypes.data, nf, C.byref(ff)))
        fed += nf
        assert ff.value <= fed
        assert np.array_equal(a_back[: ff.value * 2048 * ch], want_pcm.reshape(-1)[: ff.value * 2048 * ch])
        if fed == n:
            break
    capi.check(lib.sela_hip_decode_end(job, C.byref(ff)))
    assert ff.value == n and np.array_equal(a_back, want_pcm.reshape(-1))
    for p in (p_pcm, p_out, p_back):
        lib.sela_hip_host_free(p)
    # a second job on the thread after the first was closed, and an open job blocks another
    job = C.c_void_p()
    capi.check(lib.sela_hip_decode_begin(C.byref(job), ch, 0, None))
    other = C.c_void_p()
    assert lib.sela_hip_decode_begin(C.byref(other), ch, 0, None) == -2
    capi.check(lib.sela_hip_decode_end(job, None))


def test_host_pointer_calls_from_two_threads(gpu):
    """Two threads, each with its own context (streams, buffers, staging kernels), encode and decode different
    tracks at the same time on the one GPU: every call returns the bytes the same call returns alone.  (The
    one-launch host encoder's blocks wait for its staging kernel, and two of those pairs share the device here.)"""
    import threading
    from sela_amd import codec

    tracks = [synth_frames(1500, 2, 91), synth_frames(1100, 2, 92)]
    alone = [codec.encode_host(t) for t in tracks]
    problems = []

    def work(i):
        try:
            for _ in range(4):
                frames, offsets = codec.encode_host(tracks[i])
                if not (np.array_equal(frames, alone[i][0]) and np.array_equal(offsets, alone[i][1])):
                    problems.append("thread %d: encode differs" % i)
                if not np.array_equal(codec.decode_host(frames, offsets, 2), tracks[i]):
                    problems.append("thread %d: decode differs" % i)
        except Exception as e:  # noqa: BLE001 -- reported below, from the test's thread
            problems.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not problems, problems


@pytest.mark.parametrize("channels,n", [(1, 2500), (3, 1300), (6, 700)])
def test_host_pipeline_other_channel_counts(gpu, channels, n):
    """Several chunks of mono, three- and six-channel frames through the host-pointer pipeline (chunk sizes, frame
    offsets read from page-locked memory, one flag byte per frame and wave): the same bytes and samples as the
    device-pointer calls on the whole batch."""
    from sela_amd import codec

    pcm = synth_frames(n, channels, 90 + channels)
    frames, offsets = codec.encode_host(pcm)
    dev_frames, dev_offsets, _, _ = _encode(gpu, pcm)
    assert np.array_equal(offsets, dev_offsets) and np.array_equal(frames, dev_frames)
    back = codec.decode_host(frames, offsets, channels)
    assert np.array_equal(back, _decode(gpu, frames, offsets, channels))
    # a frame without its sync word in the last chunk is reported, the rest still decodes
    bad = frames.copy()
    bad[int(offsets[n - 2])] ^= 0xFF
    with pytest.raises(Exception):
        codec.decode_host(bad, offsets, channels)


def test_encode_feed_survives_losing_its_staging_kernel(gpu):
    """A device so busy that the staging kernel gets no compute units in time used to fail the call (the blocks' bounded
    wait ran out: SELA_HIP_ENODEV).  Now the feed -- and everything queued behind it -- is issued again with the copy
    engine in place of the stagers.  Forced here by a wait bound of zero (sela_hip_debug_stage_wait): a streaming job of
    several feeds must still return the oracle's bytes, and the library must say that it took the second path."""
    from sela_amd import capi, codec

    lib = capi.lib()
    o = oracle()
    pcm = synth_frames(2600, 2, 61)
    want_frames, want_offsets, _ = o.encode_frames(pcm, threads=os.cpu_count() or 1)
    before = lib.sela_hip_debug_reissued_feeds()
    lib.sela_hip_debug_stage_wait(0)
    try:
        # one-shot
        frames, offsets = codec.encode_host(pcm)
        assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames)
        # a streaming job: three feeds queued back to back from page-locked memory, then end
        n = pcm.shape[0]
        nbytes = pcm.nbytes
        host_pcm = lib.sela_hip_host_alloc(nbytes)
        cap = int(lib.sela_hip_encode_bound_bytes(n, 2))
        host_out = lib.sela_hip_host_alloc(cap)
        C.memmove(host_pcm, pcm.ctypes.data, nbytes)
        offs = np.zeros(n + 1, np.uint64)
        job = C.c_void_p()
        capi.check(lib.sela_hip_encode_begin(C.byref(job), 2, n, host_out, cap, offs.ctypes.data))
        cuts = [0, 900, 1700, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            capi.check(lib.sela_hip_encode_feed(job, host_pcm + a * 2048 * 2 * 2, b - a, None, None))
        total = C.c_uint64(0)
        capi.check(lib.sela_hip_encode_end(job, None, C.byref(total)))
        got = np.ctypeslib.as_array((C.c_uint8 * total.value).from_address(host_out)).copy()
        lib.sela_hip_host_free(host_pcm)
        lib.sela_hip_host_free(host_out)
        assert np.array_equal(offs, want_offsets) and np.array_equal(got, want_frames)
    finally:
        lib.sela_hip_debug_stage_wait(-1)
    assert lib.sela_hip_debug_reissued_feeds() > before, "the wait bound of zero did not send any feed down the second path"
    # and the usual path still works afterwards
    frames, offsets = codec.encode_host(pcm)
    assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames)


def test_consecutive_host_encodes_of_different_audio(gpu):
    """The host encoder's device copy of the PCM (and its per-frame ready words) is reused from call to call; a block
    that took a frame of the PREVIOUS call for its own would go unnoticed if every call fed the same samples.  Eight
    calls, every one on different audio (the same track XOR a counter), each against the oracle."""
    from sela_amd import codec

    o = oracle()
    base = synth_frames(700, 2, 62)
    for i in range(8):
        pcm = (base ^ np.int16(i * 257)).astype(np.int16)
        want_frames, want_offsets, _ = o.encode_frames(pcm, threads=os.cpu_count() or 1)
        frames, offsets = codec.encode_host(pcm)
        assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames), i


def test_four_threads_encode_on_one_gpu(gpu):
    """Four host threads with their own jobs on the one GPU (more streams than the device has hardware queues; only one
    job at a time takes the staging-kernel path, the others use the copy engine): every call returns the bytes the same
    call returns alone, nobody fails, nobody hangs."""
    import threading
    from sela_amd import codec

    tracks = [synth_frames(900 + 150 * i, 2, 70 + i) for i in range(4)]
    alone = [codec.encode_host(t) for t in tracks]
    # (against the decode done alone, not against the input: the reference's codec is off by one in a few frames)
    alone_back = [codec.decode_host(f, o, 2) for f, o in alone]
    problems = []

    def work(i):
        try:
            for _ in range(5):
                frames, offsets = codec.encode_host(tracks[i])
                if not (np.array_equal(frames, alone[i][0]) and np.array_equal(offsets, alone[i][1])):
                    problems.append("thread %d: encode differs" % i)
                if not np.array_equal(codec.decode_host(frames, offsets, 2), alone_back[i]):
                    problems.append("thread %d: decode differs" % i)
        except Exception as e:  # noqa: BLE001 -- reported below, from the test's thread
            problems.append("thread %d: %r" % (i, e))

    o = oracle()
    for i in range(4):
        ref_back, _ = o.decode_frames(alone[i][0], alone[i][1], 2, threads=8)
        assert np.array_equal(alone_back[i], ref_back)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(180)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not problems, problems


def test_threads_take_over_parked_contexts(gpu):
    """A thread's streams, events and staging buffers are parked when it ends (or calls sela_hip_thread_release) and taken
    over by the next thread on the device: ten threads one after the other create at most one context between them, three
    at once at most three, and what they encode and decode -- different audio, different sizes, mono after stereo -- is what
    the same calls give on the test's own thread."""
    import threading
    from sela_amd import capi, codec

    lib = capi.lib()
    jobs = [synth_frames(40 + 37 * i, 1 + (i % 3 != 2), 400 + i) for i in range(10)]
    expect = []
    for pcm in jobs:
        frames, offsets = codec.encode_host(pcm)
        expect.append((frames, offsets, codec.decode_host(frames, offsets, pcm.shape[2])))
    problems = []

    def work(i, release):
        try:
            frames, offsets = codec.encode_host(jobs[i])
            back = codec.decode_host(frames, offsets, jobs[i].shape[2])
            if not (np.array_equal(frames, expect[i][0]) and np.array_equal(offsets, expect[i][1]) and np.array_equal(back, expect[i][2])):
                problems.append("job %d differs" % i)
            if release:
                lib.sela_hip_thread_release()
        except Exception as e:  # noqa: BLE001 -- reported below, from the test's thread
            problems.append("job %d: %r" % (i, e))

    before = lib.sela_hip_debug_contexts_created()
    for i in range(10):
        t = threading.Thread(target=work, args=(i, i % 2 == 0))
        t.start()
        t.join(120)
        assert not t.is_alive()
    assert lib.sela_hip_debug_contexts_created() - before <= 1
    assert not problems, problems
    before = lib.sela_hip_debug_contexts_created()
    threads = [threading.Thread(target=work, args=(i, False)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads)
    assert lib.sela_hip_debug_contexts_created() - before <= 2  # (one was parked by the threads above)
    assert not problems, problems
    # shutdown frees the parked ones; the next call simply builds a new one
    lib.sela_hip_shutdown()
    frames, offsets = codec.encode_host(jobs[0])
    assert np.array_equal(frames, expect[0][0]) and np.array_equal(offsets, expect[0][1])


@pytest.mark.parametrize("channels,pinned_io", [(2, True), (2, False), (1, True), (5, False)])
def test_streaming_jobs_with_random_feeds(gpu, channels, pinned_io):
    """Encode and decode jobs fed in pieces of random sizes (1 frame to 1500, sixteen jobs each), from page-locked and from
    ordinary memory, progress polled with empty feeds in between: whatever is reported final is final and equal to the
    oracle's, the totals are the oracle's."""
    from sela_amd import capi

    lib = capi.lib()
    o = oracle()
    rng = np.random.default_rng(100 + channels + 10 * pinned_io)
    n_max = 2600 if channels <= 2 else 700
    pool = synth_frames(n_max, channels, 200 + channels)
    ref_frames, ref_offsets, _ = o.encode_frames(pool, threads=os.cpu_count() or 1)
    ref_back, _ = o.decode_frames(ref_frames, ref_offsets, channels, threads=os.cpu_count() or 1)
    frame_bytes = 2048 * channels * 2

    def buffer(nbytes):
        if pinned_io:
            p = lib.sela_hip_host_alloc(max(nbytes, 1))
            return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))
        a = np.zeros(max(nbytes, 1), np.uint8)
        return a.ctypes.data, a

    for _ in range(16):
        n = int(rng.integers(1, n_max + 1))
        start = int(rng.integers(0, n_max - n + 1))
        want_offs = ref_offsets[start: start + n + 1] - ref_offsets[start]
        want = ref_frames[int(ref_offsets[start]): int(ref_offsets[start + n])]
        p_pcm, a_pcm = buffer(n * frame_bytes)
        a_pcm[: n * frame_bytes] = pool[start: start + n].reshape(-1).view(np.uint8)
        cap = int(lib.sela_hip_encode_bound_bytes(n, channels))
        p_out, a_out = buffer(cap)
        offs = np.zeros(n + 1, np.uint64)
        job, ff, bf = C.c_void_p(), C.c_uint32(0), C.c_uint64(0)
        capi.check(lib.sela_hip_encode_begin(C.byref(job), channels, n, p_out, cap, offs.ctypes.data))
        fed = 0
        while fed < n:
            nf = min(int(rng.choice([1, 2, 7, 64, 300, 1024, 1500])), n - fed)
            capi.check(lib.sela_hip_encode_feed(job, p_pcm + fed * frame_bytes, nf, C.byref(ff), C.byref(bf)))
            fed += nf
            if rng.random() < 0.5:
                capi.check(lib.sela_hip_encode_feed(job, p_pcm, 0, C.byref(ff), C.byref(bf)))  # (only reports)
            assert ff.value <= fed and bf.value == int(want_offs[ff.value])
            assert np.array_equal(a_out[: bf.value], want[: bf.value])
        capi.check(lib.sela_hip_encode_end(job, C.byref(ff), C.byref(bf)))
        assert ff.value == n and bf.value == len(want)
        assert np.array_equal(offs, want_offs) and np.array_equal(a_out[: bf.value], want)
        # ... and back
        p_back, a_back = buffer(n * frame_bytes)
        want_pcm = ref_back[start: start + n].reshape(-1).view(np.uint8)
        job = C.c_void_p()
        capi.check(lib.sela_hip_decode_begin(C.byref(job), channels, n, p_back))
        fed = 0
        while fed < n:
            nf = min(int(rng.choice([1, 3, 50, 700, 1024, 1500])), n - fed)
            piece = np.ascontiguousarray(offs[fed: fed + nf + 1])
            capi.check(lib.sela_hip_decode_feed(job, p_out, piece.ctypes.data, nf, C.byref(ff)))
            fed += nf
            assert ff.value <= fed
            assert np.array_equal(a_back[: ff.value * frame_bytes], want_pcm[: ff.value * frame_bytes])
        capi.check(lib.sela_hip_decode_end(job, C.byref(ff)))
        assert ff.value == n and np.array_equal(a_back[: n * frame_bytes], want_pcm)
        if pinned_io:
            for p in (p_pcm, p_out, p_back):
                lib.sela_hip_host_free(p)


def test_small_calls_from_many_threads_are_coalesced_and_stay_their_own(gpu):
    """sela_hip_encode / sela_hip_decode calls of a few frames from many threads at once (a binding that keeps the
    reference's per-frame thread loop, src/sela/encoder.cpp:58-73) are merged into device batches inside the library: every
    call gets the bytes the same call gets alone -- 1 to 9 frames, mono and stereo callers mixed --, a caller whose output
    buffer is too small gets SELA_HIP_ECAPACITY and a caller with a malformed frame SELA_HIP_EFORMAT, each alone."""
    import threading
    from sela_amd import capi, codec

    lib = capi.lib()
    n_threads, rounds = 24, 5
    jobs = [[synth_frames(1 + (t + r) % 9, 1 if t % 4 == 3 else 2, 900 + 16 * t + r) for r in range(rounds)] for t in range(n_threads)]
    alone = [[codec.encode_host(p) for p in row] for row in jobs]
    alone_back = [[codec.decode_host(f, o, p.shape[2]) for (f, o), p in zip(row, prow)] for row, prow in zip(alone, jobs)]
    problems, outcomes = [], {}

    def work(t):
        try:
            for r in range(rounds):
                pcm = jobs[t][r]
                n, ch = pcm.shape[0], pcm.shape[2]
                if t == 7 and r == 2:  # an output buffer that cannot hold the frames
                    frames = np.empty(64, np.uint8)
                    offs = np.zeros(n + 1, np.uint64)
                    outcomes["cap"] = lib.sela_hip_encode(pcm.ctypes.data, n, ch, 2048, frames.ctypes.data, frames.nbytes, offs.ctypes.data)
                    continue
                frames, offs = codec.encode_host(pcm)
                if not (np.array_equal(frames, alone[t][r][0]) and np.array_equal(offs, alone[t][r][1])):
                    problems.append("thread %d round %d: encode differs" % (t, r))
                if t == 11 and r == 3:  # a frame without its sync word
                    broken = frames.copy()
                    broken[int(offs[n - 1])] ^= 0xFF
                    back = np.empty((n, 2048, ch), np.int16)
                    outcomes["format"] = lib.sela_hip_decode(broken.ctypes.data, offs.ctypes.data, n, ch, back.ctypes.data)
                    if n > 1 and not np.array_equal(back[: n - 1], alone_back[t][r][: n - 1]):
                        problems.append("the frames in front of the broken one differ")
                    continue
                if not np.array_equal(codec.decode_host(frames, offs, ch), alone_back[t][r]):
                    problems.append("thread %d round %d: decode differs" % (t, r))
        except Exception as e:  # noqa: BLE001 -- reported below, from the test's thread
            problems.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(180)
    assert not any(th.is_alive() for th in threads), "a thread is stuck"
    assert not problems, problems
    assert outcomes == {"cap": -4, "format": -5}, outcomes


# ---- the stages on their own: the reference's L1 classes on the device (sela_hip_lpc_* / sela_hip_rice_*) -------------------
def test_rice_stage_on_the_known_answers(gpu, kats):  # noqa: F811
    """rice::RiceEncoder / RiceDecoder by themselves (the reference's test/ricetests.cpp:7-25 calls them directly): the
    reference's parameter and words for every Rice KAT of tests/golden/kats.npz -- a single value, runs of ones longer than a
    word, values near 2^20 -- in ONE batched call each way, and the values back."""
    from sela_amd import codec

    names = [str(n) for n in kats["rice_names"]]
    values = [kats[f"rice/{n}/values"] for n in names]
    got = codec.rice_encode(values)
    for n, (k, words) in zip(names, got):
        assert k == int(kats[f"rice/{n}/k"]), n
        assert np.array_equal(words, kats[f"rice/{n}/words"]), n
    back = codec.rice_decode([(k, w, len(v)) for (k, w), v in zip(got, values)])
    for n, v, b in zip(names, values, back):
        assert np.array_equal(b, v), n


def test_rice_stage_against_the_oracle_on_random_streams(gpu):  # noqa: F811
    """Streams of 1 .. 5000 values of every magnitude up to 2^29, empty streams among them, against the oracle's coder."""
    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(5)
    streams = [np.zeros(0, np.int32)]
    for i in range(60):
        n = int(rng.integers(1, 5000)) if i % 7 else int(rng.integers(1, 4))
        scale = int(rng.integers(1, 30))
        streams.append(rng.integers(-(1 << scale), 1 << scale, n).astype(np.int32))
    got = codec.rice_encode(streams)
    for v, (k, words) in zip(streams, got):
        if len(v) == 0:
            continue
        rk, rw = o.rice_encode(v)
        assert k == rk and np.array_equal(words, rw), (len(v), int(np.abs(v).max()))
    back = codec.rice_decode([(k, w, len(v)) for (k, w), v in zip(got, streams)])
    for v, b in zip(streams, back):
        assert np.array_equal(b, v)
    with pytest.raises(capi.SelaHipError) as e:  # the reference's int32 zig-zag overflows: flagged, not wrapped
        codec.rice_encode([np.array([1 << 30], np.int32)])
    assert e.value.code == -6
    with pytest.raises(capi.SelaHipError) as e:  # a stream that ends before its values do
        codec.rice_decode([(3, np.array([0xFFFFFFFF], np.uint32), 5)])
    assert e.value.code == -5


def test_lpc_stage_on_the_known_answers(gpu, kats):  # noqa: F811
    """lpc::ResidueGenerator / SampleGenerator / LinearPredictor by themselves (test/lpctests.cpp:10-32): order, quantised
    coefficients, Q35 predictor and residues of every block KAT -- the 17-bit difference signal included -- and the samples
    back from them, in one batched call each way."""
    from sela_amd import codec

    names = [str(n) for n in kats["blk_names"]]
    samples = np.stack([kats[f"blk/{n}/samples"] for n in names]).astype(np.int32)
    order, q, residues = codec.lpc_encode(samples)
    for i, n in enumerate(names):
        assert order[i] == int(kats[f"blk/{n}/order"]), n
        assert np.array_equal(q[i, : order[i]], kats[f"blk/{n}/q"]), n
        assert np.array_equal(residues[i], kats[f"blk/{n}/residues"]), n
    back, coefs = codec.lpc_decode(order, q, residues, want_coefficients=True)
    o = oracle()
    for i, n in enumerate(names):
        assert np.array_equal(coefs[i, : order[i] + 1], kats[f"blk/{n}/a"]), n
        ref = o.lpc_synth(int(order[i]), q[i, : order[i]], residues[i]) if hasattr(o, "lpc_synth") else None
        if ref is not None:
            assert np.array_equal(back[i], ref), n  # (the reference's own decoder: off by one where ITS rounding differs from its encoder's)
        else:
            assert np.array_equal(back[i], samples[i]), n


def test_lpc_stage_against_the_oracle_on_random_blocks(gpu):  # noqa: F811
    """64 blocks of the synthetic album's left, right and difference signals through the stage entries against the oracle's
    analysis and synthesis; samples beyond a 16-bit difference are taken too (round 5: through the any-length kernels)."""
    from sela_amd import capi, codec

    o = oracle()
    pcm = synth_frames(22, 2, 8).astype(np.int32)
    blocks = np.concatenate([pcm[:, :, 0], pcm[:, :, 1], pcm[:, :, 0] - pcm[:, :, 1]])[:64]
    order, q, residues = codec.lpc_encode(blocks)
    for i in range(len(blocks)):
        ro, rq, rr, ra, _, _ = o.lpc_analyze(blocks[i], with_trace=True)
        assert order[i] == ro and np.array_equal(q[i, :ro], rq) and np.array_equal(residues[i], rr), i
    back = codec.lpc_decode(order, q, residues)
    for i in range(len(blocks)):
        assert np.array_equal(back[i], o.lpc_synth(int(order[i]), q[i, : order[i]], residues[i])), i
    wide = np.full((1, 2048), 70000, np.int32)
    wide[0, ::3] = -70001
    order, q, residues = codec.lpc_encode(wide)
    ro, rq, rr = o.lpc_analyze(wide[0])
    assert order[0] == ro and np.array_equal(q[0, :ro], rq) and np.array_equal(residues[0], rr)


def test_rice_stage_cuts_a_stream_beyond_2_24_bits_like_the_reference(gpu):  # noqa: F811
    """sela_hip_rice_encode on a stream whose float-rounded word count is one short: the reference's k, count and
    words -- nothing written past them (the stream behind it in the same call is intact)."""
    from sela_amd import codec

    v, k, bits = gc.long_rice_stream()
    tail = np.array([0, -1, 1, -2, 2, 100, -100, 5], np.int32)
    (k0, w0), (k1, w1) = codec.rice_encode([v, tail])
    ko, wo = oracle().rice_encode(v)
    assert k0 == ko == k and len(w0) == (bits + 31) // 32 - 1 and np.array_equal(w0, wo)
    assert k1 == 5 and [hex(x) for x in w1] == ["0xc8c10800", "0x538fc4f"]
