"""GPU parity tests added in round 3: the mode the headline is timed in (two encode->decode chains in flight on two
streams), bench.py's N > 1 code path, the decoder beyond eight channels, the path-based file verbs and the batch
verbs on album tracks against the reference's digests, and the encode feed that loses its staging kernel."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
from test_gpu_parity import _build_frame, _decode, _encode, gpu  # noqa: F401  (fixture + helpers)
from test_host_cpp import HOST, ROOT, _build, _write_wav

pytestmark = pytest.mark.gpu


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def test_two_streams_in_flight_match_oracle(gpu):
    """What bench.py's `value` is timed in: two Encoder / Decoder pairs on two HIP streams, rounds issued back to back
    with no synchronisation in between, so that launches of the two lanes are co-resident on the device -- launch
    tickets, the XCD ring pools, the mean workers' ready words and the decoder's status words are the state two
    launches could trip over.  50 rounds of different batch sizes (several above 1024 stereo frames = 3072 blocks, where
    the mean workers engage -- on both lanes), every round's frame bytes, offsets, decoded PCM and status words
    against the CPU oracle."""
    torch = gpu
    from sela_amd import capi, codec

    o = oracle()
    threads = os.cpu_count() or 1
    pools = [synth_frames(1700, 2, 21), synth_frames(1700, 2, 22)]
    expect = []
    for pool in pools:  # frames are independent: the oracle's bytes of a slice are the slice of its bytes
        blob, offs, _ = o.encode_frames(pool, threads=threads)
        back, _ = o.decode_frames(blob, offs, 2, threads=threads)
        expect.append((blob, offs, back))
    dev_pools = [torch.from_numpy(p).cuda() for p in pools]
    rng = np.random.default_rng(7)
    sizes = [int(x) for x in rng.choice([1, 3, 40, 200, 700, 1030, 1100, 1300, 1500, 1650], size=50)]
    sizes[0], sizes[1], sizes[2], sizes[3] = 1500, 1650, 1100, 1300  # the first rounds: both lanes full, workers on both
    lanes = [{"enc": codec.Encoder(1700, 2), "dec": codec.Decoder(1700, 2), "stream": torch.cuda.Stream()} for _ in range(2)]
    results = []
    torch.cuda.synchronize()
    for r, n in enumerate(sizes):
        lane = lanes[r % 2]
        start = int(rng.integers(0, 1700 - n + 1))
        st = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
        with torch.cuda.stream(lane["stream"]):
            out = lane["enc"].encode(dev_pools[r % 2][start: start + n], status=st[0])
            back = lane["dec"].decode(out.frames, out.offsets, n, status=st[1])
            # (the lane's buffers are reused two rounds on: keep this round's results, on the lane's own stream)
            results.append((r % 2, start, n, out.frames[: n * 8192 + 64].clone(), out.offsets.clone(), back.clone(), st))
    torch.cuda.synchronize()
    for which, start, n, frames, offsets, back, st in results:
        blob, offs, ref_back = expect[which]
        want_offs = offs[start: start + n + 1] - offs[start]
        got_offs = offsets.cpu().numpy().view(np.uint64)
        assert np.array_equal(got_offs, want_offs), (which, start, n)
        total = int(want_offs[-1])
        assert np.array_equal(frames[:total].cpu().numpy(), blob[int(offs[start]): int(offs[start]) + total]), (which, start, n)
        assert np.array_equal(back.cpu().numpy(), ref_back[start: start + n]), (which, start, n)
        flags = st.cpu().numpy().view(np.uint32)
        assert int(flags[0, 0]) & (capi.FLAG_WORDS_CAP | capi.FLAG_RICE_RANGE | capi.FLAG_COEF_OVERFLOW | capi.FLAG_INTERNAL) == 0 and int(flags[0, 1]) == 0
        assert int(flags[1, 0]) == 0 and int(flags[1, 1]) == 0


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_line_of_the_multi_gpu_code_path():
    """bench.py as the driver runs it for N > 1 -- process group, the RCCL all-gather of the frame sizes inside the timed
    region, the layout checks -- on ONE GPU (SELA_BENCH_FORCE_EXCHANGE=1: a one-rank group): the headline (one track
    per GPU), the album block (BASELINE.json configs[3], layout against the reference's digest) and the decode10k block
    (configs[4]); every key the driver and the judge read is there."""
    line = _bench(["--steps", "2", "--warmup", "1", "--no-host-legs"], env={"SELA_BENCH_FORCE_EXCHANGE": "1"})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "lanes", "timed_outputs", "album", "decode10k"):
        assert key in line, key
    assert line["scaling"] == "weak" and line["n_gpus"] == 1 and line["value"] > 1000
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["timed_outputs"]["lanes_compared_with_serial_step"] == 2
    assert line["cpu_baseline"]["bit_exact_vs_gpu"] is True and line["cpu_baseline"]["cores"] >= 1
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    album = line["album"]
    assert album["layout_matches_reference"] is True and album["scaling"] == "strong" and album["config"]["frames_total"] == 549365
    assert album["timed_outputs"]["equal_to_serial_step"] is True
    d10 = line["decode10k"]
    assert d10["config"]["frames_total"] == 10000 and d10["bit_exact_vs_cpu_decode"] is True and d10["timed_outputs"]["equal_to_serial_step"] is True


def test_bench_album_as_the_headline():
    """`--workload album --steps 1 --warmup 0` with the exchange forced: the album as the line's own workload."""
    line = _bench(["--workload", "album", "--steps", "1", "--warmup", "0"], env={"SELA_BENCH_FORCE_EXCHANGE": "1"})
    assert line["layout_matches_reference"] is True and line["scaling"] == "strong" and line["config"]["frames_total"] == 549365
    assert "roofline" in line and line["value"] > 1000


def test_bench_relaunches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself.  On a one-GPU box the second
    rank has no device to take: what matters here is that the command gets as far as the ranks (no assertion about
    WORLD_SIZE), and fails loudly rather than printing a line for fewer GPUs than asked."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra-legs", "--no-host-legs",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    import torch

    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["layout_matches_reference"] is True
    else:
        assert r.returncode != 0 and "AssertionError: --gpus" not in r.stderr
        assert not r.stdout.strip().startswith("{")


def test_bench_with_two_ranks_sharing_the_gpu():
    """The multi-rank logic of bench.py on a one-GPU box: two ranks under torch.distributed.run, both on GPU 0, talking
    over gloo (SELA_BENCH_RANKS_SHARE_GPU=1; RCCL refuses two ranks on one device).  Rank r's track (album track 3 r)
    against the reference's digests on every rank, the gathered layout of the two tracks, the album cut in two contiguous
    ranges (inside track 43) with its layout against the reference's, 10,000 frames decoded in two halves, one JSON line
    from rank 0 with n_gpus = 2.  Not a measurement: the two ranks share the device and the exchange goes through host
    memory."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SELA_BENCH_RANKS_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-host-legs", "--extra-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([t for t in r.stdout.strip().splitlines() if t.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_total"] == 2 * 3875
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert line["album"]["layout_matches_reference"] is True and line["album"]["config"]["frames_rank0"] in (274682, 274683)
    assert line["album"]["roundtrip_lossy_frames"] == 39  # (the reference's own lossy frames on the album, summed over the ranks)
    assert line["decode10k"]["config"]["frames_rank0"] == 5000 and line["decode10k"]["bit_exact_vs_cpu_decode"] is True


@pytest.mark.parametrize("channels,n_frames", [(9, 6), (17, 3), (32, 5), (64, 3), (255, 2)])
def test_wide_frames_encode_and_decode(gpu, channels, n_frames):
    """More than eight channels -- up to the 255 the .sela header's field carries -- through k_decode_frames_wide:
    encode and decode on the device against the oracle, and the same frames through the host-pointer pipeline."""
    from sela_amd import capi, codec

    assert capi.lib().sela_hip_decode_max_channels() == 255
    o = oracle()
    pcm = synth_frames(n_frames, channels, 300 + channels)
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=8)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
    ref_back, _ = o.decode_frames(frames, offsets, channels, threads=8)
    assert np.array_equal(_decode(gpu, frames, offsets, channels), ref_back)
    assert np.array_equal(codec.decode_host(frames, offsets, channels), ref_back)
    h_frames, h_offsets = codec.encode_host(pcm)
    assert np.array_equal(h_offsets, ref_offsets) and np.array_equal(h_frames, ref_frames)


def test_wide_frames_with_difference_subframes_and_long_streams(gpu):
    """Hand-built 12-channel frames no encoder writes: difference subframes whose parents lie in another round of the
    eight waves (before and behind them), subframes out of channel order, a Rice stream beyond the on-chip plan (the
    serial parse into the workspace), a channel nobody delivers (silence + EFORMAT) -- against the oracle's decoder."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(33)
    ch = 12

    def sub(channel, typ, parent, scale=300):
        q = rng.integers(-20, 20, size=int(rng.integers(1, 30)))
        return (channel, typ, parent, q, rng.integers(-scale, scale, size=2048))

    order = [3, 0, 1, 2, 4, 5, 6, 7, 11, 9, 10, 8]
    types = {1: (1, 10), 8: (1, 0), 5: (1, 4)}  # channel -> (type, parent): parents in the other round, before and behind
    frames = []
    for _ in range(3):
        subs = [sub(c, *types.get(c, (0, c))) for c in order]
        frames.append(_build_frame(subs))
    # one subframe with residues so wide that its stream exceeds the plan's 1072 words
    subs = [sub(c, *types.get(c, (0, c)), scale=(1 << 20) if c == 6 else 300) for c in order]
    frames.append(_build_frame(subs))
    stream = np.frombuffer(b"".join(frames), np.uint8).copy()
    offsets = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    got = _decode(gpu, stream, offsets, ch)
    for i, f in enumerate(frames):
        want, used = o.frame_decode(f, ch)
        assert used == len(f)
        assert np.array_equal(got[i], want), i
    assert np.array_equal(codec.decode_host(stream, offsets, ch), got)
    # a frame that delivers channel 2 twice and channel 7 never
    subs = [sub(2 if c == 7 else c, 0, c) for c in range(ch)]
    bad = _build_frame(subs)
    stream = np.frombuffer(bad, np.uint8).copy()
    with pytest.raises(Exception):
        _decode(gpu, stream, np.array([0, len(bad)], np.uint64), ch)


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


@pytest.mark.parametrize("form", [0, 1])
def test_decoder_recurrence_forms_give_the_same_samples(gpu, kats, form):
    """The decoder walks a subframe's recurrence in one of two forms, chosen by how alone the workgroup will be on its SIMDs
    (sela_decode.hip, synth_steps kVecShift): small launches -- every test batch -- would only ever see one of them.  Forced
    here, both forms go through the decode tests that reach every branch of the synthesis: the reference's golden frames,
    the edge blocks (order 93 and 97: the ring of 128), full-scale difference channels, residues outside 16 bits (the exact
    32-bit form and the switch to it in the middle of a subframe), unary runs of thousands of bits, random valid streams,
    frames of more than eight channels; and one batch of more workgroups than the device holds (both forms in one launch
    is what the size rule gives a 3875-frame batch: test_baseline_configs_by_digest)."""
    import test_gpu_parity as parity
    import test_gpu_round2 as round2
    from sela_amd import capi, codec

    torch = gpu
    lib = capi.lib()
    lib.sela_hip_debug_decode_recurrence(form)
    try:
        parity.test_frame_kats_encode_and_decode(gpu, kats)
        parity.test_block_kats_as_frames(gpu, kats)
        parity.test_extreme_stereo(gpu)
        parity.test_long_unary_runs_round_trip(gpu)
        parity.test_decoder_on_streams_no_encoder_would_write(gpu, kats)
        parity.test_random_batches_match_oracle(gpu, 2, 12, 150)
        parity.test_random_batches_match_oracle(gpu, 6, 14, 20)
        round2.test_segment_parallel_parser_on_hard_streams(gpu, kats)
        round2.test_parser_on_random_valid_streams(gpu, kats)
        test_wide_frames_encode_and_decode(gpu, 32, 3)
        test_wide_frames_with_difference_subframes_and_long_streams(gpu)
        # more workgroups than fit at once, one form throughout
        o = oracle()
        pcm = synth_frames(400, 2, 31)
        blob, offs, _ = o.encode_frames(pcm, threads=8)
        many = np.tile(blob, 10)
        many_offs = np.concatenate([offs[:-1] + np.uint64(i * len(blob)) for i in range(10)] + [np.array([10 * len(blob)], np.uint64)])
        ref_back, _ = o.decode_frames(blob, offs, 2, threads=8)
        dec = codec.Decoder(4000, 2)
        back = dec.decode(torch.from_numpy(many).cuda(), torch.from_numpy(many_offs.astype(np.int64)).cuda(), 4000)
        torch.cuda.synchronize()
        dec.check()
        assert np.array_equal(back.cpu().numpy().reshape(10, 400, 2048, 2), np.broadcast_to(ref_back, (10, 400, 2048, 2)))
    finally:
        lib.sela_hip_debug_decode_recurrence(-1)


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


def test_batch_verbs_on_album_tracks_match_reference_digests(gpu, tmp_path, album_digests):
    """Tracks 0..11 of BASELINE.json configs[3] (four each at 44.1 / 48 / 96 kHz, 66,120 frames) written as WAV files to
    tmpfs, `sela_mi355x -E --devices 0,0` (two workers on the one GPU: the frame space is cut inside track 7, every
    worker reads, codes and writes its own pieces) -> every .sela file's SHA-256 is the one the unmodified reference
    produced (tests/golden/album_digests.json); `-D --devices 0,0` back -> every decoded PCM's SHA-256 likewise."""
    _build()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    import shutil
    import tempfile

    work = tempfile.mkdtemp(dir=scratch)
    try:
        tracks = album_tracks()[:12]
        wavs = []
        for track, rate, frames in tracks:
            pcm = synth_frames_torch(frames, 2, track, device="cuda").cpu().numpy().reshape(-1, 2)
            p = os.path.join(work, f"track{track:02d}.wav")
            _write_wav(p, pcm, rate)
            wavs.append(p)
        enc_dir, dec_dir = os.path.join(work, "enc"), os.path.join(work, "dec")
        os.mkdir(enc_dir), os.mkdir(dec_dir)
        cli = os.path.join(HOST, "sela_mi355x")
        r = subprocess.run([cli, "-E", enc_dir, "--devices", "0,0"] + wavs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        selas = []
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            p = os.path.join(enc_dir, f"track{track:02d}.sela")
            assert os.path.getsize(p) == g["sela_bytes"], track
            assert _sha_file(p) == g["sela_sha256"], track
            selas.append(p)
        r = subprocess.run([cli, "-D", dec_dir, "--devices", "0,0"] + selas, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            with open(os.path.join(dec_dir, f"track{track:02d}.wav"), "rb") as f:
                wav = f.read()
            assert len(wav) == 44 + frames * 2048 * 2 * 2
            assert hashlib.sha256(wav[44:]).hexdigest() == g["decoded_sha256"], track
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_batch_verbs_small_tracks_in_one_job_and_three_workers(tmp_path):
    """Many small tracks (they are read into one buffer and coded as one job), a track shorter than a frame, two channel
    counts, three workers on one device: every file equals what `-e` / `-d` write for it alone."""
    _build()
    cli = os.path.join(HOST, "sela_mi355x")
    specs = [("a", 2, 44100, 9 * 2048 + 777), ("b", 2, 48000, 4 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100),
             ("e", 2, 44100, 300 * 2048), ("f", 1, 44100, 5 * 2048), ("g", 2, 44100, 2048), ("h", 2, 44100, 37 * 2048 + 1), ("i", 2, 8000, 1500 * 2048)]
    wavs = []
    for k, (name, ch, rate, n) in enumerate(specs):
        p = tmp_path / f"{name}.wav"
        _write_wav(p, synth_pcm(n, ch, 120 + k), rate)
        wavs.append(p)
    single, batch, back1, back3 = (tmp_path / d for d in ("single", "batch", "back1", "back3"))
    for d in (single, batch, back1, back3):
        d.mkdir()
    for w in wavs:
        r = subprocess.run([cli, "-e", str(w), str(single / (w.stem + ".sela"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-E", str(batch), "--devices", "0,0,0", "--io-threads", "5"] + [str(w) for w in wavs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for w in wavs:
        name = w.stem + ".sela"
        assert (batch / name).read_bytes() == (single / name).read_bytes(), name
    selas = [batch / (w.stem + ".sela") for w in wavs]
    for s in selas:
        r = subprocess.run([cli, "-d", str(s), str(back1 / (s.stem + ".wav"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-D", str(back3), "--devices", "0,0,0"] + [str(s) for s in selas], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for s in selas:
        name = s.stem + ".wav"
        assert (back3 / name).read_bytes() == (back1 / name).read_bytes(), name
    # a stream that stops early (a frame without its sync word): the batch decoder writes what the file decoder writes
    raw = bytearray((batch / "e.sela").read_bytes())
    offs = 15
    for _ in range(120):  # walk 120 frames to find the 121st's sync word
        assert raw[offs: offs + 4] == bytes([0x00, 0xFF, 0x55, 0xAA])
        p = offs + 4
        for _c in range(2):
            cw = raw[p + 4] | (raw[p + 5] << 8)
            p += 7 + 4 * cw
            rw = raw[p + 1] | (raw[p + 2] << 8)
            p += 5 + 4 * rw
        offs = p
    raw[offs] ^= 0xFF
    (tmp_path / "cut.sela").write_bytes(bytes(raw))
    r = subprocess.run([cli, "-d", str(tmp_path / "cut.sela"), str(tmp_path / "cut1.wav")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cutdir = tmp_path / "cutdir"
    cutdir.mkdir()
    r = subprocess.run([cli, "-D", str(cutdir), "--devices", "0,0", str(tmp_path / "cut.sela")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (cutdir / "cut.wav").read_bytes() == (tmp_path / "cut1.wav").read_bytes()
    assert len((tmp_path / "cut1.wav").read_bytes()) == 44 + 120 * 2048 * 2 * 2


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
