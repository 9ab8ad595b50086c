"""GPU parity tests added in round 2: the fused decoder's parser on the streams that stress it, the streaming
host-pointer jobs, file-level SHA-256 pins written by the reference's own file verbs, BASELINE.json
configs[3] (the 100-track album) at full size, the C++ multi-GPU dispatcher and the RCCL size exchange."""
import ctypes as C
import hashlib
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
from test_gpu_parity import _build_frame, _decode, _encode, _rice_words, gpu  # noqa: F401  (fixture + helpers)
from test_host_cpp import HOST, _build, _write_wav

pytestmark = pytest.mark.gpu


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def test_plain_fir_branch_of_the_product_kernel(gpu):
    """sela_hip_debug_force_plain_fir sends every block of k_encode_blocks<0> -- the instantiation the bench
    times -- down the branch that predictors beyond the fast FIR's coefficient range would take (16-bit audio
    never does: |a| stays below 2^37).  Same bytes either way."""
    from sela_amd import capi

    o = oracle()
    pcm = np.concatenate([synth_frames(40, 2, 51), np.zeros((1, 2048, 2), np.int16)])
    want_frames, want_offsets, _ = o.encode_frames(pcm, threads=4)
    lib = capi.lib()
    lib.sela_hip_debug_force_plain_fir(1)
    try:
        frames, offsets, _, _ = _encode(gpu, pcm)
    finally:
        lib.sela_hip_debug_force_plain_fir(0)
    assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames)
    frames, offsets, _, _ = _encode(gpu, pcm)
    assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames)


def _decode_frames_vs_oracle(gpu, frames, channels=1):
    o = oracle()
    stream = np.frombuffer(b"".join(frames), np.uint8).copy()
    offsets = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    got = _decode(gpu, stream, offsets, channels)
    for i, f in enumerate(frames):
        want, used = o.frame_decode(f, channels)
        assert used == len(f)
        assert np.array_equal(got[i], want), i


def test_segment_parallel_parser_on_hard_streams(gpu, kats):
    """Hand-built subframes that stay inside the decoder's LDS plan (<= 1072 aligned words), so they are
    parsed by the segment-parallel path: unary-only coding, residue streams shorter than the wave has zones,
    a unary run longer than several zones, orders 0 / 1 / 100, streams of exactly the plan's capacity and one
    word over it (generic mode), all against the oracle's decoder."""
    rng = np.random.default_rng(77)
    q_sine = kats["blk/sine_deg/q"]
    q_noise = kats["blk/white_fullscale/q"]
    spikes = np.zeros(2048, np.int64)
    spikes[[3, 700, 701, 1999]] = [12000, -9000, 9000, 4000]  # k = 0: runs of thousands of ones among single zeros
    cases = [
        _build_frame([(0, 0, 0, [0], np.zeros(2048, np.int32))]),                       # 64 words: one word per zone
        _build_frame([(0, 0, 0, [], rng.integers(-3, 4, 2048))]),                        # order 0, no coefficient words
        _build_frame([(0, 0, 0, q_sine, rng.integers(-1, 2, 2048))], res_k=0),            # unary only, inside the plan
        _build_frame([(0, 0, 0, q_sine, spikes)], res_k=0),                               # runs far longer than a zone
        _build_frame([(0, 0, 0, q_noise, rng.integers(-2, 3, 2048))], res_k=1),
        _build_frame([(0, 0, 0, np.full(100, -64, np.int32), rng.integers(-100, 100, 2048))]),  # longest coefficient stream
        _build_frame([(0, 0, 0, q_sine, rng.integers(-30000, 30000, 2048))]),             # ~16 bits per residue: ~1100 words (beyond the plan since round 3: serial parse)
        _build_frame([(0, 0, 0, q_noise[:61], rng.integers(-900, 900, 2048))], res_k=14),  # remainder-heavy: slow resynchronisation
    ]
    # exactly at the plan's capacity and one word over: pad the residue stream with zero words (a decoder
    # ignores what follows the 2048th value)
    base = rng.integers(-200, 200, 2048)
    rk = 7
    rw = _rice_words(base, rk)
    ck, cw = oracle().rice_encode(np.asarray(q_sine, np.int32))
    for total in (1072, 1073):
        pad = total - (len(cw) + 2 + len(rw))
        assert pad > 0
        words = np.concatenate([rw, np.zeros(pad, np.uint32)])
        cases.append(struct.pack("<I", 0xAA55FF00) + struct.pack("<BBBBHB", 0, 0, 0, ck, len(cw), len(q_sine)) + cw.astype("<u4").tobytes()
                     + struct.pack("<BHH", rk, len(words), 2048) + words.astype("<u4").tobytes())
    _decode_frames_vs_oracle(gpu, cases)
    # stereo: a fast-plan subframe next to one that is not sends the whole frame through generic mode
    st = _build_frame([(0, 0, 0, q_sine, rng.integers(-50, 50, 2048)), (1, 1, 0, q_noise, rng.integers(-(1 << 20), 1 << 20, 2048))])
    _decode_frames_vs_oracle(gpu, [st], channels=2)


def test_truncated_rice_stream_is_reported(gpu, kats):
    """A residue stream that ends before its 2048th value: zeros behind the end, SELA_HIP_EFORMAT at the C ABI
    (the reference reads past its vector here)."""
    from sela_amd import capi, codec

    rng = np.random.default_rng(8)
    r = rng.integers(-300, 300, 2048)
    rk = 8
    words = _rice_words(r, rk)
    q = kats["blk/sine_deg/q"]
    ck, cw = oracle().rice_encode(np.asarray(q, np.int32))
    cut = words[: len(words) // 2]
    frame = (struct.pack("<I", 0xAA55FF00) + struct.pack("<BBBBHB", 0, 0, 0, ck, len(cw), len(q)) + cw.astype("<u4").tobytes()
             + struct.pack("<BHH", rk, len(cut), 2048) + cut.astype("<u4").tobytes())
    with pytest.raises(capi.SelaHipError) as e:
        codec.decode_host(np.frombuffer(frame, np.uint8).copy(), np.array([0, len(frame)], np.uint64), 1)
    assert e.value.code == -5


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


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_cli_files_match_reference_file_digests(tmp_path, file_digests, label):
    """`sela_mi355x -e` / `-d` (the streaming file-to-file path) write byte-identical files -- headers, dropped
    tail and all -- to the ones the reference's sela::Encoder + SelaFile::writeToFile and sela::Decoder +
    WavFile::writeToFile wrote (tests/golden/file_digests.json, made by oracle/_ref's ref_encode_file /
    ref_decode_file); so do the batch verbs -E / -D."""
    _build()
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    assert _sha_file(wav) == d["wav_sha256"], "input drifted"
    cli = os.path.join(HOST, "sela_mi355x")
    r = subprocess.run([cli, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([cli, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]
    batch = tmp_path / "batch"
    batch.mkdir()
    r = subprocess.run([cli, "-E", str(batch), str(wav)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _sha_file(batch / "in.sela") == d["sela_sha256"]
    r = subprocess.run([cli, "-D", str(batch), str(batch / "in.sela")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _sha_file(batch / "in.wav") == d["decoded_wav_sha256"]


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


def test_cli_verbs_give_the_same_files_every_time(tmp_path, file_digests):
    """The host pipeline is copies, kernels and host hand-overs on four streams: a hand-over that is only almost right
    shows once in a while, not every time.  Six passes of each file verb over the 3-minute track, in fresh processes,
    every output against the reference's SHA-256."""
    _build()
    d = file_digests["config1_stereo_3min"]
    wav = tmp_path / "in.wav"
    _write_wav(wav, synth_pcm(d["samples_per_channel"], d["channels"], d["track"]), d["sample_rate"])
    cli = os.path.join(HOST, "sela_mi355x")
    for rep in range(6):
        sela, back, batch = tmp_path / f"out{rep}.sela", tmp_path / f"back{rep}.wav", tmp_path / f"batch{rep}"
        batch.mkdir()
        r = subprocess.run([cli, "-e", str(wav), str(sela)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(sela) == d["sela_sha256"], rep
        r = subprocess.run([cli, "-E", str(batch), str(wav)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(batch / "in.sela") == d["sela_sha256"], rep
        r = subprocess.run([cli, "-d", str(sela), str(back)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(back) == d["decoded_wav_sha256"], rep
        for f in (sela, back, batch / "in.sela"):
            os.remove(f)


def test_multi_gpu_dispatcher_with_two_workers_on_one_device(tmp_path):
    """sela::encodeBatch / decodeBatch with two host threads, both bound to device 0 (`--devices 0,0`): the
    flattened frame space is cut in two contiguous halves -- inside a track -- and the files must come out
    byte-identical to one worker's (src/sela/encoder.cpp:58-73 is the partition this replaces)."""
    _build()
    cli = os.path.join(HOST, "sela_mi355x")
    specs = [("a", 2, 44100, 9 * 2048 + 777), ("b", 2, 48000, 4 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100),
             ("e", 2, 44100, 1100 * 2048), ("f", 1, 44100, 5 * 2048)]
    wavs = []
    for name, ch, rate, n in specs:
        p = tmp_path / f"{name}.wav"
        _write_wav(p, synth_pcm(n, ch, 80 + len(wavs)), rate)
        wavs.append(p)
    one, two, three, back1, back2 = (tmp_path / d for d in ("one", "two", "three", "back1", "back2"))
    for d in (one, two, three, back1, back2):
        d.mkdir()
    for out_dir, devs in ((one, "0"), (two, "0,0"), (three, "0,0,0")):
        r = subprocess.run([cli, "-E", str(out_dir), "--devices", devs] + [str(w) for w in wavs], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    for w in wavs:
        name = w.stem + ".sela"
        assert (two / name).read_bytes() == (one / name).read_bytes(), name
        assert (three / name).read_bytes() == (one / name).read_bytes(), name
    selas = [one / (w.stem + ".sela") for w in wavs]
    for out_dir, devs in ((back1, "0"), (back2, "0,0")):
        r = subprocess.run([cli, "-D", str(out_dir), "--devices", devs] + [str(s) for s in selas], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    for s in selas:
        name = s.stem + ".wav"
        assert (back2 / name).read_bytes() == (back1 / name).read_bytes(), name
    # a device that does not exist is an error, not a silent single-GPU run
    r = subprocess.run([cli, "-E", str(two), "--devices", "0,99", str(wavs[0])], capture_output=True, text=True)
    assert r.returncode == 1 and "device" in r.stderr


def test_album_by_digest(gpu, album_digests):
    """BASELINE.json configs[3] at full size: the 100-track album (34 / 33 / 33 tracks at 44.1 / 48 / 96 kHz,
    549,365 stereo frames), encoded in batches of <= 65,536 frames, every track's .sela FILE (15-byte header +
    frames) and decoded PCM against SHA-256s computed with the unmodified reference
    (tests/golden/album_digests.json)."""
    import torch

    from sela_amd import codec
    from sela_amd.sharding import sela_header

    tracks = album_tracks()
    assert sum(f for _, _, f in tracks) == album_digests["n_frames"] == 549365
    cap = 65536
    enc = codec.Encoder(cap, 2)
    dec = codec.Decoder(cap, 2)
    batches, cur, cur_frames = [], [], 0
    for t in tracks:
        if cur and cur_frames + t[2] > cap:
            batches.append(cur)
            cur, cur_frames = [], 0
        cur.append(t)
        cur_frames += t[2]
    batches.append(cur)
    total = hashlib.sha256()
    lossy = 0
    for batch in batches:
        pcm = torch.cat([synth_frames_torch(frames, 2, track, device="cuda") for track, _, frames in batch])
        out = enc.encode(pcm)
        back = dec.decode(out.frames, out.offsets, pcm.shape[0])
        torch.cuda.synchronize()
        out.check()
        dec.check()
        blob, offs = out.to_host()
        back_host = back.cpu().numpy()
        f0 = 0
        for track, rate, frames in batch:
            d = album_digests["tracks"][track]
            assert d["track"] == track and d["n_frames"] == frames and d["sample_rate"] == rate
            b0, b1 = int(offs[f0]), int(offs[f0 + frames])
            sha = hashlib.sha256(sela_header(rate, 16, 2, frames) + blob[b0:b1].tobytes()).hexdigest()
            assert 15 + b1 - b0 == d["sela_bytes"] and sha == d["sela_sha256"], track
            assert hashlib.sha256(back_host[f0: f0 + frames].tobytes()).hexdigest() == d["decoded_sha256"], track
            total.update(bytes.fromhex(sha))
            f0 += frames
        lossy += int((back != pcm).reshape(pcm.shape[0], -1).any(dim=1).sum().item())
    assert total.hexdigest() == album_digests["sha256_of_track_sela_sha256s"]
    assert lossy == sum(t["lossy_frames"] for t in album_digests["tracks"])  # the reference's own rounding quirk, frame for frame


def test_sharded_encode_over_rccl_world_of_one(gpu, tmp_path):
    """sharding.encode_sharded with the `nccl` (= RCCL) backend on this GPU, world size 1: the N > 1 code path
    of bench.py and of a per-rank deployment, layout identical to the one-rank layout."""
    import torch
    import torch.distributed as dist

    from sela_amd import codec, sharding

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    if os.path.isdir("/sys/class/net/lo"): # (the bootstrap sockets over the loopback interface: the container's hostname may not resolve)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    import datetime

    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", 0))
    try:
        track_frames = [40, 0, 25]
        pcm = torch.cat([synth_frames_torch(n, 2, 90 + i, device="cuda") for i, n in enumerate(track_frames) if n])
        enc = codec.Encoder(pcm.shape[0], 2)
        out, layout = sharding.encode_sharded(pcm, pcm.shape[0], 0, 1, enc)
        # force the collective itself (encode_sharded short-cuts world == 1): one all-gather of the sizes
        sizes = (out.offsets[1:] - out.offsets[:-1]).to(torch.int64)
        gathered = torch.empty_like(sizes)
        dist.all_gather_into_tensor(gathered, sizes)
        torch.cuda.synchronize()
        frames, offsets = out.to_host()
        assert np.array_equal(gathered.cpu().numpy().astype(np.uint64), layout.frame_sizes)
        assert np.array_equal(layout.frame_offsets, offsets)
        ref_frames, ref_offsets, _ = oracle().encode_frames(pcm.cpu().numpy(), threads=8)
        assert np.array_equal(frames, ref_frames) and np.array_equal(offsets, ref_offsets)
        pieces = sharding.rank_track_pieces(layout, track_frames, 0)
        assert [(p.track, p.first_frame, p.n_frames) for p in pieces] == [(0, 0, 40), (2, 0, 25)]
        assert pieces[1].file_offset == sharding.SELA_HEADER_BYTES and pieces[1].n_bytes == int(offsets[65] - offsets[40])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("channels", [1, 2, 3])
def test_mean_workers_give_the_same_means(gpu, channels):
    """sela_hip_debug_mean_workers(self_blocks): all but the first `self_blocks` blocks of a launch take their
    2048-term sequential mean from the "mean worker" workgroups (lane = block) instead of walking the chain in
    their own wave.  Small batches never do by default, so the hook forces it: the means (bit patterns), every
    later intermediate and the frame bytes must not change.  Large batches (the configs[1] digest test, the
    10k-frame test, the album) take the worker path without the hook."""
    from sela_amd import capi
    from test_gpu_parity import _bits

    o = oracle()
    pcm = synth_frames(70, channels, 90 + channels)
    want_frames, want_offsets, _ = o.encode_frames(pcm, threads=4)
    lib = capi.lib()
    n_sig = 3 if channels == 2 else channels
    for self_blocks in (0, 8, 40):
        lib.sela_hip_debug_mean_workers(self_blocks)
        try:
            frames, offsets, enc, _ = _encode(gpu, pcm, with_trace=True)
            plain_frames, plain_offsets, _, _ = _encode(gpu, pcm)  # the product instantiation
        finally:
            lib.sela_hip_debug_mean_workers(-1)
        assert np.array_equal(offsets, want_offsets) and np.array_equal(frames, want_frames), self_blocks
        assert np.array_equal(plain_offsets, want_offsets) and np.array_equal(plain_frames, want_frames), self_blocks
        traces = enc.traces(pcm.shape[0])
        for f in (0, 1, 7, 33, 69):
            for sig in range(n_sig):
                s = (pcm[f, :, 0].astype(np.int32) - pcm[f, :, 1]) if (channels == 2 and sig == 2) else pcm[f, :, sig].astype(np.int32)
                _, _, _, _, tr, _ = o.lpc_analyze(s, with_trace=True)
                g = traces[f * n_sig + sig]
                assert np.array_equal(_bits(g.mean), _bits(tr.mean)), (self_blocks, f, sig)
                assert np.array_equal(_bits(list(g.ac)), _bits(list(tr.ac))), (self_blocks, f, sig)


def test_parser_on_random_valid_streams(gpu, kats):
    """300 hand-built mono subframes with random Rice parameters (0..15 for the residues, whatever the oracle picks
    for the coefficients), random orders and residue distributions from near-silence to spiky: all valid streams
    inside the decoder's LDS plan, every decoded sample against the oracle's decoder.  (What the segment-parallel
    parser must get right is where codewords start; these streams vary zone sizes, resynchronisation distances and
    chain shapes far beyond what the encoder's own output does.)"""
    rng = np.random.default_rng(2024)
    q_pool = [kats["blk/sine_deg/q"], kats["blk/white_fullscale/q"], np.zeros(1, np.int32), kats["blk/square_p64/q"]]
    frames = []
    while len(frames) < 300:
        k = int(rng.integers(0, 16))
        kind = int(rng.integers(0, 4))
        scale = (1 << k) * float(rng.choice([0.3, 1.0, 2.5]))
        if kind == 0:
            r = rng.normal(0, scale + 0.5, 2048)
        elif kind == 1:
            r = rng.laplace(0, scale + 0.5, 2048)
        elif kind == 2:
            r = np.where(rng.random(2048) < 0.02, rng.normal(0, 40 * (scale + 1), 2048), rng.normal(0, 0.3 * scale + 0.2, 2048))
        else:
            r = rng.integers(-int(scale) - 1, int(scale) + 2, 2048).astype(np.float64)
        r = np.clip(np.round(r), -(1 << 20), 1 << 20).astype(np.int64)
        q = q_pool[int(rng.integers(0, len(q_pool)))]
        q = np.asarray(q[: int(rng.integers(1, len(q) + 1))], np.int32)
        words = _rice_words(r, k)
        ck, cw = oracle().rice_encode(q)
        if len(cw) + 2 + len(words) > 1072:  # keep it inside the fast plan (generic mode has its own tests)
            continue
        frames.append(struct.pack("<I", 0xAA55FF00) + struct.pack("<BBBBHB", 0, 0, 0, ck, len(cw), len(q)) + cw.astype("<u4").tobytes()
                      + struct.pack("<BHH", k, len(words), 2048) + words.astype("<u4").tobytes())
    _decode_frames_vs_oracle(gpu, frames)
