"""GPU parity tests of round 4: k_encode_teams (the analysis of several blocks side by side in one wave) against the
CPU oracle -- FP64 intermediates as bit patterns, frame bytes, and BASELINE configs[1] by the reference's digest -- for
both team widths, forced through the debug hook (launch_encode picks one by launch size otherwise)."""
import hashlib
import os

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import synth_frames
from test_gpu_parity import _bits, _encode, _kat_block_frames, gpu  # noqa: F401  (gpu: fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[8, 16], ids=["teams_of_8", "teams_of_16"])
def teams(request, gpu):  # noqa: F811
    from sela_amd import capi

    capi.lib().sela_hip_debug_encode_teams(request.param)
    yield request.param
    capi.lib().sela_hip_debug_encode_teams(-1)


def _hard_blocks():
    """Blocks at the corners of the analysis: all zero (0/0 -> NaN everywhere), constant, full scale, one impulse at either
    end, alternating extremes, a ramp."""
    z = np.zeros(2048, np.int16)
    rows = [z, z + np.int16(7), np.full(2048, -32768, np.int16), np.full(2048, 32767, np.int16)]
    a = z.copy(); a[0] = 32767; rows.append(a)
    a = z.copy(); a[2047] = -32768; rows.append(a)
    a = z.copy(); a[::2] = 32767; a[1::2] = -32768; rows.append(a)
    rows.append((np.arange(2048) * 31 - 32768).astype(np.int16))
    rng = np.random.default_rng(11)
    rows.append(rng.integers(-32768, 32768, 2048).astype(np.int16))
    rows.append(rng.integers(-3, 4, 2048).astype(np.int16))
    return np.stack(rows)[:, :, None]


def test_team_analysis_stages_bit_exact(gpu, kats, teams):  # noqa: F811
    """mean / autocorrelation / reflection coefficients / order / q / a of k_encode_teams<1, P> against the oracle's trace:
    the KAT blocks, the corner blocks, 27 stereo frames (a last wave with teams to spare), 5 three-channel frames."""
    o = oracle()
    names, mono = _kat_block_frames(kats)
    for pcm in [mono, _hard_blocks(), synth_frames(27, 2, 3), synth_frames(5, 3, 4)]:
        frames, offsets, enc, out = _encode(gpu, pcm, with_trace=True)
        ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=4)
        assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
        traces = enc.traces(pcm.shape[0])
        ch = pcm.shape[2]
        n_sig = 3 if ch == 2 else ch
        for f in range(pcm.shape[0]):
            for sig in range(n_sig):
                s = (pcm[f, :, 0].astype(np.int32) - pcm[f, :, 1]) if (ch == 2 and sig == 2) else pcm[f, :, sig].astype(np.int32)
                order, q, r, a, tr, _ = o.lpc_analyze(s, with_trace=True)
                g = traces[f * n_sig + sig]
                ctx = (teams, f, sig)
                assert np.array_equal(_bits(g.mean), _bits(tr.mean)), ctx
                assert np.array_equal(_bits(list(g.ac)), _bits(list(tr.ac))), ctx
                assert np.array_equal(_bits(list(g.k)), _bits(list(tr.k))), ctx
                assert g.order == order, ctx
                assert list(g.q)[:order] == q.tolist(), ctx
                assert list(g.a)[: order + 1] == a.tolist(), ctx
                ck, cw = o.rice_encode(q)
                rk, rw = o.rice_encode(r)
                assert (g.coef_k, g.coef_words, g.res_k, g.res_words) == (ck, len(cw), rk, len(rw)), ctx
                assert g.flags == 0


@pytest.mark.parametrize("n_frames,channels", [(1, 2), (7, 2), (8, 2), (9, 2), (63, 2), (65, 2), (215, 1), (130, 2), (3, 5), (40, 4)])
def test_team_product_kernel_frames(gpu, teams, n_frames, channels):  # noqa: F811
    """The product instantiation k_encode_teams<0, P>: frame bytes and offsets against the oracle at batch sizes around the
    waves' and the launch's granules (B frames per wave, 8 waves per round of the XCDs), mono and odd channel counts."""
    pcm = synth_frames(n_frames, channels, 40 + n_frames)
    frames, offsets, enc, out = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = oracle().encode_frames(pcm, threads=8)
    assert np.array_equal(offsets, ref_offsets)
    assert np.array_equal(frames, ref_frames)


@pytest.mark.parametrize("label", ["config1_stereo_3min", "config2_1000_frames"])
def test_team_kernels_on_baseline_configs_by_digest(gpu, digests, teams, label):  # noqa: F811
    """BASELINE configs[1] and [2] through k_encode_teams: SHA-256 of the frame stream and the offsets against the digests
    computed with the unmodified reference (tests/golden/digests.json)."""
    d = digests[label]
    pcm = synth_frames(d["n_frames"], d["channels"], d["track"])
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == d["pcm_sha256"]
    frames, offsets, _, out = _encode(gpu, pcm)
    assert len(frames) == d["frames_blob_bytes"]
    assert hashlib.sha256(frames.tobytes()).hexdigest() == d["frames_blob_sha256"]
    assert hashlib.sha256(offsets.astype("<u8").tobytes()).hexdigest() == d["offsets_sha256"]


def test_team_kernels_plain_fir_branch(gpu, teams):  # noqa: F811
    """sela_hip_debug_force_plain_fir sends every block of k_encode_teams<0, P> down the 64-bit FIR loop that predictors beyond
    the fast FIR's coefficient range take (its predictions go through the block's own slot in global memory -- the same
    words a degenerate block's wide coefficients wait in): same bytes."""
    from sela_amd import capi

    pcm = synth_frames(37, 2, 91)
    lib = capi.lib()
    lib.sela_hip_debug_force_plain_fir(1)
    try:
        frames, offsets, _, _ = _encode(gpu, pcm)
    finally:
        lib.sela_hip_debug_force_plain_fir(0)
    ref_frames, ref_offsets, _ = oracle().encode_frames(pcm, threads=8)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)


def _polyphonic_frames(n, seed):
    """Loud sums of 3..40 sinusoids over a little noise: long predictors with large coefficients at large amplitudes -- the
    blocks on which sum |a[j]| x max |s| passes 2^53, where FP64 multiply-adds of integers stop being exact."""
    rng = np.random.default_rng(seed)
    t = np.arange(2048)
    pcm = np.zeros((n, 2048, 2), np.int16)
    for f in range(n):
        for ch in range(2):
            k = int(rng.integers(3, 40))
            x = sum((30000 / k) * np.sin(2 * np.pi * fr * t / 44100 + ph) for fr, ph in zip(rng.uniform(50, 20000, k), rng.uniform(0, 6.28, k)))
            x = x * rng.choice([1.0, 1.0, 0.2]) + rng.normal(0, rng.choice([0.3, 1, 3]), 2048)
            pcm[f, :, ch] = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return pcm


@pytest.mark.parametrize("team_lanes", [0, 16, 8])
def test_residue_filter_forms(gpu, team_lanes):  # noqa: F811
    """The encoder's residue filter has three forms (sela_encode_tail.inc): one pass of FP64 multiply-adds where that is exact
    (no partial sum can reach 2^53: 2^34 + sum |a[j]| x max |s| < 2^53, decided per block), two passes (the coefficients' low
    20 bits, then the rest) beyond that, the plain 64-bit loop for what neither carries.  On 40 loud polyphonic frames -- the
    oracle confirms that some of their blocks are beyond the one-pass bound and some within -- and on the corner blocks: the
    bytes by the block's own choice, with two passes forced wherever one would do, and with every block down the plain loop,
    against the oracle's, in all three encode kernels."""
    from sela_amd import capi

    o = oracle()
    pcm = _polyphonic_frames(40, 2)
    beyond = within = 0
    for f in range(0, 40, 3):
        l, r = pcm[f, :, 0].astype(np.int32), pcm[f, :, 1].astype(np.int32)
        for sig in (l, r, l - r):
            order, q = o.lpc_analyze(sig)[:2]
            a = np.asarray(o.lpc_coeffs(order, q), dtype=np.int64)
            bound = int(np.abs(a[1:order + 1]).sum()) * int(np.abs(sig).max()) + (1 << 34)
            beyond += bound >= (1 << 53)
            within += bound < (1 << 53)
    assert beyond >= 3 and within >= 3, (beyond, within)
    lib = capi.lib()
    lib.sela_hip_debug_encode_teams(team_lanes)
    try:
        for data in (pcm, np.repeat(_hard_blocks(), 2, axis=2)):
            ref_frames, ref_offsets, _ = o.encode_frames(data, threads=8)
            for form in (0, 2, 1):
                lib.sela_hip_debug_force_plain_fir(form)
                frames, offsets, _, _ = _encode(gpu, data)
                assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames), (team_lanes, form)
    finally:
        lib.sela_hip_debug_force_plain_fir(0)
        lib.sela_hip_debug_encode_teams(-1)


@pytest.mark.parametrize("team_lanes", [0, 16, 8])
def test_the_losing_stereo_candidate_leaves_its_slot_unwritten(gpu, team_lanes):  # noqa: F811
    """Of an exactly-stereo frame's second channel and its difference signal the frame keeps the smaller; the blocks tell each
    other their sizes and the one that knows it has lost does not write its slot (sela_encode_tail.inc).  The workspace --
    slots, metadata and the words the sizes travel in -- is filled with a pattern first: the bytes equal the oracle's with
    the hand-over on, and off (sela_hip_debug_keep_both_candidates); no first channel's and no winner's slot is ever left
    unwritten, never both of a pair; with the hook nothing is skipped; and the hand-over does skip a fair share of the losers
    (how many depends on which of the two waves gets there first)."""
    import torch

    from sela_amd import capi, codec

    n = 256
    pcm = synth_frames(n, 2, 17)
    ref_frames, ref_offsets, _ = oracle().encode_frames(pcm, threads=8)
    lib = capi.lib()
    lib.sela_hip_debug_encode_teams(team_lanes)
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    try:
        for keep_both in (0, 1):
            lib.sela_hip_debug_keep_both_candidates(keep_both)
            enc = codec.Encoder(n, 2)
            enc.workspace.fill_(0xA5)
            out = enc.encode(d_pcm)
            torch.cuda.synchronize()
            frames, offsets = out.to_host()
            assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames), (team_lanes, keep_both)
            ws = enc.workspace.cpu().numpy()
            base = (-enc.workspace.data_ptr()) % 256
            meta = ws[base: base + n * 3 * 8].view(np.uint16).reshape(n * 3, 4) # order|coef_k, res_k|flags, coef_words, res_words
            words = meta[:, 2].astype(np.int64) + meta[:, 3]
            slots_at = base + (n * 3 * 8 + 255) // 256 * 256
            slots = ws[slots_at: slots_at + n * 3 * 2240 * 4].view(np.uint32).reshape(n * 3, 2240)
            unwritten = (slots[:, 32] == 0xA5A5A5A5) & (slots[:, 33] == 0xA5A5A5A5) # (the first residue words)
            assert not unwritten[0::3].any()
            second, diff = unwritten[1::3], unwritten[2::3]
            assert not (second & diff).any()
            diff_wins = words[2::3] < words[1::3]
            assert not (second & ~diff_wins).any() and not (diff & diff_wins).any() # only losers
            skipped = int(second.sum() + diff.sum())
            if keep_both:
                assert skipped == 0
            else:
                assert skipped >= n // 4, skipped
    finally:
        lib.sela_hip_debug_keep_both_candidates(0)
        lib.sela_hip_debug_encode_teams(-1)


@pytest.mark.parametrize("channels,n_frames", [(9, 11), (64, 3), (255, 2)])
def test_team_kernels_many_channels(gpu, teams, channels, n_frames):  # noqa: F811
    """One signal per channel, up to the 255 the header's field carries: a wave takes one signal of B consecutive frames, so
    with few frames most teams of a wave shadow the last frame and must leave nothing behind."""
    pcm = synth_frames(n_frames, channels, 500 + channels)
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = oracle().encode_frames(pcm, threads=8)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)


def test_team_kernels_hostile_audio(gpu, teams):  # noqa: F811
    """Full-scale white noise, silence, DC, a square wave at the Nyquist rate and small noise, interleaved over 300 stereo
    frames (silent and constant blocks make 0 / 0 = NaN autocorrelations and orders of 1; the square wave drives the
    reflection coefficients to +-1): frames and decoded samples against the oracle."""
    from test_gpu_parity import _decode

    rng = np.random.default_rng(77)
    pcm = np.zeros((300, 2048, 2), np.int16)
    for f in range(300):
        kind = f % 6
        if kind == 0:
            pcm[f] = rng.integers(-32768, 32768, (2048, 2))
        elif kind == 1:
            pcm[f] = 0
        elif kind == 2:
            pcm[f, :, 0], pcm[f, :, 1] = 12345, -32768
        elif kind == 3:
            pcm[f, ::2], pcm[f, 1::2] = 32767, -32768
        elif kind == 4:
            pcm[f] = rng.integers(-2, 3, (2048, 2))
        else:
            pcm[f, :, 0] = rng.integers(-32768, 32768, 2048)
            pcm[f, :, 1] = pcm[f, :, 0]  # (the difference signal is silence)
    o = oracle()
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=16)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
    ref_back, _ = o.decode_frames(ref_frames, ref_offsets, 2, threads=16)
    assert np.array_equal(_decode(gpu, frames, offsets, 2), ref_back)


def test_the_library_picks_a_team_kernel_by_launch_size(gpu):  # noqa: F811
    """launch_encode's choice (team_lanes_for, a model of the three kernels' times in waves per SIMD of the launch's last
    round): k_encode_blocks for small launches and just behind a full round of team waves (4200 stereo frames = one fill of
    teams of 16 and a few waves), teams of 16 around one and one and a half fills, teams of 8 where their rounds are full
    or the launch is large; the hook overrides it; mono counts blocks, not frames."""
    from sela_amd import capi

    lib = capi.lib()
    lib.sela_hip_debug_encode_teams(-1)
    picks = {n: lib.sela_hip_debug_encode_kernel(n, 2) for n in (1, 1000, 3000, 3875, 4096, 4200, 5000, 8192, 9000, 16384, 61041)}
    assert picks == {1: 0, 1000: 0, 3000: 0, 3875: 16, 4096: 16, 4200: 0, 5000: 16, 8192: 8, 9000: 16, 16384: 8, 61041: 8}, picks
    assert [lib.sela_hip_debug_encode_kernel(n, 1) for n in (3000, 11625, 12288, 49152)] == [0, 16, 16, 8] # (mono: 3875 stereo frames' blocks)
    lib.sela_hip_debug_encode_teams(8)
    assert lib.sela_hip_debug_encode_kernel(1, 2) == 8
    lib.sela_hip_debug_encode_teams(-1)


# ---- the binding of INTEGRATION.md section 2, compiled against the reference's own headers (oracle/binding/) ----------
BOUND = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "sela_ref_bound")


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_reference_classes_bound_to_the_library_write_the_reference_files(tmp_path, file_digests, label):
    """oracle/_ref/sela_ref_bound = the reference's own sela::Encoder / sela::Decoder class declarations and its unmodified
    file classes (src/file/*.cpp), with processFrames() replaced by one call into libsela_hip.so (oracle/binding/bound.cpp,
    built by `make -C oracle bound` in the build container; src/lpc, src/rice, src/frame are not linked).  Its -e / -d
    write the very files the unmodified reference wrote (tests/golden/file_digests.json)."""
    import subprocess

    from sela_amd.synth import synth_pcm
    from test_gpu_round2 import _sha_file, _write_wav

    if not os.path.exists(BOUND):
        pytest.fail("oracle/_ref/sela_ref_bound is missing: run `make -C oracle bound` in the build container (it travels to the GPU box)")
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    assert _sha_file(wav) == d["wav_sha256"], "input drifted"
    r = subprocess.run([BOUND, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([BOUND, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]


MAIN_ON_HOST = os.path.join(os.path.dirname(BOUND), "sela_ref_main_on_host")


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_reference_main_compiled_against_this_host_writes_the_reference_files(tmp_path, file_digests, label):
    """The other direction of the boundary: oracle/_ref/sela_ref_main_on_host = the reference's UNCHANGED src/main.cpp
    (/root/reference/src/main.cpp:29-51: `sela::Encoder encoder = sela::Encoder(inputFile); file::SelaFile selaFile =
    encoder.process(); selaFile.writeToFile(outputFile);` ...) compiled against THIS repo's host classes through the
    reference's own header paths (host/compat/include, `make -C oracle main_on_host`).  Its -e / -d write the reference's
    files; its -p plays through sela::Player's default sink (the packets, as they are, on standard output)."""
    import subprocess

    from sela_amd.synth import synth_pcm
    from test_gpu_round2 import _sha_file, _write_wav

    if not os.path.exists(MAIN_ON_HOST):
        pytest.fail("oracle/_ref/sela_ref_main_on_host is missing: run `make -C oracle main_on_host` in the build container")
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    r = subprocess.run([MAIN_ON_HOST, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0 and "Encoding: " in r.stdout, (r.stdout, r.stderr)
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([MAIN_ON_HOST, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]
    if label == "stereo_48k_tail":  # the player's feed: banner lines, then every decoded frame's interleaved int16 bytes
        r = subprocess.run([MAIN_ON_HOST, "-p", str(sela)], capture_output=True)
        assert r.returncode == 0, r.stderr
        with open(back, "rb") as f:
            decoded = f.read()[44:]
        assert r.stdout.endswith(decoded) and len(decoded) > 0


# ---- eight-way readiness on one GPU (no 8-GPU node has been available: SCALE_r01..r03 are `skipped` records) ----------------
def test_batch_verbs_eight_workers_on_album_tracks(gpu, tmp_path, album_digests):  # noqa: F811
    """Tracks 0..22 of BASELINE.json configs[3] (eight / eight / seven at 44.1 / 48 / 96 kHz, 123,803 frames; 24 tracks would
    put every eighth of the frame space exactly on a track boundary) through
    `sela_mi355x -E / -D --devices 0,0,0,0,0,0,0,0`: EIGHT workers bound to the one GPU, the frame space cut in eight
    contiguous ranges -- the reference's static partition (src/sela/encoder.cpp:58-73) with GPUs for threads -- seven cuts,
    most of them inside a track, every worker reading, coding and writing its own pieces.  Every .sela file and every decoded
    PCM against the unmodified reference's SHA-256s (tests/golden/album_digests.json)."""
    import shutil
    import subprocess
    import tempfile

    from sela_amd.synth import album_tracks, synth_frames_torch
    from test_gpu_round2 import _sha_file
    from test_host_cpp import HOST, _build, _write_wav

    _build()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    work = tempfile.mkdtemp(dir=scratch)
    try:
        tracks = album_tracks()[:23]
        total = sum(frames for _, _, frames in tracks)
        cuts = [total * w // 8 for w in range(1, 8)]
        edges = np.cumsum([0] + [frames for _, _, frames in tracks])
        assert sum(1 for c in cuts if c not in edges) >= 5, "the cuts are supposed to fall inside tracks"
        wavs = []
        for track, rate, frames in tracks:
            pcm = synth_frames_torch(frames, 2, track, device="cuda").cpu().numpy().reshape(-1, 2)
            p = os.path.join(work, f"track{track:02d}.wav")
            _write_wav(p, pcm, rate)
            wavs.append(p)
        enc_dir, dec_dir = os.path.join(work, "enc"), os.path.join(work, "dec")
        os.mkdir(enc_dir), os.mkdir(dec_dir)
        cli = os.path.join(HOST, "sela_mi355x")
        devices = ",".join(["0"] * 8)
        r = subprocess.run([cli, "-E", enc_dir, "--devices", devices] + wavs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for p in wavs:
            os.remove(p)
        selas = []
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            p = os.path.join(enc_dir, f"track{track:02d}.sela")
            assert os.path.getsize(p) == g["sela_bytes"], track
            assert _sha_file(p) == g["sela_sha256"], track
            selas.append(p)
        r = subprocess.run([cli, "-D", dec_dir, "--devices", devices] + selas, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            with open(os.path.join(dec_dir, f"track{track:02d}.wav"), "rb") as f:
                wav = f.read()
            assert len(wav) == 44 + frames * 2048 * 2 * 2
            assert hashlib.sha256(wav[44:]).hexdigest() == g["decoded_sha256"], track
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_bench_with_eight_ranks_sharing_the_gpu():
    """bench.py --gpus 8 as the driver launches it, on a one-GPU box: eight ranks under torch.distributed.run, all on GPU 0,
    talking over gloo (SELA_BENCH_RANKS_SHARE_GPU=1).  Rank r's track (album track 3 r) against the reference's digests on
    every rank; the gathered layout of the eight tracks; the album cut in eight contiguous ranges with the gathered layout
    against the reference's; 10,000 frames decoded in eighths; ONE JSON line, from rank 0, with n_gpus = 8.  Not a
    measurement (the ranks share the device): what it proves is that every code path of an 8-rank job runs and agrees."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SELA_BENCH_RANKS_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-host-legs", "--extra-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [t for t in r.stdout.strip().splitlines() if t.startswith("{")]
    assert len(lines) == 1, "one JSON line, from rank 0"
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["frames_total"] == 8 * 3875
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert line["album"]["layout_matches_reference"] is True and line["album"]["config"]["frames_rank0"] in (68670, 68671)
    assert line["album"]["roundtrip_lossy_frames"] == 39  # (the reference's own lossy frames on the album, summed over the ranks)
    assert line["decode10k"]["config"]["frames_rank0"] == 1250 and line["decode10k"]["bit_exact_vs_cpu_decode"] is True
    assert line["decode10k"]["per_rank_share_8"] is None  # (the one-GPU line's prediction; an 8-rank line IS the thing)


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
