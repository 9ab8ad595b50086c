"""GPU parity tests, by subject: the encoder of the shape the reference's CLI produces (2048 samples of 16-bit PCM per channel and frame) --
k_encode_blocks and k_encode_teams through the C ABI against the CPU oracle and the fixtures the unmodified reference wrote.  Bit-exact: FP64
intermediates (bit patterns; hashes on the timed instantiations), .sela frame bytes, offsets, digests of BASELINE.json's configs, the forms of the
residue filter, the differential corpus.  (src/frame/frame_encoder.cpp:11-102, src/lpc/residue_generator.cpp:12-134, src/rice/rice_encoder.cpp:12-81.)"""
import ctypes as C
import hashlib
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import _bits, _build_frame, _decode, _encode, _fold, _hard_blocks, _kat_block_frames, _polyphonic_frames, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


def test_analysis_stages_bit_exact(gpu, kats):
    """mean / autocorrelation / reflection coefficients / order / q / a against the oracle's trace.

    The trace comes from k_encode_blocks<1>, a different instantiation (register allocation) from the timed
    k_encode_blocks<0>; the product instantiation's correctness rests on the frame bytes and digests compared
    everywhere else in this file -- every FP64 intermediate feeds them through q[] and the order."""
    o = oracle()
    names, mono = _kat_block_frames(kats)
    pcm_sets = [mono, synth_frames(24, 2, 3), synth_frames(5, 3, 4)]
    for pcm in pcm_sets:
        frames, offsets, enc, out = _encode(gpu, pcm, with_trace=True)
        ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=4)  # (the trace build runs the plain FIR loop)
        assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
        traces = enc.traces(pcm.shape[0])
        ch = pcm.shape[2]
        n_sig = 3 if ch == 2 else ch
        for f in range(pcm.shape[0]):
            for sig in range(n_sig):
                s = (pcm[f, :, 0].astype(np.int32) - pcm[f, :, 1]) if (ch == 2 and sig == 2) else pcm[f, :, sig].astype(np.int32)
                order, q, r, a, tr, _ = o.lpc_analyze(s, with_trace=True)
                g = traces[f * n_sig + sig]
                ctx = (f, sig)
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


def test_all_int16_sample_values(gpu):
    """Every int16 value goes through the x/32767 division and the FP64 sums at least once."""
    o = oracle()
    vals = np.arange(-32768, 32768, dtype=np.int32)
    rng = np.random.default_rng(3)
    rng.shuffle(vals)
    pcm = vals.astype(np.int16).reshape(32, 2048, 1)
    frames, offsets, enc, out = _encode(gpu, pcm, with_trace=True)
    traces = enc.traces(32)
    for f in range(32):
        _, _, _, _, tr, _ = o.lpc_analyze(pcm[f, :, 0].astype(np.int32), with_trace=True)
        assert np.array_equal(_bits(traces[f].mean), _bits(tr.mean))
        assert np.array_equal(_bits(list(traces[f].ac)), _bits(list(tr.ac)))
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=4)
    assert np.array_equal(frames, ref_frames) and np.array_equal(offsets, ref_offsets)


def test_frame_kats_encode_and_decode(gpu, kats):
    """Golden on-disk frames produced by the reference (tests/golden/make_golden.py)."""
    for name in kats["frame_names"]:
        pcm = kats[f"frame/{name}/pcm"][None]
        golden = kats[f"frame/{name}/bytes"]
        frames, offsets, _, _ = _encode(gpu, pcm)
        assert offsets.tolist() == [0, len(golden)], name
        assert np.array_equal(frames, golden), name
        back = _decode(gpu, golden, np.array([0, len(golden)], np.uint64), pcm.shape[2])
        assert np.array_equal(back[0], kats[f"frame/{name}/decoded"]), name


def test_block_kats_as_frames(gpu, kats):
    """Edge blocks (silence, constant, full-scale noise, square, impulse, ramp, ...) as mono frames."""
    o = oracle()
    names, mono = _kat_block_frames(kats)
    frames, offsets, _, _ = _encode(gpu, mono)
    ref_frames, ref_offsets, _ = o.encode_frames(mono, threads=2)
    assert np.array_equal(offsets, ref_offsets)
    assert np.array_equal(frames, ref_frames)
    for i, name in enumerate(names):  # per-block check against the reference's own numbers
        b = frames[int(offsets[i]): int(offsets[i + 1])]
        assert b[10] == int(kats[f"blk/{name}/order"]), name
        cw = int(b[8]) | int(b[9]) << 8
        assert np.array_equal(b[11: 11 + 4 * cw].view(np.uint32), kats[f"blk/{name}/coef_words"]), name
    back = _decode(gpu, frames, offsets, 1)
    assert np.array_equal(back, mono)


@pytest.mark.parametrize("channels,track,n_frames", [(1, 11, 70), (2, 12, 150), (3, 13, 40), (6, 14, 20), (9, 15, 5), (17, 16, 2)])
def test_random_batches_match_oracle(gpu, channels, track, n_frames):
    o = oracle()
    pcm = synth_frames(n_frames, channels, track)
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=8)
    assert np.array_equal(offsets, ref_offsets)
    assert np.array_equal(frames, ref_frames)
    back = _decode(gpu, frames, offsets, channels)
    ref_back, _ = o.decode_frames(ref_frames, ref_offsets, channels, threads=8)
    assert np.array_equal(back, ref_back) and np.array_equal(back, pcm)


def test_repeated_launches_are_deterministic(gpu):
    """The encoder's scalar-operand rings are handed out per launch by ticket and live in L2 between the
    stores and the scalar loads of one block (sela_encode.hip): hammer the same workspace with launches of
    different sizes, back to back without synchronising, and require every result to stay bit-identical."""
    from sela_amd import codec

    pcm = gpu.from_numpy(synth_frames(700, 2, 21)).cuda()
    enc = codec.Encoder(700, 2)
    dec = codec.Decoder(700, 2)
    want = {}
    for n in (700, 1, 64, 333):
        out = enc.encode(pcm[:n])
        gpu.cuda.synchronize()
        want[n] = (out.frames[: out.total_bytes()].clone(), out.offsets.clone())
    for it in range(60):
        for n in (333, 700, 1, 64):
            out = enc.encode(pcm[:n])
            f, o = want[n]
            assert bool((out.offsets == o).all().item()), (it, n)
            assert bool((out.frames[: f.numel()] == f).all().item()), (it, n)
        back = dec.decode(out.frames, out.offsets, 64).clone()
        first_back = back if it == 0 else first_back
        assert bool((back == first_back).all().item()), it
    gpu.cuda.synchronize()
    out.check()
    dec.check()


def test_extreme_stereo(gpu):
    """Full-scale anti-correlated channels: the difference signal uses all 17 bits."""
    o = oracle()
    rng = np.random.default_rng(21)
    l = rng.integers(-32768, 32768, (6, 2048)).astype(np.int16)
    pcm = np.stack([l, (-l.astype(np.int32)).clip(-32768, 32767).astype(np.int16)], axis=2)
    pcm[3] = np.stack([np.full(2048, 32767, np.int16), np.full(2048, -32768, np.int16)], axis=1)
    pcm[4, :, 1] = pcm[4, :, 0]
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=4)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
    # NOT compared with pcm: on this input the reference itself is off by one LSB in 9 samples -- its
    # encoder rounds the prediction half-up and its decoder half-down; parity means
    # reproducing exactly that.
    ref_back, _ = o.decode_frames(ref_frames, ref_offsets, 2, threads=4)
    assert np.array_equal(_decode(gpu, frames, offsets, 2), ref_back)


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "config2_1000_frames"])
def test_baseline_configs_by_digest(gpu, digests, label):
    """BASELINE.json configs 0-2 at full size: SHA-256 of the frame stream / offsets / decoded PCM
    against digests computed with the unmodified reference."""
    d = digests[label]
    pcm = synth_frames(d["n_frames"], d["channels"], d["track"])
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == d["pcm_sha256"]
    frames, offsets, _, out = _encode(gpu, pcm)
    assert len(frames) == d["frames_blob_bytes"]
    assert hashlib.sha256(frames.tobytes()).hexdigest() == d["frames_blob_sha256"]
    assert hashlib.sha256(offsets.astype("<u8").tobytes()).hexdigest() == d["offsets_sha256"]
    back = _decode(gpu, frames, offsets, d["channels"])
    assert hashlib.sha256(back.tobytes()).hexdigest() == d["decoded_sha256"]
    assert np.array_equal(back, pcm)  # encode -> decode round trip is lossless


def test_reference_library_agrees_when_present(gpu):
    """If the real reference travelled with the repo (oracle/_ref), compare against it directly."""
    ref = reference()
    if ref is None:
        pytest.skip("oracle/_ref/libsela_ref.so not present")
    pcm = synth_frames(64, 2, 40)
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = ref.encode_frames(pcm, threads=8)
    assert np.array_equal(frames, ref_frames) and np.array_equal(offsets, ref_offsets)
    ref_back, _ = ref.decode_frames(ref_frames, ref_offsets, 2, threads=8)
    assert np.array_equal(_decode(gpu, frames, offsets, 2), ref_back)


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


@pytest.mark.parametrize("channels", [1, 2, 3])
def test_mean_workers_give_the_same_means(gpu, channels):
    """sela_hip_debug_mean_workers(self_blocks): all but the first `self_blocks` blocks of a launch take their
    2048-term sequential mean from the "mean worker" workgroups (lane = block) instead of walking the chain in
    their own wave.  Small batches never do by default, so the hook forces it: the means (bit patterns), every
    later intermediate and the frame bytes must not change.  Large batches (the configs[1] digest test, the
    10k-frame test, the album) take the worker path without the hook."""
    from sela_amd import capi
    from gpu_common import _bits

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
    from gpu_common import _decode

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


def test_wide_differential_corpus_against_the_reference(gpu):  # noqa: F811
    """33,400 stereo frames = 100,200 analysed blocks (tests/corpus.py: AR(2..32) noise at many levels, |k| hovering at 0.05,
    clipped and faded tones, DC steps, silence <-> full scale inside a block, 17-bit differences, loud and smooth polyphony)
    through all three encode kernels and the decoder against the UNMODIFIED reference (oracle/_ref/libsela_ref.so on all host
    threads; the oracle where that library is absent): frame bytes, offsets, decoded PCM.  And which form of the residue
    filter every block took, unforced, read back from the product kernels' per-block records: equal to the documented rule
    evaluated on the oracle's predictor for a sample of blocks; one pass and two passes both occur (the plain loop does not:
    no 16-bit input was found that gets a predictor past 2^39 through the reference's quantiser -- it is reached by the
    forced-form tests of round 4 and by the trace builds)."""
    import ctypes as C
    import os
    import time

    import corpus
    from oracle_lib import reference
    from sela_amd import capi, codec

    t0 = time.time()
    pcm = corpus.build(seed=int(os.environ.get("SELA_CORPUS_SEED", "20260927")))  # (a soak: other seeds, other material)
    n = pcm.shape[0]
    ref = reference() or oracle()
    threads = os.cpu_count() or 8
    want, want_offs, _ = ref.encode_frames(pcm, threads=threads)
    want_dec, _ = ref.decode_frames(want, want_offs, 2, threads=threads)
    t1 = time.time()
    lib = capi.lib()
    enc = codec.Encoder(n, 2)
    enc.frames = gpu.empty(int(lib.sela_hip_encode_bound_bytes(n, 2)), dtype=gpu.uint8, device="cuda")
    enc.capacity = enc.frames.numel()
    d_pcm = gpu.from_numpy(pcm).cuda()
    o = oracle()
    seen = np.zeros(3, np.int64)
    try:
        for teams in (-1, 0, 16, 8):
            lib.sela_hip_debug_encode_teams(teams)
            out = enc.encode(d_pcm)
            gpu.cuda.synchronize()
            frames, offs = out.to_host()
            assert np.array_equal(offs, want_offs), teams
            assert np.array_equal(frames, want), teams
            counts = (C.c_uint32 * 3)()
            forms = np.zeros(n * 3, np.uint8)
            assert lib.sela_hip_debug_block_forms(enc.workspace.data_ptr(), n, 2, counts, forms.ctypes.data) == 0
            seen += np.array(list(counts))
            assert counts[0] > 0 and counts[1] > 0 and sum(counts) == 3 * n, (teams, list(counts))
            if teams == -1:
                picked = list(counts)
                rng = np.random.default_rng(5)
                sample = set(rng.integers(0, n, 400).tolist()) | set((np.nonzero(forms.reshape(n, 3).any(axis=1))[0][:200]).tolist())
                for f in sorted(sample):
                    l, r = pcm[f, :, 0].astype(np.int32), pcm[f, :, 1].astype(np.int32)
                    for sig, s in enumerate((l, r, l - r)):
                        order, q = o.lpc_analyze(s)[:2]
                        a = o.lpc_coeffs(order, q)
                        assert forms[3 * f + sig] == corpus.expected_form(a, order, s), (f, sig)
    finally:
        lib.sela_hip_debug_encode_teams(-1)
    dec = codec.Decoder(n, 2)
    back = dec.decode(out.frames, out.offsets, n)
    gpu.cuda.synchronize()
    dec.check()
    assert np.array_equal(back.cpu().numpy(), want_dec)
    lossy = int((want_dec != pcm).reshape(n, -1).any(axis=1).sum())
    print(f"\ncorpus: {n} stereo frames, {3 * n} blocks; reference {'libsela_ref.so' if ref.is_reference else 'oracle'} on {threads} threads "
          f"{t1 - t0:.1f} s (with generation); forms by the library's own kernel choice (one pass, two passes, plain) = {picked}; "
          f"frames the reference's own decoder does not return exactly: {lossy}; whole test {time.time() - t0:.1f} s")


@pytest.mark.parametrize("teams", [16, 8, 0], ids=["k_encode_teams<.,16>", "k_encode_teams<.,8>", "k_encode_blocks<.,false>"])
def test_fp64_intermediates_of_the_product_kernels_by_hash(gpu, kats, teams):  # noqa: F811
    """The normalised autocorrelation ac[0..100] and the reflection coefficients k[0..99] of the kernels that are TIMED --
    k_encode_teams<0,16>, <0,8>, k_encode_blocks<0,false> plus the few instructions that fold them (their kMode 3
    instantiations; round 4 checked these doubles on the trace builds only) -- as two 64-bit hashes per block against the
    oracle's trace folded the same way: the KAT blocks, the corner blocks (NaN paths), stereo and three-channel frames."""
    from sela_amd import capi, codec
    from gpu_common import _kat_block_frames
    from gpu_common import _hard_blocks

    lib = capi.lib()
    o = oracle()
    _, mono = _kat_block_frames(kats)
    lib.sela_hip_debug_encode_teams(teams)
    lib.sela_hip_debug_encode_hashes(1)
    try:
        for pcm in (mono, _hard_blocks(), synth_frames(27, 2, 3), synth_frames(5, 3, 4)):
            nf, _, ch = pcm.shape
            n_sig = 3 if ch == 2 else ch
            enc = codec.Encoder(nf, ch, with_trace=True)  # (the trace buffer is more than the 16 bytes per block used here)
            enc.trace.zero_()
            out = enc.encode(gpu.from_numpy(np.ascontiguousarray(pcm)).cuda())
            gpu.cuda.synchronize()
            frames, offs = out.to_host()
            want, want_offs, _ = o.encode_frames(pcm, threads=4)
            assert np.array_equal(frames, want) and np.array_equal(offs, want_offs)
            got = enc.trace[: nf * n_sig * 16].cpu().numpy().view(np.uint64).reshape(nf * n_sig, 2)
            for f in range(nf):
                for sig in range(n_sig):
                    s = (pcm[f, :, 0].astype(np.int32) - pcm[f, :, 1]) if (ch == 2 and sig == 2) else pcm[f, :, sig].astype(np.int32)
                    tr = o.lpc_analyze(s, with_trace=True)[4]
                    assert int(got[f * n_sig + sig, 0]) == _fold(list(tr.ac)), (teams, f, sig, "ac")
                    assert int(got[f * n_sig + sig, 1]) == _fold(list(tr.k)), (teams, f, sig, "k")
    finally:
        lib.sela_hip_debug_encode_hashes(0)
        lib.sela_hip_debug_encode_teams(-1)


def test_wave_priorities_follow_the_neighbours(gpu):  # noqa: F811
    """An encode launch takes the falling wave-priority schedule exactly when no OTHER stream has library work pending
    (sela_capi.hip, Flights): alone on its stream -- every launch; queued behind a 20,000-frame encode that another stream
    still runs -- none; and the bytes do not depend on it (forced on, forced off, by the library)."""
    from sela_amd import capi, codec

    lib = capi.lib()
    pcm = gpu.from_numpy(synth_frames(600, 2, 6)).cuda()
    big = gpu.from_numpy(np.tile(synth_frames(500, 2, 7), (40, 1, 1))).cuda()
    enc_a, enc_b = codec.Encoder(600, 2), codec.Encoder(20000, 2)
    want = None
    for forced in (0x00010203, 0, None):
        if forced is None:
            lib.sela_hip_debug_priorities_adaptive()
        else:
            lib.sela_hip_debug_priorities(forced)
        out = enc_a.encode(pcm)
        gpu.cuda.synchronize()
        frames, offs = out.to_host()
        if want is None:
            want = (frames.copy(), offs.copy())
            ref, ref_offs, _ = oracle().encode_frames(pcm.cpu().numpy(), threads=8)
            assert np.array_equal(frames, ref) and np.array_equal(offs, ref_offs)
        assert np.array_equal(frames, want[0]) and np.array_equal(offs, want[1]), forced
    gpu.cuda.synchronize()
    before = lib.sela_hip_debug_launches_alone()
    for _ in range(5):  # one stream, nothing else in flight: every launch is alone
        enc_a.encode(pcm)
    gpu.cuda.synchronize()
    assert lib.sela_hip_debug_launches_alone() - before == 5
    s1, s2 = gpu.cuda.Stream(), gpu.cuda.Stream()
    gpu.cuda.synchronize()
    before = lib.sela_hip_debug_launches_alone()
    with gpu.cuda.stream(s1):
        enc_b.encode(big)  # (several milliseconds of work: alone when it was queued)
    with gpu.cuda.stream(s2):
        for _ in range(3):
            enc_a.encode(pcm)  # queued while the other stream's launch is pending: a neighbour
    gpu.cuda.synchronize()
    assert lib.sela_hip_debug_launches_alone() - before == 1
    out = enc_a.encode(pcm)
    gpu.cuda.synchronize()
    frames, offs = out.to_host()
    assert np.array_equal(frames, want[0]) and np.array_equal(offs, want[1])


@pytest.mark.parametrize("share", [300, 520, 800])
def test_a_launch_cut_in_two_gives_the_same_bytes(gpu, share):  # noqa: F811
    """The form of sela_hip_encode_device that cuts a launch in two halves on two streams (sela_capi.hip, Splitter; an experiment
    behind a debug hook: measured slower, never taken by the library itself): frame bytes, offsets, status words and the
    per-block records are those of the whole launch and of the oracle; a launch that is not teams of 16 is never cut."""
    from sela_amd import capi, codec

    lib = capi.lib()
    o = oracle()
    pcm = synth_frames(3875, 2, 11)
    pcm[100:108, :, 1] = pcm[100:108, :, 0] // 2  # (some frames whose difference wins)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=8)
    d_pcm = gpu.from_numpy(pcm).cuda()
    enc = codec.Encoder(3875, 2)
    try:
        lib.sela_hip_debug_encode_split(0)
        whole = enc.encode(d_pcm)
        gpu.cuda.synchronize()
        f0, o0 = whole.to_host()
        forms0 = np.zeros(3875 * 3, np.uint8)
        counts0 = np.zeros(3, np.uint32)
        capi.check(lib.sela_hip_debug_block_forms(C.c_void_p(enc.workspace.data_ptr()), 3875, 2, C.c_void_p(counts0.ctypes.data), C.c_void_p(forms0.ctypes.data)))
        before = lib.sela_hip_debug_launches_split()
        lib.sela_hip_debug_encode_split(share)
        gpu.cuda.synchronize()  # (alone: nothing else pending)
        cut = enc.encode(d_pcm)
        gpu.cuda.synchronize()
        assert lib.sela_hip_debug_launches_split() == before + 1
        f1, o1 = cut.to_host()
        forms1 = np.zeros(3875 * 3, np.uint8)
        counts1 = np.zeros(3, np.uint32)
        capi.check(lib.sela_hip_debug_block_forms(C.c_void_p(enc.workspace.data_ptr()), 3875, 2, C.c_void_p(counts1.ctypes.data), C.c_void_p(forms1.ctypes.data)))
        assert np.array_equal(o1, o0) and np.array_equal(f1, f0)
        assert np.array_equal(o1, ref_offsets) and np.array_equal(f1, ref_frames)
        assert np.array_equal(forms1, forms0) and np.array_equal(counts1, counts0)
        # decode what the cut launch wrote
        dec = codec.Decoder(3875, 2)
        back = dec.decode(cut.frames, cut.offsets, 3875)
        gpu.cuda.synchronize()
        dec.check()
        ref_back, _ = o.decode_frames(ref_frames, ref_offsets, 2, threads=8)
        assert np.array_equal(back.cpu().numpy(), ref_back)
        # a small launch (k_encode_blocks) is never cut
        small = codec.Encoder(1000, 2)
        n_before = lib.sela_hip_debug_launches_split()
        out = small.encode(d_pcm[:1000])
        gpu.cuda.synchronize()
        assert lib.sela_hip_debug_launches_split() == n_before
        fs, os_ = out.to_host()
        assert np.array_equal(os_, ref_offsets[:1001]) and np.array_equal(fs, ref_frames[: int(ref_offsets[1000])])
    finally:
        lib.sela_hip_debug_encode_split(0)
