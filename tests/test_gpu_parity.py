"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bit-exact everywhere: .sela frame bytes, frame offsets, decoded PCM, and -- stage by stage -- the
FP64 intermediates of the analysis (compared as bit patterns, tolerance zero).
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from sela_amd import capi

    capi.lib()  # raises if the HIP library is missing: no fallback
    return torch


def _encode(gpu, pcm, with_trace=False):
    from sela_amd import codec

    enc = codec.Encoder(pcm.shape[0], pcm.shape[2], with_trace=with_trace)
    out = enc.encode(gpu.from_numpy(np.ascontiguousarray(pcm)).cuda())
    gpu.cuda.synchronize()
    frames, offsets = out.to_host()
    return (frames, offsets, enc, out)


def _decode(gpu, frames, offsets, channels):
    from sela_amd import codec

    n = len(offsets) - 1
    dec = codec.Decoder(n, channels)
    f = gpu.from_numpy(np.ascontiguousarray(frames)).cuda()
    o = gpu.from_numpy(np.ascontiguousarray(offsets).view(np.int64)).cuda()
    pcm = dec.decode(f, o, n)
    gpu.cuda.synchronize()
    dec.check()
    return pcm.cpu().numpy()


def _bits(x):
    """Bit patterns of doubles; NaNs are canonicalised (sign/payload of a NaN carries no meaning in
    the codec: every consumer is a comparison or isnan())."""
    v = np.atleast_1d(np.asarray(x, dtype=np.float64)).copy()
    v[np.isnan(v)] = np.nan
    return v.view(np.uint64)


def _kat_block_frames(kats):
    """The single-block KAT signals as mono frames (those that fit int16)."""
    names, frames = [], []
    for name in kats["blk_names"]:
        s = kats[f"blk/{name}/samples"]
        if s.min() >= -32768 and s.max() <= 32767:
            names.append(str(name))
            frames.append(s.astype(np.int16)[:, None])
    return names, np.stack(frames)


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


def test_decoder_rejects_corrupt_frames(gpu):
    from sela_amd import capi, codec

    pcm = synth_frames(4, 2, 31)
    frames, offsets = codec.encode_host(pcm)
    bad = frames.copy()
    bad[int(offsets[2])] ^= 0x01  # sync word of frame 2
    with pytest.raises(capi.SelaHipError) as e:
        codec.decode_host(bad, offsets, 2)
    assert e.value.code == -5


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


def test_decoder_survives_corrupt_streams(gpu):
    """Bit flips anywhere in the frame stream must end in an error code or garbage PCM, never a hang,
    a crash or an out-of-bounds access (the reference has no bounds checks here)."""
    from sela_amd import capi, codec

    pcm = synth_frames(24, 2, 33)
    frames, offsets = codec.encode_host(pcm)
    rng = np.random.default_rng(9)
    for trial in range(12):
        bad = frames.copy()
        for _ in range(1 + trial * 3):
            pos = int(rng.integers(0, len(bad)))
            bad[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        try:
            out = codec.decode_host(bad, offsets, 2)
            assert out.shape == pcm.shape
        except capi.SelaHipError as e:
            assert e.code == -5
    # all-ones payload: maximal unary runs everywhere
    bad = frames.copy()
    bad[int(offsets[3]) + 40: int(offsets[4])] = 0xFF
    try:
        codec.decode_host(bad, offsets, 2)
    except capi.SelaHipError as e:
        assert e.code == -5
    # after all that the decoder still works
    assert np.array_equal(codec.decode_host(frames, offsets, 2), pcm)


def test_long_unary_runs_round_trip(gpu):
    """Sparse full-scale impulses: residues of +-32767 next to zeros give codewords of thousands of
    bits (encoder put_codeword loop, parser slow path)."""
    o = oracle()
    rng = np.random.default_rng(4)
    pcm = np.zeros((8, 2048, 2), np.int16)
    for f in range(8):
        idx = rng.integers(0, 2048, 3 + f)
        pcm[f, idx, 0] = rng.choice([-32768, 32767], len(idx))
        pcm[f, idx[:2], 1] = 32767
    frames, offsets, _, _ = _encode(gpu, pcm)
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=4)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
    # NOT compared with pcm: on this input the reference itself is off by one LSB in 9 samples -- its
    # encoder rounds the prediction half-up and its decoder half-down; parity means
    # reproducing exactly that.
    ref_back, _ = o.decode_frames(ref_frames, ref_offsets, 2, threads=4)
    assert np.array_equal(_decode(gpu, frames, offsets, 2), ref_back)


def test_decode_only_10k_frames(gpu):
    """BASELINE.json configs[4] at full size (10k pre-encoded stereo frames): the whole encoded batch bit-exact
    against the oracle's encode, the decode of the whole batch bit-exact against the oracle's decode.  (Not
    "== pcm": the reference is off by one LSB per sample in frames 635 and 946 of this track -- half-up/half-down
    rounding, SURVEY.md App. E.)"""
    o = oracle()
    n = 10000
    pcm = synth_frames(n, 2, 2)
    frames, offsets, _, out = _encode(gpu, pcm)
    threads = os.cpu_count() or 1
    ref_frames, ref_offsets, _ = o.encode_frames(pcm, threads=threads)
    assert np.array_equal(offsets, ref_offsets) and np.array_equal(frames, ref_frames)
    back = _decode(gpu, frames, offsets, 2)
    ref_back, _ = o.decode_frames(frames, offsets, 2, threads=threads)
    assert np.array_equal(back, ref_back)
    assert (back != pcm).reshape(n, -1).any(axis=1).sum() <= 4


def _rice_words(values, k):
    """Rice-code `values` with a GIVEN parameter k (src/rice/rice_encoder.cpp:35-71 without the parameter
    search): zig-zag, u >> k ones, a zero, k remainder bits MSB first; stream bit t = bit t % 32 of word t / 32."""
    bits = []
    for v in values:
        v = int(v)
        u = -2 * v - 1 if v < 0 else 2 * v
        bits += [1] * (u >> k) + [0] + [(u >> (k - 1 - i)) & 1 for i in range(k)]
    bits += [0] * (-len(bits) % 32)
    b = np.array(bits, np.uint64).reshape(-1, 32)
    return (b << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


def _build_frame(subframes, res_k=None):
    """Hand-assemble on-disk frame bytes from (channel, type, parent, q[], residues[]) tuples, Rice-coding
    with the oracle -- or, for the residues, with the parameter `res_k` no encoder would pick
    (layout of src/file/sela_file.cpp:115-135)."""
    import struct

    o = oracle()
    out = struct.pack("<I", 0xAA55FF00)
    for channel, typ, parent, q, res in subframes:
        ck, cw = o.rice_encode(np.asarray(q, np.int32))
        if res_k is None:
            rk, rw = o.rice_encode(np.asarray(res, np.int32))
        else:
            rk, rw = res_k, _rice_words(res, res_k)
        out += struct.pack("<BBBBHB", channel, typ, parent, ck, len(cw), len(q)) + cw.astype("<u4").tobytes()
        out += struct.pack("<BHH", rk, len(rw), len(res)) + rw.astype("<u4").tobytes()
    return out


def test_decoder_on_streams_no_encoder_would_write(gpu, kats):
    """Hand-built frames: residues far outside 16 bits (forces the synthesis filter off its folded fast
    path onto the exact 32-bit one, and the parser onto very long unary runs), order 0 and order 100,
    a difference channel -- all against the oracle's decoder."""
    o = oracle()
    rng = np.random.default_rng(12)
    q_sine = kats["blk/sine_deg/q"]                      # a real order-17 predictor
    q_noise = kats["blk/white_fullscale/q"]              # order 93
    big = rng.integers(-2000, 2000, 2048).astype(np.int32)
    big[rng.choice(2048, 48, replace=False)] = rng.integers(1 << 27, 1 << 28, 48) * rng.choice([-1, 1], 48)
    mixed = rng.integers(-300, 300, 2048).astype(np.int32)
    mixed[1000] = 1 << 28                                 # one huge sample in the middle of a chunk
    frames = [
        _build_frame([(0, 0, 0, [0], big)]),                                   # order 1: samples == residues
        _build_frame([(0, 0, 0, q_sine, mixed)]),
        _build_frame([(0, 0, 0, q_noise, mixed[::-1].copy())]),
        _build_frame([(0, 0, 0, q_sine, rng.integers(-50, 50, 2048))]),         # ordinary
        _build_frame([(0, 0, 0, np.zeros(100, np.int32), rng.integers(-9, 9, 2048))]),  # order 100, all-zero q
        _build_frame([(0, 0, 0, [], rng.integers(-9, 9, 2048))]),                # order 0
        _build_frame([(0, 0, 0, q_noise[:55], mixed)]),                          # order 55: ring of 64, groups of 4, exact path
        _build_frame([(0, 0, 0, q_noise[:55], rng.integers(-50, 50, 2048))]),     # ... and its folded path
        _build_frame([(0, 0, 0, q_noise[:40], mixed[::-1].copy())]),             # order 40: groups of 16, exact path
        _build_frame([(0, 0, 0, q_sine, rng.integers(-50, 50, 2048))], res_k=27), # k too wide for packed words: slow parser path throughout
        _build_frame([(0, 0, 0, q_sine, rng.integers(-5000, 5000, 2048))], res_k=3),   # fast and slow groups mixed within blocks
        _build_frame([(0, 0, 0, q_noise, rng.integers(-300, 300, 2048))], res_k=0),    # unary only
    ]
    stream = np.frombuffer(b"".join(frames), np.uint8).copy()
    offsets = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    got = _decode(gpu, stream, offsets, 1)
    for i, f in enumerate(frames):
        want, used = o.frame_decode(f, 1)
        assert used == len(f)
        assert np.array_equal(got[i], want), i
    # stereo with a difference channel whose parent has huge samples
    st = _build_frame([(0, 0, 0, q_sine, mixed), (1, 1, 0, q_noise, rng.integers(-40, 40, 2048))])
    got = _decode(gpu, np.frombuffer(st, np.uint8).copy(), np.array([0, len(st)], np.uint64), 2)
    want, _ = o.frame_decode(st, 2)
    assert np.array_equal(got[0], want)
