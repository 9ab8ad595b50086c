"""GPU parity tests, by subject: the decoder of 2048-sample frames (k_decode_frames) through the C ABI against the CPU oracle -- corrupt and
hostile streams, the lane-parallel parser at the edges of its plan, both forms of the recurrence step, decode-only 10,000 frames.
(src/frame/frame_decoder.cpp:11-72, src/rice/rice_decoder.cpp:11-61, src/lpc/sample_generator.cpp:11-39.)"""
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
import struct
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import _build_frame, _decode, _decode_frames_vs_oracle, _encode, _rice_words, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


def test_decoder_rejects_corrupt_frames(gpu):
    from sela_amd import capi, codec

    pcm = synth_frames(4, 2, 31)
    frames, offsets = codec.encode_host(pcm)
    bad = frames.copy()
    bad[int(offsets[2])] ^= 0x01  # sync word of frame 2
    with pytest.raises(capi.SelaHipError) as e:
        codec.decode_host(bad, offsets, 2)
    assert e.value.code == -5


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


@pytest.mark.parametrize("form", [0, 1])
def test_decoder_recurrence_forms_give_the_same_samples(gpu, kats, form):
    """The decoder walks a subframe's recurrence in one of two forms, chosen by how alone the workgroup will be on its SIMDs
    (sela_decode.hip, synth_steps kVecShift): small launches -- every test batch -- would only ever see one of them.  Forced
    here, both forms go through the decode tests that reach every branch of the synthesis: the reference's golden frames,
    the edge blocks (order 93 and 97: the ring of 128), full-scale difference channels, residues outside 16 bits (the exact
    32-bit form and the switch to it in the middle of a subframe), unary runs of thousands of bits, random valid streams,
    frames of more than eight channels; and one batch of more workgroups than the device holds (both forms in one launch
    is what the size rule gives a 3875-frame batch: test_baseline_configs_by_digest)."""
    import test_gpu_encode_parity as enc
    from sela_amd import capi, codec

    torch = gpu
    lib = capi.lib()
    lib.sela_hip_debug_decode_recurrence(form)
    try:
        enc.test_frame_kats_encode_and_decode(gpu, kats)
        enc.test_block_kats_as_frames(gpu, kats)
        enc.test_extreme_stereo(gpu)
        test_long_unary_runs_round_trip(gpu)
        test_decoder_on_streams_no_encoder_would_write(gpu, kats)
        enc.test_random_batches_match_oracle(gpu, 2, 12, 150)
        enc.test_random_batches_match_oracle(gpu, 6, 14, 20)
        test_segment_parallel_parser_on_hard_streams(gpu, kats)
        test_parser_on_random_valid_streams(gpu, kats)
        enc.test_wide_frames_encode_and_decode(gpu, 32, 3)
        enc.test_wide_frames_with_difference_subframes_and_long_streams(gpu)
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
