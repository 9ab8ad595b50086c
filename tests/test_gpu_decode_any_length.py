"""GPU parity tests, by subject: the decoder for subframes of ANY length and 32-bit samples (k_decode_subframes32 in
sela_decode32.hip -- the fast decoder's lane-parallel Rice parse, cut into segments, and its tuned synthesis with the length a
run-time value -- with k_generic_decode, the serial walk, as the judge of streams it will not touch).  Everything is compared,
bit for bit, with the oracle (oracle/sela_oracle.c, pinned against the unmodified reference): src/frame/frame_decoder.cpp:11-72,
src/rice/rice_decoder.cpp:11-61, src/lpc/sample_generator.cpp:11-39."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import synth_frames, synth_pcm
from test_gpu_parity import gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _one(nbytes):
    return np.array([0, nbytes], np.uint64)


def _decode_modes(blob, offs, ch, modes=(1, 2, 0)):
    """sela_hip_decode_i32 as the product runs it (1), with every subframe parsed by segments (2) and on the serial kernel alone
    (0) -> ({mode: frames}, chunks the fast kernel took whole, subframes it parsed by segments) -- the counts over modes 1 and 2."""
    from sela_amd import capi, codec

    lib = capi.lib()
    out = {}
    chunks0, segs0 = lib.sela_hip_debug_standard_chunks(), lib.sela_hip_debug_segment_subframes()
    try:
        for m in modes:
            lib.sela_hip_debug_standard_first(m)
            out[m] = codec.decode_i32(blob, offs, ch)
    finally:
        lib.sela_hip_debug_standard_first(-1)
    return out, lib.sela_hip_debug_standard_chunks() - chunks0, lib.sela_hip_debug_segment_subframes() - segs0


def _signal(rng, kind, n, amp_bits):
    t = np.arange(n)
    amp = (1 << amp_bits) - 1
    if kind == "silence":
        return np.zeros(n, np.int32)
    if kind == "dc":
        return np.full(n, amp // 3, np.int32)
    if kind == "noise":  # incompressible: long Rice streams (beyond one segment's words from a few thousand samples)
        return rng.integers(-amp, amp + 1, n).astype(np.int32)
    if kind == "tone":
        return np.round(amp * 0.8 * np.sin(t * 0.05 + 1.0) + rng.normal(0, amp / 300 + 1, n)).astype(np.int32)
    if kind == "sparse":  # mostly zeros with rare full-scale clicks: k = 0 .. 2 with unary runs of thousands of bits
        x = np.zeros(n, np.int32)
        idx = rng.integers(0, n, max(1, n // 200))
        x[idx] = rng.integers(-amp, amp + 1, len(idx))
        return x
    raise AssertionError(kind)


LENGTHS = [1, 2, 63, 64, 65, 127, 129, 300, 1000, 2047, 2048, 2049, 4096, 5000, 20000, 65535]


@pytest.mark.parametrize("n", LENGTHS)
def test_encoder_frames_of_any_length_through_the_three_decoders(gpu, n):  # noqa: F811
    """Frames an encoder wrote, 1 .. 65535 samples per channel, mono / stereo / three channels, silence (one bit per codeword: a
    segment is cut by its codeword count), tones, clicks (unary runs of thousands of bits), full-scale noise (streams of several
    segments' words), 16- and 21-bit: the fast kernel as the product runs it, the same with every subframe by segments, and the
    serial kernel all give the oracle's 32-bit samples."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(1000 + n)
    segs_seen = coded = 0
    for kinds, bits in ((("silence",), 16), (("tone", "noise"), 16), (("sparse", "tone", "dc"), 16), (("noise",), 21), (("tone", "tone"), 21)):
        x = np.stack([_signal(rng, k, n, bits) for k in kinds])
        try:
            frames, offs = codec.encode_i32(x[None])
        except Exception:  # (a block not longer than its own predictor order: refused, like the reference's out-of-bounds read)
            assert n <= 100
            continue
        want = o.frame_encode_i32(x)
        assert frames.tobytes() == want, (n, kinds)
        coded += 1
        ref, used = o.frame_decode_i32(want, len(kinds))
        assert used == len(want)
        got, chunks, segs = _decode_modes(frames, offs, len(kinds))
        assert chunks == 2, (n, kinds, chunks)  # (modes 1 and 2: the fast kernel took the call)
        segs_seen += segs
        for m, dec in got.items():
            for c in range(len(kinds)):
                assert np.array_equal(dec[0][c], ref[c]), (n, kinds, bits, "mode", m, "channel", c)
    assert segs_seen > 0 or coded == 0
    assert coded >= 3 or n < 128


def test_a_batch_of_frames_of_different_lengths_in_one_call(gpu):  # noqa: F811
    """One call, 60 frames, every frame its own length (1 .. 9000) and signal, stereo: decoded at the stream's largest length as
    stride, counts per channel, all three decoders."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(77)
    blobs, refs = [], []
    for i in range(60):
        n = int(rng.integers(101, 9000)) if i % 5 else int(rng.choice([2048, 101, 4096, 8999]))
        kind = ["tone", "noise", "sparse", "silence"][i % 4]
        x = np.stack([_signal(rng, kind, n, 16), _signal(rng, "tone", n, 15)])
        b = o.frame_encode_i32(x)
        blobs.append(b)
        refs.append(o.frame_decode_i32(b, 2)[0])
    stream = np.frombuffer(b"".join(blobs), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.uint64)
    got, chunks, segs = _decode_modes(stream, offs, 2)
    assert chunks == 2 and segs >= 2 * 2 * 50
    for m, dec in got.items():
        for f in range(60):
            for c in range(2):
                assert np.array_equal(dec[f][c], refs[f][c]), (m, f, c)
    # the 16-bit entry point on the same stream (sela_hip_decode: the frames' samples back to back, interleaved)
    back = codec.decode_host(stream, offs, 2)
    want = np.concatenate([np.stack(r, axis=1) for r in refs]).astype(np.uint32).astype(np.uint16).view(np.int16)
    assert np.array_equal(np.asarray(back).reshape(-1, 2), want)


def _hostile_frame(rng, ch, n_lo, n_hi, big):
    subs = []
    roomy = rng.random() < 0.75
    n_frame = int(rng.integers(n_lo, n_hi))
    for c in range(ch):
        order = int(rng.integers(0, 101))
        ck = int(rng.integers(0, 12))
        rk = int(rng.integers(0, 20))
        n = n_frame if rng.random() < 0.8 else int(rng.integers(n_lo, n_hi))
        cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
        rwords = (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8)) if roomy else int(rng.integers(1, 1 + (n * (rk + 3)) // 32 + 8))
        rwords = min(rwords, 65535)
        style = rng.random()
        if style < 0.45:    # sparse words: short unary runs, most streams hold their values
            mk = lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)  # noqa: E731
        elif style < 0.8:
            mk = lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)  # noqa: E731
        else:               # stretches of all-ones words among sparse ones: unary runs across words, zones and segments
            def mk(m):
                w = (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)
                for _ in range(int(rng.integers(1, 4))):
                    if m > 4:
                        a = int(rng.integers(0, m - 2))
                        w[a: a + int(rng.integers(1, max(2, min(m - a, 80 if big else 6))))] = 0xFFFFFFFF
                return w
        cw, rw = mk(cwords), mk(rwords)
        typ = 1 if (c > 0 and rng.random() < 0.3) else 0
        parent = int(rng.integers(0, c)) if typ else c
        subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
    return bytes.fromhex("00ff55aa") + b"".join(subs)


@pytest.mark.parametrize("shape", ["short", "long"])
def test_hostile_streams_of_any_length_through_the_segment_parser(gpu, shape):  # noqa: F811
    """Frames no encoder wrote -- random words as Rice streams (sparse, dense, with stretches of all-ones words), random
    parameters, orders and lengths up to 700 / up to 30,000 samples (streams of many segments) -- through the product's decoder
    and with every subframe by segments: where the oracle reads past a stream's end or a coefficient leaves int64 the call
    fails, everywhere else every sample is the oracle's."""
    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(2024 if shape == "short" else 4048)
    trials, n_lo, n_hi = (200, 1, 700) if shape == "short" else (60, 3000, 30000)
    same = failed = 0
    for trial in range(trials):
        ch = int(rng.integers(1, 4))
        blob = _hostile_frame(rng, ch, n_lo, n_hi, shape == "long")
        fl = C.c_uint32(0)
        b = np.frombuffer(blob, np.uint8).copy()
        out = np.zeros((ch, n_hi), np.int32)
        counts = np.zeros(ch, np.uint32)
        used = o._fdec32(b, ch, out, n_hi, counts, C.byref(fl))
        assert used == len(blob)
        bad = fl.value & (8 | 2 | 32)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME
        offs = _one(len(blob))
        if bad:
            for mode in (1, 2):
                capi.lib().sela_hip_debug_standard_first(mode)
                try:
                    with pytest.raises(capi.SelaHipError) as err:
                        codec.decode_i32(b, offs, ch)
                finally:
                    capi.lib().sela_hip_debug_standard_first(-1)
                assert err.value.code in (-5, -6), (trial, mode, hex(fl.value))
            failed += 1
        else:
            got, _, _ = _decode_modes(b, offs, ch, modes=(1, 2))
            for m, dec in got.items():
                for c in range(ch):
                    assert np.array_equal(dec[0][c], out[c, : int(counts[c])]), (shape, trial, "mode", m, "channel", c, hex(fl.value))
            same += 1
    assert same >= trials // 5 and failed >= trials // 10, (same, failed)


def test_sample_generator_of_any_length_is_the_frame_kernels_recurrence(gpu):  # noqa: F811
    """sela_hip_lpc_decode_n (lpc::SampleGenerator, src/lpc/sample_generator.cpp:11-39) for lengths on both sides of every block
    of 64 and orders on both sides of the ring sizes (48 / 60 / 64 / 100), residues up to 24 bits (the folded form's range check
    fails inside a block: the exact form takes over): the oracle's samples and Q35 predictors."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(31)
    for n in (1, 63, 64, 65, 128, 191, 1000, 2048, 2049, 6000):
        for order in (0, 1, 2, 47, 48, 49, 60, 61, 64, 65, 100):
            q = np.zeros((3, 100), np.int32)
            q[:, :order] = rng.integers(-6, 7, (3, order))
            if order:
                q[:, 0] = rng.integers(-64, -40, 3)
            res = np.stack([rng.integers(-200, 201, n), rng.integers(-(1 << 23), 1 << 23, n), (rng.random(n) < 0.02) * rng.integers(-(1 << 20), 1 << 20, n)]).astype(np.int32)
            orders = np.full(3, order, np.int32)
            got, coefs = codec.lpc_decode_n(orders, q, res, want_coefficients=True)
            for b in range(3):
                want = o.lpc_synth(order, q[b, :order], res[b])
                assert np.array_equal(got[b], want), (n, order, b)
                assert np.array_equal(coefs[b, : order + 1], o.lpc_coeffs(order, q[b, :order])[: order + 1]), (n, order, b)


def test_frames_that_are_not_whole_words_are_left_to_the_serial_kernel(gpu):  # noqa: F811
    """The fast kernel reads whole words: a frame with a stray byte behind it (and the next frame, which then lies at an odd
    offset) is not its business -- the serial kernel decodes the chunk, same samples."""
    from sela_amd import capi, codec

    o = oracle()
    blobs, refs = [], []
    for seed in (5, 6):
        pcm = synth_pcm(1000, 2, seed).reshape(1000, 2)
        b = o.frame_encode_i32(np.ascontiguousarray(pcm.T.astype(np.int32)))
        blobs.append(b)
        refs.append(o.frame_decode_i32(b, 2)[0])
    stream = np.frombuffer(blobs[0] + b"\x00" + blobs[1], np.uint8)
    offs = np.array([0, len(blobs[0]) + 1, len(blobs[0]) + 1 + len(blobs[1])], np.uint64)
    before = capi.lib().sela_hip_debug_standard_chunks()
    dec = codec.decode_i32(stream, offs, 2, stride=1000)
    assert capi.lib().sela_hip_debug_standard_chunks() == before
    for f in range(2):
        for c in range(2):
            assert np.array_equal(dec[f][c], refs[f][c])
