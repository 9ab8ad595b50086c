"""GPU parity tests, by subject: the decoder for subframes of ANY length and 32-bit samples (k_decode_subframes32 in
sela_decode32.hip -- the fast decoder's lane-parallel Rice parse, cut into segments, and its tuned synthesis with the length a
run-time value -- with k_generic_decode, the serial walk, as the judge of streams it will not touch).  Everything is compared,
bit for bit, with the oracle (oracle/sela_oracle.c, pinned against the unmodified reference): src/frame/frame_decoder.cpp:11-72,
src/rice/rice_decoder.cpp:11-61, src/lpc/sample_generator.cpp:11-39."""
import ctypes as C
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
import struct
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import DECODE_LENGTHS, _both_decoders, _decode_modes, _hostile_frame, _one, _rice_words, _signal, _subframe_words, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


def test_crafted_frames_decode_like_the_reference(gpu, generic_kats):  # noqa: F811
    """Subframes spliced by hand: channels of different lengths, a chain of dependent subframes, a dependent subframe ahead of
    its parent, one channel named twice, a long parent, an unknown subframe type -- FrameDecoder's answer, channel by channel."""
    from sela_amd import capi, codec

    for name in generic_kats["crafted_names"]:
        blob = generic_kats[f"crafted/{name}/bytes"]
        ch = int(generic_kats[f"crafted/{name}/channels"])
        dec = codec.decode_i32(blob, _one(len(blob)), ch)[0]
        lengths = []
        for c in range(ch):
            want = generic_kats[f"crafted/{name}/decoded{c}"]
            assert np.array_equal(dec[c], want), (name, c)
            lengths.append(len(want))
        if len(set(lengths)) > 1:  # no interleaved PCM exists for channels of different lengths
            with pytest.raises(capi.SelaHipError) as err:
                codec.decode_host(blob, _one(len(blob)), ch)
            assert err.value.code == -5, name
        else:
            back = codec.decode_host(blob, _one(len(blob)), ch)
            want = np.stack([generic_kats[f"crafted/{name}/decoded{c}"] for c in range(ch)], axis=1)
            assert np.array_equal(back, want.astype(np.uint32).astype(np.uint16).view(np.int16)), name


def test_hostile_streams_through_the_any_length_decoder(gpu):  # noqa: F811
    """Frames no encoder wrote: random words as Rice streams, random parameters, orders and lengths, coefficient values far
    outside the dequantisation tables.  Where the reference is undefined -- it reads past a stream's end, a coefficient indexes
    past the tables or leaves int64, a subframe is not longer than its order -- the call must fail (EFORMAT / ERANGE);
    everywhere else every sample must be the oracle's -- wrap-around arithmetic, 32-bit results."""
    import ctypes as C
    import struct

    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(99)
    same = failed = 0
    for trial in range(250):
        ch = int(rng.integers(1, 4))
        subs = []
        n_frame = int(rng.integers(1, 700))
        roomy = rng.random() < 0.7  # (most frames get streams long enough for their values; the rest run dry)
        for c in range(ch):
            order = int(rng.integers(0, 101))
            ck = int(rng.integers(0, 12))
            rk = int(rng.integers(0, 20))
            n = n_frame if rng.random() < 0.8 else int(rng.integers(1, 700))
            cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
            rwords = (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8)) if roomy else int(rng.integers(1, 1 + (n * (rk + 3)) // 32 + 8))
            dense = rng.random() < 0.5  # sparse words = short unary runs, so that most streams do hold their values
            mk = (lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)
                             & (rng.integers(0, 1 << 32, m, dtype=np.uint64) if not dense else np.uint64(0xFFFFFFFF))).astype(np.uint32))
            cw, rw = mk(cwords), mk(rwords)
            if order and rng.random() < 0.7:  # coefficients the tables hold (a value outside [-64, 63] indexes past them in the reference: refused)
                cw = np.concatenate([_rice_words(rng.integers(-64, 64, order), ck), mk(int(rng.integers(0, 3)))])
                cwords = len(cw)
            typ = 1 if (c > 0 and rng.random() < 0.3) else 0
            parent = int(rng.integers(0, c)) if typ else c
            subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
        blob = bytes.fromhex("00ff55aa") + b"".join(subs)
        fl = C.c_uint32(0)
        b = np.frombuffer(blob, np.uint8).copy()
        out = np.zeros((ch, 700), np.int32)
        counts = np.zeros(ch, np.uint32)
        used = o._fdec32(b, ch, out, 700, counts, C.byref(fl))
        assert used == len(blob)
        bad = fl.value & (8 | 2 | 32 | 1 | 128)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME, Q_RANGE, SHORT_BLOCK: what the reference leaves undefined is refused
        offs = np.array([0, len(blob)], np.uint64)
        if bad:
            with pytest.raises(capi.SelaHipError) as err:
                codec.decode_i32(b, offs, ch)
            assert err.value.code in (-5, -6), (trial, hex(fl.value))
            failed += 1
        else:
            dec = codec.decode_i32(b, offs, ch)[0]
            for c in range(ch):
                assert np.array_equal(dec[c], out[c, : int(counts[c])]), (trial, c, hex(fl.value))
            same += 1
    assert same >= 60 and failed >= 60, (same, failed)


def test_standard_subframes_come_out_as_32_bit_samples_on_the_fast_parse_and_synthesis(gpu, kats):  # noqa: F811
    """sela_hip_decode_i32 (frame::FrameDecoder behind it) on 2048-sample frames: k_decode_subframes32 must give the oracle's 32-bit
    samples -- an encoder's frames (stereo with both decisions, mono, three channels; one call of many frames and calls of one),
    frames of samples far beyond 16 bits, and the KAT blocks -- and so must the any-length kernel on the same bytes."""
    from sela_amd import codec

    o = oracle()
    for pcm in (synth_frames(40, 2, 0), synth_frames(7, 1, 2), synth_frames(5, 3, 4)):
        ch = pcm.shape[2]
        frames, offs = codec.encode_host(pcm)
        offered, alone, took = _both_decoders(frames, offs, ch)
        assert took == 1
        for f in range(pcm.shape[0]):
            for c in range(ch):
                assert np.array_equal(offered[f][c], pcm[f, :, c].astype(np.int32)), (f, c)
                assert np.array_equal(alone[f][c], offered[f][c]), (f, c)
        one = codec.decode_i32(frames[int(offs[3]):int(offs[4])], _one(int(offs[4] - offs[3])), ch)[0]  # (the library's own choice)
        assert np.array_equal(np.stack(one, axis=1), pcm[3].astype(np.int32))
    # samples no WAV file holds: 21-bit tones and noise, a 32-bit frame API's business
    rng = np.random.default_rng(5)
    t = np.arange(2048)
    wide = np.stack([
        np.round((1 << 20) * 0.9 * np.sin(t * 0.01) + rng.normal(0, 3000, 2048)),
        np.round((1 << 19) * np.sin(t * 0.31 + 1) + rng.normal(0, 10, 2048)),
        rng.integers(-(1 << 20), 1 << 20, 2048).astype(np.float64),
    ]).astype(np.int32)
    taken = 0
    for chans in (wide[:1], wide[1:2], wide[:2], wide):
        frames, offs = codec.encode_i32(chans[None])
        want = o.frame_encode_i32(chans)
        assert frames.tobytes() == want
        ref_dec, used = o.frame_decode_i32(want, len(chans))
        offered, alone, took = _both_decoders(frames, offs, len(chans))
        # the fast kernel takes every clean frame: a subframe that fits the parser's plan (1072 aligned words) in one piece, the
        # uniform 21-bit noise of the third channel by segments
        assert took == 1 and used == len(want), (len(chans), _subframe_words(want, len(chans)))
        taken += took
        for c in range(len(chans)):
            assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c]), c
    assert taken >= 2
    names = [str(n) for n in kats["blk_names"]]
    blocks = np.stack([kats[f"blk/{n}/samples"] for n in names]).astype(np.int32)  # (diff_extreme is 17-bit)
    frames, offs = codec.encode_i32(blocks[:, None, :])
    taken = 0
    for i, name in enumerate(names):  # (a call per block: the standard kernel takes a chunk whole or not at all)
        blob = frames[int(offs[i]):int(offs[i + 1])]
        ref_dec, _ = o.frame_decode_i32(blob.tobytes(), 1)
        offered, alone, took = _both_decoders(blob, _one(len(blob)), 1)
        assert took == 1, name
        taken += took
        assert np.array_equal(offered[0][0], ref_dec[0]) and np.array_equal(alone[0][0], ref_dec[0]), name
    assert taken >= len(names) // 2


def test_chunks_that_mix_the_one_piece_parse_with_segments(gpu):  # noqa: F811
    """A chunk in which some subframes take the frame kernel's one-piece parse and others go by segments -- a frame of another
    length among 2048-sample ones, a Rice stream beyond the parser's plan (incompressible full-scale noise), a subframe type the
    reference ignores -- comes out of the fast kernel in one go: same answer as the serial kernel alone."""
    import struct

    from sela_amd import codec

    o = oracle()
    std = synth_frames(6, 2, 9)
    blobs = [o.frame_encode(std[f]) for f in range(6)]
    odd = synth_pcm(1000 * 2, 2, 3).reshape(2, 1000, 2)
    blobs.insert(3, o.frame_encode(odd[0]))
    stream = np.frombuffer(b"".join(blobs), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.uint64)
    offered, alone, took = _both_decoders(stream, offs, 2)
    assert took == 1
    for f in range(len(blobs)):
        ref_dec, _ = o.frame_decode_i32(blobs[f], 2)
        for c in range(2):
            assert np.array_equal(offered[f][c], ref_dec[c]) and np.array_equal(alone[f][c], ref_dec[c]), (f, c)
    # beyond the plan: more than 1072 aligned words in one subframe
    noise = np.random.default_rng(2).integers(-(1 << 19), 1 << 19, (1, 2, 2048)).astype(np.int32)
    frames, fo = codec.encode_i32(noise)
    assert max(_subframe_words(frames.tobytes(), 2)) > 1072
    offered, alone, took = _both_decoders(frames, fo, 2)
    assert took == 1
    ref_dec, _ = o.frame_decode_i32(frames.tobytes(), 2)
    for c in range(2):
        assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c])
    # a subframe of a type the reference's two passes both skip (src/frame/frame_decoder.cpp:17-69): its channel stays empty
    blob = bytearray(blobs[0])
    second = 4 + 12 + 4 * (struct.unpack_from("<H", blob, 8)[0] + struct.unpack_from("<H", blob, 4 + 7 + 4 * struct.unpack_from("<H", blob, 8)[0] + 1)[0])
    assert blob[second] == 1  # (the second subframe's channel byte)
    blob[second + 1] = 7
    b = np.frombuffer(bytes(blob), np.uint8)
    ref_dec, used = o.frame_decode_i32(bytes(blob), 2)
    offered, alone, took = _both_decoders(b, _one(len(blob)), 2)
    assert took == 1 and used == len(blob)
    for c in range(2):
        assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c]), c


def test_hostile_2048_sample_frames_through_both_decoders(gpu):  # noqa: F811
    """Frames of 2048 samples no encoder wrote (random words as Rice streams, random parameters and orders, coefficient values
    outside the tables): whatever the standard kernel takes must be the oracle's wrap-around arithmetic in 32 bits, whatever it
    leaves alone the any-length kernel's answer; failures (a stream that runs dry, a coefficient beyond int64) fail both ways."""
    import ctypes as C
    import struct

    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(123)
    same = failed = taken = 0
    import os
    trials = int(os.environ.get("SELA_HOSTILE_TRIALS", "160"))  # (a long soak: SELA_HOSTILE_TRIALS=5000)
    for trial in range(trials):
        ch = int(rng.integers(1, 4))
        subs = []
        roomy = rng.random() < 0.75
        for c in range(ch):
            order = int(rng.integers(0, 101))
            ck = int(rng.integers(0, 10))
            rk = int(rng.integers(0, 14))
            n = 2048
            cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
            rwords = min(1040 - cwords, (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8))) if roomy else int(rng.integers(1, 900))
            dense = rng.random() < 0.3
            mk = (lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)
                             & (rng.integers(0, 1 << 32, m, dtype=np.uint64) if not dense else np.uint64(0xFFFFFFFF))).astype(np.uint32))
            cw, rw = mk(cwords), mk(rwords)
            if order and rng.random() < 0.7:  # coefficients the tables hold (a value outside [-64, 63] indexes past them in the reference: refused)
                cw = np.concatenate([_rice_words(rng.integers(-64, 64, order), ck), mk(int(rng.integers(0, 3)))])
                cwords = len(cw)
            typ = 1 if (c > 0 and rng.random() < 0.3) else 0
            parent = int(rng.integers(0, c)) if typ else c
            subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
        blob = bytes.fromhex("00ff55aa") + b"".join(subs)
        fl = C.c_uint32(0)
        b = np.frombuffer(blob, np.uint8).copy()
        out = np.zeros((ch, 2048), np.int32)
        counts = np.zeros(ch, np.uint32)
        used = o._fdec32(b, ch, out, 2048, counts, C.byref(fl))
        assert used == len(blob)
        bad = fl.value & (8 | 2 | 32 | 1 | 128)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME, Q_RANGE, SHORT_BLOCK: what the reference leaves undefined is refused
        offs = np.array([0, len(blob)], np.uint64)
        if bad:
            for mode in (1, 0):
                capi.lib().sela_hip_debug_standard_first(mode)
                try:
                    with pytest.raises(capi.SelaHipError) as err:
                        codec.decode_i32(b, offs, ch)
                finally:
                    capi.lib().sela_hip_debug_standard_first(-1)
                assert err.value.code in (-5, -6), (trial, mode, hex(fl.value))
            failed += 1
        else:
            offered, alone, took = _both_decoders(b, offs, ch)
            taken += took
            for c in range(ch):
                assert np.array_equal(offered[0][c], out[c, : int(counts[c])]), (trial, c, took, hex(fl.value))
                assert np.array_equal(alone[0][c], out[c, : int(counts[c])]), (trial, c, hex(fl.value))
            same += 1
    assert same >= 40 and failed >= 20 and taken >= 20, (same, failed, taken)


@pytest.mark.parametrize("n", DECODE_LENGTHS)
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


@pytest.mark.parametrize("shape", ["short", "long"])
def test_hostile_streams_of_any_length_through_the_segment_parser(gpu, shape):  # noqa: F811
    """Frames no encoder wrote -- random words as Rice streams (sparse, dense, with stretches of all-ones words), random
    parameters, orders and lengths up to 700 / up to 30,000 samples (streams of many segments) -- through the product's decoder
    and with every subframe by segments: where the oracle reads past a stream's end or a coefficient leaves int64 the call
    fails, everywhere else every sample is the oracle's."""
    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(2024 if shape == "short" else 4048)
    import os

    scale = int(os.environ.get("SELA_HOSTILE_SCALE", "1"))  # (a long soak: SELA_HOSTILE_SCALE=20)
    trials, n_lo, n_hi = (200 * scale, 1, 700) if shape == "short" else (60 * scale, 3000, 30000)
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
        bad = fl.value & (8 | 2 | 32 | 1 | 128)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME, Q_RANGE, SHORT_BLOCK: what the reference leaves undefined is refused
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
    from sela_amd import capi, codec

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
            if n <= order:  # the reference writes samples[1 .. order] whatever the length (src/lpc/sample_generator.cpp:14-22): refused
                with pytest.raises(capi.SelaHipError) as err:
                    codec.lpc_decode_n(orders, q, res)
                assert err.value.code == -6
                continue
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
