"""GPU parity tests, by subject: the encoder for blocks of ANY length and 32-bit samples (k_generic_analyse / plan / pack /
assemble in sela_generic.hip behind sela_hip_encode_i32, sela_hip_encode with samples_per_channel != 2048 and
sela_hip_lpc_encode_n) against the oracle (oracle/sela_oracle.c, pinned against the unmodified reference):
src/lpc/residue_generator.cpp:12-134, src/lpc/linear_predictor.cpp:16-61, src/rice/rice_encoder.cpp:12-81,
src/frame/frame_encoder.cpp:11-102.  Bit-exact: frame bytes, orders, quantised coefficients, residues."""
import hashlib
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
import generic_cases as gc
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import ENCODE_LENGTHS, _signal, _wrap_taps, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", gc.LENGTHS)
def test_generic_frames_against_the_reference_fixtures(gpu, generic_digests, generic_kats, n):  # noqa: F811
    """n in {128, 1000, 2047, 2049, 4096, 65535} x {mono, stereo (difference wins / loses), three channels} x {16-bit, 17-bit}:
    frame bytes and decoded int32 channels equal the reference's (digests; whole bytes for the short ones)."""
    from sela_amd import codec

    o = oracle()
    for label, nn, kind, wide in gc.all_cases():
        if nn != n:
            continue
        g = generic_digests[label]
        x = gc.case_input(n, kind, wide)
        assert gc.sha(x) == g["input_sha256"], label
        frames, offs = codec.encode_i32(x[None])
        blob = frames.tobytes()
        if hashlib.sha256(blob).hexdigest() != g["frame_sha256"]:  # say where, with the oracle's help
            want = o.frame_encode_i32(x)
            first = next((i for i, (a, b) in enumerate(zip(blob, want)) if a != b), min(len(blob), len(want)))
            pytest.fail(f"{label}: frame bytes differ from the reference's at byte {first} ({len(blob)} vs {len(want)} bytes)")
        assert int(offs[1]) == g["frame_bytes"], label
        if f"{label}/bytes" in generic_kats:
            assert blob == generic_kats[f"{label}/bytes"].tobytes(), label
        dec = codec.decode_i32(frames, offs, x.shape[0])[0]
        assert gc.sha_channels(dec) == g["decoded_sha256"], label
        assert all(np.array_equal(a, b) for a, b in zip(dec, x)) == g["lossless"], label
        if not wide:  # the int16 entry points: the same frame, the same samples
            pcm = np.ascontiguousarray(x.T.astype(np.int16))[None]
            f16, o16 = codec.encode_host(pcm)
            assert f16.tobytes() == blob and int(o16[1]) == len(blob), label
            back = codec.decode_host(frames, offs, x.shape[0])
            assert back.shape == (n, x.shape[0]) and np.array_equal(back, pcm[0]), label
        else:  # 17-bit samples through the 16-bit writer: truncated like src/file/wav_file.cpp:248-251
            back = codec.decode_host(frames, offs, x.shape[0])
            assert np.array_equal(back, x.T.astype(np.uint32).astype(np.uint16).view(np.int16)), label


def test_batches_of_odd_frames_and_mixed_streams(gpu):  # noqa: F811
    """Many frames per call (offsets, chunking), and ONE stream whose frames have different lengths, 2048 among them."""
    from sela_amd import codec

    o = oracle()
    blobs, pcms = [], []
    for n, ch, nf, track in ((1000, 2, 37, 3), (2047, 1, 9, 4), (2049, 3, 5, 5), (4096, 2, 11, 6)):
        pcm = synth_pcm(n * nf, ch, track).reshape(nf, n, ch)
        frames, offs = codec.encode_host(pcm)
        want = [o.frame_encode(pcm[f]) for f in range(nf)]
        assert frames.tobytes() == b"".join(want), (n, ch)
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist(), (n, ch)
        back = codec.decode_host(frames, offs, ch)
        assert np.array_equal(back, pcm.reshape(-1, ch)), (n, ch)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        f32, o32 = codec.encode_i32(planar)
        assert f32.tobytes() == frames.tobytes() and np.array_equal(o32, offs)
        if ch == 2:
            blobs += want
            pcms += [pcm[f] for f in range(nf)]
    # a stereo stream of 1000-, 4096- and 2048-sample frames, interleaved
    std = synth_frames(6, 2, 9)
    blobs += [o.frame_encode(std[f]) for f in range(6)]
    pcms += [std[f] for f in range(6)]
    order = np.random.default_rng(0).permutation(len(blobs))
    stream = b"".join(blobs[i] for i in order)
    offs = np.concatenate([[0], np.cumsum([len(blobs[i]) for i in order])]).astype(np.uint64)
    so, largest = codec.index_samples(np.frombuffer(stream, np.uint8), offs, 2)
    assert largest == 4096 and so.tolist() == np.concatenate([[0], np.cumsum([len(pcms[i]) for i in order])]).tolist()
    back = codec.decode_host(np.frombuffer(stream, np.uint8), offs, 2)
    assert np.array_equal(back, np.concatenate([pcms[i] for i in order]))
    dec = codec.decode_i32(np.frombuffer(stream, np.uint8), offs, 2)
    for j, i in enumerate(order):
        assert np.array_equal(np.stack(dec[j], axis=1), pcms[i].astype(np.int32)), j


def test_2048_through_the_generic_route_equals_the_fast_kernels(gpu, kats):  # noqa: F811
    """The same frames through both routes: bytes and samples identical (stereo with both decisions, mono, the KAT blocks)."""
    from sela_amd import codec

    for pcm in (synth_frames(24, 2, 0), synth_frames(7, 1, 2), synth_frames(5, 3, 4)):
        fast, fo = codec.encode_host(pcm)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        gen, go = codec.encode_i32(planar)
        assert gen.tobytes() == fast.tobytes() and np.array_equal(go, fo)
        dec = codec.decode_i32(fast, fo, pcm.shape[2])
        for f in range(pcm.shape[0]):
            assert np.array_equal(np.stack(dec[f], axis=1), pcm[f].astype(np.int32)), f
    names = [str(n) for n in kats["blk_names"]]
    blocks = np.stack([kats[f"blk/{n}/samples"] for n in names]).astype(np.int32)  # (diff_extreme is 17-bit)
    gen, go = codec.encode_i32(blocks[:, None, :])
    o = oracle()
    assert gen.tobytes() == b"".join(o.frame_encode_i32(b[None]) for b in blocks)


def test_what_the_reference_cannot_answer_is_refused(gpu):  # noqa: F811
    """A block not longer than its own order (the reference reads past its vector), lengths the u16 field cannot say,
    a stride too small, residues beyond the int32 zig-zag."""
    from sela_amd import capi, codec

    rng = np.random.default_rng(2)
    noise = rng.integers(-20000, 20000, (1, 1, 40)).astype(np.int32)  # white noise: the order comes out above 40
    assert oracle().lpc_analyze(noise[0, 0])[0] >= 40
    with pytest.raises(capi.SelaHipError) as err:
        codec.encode_i32(noise)
    assert err.value.code == -6
    for bad in (0, 65536):
        lib = capi.lib()
        buf = np.zeros(16, np.int16)
        out = np.zeros(4096, np.uint8)
        offs = np.zeros(2, np.uint64)
        assert lib.sela_hip_encode(buf.ctypes.data, 1, 1, bad, out.ctypes.data, out.nbytes, offs.ctypes.data) == -2
    x = gc.case_input(1000, "mono", False)
    frames, offs = codec.encode_i32(x[None])
    with pytest.raises(capi.SelaHipError) as err:
        codec.decode_i32(frames, offs, 1, stride=999)
    assert err.value.code == -4
    wild = (rng.integers(-(1 << 31), 1 << 31, (1, 1, 500))).astype(np.int32)  # full-range int32: residues overflow the zig-zag
    with pytest.raises(capi.SelaHipError) as err:
        codec.encode_i32(wild)
    assert err.value.code == -6
    # a short frame that IS longer than its order: a constant block has order 1
    flat = np.full((1, 1, 2), 5, np.int32)
    frames, offs = codec.encode_i32(flat)
    assert frames.tobytes() == oracle().frame_encode_i32(flat[0])


def test_lpc_stages_of_any_length(gpu):  # noqa: F811
    """lpc::ResidueGenerator / SampleGenerator on vectors of any length and any 32-bit value, against the oracle."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(8)
    for n, amp in ((101, 3000), (500, 32767), (2048, 1 << 20), (4096, 65535), (10000, 1 << 22)):
        t = np.arange(n)
        blocks = np.stack([np.clip(np.round(amp * 0.6 * np.sin(t * (0.01 + 0.003 * b)) + rng.normal(0, amp * 0.02, n)), -amp, amp) for b in range(6)]).astype(np.int32)
        order, q, res = codec.lpc_encode_n(blocks)
        for b in range(len(blocks)):
            wo, wq, wr = o.lpc_analyze(blocks[b])
            assert order[b] == wo and np.array_equal(q[b, :wo], wq) and not q[b, wo:].any() and np.array_equal(res[b], wr), (n, b)
        back, coefs = codec.lpc_decode_n(order, q, res, want_coefficients=True)
        for b in range(len(blocks)):
            assert np.array_equal(back[b], o.lpc_synth(int(order[b]), q[b, : order[b]], res[b])), (n, b)
            assert np.array_equal(coefs[b, : order[b] + 1], o.lpc_coeffs(int(order[b]), q[b, : order[b]])), (n, b)
    # the 2048 entry with samples beyond 17 bits goes the same way, and zeroes q beyond the order
    big = (rng.integers(-(1 << 19), 1 << 19, (3, 2048))).astype(np.int32)
    order, q, res = codec.lpc_encode(big)
    for b in range(3):
        wo, wq, wr = o.lpc_analyze(big[b])
        assert order[b] == wo and np.array_equal(q[b, :wo], wq) and not q[b, wo:].any() and np.array_equal(res[b], wr)


def test_random_shapes_against_the_oracle(gpu):  # noqa: F811
    """300 frames of random shape -- 1 .. 6000 samples, 1 .. 6 channels, amplitudes from 1 to 2^20, silent and constant
    channels, near-copies (difference coding), lengths around the analysis' landmarks (63 .. 65, 100 .. 102, 127 .. 129) --
    through sela_hip_encode_i32 / sela_hip_decode_i32 against the oracle; a frame with a block not longer than its own order
    must be refused with SELA_HIP_ERANGE, exactly those."""
    from sela_amd import capi, codec

    import os

    o = oracle()
    trials = int(os.environ.get("SELA_SHAPES_TRIALS", "300"))  # (a soak: SELA_SHAPES_TRIALS=6000 SELA_SHAPES_SEED=...)
    rng = np.random.default_rng(int(os.environ.get("SELA_SHAPES_SEED", "77")))
    landmarks = [1, 2, 3, 31, 63, 64, 65, 100, 101, 102, 103, 127, 128, 129, 2047, 2048, 2049]
    refused = coded = 0
    for trial in range(trials):
        n = int(rng.choice(landmarks)) if trial % 3 == 0 else int(rng.integers(1, 6001))
        ch = int(rng.integers(1, 7))
        amp = int(2 ** rng.uniform(0, 20))
        t = np.arange(n)
        x = np.zeros((ch, n), np.int64)
        for c in range(ch):
            kind = rng.integers(0, 6)
            if kind == 0:
                x[c] = 0
            elif kind == 1:
                x[c] = rng.integers(-amp, amp + 1)
            elif kind == 2:
                x[c] = rng.integers(-amp, amp + 1, n)
            else:
                x[c] = np.round(amp * 0.7 * np.sin(t * rng.uniform(0.001, 1.5) + rng.uniform(0, 6)) + rng.normal(0, amp * rng.choice([0.0, 0.01, 0.2]), n))
        if ch == 2 and rng.random() < 0.5:
            x[1] = x[0] - rng.integers(-2, 3, n)
        x = np.clip(x, -(1 << 20), 1 << 20).astype(np.int32)
        # the orders of every block the frame encoder analyses (the difference signal of a stereo frame included)
        signals = [x[c] for c in range(ch)] + ([(x[0] - x[1]).astype(np.int32)] if ch == 2 else [])
        short = any(o.lpc_analyze(s)[0] >= n for s in signals)
        if short:
            with pytest.raises(capi.SelaHipError) as err:
                codec.encode_i32(x[None])
            assert err.value.code == -6, (trial, n, ch)
            refused += 1
            continue
        frames, offs = codec.encode_i32(x[None])
        want = o.frame_encode_i32(x)
        assert frames.tobytes() == want, (trial, n, ch, amp)
        dec = codec.decode_i32(frames, offs, ch)[0]
        ref_dec, used = o.frame_decode_i32(want, ch)
        assert used == len(want)
        for c in range(ch):
            assert np.array_equal(dec[c], ref_dec[c]), (trial, n, ch, c)
        coded += 1
    assert refused >= 5 and coded >= 2 * trials // 3, (refused, coded)


@pytest.mark.parametrize("n", ENCODE_LENGTHS)
def test_frames_of_any_length_equal_the_oracles_bytes_in_both_forms_of_the_residue_filter(gpu, n):  # noqa: F811
    """Lengths on both sides of every boundary the kernel has (chunks of 64, the ring of 256, stretches of 1024 with 128 samples
    of history), mono / stereo / three channels, silence, DC, tones, clicks, noise, 16- and 21-bit: the frame bytes are the
    oracle's with the residue filter chosen by the block's bound (FP64 taps where exact) and forced onto the 64-bit taps."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(5000 + n)
    coded = 0
    for kinds, bits in ((("silence",), 16), (("dc", "tone"), 16), (("tone", "noise"), 16), (("sparse", "tone", "noise"), 16), (("noise",), 21), (("tone", "tone"), 21),
                        (("tone",), 24)):
        x = np.stack([_signal(rng, k, n, bits) for k in kinds])
        want = None
        for wrap in (False, True):
            _wrap_taps(wrap)
            try:
                try:
                    frames, offs = codec.encode_i32(x[None])
                except Exception:  # a block not longer than its own order (the reference reads past its vector), or residues beyond the zig-zag
                    frames = None
            finally:
                _wrap_taps(False)
            if want is None:
                fl_ok = True
                try:
                    want = o.frame_encode_i32(x)
                except Exception:
                    fl_ok = False
                if frames is None:
                    break
                assert fl_ok
            if frames is None:
                pytest.fail(f"n {n} {kinds} {bits}: refused on the wrap-around taps only")
            assert frames.tobytes() == want, (n, kinds, bits, "wrap taps" if wrap else "by the bound")
            coded += 1
    assert coded >= 8 or n < 128


def test_the_lpc_stage_of_any_length_equals_the_oracle(gpu):  # noqa: F811
    """sela_hip_lpc_encode_n: order, quantised coefficients and residues of blocks of many lengths, both forms."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(8)
    for n in (130, 1000, 2048, 2500, 5000, 70000):
        blocks = np.stack([_signal(rng, k, n, b) for k, b in (("tone", 16), ("noise", 16), ("sparse", 16), ("tone", 21), ("noise", 21), ("dc", 12))])
        for wrap in (False, True):
            _wrap_taps(wrap)
            try:
                order, q, res = codec.lpc_encode_n(blocks)
            finally:
                _wrap_taps(False)
            for i in range(len(blocks)):
                wo, wq, wr = o.lpc_analyze(blocks[i])
                assert order[i] == wo and np.array_equal(q[i, :wo], wq) and not q[i, wo:].any(), (n, i, wrap)
                assert np.array_equal(res[i], wr), (n, i, wrap)


def test_a_batch_of_frames_equals_frame_by_frame(gpu):  # noqa: F811
    """3000 stereo frames of 777 samples in one call (more waves than the device holds at once) = the oracle frame by frame."""
    from sela_amd import codec
    from sela_amd.synth import synth_pcm

    o = oracle()
    n, nf = 777, 3000
    pcm = synth_pcm(n * nf, 2, 41).reshape(nf, n, 2)
    planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
    frames, offs = codec.encode_i32(planar)
    for f in list(range(0, nf, 97)) + [nf - 1]:
        assert frames[int(offs[f]):int(offs[f + 1])].tobytes() == o.frame_encode_i32(planar[f]), f
    f16, o16 = codec.encode_host(pcm)
    assert np.array_equal(o16, offs) and np.array_equal(f16, frames)


def test_frames_whose_channels_differ_in_length_equal_the_references_bytes(gpu, ragged_digests, ragged_kats):  # noqa: F811
    """frame::FrameEncoder on channels of different lengths (src/frame/frame_encoder.cpp:20-24,73-98: every channel at its own
    samples[i].size(), the stereo difference over channel 1's length): the frame bytes the unmodified reference wrote
    (tests/golden/ragged.json, ragged_kats.npz), decoded back to the reference's channels; and the one shape on which the
    reference indexes past its vector -- stereo, channel 0 the shorter -- refused."""
    import hashlib

    import generic_cases as gc
    from sela_amd import capi, codec

    for label, chans in gc.ragged_cases():
        g = ragged_digests[label]
        assert gc.sha_channels(chans) == g["input_sha256"], label
        blob = codec.encode_ragged(chans)
        assert len(blob) == g["frame_bytes"] and hashlib.sha256(blob).hexdigest() == g["frame_sha256"], label
        if f"{label}/bytes" in ragged_kats:
            assert blob == ragged_kats[f"{label}/bytes"].tobytes(), label
        dec = codec.decode_i32(np.frombuffer(blob, np.uint8), np.array([0, len(blob)], np.uint64), len(chans))[0]
        assert gc.sha_channels(dec) == g["decoded_sha256"], label
    short_first = [np.arange(500, dtype=np.int32), np.arange(900, dtype=np.int32)]
    with pytest.raises(capi.SelaHipError) as err:
        codec.encode_ragged(short_first)
    assert err.value.code == -2  # SELA_HIP_EINVAL
    # equal lengths through the same entry = sela_hip_encode_i32
    x = gc.case_input(1000, "stereo_diff", False)
    assert codec.encode_ragged([x[0], x[1]]) == codec.encode_i32(x[None])[0].tobytes()
