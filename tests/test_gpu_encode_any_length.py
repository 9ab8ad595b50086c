"""GPU parity tests, by subject: the encoder for blocks of ANY length and 32-bit samples (k_generic_analyse / plan / pack /
assemble in sela_generic.hip behind sela_hip_encode_i32, sela_hip_encode with samples_per_channel != 2048 and
sela_hip_lpc_encode_n) against the oracle (oracle/sela_oracle.c, pinned against the unmodified reference):
src/lpc/residue_generator.cpp:12-134, src/lpc/linear_predictor.cpp:16-61, src/rice/rice_encoder.cpp:12-81,
src/frame/frame_encoder.cpp:11-102.  Bit-exact: frame bytes, orders, quantised coefficients, residues."""
import numpy as np
import pytest

from oracle_lib import oracle
from test_gpu_decode_any_length import _signal
from test_gpu_parity import gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

LENGTHS = [2, 63, 64, 65, 101, 127, 128, 129, 191, 255, 256, 257, 300, 1000, 2047, 2048, 2049, 2175, 2176, 2177, 4095, 4096, 4097, 6144, 20000, 65535]


def _wrap_taps(on):
    from sela_amd import capi

    capi.lib().sela_hip_debug_generic_wrap_taps(1 if on else 0)


@pytest.mark.parametrize("n", LENGTHS)
def test_frames_of_any_length_equal_the_oracles_bytes_in_both_forms_of_the_residue_filter(gpu, n):  # noqa: F811
    """Lengths on both sides of every boundary the kernel has (chunks of 64, the ring of 256, stretches of 2048 with 128 samples
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
