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
