"""The CPU restatement (oracle/sela_oracle.c) against the committed golden fixtures.

The fixtures were produced by the unmodified reference (tests/golden/make_golden.py), so these
tests pin the oracle to the reference on every box, including ones without /root/reference.
Mirrors the reference's own round-trip tests (test/lpctests.cpp, test/ricetests.cpp,
test/frametests.cpp) and adds the known-answer bitstreams the reference never had.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import synth_frames


def test_block_kats(kats):
    o = oracle()
    for name in kats["blk_names"]:
        s = kats[f"blk/{name}/samples"]
        order, q, r, a, tr, flags = o.lpc_analyze(s, with_trace=True)
        assert order == int(kats[f"blk/{name}/order"]), name
        assert np.array_equal(q, kats[f"blk/{name}/q"]), name
        assert np.array_equal(a, kats[f"blk/{name}/a"]), name
        assert np.array_equal(r, kats[f"blk/{name}/residues"]), name
        ck, cw = o.rice_encode(q)
        rk, rw = o.rice_encode(r)
        assert ck == int(kats[f"blk/{name}/coef_k"]) and np.array_equal(cw, kats[f"blk/{name}/coef_words"]), name
        assert rk == int(kats[f"blk/{name}/res_k"]) and np.array_equal(rw, kats[f"blk/{name}/res_words"]), name
        assert np.array_equal(o.lpc_synth(order, q, r), kats[f"blk/{name}/synth"]), name


def test_survey_known_answers(kats):
    """SURVEY.md App. C values, captured from the reference during the survey."""
    o = oracle()
    order, q, r, a, _, _ = o.lpc_analyze(kats["blk/sine_deg/samples"], with_trace=True)
    assert order == 17
    assert q.tolist() == [-62, 14, 12, 10, 9, 7, 7, 6, 5, 5, 4, 4, 4, 3, 3, 3, 3]
    assert a[:6].tolist() == [0, 37244248064, -830065604, -114864354, 199925996, -590349670]
    assert r[:8].tolist() == [0, 571, 524, 489, 457, 420, 394, 360]
    k, w = o.rice_encode(q)
    assert k == 4 and [hex(x) for x in w] == ["0x4914dd7f", "0x4a519ce4", "0x6318c108"]
    k, w = o.rice_encode(r)
    assert k == 7 and len(w) == 552 and [hex(x) for x in w[:3]] == ["0xff6eff00", "0x3fa57f18", "0xa3f12fd2"]
    k, w = o.rice_encode(np.array([0, -1, 1, -2, 2, 100, -100, 5], np.int32))
    assert k == 5 and [hex(x) for x in w] == ["0xc8c10800", "0x538fc4f"]


def test_lpc_round_trip_like_reference_test(kats):
    """test/lpctests.cpp:10-32."""
    o = oracle()
    s = kats["blk/sine_deg/samples"]
    order, q, r = o.lpc_analyze(s)
    assert np.array_equal(o.lpc_synth(order, q, r), s)


def test_rice_kats(kats):
    o = oracle()
    for name in kats["rice_names"]:
        v = kats[f"rice/{name}/values"]
        k, w = o.rice_encode(v)
        assert k == int(kats[f"rice/{name}/k"]), name
        assert np.array_equal(w, kats[f"rice/{name}/words"]), name
        assert np.array_equal(o.rice_decode(w, len(v), k), v), name  # test/ricetests.cpp:7-25


def test_frame_kats(kats):
    o = oracle()
    for name in kats["frame_names"]:
        pcm = kats[f"frame/{name}/pcm"]
        blob = o.frame_encode(pcm)
        assert blob == kats[f"frame/{name}/bytes"].tobytes(), name
        dec, used = o.frame_decode(blob, pcm.shape[1])
        assert used == len(blob)
        assert np.array_equal(dec, kats[f"frame/{name}/decoded"]), name
        assert np.array_equal(dec, pcm), name  # test/frametests.cpp:8-70 (lossless)


def test_frame_kat_structure(kats):
    """same sine on both channels -> second channel is a silent difference."""
    b = kats["frame/stereo_same_sine/bytes"]
    assert b[:4].tolist() == [0x00, 0xFF, 0x55, 0xAA]
    assert b[4:8].tolist() == [0, 0, 0, 4] and int(b[8]) | int(b[9]) << 8 == 3 and b[10] == 17
    p = 4 + 7 + 12
    assert b[p] == 7 and int(b[p + 1]) | int(b[p + 2]) << 8 == 552 and int(b[p + 3]) | int(b[p + 4]) << 8 == 2048
    p += 5 + 4 * 552
    assert b[p : p + 3].tolist() == [1, 1, 0] and b[p + 3] == 0 and b[p + 6] == 1


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "config2_1000_frames"])
def test_config_digests(digests, label):
    """BASELINE.json configs 0-2: whole-job bitstream equality via SHA-256 of the frames blob."""
    d = digests[label]
    o = oracle()
    pcm = synth_frames(d["n_frames"], d["channels"], d["track"])
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == d["pcm_sha256"], "synthetic generator drifted"
    threads = os.cpu_count() or 1
    blob, offs, _ = o.encode_frames(pcm, threads=threads)
    assert len(blob) == d["frames_blob_bytes"]
    assert hashlib.sha256(blob.tobytes()).hexdigest() == d["frames_blob_sha256"]
    assert hashlib.sha256(offs.astype("<u8").tobytes()).hexdigest() == d["offsets_sha256"]
    dec, _ = o.decode_frames(blob, offs, d["channels"], threads=threads)
    assert hashlib.sha256(dec.tobytes()).hexdigest() == d["decoded_sha256"]
    assert np.array_equal(dec, pcm)


def test_thread_count_does_not_change_output():
    """The reference's output is independent of the thread count (ordered concatenation,
    src/sela/encoder.cpp:75-84), including more threads than frames."""
    o = oracle()
    pcm = synth_frames(13, 2, 2)
    base = o.encode_frames(pcm, threads=1)
    for t in (2, 5, 13, 32):
        blob, offs, _ = o.encode_frames(pcm, threads=t)
        assert np.array_equal(blob, base[0]) and np.array_equal(offs, base[1])
