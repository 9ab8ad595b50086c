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

import struct

from oracle_lib import oracle
from sela_amd.synth import synth_frames, synth_frames_torch, synth_pcm


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


def wav_header(rate, channels, data_bytes):
    """The canonical 44-byte header the reference writes (src/file/wav_file.cpp:222-242)."""
    return (b"RIFF" + struct.pack("<I", 36 + data_bytes) + b"WAVE" + b"fmt " + struct.pack("<IhHIIHH", 16, 1, channels, rate,
            rate * channels * 2, channels * 2, 16) + b"data" + struct.pack("<I", data_bytes))


def sela_header(rate, channels, n_frames):
    """15 bytes, src/file/sela_file.cpp:108-114."""
    return b"SeLa" + struct.pack("<IHBI", rate, 16, channels, n_frames)


@pytest.mark.parametrize("label", ["config0_mono_10s", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_file_digests(file_digests, label):
    """FILE level: header + frame stream of the restatement == the .sela file the reference's own
    sela::Encoder + SelaFile::writeToFile wrote, and canonical header + decoded PCM == the .wav its
    sela::Decoder + WavFile::writeToFile wrote (tail dropped: src/file/wav_file.cpp:184,203)."""
    d = file_digests[label]
    o = oracle()
    n, ch, rate = d["samples_per_channel"], d["channels"], d["sample_rate"]
    pcm = synth_pcm(n, ch, d["track"])
    assert hashlib.sha256(wav_header(rate, ch, pcm.nbytes) + pcm.tobytes()).hexdigest() == d["wav_sha256"], "input drifted"
    frames = n // 2048
    blob, offs, _ = o.encode_frames(pcm[: frames * 2048].reshape(frames, 2048, ch), threads=4)
    sela = sela_header(rate, ch, frames) + blob.tobytes()
    assert len(sela) == d["sela_bytes"] and hashlib.sha256(sela).hexdigest() == d["sela_sha256"]
    dec, _ = o.decode_frames(blob, offs, ch, threads=4)
    wav = wav_header(rate, ch, dec.nbytes) + dec.tobytes()
    assert len(wav) == d["decoded_wav_bytes"] and hashlib.sha256(wav).hexdigest() == d["decoded_wav_sha256"]


def test_album_digest_sample(album_digests):
    """BASELINE.json configs[3] on the CPU: three of the 100 tracks (one per sample rate), first 96 frames
    bit-exact through the restatement; the whole album is checked on the GPU (tests/test_album.py)."""
    assert album_digests["n_tracks"] == 100 and album_digests["n_frames"] == 549365
    o = oracle()
    for t in album_digests["tracks"][:3]:
        pcm = synth_frames_torch(96, 2, t["track"]).numpy()
        blob, offs, _ = o.encode_frames(pcm, threads=4)
        dec, _ = o.decode_frames(blob, offs, 2, threads=4)
        assert np.array_equal(dec, pcm) or t["lossy_frames"] > 0
    total = hashlib.sha256()
    for t in album_digests["tracks"]:
        total.update(bytes.fromhex(t["sela_sha256"]))
    assert total.hexdigest() == album_digests["sha256_of_track_sela_sha256s"]
