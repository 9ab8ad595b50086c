"""The CPU restatement on the domain the reference's frame CLASSES accept beyond its CLI: any block length, 32-bit samples
(data::WavFrame, src/include/data/wav_frame.hpp:8-16; src/lpc/residue_generator.cpp:98-119; src/frame/frame_decoder.cpp:11-72),
against fixtures the unmodified reference wrote (tests/golden/generic.json, generic_kats.npz) -- and, where
oracle/_ref/libsela_ref.so exists, against the reference itself on more shapes."""
import hashlib

import numpy as np
import pytest

import generic_cases as gc
from oracle_lib import oracle, reference


def test_generic_frames_against_the_reference_fixtures(generic_digests, generic_kats):
    o = oracle()
    for label, n, kind, wide in gc.all_cases():
        g = generic_digests[label]
        x = gc.case_input(n, kind, wide)
        assert gc.sha(x) == g["input_sha256"], label
        blob = o.frame_encode_i32(x)
        assert len(blob) == g["frame_bytes"] and hashlib.sha256(blob).hexdigest() == g["frame_sha256"], label
        if f"{label}/bytes" in generic_kats:
            assert blob == generic_kats[f"{label}/bytes"].tobytes(), label
        dec, used = o.frame_decode_i32(blob, x.shape[0])
        assert used == len(blob) and gc.sha_channels(dec) == g["decoded_sha256"], label
        assert all(np.array_equal(a, b) for a, b in zip(dec, x)) == g["lossless"], label
        if not wide:  # the int16 entry is the same frame
            assert o.frame_encode(np.ascontiguousarray(x.T.astype(np.int16))) == blob, label


def test_crafted_frames_decode_like_the_reference(generic_kats):
    o = oracle()
    for name in generic_kats["crafted_names"]:
        blob = generic_kats[f"crafted/{name}/bytes"].tobytes()
        ch = int(generic_kats[f"crafted/{name}/channels"])
        dec, used = o.frame_decode_i32(blob, ch)
        assert used == len(blob), name
        for c in range(ch):
            assert np.array_equal(dec[c], generic_kats[f"crafted/{name}/decoded{c}"]), (name, c)


def test_frames_whose_channels_differ_in_length_against_the_reference_fixtures(ragged_digests, ragged_kats):
    """src/frame/frame_encoder.cpp:20-24,73-98: every channel at its own length, the stereo difference over channel 1's."""
    o = oracle()
    for label, chans in gc.ragged_cases():
        g = ragged_digests[label]
        assert gc.sha_channels(chans) == g["input_sha256"] and [len(c) for c in chans] == g["lengths"], label
        blob = o.frame_encode_ragged(chans)
        assert len(blob) == g["frame_bytes"] and hashlib.sha256(blob).hexdigest() == g["frame_sha256"], label
        if f"{label}/bytes" in ragged_kats:
            assert blob == ragged_kats[f"{label}/bytes"].tobytes(), label
        assert [s[1] for s in gc.subframes_of(blob, len(chans))] == g["subframe_types"], label
        dec, used = o.frame_decode_i32(blob, len(chans))
        assert used == len(blob) and gc.sha_channels(dec) == g["decoded_sha256"], label


def test_oracle_against_the_reference_on_ragged_frames():
    ref = reference()
    if ref is None:
        pytest.skip("oracle/_ref/libsela_ref.so not built (needs /root/reference)")
    o = oracle()
    rng = np.random.default_rng(12)
    for trial in range(40):
        ch = int(rng.integers(2, 6))
        lengths = [int(rng.integers(101, 3000)) for _ in range(ch)]
        if ch == 2:
            lengths.sort(reverse=True)  # (channel 0 the longer one: the reference reads it up to channel 1's length)
        amp = int(rng.choice([300, 32767, 65535]))
        chans = []
        for c, n in enumerate(lengths):
            t = np.arange(n)
            chans.append(np.clip(np.round(amp * 0.5 * np.sin(t * 0.03 * (c + 1)) + rng.normal(0, amp * 0.02 + 1, n)), -amp, amp).astype(np.int32))
        if ch == 2 and trial % 2:
            chans[1] = (chans[0][: lengths[1]] - rng.integers(-2, 3, lengths[1])).astype(np.int32)
        assert o.frame_encode_ragged(chans) == ref.frame_encode_ragged(chans), (trial, lengths, amp)


def test_oracle_against_the_reference_on_odd_shapes():
    ref = reference()
    if ref is None:
        pytest.skip("oracle/_ref/libsela_ref.so not built (needs /root/reference)")
    o = oracle()
    rng = np.random.default_rng(3)
    for n in (101, 102, 127, 333, 2048, 5000, 9999):
        for ch in (1, 2, 3, 5):
            for amp in (100, 32767, 65535, 1 << 20):
                t = np.arange(n)
                x = np.stack([np.round(amp * 0.5 * np.sin(t * 0.02 * (c + 1)) + rng.normal(0, amp * 0.01 + 1, n)) for c in range(ch)]).astype(np.int64)
                x = np.clip(x, -amp, amp).astype(np.int32)
                if ch == 2:
                    x[1] = x[0] - rng.integers(-2, 3, n)
                a, b = o.frame_encode_i32(x), ref.frame_encode_i32(x)
                assert a == b, (n, ch, amp)
                da, ua = o.frame_decode_i32(a, ch)
                db, ub = ref.frame_decode_i32(a, ch)
                assert ua == ub == len(a)
                for u, v in zip(da, db):
                    assert np.array_equal(u, v), (n, ch, amp)


def test_a_rice_stream_beyond_2_24_bits_is_cut_like_the_reference():
    """ceil((float)bits / 32) is one word short for this stream (src/rice/rice_encoder.cpp:37,63): the oracle returns the
    reference's words, where oracle/_ref exists compared with the real thing."""
    v, k, bits = gc.long_rice_stream()
    ko, wo = oracle().rice_encode(v)
    assert ko == k and len(wo) == (bits + 31) // 32 - 1
    ref = reference()
    if ref is not None:
        kr, wr = ref.rice_encode(v)
        assert kr == ko and np.array_equal(wr, wo)
