"""Pin the CPU restatement to the REAL reference where it is available.

oracle/_ref/libsela_ref.so is the unmodified reference compiled by `make -C oracle ref`
(only possible where /root/reference exists; the prebuilt library travels to the GPU box).
Skipped when the library is absent -- tests/test_oracle_golden.py covers that case.
"""
import numpy as np
import pytest

from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames

ref = reference()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsela_ref.so not built")


def _blocks(seed, count):
    rng = np.random.default_rng(seed)
    i = np.arange(2048)
    for c in range(count):
        kind = c % 6
        if kind == 0:  # coloured noise, random level
            x = np.cumsum(rng.integers(-300, 301, 2048)) // (1 + c % 7)
        elif kind == 1:  # white noise at a random level (k hovering near the 0.05 threshold)
            x = rng.integers(-(1 << (3 + c % 13)), 1 << (3 + c % 13), 2048)
        elif kind == 2:  # sinusoid mix computed in integers
            x = (20000 * np.sin(i * (0.01 + 0.002 * c)) + rng.integers(-50, 51, 2048)).astype(np.int64)
        elif kind == 3:  # 17-bit difference-like signal
            x = rng.integers(-65535, 65536, 2048)
        elif kind == 4:  # sparse
            x = np.where(rng.random(2048) < 0.01, rng.integers(-32768, 32768, 2048), 0)
        else:  # constant + tiny dither
            x = 1000 * (c % 30) + rng.integers(0, 2, 2048)
        yield np.clip(x, -65535, 65535).astype(np.int32)


def test_lpc_and_rice_stages_match():
    o = oracle()
    for s in _blocks(11, 120):
        order, q, r = o.lpc_analyze(s)
        order_r, q_r, r_r = ref.lpc_analyze(s)
        assert order == order_r and np.array_equal(q, q_r) and np.array_equal(r, r_r)
        assert np.array_equal(o.lpc_coeffs(order, q), ref.lpc_coeffs(order, q))
        assert np.array_equal(o.lpc_synth(order, q, r), ref.lpc_synth(order, q, r))
        for v in (q, r):
            k, w = o.rice_encode(v)
            k_r, w_r = ref.rice_encode(v)
            assert k == k_r and np.array_equal(w, w_r)
            assert np.array_equal(o.rice_decode(w, len(v), k), ref.rice_decode(w, len(v), k))


def test_synth_decoder_inputs_not_from_encoder():
    """Decoder-side functions on coefficient sets the encoder would not emit."""
    o = oracle()
    rng = np.random.default_rng(5)
    for _ in range(60):
        order = int(rng.integers(1, 101))
        q = rng.integers(-64, 64, order).astype(np.int32)
        q[2:] = rng.integers(-12, 12, max(order - 2, 0))
        r = rng.integers(-2000, 2000, 2048).astype(np.int32)
        assert np.array_equal(o.lpc_coeffs(order, q), ref.lpc_coeffs(order, q))
        assert np.array_equal(o.lpc_synth(order, q, r), ref.lpc_synth(order, q, r))


@pytest.mark.parametrize("channels", [1, 2, 3])
def test_frames_match(channels):
    o = oracle()
    pcm = synth_frames(24, channels, 9 + channels)
    blob, offs, _ = o.encode_frames(pcm, threads=4)
    blob_r, offs_r, _ = ref.encode_frames(pcm, threads=4)
    assert np.array_equal(blob, blob_r) and np.array_equal(offs, offs_r)
    dec, _ = o.decode_frames(blob, offs, channels, threads=4)
    dec_r, _ = ref.decode_frames(blob_r, offs_r, channels, threads=4)
    assert np.array_equal(dec, dec_r) and np.array_equal(dec, pcm)
