"""CPU-only checks of host-side logic and of the mathematical claims the kernels rely on."""
import os

import numpy as np

from oracle_lib import oracle
from sela_amd.synth import synth_frames, synth_pcm


def _first_min_by_scan(u, n):
    """src/rice/rice_encoder.cpp:20-33: evaluate all 20 parameters, first minimum wins."""
    bits = [int((u >> k).sum()) + n * (1 + k) for k in range(20)]
    return bits.index(min(bits)), min(bits)


def _first_min_by_convexity(u, n):
    """What rice_plan() in sela_amd/csrc/sela_encode.hip does: walk from a guess to the smallest k whose
    one-step decrease d(k) = T(k) - T(k+1) is <= n."""
    T = lambda k: int((u >> k).sum())
    t0 = T(0)
    mean = t0 // n if n else 0
    k = min(mean.bit_length() - 1 if mean else 0, 18)
    ta, tb = T(k), T(k + 1)
    if ta - tb <= n:
        while k > 0:
            tc = T(k - 1)
            if tc - ta > n:
                break
            k, tb, ta = k - 1, ta, tc
    else:
        while True:
            k, ta = k + 1, tb
            if k == 19:
                break
            tb = T(k + 1)
            if ta - tb <= n:
                break
    return k, ta + n * (1 + k)


def test_rice_parameter_search_matches_full_scan():
    rng = np.random.default_rng(0)
    cases = [np.zeros(2048, np.uint64), np.full(100, 127, np.uint64), np.array([1 << 31] + [0] * 63, np.uint64)]
    for _ in range(300):
        scale = int(rng.integers(0, 31))
        n = int(rng.integers(1, 2049))
        kind = rng.integers(0, 3)
        if kind == 0:
            u = rng.integers(0, 1 << scale, n, dtype=np.uint64) if scale else np.zeros(n, np.uint64)
        elif kind == 1:  # heavy tail
            u = (rng.exponential(float(1 << min(scale, 24)), n)).astype(np.uint64)
        else:  # mostly small with outliers
            u = rng.integers(0, 8, n, dtype=np.uint64)
            u[rng.integers(0, n, max(1, n // 50))] = (1 << scale) - (1 if scale else 0)
        cases.append(np.minimum(u, np.uint64((1 << 32) - 1)))
    for u in cases:
        assert _first_min_by_convexity(u, len(u)) == _first_min_by_scan(u, len(u))


def test_rice_parameter_matches_oracle():
    o = oracle()
    rng = np.random.default_rng(1)
    for _ in range(50):
        v = (rng.normal(0, 10 ** rng.uniform(0, 4.5), int(rng.integers(1, 2049)))).astype(np.int32)
        k, words = o.rice_encode(v)
        u = np.where(v < 0, -2 * v.astype(np.int64) - 1, 2 * v.astype(np.int64)).astype(np.uint64)
        kk, bits = _first_min_by_convexity(u, len(u))
        assert kk == k and len(words) == -(-bits // 32)


def test_synthetic_generator_is_deterministic_and_integer_only():
    a = synth_frames(3, 2, 7)
    b = synth_pcm(3 * 2048, 2, 7).reshape(3, 2048, 2)
    assert a.dtype == np.int16 and np.array_equal(a, b)
    assert not np.array_equal(synth_frames(3, 2, 8), a)
    # mono and multichannel variants are prefixes of the same per-channel streams
    assert np.array_equal(synth_frames(3, 1, 7)[:, :, 0], a[:, :, 0])


def test_frame_stream_layout_constants():
    """include/sela_format.h agrees with what the reference wrote (golden frame)."""
    import os

    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "kats.npz"))
    b = k["frame/stereo_synth0/bytes"]
    total_words = 0
    p = 4
    for _ in range(2):
        cw = int(b[p + 4]) | int(b[p + 5]) << 8
        p += 7 + 4 * cw
        rw = int(b[p + 1]) | int(b[p + 2]) << 8
        p += 5 + 4 * rw
        total_words += cw + rw
    assert p == len(b) == 4 + 2 * 12 + 4 * total_words and len(b) % 4 == 0


def test_scale_division_is_exact(tmp_path):
    """sela_encode.hip:scale_sample replaces x = s / 32767 (src/lpc/residue_generator.cpp:12-18) by a
    multiply and two fmas; tests/c/scale_division.c proves bit-equality with the IEEE quotient for
    every |s| <= 70000 (16-bit samples and stereo differences stay within 65535)."""
    import subprocess

    exe = tmp_path / "scale_division"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "scale_division.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"])
    assert subprocess.check_output([str(exe)]).decode().strip() == "0"


def test_torch_synth_matches_numpy():
    """The torch generator (album-sized workloads, any device) is bit-identical to the numpy one, at any offset."""
    import numpy as np

    from sela_amd.synth import album_tracks, synth_pcm, synth_pcm_torch

    for track, ch in [(5, 2), (0, 1), (99, 3)]:
        a = synth_pcm(70000, ch, track)
        assert np.array_equal(a, synth_pcm_torch(70000, ch, track).numpy())
        assert np.array_equal(a[12345:20000], synth_pcm_torch(7655, ch, track, start=12345).numpy())
    tracks = album_tracks()
    assert len(tracks) == 100 and sum(f for _, _, f in tracks) == 549365
    assert [tracks[i][2] for i in range(3)] == [3875, 4218, 8437]
