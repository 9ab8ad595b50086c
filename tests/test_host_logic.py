"""CPU-only checks of host-side logic and of the mathematical claims the kernels rely on."""
import os
import sys

import numpy as np
import pytest

from oracle_lib import oracle
from sela_amd.synth import synth_frames, synth_pcm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_residue_filter_in_fp64_is_exact_under_its_bounds():
    """What sela_encode_tail.inc relies on (round 4).  One pass: while 2^34 + sum |a[j]| x max |s| < 2^53 every partial sum of
    2^34 + sum a[j] s[i-j] is an integer below 2^53, so float64 arithmetic carries it exactly and floor(sum / 2^35) is the
    reference's (int64 sum) >> 35.  Two passes: with a = a_hi 2^20 + a_lo (0 <= a_lo < 2^20) and L = 2^34 + sum a_lo s,
    (sum a_hi s 2^20 + L) >> 35 == (sum a_hi s + (L >> 20)) >> 15 -- the dropped fraction of L cannot carry -- and with
    order x (max |a| / 2^20 + 1) x max |s| < 2^42 the whole sum stays below 2^63 (the reference's int64 does not wrap) and the
    prediction inside 32 bits.  Checked against Python integers on
    random predictors at the edges of both bounds, negative coefficients and samples included."""
    rng = np.random.default_rng(5)
    for trial in range(300):
        order = int(rng.integers(1, 101))
        s_mag = int(rng.choice([1, 300, 32768, 65536]))
        s = rng.integers(-s_mag, s_mag + 1, 64 + order).astype(np.int64)
        s[int(rng.integers(0, len(s)))] = s_mag * int(rng.choice([-1, 1]))
        two_pass = trial % 2 == 1
        if not two_pass: # one pass: sum |a| as large as the bound allows
            budget = ((1 << 53) - (1 << 34) - 1) // s_mag
            w = rng.random(order) + 1e-3
            a = np.minimum((w / w.sum() * budget).astype(np.float64), float((1 << 39) - 1)).astype(np.int64)
            a = a * rng.choice([-1, 1], order)
            assert int(np.abs(a).sum()) * s_mag + (1 << 34) < (1 << 53)
        else:            # two passes: max |a| as large as THAT bound allows (up to 2^55)
            top = min(((1 << 42) // (order * s_mag) - 1) << 20, (1 << 55) - 1)
            a = rng.integers(-top, top + 1, order).astype(np.int64)
            a[int(rng.integers(0, order))] = top * int(rng.choice([-1, 1]))
            a_top = int(np.bitwise_or.reduce(np.abs(a)))
            if not (a_top < (1 << 55) and order * ((a_top >> 20) + 1) * s_mag < (1 << 42)):
                continue # (the OR of the magnitudes may exceed the maximum: such a block takes the plain loop)
        for i in range(order, len(s)):
            taps = [(int(a[j - 1]), int(s[i - j])) for j in range(1, order + 1)]
            total = (1 << 34) + sum(c * x for c, x in taps)
            assert abs(total) < (1 << 63)                       # the reference's int64 does not wrap ...
            want = total >> 35
            assert -(1 << 31) <= want < (1 << 31)               # ... and its (int32) cast keeps the value
            if not two_pass:
                acc = float(1 << 34)
                for c, x in taps:
                    acc = float(c) * float(x) + acc
                    assert abs(acc) < 2.0 ** 53 and acc == int(acc)
                got = int(np.floor(acc * 2.0 ** -35))
            else:
                acc = float(1 << 34)
                for c, x in taps:
                    acc = float(c & 0xFFFFF) * float(x) + acc
                    assert abs(acc) < 2.0 ** 53
                acc = float(np.floor(acc * 2.0 ** -20))
                for c, x in taps:
                    acc = float(c >> 20) * float(x) + acc
                    assert abs(acc) < 2.0 ** 53
                got = int(np.floor(acc * 2.0 ** -15))
            assert got == want, (trial, i)


def test_launch_size_model_of_the_encode_kernels():
    """team_lanes_for (sela_encode.hip): which of the three encode kernels launch_encode takes for a batch -- a host-side
    model of their times (the team kernels' times are a staircase in waves per SIMD of the last round); no GPU needed to ask.
    k_encode_blocks for small launches and just behind a full round of team waves, teams of 16 around one fill, teams of 8
    where their rounds are full and for large launches; monotone in nothing, so the landmarks are spelled out."""
    from sela_amd import capi

    lib = capi.lib()
    lib.sela_hip_debug_encode_teams(-1)
    picks = {n: lib.sela_hip_debug_encode_kernel(n, 2) for n in (1, 1000, 3000, 3875, 4096, 4200, 5000, 8192, 9000, 16384, 61041)}
    assert picks == {1: 0, 1000: 0, 3000: 0, 3875: 16, 4096: 16, 4200: 0, 5000: 16, 8192: 8, 9000: 16, 16384: 8, 61041: 8}, picks
    for n in range(1, 70000, 997):
        assert lib.sela_hip_debug_encode_kernel(n, 2) in (0, 8, 16)
        assert lib.sela_hip_debug_encode_kernel(n, 1) in (0, 8, 16)


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


def _model_case(o, q, r, stats=None, coef_lanes=4):
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parse_model import parse, subframe_words

    q, r = np.asarray(q, np.int32), np.asarray(r, np.int32)
    ck, cwords = o.rice_encode(q) if len(q) else (0, np.zeros(0, np.uint32))
    rk, rwords = o.rice_encode(r)
    words, cw, rw = subframe_words(cwords, rwords)
    got_q, got_r, over_c, over_r = parse(words, cw, rw, ck, rk, len(q), len(r), coef_lanes, stats)
    assert np.array_equal(got_q, q) and np.array_equal(got_r, r) and not over_c and not over_r
    return cw, rw


def test_segment_parallel_parse_model():
    """tools/parse_model.py -- the algorithm of the decoder's segment-parallel Rice parser (zones, start
    bitmap, merge points, chain walk, counts) -- against the oracle's encoder on real subframes and on the
    streams that stress it: unary-only coding, runs longer than a zone, streams shorter than the wave,
    empty coefficient streams, all-zero and single-value inputs."""
    import numpy as np

    from oracle_lib import oracle
    from sela_amd.synth import synth_frames

    o = oracle()
    rng = np.random.default_rng(5)
    stats = []
    pcm = synth_frames(3, 2, 7)
    for f in range(3):
        for c in range(2):
            order, q, r = o.lpc_analyze(pcm[f, :, c].astype(np.int32))
            _model_case(o, q, r, stats)
    assert max(s[3] for s in stats) <= 64 + 4  # chain hops never exceed the lanes
    cases = [
        ([0], np.zeros(2048)),                                   # silence: k = 0, 64 words, zones of one word
        ([], rng.integers(-3, 4, 2048)),                         # order 0: no coefficient stream at all
        ([5, -3], rng.integers(-1, 2, 2048)),                    # unary-heavy
        (rng.integers(-64, 64, 100), rng.integers(-20000, 20000, 2048)),  # order 100, wide residues
        ([1, 2, 3], np.where(rng.random(2048) < 0.01, 30000, 0)),        # long unary runs among zeros
        ([-64] * 7, np.concatenate([np.full(5, 1 << 14), np.zeros(2043)])),
        ([3], rng.integers(-(1 << 16), 1 << 16, 2048)),          # 17-bit residues: near the LDS plan's cap
    ]
    for q, r in cases:
        for lanes in (1, 4, 8):
            _model_case(o, q, np.asarray(r, np.int64), None, lanes)
    # short value counts (the model is generic in n): zones longer than the stream
    for n in (1, 2, 63, 64, 65, 130):
        _model_case(o, [1], rng.integers(-40, 40, n))


def test_segmented_parse_model_on_streams_of_any_length():
    """tools/parse_model.py parse_segments -- the algorithm of k_decode_subframes32's parse by segments (sela_decode32.hip: an entry
    anywhere in a word, a limit behind which a codeword is the next segment's, a cap on the codewords listed, segments sized by
    the stream's own words per value) -- against the oracle's encoder and decoder: long noise (segments cut by their words),
    silence (cut by their codeword count), clicks whose unary runs span zones and segments, and tiny segments that put every
    boundary case on every few codewords."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from oracle_lib import oracle
    from parse_model import parse_segments

    o = oracle()
    rng = np.random.default_rng(8)

    def check(values, seg_words=1072, seg_values=2048, min_segments=1):
        v = np.asarray(values, np.int32)
        k, words = o.rice_encode(v)
        got, over, segs = parse_segments(words, 0, 32 * len(words), k, len(v), seg_words, seg_values)
        assert not over and np.array_equal(got, v), (len(v), k, seg_words, seg_values)
        assert segs >= min_segments, (segs, min_segments)
        # ... and from an entry in the middle of a word: the same stream behind 24 junk bits (how the coefficient stream lies)
        shifted = np.zeros(len(words) + 1, np.uint64)
        shifted[:-1] |= (words.astype(np.uint64) << np.uint64(24)) & np.uint64(0xFFFFFFFF)
        shifted[1:] |= words.astype(np.uint64) >> np.uint64(8)
        shifted[0] |= np.uint64(0x00ABCDEF)
        got2, over2, _ = parse_segments(shifted.astype(np.uint32), 24, 24 + 32 * len(words), k, len(v), seg_words, seg_values)
        assert not over2 and np.array_equal(got2, v)

    check(rng.integers(-20000, 20000, 9000), min_segments=3)               # ~17 bits per value: several segments' words
    check(np.zeros(10000), min_segments=4)                                   # k = 0, a bit per value: cut by the 2048-codeword cap
    check(np.where(rng.random(6000) < 0.004, rng.integers(-(1 << 17), 1 << 17, 6000), rng.integers(-2, 3, 6000)))  # long unary runs
    for seg_words, seg_values in ((3, 5), (1, 1), (2, 64), (7, 3), (64, 2048)):
        check(rng.integers(-300, 300, 700), seg_words, seg_values, min_segments=2)
        check(np.where(rng.random(500) < 0.05, 5000, 0), seg_words, seg_values, min_segments=2)  # runs longer than a whole segment
    # a stream that runs dry: what is missing reads as zeros and the overrun is reported
    v = rng.integers(-300, 300, 3000).astype(np.int32)
    k, words = o.rice_encode(v)
    got, over, _ = parse_segments(words[: len(words) // 2], 0, 32 * (len(words) // 2), k, 3000)
    n_ok = int((got == v).cumprod().sum())
    assert over and n_ok > 1300 and np.all(got[n_ok + 1:] == 0)


def test_parse_model_reports_truncated_streams():
    """A residue stream cut short decodes zeros behind its end and raises the overrun flag -- the behaviour
    of the kernel (reads beyond the stream are zero), which the reference leaves undefined."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from oracle_lib import oracle
    from parse_model import parse, subframe_words

    o = oracle()
    rng = np.random.default_rng(6)
    r = rng.integers(-300, 300, 2048).astype(np.int32)
    rk, rwords = o.rice_encode(r)
    ck, cwords = o.rice_encode(np.array([4, -2], np.int32))
    cut = len(rwords) // 2
    words, cw, rw = subframe_words(cwords, rwords[:cut])
    q, got, over_c, over_r = parse(words, cw, rw, ck, rk, 2, 2048)
    assert not over_c and over_r and q.tolist() == [4, -2]
    full = o.rice_decode(rwords, 2048, rk)
    n_ok = int((got == full).cumprod().sum())
    assert n_ok > 900 and np.all(got[n_ok + 1:] == 0)


def test_album_ranges_tile_the_album():
    """bench.py --gpus N (N > 1) gives every rank a contiguous range of the album's (track, frame) space and has
    it synthesise just that: the ranks' pieces, in rank order, must be the tracks' frames in job order -- for
    every world size the driver uses -- and rank_track_pieces must place them in the right files."""
    import numpy as np

    from sela_amd import sharding
    from sela_amd.synth import album_tracks, synth_frames_torch

    tracks = album_tracks(9, seconds=0.5)  # 10 / 11 / 23 frames per track at 44.1 / 48 / 96 kHz
    frames = [f for _, _, f in tracks]
    starts = np.concatenate([[0], np.cumsum(frames)])
    n_total = int(starts[-1])
    whole = np.concatenate([synth_frames_torch(f, 2, t).numpy() for t, _, f in tracks])
    for world in (1, 2, 4, 8):
        got = []
        for rank in range(world):
            b0, e0 = sharding.my_range(n_total, rank, world)
            for track, _, nf in tracks:  # (the loop of bench.py)
                b, e = max(b0, int(starts[track])), min(e0, int(starts[track + 1]))
                if b < e:
                    got.append(synth_frames_torch(e - b, 2, track, first_frame=b - int(starts[track])).numpy())
        assert np.array_equal(np.concatenate(got), whole), world
        sizes = np.arange(1, n_total + 1, dtype=np.uint64) * 4  # any sizes: the layout arithmetic is what is checked
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        layout = sharding.FileLayout(sizes, offsets, sharding.partition(n_total, world))
        covered = np.zeros(n_total, bool)
        for rank in range(world):
            for p in sharding.rank_track_pieces(layout, frames, rank):
                assert p.job_frame == int(starts[p.track]) + p.first_frame
                assert p.file_offset == sharding.SELA_HEADER_BYTES + int(offsets[p.job_frame] - offsets[int(starts[p.track])])
                assert not covered[p.job_frame: p.job_frame + p.n_frames].any()
                covered[p.job_frame: p.job_frame + p.n_frames] = True
        assert covered.all()


def test_bench_with_more_than_one_gpu_and_no_launcher_starts_its_own():
    """`python bench.py --gpus 2` without WORLD_SIZE used to die on an assertion before anything ran; now it starts
    torch.distributed.run itself.  Without GPUs the ranks fail -- loudly, with a non-zero exit code and no JSON line --
    but they ARE started (the failure is torch's, about the device, not bench.py's about WORLD_SIZE)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra-legs", "--no-host-legs",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the GPU tests run the real thing")
    assert r.returncode != 0
    assert "AssertionError: --gpus" not in r.stderr
    assert not [t for t in r.stdout.splitlines() if t.startswith("{")]


def test_reference_binding_is_built_and_fails_loudly_without_a_gpu(tmp_path):
    """oracle/_ref/sela_ref_bound (oracle/binding/bound.cpp: the reference's sela::Encoder / Decoder declarations with
    processFrames() bound to libsela_hip.so, built where /root/reference exists) links, and -- no GPU here -- reports the
    library's error through the reference's data::Exception instead of falling back to anything."""
    import os
    import shutil
    import subprocess

    import torch

    from sela_amd.synth import synth_pcm
    from test_host_cpp import _write_wav

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "sela_ref_bound")
    if not os.path.exists(exe):
        if not os.path.isdir("/root/reference/src") or shutil.which("g++") is None:
            pytest.skip("no reference checkout here: the binding is built in the build container")
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "bound"])
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: tests/test_gpu_host_and_files.py runs the binding for real")
    wav = tmp_path / "in.wav"
    _write_wav(wav, synth_pcm(5000, 2, 1), 44100)
    r = subprocess.run([exe, "-e", str(wav), str(tmp_path / "out.sela")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)
