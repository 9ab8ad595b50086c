"""What the GPU test files share: the `gpu` fixture (skip without a device, library built), the encode / decode helpers over the C ABI's
device-pointer calls, hand-built frames and Rice streams, and the paths of the reference-side binaries.  Not a test file."""
import ctypes as C
import hashlib
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
from sela_amd.synth import synth_frames
import struct
import subprocess
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
import json
import sys
import generic_cases as gc
from sela_amd.synth import synth_frames, synth_pcm


from test_host_cpp import HOST, ROOT, _build, _write_wav  # noqa: F401


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from sela_amd import capi

    capi.lib()  # raises if the HIP library is missing: no fallback
    return torch


def _encode(gpu, pcm, with_trace=False):
    from sela_amd import codec

    enc = codec.Encoder(pcm.shape[0], pcm.shape[2], with_trace=with_trace)
    out = enc.encode(gpu.from_numpy(np.ascontiguousarray(pcm)).cuda())
    gpu.cuda.synchronize()
    frames, offsets = out.to_host()
    return (frames, offsets, enc, out)


def _decode(gpu, frames, offsets, channels):
    from sela_amd import codec

    n = len(offsets) - 1
    dec = codec.Decoder(n, channels)
    f = gpu.from_numpy(np.ascontiguousarray(frames)).cuda()
    o = gpu.from_numpy(np.ascontiguousarray(offsets).view(np.int64)).cuda()
    pcm = dec.decode(f, o, n)
    gpu.cuda.synchronize()
    dec.check()
    return pcm.cpu().numpy()


def _bits(x):
    """Bit patterns of doubles; NaNs are canonicalised (sign/payload of a NaN carries no meaning in
    the codec: every consumer is a comparison or isnan())."""
    v = np.atleast_1d(np.asarray(x, dtype=np.float64)).copy()
    v[np.isnan(v)] = np.nan
    return v.view(np.uint64)


def _kat_block_frames(kats):
    """The single-block KAT signals as mono frames (those that fit int16)."""
    names, frames = [], []
    for name in kats["blk_names"]:
        s = kats[f"blk/{name}/samples"]
        if s.min() >= -32768 and s.max() <= 32767:
            names.append(str(name))
            frames.append(s.astype(np.int16)[:, None])
    return names, np.stack(frames)


def _rice_words(values, k):
    """Rice-code `values` with a GIVEN parameter k (src/rice/rice_encoder.cpp:35-71 without the parameter
    search): zig-zag, u >> k ones, a zero, k remainder bits MSB first; stream bit t = bit t % 32 of word t / 32."""
    bits = []
    for v in values:
        v = int(v)
        u = -2 * v - 1 if v < 0 else 2 * v
        bits += [1] * (u >> k) + [0] + [(u >> (k - 1 - i)) & 1 for i in range(k)]
    bits += [0] * (-len(bits) % 32)
    b = np.array(bits, np.uint64).reshape(-1, 32)
    return (b << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


def _build_frame(subframes, res_k=None):
    """Hand-assemble on-disk frame bytes from (channel, type, parent, q[], residues[]) tuples, Rice-coding
    with the oracle -- or, for the residues, with the parameter `res_k` no encoder would pick
    (layout of src/file/sela_file.cpp:115-135)."""
    import struct

    o = oracle()
    out = struct.pack("<I", 0xAA55FF00)
    for channel, typ, parent, q, res in subframes:
        ck, cw = o.rice_encode(np.asarray(q, np.int32))
        if res_k is None:
            rk, rw = o.rice_encode(np.asarray(res, np.int32))
        else:
            rk, rw = res_k, _rice_words(res, res_k)
        out += struct.pack("<BBBBHB", channel, typ, parent, ck, len(cw), len(q)) + cw.astype("<u4").tobytes()
        out += struct.pack("<BHH", rk, len(rw), len(res)) + rw.astype("<u4").tobytes()
    return out


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def _decode_frames_vs_oracle(gpu, frames, channels=1):
    o = oracle()
    stream = np.frombuffer(b"".join(frames), np.uint8).copy()
    offsets = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    got = _decode(gpu, stream, offsets, channels)
    for i, f in enumerate(frames):
        want, used = o.frame_decode(f, channels)
        assert used == len(f)
        assert np.array_equal(got[i], want), i


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


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


def _polyphonic_frames(n, seed):
    """Loud sums of 3..40 sinusoids over a little noise: long predictors with large coefficients at large amplitudes -- the
    blocks on which sum |a[j]| x max |s| passes 2^53, where FP64 multiply-adds of integers stop being exact."""
    rng = np.random.default_rng(seed)
    t = np.arange(2048)
    pcm = np.zeros((n, 2048, 2), np.int16)
    for f in range(n):
        for ch in range(2):
            k = int(rng.integers(3, 40))
            x = sum((30000 / k) * np.sin(2 * np.pi * fr * t / 44100 + ph) for fr, ph in zip(rng.uniform(50, 20000, k), rng.uniform(0, 6.28, k)))
            x = x * rng.choice([1.0, 1.0, 0.2]) + rng.normal(0, rng.choice([0.3, 1, 3]), 2048)
            pcm[f, :, ch] = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return pcm


BOUND = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "sela_ref_bound")


MAIN_ON_HOST = os.path.join(os.path.dirname(BOUND), "sela_ref_main_on_host")


def _one(offs):
    return np.array([0, offs], np.uint64)


def _fold(values):
    """hash_term of sela_encode.hip over an array of doubles, position-keyed, XOR-ed."""
    v = np.asarray(values, dtype=np.float64).copy()
    v[np.isnan(v)] = np.float64("nan")
    x = v.view(np.uint64).copy()
    x[np.isnan(v)] = np.uint64(0x7FF8000000000000)
    with np.errstate(over="ignore"):
        x ^= np.uint64(0x9E3779B97F4A7C15) * (np.arange(len(x), dtype=np.uint64) + np.uint64(1))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return int(np.bitwise_xor.reduce(x))


def _subframe_words(blob, ch):
    """aligned words (coefficient words + 2 + residue words) of every subframe of one frame"""
    import struct

    out, p = [], 4
    for _ in range(ch):
        cw = struct.unpack_from("<H", blob, p + 4)[0]
        rw = struct.unpack_from("<H", blob, p + 7 + 4 * cw + 1)[0]
        out.append(cw + 2 + rw)
        p += 12 + 4 * (cw + rw)
    return out


def _both_decoders(blob, offs, ch):
    """decode_i32 with the standard kernel offered and with the any-length kernel alone -> (offered, alone, chunks the standard
    kernel decoded)."""
    from sela_amd import capi, codec

    lib = capi.lib()
    try:
        lib.sela_hip_debug_standard_first(1)
        before = lib.sela_hip_debug_standard_chunks()
        offered = codec.decode_i32(blob, offs, ch)
        took = lib.sela_hip_debug_standard_chunks() - before
        lib.sela_hip_debug_standard_first(0)
        alone = codec.decode_i32(blob, offs, ch)
    finally:
        lib.sela_hip_debug_standard_first(-1)
    return offered, alone, took


ENCODE_LENGTHS = [2, 63, 64, 65, 101, 127, 128, 129, 191, 255, 256, 257, 300, 1000, 1023, 1024, 1025, 1151, 1152, 1153, 2047, 2048, 2049, 2175, 2176, 2177, 3071, 3072, 3073, 4095, 4096, 4097, 6144, 20000, 65535]


def _wrap_taps(on):
    from sela_amd import capi

    capi.lib().sela_hip_debug_generic_wrap_taps(1 if on else 0)


def _decode_modes(blob, offs, ch, modes=(1, 2, 0)):
    """sela_hip_decode_i32 as the product runs it (1), with every subframe parsed by segments (2) and on the serial kernel alone
    (0) -> ({mode: frames}, chunks the fast kernel took whole, subframes it parsed by segments) -- the counts over modes 1 and 2."""
    from sela_amd import capi, codec

    lib = capi.lib()
    out = {}
    chunks0, segs0 = lib.sela_hip_debug_standard_chunks(), lib.sela_hip_debug_segment_subframes()
    try:
        for m in modes:
            lib.sela_hip_debug_standard_first(m)
            out[m] = codec.decode_i32(blob, offs, ch)
    finally:
        lib.sela_hip_debug_standard_first(-1)
    return out, lib.sela_hip_debug_standard_chunks() - chunks0, lib.sela_hip_debug_segment_subframes() - segs0


def _signal(rng, kind, n, amp_bits):
    t = np.arange(n)
    amp = (1 << amp_bits) - 1
    if kind == "silence":
        return np.zeros(n, np.int32)
    if kind == "dc":
        return np.full(n, amp // 3, np.int32)
    if kind == "noise":  # incompressible: long Rice streams (beyond one segment's words from a few thousand samples)
        return rng.integers(-amp, amp + 1, n).astype(np.int32)
    if kind == "tone":
        return np.round(amp * 0.8 * np.sin(t * 0.05 + 1.0) + rng.normal(0, amp / 300 + 1, n)).astype(np.int32)
    if kind == "sparse":  # mostly zeros with rare full-scale clicks: k = 0 .. 2 with unary runs of thousands of bits
        x = np.zeros(n, np.int32)
        idx = rng.integers(0, n, max(1, n // 200))
        x[idx] = rng.integers(-amp, amp + 1, len(idx))
        return x
    raise AssertionError(kind)


DECODE_LENGTHS = [1, 2, 63, 64, 65, 127, 129, 300, 1000, 2047, 2048, 2049, 4096, 5000, 20000, 65535]


def _hostile_frame(rng, ch, n_lo, n_hi, big):
    subs = []
    roomy = rng.random() < 0.75
    n_frame = int(rng.integers(n_lo, n_hi))
    for c in range(ch):
        order = int(rng.integers(0, 101))
        ck = int(rng.integers(0, 12))
        rk = int(rng.integers(0, 20))
        n = n_frame if rng.random() < 0.8 else int(rng.integers(n_lo, n_hi))
        cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
        rwords = (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8)) if roomy else int(rng.integers(1, 1 + (n * (rk + 3)) // 32 + 8))
        rwords = min(rwords, 65535)
        style = rng.random()
        if style < 0.45:    # sparse words: short unary runs, most streams hold their values
            mk = lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)  # noqa: E731
        elif style < 0.8:
            mk = lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)  # noqa: E731
        else:               # stretches of all-ones words among sparse ones: unary runs across words, zones and segments
            def mk(m):
                w = (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)).astype(np.uint32)
                for _ in range(int(rng.integers(1, 4))):
                    if m > 4:
                        a = int(rng.integers(0, m - 2))
                        w[a: a + int(rng.integers(1, max(2, min(m - a, 80 if big else 6))))] = 0xFFFFFFFF
                return w
        cw, rw = mk(cwords), mk(rwords)
        if order and rng.random() < 0.7:  # coefficients the tables hold (a value outside [-64, 63] indexes past them in the reference: refused)
            cw = np.concatenate([_rice_words(rng.integers(-64, 64, order), ck), mk(int(rng.integers(0, 3)))])
            cwords = len(cw)
        typ = 1 if (c > 0 and rng.random() < 0.3) else 0
        parent = int(rng.integers(0, c)) if typ else c
        subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
    return bytes.fromhex("00ff55aa") + b"".join(subs)
