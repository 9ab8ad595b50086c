"""A wide differential corpus of stereo frames (test data only; numpy, seeded): material chosen to walk the analysis and both
coders through their corners -- AR noise of orders 2..32 at many levels, reflection coefficients hovering at the 0.05 order
threshold, clipped and faded tones, DC steps, silence <-> full scale inside a block, 17-bit difference extremes, loud
polyphony (predictors beyond the one-pass bound of the residue filter) and smooth low-frequency polyphony (predictor
coefficients in the 2^45 .. 2^55 range: the plain 64-bit loop)."""
import numpy as np
from scipy.signal import lfilter

N = 2048


def _clip16(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def _ar_frames(rng, n_frames):
    """AR(p) noise, p = 2..32: random stable poles, level from -60 dB to clipping; the two channels share part of the
    excitation so that the difference signal sometimes wins."""
    out = np.zeros((n_frames, N, 2), np.int16)
    group = 32
    for g0 in range(0, n_frames, group):
        g1 = min(n_frames, g0 + group)
        p = int(rng.integers(2, 33))
        radii = rng.uniform(0.3, 0.995, p // 2)
        angles = rng.uniform(0.02, 3.1, p // 2)
        poles = np.concatenate([radii * np.exp(1j * angles), radii * np.exp(-1j * angles)])
        if p % 2:
            poles = np.append(poles, rng.uniform(-0.95, 0.95))
        a = np.real(np.poly(poles))
        n = (g1 - g0) * N
        common = rng.normal(0, 1, n)
        share = rng.choice([0.0, 0.5, 0.9, 1.0])
        for ch in range(2):
            e = share * common + (1 - share) * rng.normal(0, 1, n)
            x = lfilter([1.0], a, e)
            x = x / (np.abs(x).max() + 1e-9)
            level = 32767 * 10 ** (-rng.uniform(0, 60, g1 - g0) / 20) * rng.choice([1.0, 1.0, 4.0], g1 - g0)  # (x4: clipped)
            out[g0:g1, :, ch] = _clip16(x.reshape(g1 - g0, N) * level[:, None])
    return out


def _threshold_frames(rng, n_frames):
    """AR(1..4) with coefficients around 0.05: the order picker's strict |k| > 0.05 decides at the last place."""
    out = np.zeros((n_frames, N, 2), np.int16)
    for f in range(n_frames):
        p = int(rng.integers(1, 5))
        a = np.concatenate([[1.0], rng.normal(0, 0.05, p)])
        for ch in range(2):
            x = lfilter([1.0], a, rng.normal(0, 1, N))
            out[f, :, ch] = _clip16(x * rng.choice([30, 300, 3000, 12000]))
    return out


def _tones(rng, n_frames, lo, hi, kmin, kmax, noise):
    t = np.arange(N)
    out = np.zeros((n_frames, N, 2), np.int16)
    for f in range(n_frames):
        for ch in range(2):
            k = int(rng.integers(kmin, kmax + 1))
            fr, ph = rng.uniform(lo, hi, k), rng.uniform(0, 6.28, k)
            x = (np.sin(2 * np.pi * fr[:, None] * t[None, :] / 44100 + ph[:, None]) * (30000.0 / k)).sum(axis=0)
            x = x * rng.choice([1.0, 1.0, 0.25, 2.5]) + rng.normal(0, rng.choice(noise), N)
            out[f, :, ch] = _clip16(x)
    return out


def _shaped_frames(rng, n_frames):
    """Fades, DC steps, silence <-> full scale inside a block, impulses, alternating extremes, 17-bit differences."""
    t = np.arange(N)
    out = np.zeros((n_frames, N, 2), np.int16)
    for f in range(n_frames):
        kind = f % 8
        cut = int(rng.integers(1, N - 1))
        for ch in range(2):
            tone = 25000 * np.sin(2 * np.pi * rng.uniform(40, 9000) * t / 44100 + rng.uniform(0, 6.28)) + rng.normal(0, 20, N)
            if kind == 0:  # fade in / out
                x = tone * (t / N if ch == 0 else 1 - t / N) ** rng.choice([1, 2, 6])
            elif kind == 1:  # DC step
                x = np.where(t < cut, rng.integers(-30000, 30000), rng.integers(-30000, 30000)) + rng.normal(0, rng.choice([0, 1, 40]), N)
            elif kind == 2:  # silence -> full scale
                x = np.where(t < cut, 0, tone * 1.31)
            elif kind == 3:  # full scale -> silence (exact zeros)
                x = np.where(t < cut, tone * 1.31, 0)
            elif kind == 4:  # sparse impulses
                x = np.zeros(N)
                x[rng.integers(0, N, int(rng.integers(1, 6)))] = rng.choice([32767, -32768, 1, -1, 12345])
            elif kind == 5:  # alternating extremes, opposite in the two channels: the difference signal is 17-bit
                period = int(rng.integers(1, 9))
                x = np.where((t // period) % 2 == 0, 32767, -32768) * (1 if ch == 0 else -1)
            elif kind == 6:  # constant / near-constant
                x = np.full(N, rng.choice([0, 1, -1, 1234, -32768, 32767])) + (rng.integers(-1, 2, N) if rng.random() < 0.5 else 0)
            else:  # hard-clipped loud noise
                x = rng.normal(0, 60000, N)
            out[f, :, ch] = _clip16(x)
    return out


def build(n_frames=33400, seed=20260927):
    """-> int16 [n_frames, 2048, 2] (n_frames stereo frames = 3 n_frames analysed blocks)."""
    rng = np.random.default_rng(seed)
    parts = [
        _ar_frames(rng, int(n_frames * 0.62)),
        _threshold_frames(rng, int(n_frames * 0.12)),
        _shaped_frames(rng, int(n_frames * 0.14)),
        _tones(rng, int(n_frames * 0.05), 50, 20000, 3, 40, [0.3, 1, 3]),       # loud polyphony: beyond the one-pass bound
        _tones(rng, int(n_frames * 0.03), 15, 400, 2, 12, [0.0, 0.01, 0.3]),    # smooth low-frequency polyphony: huge predictors
    ]
    have = sum(len(p) for p in parts)
    parts.append(_ar_frames(rng, n_frames - have))
    pcm = np.concatenate(parts)
    return np.ascontiguousarray(pcm[rng.permutation(len(pcm))])


def expected_form(a, order, s):
    """The residue filter's form for a block by the rule of sela_encode_tail.inc: 0 one pass, 1 two passes, 2 the plain loop."""
    mags = [abs(int(v)) for v in a[1:order + 1]]
    s_mag = int(np.abs(s.astype(np.int64)).max())
    if s_mag <= 65536:
        if all(m < (1 << 39) for m in mags) and sum(mags) * s_mag + (1 << 34) < (1 << 53):
            return 0
        a_top = 0
        for m in mags:
            a_top |= m
        if a_top < (1 << 55) and order * ((a_top >> 20) + 1) * s_mag < (1 << 42):
            return 1
    return 2
