"""Integer-only synthetic 16-bit PCM.

The generator uses no floating point and no libm, so the samples are bit-identical on
every host: two table-lookup sinusoids per channel (table = data/sin4096.npy, a committed
4096-entry int16 sine table) plus counter-hashed noise.  It mimics the sine+noise material
the reference was surveyed on (LPC order ~30-100, residue Rice k ~9-11, mixed independent /
difference-coded second channels).
"""
from __future__ import annotations

import os

import numpy as np

_TABLE = None
BLOCK = 2048  # samplesPerChannelPerFrame, reference src/include/file/wav_file.hpp:12


def _sin_table() -> np.ndarray:
    global _TABLE
    if _TABLE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sin4096.npy")
        _TABLE = np.load(path).astype(np.int64)
    return _TABLE


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = x.copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _noise(n: np.ndarray, seed: np.uint64, stream: int) -> np.ndarray:
    """Centred Irwin-Hall sum of four 10-bit uniforms from a counter hash (sigma ~ 590)."""
    ctr = (n.astype(np.uint64) << np.uint64(4)) | np.uint64(stream)
    with np.errstate(over="ignore"):
        h = _mix64(ctr + seed * np.uint64(0x9E3779B97F4A7C15))
    m = np.uint64(1023)
    return ((h & m).astype(np.int64) + ((h >> np.uint64(10)) & m).astype(np.int64)
            + ((h >> np.uint64(20)) & m).astype(np.int64) + ((h >> np.uint64(30)) & m).astype(np.int64) - 2046)


def synth_pcm(n_samples: int, channels: int = 2, track: int = 0, noise_shift: int = 0) -> np.ndarray:
    """Return int16 PCM of shape [n_samples, channels] (interleaved, WAV data-chunk order).

    track selects the seed (0x5E1A0000 + track); noise_shift attenuates the noise
    (0 -> sigma ~ 590 LSB ~ 0.018 FS).
    """
    tab = _sin_table()
    n = np.arange(n_samples, dtype=np.int64)
    out = np.empty((n_samples, channels), dtype=np.int16)
    seed = np.uint64(0x5E1A0000 + track)
    for c in range(channels):
        f1 = 37 + 2 * track % 17
        f2 = 211 + 82 * c + 3 * (track % 5)
        phi = 977 * c + 131 * track
        # slow integer frequency drift so that frames differ from each other
        p1 = (n * f1 + ((n >> 9) * (n >> 9) >> 7)) & 4095
        p2 = (n * f2 + phi + (n >> 6)) & 4095
        a1 = 19660 - 2000 * c
        a2 = 6553 + 1500 * c
        tone = (a1 * tab[p1] + a2 * tab[p2]) >> 15
        own = _noise(n, seed, c + 1)
        common = _noise(n, seed, 0)
        # the share of noise common to all channels steps through 0, 1/4 .. 4/4 every four
        # frames, so a stereo second channel is sometimes cheaper difference-coded and sometimes not
        w = (n >> 13) % 5
        noise = (common * w + own * (4 - w)) >> 2
        x = tone + (noise >> noise_shift)
        out[:, c] = np.clip(x, -32768, 32767).astype(np.int16)
    return out


# ---- the same generator on torch tensors (any device) --------------------------------------------------
# Album-sized workloads (BASELINE.json configs[3]: 1.1 G stereo samples) take minutes in numpy on one core;
# on torch the same integer arithmetic runs on all host cores or on the GPU.  int64 two's-complement
# wrap-around stands in for numpy's uint64 (logical shifts are masked arithmetic shifts); the result is
# bit-identical to synth_pcm (tests/test_host_logic.py::test_torch_synth_matches_numpy).
def _i64(c: int) -> int:
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(x, s: int):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix64_t(x):
    x = x ^ _lsr(x, 30)
    x = x * _i64(0xBF58476D1CE4E5B9)
    x = x ^ _lsr(x, 27)
    x = x * _i64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _noise_t(n, seed: int, stream: int):
    h = _mix64_t(((n << 4) | stream) + _i64((seed * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)))
    return (h & 1023) + (_lsr(h, 10) & 1023) + (_lsr(h, 20) & 1023) + (_lsr(h, 30) & 1023) - 2046


def synth_pcm_torch(n_samples: int, channels: int = 2, track: int = 0, start: int = 0, device="cpu"):
    """Samples [start, start + n_samples) of synth_pcm(..., channels, track) as an int16 torch tensor
    [n_samples, channels] on `device`."""
    import torch

    tab = torch.from_numpy(_sin_table()).to(device)
    n = torch.arange(start, start + n_samples, dtype=torch.int64, device=device)
    out = torch.empty((n_samples, channels), dtype=torch.int16, device=device)
    seed = 0x5E1A0000 + track
    common = _noise_t(n, seed, 0)
    w = (n >> 13) % 5
    for c in range(channels):
        f1 = 37 + 2 * track % 17
        f2 = 211 + 82 * c + 3 * (track % 5)
        phi = 977 * c + 131 * track
        p1 = (n * f1 + (((n >> 9) * (n >> 9)) >> 7)) & 4095
        p2 = (n * f2 + phi + (n >> 6)) & 4095
        tone = ((19660 - 2000 * c) * tab[p1] + (6553 + 1500 * c) * tab[p2]) >> 15
        noise = (common * w + _noise_t(n, seed, c + 1) * (4 - w)) >> 2
        out[:, c] = torch.clamp(tone + noise, -32768, 32767).to(torch.int16)
    return out


def synth_frames_torch(n_frames: int, channels: int = 2, track: int = 0, first_frame: int = 0, device="cpu"):
    """Frames [first_frame, first_frame + n_frames) of a track: int16 tensor [n_frames, BLOCK, channels]."""
    return synth_pcm_torch(n_frames * BLOCK, channels, track, first_frame * BLOCK, device).reshape(n_frames, BLOCK, channels)


# ---- BASELINE.json configs[3]: the 100-track album ---------------------------------
ALBUM_TRACKS = 100
ALBUM_RATES = (44100, 48000, 96000)
ALBUM_SECONDS = 180


def album_tracks(n_tracks: int = ALBUM_TRACKS, seconds: float = ALBUM_SECONDS):
    """[(track id, sample rate, whole frames)]: rates round-robin over 44.1 / 48 / 96 kHz, 16-bit stereo.
    The full album is 34 / 33 / 33 tracks of 3875 / 4218 / 8437 frames = 549,365 frames."""
    return [(t, ALBUM_RATES[t % 3], frames_for_seconds(seconds, ALBUM_RATES[t % 3])) for t in range(n_tracks)]


def synth_frames(n_frames: int, channels: int = 2, track: int = 0) -> np.ndarray:
    """int16 PCM of shape [n_frames, BLOCK, channels]."""
    return synth_pcm(n_frames * BLOCK, channels, track).reshape(n_frames, BLOCK, channels)


def frames_for_seconds(seconds: float, sample_rate: int = 44100) -> int:
    """Whole 2048-sample frames in a track; the tail is dropped (reference
    src/file/wav_file.cpp:184,203)."""
    return int(seconds * sample_rate) // BLOCK
