#!/usr/bin/env python3
"""Times the any-length route piece by piece (run on the GPU box; under rocprofv3 --kernel-trace --stats it gives the
k_generic_* kernels' durations for profiles/rNN/generic_kernels.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from sela_amd import codec  # noqa: E402
from sela_amd.synth import synth_pcm  # noqa: E402


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


if "counters" in sys.argv[1:]:
    # one encode and one decode of 3875 stereo frames of 2048 samples (11,625 blocks / 7750 subframes: the bench's launch size)
    # for tools/generic_counters.sh; the decode twice: the product (one-piece parse) and every subframe by segments
    from sela_amd import capi

    pcm = synth_pcm(2048 * 3875, 2, 23).reshape(3875, 2048, 2)
    planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
    frames, offs = codec.encode_i32(planar)
    for mode in (1, 2):
        capi.lib().sela_hip_debug_standard_first(mode)
        dec = codec.decode_i32(frames, offs, 2)
    capi.lib().sela_hip_debug_standard_first(-1)
    assert all(np.array_equal(dec[f][c], planar[f, c]) for f in (0, 17, 3874) for c in (0, 1))
    sys.exit(0)

for n, nf in ((1000, 256), (2048, 32), (4096, 64), (65535, 4)):
    pcm = synth_pcm(n * nf, 2, 21).reshape(nf, n, 2)
    planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
    e_ms, (frames, offs) = timed(lambda: codec.encode_i32(planar))
    d32_ms, dec = timed(lambda: codec.decode_i32(frames, offs, 2))
    d16_ms, back = timed(lambda: codec.decode_host(frames, offs, 2))
    assert np.array_equal(np.asarray(back).reshape(-1, 2), pcm.reshape(-1, 2))
    one_e, (f1, o1) = timed(lambda: codec.encode_i32(planar[:1]))
    one_d, _ = timed(lambda: codec.decode_i32(f1, o1, 2))
    print(f"n {n:6d} x {nf:4d} stereo frames: encode_i32 {e_ms:8.3f} ms  decode_i32 {d32_ms:8.3f} ms  sela_hip_decode (fast kernels tried first) {d16_ms:8.3f} ms"
          f"   ONE frame: encode {one_e:6.3f} ms decode {one_d:6.3f} ms   = {n * nf / e_ms / 1e3:7.1f} / {n * nf / d32_ms / 1e3:7.1f} M samples/s")

# Large batches (argument "big"): what the route sustains when the device is full -- host pointers in and out (pageable), so
# the call times include the copies; the kernels' own durations come from running this under rocprofv3 --kernel-trace --stats.
if "big" in sys.argv[1:]:
    for n, nf in ((1000, 8000), (2048, 3875), (4096, 2000)):
        pcm = synth_pcm(n * nf, 2, 23).reshape(nf, n, 2)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        e_ms, (frames, offs) = timed(lambda: codec.encode_i32(planar), reps=2)
        d32_ms, dec = timed(lambda: codec.decode_i32(frames, offs, 2), reps=2)
        print(f"big: n {n:6d} x {nf:5d} stereo frames: encode_i32 {e_ms:8.3f} ms  decode_i32 {d32_ms:8.3f} ms   = {n * nf / e_ms / 1e3:7.1f} / "
              f"{n * nf / d32_ms / 1e3:7.1f} M samples/s (copies from and to pageable host memory included)")
