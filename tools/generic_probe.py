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

    nf = int(os.environ.get("SELA_PROBE_FRAMES", "3875"))
    pcm = synth_pcm(2048 * nf, 2, 23).reshape(nf, 2048, 2)
    planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
    frames, offs = codec.encode_i32(planar)
    for mode in (1, 2):
        capi.lib().sela_hip_debug_standard_first(mode)
        dec = codec.decode_i32(frames, offs, 2)
    capi.lib().sela_hip_debug_standard_first(-1)
    assert all(np.array_equal(dec[f][c], planar[f, c]) for f in (0, 17, nf - 1) for c in (0, 1))
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

# Large batches (argument "big"): what the route sustains when the device is full -- host pointers in and out, so the call times
# include the copies: from and to ordinary (pageable) memory that the caller has touched before (a first touch costs a page
# fault per 4 KB: the caller's, not the call's), and from and to page-locked memory (sela_hip_host_alloc).  The kernels' own
# durations come from running this under rocprofv3 --kernel-trace --stats.
if "big" in sys.argv[1:]:
    import ctypes as C

    from sela_amd import capi

    lib = capi.lib()

    def pinned(shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = lib.sela_hip_host_alloc(C.c_size_t(nbytes))
        assert ptr
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype).reshape(shape), ptr

    for n, nf in ((1000, 8000), (2048, 3875), (4096, 2000)):
        pcm = synth_pcm(n * nf, 2, 23).reshape(nf, n, 2)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        cap = int(lib.sela_hip_encode_bound_bytes_n(nf, 2, n))
        cap = min(cap, planar.nbytes * 2 + (1 << 20))  # (the format's bound is a quarter of a megabyte per subframe: room for the data at hand)
        rows = {}
        for kind in ("pageable", "page-locked"):
            if kind == "pageable":
                src, frames, offs = planar.copy(), np.ones(cap, np.uint8), np.zeros(nf + 1, np.uint64)
                out, counts, keep = np.ones((nf, 2, n), np.int32), np.zeros((nf, 2), np.uint32), []
            else:
                (src, p0), (frames, p1), (out, p2) = pinned(planar.shape, np.int32), pinned((cap,), np.uint8), pinned((nf, 2, n), np.int32)
                src[...] = planar
                offs, counts, keep = np.zeros(nf + 1, np.uint64), np.zeros((nf, 2), np.uint32), [p0, p1, p2]
            e_ms, _ = timed(lambda: capi.check(lib.sela_hip_encode_i32(src.ctypes.data, nf, 2, n, frames.ctypes.data, cap, offs.ctypes.data)), reps=3)
            d_ms, _ = timed(lambda: capi.check(lib.sela_hip_decode_i32(frames.ctypes.data, offs.ctypes.data, nf, 2, out.ctypes.data, n, counts.ctypes.data)), reps=3)
            assert np.array_equal(out, planar) and int(counts.min()) == n
            rows[kind] = (e_ms, d_ms)
            for ptr in keep:
                lib.sela_hip_host_free(C.c_void_p(ptr))
        print(f"big: n {n:6d} x {nf:5d} stereo frames ({planar.nbytes / 1e6:.0f} MB of samples, {int(offs[nf]) / 1e6:.0f} MB of frames): "
              + "  ".join(f"{kind}: encode_i32 {e:7.3f} ms decode_i32 {d:7.3f} ms = {n * nf / e / 1e3:7.1f} / {n * nf / d / 1e3:7.1f} M samples/s" for kind, (e, d) in rows.items()))
