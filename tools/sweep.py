#!/usr/bin/env python3
"""Encode / decode wall time (HBM-resident, stream-synchronised) against batch size.  Not a bench line --
a scaling sanity check quoted in DESIGN.md."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sela_amd import codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 1000, 3875, 10000, 40000]
    print(f"{'frames':>8} {'enc ms':>9} {'dec ms':>9} {'enc Gs/s':>9} {'dec Gs/s':>9}")
    for n in sizes:
        pcm = torch.from_numpy(synth_frames(n, 2, 1)).cuda()
        enc, dec = codec.Encoder(n, 2), codec.Decoder(n, 2)
        out = enc.encode(pcm)
        reps = 20 if n <= 10000 else 5
        te = timed(lambda: enc.encode(pcm), reps)
        td = timed(lambda: dec.decode(out.frames, out.offsets, n), reps)
        print(f"{n:8d} {te * 1e3:9.3f} {td * 1e3:9.3f} {n * 2048 / te / 1e9:9.2f} {n * 2048 / td / 1e9:9.2f}")


if __name__ == "__main__":
    main()


def host_api(n=3875):
    """PCIe-inclusive rate of the host-pointer API through the Python wrapper (pageable numpy buffers: H2D + kernels + D2H."""
    import numpy as np

    pcm = synth_frames(n, 2, 0)
    frames, offs = codec.encode_host(pcm)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        codec.encode_host(pcm)
    te = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        codec.decode_host(frames, offs, 2)
    td = (time.perf_counter() - t0) / reps
    print(f"host-pointer API through the Python wrapper, {n} frames: encode {te * 1e3:.2f} ms, decode {td * 1e3:.2f} ms "
          f"(each call allocates and first-touches its numpy output; host/sela_filebench times the C++ path on page-locked buffers)")


if __name__ == "__main__" and os.environ.get("SELA_SWEEP_HOST", "1") == "1":
    host_api()
