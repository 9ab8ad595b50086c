#!/usr/bin/env python3
"""Would ONE encode call gain by splitting itself over two streams with different kernels?  (not product
code.)  Times the analysis part of a 3875-frame encode as one k_encode_teams<0,16> launch against teams of 16 on the first
share of the frames beside k_encode_blocks on the rest, on two streams, for several shares and with / without the falling
wave priorities.  Whole encode calls are timed (their plan + assemble kernels, ~37 us, are in every figure)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sela_amd import capi, codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402

lib = capi.lib()
N = 3875
pcm = torch.from_numpy(synth_frames(N, 2, 0)).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(share, prio_teams, prio_blocks, second=0, reps=60):
    a = int(N * share)
    ea, eb = codec.Encoder(max(a, 1), 2), codec.Encoder(max(N - a, 1), 2)
    best = []
    for rep in range(reps + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if a:
            lib.sela_hip_debug_encode_teams(16)
            lib.sela_hip_debug_priorities(prio_teams)
            with torch.cuda.stream(s1):
                ea.encode(pcm[:a])
        if a < N:
            lib.sela_hip_debug_encode_teams(second)
            lib.sela_hip_debug_priorities(prio_blocks)
            with torch.cuda.stream(s2):
                eb.encode(pcm[a:])
        torch.cuda.synchronize()
        if rep >= 5:
            best.append(time.perf_counter() - t0)
    best.sort()
    return best[len(best) // 2] * 1e3


for share, pt, pb, second in ((1.0, 0x00010203, 0, 0), (1.0, 0, 0, 0), (0.8, 0, 0, 0), (0.7, 0, 0, 0), (0.6, 0, 0, 0), (0.5, 0, 0, 0), (0.7, 0x00010203, 0, 0),
                              (0.7, 0x00010203, 0x00010203, 0), (0.6, 0x00010203, 0x00010203, 0), (0.5, 0, 0, 16), (0.5, 0x00010203, 0x00010203, 16), (0.0, 0, 0, 0)):
    print(f"teams of 16 on {share:4.0%} of the frames (priorities {pt:08x}) beside {'k_encode_blocks' if second == 0 else 'teams of 16'} on the rest ({pb:08x}): {run(share, pt, pb, second):.4f} ms (host clock, median of 60)")
lib.sela_hip_debug_encode_teams(-1)
lib.sela_hip_debug_priorities_adaptive()
