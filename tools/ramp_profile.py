#!/usr/bin/env python3
"""When the waves of a k_encode_teams launch start and end (s_memtime, absolute), from the instrumented build:
    python tools/ramp_profile.py [n_frames [team_lanes]]
Prints the spread of the start times (the dispatcher's ramp), of the end times, and the waves' own durations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sela_amd import capi, codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3875
teams = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = capi.lib()
lib.sela_hip_debug_encode_teams(teams)
pcm = torch.from_numpy(synth_frames(n, 2, 0)).cuda()
enc = codec.Encoder(n, 2)
enc.encode(pcm)
torch.cuda.synchronize()
buf = torch.zeros(n * 3 * 16, dtype=torch.int64, device="cuda")
lib.sela_hip_debug_phase_buffer(buf.data_ptr())
for _ in range(2):
    buf.zero_()
    enc.encode(pcm)
    torch.cuda.synchronize()
lib.sela_hip_debug_phase_buffer(None)
raw = buf.cpu().numpy().reshape(-1, 16)
start, end = raw[:, 13].astype(np.float64), raw[:, 14].astype(np.float64)
ok = start > 0
start, end = start[ok], end[ok]
t0 = start.min()
per_wave_start = np.unique(start)
print(f"{len(per_wave_start)} waves; launch spans {(end.max() - t0):.0f} cycles of s_memtime (100 MHz ticks x ... see below)")
q = lambda a, p: float(np.percentile(a, p))
print("wave start after the first wave's: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % tuple(q(per_wave_start - t0, p) for p in (10, 50, 90, 100)))
last_end = np.array([end[start == s0].max() for s0 in per_wave_start])
print("wave end   after the first wave's start: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % tuple(q(last_end - t0, p) for p in (10, 50, 90, 100)))
dur = last_end - per_wave_start
print("wave duration: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % tuple(q(dur, p) for p in (10, 50, 90, 100)))
