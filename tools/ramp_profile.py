#!/usr/bin/env python3
"""How long the waves of a k_encode_teams launch run (s_memtime at a wave's start and end), from the instrumented build:
    python tools/ramp_profile.py [n_frames [team_lanes]]
Prints the spread of the waves' own durations."""
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
# s_memtime has a base of its own per XCD (and, by what the start times show, per shader engine): start times of waves on
# different ones are not comparable, and no grouping by gaps separates them reliably -- only a wave's OWN duration is
# printed (start and end are read by the same wave).
per_wave_start = np.unique(start)
last_end = np.array([end[start == s0].max() for s0 in per_wave_start])
dur = last_end - per_wave_start
q = lambda a, p: float(np.percentile(a, p))
print(f"{len(per_wave_start)} waves; duration in s_memtime ticks (= shader clocks here: the slowest wave is the launch's length): "
      "min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f  mean %.0f" % (dur.min(), q(dur, 10), q(dur, 50), q(dur, 90), dur.max(), dur.mean()))
