#!/usr/bin/env python3
"""Encode n stereo frames `reps` times with the block kernel forced (sela_hip_debug_encode_teams): something to put under
rocprofv3.      python tools/teams_run.py n_frames team_lanes [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sela_amd import capi, codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402

n, teams = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
capi.lib().sela_hip_debug_encode_teams(teams)
pcm = torch.from_numpy(synth_frames(n, 2, 1)).cuda()
enc = codec.Encoder(n, 2)
for _ in range(reps):
    enc.encode(pcm)
torch.cuda.synchronize()
