#!/usr/bin/env python3
"""k_decode_frames per launch against the batch size with the recurrence forced into each of its two forms and picked by launch
size (profiles/r03/decode_forms.txt).  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sela_amd import capi, codec
from sela_amd.synth import synth_frames_torch
lib = capi.lib()
print("frames | dec ms scalar-shift | dec ms vector-shift | dec ms by size")
for n in (1, 64, 500, 1000, 1250, 1500, 2000, 3000, 3875, 10000, 40000):
    pcm = synth_frames_torch(n, 2, 0, device="cuda")
    enc = codec.Encoder(n, 2); dec = codec.Decoder(n, 2)
    out = enc.encode(pcm); torch.cuda.synchronize()
    row = []
    for form in (0, 1, -1):
        lib.sela_hip_debug_decode_recurrence(form)
        for _ in range(3): dec.decode(out.frames, out.offsets, n)
        torch.cuda.synchronize()
        reps = 30 if n <= 10000 else 8
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): dec.decode(out.frames, out.offsets, n)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        row.append(best)
    lib.sela_hip_debug_decode_recurrence(-1)
    print(f"{n:6d} | {row[0]:.4f} | {row[1]:.4f} | {row[2]:.4f}")
