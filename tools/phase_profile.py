#!/usr/bin/env python3
"""Per-phase cycle breakdown of the encode/decode kernels on the bench workload (debug hook).

    python tools/phase_profile.py [n_frames [team_lanes]]     (team_lanes 8 / 16: k_encode_teams, where the mean / autocorr /
                                                              Schur columns are a WAVE's phases, shared by its 64 / team_lanes blocks,
                                                              and the last column is the time since the end of the wave's analysis)

Prints the mean s_memtime cycles each phase takes per block (encoder: per (frame, signal);
decoder: per subframe).  Quoted in DESIGN.md; not part of the timed path.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sela_amd import capi, codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402

ENC = ["load+x/32767", "mean chain", "centre", "autocorr", "normalise+Schur", "order/quant/dequant", "step-up",
       "FIR residues", "Rice k search", "pack coefs", "transpose+pack residues", "store slot/meta"]
DEC = ["headers + mode barrier", "stream -> LDS", "parse A (own zone)", "parse B (merge)", "resolve chain",
       "pass 2 (decode)", "dequant + step-up + table", "synthesis", "barrier + combine + store"]


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3875
    lib = capi.lib()
    teams = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    lib.sela_hip_debug_encode_teams(teams)
    pcm = torch.from_numpy(synth_frames(n_frames, 2, 0)).cuda()
    enc, dec = codec.Encoder(n_frames, 2), codec.Decoder(n_frames, 2)
    out = enc.encode(pcm)
    torch.cuda.synchronize()
    buf = torch.zeros(n_frames * 3 * 16, dtype=torch.int64, device="cuda")
    lib.sela_hip_debug_phase_buffer(buf.data_ptr())
    enc.encode(pcm)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(-1, 16)
    e = raw[:, :12].astype(np.float64)
    if teams > 0:
        print(f"k_encode_teams<{teams}>: since the end of the wave's analysis, per block (its own tail and those before it): mean {raw[:, 12].mean():.0f}, max {raw[:, 12].max()}")
    buf.zero_()
    dec.decode(out.frames, out.offsets, n_frames)
    torch.cuda.synchronize()
    lib.sela_hip_debug_phase_buffer(None)
    dd = buf.cpu().numpy().reshape(-1, 16)[: n_frames * 2].astype(np.float64)
    d = dd[:, :9]
    print(f"encoder, mean cycles per (frame, signal) block over {len(e)} blocks (total {e.sum(1).mean():.0f}):")
    for name, v in zip(ENC, e.mean(0)):
        print(f"  {name:28s} {v:10.0f}  {100 * v / e.sum(1).mean():5.1f}%")
    if teams <= 0:
        tot = e.sum(1)
        print("  a block's total: " + "  ".join(f"p{q} {np.percentile(tot, q):.0f}" for q in (0, 10, 50, 90, 100))
              + "   (p100 against the kernel's duration = what the launch and the dispatch of its waves add)")
        for name, col in (("mean chain", 1), ("autocorr", 3), ("FIR residues", 7)):
            print(f"  {name}: " + "  ".join(f"p{q} {np.percentile(e[:, col], q):.0f}" for q in (0, 10, 50, 90, 100)))
    print(f"decoder (k_decode_frames), mean cycles per subframe over {len(d)} subframes (total {d.sum(1).mean():.0f}):")
    for name, v in zip(DEC, d.mean(0)):
        print(f"  {name:28s} {v:10.0f}  {100 * v / d.sum(1).mean():5.1f}%")
    tot = d.sum(1)
    print("  a subframe's total: " + "  ".join(f"p{q} {np.percentile(tot, q):.0f}" for q in (0, 10, 50, 90, 100)))
    print("  synthesis: " + "  ".join(f"p{q} {np.percentile(d[:, 7], q):.0f}" for q in (0, 10, 50, 90, 100)))


if __name__ == "__main__":
    main()
