#!/usr/bin/env python3
"""k_encode_blocks against k_encode_teams<8 / 16>: kernel time of the block kernel (the library's own HIP events) and wall
time of the whole encode against batch size, the kernel forced through sela_hip_debug_encode_teams.  A tuning aid for
launch_encode's choice (team_lanes_for), quoted in DESIGN.md."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sela_amd import capi, codec  # noqa: E402
from sela_amd.synth import synth_frames  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [250, 500, 1000, 2000, 3000, 3875, 6000, 10000, 20000, 40000]
    lib = capi.lib()
    lib.sela_hip_kernel_times.argtypes = [C.POINTER(C.c_float), C.c_int]
    lib.sela_hip_kernel_times.restype = C.c_int
    print(f"{'frames':>8} {'teams':>6} {'blocks kernel ms':>17} {'encode wall ms':>15} {'G samples/s':>12}")
    for n in sizes:
        pcm = torch.from_numpy(synth_frames(n, 2, 1)).cuda()
        enc = codec.Encoder(n, 2)
        ref = None
        lib.sela_hip_debug_encode_teams(-1)
        pick = lib.sela_hip_debug_encode_kernel(n, 2) # the library's own choice for this size (team_lanes_for)
        for teams in (0, 16, 8):
            lib.sela_hip_debug_encode_teams(teams)
            out = enc.encode(pcm)
            torch.cuda.synchronize()
            got = (out.frames[: out.total_bytes()].clone(), out.offsets.clone())
            if ref is None:
                ref = got
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (n, teams)
            reps = 20 if n <= 10000 else 5
            ks = []
            t0 = time.perf_counter()
            for _ in range(reps):
                enc.encode(pcm)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            lib.sela_hip_enable_kernel_timing(1)
            for _ in range(5):
                enc.encode(pcm)
                torch.cuda.synchronize()
                ms = (C.c_float * 8)()
                k = lib.sela_hip_kernel_times(ms, 8)
                ks.append(ms[0] if k else float("nan"))
            lib.sela_hip_enable_kernel_timing(0)
            ks.sort()
            print(f"{n:8d} {teams:6d} {ks[len(ks) // 2]:17.4f} {wall * 1e3:15.4f} {n * 2048 / wall / 1e9:12.2f}" + ("   <- the library's pick" if teams == pick else ""), flush=True)
        lib.sela_hip_debug_encode_teams(-1)


if __name__ == "__main__":
    main()
