#!/usr/bin/env python3
"""bench.py -- SELA frame encode+decode throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic PCM that is already resident in
HBM: encode the batch to the .sela frame stream, (N > 1: all-gather the per-rank compressed sizes
over RCCL, the only exchange the path has), decode the stream back to PCM.  The N = 1 workload is
BASELINE.json configs[1]: one 3-minute 16-bit stereo 44.1 kHz track = 3875 frames of 2048 samples.
With N ranks every rank processes its own track (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line: metric/value as BASELINE.json names them (Msamples/s, a sample = one
stereo pair that went through encode AND decode), plus
  "roofline"     -- the dominant kernel (k_encode_blocks): algorithmic bytes per launch / its average
                    duration measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  "cpu_baseline" -- the reference (oracle/_ref, kind "reference") or the CPU restatement (kind "port")
                    timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_QUARTER_RATE_GIPS = 256 * 4 * 2.4 / 4  # wave-instructions/ns the 1024 SIMDs can retire at 4 cycles each, 2.4 GHz peak clock
FP64_UNFUSED_PEAK_TOPS = 39.3  # vector FP64: 78.6 TFLOP/s counts an FMA as 2; mul and add issue separately here
# unfused FP64 operations the reference's analysis needs per 2048-sample block:
# 101 lags x (2048 - lag) x (mul + add) for the autocorrelation + 4950 Schur column updates x 4
FP64_OPS_PER_BLOCK = 2 * sum(2048 - i for i in range(101)) + 4 * 4950
TRACK_SECONDS, SAMPLE_RATE, CHANNELS = 180, 44100, 2


def cpu_baseline(pcm, repeats_target_s=12.0, gpu_frames=None, gpu_offsets=None, gpu_decoded=None):
    """Time the CPU path (encode + decode of the same frames) on the host cores.

    Uses the unmodified reference when oracle/_ref/libsela_ref.so travelled with the repo, else the
    CPU restatement.  Thread fan-out = the reference's static contiguous partition over
    hardware_concurrency() threads (src/sela/encoder.cpp:58-73).  Bounded sample: the whole
    3875-frame track, repeated until ~12 s of wall time or 3 repeats, whichever comes first.
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle, reference

    impl = reference()
    kind = "reference"
    if impl is None:
        impl, kind = oracle(), "port"
    cores = os.cpu_count() or 1
    n_frames, n, ch = pcm.shape
    enc_s = dec_s = 0.0
    reps = 0
    t_start = time.time()
    while reps < 3 and (reps == 0 or time.time() - t_start < repeats_target_s):
        blob, offs, es = impl.encode_frames(pcm, threads=cores)
        dec, ds = impl.decode_frames(blob, offs, ch, threads=cores)
        enc_s += es
        dec_s += ds
        reps += 1
    samples = reps * n_frames * n
    bit_exact = None
    if gpu_frames is not None:  # the CPU output doubles as the checker of what the GPU just produced
        import numpy as np

        bit_exact = bool(np.array_equal(blob, gpu_frames) and np.array_equal(offs, gpu_offsets) and np.array_equal(dec, gpu_decoded))
    return {
        "bit_exact_vs_gpu": bit_exact,
        "value": samples / (enc_s + dec_s) / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": kind,
        "encode_msps": samples / enc_s / 1e6,
        "decode_msps": samples / dec_s / 1e6,
        "sample": f"{reps} x the full {n_frames}-frame stereo track, encode+decode, {cores} threads "
                  f"(static contiguous frame partition as src/sela/encoder.cpp:58-73)",
    }


def _flush_c_stdio():
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def measured_valu_instructions(kernel: str):
    """Vector instructions per launch of `kernel` (SQ_INSTS_VALU, whole GPU) from the newest committed counter
    summary (profiles/rNN/valu_counters.txt, written by tools/valu_counters.sh).  None if there is none."""
    import ast
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "valu_counters.txt")))
    if not files:
        return None
    with open(files[-1]) as f:
        for line in f:
            if kernel in line and "SQ_INSTS_VALU" in line:
                try:
                    return float(ast.literal_eval(line[line.index("{"):].strip())["SQ_INSTS_VALU"])
                except (ValueError, SyntaxError, KeyError):
                    return None
    return None


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN/traffic.json,
    written by tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes of this same command,
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if no summary is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        data = json.load(f)
    for name, d in data.items():
        if kernel in name and "hbm_bytes_per_launch_fetch_x2" in d:
            return d["hbm_bytes_per_launch_fetch_x2"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from sela_amd import capi, codec
    from sela_amd.synth import frames_for_seconds, synth_frames

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("SELA_BENCH_FORCE_EXCHANGE") == "1":  # (the override runs the N>1 code path on one GPU)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    lib = capi.lib()  # raises if the HIP library is missing: there is no CPU fallback
    n_frames = frames_for_seconds(TRACK_SECONDS, SAMPLE_RATE)  # 3875
    pcm_host = synth_frames(n_frames, CHANNELS, track=rank)
    pcm = torch.from_numpy(pcm_host).cuda()
    enc = codec.Encoder(n_frames, CHANNELS)
    dec = codec.Decoder(n_frames, CHANNELS)
    all_ends = torch.zeros(world * n_frames, dtype=torch.int64, device="cuda")
    exchange = torch.cuda.Stream() if dist is not None else None

    def step():
        out = enc.encode(pcm)
        if dist is not None:
            # the path's only exchange: every rank learns where every frame of the job
            # lands in the output stream.  The encoder has already scanned its own frame sizes, so what is
            # gathered are each rank's local frame END offsets (8 bytes x frames per rank, latency bound --
            # RCCL over xGMI); a frame's global position is its local offset plus the totals of the ranks
            # before it, which is O(ranks) arithmetic on the gathered array (sela_amd/sharding.py does the
            # same with sizes).  Decoding the local frames does not need the layout, so the collective
            # runs beside it on its own stream and is joined at the end of the step, inside the timed region.
            exchange.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(exchange):
                dist.all_gather_into_tensor(all_ends, out.offsets[1:])
        back = dec.decode(out.frames, out.offsets, n_frames)
        if dist is not None:
            torch.cuda.current_stream().wait_stream(exchange)
        return out, back

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, back = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: status words + round trip.  The reference codec is not lossless on
    # every frame (its encoder rounds the prediction half-up, its decoder half-down: a frame whose Q35
    # sum hits 2^34 mod 2^35 comes back off by one), and parity means reproducing
    # that -- so the round trip may differ from the input in a handful of frames, never in many.
    out.check()
    dec.check()
    lossy_frames = int((back != pcm).reshape(n_frames, -1).any(dim=1).sum().item())
    assert lossy_frames <= max(1, n_frames // 500), f"decode(encode(x)) differs from x in {lossy_frames} frames"
    payload_bytes = out.total_bytes()

    # ---- per-kernel timing leg (separate from the timed region: events add launch gaps) --------------
    lib.sela_hip_enable_kernel_timing(1)
    k_enc, k_dec = [], []
    for _ in range(max(5, min(args.steps, 20))):
        o2 = enc.encode(pcm)
        k_enc.append(capi.kernel_times(3))
        dec.decode(o2.frames, o2.offsets, n_frames)
        k_dec.append(capi.kernel_times(2))
    lib.sela_hip_enable_kernel_timing(0)
    torch.cuda.synchronize()
    k_enc = np.array(k_enc)  # [reps, 3] ms: blocks, plan, assemble
    k_dec = np.array(k_dec)  # [reps, 2] ms: parse, synthesize
    enc_blocks_ms = float(k_enc[:, 0].mean())
    pcm_bytes = pcm_host.nbytes
    algo_bytes = pcm_bytes + payload_bytes  # PCM16 read + .sela frame bytes written
    achieved = algo_bytes / (enc_blocks_ms * 1e-3) / 1e9
    traffic = measured_traffic("k_encode_blocks")
    valu_instr = measured_valu_instructions("k_encode_blocks")

    if rank == 0:
        samples = n_frames * 2048
        ms_per_step = elapsed / args.steps * 1e3
        enc_ms = float(k_enc.sum(axis=1).mean())
        dec_ms = float(k_dec.sum(axis=1).mean())
        result = {
            "metric": "Msamples/s encode+decode, 16-bit stereo 44.1kHz",
            "value": world * samples / (elapsed / args.steps) / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64+int64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: one 3-min 16-bit stereo 44.1 kHz track per GPU "
                            "(3875 frames x 2048 stereo samples), encode to .sela frames then decode, bit-exact",
                "frames_per_gpu": n_frames, "channels": CHANNELS, "sharding": f"track-per-rank x{world}",
                "sela_bytes_per_gpu": payload_bytes, "pcm_bytes_per_gpu": pcm_bytes,
            },
            "encode_msps_kernels": samples / (enc_ms * 1e-3) / 1e6,
            "decode_msps_kernels": samples / (dec_ms * 1e-3) / 1e6,
            "kernel_ms": {"encode_blocks": enc_blocks_ms, "encode_plan": float(k_enc[:, 1].mean()),
                          "encode_assemble": float(k_enc[:, 2].mean()), "decode_parse": float(k_dec[:, 0].mean()),
                          "decode_synthesize": float(k_dec[:, 1].mean())},
            "roundtrip_lossy_frames": lossy_frames,
            "fp64_valu": {  # the resource that actually binds k_encode_blocks
                "achieved": FP64_OPS_PER_BLOCK * n_frames * 3 / (enc_blocks_ms * 1e-3) / 1e12, "peak": FP64_UNFUSED_PEAK_TOPS,
                "unit": "T unfused FP64 op/s", "frac": FP64_OPS_PER_BLOCK * n_frames * 3 / (enc_blocks_ms * 1e-3) / 1e12 / FP64_UNFUSED_PEAK_TOPS,
            },
            # what binds the kernel in practice: issue slots of the vector ALU.  Quarter-rate instructions
            # (FP64, 64-bit integer multiply-add: most of this kernel) take 4 cycles of a SIMD each.
            "valu_issue": None if valu_instr is None else {
                "achieved": valu_instr / (enc_blocks_ms * 1e-3) / 1e9, "peak": VALU_QUARTER_RATE_GIPS, "unit": "G wave-instructions/s",
                "frac": valu_instr / (enc_blocks_ms * 1e-3) / 1e9 / VALU_QUARTER_RATE_GIPS,
                "instructions_per_launch": valu_instr,
                "note": "SQ_INSTS_VALU from profiles/ (committed counter pass) / live kernel time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles",
            },
            "roofline": {
                "kernel": "k_encode_blocks", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "the path is FP64-issue/latency bound, not HBM bound (DESIGN.md): 7 B per stereo sample",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            g_frames, g_offsets = out.to_host()
            result["cpu_baseline"] = cpu_baseline(pcm_host, gpu_frames=g_frames, gpu_offsets=g_offsets, gpu_decoded=back.cpu().numpy())
            assert result["cpu_baseline"]["bit_exact_vs_gpu"], "GPU output differs from the CPU reference"
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()  # RCCL prints its version banner through C stdio; keep the JSON line the LAST line of stdout
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
