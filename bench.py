#!/usr/bin/env python3
"""bench.py -- SELA frame encode+decode throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over synthetic PCM that is already resident in HBM: encode to the
.sela frame stream, decode the stream back to PCM.

  N = 1   BASELINE.json configs[1]: one 3-minute 16-bit stereo 44.1 kHz track = 3875 frames of 2048 samples.
  N > 1   BASELINE.json configs[3]: the 100-track album (34/33/33 tracks at 44.1/48/96 kHz, 549,365 frames),
          STRONG scaling: the (track, frame) space is cut into N contiguous balanced ranges
          (sela_amd.sharding.partition -- the reference's static partition, src/sela/encoder.cpp:58-73), every
          rank encodes and decodes its range in batches of <= 65,536 frames, and the ranks all-gather the
          compressed frame sizes over RCCL -- the only exchange the path has -- inside the
          timed region, beside the decode of the last batch.  The gathered layout is checked against the
          digest of the one-GPU (reference) layout.  `--workload album` runs the same job on one GPU.

K steps are timed, bracketed by barrier + torch.cuda.synchronize(), MAX over ranks; that measurement is
repeated until at least ~0.5 s has been timed and `value` is the median repetition (min / max reported; the
first repetition, which runs while the clocks still settle, is reported separately).

Rank 0 prints ONE JSON line: metric/value as BASELINE.json names them (Msamples/s, a sample = one stereo pair
that went through encode AND decode), plus
  "roofline"      the dominant kernel (k_encode_blocks): algorithmic bytes per launch / its average duration
                  measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  "cpu_baseline"  the reference (oracle/_ref, kind "reference") or the CPU restatement (kind "port") timed on
                  this box's host cores on a bounded sample of the same workload
  "e2e"           host-pointer API (H2D + kernels + D2H, steady_clock) and "file_to_file" (file read .. file
                  write, the reference's `sela -e` / `-d`), measured by host/sela_filebench (N = 1 only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
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
ALBUM_BATCH_FRAMES = 65536
ENCODE_TARGET_MSPS = 1000.0  # BASELINE.json north_star: >= 1 G stereo samples/s encode on one MI355X


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


def _newest_profile(name):
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    return files[-1] if files else None


def committed_valu_instructions(kernel: str):
    """(instructions per launch, source file): SQ_INSTS_VALU of `kernel` from the newest COMMITTED counter
    summary (profiles/rNN/valu_counters.txt, written by tools/valu_counters.sh) -- not measured in this run."""
    import ast

    path = _newest_profile("valu_counters.txt")
    if not path:
        return None, None
    with open(path) as f:
        for line in f:
            if kernel in line and "SQ_INSTS_VALU" in line:
                try:
                    return float(ast.literal_eval(line[line.index("{"):].strip())["SQ_INSTS_VALU"]), os.path.relpath(path, ROOT)
                except (ValueError, SyntaxError, KeyError):
                    return None, None
    return None, None


def committed_traffic(kernel: str):
    """(HBM bytes per launch, source file) of `kernel` from the newest COMMITTED PMC summary
    (profiles/rNN/traffic.json, written by tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes of
    this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md) -- not measured in this run."""
    path = _newest_profile("traffic.json")
    if not path:
        return None, None
    with open(path) as f:
        data = json.load(f)
    for name, d in data.items():
        if kernel in name and "hbm_bytes_per_launch_fetch_x2" in d:
            return d["hbm_bytes_per_launch_fetch_x2"], os.path.relpath(path, ROOT)
    return None, None


def host_legs(pcm_host, repeats=9):
    """e2e (host-pointer API) and file-to-file numbers from the C++ host (host/sela_filebench), or None."""
    import struct

    exe = os.path.join(ROOT, "host", "sela_filebench")
    if not os.path.exists(exe):
        return None
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        wav = os.path.join(tmp, "track.wav")
        data = pcm_host.astype("<i2").tobytes()
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt "
                    + struct.pack("<IhHIIHH", 16, 1, CHANNELS, SAMPLE_RATE, SAMPLE_RATE * CHANNELS * 2, CHANNELS * 2, 16)
                    + b"data" + struct.pack("<I", len(data)) + data)
        try:
            out = subprocess.run([exe, wav, tmp, str(repeats)], capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                return {"error": (out.stderr or out.stdout).strip()[-300:]}
            res = json.loads(out.stdout.strip().splitlines()[-1])
        except (subprocess.TimeoutExpired, ValueError, OSError) as e:
            return {"error": str(e)[-300:]}
    res["files_on"] = "tmpfs (/dev/shm)" if scratch else "the temp directory's file system"
    return res


def timed_repetitions(step, barrier, steps, dist, min_total_s=0.5, max_reps=40):
    """Seconds per K-step measurement, each bracketed by barrier + synchronize, MAX over ranks."""
    import torch

    out = []
    total = 0.0
    while len(out) < max_reps and (not out or total < min_total_s):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            result = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed, total + elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the same numbers -- and the same loop exit -- on every rank
            elapsed, agreed_total = float(t[0].item()), float(t[1].item())
        else:
            agreed_total = total + elapsed
        out.append(elapsed)
        total = agreed_total
    return out, result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["track", "album"], default=None, help="default: track for --gpus 1, album otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, help="batches in flight: independent encode->decode chains on their own HIP streams")
    args = ap.parse_args()

    import numpy as np
    import torch

    from sela_amd import capi, codec, sharding
    from sela_amd.synth import album_tracks, frames_for_seconds, synth_frames, synth_frames_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    workload = args.workload or ("track" if world == 1 else "album")
    assert workload == "album" or world == 1, "the single track is the one-GPU workload"
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
    exchange = torch.cuda.Stream() if dist is not None else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the workload, resident in HBM ---------------------------------------------------------------------
    if workload == "track":
        n_total = frames_for_seconds(TRACK_SECONDS, SAMPLE_RATE)  # 3875
        pcm_host = synth_frames(n_total, CHANNELS, track=0)
        batches = [torch.from_numpy(pcm_host).cuda()]
        my_begin, my_end = 0, n_total
        tracks = [(0, SAMPLE_RATE, n_total)]
    else:
        tracks = album_tracks()
        starts = np.concatenate([[0], np.cumsum([f for _, _, f in tracks])]).astype(np.int64)
        n_total = int(starts[-1])  # 549,365
        my_begin, my_end = sharding.my_range(n_total, rank, world)
        parts = []  # this rank's contiguous range of the album's (track, frame) space, generated on the GPU
        for track, _, frames in tracks:
            b, e = max(my_begin, int(starts[track])), min(my_end, int(starts[track + 1]))
            if b < e:
                parts.append(synth_frames_torch(e - b, CHANNELS, track, first_frame=b - int(starts[track]), device="cuda"))
        local = torch.cat(parts) if parts else torch.zeros((0, 2048, CHANNELS), dtype=torch.int16, device="cuda")
        del parts
        n_batches = max(1, -(-int(local.shape[0]) // ALBUM_BATCH_FRAMES))  # equal batches of <= 65,536 frames
        per_batch = max(1, -(-int(local.shape[0]) // n_batches))
        batches = [local[i: i + per_batch] for i in range(0, local.shape[0], per_batch)] or [local]
        pcm_host = None
    n_local = my_end - my_begin
    max_batch = max(int(b.shape[0]) for b in batches)
    # Lanes: consecutive batches (N = 1: consecutive steps) are independent encode -> decode chains, so they run on
    # alternating HIP streams with their own buffers: the decode of one batch fills the launch tail of the next
    # batch's encode and the other way round.  Everything issued inside the timed region completes inside it
    # (device-wide synchronize on both sides); `--lanes 1` is the strictly serial form, reported beside it.
    n_lanes = max(1, args.lanes)
    lanes = [{"enc": codec.Encoder(max(max_batch, 1), CHANNELS), "dec": codec.Decoder(max(max_batch, 1), CHANNELS),
              "stream": torch.cuda.Stream()} for _ in range(n_lanes)]
    enc, dec = lanes[0]["enc"], lanes[0]["dec"]
    slot = {"next": 0}
    max_local = max(e - b for b, e in sharding.partition(n_total, world))
    local_sizes = torch.zeros(max_local, dtype=torch.int64, device="cuda")
    all_sizes = torch.zeros(world * max_local, dtype=torch.int64, device="cuda")
    state = {"lossy": 0, "bytes": 0}

    def step(check=False, serial=False):
        at = 0
        for i, pcm in enumerate(batches):
            nb = int(pcm.shape[0])
            lane = lanes[0 if serial else slot["next"] % n_lanes]
            slot["next"] += 1
            with torch.cuda.stream(lane["stream"]):
                out = lane["enc"].encode(pcm)
                if dist is not None:
                    local_sizes[at: at + nb] = out.offsets[1:] - out.offsets[:-1]
                    if i == len(batches) - 1:
                        # the path's only exchange: every rank learns the size of every frame of the
                        # job, i.e. where its bytes land in every output file (8 bytes x frames, latency bound -- RCCL
                        # over xGMI).  Decoding does not need the layout, so the collective runs beside the last
                        # decode on its own stream and is joined at the end of the step, inside the timed region.
                        for other in lanes:  # (the sizes of the earlier batches were written on the other lanes' streams)
                            exchange.wait_stream(other["stream"])
                        with torch.cuda.stream(exchange):
                            dist.all_gather_into_tensor(all_sizes, local_sizes)
                back = lane["dec"].decode(out.frames, out.offsets, nb)
                if dist is not None and i == len(batches) - 1:
                    lane["stream"].wait_stream(exchange)
            if check:  # (outside the timed region) status words + round trip of every batch
                torch.cuda.synchronize()
                out.check()
                lane["dec"].check()
                state["lossy"] += int((back != pcm).reshape(nb, -1).any(dim=1).sum().item()) if nb else 0
                state["bytes"] += out.total_bytes()
            at += nb
        return out, back

    for _ in range(args.warmup):
        step()
    reps, (out, back) = timed_repetitions(step, barrier, args.steps, dist)
    serial_reps, _ = timed_repetitions(lambda: step(serial=True), barrier, args.steps, dist, min_total_s=0.15) if n_lanes > 1 else (reps, None)
    # the first K-step measurement runs a few % slow while the clocks settle behind the W warm-up steps: with four
    # measurements or more it is reported (ms_per_step_first) but kept out of median / min / max
    settled = reps[1:] if len(reps) >= 4 else reps
    per_step = sorted(r / args.steps for r in settled)
    median_s = per_step[len(per_step) // 2]

    # ---- correctness of what was timed ------------------------------------------------------------------------
    # status words + round trip.  The reference codec is not lossless on every frame (its encoder rounds the
    # prediction half-up, its decoder half-down: a frame whose Q35 sum hits 2^34 mod 2^35 comes back off by one;
    # DESIGN.md section 2), and parity means reproducing that -- so the round trip may differ from the input in a
    # handful of frames, never in many.
    out, back = step(check=True)
    torch.cuda.synchronize()
    lossy_frames, payload_bytes = state["lossy"], state["bytes"]
    assert lossy_frames <= max(1, n_local // 500), f"decode(encode(x)) differs from x in {lossy_frames} frames"
    layout_ok = None
    if dist is not None:
        # the gathered layout must be the one-GPU layout: its digest was computed with the unmodified reference
        ranges = sharding.partition(n_total, world)
        g = all_sizes.cpu().numpy().reshape(world, max_local)
        sizes = np.concatenate([g[r, : e - b] for r, (b, e) in enumerate(ranges)]).astype("<u8")
        if workload == "album":
            with open(os.path.join(ROOT, "tests", "golden", "album_digests.json")) as f:
                golden = json.load(f)
            offs = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64))])
            for t, (track, _, frames) in enumerate(tracks):
                b, e = int(starts[t]), int(starts[t + 1])
                assert 15 + int(offs[e] - offs[b]) == golden["tracks"][t]["sela_bytes"], f"track {track}: file size differs from the reference's"
            layout_ok = golden.get("frame_sizes_sha256") in (None, hashlib.sha256(sizes.tobytes()).hexdigest())
            assert layout_ok, "the gathered frame-size layout differs from the one-GPU (reference) layout"
        t = torch.tensor([lossy_frames, payload_bytes], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        lossy_frames, payload_bytes = int(t[0].item()), int(t[1].item())

    # ---- per-kernel timing leg (separate from the timed region: events add launch gaps) --------------
    lib.sela_hip_enable_kernel_timing(1)
    k_enc, k_dec = [], []
    pcm0 = batches[0]
    n0 = int(pcm0.shape[0])
    o2 = enc.encode(pcm0) if n0 else None
    for _ in range(max(5, min(args.steps, 20)) if n0 else 0):
        # The timed kernel runs where it runs in a step: the encoder behind a decode, the decoder behind an encode, two
        # steps queued so that it starts on a busy, clocked-up device (a launch onto an idle device runs ~10 % slower;
        # an encoder behind an encoder 6 % slower than behind a decoder -- the device's clock follows the power the last
        # kernel drew -- and it is behind a decoder that the rocprofv3 averages of the timed loop see it).
        for _ in range(2):
            dec.decode(o2.frames, o2.offsets, n0)
            o2 = enc.encode(pcm0)
        k_enc.append(capi.kernel_times(3))
        for _ in range(2):
            o2 = enc.encode(pcm0)
            dec.decode(o2.frames, o2.offsets, n0)
        k_dec.append(capi.kernel_times(1))
    lib.sela_hip_enable_kernel_timing(0)
    torch.cuda.synchronize()

    if rank == 0:
        k_enc = np.array(k_enc)  # [reps, 3] ms: blocks, plan, assemble
        k_dec = np.array(k_dec)  # [reps, 1] ms: the fused decode kernel
        enc_blocks_ms = float(k_enc[:, 0].mean())
        pcm_bytes0 = n0 * 2048 * CHANNELS * 2
        algo_bytes = pcm_bytes0 + o2.total_bytes()  # PCM16 read + .sela frame bytes written, one launch
        achieved = algo_bytes / (enc_blocks_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic("k_encode_blocks")
        valu_instr, valu_src = committed_valu_instructions("k_encode_blocks")
        if workload != "track":
            valu_instr = traffic = None  # the committed counter passes are of the single-track launch
        samples = n_total * 2048
        samples0 = n0 * 2048
        enc_ms = float(k_enc.sum(axis=1).mean())
        dec_ms = float(k_dec.sum(axis=1).mean())
        fp64 = FP64_OPS_PER_BLOCK * n0 * 3 / (enc_blocks_ms * 1e-3) / 1e12
        result = {
            "metric": "Msamples/s encode+decode, 16-bit stereo 44.1kHz" if workload == "track"
                      else "Msamples/s encode+decode, 16-bit stereo 44.1/48/96kHz album",
            "value": samples / median_s / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": median_s * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if workload == "track" else "strong",
            "vs_baseline": None,
            "dtype": "f64+int64",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: one 3-min 16-bit stereo 44.1 kHz track (3875 frames x 2048 stereo samples), "
                             "encode to .sela frames then decode, bit-exact") if workload == "track" else
                            ("BASELINE.json configs[3]: 100-track synthetic album (34/33/33 tracks at 44.1/48/96 kHz, 549,365 frames x 2048 "
                             f"stereo samples) sharded over {world} GPU(s) in contiguous frame ranges, encode + RCCL all-gather of the "
                             "frame sizes + decode, layout checked against the reference's"),
                "frames_total": n_total, "frames_rank0": n_local, "channels": CHANNELS,
                "sharding": f"contiguous frame ranges x{world}" if workload == "album" else "single track",
                "batch_frames": max_batch, "sela_bytes_total": payload_bytes, "pcm_bytes_total": n_total * 2048 * CHANNELS * 2,
            },
            "lanes": {"in_flight": n_lanes, "ms_per_step_one_lane": sorted(serial_reps)[len(serial_reps) // 2] / args.steps * 1e3,
                      "value_one_lane": samples / (sorted(serial_reps)[len(serial_reps) // 2] / args.steps) / 1e6,
                      "what": "consecutive batches are independent encode->decode chains on alternating HIP streams; one lane = strictly serial"},
            "repetitions": {"count": len(reps), "timed_s": sum(reps), "ms_per_step_first": reps[0] / args.steps * 1e3,
                            "ms_per_step_min": per_step[0] * 1e3,
                            "ms_per_step_median": median_s * 1e3, "ms_per_step_max": per_step[-1] * 1e3,
                            "spread_frac": (per_step[-1] - per_step[0]) / median_s,
                            "spread_frac_p10_p90": (per_step[(9 * len(per_step)) // 10 - (1 if len(per_step) >= 10 else 0)] - per_step[len(per_step) // 10]) / median_s},
            "encode_msps": samples0 / (enc_ms * 1e-3) / 1e6,   # kernels of the first batch, HBM resident
            "decode_msps": samples0 / (dec_ms * 1e-3) / 1e6,
            "encode_target": {"msps": ENCODE_TARGET_MSPS, "met": bool(samples0 / (enc_ms * 1e-3) / 1e6 >= ENCODE_TARGET_MSPS),
                              "ratio": samples0 / (enc_ms * 1e-3) / 1e6 / ENCODE_TARGET_MSPS},
            "kernel_ms": {"encode_blocks": enc_blocks_ms, "encode_plan": float(k_enc[:, 1].mean()),
                          "encode_assemble": float(k_enc[:, 2].mean()), "decode_frames": dec_ms, "frames_in_launch": n0},
            "roundtrip_lossy_frames": lossy_frames,
            "layout_matches_reference": layout_ok,
            "fp64_valu": {  # the resource that actually binds k_encode_blocks
                "achieved": fp64, "peak": FP64_UNFUSED_PEAK_TOPS, "unit": "T unfused FP64 op/s", "frac": fp64 / FP64_UNFUSED_PEAK_TOPS,
            },
            # what binds the kernel in practice: issue slots of the vector ALU.  Quarter-rate instructions
            # (FP64, 64-bit integer multiply-add: most of this kernel) take 4 cycles of a SIMD each.
            "valu_issue": None if valu_instr is None else {
                "achieved": valu_instr / (enc_blocks_ms * 1e-3) / 1e9, "peak": VALU_QUARTER_RATE_GIPS, "unit": "G wave-instructions/s",
                "frac": valu_instr / (enc_blocks_ms * 1e-3) / 1e9 / VALU_QUARTER_RATE_GIPS,
                "instructions_per_launch_from_profiles": valu_instr, "profiles_source": valu_src,
                "note": "SQ_INSTS_VALU from the committed counter pass named in profiles_source / live kernel time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles",
            },
            "roofline": {
                "kernel": "k_encode_blocks", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from_profiles": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "the path is FP64-issue/latency bound, not HBM bound (DESIGN.md): 7 B per stereo sample; traffic is the committed PMC pass, not this run",
            },
        }
        if workload == "track" and world == 1:
            if not args.no_host_legs:
                legs = host_legs(pcm_host.reshape(-1, CHANNELS))
                if legs is not None and "error" not in legs:
                    result["e2e"] = {"encode_ms": legs["e2e_encode_ms"], "decode_ms": legs["e2e_decode_ms"],
                                     "encode_msps": legs["e2e_encode_msps"], "decode_msps": legs["e2e_decode_msps"],
                                     "what": "sela_hip_encode / sela_hip_decode on page-locked host buffers: H2D + kernels + D2H, steady_clock, median of %d" % legs["repeats"]}
                    result["file_to_file"] = {"encode_ms": legs["file_encode_ms"], "decode_ms": legs["file_decode_ms"],
                                              "encode_msps": legs["file_encode_msps"], "decode_msps": legs["file_decode_msps"],
                                              "equals_e2e_bytes": legs["file_equals_e2e"], "files_on": legs["files_on"],
                                              "what": "sela::encodeFile / decodeFile (the reference's `sela -e` / `-d`, src/main.cpp:29-41): file read, "
                                                      "H2D, kernels, D2H, file write overlapped; in-process, HIP initialised"}
                else:
                    result["e2e"] = result["file_to_file"] = legs
            if not args.no_cpu_baseline:
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
