#!/usr/bin/env python3
"""Rate of the batch verbs (sela::encodeFiles / decodeFiles: every GPU worker reading, coding and writing its own pieces) on
album tracks in tmpfs, in-process (host/sela_filebench batch: HIP initialised, buffers pinned by an untimed first pass).
Run on the GPU box; not a bench line (DESIGN.md section 7 quotes it)."""
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sela_amd.synth import album_tracks, synth_frames_torch  # noqa: E402


def write_wav(path, pcm, rate):
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IhHIIHH", 16, 1, 2, rate, rate * 4, 4, 16)
                + b"data" + struct.pack("<I", len(data)) + data)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    work = tempfile.mkdtemp(dir="/dev/shm")
    exe = os.path.join(ROOT, "host", "sela_filebench")
    try:
        tracks = album_tracks()[:n]
        wavs, frames = [], 0
        for track, rate, nf in tracks:
            p = os.path.join(work, f"t{track:03d}.wav")
            write_wav(p, synth_frames_torch(nf, 2, track, device="cuda").cpu().numpy().reshape(-1, 2), rate)
            wavs.append(p)
            frames += nf
        print(f"{n} album tracks, {frames} frames, in tmpfs; sela::encodeFiles / decodeFiles in-process (host/sela_filebench batch), median of 5")
        print("workers (all on GPU 0) | encode ms | G samples/s | GB/s in | GB/s out | decode ms | G samples/s | GB/s in | GB/s out")
        for devices in ("0", "0,0", "0,0,0,0"):
            r = subprocess.run([exe, "batch", work, devices, "5"] + wavs, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            j = json.loads(r.stdout.strip().splitlines()[-1])
            e, d = j["encode_ms"] / 1e3, j["decode_ms"] / 1e3
            print(f"{j['workers']} | {j['encode_ms']:.1f} | {frames * 2048 / e / 1e9:.2f} | {j['wav_bytes'] / e / 1e9:.2f} | {j['sela_bytes'] / e / 1e9:.2f} | "
                  f"{j['decode_ms']:.1f} | {frames * 2048 / d / 1e9:.2f} | {j['sela_bytes'] / d / 1e9:.2f} | {j['wav_bytes'] / d / 1e9:.2f}", flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
