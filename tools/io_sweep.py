#!/usr/bin/env python3
"""File-to-file legs of host/sela_filebench against the size of the I/O pool (run on the GPU box; not a bench line)."""
import json
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sela_amd.synth import synth_frames  # noqa: E402


def main():
    threads = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
    pcm = synth_frames(3875, 2, 0).reshape(-1, 2)
    d = "/dev/shm/sela_io"
    os.makedirs(d, exist_ok=True)
    data = pcm.astype("<i2").tobytes()
    with open(d + "/track.wav", "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IhHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16)
                + b"data" + struct.pack("<I", len(data)) + data)
    print("io threads | file encode ms | file decode ms | e2e encode ms | e2e decode ms | equal")
    for n in threads:
        r = subprocess.run([os.path.join(ROOT, "host", "sela_filebench"), d + "/track.wav", d, "9", "all", str(n)], capture_output=True, text=True, timeout=300)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print(n, "|", j["file_encode_ms"], "|", j["file_decode_ms"], "|", j["e2e_encode_ms"], "|", j["e2e_decode_ms"], "|", j["file_equals_e2e"], flush=True)
        except Exception as e:  # noqa: BLE001
            print(n, "| error", r.stderr[-300:], e, flush=True)


if __name__ == "__main__":
    main()
