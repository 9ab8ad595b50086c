#!/usr/bin/env python3
"""Run the CLI's verbs over and over with a short limit each and say which invocation, if any, did not come back:
    python tools/stress_cli.py [rounds [seconds per call]]
(the hunt for a test run that once held the GPU box for 23 minutes; not part of the tests)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sela_amd.synth import synth_pcm  # noqa: E402
from test_host_cpp import _write_wav  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
cli = os.path.join(ROOT, "host", "sela_mi355x")
work = tempfile.mkdtemp(prefix="sela_stress_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
specs = [("a", 2, 44100, 9 * 2048 + 777), ("b", 2, 48000, 4 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100),
         ("e", 2, 44100, 300 * 2048), ("f", 1, 44100, 5 * 2048), ("g", 2, 44100, 2048), ("h", 2, 44100, 37 * 2048 + 1), ("i", 2, 8000, 1500 * 2048)]
wavs = []
for k, (name, ch, rate, n) in enumerate(specs):
    p = os.path.join(work, name + ".wav")
    _write_wav(p, synth_pcm(n, ch, 120 + k), rate)
    wavs.append(p)
hung = 0
t_all = time.time()


def call(tag, args):
    global hung
    t0 = time.time()
    try:
        r = subprocess.run([cli] + args, capture_output=True, text=True, timeout=limit)
        if r.returncode != 0:
            print(f"{tag}: rc {r.returncode}: {r.stderr[-300:]}", flush=True)
    except subprocess.TimeoutExpired:
        hung += 1
        print(f"{tag}: NO RETURN after {limit:.0f} s: {' '.join(args)[:200]}", flush=True)
    return time.time() - t0


slow = []
for r in range(rounds):
    enc, dec = os.path.join(work, f"enc{r}"), os.path.join(work, f"dec{r}")
    os.makedirs(enc), os.makedirs(dec)
    devs = ["0", "0,0", "0,0,0", "0,0,0,0,0,0,0,0"][r % 4]
    slow.append((call(f"round {r} -E {devs}", ["-E", enc, "--devices", devs] + (["--io-threads", "5"] if r % 2 else []) + wavs), "E"))
    selas = [os.path.join(enc, os.path.basename(w)[:-4] + ".sela") for w in wavs]
    slow.append((call(f"round {r} -D {devs}", ["-D", dec, "--devices", devs] + selas), "D"))
    slow.append((call(f"round {r} -e", ["-e", wavs[8], os.path.join(work, "one.sela")]), "e"))
    slow.append((call(f"round {r} -d", ["-d", os.path.join(work, "one.sela"), os.path.join(work, "one.wav")]), "d"))
    slow.append((call(f"round {r} -p", ["-p", os.path.join(work, "one.sela"), os.path.join(work, "one.raw")]), "p"))
print(f"{rounds} rounds, {len(slow)} calls in {time.time() - t_all:.0f} s; {hung} did not return; slowest: " + ", ".join(f"{v}: {t:.2f} s" for t, v in sorted(slow, reverse=True)[:5]))
