#!/usr/bin/env python3
"""tools/e2e_trace.py (run ON THE GPU BOX): copies and kernels of the LAST host-pointer encode and decode call of
host/sela_filebench, on one time axis (rocprofv3 --kernel-trace --memory-copy-trace).  Prints the untraced
e2e numbers first.  Environment knobs of the pipeline (SELA_HOST_*) are passed through."""
import csv
import glob
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_track(path):
    import torch
    from sela_amd import synth

    pcm = synth.synth_frames_torch(3875, 2, 1, 0, torch.device("cuda")).cpu().numpy()
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt "
                + struct.pack("<IhHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) + b"data" + struct.pack("<I", len(data)) + data)


def main():
    exe = os.path.join(ROOT, "host", "sela_filebench")
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        wav = os.path.join(tmp, "track.wav")
        write_track(wav)
        try:
            out = subprocess.run([exe, wav, tmp, os.environ.get("E2E_REPS", "15"), "e2e"], capture_output=True, text=True,
                                 timeout=float(os.environ.get("E2E_TIMEOUT", "120")))
            print("untraced:", out.stdout.strip(), out.stderr.strip()[-400:])
        except subprocess.TimeoutExpired as t:
            print("untraced: TIMEOUT", (t.stdout or b"")[-400:], (t.stderr or b"")[-400:])
            return
        if "--no-trace" in sys.argv:
            return
        prof = os.path.join(tmp, "prof")
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", prof, "-o", "t", "--",
                        exe, wav, tmp, "2", "e2e"], capture_output=True, text=True, timeout=300, cwd="/tmp", env=env)
        kern = glob.glob(prof + "/**/*kernel_trace.csv", recursive=True)
        copy = glob.glob(prof + "/**/*memory_copy_trace.csv", recursive=True)
        if not kern:
            print("no trace files", os.listdir(prof) if os.path.isdir(prof) else "")
            return
        ev = []
        for r in csv.DictReader(open(kern[0])):
            name = r["Kernel_Name"].split("(")[0].replace("void sela::", "")[:28]
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, ""))
        for r in (csv.DictReader(open(copy[0])) if copy else []):  # (no copy trace: the call made no copies)
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""),
                       r.get("Bytes", r.get("Size", "?"))))
        ev.sort()
        # which call an operation belongs to: copies in and fills go with the next codec kernel, copies out with the last one
        def kind_of(name):
            if "k_decode" in name:
                return "decode"
            if "k_encode" in name or "k_plan" in name or "k_assemble" in name:
                return "encode"
            return None
        kinds = [kind_of(e[2]) for e in ev]
        nxt, prv = [None] * len(ev), [None] * len(ev)
        k = None
        for i in range(len(ev) - 1, -1, -1):
            k = kinds[i] or k
            nxt[i] = k
        k = None
        for i in range(len(ev)):
            k = kinds[i] or k
            prv[i] = k
        call = [kinds[i] or (prv[i] if ("DEVICE_TO_HOST" in ev[i][2] or "copyBuffer" in ev[i][2]) else nxt[i]) or prv[i] for i in range(len(ev))]
        groups = []
        for i, e in enumerate(ev):
            if not groups or groups[-1][0] != call[i]:
                groups.append((call[i], []))
            groups[-1][1].append(e)
        shown = set()
        for what, g in reversed(groups):
            if what in shown or what is None:
                continue
            shown.add(what)
            t0 = g[0][0]
            print(f"---- last {what} call: {len(g)} operations, {(max(e[1] for e in g) - t0) / 1e3:.1f} us from first start to last end")
            for s, e, n, b in g:
                print(f"{n:30s} {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:7.1f})  {b}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
