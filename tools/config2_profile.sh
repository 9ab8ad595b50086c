#!/bin/bash
# tools/config2_profile.sh  (run ON THE GPU BOX through gpurun): BASELINE.json configs[2] -- a batch of 1000 stereo frames on
# one GPU -- under rocprofv3: per-kernel durations (--kernel-trace --stats) and HBM bytes (PMC FETCH_SIZE / WRITE_SIZE, each in
# its own run, kernel trace only), as GB/s per kernel and for the batch.  Writes gpurun_out/config2_1000_frames.txt.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SELA_SWEEP_HOST=0   # device-pointer path only: the batch is resident in HBM
CMD="python $ROOT/tools/sweep.py 1000"
rm -rf /tmp/c2_stats /tmp/c2_FETCH_SIZE /tmp/c2_WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2_stats -o stats -- $CMD > "$OUT/config2_stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/c2_$C -o pmc -- $CMD > "$OUT/config2_pmc_$C.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for path in glob.glob("/tmp/c2_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "sela::" in r["Name"]:
            dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]))
pmc = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"/tmp/c2_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") == c and "sela::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k][c] = sum(v) / len(v) * 1024  # KiB -> bytes, per launch
lines = ["BASELINE.json configs[2]: 1000 synthetic 16-bit stereo frames (2,048,000 stereo samples, 8.19 MB of PCM), one MI355X;",
         "`python tools/sweep.py 1000` under rocprofv3 (--kernel-trace --stats; --pmc FETCH_SIZE and --pmc WRITE_SIZE in runs of their own).",
         "HBM bytes per launch: raw = FETCH + WRITE, x2 = 2 x FETCH + WRITE (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads on gfx950).",
         "", "kernel | launches | avg us | FETCH MB | WRITE MB | GB/s raw | GB/s x2 | % of 8 TB/s (x2)"]
tot_t = tot_raw = tot_x2 = 0.0
enc_t = 0.0
for k in sorted(dur):
    calls, ns = dur[k]
    f, w = pmc.get(k, {}).get("FETCH_SIZE", 0.0), pmc.get(k, {}).get("WRITE_SIZE", 0.0)
    raw, x2 = f + w, 2 * f + w
    lines.append(f"{k} | {calls} | {ns / 1e3:.1f} | {f / 1e6:.2f} | {w / 1e6:.2f} | {raw / ns:.1f} | {x2 / ns:.1f} | {100 * x2 / ns / 8000:.2f}")
    tot_t += ns
    tot_raw += raw
    tot_x2 += x2
    if "decode" not in k:
        enc_t += ns
samples = 1000 * 2048
lines += ["", f"encode (blocks + plan + assemble): {enc_t / 1e3:.1f} us of kernels = {samples / enc_t:.2f} G stereo samples/s; "
              f"encode + decode: {tot_t / 1e3:.1f} us = {samples / tot_t:.2f} G samples/s",
          f"HBM traffic of the four kernels together: {tot_raw / 1e6:.1f} MB raw / {tot_x2 / 1e6:.1f} MB x2 per batch = "
          f"{tot_raw / tot_t:.1f} / {tot_x2 / tot_t:.1f} GB/s = {100 * tot_x2 / tot_t / 8000:.2f} % of the 8 TB/s peak",
          "(algorithmic: 8.19 MB PCM + ~5.9 MB of frames, in and out once each way per direction = ~28 MB per encode + decode)"]
open(f"{out}/config2_1000_frames.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
