#!/bin/bash
# tools/album_profile.sh  (run ON THE GPU BOX): BASELINE configs[3] -- the 100-track album, batches of 61,041 frames, ONE lane --
# under rocprofv3: kernel durations (--kernel-trace --stats) and HBM bytes per launch (--pmc FETCH_SIZE / WRITE_SIZE, runs
# of their own; FETCH x 2.0, WRITE x 1.0: tools/traffic_calib.hip).  Prints a table; quoted in profiles/rNN/album_kernels.txt.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload album --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --no-extra-legs --lanes 1"
rm -rf /tmp/prof_alb_*
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_alb_stats -o stats -- $CMD > /tmp/prof_alb_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_alb_$C -o pmc -- $CMD > /tmp/prof_alb_$C.log 2>&1
done
python - <<'PY'
import csv, glob, collections
stats = {}
for f in glob.glob("/tmp/prof_alb_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sela::" in r["Name"]:
            stats[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/prof_alb_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c and "sela::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        big = [x for x in v if x > 0.5 * max(v)]  # (the album's last batch is shorter, the warm-up batches are the same size)
        pmc.setdefault(k, {})[c] = sum(big) / len(big) * 1024
print("BASELINE.json configs[3]: 549,365 frames in batches of 61,041 (the last one 60,037), one lane; per launch of the full batches")
print("kernel | launches | avg us | FETCH MB (x2) | WRITE MB | MB per launch | x algorithmic (PCM 500.0 MB + frames ~360 MB per batch)")
for k, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    p = pmc.get(k, {})
    fe, wr = 2.0 * p.get("FETCH_SIZE", 0) / 1e6, p.get("WRITE_SIZE", 0) / 1e6
    print(f"{k} | {calls} | {us:.1f} | {fe:.1f} | {wr:.1f} | {fe + wr:.1f} |")
PY
tail -1 /tmp/prof_alb_stats.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('bench line of the profiled run:', d['value'], 'M samples/s', d['ms_per_step'], 'ms per step', d['kernel_ms'])
except Exception as e: print('no bench line', e)"
