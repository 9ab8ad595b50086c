#!/bin/bash
# tools/collect_profiles.sh <round-tag>   (run ON THE GPU BOX, e.g. through gpurun)
#
# Collects the rocprofv3 evidence that DESIGN.md / bench.py quote, into gpurun_out/profiles_<tag>/:
#   1. --kernel-trace --stats of the default bench command        -> kernel_stats.csv
#   2. PMC pass FETCH_SIZE (own run, kernel-trace only)            -> pmc_fetch.csv
#   3. PMC pass WRITE_SIZE (own run, kernel-trace only)            -> pmc_write.csv
# and a traffic.json with per-launch HBM bytes of the dominant kernels (see MI355X_MICROARCH.md
# "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads
# by 2x, so read bytes are given raw and x2).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-legs --no-extra-legs --lanes 1"   # one lane: kernels back to back, so a duration is the kernel's own

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o stats -- $CMD > "$OUT/bench_under_rocprof.log" 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o pmc -- $CMD > "$OUT/bench_under_pmc_$C.log" 2>&1
    find /tmp/prof_$C -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_$C.csv" \;
done
python - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    try:
        with open(f"{out}/pmc_{c}.csv") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == c:
                    acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    except FileNotFoundError:
        continue
    for k, v in acc.items():
        if "sela::" in k:
            res.setdefault(k, {})[c + "_KiB_per_launch"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
for k, d in res.items():
    f, w = d.get("FETCH_SIZE_KiB_per_launch"), d.get("WRITE_SIZE_KiB_per_launch")
    if f is not None and w is not None:
        d["hbm_bytes_per_launch_raw"] = (f + w) * 1024
        d["hbm_bytes_per_launch_fetch_x2"] = (2 * f + w) * 1024
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
ls -la "$OUT"
