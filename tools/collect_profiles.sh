#!/bin/bash
# tools/collect_profiles.sh <round-tag>   (run ON THE GPU BOX, e.g. through gpurun)
#
# Collects the rocprofv3 evidence that DESIGN.md / bench.py quote, into gpurun_out/profiles_<tag>/:
#   1. --kernel-trace --stats of the default bench command        -> kernel_stats.csv
#   2. PMC pass FETCH_SIZE (own run, kernel-trace only)            -> pmc_fetch.csv
#   3. PMC pass WRITE_SIZE (own run, kernel-trace only)            -> pmc_write.csv
#   4. tools/traffic_calib.sh (known-byte kernels at 4 / 8 / 16 B per lane)  -> traffic_calibration.json
# and a traffic.json with per-launch HBM bytes of the dominant kernels (see MI355X_MICROARCH.md
# "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads
# by 2x, so read bytes are given raw, x2, and scaled by the factors the calibration run measured), and
# sources.sha256 = the digest of the kernel sources these numbers belong to (bench.py: profiles_stale).
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
bash "$ROOT/tools/traffic_calib.sh" "$OUT" > "$OUT/calib.log" 2>&1
(cd "$ROOT" && python -c "import bench; print(bench.kernel_sources_sha256(), ' sela_amd/csrc/* include/*')") > "$OUT/sources.sha256"
python - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
try:
    calib = json.load(open(f"{out}/traffic_calibration.json"))
    reads = [v["factor"] for k, v in calib.items() if k.startswith("read_") and v.get("factor")]
    writes = [v["factor"] for k, v in calib.items() if k.startswith("write_") and v.get("factor")]
    # (one factor per direction if the widths agree to 2 %: they do on gfx950, 2.0 and 1.0)
    f_read = sum(reads) / len(reads) if reads and max(reads) - min(reads) < 0.02 * max(reads) else None
    f_write = sum(writes) / len(writes) if writes and max(writes) - min(writes) < 0.02 * max(writes) else None
except (OSError, ValueError):
    calib, f_read, f_write = None, None, None
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
        if f_read and f_write:
            d["hbm_bytes_per_launch_calibrated"] = (f_read * f + f_write * w) * 1024
if f_read and f_write:
    res["_calibration"] = {"file": "traffic_calibration.json", "fetch_factor": f_read, "write_factor": f_write,
                           "what": "true bytes / counter bytes of known-byte streaming kernels at 4, 8 and 16 B per lane (tools/traffic_calib.hip); the widths agree"}
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
ls -la "$OUT"
