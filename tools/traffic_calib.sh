#!/bin/bash
# tools/traffic_calib.sh <out-dir>   (run ON THE GPU BOX): FETCH_SIZE / WRITE_SIZE of tools/traffic_calib's known-byte kernels, one
# rocprofv3 --pmc pass per counter (kernel trace only) -> <out-dir>/traffic_calibration.json with
#   factor = true bytes / (counter x 1024)   per access width (b32 / b64 / b128), for reads and for writes.
OUT=${1:-gpurun_out/calib}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac
mkdir -p "$OUT"
MIB=${2:-1024}
cd /tmp && export TMPDIR=/tmp
[ -x "$ROOT/tools/traffic_calib" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$ROOT/tools/traffic_calib" "$ROOT/tools/traffic_calib.hip"
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/calib_$C
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/calib_$C -o pmc -- "$ROOT/tools/traffic_calib" $MIB > /tmp/calib_$C.log 2>&1
    find /tmp/calib_$C -name "*counter_collection.csv" -exec cp {} "$OUT/calib_pmc_$C.csv" \;
done
python - "$OUT" $MIB <<'PY'
import csv, json, sys, collections
out, mib = sys.argv[1], int(sys.argv[2])
true_bytes = mib << 20
res = {"bytes_per_kernel": true_bytes, "what": "tools/traffic_calib.hip under rocprofv3 --pmc; factor = true bytes / (counter x 1024)"}
for c, kind in (("FETCH_SIZE", "k_read"), ("WRITE_SIZE", "k_write")):
    acc = collections.defaultdict(list)
    try:
        for row in csv.DictReader(open(f"{out}/calib_pmc_{c}.csv")):
            if row.get("Counter_Name") == c and kind in row["Kernel_Name"]:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    except FileNotFoundError:
        continue
    for k, v in acc.items():
        width = "b128" if ("uint4" in k or "HIP_vector_type<unsigned int, 4" in k) else ("b64" if ("uint2" in k or "HIP_vector_type<unsigned int, 2" in k) else "b32")
        mean = sum(v) / len(v)
        res[f"{'read' if kind == 'k_read' else 'write'}_{width}"] = {"counter_KiB": mean, "factor": true_bytes / (mean * 1024) if mean else None, "launches": len(v)}
json.dump(res, open(f"{out}/traffic_calibration.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
