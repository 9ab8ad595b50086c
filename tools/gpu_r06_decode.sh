#!/bin/bash
# (one gpurun call) the GPU suite, then the any-length route's probe alone and under rocprofv3, launch by launch
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r06"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/suite.txt" 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" "$OUT/suite.txt" | head -30
timeout 300 python tools/generic_probe.py big 2>&1 | grep -v amdgpu.ids | tee "$OUT/generic_route.txt" | cut -c1-260
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python "$ROOT/tools/generic_probe.py" big > /tmp/gp.log 2>&1; find /tmp/gp -name "*kernel_trace.csv" -exec cp {} /tmp/gp_trace.csv \;)
python - /tmp/gp_trace.csv <<'PY' | tee "$OUT/generic_launches.txt"
import collections, csv, sys
groups = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if "sela" not in name:
        continue
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    groups.setdefault((name, grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in groups.items():
    print(f"{name:48s} workgroups {grid:7d}  launches {len(d):3d}  min {min(d):10.1f} us  mean {sum(d)/len(d):10.1f} us")
PY
