#!/bin/bash
# round 5, late: the any-length route at large batches, kernel by kernel and launch by launch
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r05"; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -k "lpc or refus or golden or hostile" > "$OUT/h_tests.txt" 2>&1; echo "tests rc $?"
grep -n "passed\|failed\|^E " "$OUT/h_tests.txt" | head
timeout 300 python tools/generic_probe.py big > "$OUT/generic_big.txt" 2>&1; echo "probe rc $?"; grep "big:" "$OUT/generic_big.txt"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python "$ROOT/tools/generic_probe.py" big > /tmp/gp.log 2>&1; find /tmp/gp -name "*kernel_trace.csv" -exec cp {} "$OUT/generic_trace.csv" \;)
python - "$OUT/generic_trace.csv" <<'PY' | tee "$OUT/generic_launches.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
groups = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if "generic" not in name and "sela" not in name:
        continue
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    groups.setdefault((name, grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in groups.items():
    print(f"{name:48s} workgroups {grid:7d}  launches {len(d):3d}  min {min(d):10.1f} us  mean {sum(d)/len(d):10.1f} us")
PY
