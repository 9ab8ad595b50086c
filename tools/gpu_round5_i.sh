#!/bin/bash
# round 5, late: k_decode_subframes32 (standard subframes to 32-bit samples) -- tests, the frame classes' fan-out, the probe
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r05"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > "$OUT/i_tests.txt" 2>&1; echo "round5 rc $?"
grep -n "passed\|failed\|^E " "$OUT/i_tests.txt" | head -20
timeout 600 python -m pytest tests/test_host_cpp.py tests/test_gpu_round4.py -x -q -m gpu > "$OUT/i_tests2.txt" 2>&1; echo "host/round4 rc $?"
grep -n "passed\|failed\|^E " "$OUT/i_tests2.txt" | head -20
for t in 1 4 16 64; do timeout 120 host/sela_filebench frames $t 16; done 2>&1 | tee "$OUT/i_fanout.txt" | cut -c1-150
for t in 1 4 16 64; do timeout 120 host/sela_filebench frames $t 16 fast; done 2>&1 | tee -a "$OUT/i_fanout.txt" | cut -c1-150
timeout 300 python tools/generic_probe.py big > "$OUT/generic_big.txt" 2>&1; echo "probe rc $?"; cat "$OUT/generic_big.txt" | cut -c1-250
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python "$ROOT/tools/generic_probe.py" big > /tmp/gp.log 2>&1; find /tmp/gp -name "*kernel_trace.csv" -exec cp {} "$OUT/generic_trace.csv" \;)
python - "$OUT/generic_trace.csv" <<'PY' | tee "$OUT/generic_launches.txt" | grep "decode\|combine"
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
