#!/bin/bash
# tools/valu_counters.sh  (run ON THE GPU BOX): SQ instruction-issue counters of the bench kernels, one
# rocprofv3 --pmc pass (kernel trace only).  Prints per-kernel averages; quoted in DESIGN.md.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-extra-legs --lanes 1"
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/pc_$TAG
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pc_$TAG -o pmc -- $CMD > /tmp/pc_$TAG.log 2>&1
  python - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
fs = glob.glob(f"/tmp/pc_{tag}/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file for", tag); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sela::" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
