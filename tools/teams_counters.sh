#!/bin/bash
# tools/teams_counters.sh [n_frames]  (run ON THE GPU BOX): SQ counters of the block kernels, one rocprofv3 --pmc pass per
# counter set (kernel trace only), for k_encode_blocks (0) and k_encode_teams<16 / 8>.
N=${1:-10000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for TEAMS in 0 16 8; do
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  TAG=t${TEAMS}_$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/pc_$TAG
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pc_$TAG -o pmc -- python $ROOT/tools/teams_run.py $N $TEAMS 4 > /tmp/pc_$TAG.log 2>&1
  python - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
fs = glob.glob(f"/tmp/pc_{tag}/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file for", tag); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_encode_" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
done
