#!/bin/bash
# tools/generic_counters.sh  (run ON THE GPU BOX): SQ instruction-issue counters and durations of the any-length route's kernels
# at the bench's launch size (3875 stereo frames of 2048 samples), rocprofv3 --pmc passes with the kernel trace only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/generic_probe.py counters"
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/gc_$TAG
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/gc_$TAG -o pmc -- $CMD > /tmp/gc_$TAG.log 2>&1
  python - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
fs = glob.glob(f"/tmp/gc_{tag}/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file for", tag); sys.exit(0)
acc = collections.OrderedDict()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sela::" in k:
        acc.setdefault((k, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (k, d), c in acc.items():
    print(f"{k:40s}", {n: round(v) for n, v in c.items()})
PY
done
rm -rf /tmp/gc_t
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gc_t -o t -- $CMD > /tmp/gc_t.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/gc_t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sela::" in k:
            print(f"{k:40s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}")
PY
