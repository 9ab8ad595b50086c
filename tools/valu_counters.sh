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
# The same counters with the bench's default TWO lanes (both streams in flight): counter collection serialises the
# dispatches it instruments, so what this pass shows is (a) that the per-kernel instruction counts do not depend on the
# lanes, and (b) the step time the bench itself reports under instrumentation, next to the un-instrumented one.
rm -rf /tmp/pc_two
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc_two -o pmc -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-extra-legs --lanes 2 > /tmp/pc_two.log 2>&1
python - <<'PY'
import csv, glob, json, collections
fs = glob.glob("/tmp/pc_two/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in (csv.DictReader(open(fs[0])) if fs else []):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sela::" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("two lanes:", k, {c: round(sum(v) / len(v)) for c, v in d.items()})
try:
    line = json.loads(open("/tmp/pc_two.log").read().strip().splitlines()[-1])
    print("two lanes: bench under --pmc reports ms_per_step", round(line["ms_per_step"], 4), "one lane", round(line["lanes"]["ms_per_step_one_lane"], 4))
except Exception as e:  # noqa: BLE001
    print("two lanes: no bench line under --pmc:", e)
PY
