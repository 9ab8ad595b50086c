#!/bin/bash
# tools/lds_counters.sh  (run ON THE GPU BOX): LDS bank-conflict, wait and issue counters of the headline kernels (bench.py, two steps),
# rocprofv3 --pmc passes with the kernel trace only; per-launch averages.  Round 6: found the 2-way conflict of k_encode_teams<0,16>.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-host-legs"
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  TAG=$(echo $SET | cut -d' ' -f1); rm -rf /tmp/l_$TAG
  timeout 400 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/l_$TAG -o pmc -- $CMD > /tmp/l_$TAG.log 2>&1
  tail -2 /tmp/l_$TAG.log | cut -c1-200
  python - $TAG <<'PY'
import csv, glob, sys, collections
fs = glob.glob(f"/tmp/l_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
if not fs: print("no file", sys.argv[1]); sys.exit(0)
acc = collections.OrderedDict(); cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sela::" in k:
        d = acc.setdefault(k, collections.Counter()); d[r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == list(d)[0]: cnt[k] += 1
for k, c in acc.items(): print(f"{k:40s} x{cnt[k]:4d}", {n: round(v / cnt[k]) for n, v in c.items()})
PY
done
