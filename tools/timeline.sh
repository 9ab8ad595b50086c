#!/bin/bash
# tools/timeline.sh  (run ON THE GPU BOX): kernel start/end times of the LAST bench step, relative to
# its encode kernel's start, from a rocprofv3 kernel trace (encode: three kernels; decode: a memset and one kernel).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-extra-legs --lanes ${LANES:-1} > /tmp/prof_tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sela::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# bench's timed steps come before its per-kernel timing pass; take the 6th encode from the end of the timed run
enc = [i for i, r in enumerate(rows) if "k_encode_blocks" in r["Kernel_Name"] or "k_encode_teams" in r["Kernel_Name"]]
i0 = enc[5]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 8]:
    n = r["Kernel_Name"].split("(")[0].replace("void sela::", "")[:32]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{n:34s} start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  stream {r.get('Stream_Id', r.get('Queue_Id', '?'))}")
PY
