#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -s > gpurun_out/r05/generic_tests.txt 2>&1; echo "round5 rc $?"
grep -n "corpus:\|passed\|failed\|^E " gpurun_out/r05/generic_tests.txt | head -20
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round5.py > gpurun_out/r05/gpu_tests.txt 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/gpu_tests.txt | head -40
for v in "" "--lanes 1" "--priorities 0" "--priorities 0 --lanes 1" "--priorities 00010203 --lanes 1"; do
  timeout 600 python bench.py $v > gpurun_out/r05/bench_b.json 2> gpurun_out/r05/bench_b.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r05/bench_b.json').read().strip().splitlines()[-1])
    print("bench.py %-36s %7.0f M  %.4f ms  one lane %7.0f  k %s" % (sys.argv[1], d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], {k:round(v,4) if isinstance(v,float) else v for k,v in d['kernel_ms'].items()}))
except Exception as e:
    print("bench failed", sys.argv[1], e); print(open('gpurun_out/r05/bench_b.err').read()[-2000:])
PY
done
