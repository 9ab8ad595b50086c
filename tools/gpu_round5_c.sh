#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "hash or forms or corpus" > gpurun_out/r05/hash_tests.txt 2>&1; echo "hash tests rc $?"
grep -n "passed\|failed\|^E " gpurun_out/r05/hash_tests.txt | head -20
timeout 300 tools/tmp/chain_ubench_bin > gpurun_out/r05/chain_ubench.txt 2>&1; echo "ubench rc $?"
cat gpurun_out/r05/chain_ubench.txt
for v in "" "--encode-teams 16"; do
  timeout 600 python bench.py $v > gpurun_out/r05/bench_c.json 2> gpurun_out/r05/bench_c.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r05/bench_c.json').read().strip().splitlines()[-1])
    print("bench.py %-24s %7.0f M  album %s" % (sys.argv[1], d['value'], json.dumps(d.get('album'))[:900]))
except Exception as e:
    print("bench failed", sys.argv[1], e); print(open('gpurun_out/r05/bench_c.err').read()[-2000:])
PY
done
