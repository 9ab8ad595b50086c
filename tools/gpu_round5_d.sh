#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05/gpu_tests.txt 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/gpu_tests.txt | head -40
timeout 600 python bench.py > gpurun_out/r05/bench_d.json 2> gpurun_out/r05/bench_d.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench_d.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','kernel_ms','lanes','any_length','e2e','file_to_file','decode10k') if k in d})
except Exception as e:
    print("bench failed", e); print(open('gpurun_out/r05/bench_d.err').read()[-3000:])
PY
