#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_round3.py -q -k "bench_line" > gpurun_out/r05/gpu_tests_e.txt 2>&1; echo "test rc $?"
grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/gpu_tests_e.txt | head -20
timeout 600 python bench.py > gpurun_out/r05/bench_d.json 2> gpurun_out/r05/bench_d.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench_d.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','kernel_ms','lanes','any_length','e2e','file_to_file') if k in d})
    print({k:d['decode10k'][k] for k in ('value','ms_per_step')}, d['decode10k'].get('per_rank_share_8'))
except Exception as e:
    print("bench failed", e); print(open('gpurun_out/r05/bench_d.err').read()[-3000:])
PY
host/sela_filebench frames 16 16 2>&1 | tail -5
