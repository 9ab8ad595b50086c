#!/bin/bash
# first GPU call of round 5: the new any-length tests, then the whole GPU suite, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r05/generic_tests.txt 2>&1; echo "generic rc $?"
tail -30 gpurun_out/r05/generic_tests.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r05/gpu_tests.txt 2>&1; echo "suite rc $?"
tail -5 gpurun_out/r05/gpu_tests.txt
timeout 600 python bench.py > gpurun_out/r05/bench0.json 2> gpurun_out/r05/bench0.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench0.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','roofline','kernel_ms','lanes') if k in d})
PY
