#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_host_cpp.py -x -q -m gpu > gpurun_out/r05/generic_tests.txt 2>&1; echo "round5+host rc $?"
grep -n "passed\|failed\|^E " gpurun_out/r05/generic_tests.txt | head -20
for T in 1 4 16 64; do host/sela_filebench frames $T 16 2>&1 | tail -1; done
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, bench
print(bench.any_length_leg(np))
PY
