#!/bin/bash
# the whole GPU suite (round-5 tests first, verbosely on failure)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r05/generic_tests.txt 2>&1; echo "generic rc $?"
tail -40 gpurun_out/r05/generic_tests.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05/gpu_tests.txt 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/gpu_tests.txt | head -40
