#!/bin/bash
# (one gpurun call) the whole GPU suite, then tools/collect_r05.sh: what every profiles/r05 refresh of round 5 ran
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05/final_tests.txt 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/final_tests.txt | head -30
bash tools/collect_r05.sh 2>&1 | tail -120
