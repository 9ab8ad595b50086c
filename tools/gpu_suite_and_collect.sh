#!/bin/bash
# tools/gpu_suite_and_collect.sh  (one gpurun call): smoke, the whole GPU suite, then tools/collect_round.sh r06 -- what every refresh of profiles/r06 ran
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06/suite.txt 2>&1; echo "suite rc $?"
grep -n "^E  \|FAILED\|passed\|failed" gpurun_out/r06/suite.txt | head -20
bash tools/collect_round.sh r06 > gpurun_out/r06/collect_stdout.txt 2>&1
tail -60 gpurun_out/r06/collect_stdout.txt | cut -c1-250
