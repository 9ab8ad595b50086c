#!/bin/bash
# (one gpurun call) smoke, the whole GPU suite, then the round's profile collection
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06/suite.txt 2>&1; echo "suite rc $?"
grep -n "^E  \|FAILED\|passed\|failed" gpurun_out/r06/suite.txt | head -20
bash tools/collect_round.sh r06 > gpurun_out/r06/collect_stdout.txt 2>&1
tail -60 gpurun_out/r06/collect_stdout.txt | cut -c1-250
