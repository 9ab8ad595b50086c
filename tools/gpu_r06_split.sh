#!/bin/bash
# (one gpurun call) the cut launch: the encode parity file (its test among them)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_encode_parity.py tests/test_gpu_multi_process.py -q -p no:cacheprovider > $OUT/t_split.txt 2>&1; echo "rc $?"
grep -n "^E \|FAILED\|passed\|failed\|Error" $OUT/t_split.txt | head -20
