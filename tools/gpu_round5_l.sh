#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r05"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_host_cpp.py -x -q -m gpu > "$OUT/l_tests.txt" 2>&1; echo "round5+host rc $?"
grep -n "^E \|FAILED\|passed\|failed" "$OUT/l_tests.txt" | head -30
for rep in 1 2 3; do for t in 1 4 16 64; do timeout 120 host/sela_filebench frames $t 16; done; done 2>&1 | tee "$OUT/l_fanout.txt" | cut -c1-150
for t in 1 4 16 64; do timeout 120 host/sela_filebench frames $t 16 fast; done 2>&1 | tee -a "$OUT/l_fanout.txt" | cut -c1-150
