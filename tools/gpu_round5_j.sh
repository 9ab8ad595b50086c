#!/bin/bash
# round 5, late: the coalescer's first wait (variants under tools/tmp/vNN), the round-5 tests on the final kernels
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r05"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > "$OUT/j_tests.txt" 2>&1; echo "round5 rc $?"
grep -n "passed\|failed\|^E " "$OUT/j_tests.txt" | head -20
: > "$OUT/j_fanout.txt"
for rep in 1 2; do
for v in default v60 v120; do
  for mode in "" fast; do
    for t in 4 16 64; do
      if [ "$v" = default ]; then line=$(timeout 120 host/sela_filebench frames $t 16 $mode); else line=$(LD_LIBRARY_PATH="$ROOT/tools/tmp/$v" timeout 120 host/sela_filebench frames $t 16 $mode); fi
      echo "$v ${mode:-exact} $line" | tee -a "$OUT/j_fanout.txt" | cut -c1-160
    done
  done
done
done
