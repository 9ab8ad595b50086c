#!/bin/bash
# round 5, late: the whole GPU suite on the final sources, then the frame classes' fan-out
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r05"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/k_tests.txt" 2>&1; echo "suite rc $?"
grep -n "^E \|FAILED\|passed\|failed" "$OUT/k_tests.txt" | head -30
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for t in 1 4 16 64; do timeout 120 host/sela_filebench frames $t 16; done 2>&1 | tee "$OUT/k_fanout.txt" | cut -c1-150
