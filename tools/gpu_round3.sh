#!/bin/bash
# tools/gpu_round3.sh <tag>  (run ON THE GPU BOX through gpurun): smoke, the GPU test suite, the bench line (default and
# with the exchange forced), the file-to-file legs against the number of I/O threads; logs under gpurun_out/<tag>/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee "$OUT/summary.txt"
if ! grep -q "smoke ok" "$OUT/smoke.log"; then
    tail -30 "$OUT/smoke.log"
    echo "smoke failed: skipping the rest" | tee -a "$OUT/summary.txt"
    exit 1
fi
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=15 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -60 "$OUT/pytest.log"
( time timeout 600 python bench.py ) > "$OUT/bench.log" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"
tail -1 "$OUT/bench.log" > "$OUT/bench_line.json"
tail -c 6000 "$OUT/bench.log"
tail -5 "$OUT/bench.err"
# file-to-file against the I/O pool's size
timeout 600 python tools/io_sweep.py > "$OUT/io_threads.txt" 2>&1
cat "$OUT/io_threads.txt"
