#!/bin/bash
# tools/gpu_check.sh <tag>   (run ON THE GPU BOX through gpurun): smoke, the GPU test suite, the bench line, the
# batch-size sweep and the phase profile, each under its own timeout, logs under gpurun_out/<tag>/.
TAG=${1:-check}
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
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -40 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench.log" 2>&1
echo "bench rc=$?" | tee -a "$OUT/summary.txt"
tail -1 "$OUT/bench.log" > "$OUT/bench_line.json"
tail -c 3000 "$OUT/bench.log"
timeout 300 python tools/sweep.py > "$OUT/sweep.txt" 2>&1
cat "$OUT/sweep.txt"
timeout 300 python tools/phase_profile.py > "$OUT/phase_cycles.txt" 2>&1
cat "$OUT/phase_cycles.txt"
