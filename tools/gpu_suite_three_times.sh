#!/bin/bash
# (one gpurun call) the GPU suite three times over -- flaky tests show -- then smoke() and bench.py the way the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05/flake_$i.txt 2>&1; echo "run $i rc $?"
  grep -n "^E \|FAILED\|passed\|failed" gpurun_out/r05/flake_$i.txt | head -10
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_like_bench.json 2> gpurun_out/r05/driver_like_bench.err; echo "bench rc $?"
python -c "
import json; d=json.loads(open('gpurun_out/r05/driver_like_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['profiles_stale'], d['roofline']['frac'], d['cpu_baseline']['value'], len(json.dumps(d)))"
