#!/bin/bash
# (one gpurun call) the any-length route's tests, its probe, its kernels' counters and durations at the bench's launch size
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r06"; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_encode_any_length.py tests/test_gpu_decode_any_length.py tests/test_gpu_decode_parity.py tests/test_gpu_boundary.py -x -q -p no:cacheprovider > $OUT/t_any.txt 2>&1; echo "any rc $?"
grep -n "^E \|FAILED\|passed\|failed\|Error" $OUT/t_any.txt | head -40
timeout 300 python tools/generic_probe.py big 2>&1 | grep -v amdgpu.ids | tee "$OUT/generic_route.txt" | cut -c1-330
bash tools/generic_counters.sh 2>&1 | grep -v amdgpu.ids | tail -9
