#!/bin/bash
# (one gpurun call) the frame classes' thread loop on the CLI's shape and on odd shapes
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_encode_any_length.py tests/test_host_cpp.py -m gpu -q -p no:cacheprovider > $OUT/t_host.txt 2>&1; echo "rc $?"
grep -n "^E \|FAILED\|passed\|failed" $OUT/t_host.txt | head -30
for shape in "2048 16" "1000 17" "4096 16"; do
  for T in 1 4 16 64; do
    echo -n "shape $shape threads $T: "; timeout 120 host/sela_filebench frames $T 16 $shape 2>&1 | tail -1
  done
done | tee $OUT/frame_classes_fanout.txt
