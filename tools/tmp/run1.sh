mkdir -p gpurun_out
E2E_REPS=5 E2E_TIMEOUT=20 timeout 80 python tools/e2e_trace.py --no-trace > gpurun_out/mf_e2e.txt 2>&1
cat gpurun_out/mf_e2e.txt | tail -3
if grep -q e2e_encode_ms gpurun_out/mf_e2e.txt; then
  timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/mf_pytest.txt
  cat gpurun_out/mf_pytest.txt
  E2E_REPS=15 E2E_TIMEOUT=30 timeout 60 python tools/e2e_trace.py --no-trace 2>&1 | tail -1 | tee gpurun_out/mf_e2e2.txt
fi
