for v in base V base V; do
  if [ $v = base ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$PWD/tools/tmp/v$v; fi
  echo "== $v: $(E2E_REPS=15 E2E_TIMEOUT=40 SELA_FILEBENCH_ENCODE_ONLY=1 timeout 80 python tools/e2e_trace.py --no-trace 2>&1 | tail -1)"
done
unset LD_LIBRARY_PATH
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
