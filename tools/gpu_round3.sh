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
# file-to-file against the I/O pool's size (host/sela_filebench <wav> <dir> <repeats> all <threads>)
python - "$OUT" <<'PY'
import os, struct, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from sela_amd.synth import synth_frames
out = sys.argv[1]
pcm = synth_frames(3875, 2, 0).reshape(-1, 2)
d = "/dev/shm/sela_io"
os.makedirs(d, exist_ok=True)
data = pcm.astype("<i2").tobytes()
with open(d + "/track.wav", "wb") as f:
    f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IhHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) + b"data" + struct.pack("<I", len(data)) + data)
rows = []
for threads in (1, 2, 4, 6, 8, 12, 16, 24, 32):
    r = subprocess.run(["host/sela_filebench", d + "/track.wav", d, "9", "all", str(threads)], capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append((threads, j["file_encode_ms"], j["file_decode_ms"], j["e2e_encode_ms"], j["e2e_decode_ms"], j["file_equals_e2e"]))
    except Exception as e:
        rows.append((threads, "error", r.stderr[-200:], str(e), "", ""))
with open(out + "/io_threads.txt", "w") as f:
    f.write("io threads | file encode ms | file decode ms | e2e encode ms | e2e decode ms | equal\n")
    for row in rows:
        f.write(" | ".join(str(x) for x in row) + "\n")
print(open(out + "/io_threads.txt").read())
PY
