#!/bin/bash
# tools/collect_all.sh <round-tag>  (run ON THE GPU BOX through gpurun): everything profiles/<tag>/ holds.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1
bash tools/valu_counters.sh > "$OUT/valu_counters.txt" 2>&1
bash tools/timeline.sh > "$OUT/timeline.txt" 2>&1
LANES=2 bash tools/timeline.sh > "$OUT/timeline_two_lanes.txt" 2>&1
python tools/phase_profile.py > "$OUT/phase_cycles.txt" 2>&1
python tools/phase_profile.py 1000 > "$OUT/phase_cycles_1000_frames.txt" 2>&1
SELA_SWEEP_HOST=0 python tools/sweep.py > "$OUT/sweep.txt" 2>&1
# the decoder's two forms of the recurrence step against the batch size, and the instruction-level chain walk behind them
python tools/decode_forms.py 2>&1 | grep -v amdgpu.ids > "$OUT/decode_forms.txt"
python tools/chain_ubench.py >> "$OUT/decode_forms.txt" 2>&1
bash tools/config2_profile.sh > /dev/null 2>&1; cp "$ROOT/gpurun_out/config2_1000_frames.txt" "$OUT/config2_1000_frames.txt"   # BASELINE configs[2] under rocprofv3
cd "$ROOT"
python bench.py > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/bench_line.json"
SELA_BENCH_FORCE_EXCHANGE=1 python bench.py --workload album --steps 3 --warmup 1 > "$OUT/bench_album.log" 2>&1
tail -1 "$OUT/bench_album.log" > "$OUT/bench_album_1gpu_line.json"
# the file-to-file verbs: what the file system gives (tools/io_probe), one encodeFile / decodeFile on a time axis, and
# the legs against the size of the I/O pool
[ -x tools/io_probe ] || g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/io_probe.cpp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o tools/io_probe
tools/io_probe /dev/shm 32 > "$OUT/io_probe.txt" 2>&1
python tools/batch_rate.py 12 > "$OUT/batch_rate.txt" 2>&1
python tools/io_sweep.py 1 2 4 8 16 > "$OUT/io_threads.txt" 2>&1
host/sela_filebench /dev/shm/sela_io/track.wav /dev/shm/sela_io 3 trace > "$OUT/io_trace.txt" 2>&1
python tools/e2e_trace.py > "$OUT/e2e_trace.txt" 2>&1
ls -la "$OUT"
cat "$OUT/valu_counters.txt" "$OUT/timeline.txt" | grep -v amdgpu.ids
cat "$OUT/traffic.json"
head -20 "$OUT/kernel_stats.csv"
