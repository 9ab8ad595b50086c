#!/bin/bash
# tools/collect_all.sh <round-tag>  (run ON THE GPU BOX through gpurun): everything profiles/<tag>/ holds.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1
bash tools/valu_counters.sh > "$OUT/valu_counters.txt" 2>&1
bash tools/timeline.sh > "$OUT/timeline.txt" 2>&1
LANES=2 bash tools/timeline.sh > "$OUT/timeline_two_lanes.txt" 2>&1
python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids > "$OUT/phase_cycles.txt"
python tools/phase_profile.py 1000 2>&1 | grep -v amdgpu.ids > "$OUT/phase_cycles_1000_frames.txt"
# round 4: k_encode_teams against k_encode_blocks (kernel and wall time by batch size, the kernels forced), the teams' phases,
# when the waves of a launch start and end, and the instruction counters of the three block kernels at 10,000 frames
python tools/teams_sweep.py 500 1000 2000 3000 3875 6000 10000 20000 40000 2>&1 | grep -v amdgpu.ids > "$OUT/teams_sweep.txt"
python tools/teams_sweep.py 3300 3500 3875 4096 4200 4400 4700 5000 5500 6000 7000 8192 8300 9000 10000 12288 12500 15000 16384 16600 2>&1 | grep -v amdgpu.ids > "$OUT/teams_fine_sweep.txt"
for T in 16 8; do python tools/phase_profile.py 3875 $T 2>&1 | grep -v amdgpu.ids | sed -n "/^k_encode_teams/,/store slot/p"; done > "$OUT/phase_cycles_teams.txt"
{ for T in 16 8; do echo "teams of $T, 3875 frames:"; python tools/ramp_profile.py 3875 $T 2>&1 | grep -v amdgpu.ids; done
  echo "teams of 16, 10000 frames:"; python tools/ramp_profile.py 10000 16 2>&1 | grep -v amdgpu.ids; } > "$OUT/ramp_teams.txt"
python tools/skip_rate.py 2>&1 | grep -v amdgpu.ids > "$OUT/skip_rate.txt"
bash tools/teams_counters.sh 10000 2>&1 | grep -v amdgpu.ids > "$OUT/teams_counters_10000_frames.txt"
# what the block kernel, the lanes and the wave priorities do to the headline (experiments; the first line is the default)
for CFG in "" "--lanes 1" "--encode-teams 0" "--encode-teams 8" "--encode-teams 8 --lanes 4" "--encode-fused" "--encode-fused --lanes 1" "--encode-teams 0 --lanes 1" "--priorities 00010203" "--priorities 00010203 --lanes 1"; do
  python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs $CFG 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('bench.py %-40s %6.0f M samples/s  %.4f ms/step  one lane %6.0f  kernels %s' % ('$CFG', d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['kernel_ms'].items()}))"
done > "$OUT/headline_variants.txt" 2>&1
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
