#!/bin/bash
# tools/collect_round.sh <tag>  (run ON THE GPU BOX through gpurun): what profiles/<tag>/ holds, on the tree as it is.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
# 1. the bench command under rocprofv3 --kernel-trace --stats, the PMC traffic passes, the calibration, the sources' digest
bash tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1
# 2. SQ instruction counters of the bench kernels, and of the any-length route's kernels at the bench's launch size
bash tools/valu_counters.sh > "$OUT/valu_counters.txt" 2>&1
bash tools/generic_counters.sh 2>&1 | grep -v amdgpu.ids > "$OUT/generic_counters.txt"
bash tools/lds_counters.sh 2>&1 | grep "sela::" > "$OUT/lds_counters.txt"
# 3. what the lanes and the wave priorities do to the headline (the first line is the default: priorities by the library)
for CFG in "" "--lanes 1" "--priorities 0" "--priorities 0 --lanes 1" "--priorities 00010203" "--priorities 00010203 --lanes 1" "--encode-teams 0 --lanes 1"; do
  python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs $CFG 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('bench.py %-36s %6.0f M samples/s  %.4f ms/step  one lane %6.0f  kernels %s' % ('$CFG', d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['kernel_ms'].items()}))"
done > "$OUT/headline_variants.txt" 2>&1
SELA_SWEEP_HOST=0 python tools/sweep.py > "$OUT/sweep.txt" 2>&1
# 4. the any-length route: calls (small batches, ONE frame, large batches) and kernels, launch by launch
python tools/generic_probe.py big 2>&1 | grep -v amdgpu.ids > "$OUT/generic_route.txt"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python "$ROOT/tools/generic_probe.py" big > /tmp/gp.log 2>&1
 find /tmp/gp -name "*kernel_stats.csv" -exec cp {} "$OUT/generic_kernel_stats.csv" \;
 find /tmp/gp -name "*kernel_trace.csv" -exec cp {} /tmp/gp_trace.csv \;)
python - /tmp/gp_trace.csv > "$OUT/generic_launches.txt" <<'PY'
import collections, csv, sys
groups = collections.OrderedDict()  # (kernel, workgroups) in order of first launch -> durations
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if "sela" not in name:
        continue
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    groups.setdefault((name, grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# tools/generic_probe.py big under rocprofv3 --kernel-trace: the route's kernels by launch size (workgroups = blocks, subframes or frame slices)")
for (name, grid), d in groups.items():
    print(f"{name:48s} workgroups {grid:7d}  launches {len(d):3d}  min {min(d):10.1f} us  mean {sum(d)/len(d):10.1f} us")
PY
# 5. the frame classes from many threads (the reference's own loop, src/sela/encoder.cpp:58-73): the CLI's shape and odd shapes
{ for R in 1 2 3; do for shape in "2048 16" "1000 17" "4096 16"; do for T in 1 16 64 256; do echo -n "run $R shape $shape: "; host/sela_filebench frames $T 128 $shape 2>&1 | tail -1; done; done; done; } > "$OUT/frame_classes_fanout.txt" 2>&1
# 6. BASELINE configs[2] (1000 frames) and configs[3] (the album's launches) under rocprofv3 with the PMC passes
bash tools/config2_profile.sh > /dev/null 2>&1; cp "$ROOT/gpurun_out/config2_1000_frames.txt" "$OUT/config2_1000_frames.txt" 2>/dev/null
bash tools/album_profile.sh > "$OUT/album_kernels.txt" 2>&1
# 6b. an encode launch cut in two inside the library (an experiment behind --encode-split), one lane
for S in 0 400 500 520 600 700; do
  python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs --lanes 1 --encode-split $S 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('--lanes 1 --encode-split %-5s %6.0f M samples/s  %.4f ms/step  launches cut %s' % ('$S', d['value'], d['ms_per_step'], d['lanes']['encode_split']['launches_cut_in_two']))"
done > "$OUT/split_sweep.txt" 2>&1
# 7. the FP64 matrix pipe beside the vector pipe (the gate of VERDICT r5 item 3), the differential corpus' summary
[ -x tools/mfma_overlap ] && tools/mfma_overlap > "$OUT/mfma_overlap.txt" 2>&1
python -m pytest tests/test_gpu_encode_parity.py -q -s -k corpus -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 > "$OUT/corpus.txt"
# 8. the default line with these profiles in place
mkdir -p "$ROOT/profiles/$TAG"
cp "$OUT"/kernel_stats.csv "$OUT"/traffic.json "$OUT"/traffic_calibration.json "$OUT"/valu_counters.txt "$OUT"/sources.sha256 "$ROOT/profiles/$TAG/" 2>/dev/null
python bench.py > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/bench_line.json"
ls -la "$OUT"
cat "$OUT/headline_variants.txt" "$OUT/generic_route.txt" "$OUT/frame_classes_fanout.txt" "$OUT/corpus.txt" "$OUT/mfma_overlap.txt"
grep -v amdgpu.ids "$OUT/valu_counters.txt"
cat "$OUT/generic_counters.txt"
python -c "import json; d=json.load(open('$OUT/traffic.json')); print(json.dumps({k:{kk:vv for kk,vv in v.items() if 'calibrated' in kk or kk=='launches'} for k,v in d.items() if k!='_calibration'}, indent=0))"
head -8 "$OUT/kernel_stats.csv"
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print({k:d[k] for k in ('value','ms_per_step','profiles_stale','roofline','e2e','any_length')}); print(d['decode10k']['value'], d['album']['value'])"
