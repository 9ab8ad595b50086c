#!/bin/bash
# tools/collect_r05.sh  (run ON THE GPU BOX through gpurun): what profiles/r05/ holds, on the tree as it is.
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1
bash tools/valu_counters.sh > "$OUT/valu_counters.txt" 2>&1
# what the lanes and the wave priorities do to the headline (the first line is the default: priorities by the library)
for CFG in "" "--lanes 1" "--priorities 0" "--priorities 0 --lanes 1" "--priorities 00010203" "--priorities 00010203 --lanes 1" "--encode-teams 0 --lanes 1"; do
  python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs $CFG 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('bench.py %-36s %6.0f M samples/s  %.4f ms/step  one lane %6.0f  kernels %s' % ('$CFG', d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['kernel_ms'].items()}))"
done > "$OUT/headline_variants.txt" 2>&1
SELA_SWEEP_HOST=0 python tools/sweep.py > "$OUT/sweep.txt" 2>&1
[ -x tools/tmp/chain_ubench_bin ] && tools/tmp/chain_ubench_bin > "$OUT/chain_ubench.txt" 2>&1
# the any-length route: calls and kernels
python tools/generic_probe.py 2>&1 | grep -v amdgpu.ids > "$OUT/generic_route.txt"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python "$ROOT/tools/generic_probe.py" > /tmp/gp.log 2>&1; find /tmp/gp -name "*kernel_stats.csv" -exec cp {} "$OUT/generic_kernel_stats.csv" \;)
# the frame classes from many threads: the decoder class on its exact route and on the coalesced fast kernels
{ for T in 1 4 16 64; do host/sela_filebench frames $T 16; done; for T in 1 4 16 64; do host/sela_filebench frames $T 16 fast; done; } > "$OUT/frame_classes_fanout.txt" 2>&1
# the differential corpus (its printed summary)
python -m pytest tests/test_gpu_round5.py -q -s -k corpus 2>&1 | grep -v amdgpu.ids | tail -6 > "$OUT/corpus.txt"
# the default line with these profiles in place
mkdir -p "$ROOT/profiles/$TAG"
cp "$OUT"/kernel_stats.csv "$OUT"/traffic.json "$OUT"/traffic_calibration.json "$OUT"/valu_counters.txt "$OUT"/sources.sha256 "$ROOT/profiles/$TAG/" 2>/dev/null
python bench.py > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/bench_line.json"
ls -la "$OUT"
cat "$OUT/headline_variants.txt" "$OUT/generic_route.txt" "$OUT/frame_classes_fanout.txt" "$OUT/corpus.txt"
grep -v amdgpu.ids "$OUT/valu_counters.txt"
python -c "import json; d=json.load(open('$OUT/traffic.json')); print(json.dumps({k:{kk:vv for kk,vv in v.items() if 'calibrated' in kk or kk=='launches'} for k,v in d.items() if k!='_calibration'}, indent=0))"
head -8 "$OUT/kernel_stats.csv"
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print({k:d[k] for k in ('value','ms_per_step','profiles_stale','roofline','e2e','any_length')}); print(d['decode10k']['value'], d['album']['value'])"
