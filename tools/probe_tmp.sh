cd $GRAFT_REPO_ROOT
for L in 1 2 3 4; do python bench.py --lanes $L --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes', d['lanes']['in_flight'], 'value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'one-lane', round(d['lanes']['value_one_lane']), d['repetitions']['spread_frac'], d['roundtrip_lossy_frames'])"; done
SELA_BENCH_FORCE_EXCHANGE=1 python bench.py --workload album --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('album lanes', d['lanes']['in_flight'], 'value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'one-lane', round(d['lanes']['value_one_lane']), d['layout_matches_reference'], d['roundtrip_lossy_frames'])"
python -m pytest tests/test_gpu_round2.py -q -k random_valid -p no:cacheprovider 2>&1 | tail -2
