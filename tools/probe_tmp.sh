cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['value'], d['roofline']['frac'], d['fp64_valu']['frac'], d['valu_issue']['frac'])"
