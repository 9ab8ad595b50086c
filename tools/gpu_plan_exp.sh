#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for R in 1 2 3; do for E in "" "SELA_EXP_PLAN_SMALL=1"; do for CFG in "" "--lanes 1"; do
  env $E python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs $CFG 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('%-22s bench.py %-10s %6.0f M  %.4f ms/step  one lane %6.0f  plan %.4f enc %.4f dec %.4f' % ('$E', '$CFG', d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], d['kernel_ms']['encode_plan'], d['kernel_ms']['encode_blocks'], d['kernel_ms']['decode_frames']))"
done; done; done
