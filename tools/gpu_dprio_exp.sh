#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for R in 1 2; do for E in "" "SELA_EXP_SYNTH_PRIO=020100" "SELA_EXP_SYNTH_PRIO=010000" "SELA_EXP_SYNTH_PRIO=030201" "SELA_EXP_SYNTH_PRIO=030100" "SELA_EXP_SYNTH_PRIO=000102"; do for CFG in "" "--lanes 1"; do
  env $E python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs $CFG 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('%-28s bench.py %-10s %6.0f M  %.4f ms/step  one lane %6.0f  enc %.4f dec %.4f' % ('$E', '$CFG', d['value'], d['ms_per_step'], d['lanes']['value_one_lane'], d['kernel_ms']['encode_blocks'], d['kernel_ms']['decode_frames']))"
done; done; done
