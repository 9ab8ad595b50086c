#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
enc() { python - "$@" <<'PY'
import sys
# args: three schedules "3210" per class -> hex
c=[sys.argv[1],sys.argv[2],sys.argv[3]]
v=0x80000000
for q in range(4):
    b=int(c[0][q])|int(c[1][q])<<2|int(c[2][q])<<4
    v|=b<<(8*q)
print("%08x"%v)
PY
}
for S in "3210 3210 3210" "3210 2210 1210" "3210 3310 3320" "3210 2100 3321" "1110 2210 3320" "3210 3210 0210" "3320 2210 1100" "3210 2321 1232" "2210 3210 2210" "3211 3210 3100"; do
  H=$(enc $S)
  for L in 1 2; do
    python bench.py --no-cpu-baseline --no-host-legs --no-extra-legs --priorities $H --lanes $L 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('%-16s %s lanes %d  %6.0f M  one lane %6.0f  enc kernel %.4f' % ('$S', '$H', $L, d['value'], d['lanes']['value_one_lane'], d['kernel_ms']['encode_blocks']))"
  done
done
