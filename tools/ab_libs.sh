#!/bin/bash
# same-box A/B of several builds of the library (make TAG=x ...): tools/ab_libs.sh <rounds> <lib.so | default> ... [-- bench args]
# (bench.py value, interleaved over the rounds)
N=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $N); do
  for l in "${LIBS[@]}"; do
    if [ "$l" = default ]; then unset RLDM_LIB; else export RLDM_LIB=$PWD/$l; fi
    v=$(python bench.py --no-cpu-baseline --no-pipelined --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))")
    echo "$l $v"
  done
done
