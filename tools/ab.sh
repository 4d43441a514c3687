#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <prev.so> [rounds]  (bench.py value, interleaved)
PREV=$1; N=${2:-3}
for i in $(seq $N); do
  for which in prev new; do
    if [ $which = prev ]; then export RLDM_LIB=$PWD/$PREV; else unset RLDM_LIB; fi
    v=$(python bench.py --no-cpu-baseline --no-pipelined --no-other-configs --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))")
    echo "$which $v"
  done
done
