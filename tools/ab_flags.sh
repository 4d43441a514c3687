#!/bin/bash
# same-box A/B of one build with and without a debug flag set: tools/ab_flags.sh <RLDM_DBG_FLAGS value> [rounds] [bench args]
# (bench.py value, interleaved; "off" = the flag's route disabled, "on" = the default build)
FLAGS=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for which in off on; do
    if [ $which = off ]; then export RLDM_DBG_FLAGS=$FLAGS; else unset RLDM_DBG_FLAGS; fi
    v=$(python bench.py --no-cpu-baseline --no-pipelined --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d.get('unet_launches_per_step'))")
    echo "$which $v"
  done
done
