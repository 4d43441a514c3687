#!/bin/bash
# same-box A/B of routing switches given as environment settings: tools/ab_env.sh <rounds> "VAR=1 VAR2=2" "VAR=3" ... [-- bench args]
# (bench.py value, interleaved over the rounds; "-" = no settings)
N=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $N); do
  for s in "${SETS[@]}"; do
    if [ "$s" = "-" ]; then e=""; else e="$s"; fi
    v=$(env $e python bench.py --no-cpu-baseline --no-pipelined --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d.get('unet_launches_per_step'))")
    echo "[$s] $v"
  done
done
