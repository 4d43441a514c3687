#!/bin/bash
# per-UNet-step slope and decode intercept of the sampler: bench.py at 10 / 50 / 100 inference steps
for n in 10 50 100; do
  python bench.py --no-cpu-baseline --no-pipelined --no-other-configs --steps 6 --warmup 2 --inference-steps $n 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3))"
done
