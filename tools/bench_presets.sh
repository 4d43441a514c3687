#!/bin/bash
# one bench.py line per BASELINE config that fits a single GPU (summary view); usage: tools/bench_presets.sh [extra bench.py flags]
for a in "--preset RangeLDM --batch 16" "--preset upsample --batch 16" "--preset nuscenes --batch 4" "--preset nuscenes --batch 32" "--preset RangeDM --batch 1 --inference-steps 10" "--preset RangeDM --batch 4 --inference-steps 10"; do
  echo "== $a $*"
  timeout 500 python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-other-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print(round(d['value'],2),'img/s', round(d['ms_per_step'],2),'ms', d.get('end_to_end_tflops'),'TFLOP/s', d['roofline']['kernel'], d['roofline']['frac']); print({k:(v['share'],v['avg_us'],v['tflops']) for k,v in list(d['kernels'].items())[:6]})
except Exception as e: print('ERR', l[-600:])
"
done
