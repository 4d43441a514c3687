#!/bin/bash
# One pass over everything profiles/ holds for a version tag (run on the GPU box through gpurun):
#   tools/collect_profiles.sh v24   ->  gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
# Kernel trace and the PMC passes are separate rocprofv3 runs (counters are never combined with the trace domains).
set -u
TAG=${1:-vX}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-other-configs"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/${TAG}_gpu_tests.txt
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- $BENCH > /tmp/prof_$TAG/kt.log 2>&1
DB=$(find /tmp/prof_$TAG/kt -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > $OUT/${TAG}_rocprof_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$TAG/$c -o pmc -- $BENCH > /tmp/prof_$TAG/$c.log 2>&1
  python tools/pmc_summary.py /tmp/prof_$TAG/$c > $OUT/${TAG}_pmc_$(echo $c | tr A-Z a-z).txt
done
python tools/traffic_json.py /tmp/prof_$TAG/FETCH_SIZE /tmp/prof_$TAG/WRITE_SIZE > $OUT/${TAG}_traffic.json; cp $OUT/${TAG}_traffic.json profiles/round6_traffic.json 2>/dev/null
# (bench.py after the traffic passes: its roofline.traffic reads the newest profiles/round*_traffic.json)
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --train --steps 20 --warmup 3 > $OUT/${TAG}_bench_train.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_$TAG/mfma -o pmc -- $BENCH > /tmp/prof_$TAG/mfma.log 2>&1
python tools/pmc_summary.py /tmp/prof_$TAG/mfma > $OUT/${TAG}_pmc_mfma.txt
python tools/mfma_util.py $OUT/${TAG}_pmc_mfma.txt > $OUT/${TAG}_mfma_util.txt
# VALU / wait split per kernel (round 5: the attention launch is VALU-issue bound; quad-cycle units, MI355X_MICROARCH.md "rocprofv3 PMC slots")
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/prof_$TAG/valu -o pmc -- $BENCH > /tmp/prof_$TAG/valu.log 2>&1
python tools/pmc_summary.py /tmp/prof_$TAG/valu > $OUT/${TAG}_pmc_valu.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/tr -o tr -- python tools/bench_train.py --steps 5 --warmup 2 --eager > /tmp/prof_$TAG/tr.log 2>&1
DB=$(find /tmp/prof_$TAG/tr -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > $OUT/${TAG}_train_rocprof_kernel_stats.txt
[ -n "$DB" ] && python tools/rocprof_summary.py $DB --by-grid > $OUT/${TAG}_train_by_grid.txt
# (the trace holds 7 steps: 2 warm-ups + 5; TOTAL calls / 7 = launches per step, VAE encode included)
python tools/graph_trace.py --top 120 > $OUT/${TAG}_graph_trace.txt 2>&1
tail -3 $OUT/${TAG}_gpu_tests.txt; head -c 400 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_bench_train.json; echo
ls -la $OUT | grep ${TAG}_
