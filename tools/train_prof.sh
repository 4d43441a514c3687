#!/bin/bash
# rocprofv3 kernel trace of the training step, one row per (kernel, grid): tools/train_prof.sh <tag> [env assignments...]
# -> gpurun_out/<tag>_train_by_grid.txt, gpurun_out/<tag>_train_stats.txt
TAG=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 5 --warmup 2 --eager --no-vae > /tmp/prof_$TAG.log 2>&1)
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
python tools/rocprof_summary.py $DB --by-grid > gpurun_out/${TAG}_train_by_grid.txt
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_train_stats.txt
tail -1 gpurun_out/${TAG}_train_stats.txt
