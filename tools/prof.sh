#!/bin/bash
# usage: tools/prof.sh <tag> ["pmc groups, comma-separated counters, space-separated passes"]   (run on the GPU box through gpurun)
# kernel-trace + stats of a short bench run, then one PMC pass per group; everything lands in gpurun_out/<tag>/
# BENCH_ARGS selects the workload (default: the inference headline with one batch in flight; "--train" = the training step)
set -o pipefail
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS=${BENCH_ARGS:---pipeline 1 --no-next --no-alt --no-c1}
echo "[prof] kernel trace: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS" | tee $OUT/log.txt
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS >> $OUT/log.txt 2>&1
echo "[prof] trace rc=$?" | tee -a $OUT/log.txt
if [ -n "$2" ]; then
  i=0
  for grp in $2; do
    i=$((i+1))
    timeout -k 10 420 rocprofv3 --kernel-trace --pmc ${grp//,/ } --output-format csv -d $OUT/pmc$i -o pmc -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS >> $OUT/log.txt 2>&1
    echo "[prof] pmc$i ($grp) rc=$?" | tee -a $OUT/log.txt
  done
fi
