#!/bin/bash
# One gpurun call: GPU tests, MFMA-shape microbench, the default bench line, a kernel-trace profile.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [tests|notests]'
TAG=${1:-r02}
MODE=${2:-tests}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$MODE" = tests ]; then
  timeout 1100 python -m pytest tests -m gpu -q -s --durations=12 > $O/${TAG}_gpu_tests.log 2>&1
  echo "pytest rc $?" >> $O/${TAG}_gpu_tests.log
  tail -5 $O/${TAG}_gpu_tests.log
fi
if [ -f tools/exp/mfma_rate.so ] && [ "$MODE" = tests ]; then timeout 120 python tools/exp/mfma_rate.py > $O/${TAG}_mfma_rate.log 2>&1; fi
timeout 400 python bench.py --steps 30 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench rc $?"; cut -c1-600 $O/${TAG}_bench.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -- python $R/bench.py --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof.log 2>&1)
python tools/prof_summary.py $O/${TAG}_prof 25 $O/${TAG}_kernel_stats.csv 40 > $O/${TAG}_kernel_stats.txt 2>&1
# keep only the summaries of the trace (the raw db is tens of MB)
find $O/${TAG}_prof -name '*.db' -size +20M -delete 2>/dev/null
tail -3 $O/${TAG}_kernel_stats.txt
