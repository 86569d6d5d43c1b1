#!/bin/bash
# round 5, GPU call B: load-rate microbenchmark (which operand path can feed a small-M GEMM), 24-cut kernel profile
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/exp/load_rate.py > $O/${TAG}_load_rate.txt 2>&1
cat $O/${TAG}_load_rate.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_s26 -- python $R/bench.py --vit-path rs=0 --samples 26 --steps 25 --warmup 5 --no-cpu-baseline --no-legs --no-roofline --no-graph > $O/${TAG}_prof_s26.log 2>&1
python $R/tools/prof_summary.py $O/${TAG}_prof_s26 30 $O/${TAG}_kernel_stats_s26.csv 60 > $O/${TAG}_kernel_stats_s26.txt 2>&1 || true
head -70 $O/${TAG}_kernel_stats_s26.txt
