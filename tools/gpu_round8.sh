#!/bin/bash
# Round-3, the last two GPU-minutes: kernel trace of the C2 bench line on the final library (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
(cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats -d $O/r03h_prof -- python $R/bench.py --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/r03h_prof.log 2>&1)
echo "rocprof rc $?"
timeout 20 python tools/prof_summary.py $O/r03h_prof 25 $O/r03h_kernel_stats.csv 48 > $O/r03h_kernel_stats.txt 2>&1
tail -n 4 $O/r03h_kernel_stats.txt
rm -rf $O/r03h_prof
