#!/bin/bash
# Round-3, what was left of the GPU budget: kernel trace of the C4 line (3840x2160 DWT db3, ViT-B/16) on the final library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
(cd /tmp && timeout 70 rocprofv3 --kernel-trace --stats -d $O/r03i_prof -- python $R/bench.py --config c4 --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/r03i_prof.log 2>&1)
echo "rocprof rc $?"
timeout 20 python tools/prof_summary.py $O/r03i_prof 25 $O/r03i_kernel_stats_c4.csv 60 > $O/r03i_kernel_stats_c4.txt 2>&1
grep -i "idwt\|total" $O/r03i_kernel_stats_c4.txt | cut -c1-150
rm -rf $O/r03i_prof
