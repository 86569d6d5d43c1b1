#!/bin/bash
# round 5: kernel trace of the C1 step (one 224x224 cut: every launch is latency)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r05t_prof_c1 -- python $R/bench.py --config c1 --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/r05t_prof_c1.log 2>&1)
python tools/prof_summary.py $O/r05t_prof_c1 65 $O/r05t_kernel_stats_c1.csv 60 > $O/r05t_kernel_stats_c1.txt 2>&1
tail -70 $O/r05t_kernel_stats_c1.txt
