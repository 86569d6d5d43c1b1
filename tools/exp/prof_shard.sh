#!/bin/bash
# kernel-time breakdown of one rank's share of the C2 step at 8 / 2 ranks (24 / 95 cuts): where the strong-scaling ceiling comes from
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp
cd $R
for s in 26 100; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_s$s -- python $R/bench.py --samples $s --no-cpu-baseline --no-roofline --no-legs --steps 20 > $O/prof_s$s.log 2>&1)
python tools/prof_summary.py $O/prof_s$s 25 $O/r03_kernel_stats_s$s.csv 48 > $O/r03_kernel_stats_s$s.txt 2>&1
rm -rf $O/prof_s$s
grep '^{' $O/prof_s$s.log | cut -c1-260
done
