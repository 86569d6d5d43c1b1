R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_smp -- python $R/tools/sampler_bench.py > $O/prof_smp.log 2>&1)
cd $R; python tools/prof_summary.py $O/prof_smp 1 $O/smp_kernel_stats.csv 20 2>&1 | cut -c40-200
rm -rf $O/prof_smp
