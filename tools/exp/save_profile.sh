cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for s in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/r06l_saveprof_$s -- python $R/tools/exp/save_profile.py $s > $O/r06l_saveprof_$s.log 2>&1
  tail -1 $O/r06l_saveprof_$s.log
  python $R/tools/prof_summary.py $O/r06l_saveprof_$s 70 $O/r06l_saveprof_kernels_$s.csv 14 2>&1 | head -18 | cut -c1-200
  find $O/r06l_saveprof_$s -name '*memory_copy*' | head -3
  find $O/r06l_saveprof_$s -name '*.db' -size +20M -delete
done
ls $O/r06l_saveprof_1/* | head
