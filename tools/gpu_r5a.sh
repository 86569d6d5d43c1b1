#!/bin/bash
# round 5, GPU call A: per-wave split-K small-M GEMM (vit_gemm_rs.h) -- correctness on hardware, shape sweep, shard-size step times on/off
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or vit" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -n 5 $O/${TAG}_tests.log
for M in 50 300 1200 2400 4750; do
  echo "== M=$M" >> $O/${TAG}_gemm_shapes.txt
  M=$M timeout 300 python tools/gemm_shapes_bench.py 1 10 14 15 16 17 5 >> $O/${TAG}_gemm_shapes.txt 2>&1
done
cat $O/${TAG}_gemm_shapes.txt
for s in 1 6 26 46 51 100; do
  for rs in 1 0; do
    timeout 300 python bench.py --vit-path rs=$rs --samples $s --steps 60 --warmup 10 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('samples $s rs $rs: cuts %d  %.3f ms/step  %.1f steps/s  loss %.5f skipped %d' % (d['config']['samples_effective'], d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['skipped_steps']))
" >> $O/${TAG}_steps.txt
  done
done
cat $O/${TAG}_steps.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_s26 -- python $R/bench.py --samples 26 --steps 25 --warmup 5 --no-cpu-baseline --no-legs --no-roofline --no-graph > $O/${TAG}_prof_s26.log 2>&1
python $R/tools/prof_summary.py $O/${TAG}_prof_s26 30 $O/${TAG}_kernel_stats_s26.csv 50 > $O/${TAG}_kernel_stats_s26.txt 2>&1 || true
head -60 $O/${TAG}_kernel_stats_s26.txt
