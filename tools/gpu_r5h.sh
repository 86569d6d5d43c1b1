#!/bin/bash
# round 5, GPU call: fused forward / backward block kernels v2 (DPP LayerNorm prologues) -- tests, step times by path, 24-cut profile
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or vit or attention" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -n 8 $O/${TAG}_tests.log
for s in 6 26 51; do
  for path in "fused=0" "fused=100000,fattn=0" "fused=100000,fattn=1"; do
    timeout 300 python bench.py --f16 --reps 1 --vit-path $path --samples $s --steps 60 --warmup 10 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('samples $s $path: cuts %d  %.3f ms/step  %.1f steps/s  loss %.5f skipped %d' % (d['config']['samples_effective'], d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['skipped_steps']))
" >> $O/${TAG}_steps.txt
  done
done
cat $O/${TAG}_steps.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_s26 -- python $R/bench.py --f16 --reps 1 --vit-path fused=100000,fattn=0 --samples 26 --steps 25 --warmup 5 --no-cpu-baseline --no-legs --no-roofline --no-graph > $O/${TAG}_prof_s26.log 2>&1
python $R/tools/prof_summary.py $O/${TAG}_prof_s26 30 $O/${TAG}_kernel_stats_s26.csv 22 > $O/${TAG}_kernel_stats_s26.txt 2>&1 || true
head -30 $O/${TAG}_kernel_stats_s26.txt
cd $R
for M in 1200; do M=$M timeout 200 python tools/gemm_shapes_bench.py 1 10 14 16 17 2>&1 | grep -v amdgpu; done | tee $O/${TAG}_gemm_shapes.txt
MS=1200 python tools/exp/gemm_rs_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_rs_trace.txt
for path in "rs=2,fused=0" "rs=2,fused=100000,fattn=0"; do
    timeout 300 python bench.py --f16 --reps 1 --vit-path $path --samples 26 --steps 60 --warmup 10 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('samples 26 $path: cuts %d  %.3f ms/step  %.1f steps/s' % (d['config']['samples_effective'], d['ms_per_step'], d['value']))
"
done | tee -a $O/${TAG}_steps.txt
