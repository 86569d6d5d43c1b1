#!/bin/bash
# round 5: 64x64 tiles on eight waves (tile_cfg 13) -- shapes at the shard sizes, step times on / off
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -n 3 $O/${TAG}_tests.log
for M in 300 600 1200 2400 4750; do echo "== M=$M"; M=$M timeout 200 python tools/gemm_shapes_bench.py 1 13 10 2>&1 | grep -v amdgpu; done | tee $O/${TAG}_gemm_shapes.txt
for s in 6 13 26 51 100; do
  for path in "small8=0" "small8=256" "small8=512" "small8=100000"; do
    timeout 300 python bench.py --f16 --reps 1 --vit-path $path --samples $s --steps 60 --warmup 10 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('samples $s $path: cuts %d  %.3f ms/step  %.1f steps/s  loss %.5f' % (d['config']['samples_effective'], d['ms_per_step'], d['value'], d['config']['final_loss']))
"
  done
done | tee $O/${TAG}_steps.txt
