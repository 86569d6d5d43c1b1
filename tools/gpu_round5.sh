#!/bin/bash
# GPU call: persistent phased GEMM (next tile's first k-tile under the epilogue) A/B against one tile per workgroup
TAG=${1:-r02e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -m gpu -q -x -k "gemm or vit or reproduc or bitwise or full_size" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -n 3 $O/${TAG}_tests.log
for p in 0 1 0 1; do
  echo "== APH_GEMM8_PERSIST=$p" >> $O/${TAG}_ab.txt
  APH_GEMM8_PERSIST=$p timeout 120 python tools/gemm_cfg_bench.py 4 >> $O/${TAG}_ab.txt 2>&1
  APH_GEMM8_PERSIST=$p timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('c2 %.1f steps/s  gemm %.3f ms/step %.0f TF/s' % (d['value'], r.get('gemm_ms_per_step', 0), r['achieved']))
" >> $O/${TAG}_ab.txt
done
for p in 0 1; do
  echo "== c4 APH_GEMM8_PERSIST=$p" >> $O/${TAG}_ab.txt
  APH_GEMM8_PERSIST=$p timeout 300 python bench.py --config c4 --steps 40 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('c4 %.1f steps/s  gemm %.3f ms/step %.0f TF/s' % (d['value'], r.get('gemm_ms_per_step', 0), r['achieved']))
" >> $O/${TAG}_ab.txt
done
cat $O/${TAG}_ab.txt
