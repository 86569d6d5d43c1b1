#!/bin/bash
# GPU call: selected tests, MFMA-shape / tile-choice A/B on the whole step, kernel-trace profiles (full batch and the 8-rank shard size)
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py -m gpu -q -s -k "comm or allreduce or ranks or gemm or sampler or augment or fft_pair or enforce or sharp or illustrip or fast_transform or vit_tiny" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -4 $O/${TAG}_tests.log
B="python bench.py --steps 40 --no-cpu-baseline --no-legs"
for v in 0 1; do APH_GEMM_MFMA32=$v timeout 200 $B > $O/${TAG}_bench_mfma32_$v.json 2>> $O/${TAG}_bench.err; done
APH_GEMM8_MIN_TILES=100000 timeout 200 $B > $O/${TAG}_bench_nogemm8.json 2>> $O/${TAG}_bench.err
APH_GEMM8_MIN_TILES=100000 APH_GEMM_MFMA32=1 timeout 200 $B > $O/${TAG}_bench_nogemm8_m32.json 2>> $O/${TAG}_bench.err
for f in $O/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print('%-40s %.1f steps/s  gemm %.3f ms/step  %.0f TF/s' % (sys.argv[1].split('/')[-1], j['value'], r['gemm_ms_per_step'], r['achieved']))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -- python $R/bench.py --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof.log 2>&1)
python tools/prof_summary.py $O/${TAG}_prof 25 $O/${TAG}_kernel_stats.csv 44 > $O/${TAG}_kernel_stats.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof24 -- python $R/bench.py --steps 20 --samples 26 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof24.log 2>&1)
python tools/prof_summary.py $O/${TAG}_prof24 25 $O/${TAG}_kernel_stats_24cuts.csv 44 > $O/${TAG}_kernel_stats_24cuts.txt 2>&1
find $O -name '*.db' -size +20M -delete 2>/dev/null
tail -2 $O/${TAG}_kernel_stats.txt $O/${TAG}_kernel_stats_24cuts.txt
