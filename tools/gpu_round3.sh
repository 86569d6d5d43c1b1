#!/bin/bash
# GPU call: re-run the tests touched since the last call, bench lines for every config, kernel-trace profile of the headline
TAG=${1:-r02c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py tests/test_gpu_step.py -m gpu -q -s -k "sampler or augment or linear_head or aesthetic or illustrip or enforce or fast_transform or full_size or bitwise or sharp or dwt_and_pixel" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -4 $O/${TAG}_tests.log
timeout 300 python bench.py --steps 40 --no-cpu-baseline > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench.err
for c in c1 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline --no-legs > $O/${TAG}_bench_$c.json 2>> $O/${TAG}_bench.err; done
for f in $O/${TAG}_bench_c*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline'] or {}
    print('%-28s %.1f steps/s  gemm %.3f ms/step  %.0f TF/s step_frac %.3f %s' % (sys.argv[1].split('/')[-1], j['value'], r.get('gemm_ms_per_step', 0), r.get('achieved', 0), r.get('step_frac', 0), {k: round(v['value'], 1) for k, v in (j.get('legs') or {}).items()}))
    if 'irdwt' in r: print('   irdwt', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r['irdwt'].items() if k != 'kernel'})
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -- python $R/bench.py --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof.log 2>&1)
python tools/prof_summary.py $O/${TAG}_prof 25 $O/${TAG}_kernel_stats.csv 30 > $O/${TAG}_kernel_stats.txt 2>&1
find $O -name '*.db' -size +20M -delete 2>/dev/null
tail -n 2 $O/${TAG}_kernel_stats.txt
