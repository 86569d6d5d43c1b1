#!/bin/bash
# Round-3 last GPU call (6 GPU-minutes were left): the LDS-resident coarse tail of the irDWT, the fused-vs-separate LayerNorm
# difference on real hardware, the C4 line, then the whole -m gpu suite on the final library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r03g_timeline.txt; }
stamp start
timeout 60 python tools/exp/ln_fuse_diff.py > $O/r03g_ln_fuse_diff.txt 2>&1
stamp "ln fuse diff rc $?"; grep -v amdgpu.ids $O/r03g_ln_fuse_diff.txt | cut -c1-200
timeout 120 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py tests/test_gpu_step.py -m gpu -q -s -k "dwt or c4 or vit_base" > $O/r03g_tests_quick.log 2>&1
stamp "dwt / vit tests rc $?"; grep "fused vs separate\|passed\|failed" $O/r03g_tests_quick.log | tail -n 6
timeout 60 python tools/exp/dwt_levels.py > $O/r03g_dwt_levels.txt 2>&1
stamp "dwt levels rc $?"; grep "all levels" $O/r03g_dwt_levels.txt
timeout 100 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $O/r03g_bench_c4.json 2> $O/r03g_bench_c4.err
stamp "c4 bench rc $?"; python -c "
import json,sys
for l in open('$O/r03g_bench_c4.json'):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']['irdwt']
        print('c4 %.1f steps/s  irdwt fwd %.1f us %.2f  adj %.1f us %.2f' % (d['value'], r['fwd_us'], r['frac'], r['bwd_us'], r['frac_adjoint']))
"
timeout 330 python -m pytest tests -m gpu -q > $O/r03g_gpu_tests.log 2>&1
stamp "full gpu tests rc $?"; tail -n 3 $O/r03g_gpu_tests.log
