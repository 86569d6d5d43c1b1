#!/bin/bash
# The closing GPU calls of round 3 (15 GPU-minutes were left), one mode per call:
#   tests1  targeted tests, C4 line with the coarse irDWT tail on / off, per-level irDWT, whole -m gpu suite, default bench line
#   tests2  fused-vs-separate LayerNorm difference, DWT / ViT tests, irDWT per level, C4 line, whole -m gpu suite (final library)
#   trace   rocprofv3 --kernel-trace --stats of the C2 bench line        trace_c4   the same for --config c4
#   dwt     DWT tests + all-levels irDWT time (after the coarse kernel's level table moved to scalar registers)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r03_closing_timeline.txt; }
c4_line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']['irdwt']
        print('c4 %.1f steps/s  irdwt fwd %.1f us %.2f  adj %.1f us %.2f' % (d['value'], r['fwd_us'], r['frac'], r['bwd_us'], r['frac_adjoint']))
"; }
case "${1:-tests2}" in
tests1)
  timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dwt or adam or synth or fft_pair or vit_tiny or vit_base" > $O/r03f_tests_quick.log 2>&1
  stamp "quick tests rc $?"; tail -n 2 $O/r03f_tests_quick.log
  for c in 1 0; do
    APH_IDWT_COARSE=$c timeout 150 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $O/r03f_bench_c4_coarse$c.json 2> $O/r03f_bench_c4_coarse$c.err
    stamp "c4 coarse=$c rc $?"
  done
  timeout 90 python tools/exp/dwt_levels.py > $O/r03f_dwt_levels.txt 2>&1; stamp "dwt levels rc $?"
  timeout 420 python -m pytest tests -m gpu -q > $O/r03f_gpu_tests.log 2>&1; stamp "full gpu tests rc $?"; tail -n 3 $O/r03f_gpu_tests.log
  timeout 240 python bench.py > $O/r03f_bench_c2.json 2> $O/r03f_bench_c2.err; stamp "default bench rc $?"
  ;;
tests2)
  timeout 60 python tools/exp/ln_fuse_diff.py > $O/r03g_ln_fuse_diff.txt 2>&1; stamp "ln fuse diff rc $?"
  timeout 120 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py tests/test_gpu_step.py -m gpu -q -s -k "dwt or c4 or vit_base" > $O/r03g_tests_quick.log 2>&1
  stamp "dwt / vit tests rc $?"; grep "fused vs separate\|passed\|failed" $O/r03g_tests_quick.log | tail -n 6
  timeout 60 python tools/exp/dwt_levels.py > $O/r03g_dwt_levels.txt 2>&1; stamp "dwt levels rc $?"; grep "all levels" $O/r03g_dwt_levels.txt
  timeout 100 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $O/r03g_bench_c4.json 2> $O/r03g_bench_c4.err
  stamp "c4 bench rc $?"; c4_line $O/r03g_bench_c4.json
  timeout 330 python -m pytest tests -m gpu -q > $O/r03g_gpu_tests.log 2>&1; stamp "full gpu tests rc $?"; tail -n 3 $O/r03g_gpu_tests.log
  ;;
trace|trace_c4)
  CFG=""; TAG=r03h; [ "$1" = trace_c4 ] && { CFG="--config c4"; TAG=r03i; }
  (cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -- python $R/bench.py $CFG --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof.log 2>&1)
  echo "rocprof rc $?"
  timeout 20 python tools/prof_summary.py $O/${TAG}_prof 25 $O/${TAG}_kernel_stats.csv 60 > $O/${TAG}_kernel_stats.txt 2>&1
  tail -n 4 $O/${TAG}_kernel_stats.txt
  rm -rf $O/${TAG}_prof
  ;;
dwt)
  timeout 30 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "test_dwt" > $O/r03j_tests.log 2>&1; echo "test_dwt rc $?"; tail -n 1 $O/r03j_tests.log
  timeout 15 python tools/exp/dwt_levels.py 2>&1 | grep "all levels" | tee $O/r03j_dwt_all_levels.txt
  timeout 40 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -k "c4_irdwt_full_size" > $O/r03j_tests_c4.log 2>&1; echo "c4 full size rc $?"; tail -n 1 $O/r03j_tests_c4.log
  ;;
esac
