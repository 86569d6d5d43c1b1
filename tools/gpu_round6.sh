#!/bin/bash
# Round-3 closing GPU call (10 GPU-minutes were left): the batched-load irDWT with its one-launch coarse tail, the 16-byte Adam /
# guard kernels, the batched FFT load phases and the fused first-block LayerNorm pairs on real hardware -- targeted tests first,
# the C4 line with the coarse tail on and off, per-level irDWT times, then the whole -m gpu suite and the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r03f_timeline.txt; }
stamp start
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dwt or adam or synth or fft_pair or vit_tiny or vit_base" > $O/r03f_tests_quick.log 2>&1
stamp "quick tests rc $?"; tail -n 2 $O/r03f_tests_quick.log
for c in 1 0; do
  APH_IDWT_COARSE=$c timeout 150 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $O/r03f_bench_c4_coarse$c.json 2> $O/r03f_bench_c4_coarse$c.err
  stamp "c4 coarse=$c rc $?"
done
timeout 90 python tools/exp/dwt_levels.py > $O/r03f_dwt_levels.txt 2>&1
stamp "dwt levels rc $?"; head -n 1 $O/r03f_dwt_levels.txt
timeout 420 python -m pytest tests -m gpu -q > $O/r03f_gpu_tests.log 2>&1
stamp "full gpu tests rc $?"; tail -n 3 $O/r03f_gpu_tests.log
timeout 240 python bench.py > $O/r03f_bench_c2.json 2> $O/r03f_bench_c2.err
stamp "default bench rc $?"; cut -c1-300 $O/r03f_bench_c2.json
