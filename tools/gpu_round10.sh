#!/bin/bash
# Round-3, the very last GPU seconds: the coarse irDWT kernel with its level table in scalar registers -- DWT tests and the all-levels time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 30 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "test_dwt" > $O/r03j_tests.log 2>&1; echo "test_dwt rc $?"; tail -n 1 $O/r03j_tests.log
timeout 15 python tools/exp/dwt_levels.py 2>&1 | grep "all levels" | tee $O/r03j_dwt_all_levels.txt
timeout 40 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -k "c4_irdwt_full_size" > $O/r03j_tests_c4.log 2>&1; echo "c4 full size rc $?"; tail -n 1 $O/r03j_tests_c4.log
