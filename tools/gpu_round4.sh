#!/bin/bash
# GPU call: fixed tests + PMC passes over the step (where do the sampler kernels' cycles go?)
TAG=${1:-r02d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py tests/test_gpu_step.py -m gpu -q -s -k "linear_head or aesthetic or illustrip_frame_loop_vs or sharp_expand or full_size_properties" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -3 $O/${TAG}_tests.log
export TMPDIR=/tmp
rocprofv3 -L > $O/${TAG}_counters.txt 2>&1
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-legs --no-graph"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmcA -- $B > $O/${TAG}_pmcA.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/${TAG}_pmcB -- $B > $O/${TAG}_pmcB.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $O/${TAG}_pmcC -- $B > $O/${TAG}_pmcC.log 2>&1)
python tools/pmc_table.py $O/${TAG}_pmc_table.csv $O/${TAG}_pmcA $O/${TAG}_pmcB $O/${TAG}_pmcC > $O/${TAG}_pmc_table.txt 2>&1
find $O -name '*counter_collection.csv' -size +30M -delete 2>/dev/null
find $O -name '*.db' -size +20M -delete 2>/dev/null
tail -n 30 $O/${TAG}_pmc_table.txt | cut -c1-260
