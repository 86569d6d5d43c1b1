#!/bin/bash
# Round 3: SQ / LDS / MFMA / TA / TCC counters per kernel over the C2 step of the final build (separate --pmc passes, no trace domains)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-legs --no-graph"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmcA -- $B > $O/${TAG}_pmcA.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/${TAG}_pmcB -- $B > $O/${TAG}_pmcB.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/${TAG}_pmcC -- $B > $O/${TAG}_pmcC.log 2>&1)
python tools/pmc_table.py $O/${TAG}_pmc_table.csv $O/${TAG}_pmcA $O/${TAG}_pmcB $O/${TAG}_pmcC > $O/${TAG}_pmc_table.txt 2>&1
find $O -name '*counter_collection.csv' -size +30M -delete 2>/dev/null
find $O -name '*.db' -size +20M -delete 2>/dev/null
rm -rf $O/${TAG}_pmcA $O/${TAG}_pmcB $O/${TAG}_pmcC
head -14 $O/${TAG}_pmc_table.txt | cut -c1-250
