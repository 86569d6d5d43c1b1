#!/bin/bash
# THE gpurun command line of this repository: one parameterised script instead of one copy per round (VERDICT r5 weak #12 -- the
# copy-pasted per-round scripts are where the stale "5 steps" of round 5's C4 traffic figure came from).
#   gpurun --timeout 2400 -- 'bash tools/gpu.sh <tag> <stage> [<stage> ...]'
# Everything lands in gpurun_out/<tag>_*; what is to be judged gets copied into profiles/ afterwards.  Stages:
#   tests        pytest -m gpu -s (printed parity numbers)              smoke      __graft_entry__.smoke()
#   bench        the default bench.py line (C2 headline, legs, cpu_baseline)    benchall   bench.py --config c1 | c3 | c4 | c5
#   prof         rocprofv3 --kernel-trace --stats of C2 and C4 -> kernel_stats_c{2,4}.{csv,txt}
#   calib        FETCH_SIZE / WRITE_SIZE calibration on known 1 GiB streams (tools/pmc/pmc_calib) -> pmc_calibration.json
#   pmc          FETCH / WRITE passes over C2 and C4 -> profiles-ready <tag>[_c4]_pmc_hbm_traffic.{csv,json} (+ bench lines quoting them)
#   pmctable     SQ / LDS / MFMA / TA / TCC counter table of the C2 step (separate --pmc passes)
#   ensemble     tools/loss_ensemble.py: both precision modes against the oracle ensemble of tests/golden/ensemble
#   mr2          bench.py --gpus 2 on this one GPU through gloo (the N > 1 code path; RCCL refuses two ranks on one device)
#   mr2torchrun  the same launched as the driver launches N > 1: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2
#   mr2fail      two RCCL ranks requested on this ONE-GPU box: every rung fails, the line must still appear (value null + ladder record)
#   mr2hang      the same with a test-only hang injected into the first rung of the supervised ladder
#   py:<script> [args, '+' for spaces]   any python script, output to <tag>_<script name>.txt      (e.g. py:tools/exp/foo.py+--x+1)
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-roofline --no-legs"
summ() { python - "$@" <<'PY'
import json,sys
for p in sys.argv[1:]:
    try:
        j=json.load(open(p)); r=j.get('roofline') or {}
        print('%-34s %.1f %s  gemm %.3f ms %.0f TF/s frac %.3f step_frac %.3f skipped %s stale %s %s' % (p.split('/')[-1], j['value'], j['unit'], r.get('gemm_ms_per_step', 0) or 0, r.get('achieved', 0) or 0, r.get('frac', 0) or 0, r.get('step_frac', 0) or 0, j['config'].get('skipped_steps'), r.get('traffic_stale'), {k: round(v['value'], 1) for k, v in (j.get('legs') or {}).items() if isinstance(v, dict) and 'value' in v}))
    except Exception as e: print(p, 'failed', e)
PY
}
for STAGE in "$@"; do
  echo "== $STAGE ($(date +%T))"
  case $STAGE in
    tests)
      timeout 1700 python -m pytest tests -m gpu -q -s --durations=8 > $O/${TAG}_gpu_tests.log 2>&1
      echo "pytest rc $?" >> $O/${TAG}_gpu_tests.log; tail -4 $O/${TAG}_gpu_tests.log ;;
    smoke)
      timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log ;;
    bench)
      timeout 600 python bench.py --steps 40 > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench.err; summ $O/${TAG}_bench_c2.json ;;
    benchall)
      for c in c1 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline > $O/${TAG}_bench_$c.json 2>> $O/${TAG}_bench.err; done
      summ $O/${TAG}_bench_c1.json $O/${TAG}_bench_c3.json $O/${TAG}_bench_c4.json $O/${TAG}_bench_c5.json ;;
    prof)
      for c in c2 c4; do
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$c -- python $R/bench.py --config $c --steps 20 $NOX > $O/${TAG}_prof_$c.log 2>&1)
        python tools/prof_summary.py $O/${TAG}_prof_$c 65 $O/${TAG}_kernel_stats_$c.csv 48 > $O/${TAG}_kernel_stats_$c.txt 2>&1
        find $O/${TAG}_prof_$c -name '*.db' -size +5M -delete 2>/dev/null
        head -30 $O/${TAG}_kernel_stats_$c.txt | cut -c1-220
      done ;;
    calib)
      [ -x tools/pmc/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc/pmc_calib.hip -o tools/pmc/pmc_calib
      (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_calib_fetch -- $R/tools/pmc/pmc_calib > $O/${TAG}_calib.log 2>&1)
      (cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_calib_write -- $R/tools/pmc/pmc_calib >> $O/${TAG}_calib.log 2>&1)
      python tools/pmc/calib_summary.py $O/${TAG}_calib_fetch $O/${TAG}_calib_write $O/${TAG}_pmc_calibration.json 2>&1 | tee $O/${TAG}_pmc_calibration.txt ;;
    pmc)
      CAL=""; [ -f $O/${TAG}_pmc_calibration.json ] && CAL="--calib $O/${TAG}_pmc_calibration.json"
      [ -z "$CAL" ] && ls profiles/r*_pmc_calibration.json >/dev/null 2>&1 && CAL="--calib $(ls profiles/r*_pmc_calibration.json | tail -1)"
      B="python $R/bench.py --steps 3 --warmup 2 --reps 1 $NOX --no-graph"
      for c in c2 c4; do
        (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_$c -- $B --config $c > $O/${TAG}_pmc_fetch_$c.log 2>&1)
        (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_$c -- $B --config $c > $O/${TAG}_pmc_write_$c.log 2>&1)
      done
      python tools/pmc_traffic.py $O/${TAG}_pmc_fetch_c2 $O/${TAG}_pmc_write_c2 ${TAG} "bench.py --steps 3 --warmup 2 --reps 1 --no-graph (C2: 1280x720, ViT-B/32, 190 cuts, -tf fast)" $CAL > $O/${TAG}_pmc_traffic.txt 2>&1
      python tools/pmc_traffic.py $O/${TAG}_pmc_fetch_c4 $O/${TAG}_pmc_write_c4 ${TAG}_c4 "bench.py --config c4 --steps 3 --warmup 2 --reps 1 --no-graph (C4: 3840x2160 DWT db3, ViT-B/16, 95 cuts)" $CAL > $O/${TAG}_pmc_traffic_c4.txt 2>&1
      cp profiles/${TAG}_pmc_hbm_traffic.* profiles/${TAG}_c4_pmc_hbm_traffic.* $O/ 2>/dev/null
      timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-legs > $O/${TAG}_bench_c2_with_traffic.json 2>> $O/${TAG}_bench.err
      timeout 300 python bench.py --config c4 --steps 30 --no-cpu-baseline --no-legs > $O/${TAG}_bench_c4_with_traffic.json 2>> $O/${TAG}_bench.err
      find $O -name '*counter_collection.csv' -size +30M -delete 2>/dev/null
      summ $O/${TAG}_bench_c2_with_traffic.json $O/${TAG}_bench_c4_with_traffic.json
      tail -n 3 $O/${TAG}_pmc_traffic.txt; tail -n 3 $O/${TAG}_pmc_traffic_c4.txt ;;
    pmctable)
      B="python $R/bench.py --steps 3 --warmup 2 --reps 1 $NOX --no-graph"
      (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmcA -- $B > $O/${TAG}_pmcA.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/${TAG}_pmcB -- $B > $O/${TAG}_pmcB.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/${TAG}_pmcC -- $B > $O/${TAG}_pmcC.log 2>&1)
      python tools/pmc_table.py $O/${TAG}_pmc_table.csv $O/${TAG}_pmcA $O/${TAG}_pmcB $O/${TAG}_pmcC > $O/${TAG}_pmc_table.txt 2>&1
      rm -rf $O/${TAG}_pmcA $O/${TAG}_pmcB $O/${TAG}_pmcC; head -14 $O/${TAG}_pmc_table.txt | cut -c1-250 ;;
    ensemble)
      timeout 1500 python tools/loss_ensemble.py $O/${TAG}_precision_ensemble > $O/${TAG}_precision_ensemble.log 2>&1; tail -25 $O/${TAG}_precision_ensemble.log ;;
    mr2)
      APH_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs > $O/${TAG}_mr_gloo2.json 2> $O/${TAG}_mr_gloo2.err
      echo "rc $?"; cut -c1-600 $O/${TAG}_mr_gloo2.json; tail -3 $O/${TAG}_mr_gloo2.err ;;
    mr2torchrun)
      # the driver's own launch line for N > 1 (torchrun), two ranks on this one GPU through gloo: the supervisors run under torchrun's agent
      APH_BENCH_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs --no-roofline > $O/${TAG}_mr_torchrun2.json 2> $O/${TAG}_mr_torchrun2.err
      echo "rc $?"; cut -c1-300 $O/${TAG}_mr_torchrun2.json; python -c "import json,sys; j=json.loads([l for l in open('$O/${TAG}_mr_torchrun2.json') if l.startswith('{')][-1]); print(j['config'].get('multi_rank_mode'), j['config'].get('multi_rank_ladder'), j['config'].get('params_identical_across_ranks'))"; tail -3 $O/${TAG}_mr_torchrun2.err ;;
    mr2fail)
      # every rung MUST fail here (two RCCL ranks requested on a one-GPU box: rank 1 has no device, RCCL refuses duplicates): the supervisors still print ONE line
      # (value null + the ladder's record) and exit non-zero -- the "cannot end without a line" path on real hardware
      APH_BENCH_RUNG_BUDGET=45 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs --no-roofline > $O/${TAG}_mr_fail2.json 2> $O/${TAG}_mr_fail2.err
      echo "rc $? (non-zero expected)"; grep "^{" $O/${TAG}_mr_fail2.json | cut -c1-1200 ;;
    mr2hang)
      # the ladder with a test-only hang injected into the first rung: the line must come from the second rung
      APH_BENCH_BACKEND=gloo APH_BENCH_INJECT_HANG=graph+rccl APH_BENCH_RUNG_BUDGET=75 timeout 500 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs --no-roofline > $O/${TAG}_mr_gloo2_hang.json 2> $O/${TAG}_mr_gloo2_hang.err
      echo "rc $?"; cut -c1-300 $O/${TAG}_mr_gloo2_hang.json; python -c "import json,sys; j=json.load(open('$O/${TAG}_mr_gloo2_hang.json')); print(j['config'].get('multi_rank_mode'), j['config'].get('multi_rank_ladder'))"; grep -c "INJECT_HANG" $O/${TAG}_mr_gloo2_hang.err ;;
    py:*)
      CMD=${STAGE#py:}; CMD=${CMD//+/ }; NAME=$(basename ${CMD%% *} .py)
      timeout 1500 python $CMD > $O/${TAG}_$NAME.txt 2>&1; echo "rc $?"; tail -40 $O/${TAG}_$NAME.txt | cut -c1-300 ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
find $O -name '*.db' -size +20M -delete 2>/dev/null
echo "== done ($(date +%T))"
