#!/bin/bash
# Full validation of one build (round 5): GPU test suite (-s: printed parity numbers), smoke, bench line per configuration, kernel-trace summaries (C2, C4),
# FETCH / WRITE PMC passes for C2 (GEMM family) and C4 (irDWT) -> profiles-ready files under gpurun_out/
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $O/${TAG}_gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_gpu_tests.log; tail -4 $O/${TAG}_gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
timeout 500 python bench.py --steps 40 > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench.err
for c in c1 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline > $O/${TAG}_bench_$c.json 2>> $O/${TAG}_bench.err; done
export TMPDIR=/tmp
for c in c2 c4; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$c -- python $R/bench.py --config $c --steps 20 --no-cpu-baseline --no-roofline --no-legs > $O/${TAG}_prof_$c.log 2>&1)
  python tools/prof_summary.py $O/${TAG}_prof_$c 65 $O/${TAG}_kernel_stats_$c.csv 48 > $O/${TAG}_kernel_stats_$c.txt 2>&1
done
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-legs --no-graph"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $B > $O/${TAG}_pmc_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $B > $O/${TAG}_pmc_write.log 2>&1)
python tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write 5 ${TAG} "bench.py --steps 3 --warmup 2 --no-graph (C2: 1280x720, ViT-B/32, 190 cuts, -tf fast)" > $O/${TAG}_pmc_traffic.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_c4 -- $B --config c4 > $O/${TAG}_pmc_fetch_c4.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_c4 -- $B --config c4 > $O/${TAG}_pmc_write_c4.log 2>&1)
python tools/pmc_traffic.py $O/${TAG}_pmc_fetch_c4 $O/${TAG}_pmc_write_c4 5 ${TAG}_c4 "bench.py --config c4 --steps 3 --warmup 2 --no-graph (C4: 3840x2160 DWT db3, ViT-B/16, 95 cuts)" > $O/${TAG}_pmc_traffic_c4.txt 2>&1
cp profiles/${TAG}_pmc_hbm_traffic.* profiles/${TAG}_c4_pmc_hbm_traffic.* $O/ 2>/dev/null
# second bench lines, now quoting the traffic of THIS build
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-legs > $O/${TAG}_bench_c2_with_traffic.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --config c4 --steps 30 --no-cpu-baseline --no-legs > $O/${TAG}_bench_c4_with_traffic.json 2>> $O/${TAG}_bench.err
find $O -name '*counter_collection.csv' -size +30M -delete 2>/dev/null
find $O -name '*.db' -size +5M -delete 2>/dev/null
for f in $O/${TAG}_bench_c*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline'] or {}
    print('%-34s %.1f steps/s gemm %.3f ms %.0f TF/s frac %.3f step_frac %.3f skipped %s stale %s %s' % (sys.argv[1].split('/')[-1], j['value'], r.get('gemm_ms_per_step', 0), r.get('achieved', 0), r.get('frac', 0), r.get('step_frac', 0), j['config'].get('skipped_steps'), r.get('traffic_stale'), {k: round(v['value'], 1) for k, v in (j.get('legs') or {}).items() if isinstance(v, dict) and 'value' in v}))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
tail -n 3 $O/${TAG}_pmc_traffic.txt; tail -n 3 $O/${TAG}_pmc_traffic_c4.txt
