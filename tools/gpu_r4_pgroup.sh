mkdir -p gpurun_out; O=gpurun_out
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wave_specialised" > $O/r04_pg_test.txt 2>&1; tail -3 $O/r04_pg_test.txt
for g in 1 4 2 8 1 4; do echo "APH_GEMM_WS_PGROUP=$g"; APH_GEMM_WS_PGROUP=$g timeout 100 python tools/gemm_shapes_bench.py 5 2>&1 | grep -v amdgpu; done > $O/r04_pg_shapes.txt 2>&1; cat $O/r04_pg_shapes.txt
export TMPDIR=/tmp
for g in 1 4; do for shape in "9500 3072 768" "9500 2304 768"; do
  n=$(echo $shape | tr ' ' x)
  (cd /tmp && APH_GEMM_WS_PGROUP=$g timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pg_pmc_${g}_$n -- python $GRAFT_REPO_ROOT/tools/gemm_one.py $shape > /dev/null 2>&1)
  python - $O/pg_pmc_${g}_$n $g "$shape" <<'PY'
import csv, glob, sys
csv.field_size_limit(1 << 30)
tot = n = 0
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'gemm_ws' in r['Kernel_Name']:
            tot += float(r['Counter_Value']); n += 1
print('PGROUP=%s shape %s: fetch %.1f MB / launch (2 x FETCH_SIZE KiB) over %d launches' % (sys.argv[2], sys.argv[3], 2 * 1024 * tot / max(n, 1) / 1e6, n))
PY
done; done 2>&1 | tee $O/r04_pg_fetch.txt
rm -rf $O/pg_pmc_*
for g in 1 4 1 4; do APH_GEMM_WS_PGROUP=$g timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-legs > $O/r04_pg_bench_$g.json 2>$O/r04_pg_bench_$g.err; python -c "
import json; j=json.load(open('$O/r04_pg_bench_$g.json')); r=j['roofline']; print('PGROUP=$g', round(j['value'],1), 'steps/s gemm', round(r['gemm_ms_per_step'],3), 'ms', round(r['achieved']), 'TF/s frac', round(r['frac'],3), 'skipped', j['config'].get('skipped_steps'))"; done 2>&1 | tee $O/r04_pg_bench.txt
