cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python bench.py --steps 40 > $O/r06_bench_c2.json 2>> $O/r06_bench.err
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-legs > $O/r06_bench_c2_with_traffic.json 2>> $O/r06_bench.err
timeout 300 python bench.py --config c4 --steps 30 --no-cpu-baseline --no-legs > $O/r06_bench_c4_with_traffic.json 2>> $O/r06_bench.err
for c in c1 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline > $O/r06_bench_$c.json 2>> $O/r06_bench.err; done
python - <<'PY'
import json
for f in ('c2','c2_with_traffic','c4_with_traffic','c1','c3','c4','c5'):
    j=json.load(open('gpurun_out/r06_bench_%s.json'%f)); r=j.get('roofline') or {}
    print(f, round(j['value'],1), r.get('bound'), r.get('binding_floor'), round(r.get('frac',0),3), r.get('traffic_stale'))
PY
