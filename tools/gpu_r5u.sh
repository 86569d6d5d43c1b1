#!/bin/bash
# round 5: C1 (one cut) with the split-K small-M kernel also on the K = 2304 / 3072 GEMMs (rs=3) against the default (rs=1), alternating
# (record of a rejected experiment: mode 3 of aph_gemm_set_rs existed for this run only and was removed again -- today the call clamps 3 to 2;
# profiles/r05_c1_long_k_ab.txt has the numbers)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
  for m in 1 3; do
    echo -n "rs=$m: "; timeout 200 python bench.py --config c1 --steps 60 --no-cpu-baseline --no-roofline --no-legs --vit-path rs=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['steps_per_s'], 'loss', d.get('loss'))"
  done
done
} > $O/r05u_c1_rs3.txt 2>&1
cat $O/r05u_c1_rs3.txt
