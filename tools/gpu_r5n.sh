#!/bin/bash
# round 5: one-kernel blocked attention backward (ViT-B/16) -- tests, kernel alone old / new, C4 and C3 lines old / new
TAG=${1:-r05n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention or vit" > $O/${TAG}_tests.log 2>&1
echo "pytest rc $?" >> $O/${TAG}_tests.log; tail -n 3 $O/${TAG}_tests.log
python - > $O/${TAG}_attn_bench.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
for (S, T, heads) in ((95, 197, 12), (43, 197, 12), (24, 197, 12), (95, 82, 12), (95, 135, 12)):
    D = heads * 64
    qkv = torch.randn(S * T, 3 * D, device='cuda').half()
    datt = torch.randn(S * T, D, device='cuda').half()
    att = torch.empty(S * T, D, dtype=torch.float16, device='cuda')
    lse = torch.empty(S * heads * T, device='cuda'); delta = torch.empty_like(lse)
    st = _stream(qkv)
    L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), None, None, None, S, T, heads, 0, st)
    outs = []
    for one in (0, 1, 0, 1):
        L.cdll.aph_attn_set_bwd_one(one)
        dqkv = torch.zeros_like(qkv)
        bwd = lambda: L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), ptr(datt), ptr(delta), ptr(dqkv), S, T, heads, 1, st)
        for _ in range(5): bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): bwd()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        outs.append(dqkv.float().clone())
        print('S=%3d T=%3d backward %s: %7.1f us' % (S, T, 'ONE kernel' if one else 'dQ + dK/dV pair', us), flush=True)
    d = (outs[0] - outs[1]).abs().max().item()
    print('   max |pair - one| = %.3e (max |dqkv| %.3e); one-kernel bitwise repeatable: %s' % (d, outs[0].abs().max().item(), bool(torch.equal(outs[1], outs[3]))), flush=True)
L.cdll.aph_attn_set_bwd_one(1)
PY
cat $O/${TAG}_attn_bench.txt | grep -v amdgpu
for c in c4 c3; do
  timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$c (split precision) one-kernel attention backward: %.2f steps/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
"
done | tee $O/${TAG}_bench.txt
