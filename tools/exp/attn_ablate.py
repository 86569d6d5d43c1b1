"""What bounds the T = 50 attention backward (attn_bwd_mfma_kernel<56>): the kernel against its memory skeleton (no products), its compute
skeleton (no stores) and its loads alone, at C2's shape -- needs a -DAPH_EXPERIMENTS build (aph_attn_set_ablate)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
assert L.experiments, 'python -m aphantasia_amd._build --experiments'
S, T, heads = 190, 50, 12
D = heads * 64
qkv = torch.randn(S * T, 3 * D, device='cuda').half()
datt = torch.randn(S * T, D, device='cuda').half()
att = torch.empty(S * T, D, dtype=torch.float16, device='cuda')
lse = torch.empty(S * heads * T, device='cuda'); delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)
st = _stream(qkv)
L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), None, None, None, S, T, heads, 0, st)
bwd = lambda: L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), ptr(datt), ptr(delta), ptr(dqkv), S, T, heads, 1, st)
rd, wr = (qkv.numel() + datt.numel()) * 2 + lse.numel() * 4, dqkv.numel() * 2
for rep in range(2):
    for mode, name, nbytes in ((0, 'the kernel', rd + wr), (1, 'no products (loads + staging + zero stores)', rd + wr), (2, 'no stores (loads + staging + products)', rd), (3, 'loads + staging only', rd)):
        L.cdll.aph_attn_set_ablate(mode)
        for _ in range(5): bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): bwd()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print('%-48s %6.1f us  %5.2f TB/s on %5.1f MB' % (name, us, nbytes / us / 1e6, nbytes / 1e6), flush=True)
L.cdll.aph_attn_set_ablate(0)
