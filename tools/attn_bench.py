"""The attention kernels alone at C2's shape (190 cuts x 12 heads, T = 50) and C4's (95 cuts, T = 197): microseconds per launch and
the HBM rate on the algorithmic bytes (forward: qkv in, att + lse out; backward: qkv, att, datt, lse in, dqkv out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
for (S, T, heads) in ((190, 50, 12), (95, 197, 12)):
    D = heads * 64
    qkv = torch.randn(S * T, 3 * D, device='cuda').half()
    datt = torch.randn(S * T, D, device='cuda').half()
    att = torch.empty(S * T, D, dtype=torch.float16, device='cuda')
    lse = torch.empty(S * heads * T, device='cuda'); delta = torch.empty_like(lse)
    dqkv = torch.empty_like(qkv)
    st = _stream(qkv)
    fwd = lambda: L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), None, None, None, S, T, heads, 0, st)
    bwd = lambda: L.call('aph_attn_test', ptr(qkv), ptr(att), ptr(lse), ptr(datt), ptr(delta), ptr(dqkv), S, T, heads, 1, st)
    for name, f, nbytes in (('forward', fwd, qkv.numel() * 2 + att.numel() * 2 + lse.numel() * 4),
                            ('backward', bwd, 2 * qkv.numel() * 2 + 2 * att.numel() * 2 + lse.numel() * 4)):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print('S=%3d T=%3d %-8s %7.1f us  %5.2f TB/s on %5.1f MB' % (S, T, name, us, nbytes / us / 1e6, nbytes / 1e6), flush=True)
