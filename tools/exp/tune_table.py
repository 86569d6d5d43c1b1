"""Experiment: best tile configuration per (M, N, K) of the ViT-B linears at shard sizes (1, 2, 4, 8 ranks of C2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
for M in (1200, 2400):
    for (N, K) in ((768, 768), (2304, 768), (3072, 768), (768, 2304), (768, 3072)):
        A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M, N, device='cuda')
        line = 'M %5d N %5d K %5d:' % (M, N, K)
        best = None
        for cfg in (0, 1, 2, 4, 10):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                st = _stream(A)
                f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M, N, K, ptr(C), cfg, st)
                for _ in range(3): f()
                s.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(30): f()
                e1.record(s); s.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            line += '  cfg%d %6.1f' % (cfg, us)
            if cfg and (best is None or us < best[1]): best = (cfg, us)
        print(line + '   best cfg%d' % best[0])
