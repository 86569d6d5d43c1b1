"""Experiment: fixed vs per-k-tile cost of the GEMM configs at shard-sized M."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
g = torch.cuda.CUDAGraph()
for M in (190, 1200, 2400):
  for N in (768, 3072):
    for K in (768, 2304, 3072):
        A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M, N, device='cuda')
        line = 'M %5d N %5d K %5d:' % (M, N, K)
        for cfg in (1, 8, 9, 10):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                st = _stream(A)
                f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M, N, K, ptr(C), cfg, st)
                for _ in range(3): f()
                s.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(50): f()
                e1.record(s); s.synchronize()
            line += '  cfg%d %6.1f us' % (cfg, e0.elapsed_time(e1) / 50 * 1e3)
        print(line)
