"""Experiment: cost of the epilogue stores (cfg4 vs cfg6 = same kernel, stores predicated off) vs number of active tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
N, K = 3072, 768
for M in (256, 1024, 2048, 4096, 5376, 9500):
    A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M, N, device='cuda')
    st = _stream(A)
    line = 'M %5d (%3d tiles):' % (M, (M + 255) // 256 * 12)
    for cfg in (4, 6, 2, 7):
        f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M, N, K, ptr(C), cfg, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        line += '  cfg%d %6.1f us' % (cfg, e0.elapsed_time(e1) / 30 * 1e3)
    print(line)
