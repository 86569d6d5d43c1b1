"""Tile-configuration sweep of the GEMM core on the ViT shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
for (M, N, K) in [(9500, 3072, 768), (18715, 3072, 768), (3072, 3072, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M, N, device='cuda')
    st = _stream(A)
    line = '%5d x %5d x %5d :' % (M, N, K)
    for cfg in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 10)):
        f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M, N, K, ptr(C), cfg, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        n = 30 if M * N * K < 1e11 else 6
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        line += '   cfg%d %7.1f us %6.0f TF' % (cfg, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
    print(line)
