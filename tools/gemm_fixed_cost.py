"""Fixed cost of a GEMM launch: time vs k-tiles at the out-proj shape (9500 x 768, 228 tiles of 256x128 = one round), and the
back-to-back launch floor of this stream (a 1-tile GEMM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()


def t(M, N, K, cfg, n=60):
    A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M, N, device='cuda')
    st = _stream(A)
    f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M, N, K, ptr(C), cfg, st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('launch floor (64x128x64, 1 tile, cfg1): %.1f us' % t(64, 128, 64, 1, 200))
for cfg in (2, 4):
    N = 768 if cfg == 2 else 3072
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        print('cfg%d 9500 x %4d x %4d : %6.1f us' % (cfg, N, K, t(9500, N, K, cfg)), flush=True)
