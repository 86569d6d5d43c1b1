"""Does the operand row pitch matter (L2 channel aliasing)?  Same GEMM with padded leading dimensions."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
for (M, N, K) in [(9500, 768, 3072), (9500, 3072, 768), (9500, 768, 768), (8192, 8192, 8192)]:
    for pad in (0, 8, 32, 64, 72):
        A = torch.randn(M, K + pad, device='cuda').half()
        B = torch.randn(N, K + pad, device='cuda').half()
        C = torch.empty(M, N, device='cuda')
        st = _stream(A)
        f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K + pad, ptr(B), K + pad, M, N, K, ptr(C), 0, st)
        for _ in range(5): f()
        torch.cuda.synchronize()
        n = 30 if M * N * K < 1e11 else 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print('%5d x %5d x %5d pad %3d : %8.1f us %7.1f TF/s' % (M, N, K, pad, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
