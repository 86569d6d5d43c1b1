"""GEMM core micro-benchmark on the ViT's shapes (HIP events, preallocated output, random operands)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
dev = 'cuda'
shapes = [(9500, 768, 768), (9500, 2304, 768), (9500, 3072, 768), (9500, 768, 3072), (9500, 768, 2304), (9310, 768, 3072),
          (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev)
    st = _stream(A)
    for _ in range(5):
        L.call('aph_gemm_f16', ptr(A), ptr(B), M, N, K, ptr(C), st)
    torch.cuda.synchronize()
    n = 50 if M * N * K < 1e11 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        L.call('aph_gemm_f16', ptr(A), ptr(B), M, N, K, ptr(C), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('%5d x %5d x %5d : %8.1f us  %7.1f TF/s' % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
print('--- per-rank shard shapes (24 / 48 cuts) ---')
for (M, N, K) in [(1200, 768, 768), (1200, 2304, 768), (1200, 3072, 768), (1200, 768, 3072), (2400, 768, 768), (2400, 3072, 768), (2400, 768, 3072)]:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev)
    st = _stream(A)
    for _ in range(5):
        L.call('aph_gemm_f16', ptr(A), ptr(B), M, N, K, ptr(C), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.call('aph_gemm_f16', ptr(A), ptr(B), M, N, K, ptr(C), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print('%5d x %5d x %5d : %8.1f us  %7.1f TF/s' % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
