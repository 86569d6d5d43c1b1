import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from aphantasia_amd.ops import ptr, _stream
L = ctypes.CDLL(os.path.join(HERE, 'gemm_ablate.so'))
for (M, N, K) in [(9500, 768, 3072), (9500, 3072, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half()
    out = torch.empty(((M + 255) // 256) * (N // 128) * 512, device='cuda')
    for mode, name in [(0, 'full'), (1, 'no-DMA (compute+barriers)'), (2, 'DMA+barriers only'), (3, 'MFMA only'), (4, 'full, staggered DMA issue')]:
        f = lambda: L.ablate(mode, ptr(A), ptr(B), M, N, K, ptr(out), _stream(A))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%5dx%5dx%5d %-28s %8.1f us  (%6.1f TF/s-equivalent)' % (M, N, K, name, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
