import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from aphantasia_amd.ops import ptr, _stream
L = ctypes.CDLL(os.path.join(HERE, 'dma_rate.so'))
K, rows, nk = 8192, 16384, 128          # 16384 x 8192 halfs = 268 MB source (beyond L2, within MALL)
A = torch.randn(rows, K, device='cuda').half()
out = torch.empty(512 * 512, device='cuda')
for rows_used, label in [(16384, 'source 268 MB'), (512, 'source 8 MB (L2/MALL resident)')]:
    for mode in (0, 1):
        for blocks in (8, 32, 64, 128, 256):
            f = lambda: L.dma_rate(mode, ptr(A), K, rows_used, nk, blocks, ptr(out), _stream(A))
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            byts = blocks * nk * 48 * 1024
            print('%-32s %-12s blocks %3d : %7.1f us  %6.1f GB/s per CU  %6.2f TB/s total' % (label, ['LDS-DMA', 'via VGPR'][mode], blocks, us, byts / blocks / us / 1e3, byts / us / 1e6))
