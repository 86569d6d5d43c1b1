import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from aphantasia_amd.ops import ptr, _stream
L = ctypes.CDLL(os.path.join(HERE, 'mfma_rate.so'))
out = torch.empty(1024, device='cuda')
for label, src in [('zeros', torch.zeros(8192 * 8, device='cuda').half()), ('randn', torch.randn(8192 * 8, device='cuda').half())]:
    for nacc in (8, 32):          # 8: 16x16x32 with 32 accumulator tiles; 32: 32x32x16 with 8 accumulator tiles (128 registers both)
        for blocks in (64, 256):
            per_iter = nacc * 4 * 16384 if nacc != 32 else 8 * 32768
            iters = 40000 // 8
            f = lambda: L.mfma_rate(nacc, blocks, iters, ptr(src), ptr(out), _stream(out))
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            fl = blocks * 8.0 * iters * per_iter
            print('%-6s %s blocks %3d : %8.1f us  %7.1f TF total  %6.2f TF per block' % (label, '32x32x16' if nacc == 32 else '16x16x32', blocks, us, fl / us / 1e6, fl / us / 1e6 / blocks), flush=True)
