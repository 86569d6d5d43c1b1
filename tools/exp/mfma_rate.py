import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from aphantasia_amd.ops import ptr, _stream
L = ctypes.CDLL(os.path.join(HERE, 'mfma_rate.so'))
out = torch.empty(1024, device='cuda')
for label, src in [('zeros', torch.zeros(8192 * 8, device='cuda').half()), ('ones', torch.ones(8192 * 8, device='cuda').half()),
                   ('randn', torch.randn(8192 * 8, device='cuda').half())]:
    for nacc in (4, 8):
        for blocks in (64, 256):
            iters = 40000 // nacc
            f = lambda: L.mfma_rate(nacc, blocks, iters, ptr(src), ptr(out), _stream(out))
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            fl = blocks * 8.0 * iters * nacc * 4 * 16384
            print('%-6s acc tiles %2d blocks %3d : %8.1f us  %7.1f TF total  %6.2f TF per block' % (label, nacc * 4, blocks, us, fl / us / 1e6, fl / us / 1e6 / blocks))
