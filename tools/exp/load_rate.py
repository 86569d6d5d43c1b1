"""Experiment: per-CU load rates by access pattern / waves per CU / loads in flight (tools/exp/load_rate.hip).  Needs a GPU."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, 'load_rate.so'))
L.load_rate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
src = torch.randint(0, 2 ** 31 - 1, (256 * 1024 * 1024 + 262144,), device='cuda', dtype=torch.int32)       # 1 GiB
out = torch.zeros(1024 * 1024, device='cuda', dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
names = ['DMA 8x128B', 'DMA 16x64B', 'VGPR contiguous', 'VGPR fragment', 'VGPR 8x128B']
iters = 512
for label, region, G in (('512 KB region per XCD (L2-hot, shared)', 512 * 1024, 8), ('4 MB region per workgroup-slot x 256 (1 GiB: HBM)', 4 * 1024 * 1024, 256),
                         ('256 KB region x 256 (64 MB: MALL)', 256 * 1024, 256)):
    print('== ' + label)
    for mode in range(5):
        for threads in (256, 512, 1024):
            line = '%-16s %2d waves/CU:' % (names[mode], threads // 64)
            for U in (4, 8, 16):
                def f():
                    return L.load_rate(mode, U, src.data_ptr(), region, G, 2048, iters, 256, threads, out.data_ptr(), st)
                rc = f()
                if rc != 0:
                    line += '   U=%-2d   --    ' % U
                    continue
                for _ in range(2): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): f()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 5 * 1e3
                byts = (threads // 64) * iters * 1024
                line += '   U=%-2d %6.1f GB/s' % (U, byts / us / 1e3)
            print(line + '   (per CU; x256 CUs)', flush=True)
