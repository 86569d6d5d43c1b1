import ctypes, os, sys, torch
HERE = '/root/repo/tools/exp'
L = ctypes.CDLL(os.path.join(HERE, 'load_rate.so'))
L.load_rate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
src = torch.randint(0, 2 ** 31 - 1, (256 * 1024 * 1024 + 262144,), device='cuda', dtype=torch.int32)
out = torch.zeros(1024 * 1024, device='cuda', dtype=torch.int32)
flush = torch.empty(1 << 28, device='cuda', dtype=torch.int32)     # 1 GiB: pushed through the caches between launches ("cold" rows)
st = torch.cuda.current_stream().cuda_stream
names = ['DMA 8x128B', 'DMA 16x64B', 'VGPR contiguous', 'VGPR fragment', 'VGPR 8x128B']
iters = 128       # KiB per wave per launch: a GEMM-sized stream (512 KB per 4-wave workgroup)
for label, region, G in (('4 MB region, ONE region for all workgroups (every XCD pulls the same lines)', 4 << 20, 1),
                         ('4 MB region per XCD (8 regions, 32 MB)', 4 << 20, 8),
                         ('512 KB region, ONE for all', 512 << 10, 1), ('512 KB region per XCD', 512 << 10, 8)):
    for cold in (0, 1):
        print('== %s%s' % (label, ' -- caches flushed before every launch' if cold else ' -- warm'))
        for mode in (0, 2, 4):
            for threads in (256, 512):
                line = '%-16s %2d waves/CU:' % (names[mode], threads // 64)
                for U in (8,):
                    f = lambda: L.load_rate(mode, U, src.data_ptr(), region, G, 2048, iters, 256, threads, out.data_ptr(), st)
                    f(); torch.cuda.synchronize()
                    tot = 0.0
                    n = 6
                    for _ in range(n):
                        if cold:
                            flush.add_(1)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); f(); e1.record(); torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1)
                    us = tot / n * 1e3
                    byts = (threads // 64) * iters * 1024
                    line += '   U=%-2d %7.1f us/launch %6.1f GB/s per CU' % (U, us, byts / us / 1e3)
                print(line, flush=True)
