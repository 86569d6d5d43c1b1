"""Experiment: two CU-masked streams running half-batch GEMM chains concurrently vs one full-GPU stream."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr
L = _ffi.lib()
hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return s

def chain(M, st, reps, bufs, swap=False):
    A, B1, B2, C1, C2 = bufs
    for _ in range(reps):
        if swap: L.call('aph_gemm_f16_ld', ptr(A), 768, ptr(B2), 768, M, 768, 768, ptr(C2), 0, st)
        L.call('aph_gemm_f16_ld', ptr(A), 768, ptr(B1), 768, M, 3072, 768, ptr(C1), 0, st)
        if not swap: L.call('aph_gemm_f16_ld', ptr(A), 768, ptr(B2), 768, M, 768, 768, ptr(C2), 0, st)

def mk(M):
    return (torch.randn(M, 768, device='cuda').half(), torch.randn(3072, 768, device='cuda').half(), torch.randn(768, 768, device='cuda').half(),
            torch.empty(M, 3072, device='cuda'), torch.empty(M, 768, device='cuda'))

full = mk(9500); ha = mk(4750); hb = mk(4750)
torch.cuda.synchronize()
s0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, wa, wb in [('alt-bits', [0x55555555] * 8, [0xAAAAAAAA] * 8), ('lo/hi', [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
                     ('alt-words', [0xFFFFFFFF, 0] * 4, [0, 0xFFFFFFFF] * 4), ('none', [0xFFFFFFFF] * 8, [0xFFFFFFFF] * 8)]:
    sa, sb = masked_stream(wa), masked_stream(wb)
    for it in range(2):
        chain(9500, s0, 20, full); torch.cuda.synchronize()
        t0 = time.perf_counter(); chain(9500, s0, 20, full); torch.cuda.synchronize(); tf = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(20):
            chain(4750, sa, 1, ha); chain(4750, sb, 1, hb, True)
        hip.hipStreamSynchronize(sa); hip.hipStreamSynchronize(sb); th = time.perf_counter() - t0
        t0 = time.perf_counter(); chain(4750, sa, 20, ha); hip.hipStreamSynchronize(sa); t1 = time.perf_counter() - t0
    print('%-10s full-GPU 20x(M=9500): %.3f ms | two masked streams 20x(M=4750) each: %.3f ms | one masked stream alone: %.3f ms' % (name, tf * 1e3, th * 1e3, t1 * 1e3))
