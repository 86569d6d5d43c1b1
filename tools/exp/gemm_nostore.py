"""Experiment (round 3): the ViT GEMM shapes with and without their output stores (test hook tile_cfg | 0x100): how much
of a launch is the store phase, i.e. the most an overlapped epilogue could gain.  fp32 test epilogue (2x the bytes of the
f16 epilogues the ViT uses)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
M = int(os.environ.get('M', 9500))
SHAPES = [('qkv', M, 2304, 768), ('outproj', M, 768, 768), ('fc1/dfc2', M, 3072, 768), ('fc2/dfc1', M, 768, 3072), ('dqkv', M, 768, 2304)]
for (name, M_, N, K) in SHAPES:
    A = torch.randn(M_, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half(); C = torch.empty(M_, N, device='cuda')
    st = _stream(A)
    line = '%-9s %5d x %5d x %5d :' % (name, M_, N, K)
    for cfg in (2, 2 | 0x100, 4, 4 | 0x100):
        if (cfg & 0xff) == 4 and N % 256:
            continue
        f = lambda: L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M_, N, K, ptr(C), cfg, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        line += '  cfg%d%s %6.1f us %5.0f TF' % (cfg & 0xff, '-nostore' if cfg & 0x100 else '        ', ms * 1e3, 2.0 * M_ * N * K / ms / 1e9)
    print(line, flush=True)
