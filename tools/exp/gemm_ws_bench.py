"""Round 3: the wave-specialised persistent GEMM (test hook tile_cfg 5, vit_gemm_ws.h) against the round-2 kernels (2 = 256x128
ring, 4 = 256x256 phased) on the ViT-B/32 shapes at full batch; correctness against an fp32 matmul and bitwise
reproducibility first (the emulator cannot see vmcnt under-waits)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
M = int(os.environ.get('M', 9500))
CFGS = [int(c, 0) for c in os.environ.get('CFGS', '2,4,5,0x105').split(',')]
SHAPES = [('qkv', M, 2304, 768), ('outproj', M, 768, 768), ('fc1/dfc2', M, 3072, 768), ('fc2/dfc1', M, 768, 3072), ('dqkv', M, 768, 2304)]
torch.manual_seed(0)
for (name, M_, N, K) in SHAPES:
    A = torch.randn(M_, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half()
    st = _stream(A)
    want = A.float() @ B.float().T
    outs = []
    for rep in range(3):
        C = torch.full((M_, N), float('nan'), device='cuda')
        L.call('aph_gemm_f16_ld', ptr(A), K, ptr(B), K, M_, N, K, ptr(C), int(os.environ.get('CHECK_CFG', 5)), st)
        torch.cuda.synchronize()
        outs.append(C)
    err = (outs[0] - want).abs().max().item()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    C = torch.empty(M_, N, device='cuda')
    line = '%-9s %5d x %5d x %5d : check-cfg max err %.2e (tol %.2e) bitwise-repro %s |' % (name, M_, N, K, err, 2e-3 * (K / 64) ** 0.5, same)
    for cfg in CFGS:
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
        line += '  cfg%d%s %6.1f us %5.0f TF' % (cfg & 0xff, '-ns' if cfg & 0x100 else '', ms * 1e3, 2.0 * M_ * N * K / ms / 1e9)
    print(line, flush=True)
