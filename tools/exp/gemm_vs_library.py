"""Round 5: calibration of the wave-specialised GEMM (vit_gemm_ws.h, the kernel behind roofline.frac) against the vendor library on
the SAME shapes, layout (A [M,K] x Bt [N,K]^T), f16 output + bias: torch.nn.functional.linear -> hipBLASLt / rocBLAS.  Not a product
path (the product never calls a library GEMM): it answers "what does the best available kernel reach on these shapes on this part"."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
SHAPES = []
for tag, M in (('B/32 190 cuts', 9500), ('B/16 190 cuts', 37430)):
    SHAPES += [(tag + ' qkv', M, 2304, 768), (tag + ' proj', M, 768, 768), (tag + ' fc1', M, 3072, 768), (tag + ' fc2', M, 768, 3072), (tag + ' dqkv', M, 768, 2304)]
SHAPES += [('16384 x 4096 x 4096', 16384, 4096, 4096), ('square 4096', 4096, 4096, 4096)]      # (bias staging of the kernel: N <= 4096)
NBUF, REPS = 4, 30
torch.manual_seed(0)


def timed(f):
    for i in range(4): f(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(REPS): f(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best


print('%-22s %6s %5s %5s | %-22s | %-22s | ratio' % ('shape', 'M', 'N', 'K', 'this repo (ws kernel)', 'library (F.linear f16)'))
for (name, M, N, K) in SHAPES:
    As = [torch.randn(M, K, device='cuda').half() for _ in range(NBUF)]
    B = (torch.randn(N, K, device='cuda') * 0.05).half()
    bias = torch.randn(N, device='cuda')
    bias_h = bias.half()
    out = torch.empty(M, N, device='cuda', dtype=torch.float16)
    st = _stream(out)
    mine = lambda i: L.call('aph_gemm_ws_probe', ptr(As[i % NBUF]), ptr(B), M, N, K, ptr(out), None, ptr(bias), 0, None, st)
    lib = lambda i: torch.nn.functional.linear(As[i % NBUF], B, bias_h)
    mine(0); torch.cuda.synchronize()
    ref = lib(0)
    err = (out.float() - ref.float()).abs().max().item()
    tm, tl = timed(mine), timed(lib)
    fl = 2.0 * M * N * K
    print('%-22s %6d %5d %5d | %7.1f us %6.0f TF/s | %7.1f us %6.0f TF/s | %.2f   (max |diff| %.3g)' % (name, M, N, K, tm * 1e3, fl / tm / 1e9, tl * 1e3, fl / tl / 1e9, tl / tm, err), flush=True)
    del As, B, out
