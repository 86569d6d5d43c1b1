"""Round 3: per-level time of the inverse DWT and its adjoint at C4's size (3840x2160 db3, 11 levels), HIP events around each C-ABI call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import ops
from aphantasia_amd.dwt import DWTSynth
h, w = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (2160, 3840)))
syn = DWTSynth(h, w, 'db3', 0.3, 'cuda')
flat = torch.randn(syn.numel, device='cuda') * 0.01
grad = torch.empty_like(flat)
graw = torch.randn(3, syn.H, syn.W, device='cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('all levels: forward %.1f us, adjoint %.1f us' % (timeit(lambda: syn.forward(flat)), timeit(lambda: syn.backward(graw, grad))))
ys = syn.views(flat); gs = syn.views(grad)
st = ops._stream(flat)
ll, llh, llw = ys[0], syn.sizes[-1][0], syn.sizes[-1][1]
for j in range(syn.J - 1, -1, -1):
    hh, ww = syn.sizes[j]
    f = lambda: syn.lib.call('aph_idwt_level_fwd', ops.ptr(ll), llh, llw, ops.ptr(ys[1 + j]), hh, ww, syn.C, ops.ptr(syn.g0), ops.ptr(syn.g1), syn.L, float(syn.scale[j]), ops.ptr(syn.bufs[j]), st)
    ho, wo = syn.out_sizes[j]
    by = 4.0 * 3 * (4 * hh * ww + ho * wo)
    us = timeit(f)
    print('fwd level %2d: in %4dx%4d -> out %4dx%4d  %7.1f us  %6.2f TB/s (%.1f MB)' % (j, hh, ww, ho, wo, us, by / us / 1e6, by / 1e6))
    ll, (llh, llw) = syn.bufs[j], syn.out_sizes[j]
g = graw
for j in range(syn.J):
    hh, ww = syn.sizes[j]
    if j + 1 < syn.J: dst, (llh, llw) = syn.gbufs[j + 1], syn.out_sizes[j + 1]
    else: dst, (llh, llw) = gs[0], syn.sizes[-1]
    f = lambda: syn.lib.call('aph_idwt_level_bwd', ops.ptr(g), hh, ww, syn.C, ops.ptr(syn.g0), ops.ptr(syn.g1), syn.L, float(syn.scale[j]), ops.ptr(dst), llh, llw, ops.ptr(gs[1 + j]), st)
    ho, wo = syn.out_sizes[j]
    by = 4.0 * 3 * (4 * hh * ww + ho * wo)
    us = timeit(f)
    print('adj level %2d: %7.1f us  %6.2f TB/s (%.1f MB)' % (j, us, by / us / 1e6, by / 1e6))
    g = dst
