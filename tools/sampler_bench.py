"""Sampler kernels alone at the headline size (1280x720, 190 cuts, ViT-B/32 patch layout): forward / adjoint timings for
-tf none and -tf fast, HIP events around the C-ABI calls.  APH_SAMPLER_DBG=<bits> selects ablation variants (experiments).
    python tools/sampler_bench.py [S] [H] [W]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import _ffi, ops, transforms
from aphantasia_amd.utils import draw_crop_params_bulk
S = int(sys.argv[1]) if len(sys.argv) > 1 else 190
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
dev = 'cuda'
rng = np.random.default_rng(0)
img = torch.rand(3, H, W, device=dev)
geom = ops.make_geom(H, W, S, 224, 32)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, trf in (('none', transforms.normalize()), ('fast', transforms.transforms_fast)):
    table, aug = draw_crop_params_bulk(S, 224, H, W, 'uniform', 0.4, trf, rng)
    tb = torch.from_numpy(table).to(dev)
    ag = torch.from_numpy(aug).to(dev) if aug is not None else None
    ws = ops.sample_ws(geom, ag is not None, dev)
    out = torch.empty(S * 49, 3072, dtype=torch.float16, device=dev)
    g = torch.randn(S * 49, 3072, device=dev)
    grgb = torch.empty(3, H, W, device=dev)
    f = lambda: ops.sample_fwd(geom, img, tb, ag, ws, out, _ffi.APH_OUT_PATCH_F16)
    b = lambda: ops.sample_bwd(geom, g, tb, ag, ws, grgb, _ffi.APH_OUT_PATCH_F16)
    tb_new = timeit(b)
    prev = _ffi.lib().cdll.aph_crop_adjoint_set_gather(1)
    tb_old = timeit(b)
    _ffi.lib().cdll.aph_crop_adjoint_set_gather(prev)
    print('-tf %-5s S=%d %dx%d dbg=%s: forward %7.1f us   adjoint %7.1f us  (with the round-2 gather crop adjoint: %7.1f us)' % (name, S, W, H, os.environ.get('APH_SAMPLER_DBG', '0'), timeit(f), tb_new, tb_old), flush=True)
