"""Round 3: whole C2 step with the separable row-block crop adjoint (default on frames without wrap padding) vs the round-2 gather kernel
(aph_crop_adjoint_set_gather), same process, alternating."""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aphantasia_amd import clip as aclip, transforms, _ffi
from aphantasia_amd.engine import Engine
L = _ffi.lib()
S, h, w = 190, 720, 1280
def mk(tf):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
    torch.manual_seed(0); np.random.seed(0)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    return Engine(leaf, h, w, model, S if tf == 'fast' else 200, [(tgt, -1.0)], sim='mix', transform=transforms.transforms_fast if tf == 'fast' else transforms.normalize())
def run(e, n=30):
    for _ in range(6): e.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for tf in ('fast', 'none'):
    for gather in (1, 0, 1, 0):
        L.cdll.aph_crop_adjoint_set_gather(gather)
        e = mk(tf); ms = run(e)
        print('-tf %s crop adjoint %s: %.3f ms/step (%.1f steps/s) loss %.5f' % (tf, 'gather (r2)' if gather else 'row-block (r3)', ms, 1e3 / ms, float(e.loss)), flush=True)
        del e
