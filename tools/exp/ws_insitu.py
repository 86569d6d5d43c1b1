"""Round 3: whole C2 step with the wave-specialised GEMM switched on from N tiles (aph_gemm_set_ws_min_tiles) vs off, same process,
alternating; plus the tiny ViT parity check with it forced on."""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aphantasia_amd import clip as aclip, transforms, _ffi
from aphantasia_amd.engine import Engine
import kernel_checks as K
L = _ffi.lib()
prev = L.cdll.aph_gemm_set_ws_min_tiles(1)
print('tiny ViT with the ws kernel forced: fwd/bwd rel err', K.check_vit(None, 'cuda'))
cfg = dict(input_resolution=64, patch_size=16, width=256, layers=2, heads=4, output_dim=128)
print('T=17, S=40 (ragged, 3 row tiles):', K.check_vit(None, 'cuda', cfg, S=40))
L.cdll.aph_gemm_set_ws_min_tiles(prev)
S, h, w = int(os.environ.get('S', 190)), 720, 1280
steps = 30
def mk():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
    torch.manual_seed(0); np.random.seed(0)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    return Engine(leaf, h, w, model, S, [(tgt, -1.0)], sim='mix', transform=transforms.transforms_fast, use_graph=True)
def run(e, n):
    for _ in range(6): e.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for thr in [int(v) for v in os.environ.get('THR', '0,160,100,0,160').split(',')]:
    L.cdll.aph_gemm_set_ws_min_tiles(thr)
    e = mk()
    ms = run(e, steps)
    print('ws_min_tiles %4d : %.3f ms/step  %.1f steps/s  loss %.5f skipped %d' % (thr, ms, 1e3 / ms, float(e.loss), int(e.guard[0])), flush=True)
    del e
