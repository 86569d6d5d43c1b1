"""Round 3: the ViT input-gradient handed to the sampler adjoint as f16 (aph_vit_backward_h, carrying the loss scale) vs f32: step time at
C2 (both -tf) and what it does to the free-running loss curves against the oracle fixtures."""
import os, sys, time, warnings, importlib.util
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
spec = importlib.util.spec_from_file_location('lc', os.path.join(ROOT, 'tools', 'loss_curve.py')); lc = importlib.util.module_from_spec(spec); spec.loader.exec_module(lc)
h, w = 720, 1280
def mk(S, tf, f16):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
    torch.manual_seed(0); np.random.seed(0)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    trf = transforms.transforms_fast if tf == 'fast' else transforms.normalize()
    return Engine(leaf, h, w, model, S, [(tgt, -1.0)], sim='mix', transform=trf, grad_f16=f16)
def run(e, n=30):
    for _ in range(6): e.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for tf, S in (('fast', 190), ('none', 200)):
    for rep in range(2):
        for f16 in (False, True):
            e = mk(S, tf, f16); ms = run(e)
            print('-tf %s S=%d patch gradient %s: %.3f ms/step (%.1f steps/s), skipped %d' % (tf, S, 'f16' if f16 else 'f32', ms, 1e3 / ms, int(e.guard[0])), flush=True)
            del e
for name in ('c2_s200', 'c2_s32', 'c2_s32_stress'):
    for f16 in (False, True):
        mx, first, rms, _ = lc.run_fixture(name, grad_f16=f16)
        print('%s patch gradient %s: max |d loss| %.2e, first step past 1e-3: %s, block-mean RMS %.4f' % (name, 'f16' if f16 else 'f32', mx, first, rms), flush=True)
