"""debug: FrameLoop with / without hipGraph replay -- per-frame step size and guard"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.illustrip_loop import FrameLoop
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=8)
tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
h, w, S = 256, 320, 6
for gen in ('RGB', 'FFT'):
    for graph in (False, True):
        torch.manual_seed(0); np.random.seed(0)
        p0 = torch.randn(1, 3, h, w) * 0.3 if gen == 'RGB' else 0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)
        kw = dict(sim='mix', transform=transforms.normalize(), rng='reference', lr=0.1, use_graph=graph)
        if gen == 'RGB': kw.update(param_kind='pixel', rgb_priors=True)
        eng = Engine(p0.cuda().contiguous(), h, w, model, S, [(tgt, -1.0)], **kw)
        loop = FrameLoop(eng, gen=gen, opt_step=1)
        out = []
        for frame in range(6):
            torch.manual_seed(100 + frame); np.random.seed(100 + frame)
            loop.reparameterise(1.03, (3, -1), 2.0, 1.0)
            before = eng.params.clone()
            eng.step()
            torch.cuda.synchronize()
            d = (eng.params - before).abs()
            out.append('f%d loss %.5f |dp| mean %.4f max %.4f guard %d gradmax %.3g vmax %.3g' % (frame, float(eng.loss), d.mean().item(), d.max().item(), int(eng.guard[0]), float(eng.grad.abs().max()), float(eng.v.max())))
        print(gen, 'graph' if graph else 'eager')
        for o in out: print('   ', o)
