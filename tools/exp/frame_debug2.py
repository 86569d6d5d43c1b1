"""debug: FrameLoop graph replay with (a) a large-max_batch model, (b) a host round trip of the parameters every frame"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.illustrip_loop import FrameLoop
tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
h, w, S = 256, 320, 6
for mb, resync in ((200, False), (8, True), (200, True)):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=mb)
    for gen in ('RGB', 'FFT'):
        torch.manual_seed(0); np.random.seed(0)
        p0 = torch.randn(1, 3, h, w) * 0.3 if gen == 'RGB' else 0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)
        kw = dict(sim='mix', transform=transforms.normalize(), rng='reference', lr=0.1, use_graph=True)
        if gen == 'RGB': kw.update(param_kind='pixel', rgb_priors=True)
        eng = Engine(p0.cuda().contiguous(), h, w, model, S, [(tgt, -1.0)], **kw)
        loop = FrameLoop(eng, gen=gen, opt_step=1)
        out = []
        cur = p0
        for frame in range(6):
            if resync:
                with torch.no_grad():
                    eng.params.copy_(cur.reshape(eng.params.shape).to('cuda'))
            torch.manual_seed(100 + frame); np.random.seed(100 + frame)
            loop.frame(1.03, (3, -1), 2.0, 1.0)
            l = float(eng.loss)
            gd = int(eng.guard[0])
            cur = eng.params.detach().cpu().clone()
            out.append('f%d loss %.5f guard %d gradmax %.3g vmax %.3g pmax %.3g' % (frame, l, gd, float(eng.grad.abs().max()), float(eng.v.max()), float(cur.abs().max())))
        print('max_batch', mb, 'resync', resync, gen)
        for o in out: print('   ', o)
