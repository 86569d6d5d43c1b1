"""Debug aid: are two identical runs bitwise identical?  Which buffer differs first?"""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', weights=None, seed=1, max_batch=8)
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))

def run(h, w, S, steps, graph, tf):
    torch.manual_seed(0); np.random.seed(0)
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=tf, macro=0.4, use_graph=graph)
    snaps = []
    for i in range(steps):
        eng.step()
        torch.cuda.synchronize()
        snaps.append(dict(enc=eng.enc.clone(), genc=eng.genc.clone(), gpatch=eng.gpatch.clone(), grgb=eng.grgb.clone(), grad=eng.grad.clone(),
                          params=eng.params.clone(), loss=eng.loss.clone(), rgb=eng.rgb.clone(), patches=eng.patches.clone()))
    return snaps

for (h, w, S, tf, name) in ((360, 640, 16, transforms.transforms_fast, 'small fast'), (720, 1280, 190, transforms.transforms_fast, 'C2 fast'),
                            (720, 1280, 200, transforms.normalize(), 'C2 none')):
    a = run(h, w, S, 4, False, tf)
    b = run(h, w, S, 4, False, tf)
    for i, (x, y) in enumerate(zip(a, b)):
        bad = [k for k in x if not torch.equal(x[k], y[k])]
        nan = [k for k in x if torch.isnan(x[k].float()).any()]
        print('%-10s step %d: differing buffers %s  nan %s  loss %.6f / %.6f' % (name, i, bad, nan, float(x['loss']), float(y['loss'])), flush=True)
