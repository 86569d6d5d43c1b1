"""Check: single-rank graph replay followed by eager dependent kernels (per-step frame synthesis), with device syncs."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', weights=None, seed=1, max_batch=8)
h, w, S = 360, 640, 16
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
def run(graph):
    torch.manual_seed(0); np.random.seed(0)
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph=graph, expand=0.3)
    frames = []
    for i in range(14):
        eng.step()
        eng.set_prev_enc()                       # eager copy of the graph's output
        frames.append(eng.synthesize(1.1).clone())   # eager kernels reading the graph's Adam result
        if i in (4, 9):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return frames, eng.params.clone()
fa, pa = run(False)
fb, pb = run(True)
print('frames equal:', [bool(torch.equal(a, b)) for a, b in zip(fa, fb)], 'params equal:', bool(torch.equal(pa, pb)))
