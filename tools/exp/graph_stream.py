"""Check: single-rank hipGraph replay on a NON-default stream with a device-wide synchronize in the middle."""
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
for name, own, graph, sync in (('default+graph+devsync', False, True, 'dev'), ('own+graph+devsync', True, True, 'dev'), ('own+graph+streamsync', True, True, 'stream'),
                               ('own+eager+devsync', True, False, 'dev')):
    torch.manual_seed(0); np.random.seed(0)
    st = torch.cuda.Stream() if own else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
        eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph=graph)
        losses = []
        for i in range(12):
            eng.step()
            if i == 5:
                torch.cuda.synchronize() if sync == 'dev' else st.synchronize()
            losses.append(float(eng.loss))
    print('%-24s %s' % (name, ['%.5f' % v for v in losses[5:]]), flush=True)
