"""Race tripwire at whole-step level: two identical 300-step runs of the C2 step (hipGraph replay, `-tf fast` and `-tf none`) must end in
bitwise identical parameters and loss histories (every kernel is deterministic by construction; an under-waited vmcnt / barrier shows up here)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', weights=None, seed=1, max_batch=200)
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))


def run(h, w, S, tf):
    torch.manual_seed(0); np.random.seed(0)
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=tf, macro=0.4)
    losses = []
    for i in range(STEPS):
        losses.append(eng.step().clone())
    torch.cuda.synchronize()
    return eng.params.clone(), torch.cat(losses).cpu(), int(eng.guard[0])


for (h, w, S, tf, name) in ((720, 1280, 190, transforms.transforms_fast, 'C2 -tf fast'), (720, 1280, 200, transforms.normalize(), 'C2 -tf none')):
    p1, l1, g1 = run(h, w, S, tf)
    p2, l2, g2 = run(h, w, S, tf)
    print('%s, %d steps twice: parameters bitwise equal %s, loss histories bitwise equal %s, finite %s, skipped steps %d / %d, last loss %.6f'
          % (name, STEPS, torch.equal(p1, p2), torch.equal(l1, l2), bool(torch.isfinite(l1).all()), g1, g2, float(l1[-1])), flush=True)
