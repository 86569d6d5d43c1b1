"""Repro of the runtime defect behind Engine.step()'s dedicated stream: on the legacy default (NULL) stream the sequence
   eager kernels -> device synchronize -> async H2D copies -> hipGraph replay
gives NaN gradients from the second replay on (the guarded Adam skips those steps: `guard` counts them); the same sequence on
a non-NULL stream is clean.  Engine._step is called directly here to stay on the caller's stream (Engine.step() itself always
switches to its own stream)."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.illustrip_loop import FrameLoop
tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
h, w, S = 256, 320, 6
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=8)


def run(own_stream):
    torch.manual_seed(0); np.random.seed(0)
    p0 = torch.randn(1, 3, h, w) * 0.3
    eng = Engine(p0.cuda().contiguous(), h, w, model, S, [(tgt, -1.0)], sim='mix', transform=transforms.normalize(), rng='reference', lr=0.1,
                 use_graph=True, param_kind='pixel', rgb_priors=True)
    loop = FrameLoop(eng, gen='RGB', opt_step=1)
    if not own_stream:
        def on_callers_stream(table=None, augs=None, lr=None, shift=None, tables2=None):
            torch.cuda.synchronize()
            return eng._step(table, augs, lr, shift, tables2)
        eng.step = on_callers_stream
    else:
        orig = eng.step
        def synced(*a, **k):
            torch.cuda.synchronize()
            return orig(*a, **k)
        eng.step = synced
    skipped = []
    for frame in range(8):
        torch.manual_seed(100 + frame); np.random.seed(100 + frame)
        loop.frame(1.03, (3, -1), 2.0, 1.0)
        skipped.append(int(eng.guard[0]))
    return skipped


print('step on the NULL stream (Engine._step)      skipped-step counter per frame', run(False))
print('step on the engine stream (Engine.step)     skipped-step counter per frame', run(True))
