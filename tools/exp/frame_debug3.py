"""Repro harness for the hipGraph memset-node defect (ROCm 7.2 / gfx950): a captured step that contained hipMemsetAsync nodes,
replayed in the sequence
   eager kernels -> device synchronize -> async H2D copies -> graph replay
produced NaN gradients from the second replay on (the guarded Adam skipped those steps: `guard` counts them), on the NULL
stream and on a dedicated stream alike when the eager kernels ran on the NULL stream.  With the memsets replaced by fill
kernels (csrc/aph_device.h zero_fill_async) both variants below print all zeros; check out the commit before that change to
see the counters climb from frame 3."""
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
    orig = eng.step
    def synced(*a, **k):
        torch.cuda.synchronize()
        if own_stream:
            with torch.cuda.stream(STREAM):
                return orig(*a, **k)
        return orig(*a, **k)
    eng.step = synced
    skipped = []
    for frame in range(8):
        torch.manual_seed(100 + frame); np.random.seed(100 + frame)
        loop.frame(1.03, (3, -1), 2.0, 1.0)
        skipped.append(int(eng.guard[0]))
    return skipped


STREAM = torch.cuda.Stream()
print('step on the NULL stream       skipped-step counter per frame', run(False))
print('step on a dedicated stream    skipped-step counter per frame', run(True))
