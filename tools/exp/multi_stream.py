"""Experiment (round 5): N engines of S/N cuts each on N HIP streams against one engine of S cuts -- at a 24-cut shard a step is ~225
dependent launches of 5-15 us each whose fixed costs (boundary, cold first loads) dominate; independent chains on separate hardware queues
overlap them.     python tools/exp/multi_stream.py [S] [steps]"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
h, w = 720, 1280
dev = torch.device('cuda')


def mk(S_, graph=True):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S_)
    torch.manual_seed(0); np.random.seed(0)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(dev).contiguous()
    tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    return Engine(leaf, h, w, model, S_, [(tgt, -1.0)], sim='mix', transform=transforms.transforms_fast, use_graph=graph)


def bench(engs, streams, n):
    def one():
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step()
    for _ in range(6): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = None
for N in (1, 2, 3, 4, 6, 8):
    sizes = [S // N + (1 if i < S % N else 0) for i in range(N)]
    engs = [mk(s) for s in sizes]
    streams = [torch.cuda.Stream() for _ in range(N)]
    t = bench(engs, streams, steps)
    base = base or t
    print('S=%d as %d engine(s) of %s cuts on %d stream(s): %.3f ms per full step  (x%.2f vs one engine)' % (S, N, sizes, N, t, base / t), flush=True)
    del engs
