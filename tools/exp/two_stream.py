"""Experiment (round 3): does running two half-batch steps CONCURRENTLY on two HIP streams beat one full-batch step?
Every GEMM launch is two phases that do not overlap on a CU (MFMA main loops, then an HBM-bound store burst, all CUs in
lockstep -- DESIGN.md section 4).  Two independent kernel streams are out of phase with each other, fill each other's tails
and overlap the memory-bound LN / attention kernels of one half with the GEMMs of the other.

    python tools/exp/two_stream.py [S] [steps]

Prints steps/s-equivalent for: one engine with S cuts; two engines with S/2 cuts each on two streams (eager and graph)."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 190
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
h, w = 720, 1280
dev = torch.device('cuda')


def mk(S_, graph, tf='fast'):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S_)
    torch.manual_seed(0); np.random.seed(0)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(dev).contiguous()
    tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    trf = transforms.transforms_fast if tf == 'fast' else transforms.normalize()
    return Engine(leaf, h, w, model, S_, [(tgt, -1.0)], sim='mix', transform=trf, use_graph=graph)


def bench(engs, streams, n):
    def one():
        if streams is None:
            for e in engs: e.step()
        else:
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.step()
    for _ in range(6): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for graph in (True, False):
    for tf in ('fast', 'none'):
        e_full = mk(S, graph, tf)
        t_full = bench([e_full], None, steps)
        del e_full
        ea, eb = mk(S // 2, graph, tf), mk(S - S // 2, graph, tf)
        t_seq = bench([ea, eb], None, steps)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        t_par = bench([ea, eb], [sa, sb], steps)
        del ea, eb
        print('graph=%d tf=%s: one engine S=%d %.3f ms | two engines S/2 sequential %.3f ms | two engines on two streams %.3f ms  (x%.3f vs full)'
              % (graph, tf, S, t_full, t_seq, t_par, t_full / t_par), flush=True)
