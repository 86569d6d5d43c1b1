"""Debug aid: is the gloo all-reduce of a CUDA tensor (2 ranks on one GPU) correct / ordered after kernels launched through the C ABI?"""
import os, sys, warnings
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', weights=None, seed=1, max_batch=8)
h, w, S = 720, 1280, 190
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
torch.manual_seed(0); np.random.seed(0)
params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, rank=rank, world=world, use_graph=False)
for i in range(6):
    table, augs = eng.draw()
    eng._state['step'][0] += 1
    eng._calls += 1
    from aphantasia_amd import ops
    hy = ops.adam_hyper(eng._state['step'][0], eng.lr, eng.beta1, 0.999, 1e-8, eng.wd, 1.0)
    eng.hyper.copy_(torch.tensor(hy, dtype=torch.float32))
    eng.table.copy_(torch.from_numpy(np.ascontiguousarray(table[eng.lo:eng.hi])))
    eng.aug.copy_(torch.from_numpy(np.ascontiguousarray(augs[eng.lo:eng.hi])))
    eng._enqueue_grad(None)
    if os.environ.get('CHK_SYNC'):
        torch.cuda.synchronize()
    dist.all_reduce(eng.grad)                       # as Engine.step does (no host sync before it)
    got = eng.grad.clone()
    torch.cuda.synchronize()
    # reference: recompute the local gradient, gather through the host
    eng._enqueue_grad(None)
    torch.cuda.synchronize()
    loc = eng.grad.cpu()
    parts = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(parts, loc)
    want = sum(parts)
    err = (got.cpu() - want).abs().max().item()
    print('rank %d step %d: |allreduce - host sum| max %.3e (|want| max %.3e) nan %d' % (rank, i, err, want.abs().max().item(), int(torch.isnan(got).sum())), flush=True)
    eng.grad.copy_(want.cuda())
    eng._enqueue_adam()
dist.destroy_process_group()
