"""Check: hipGraph capture/replay of the step in a process that has RCCL initialised and collectives in flight (world 1)."""
import os, sys, warnings
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', weights=None, seed=1, max_batch=8)
h, w, S = 360, 640, 16
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
torch.manual_seed(0); np.random.seed(0)
params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
mode = os.environ.get('DBG', '')
eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph='eager' not in mode)
if 'noar' in mode:
    dist.all_reduce = lambda *a, **k: None
eng.world = 2; eng.use_graph = 'eager' not in mode and 'forcegraph' in mode          # exercise the multi-rank code path (own stream, all-reduce between graph and Adam) on one rank
eng.S_loc = S
losses = []
if 'defstream' in mode:
    eng.step = lambda *a, **k: eng._step(None, None, None, None, None)
for i in range(12):
    eng.step()
    if i == 5:
        if 'nobarrier' not in mode: dist.barrier()
        if 'nosync' not in mode:
            if 'ownsync' in mode: eng._own_stream.synchronize()
            else: torch.cuda.synchronize()
        if 'sleep' in mode:
            import time; time.sleep(0.5)
    losses.append(float(eng.loss) if 'noread' not in mode or i == 11 else 0.0)
print(mode, 'losses', ['%.5f' % v for v in losses], 'graph captured:', eng._graphs is not None, 'nan:', bool(torch.isnan(eng.params).any()))
dist.destroy_process_group()
