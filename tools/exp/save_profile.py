"""The per-step frame save under a kernel + memory-copy trace: 40 steps with the frame written every step (clip_fft.py:297-306), to see what
the 3-5 % it costs is made of on the device side.  rocprofv3 --kernel-trace --memory-copy-trace --stats -- python tools/exp/save_profile.py [save 0|1]"""
import os, shutil, sys, tempfile, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import clip_fft
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
save = (sys.argv[1] if len(sys.argv) > 1 else '1') == '1'
steps, h, w, S = 60, 720, 1280, 190
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
torch.manual_seed(0); np.random.seed(0)
leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
eng = Engine(leaf, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4)
tmp = tempfile.mkdtemp(prefix='aph_save_prof_')
writer = clip_fft.FrameWriter(h, w) if save else None
for i in range(10):
    eng.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    eng.step()
    if writer is not None:
        writer.put(eng.synthesize(1.1).reshape(3, h, w), os.path.join(tmp, '%04d.jpg' % i), 1.0)
if writer is not None:
    writer.drain()
torch.cuda.synchronize()
print('save %s: %.1f steps/s' % (save, steps / (time.perf_counter() - t0)))
if writer is not None:
    writer.close()
shutil.rmtree(tmp, ignore_errors=True)
