"""A/B of the per-step frame save (clip_fft.py:297-306, opt_step 1) at the headline workload: no save, save with the interpreter's default GIL
switch interval (5 ms), save with FrameWriter's 0.5 ms.  python tools/exp/save_ab.py [steps]"""
import os, shutil, sys, tempfile, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import clip_fft
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
h, w, S = 720, 1280, 190
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))


def run(save, interval):
    torch.manual_seed(0); np.random.seed(0)
    sys.setswitchinterval(0.005)
    leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    eng = Engine(leaf, h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4)
    tmp = tempfile.mkdtemp(prefix='aph_save_ab_')
    writer = clip_fft.FrameWriter(h, w, switch_interval=interval) if save else None
    try:
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
        dt = time.perf_counter() - t0
    finally:
        if writer is not None:
            writer.close()
        shutil.rmtree(tmp, ignore_errors=True)
    return steps / dt


for rep in range(2):
    print('no save %.1f   save, 5 ms switch interval %.1f   save, 0.5 ms %.1f   save, 0.1 ms %.1f  steps/s' % (run(False, None), run(True, None), run(True, 5e-4), run(True, 1e-4)), flush=True)
