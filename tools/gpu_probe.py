"""Per-kernel timing probe on one MI355X (not a test, not the bench): torch.cuda events around each C-ABI call."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from aphantasia_amd import _ffi, ops
from aphantasia_amd.weights import synthetic_visual_weights, visual_config

dev = 'cuda'


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for (M, N, K) in [(9500, 768, 768), (9500, 2304, 768), (9500, 3072, 768), (9500, 768, 3072), (9310, 768, 3072), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    ms = timeit(lambda: ops.gemm_f16(A, B))
    res['gemm_%dx%dx%d' % (M, N, K)] = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)

H, W, S = 720, 1280, 190
plan = ops.SynthPlan(3, H, W)
params = (0.01 * torch.randn(3, H, W // 2 + 1, 2)).to(dev)
scale = torch.rand(H, W // 2 + 1).to(dev) + 0.5
cc = [0.56, 0.58, 0.58, 0.35, 0, -0.35, 0.078, -0.19, 0.117]
raw, rgb = ops.synth_fft_fwd(plan, params, scale, None, 1.0, cc)
res['synth_fwd'] = dict(ms=timeit(lambda: ops.synth_fft_fwd(plan, params, scale, None, 1.0, cc)))
g = torch.randn_like(rgb)
gp = torch.empty_like(params)
res['synth_bwd'] = dict(ms=timeit(lambda: ops.synth_fft_bwd(plan, g, rgb, raw, scale, 1.0, cc, out=gp)))

geom = ops.make_geom(H, W, S, 224, 32)
tab = torch.stack([torch.randint(224, 720, (S,)), torch.zeros(S, dtype=torch.long), torch.zeros(S, dtype=torch.long)], 1)
tab[:, 1] = (torch.rand(S) * (W - tab[:, 0])).long()
tab[:, 2] = (torch.rand(S) * (H - tab[:, 0])).long()
tab = tab.int().to(dev)
patches = ops.sample_fwd(geom, rgb, tab, out_mode=_ffi.APH_OUT_PATCH_F16)
res['sample_fwd'] = dict(ms=timeit(lambda: ops.sample_fwd(geom, rgb, tab, out=patches, out_mode=_ffi.APH_OUT_PATCH_F16)))
gpatch = torch.randn(patches.shape, device=dev)
grgb = torch.empty_like(rgb)
res['sample_bwd'] = dict(ms=timeit(lambda: ops.sample_bwd(geom, gpatch, tab, out=grgb, out_mode=_ffi.APH_OUT_PATCH_F16), n=5, warm=1))

cfg = visual_config('ViT-B/32')
vit = ops.VitHandle(cfg, synthetic_visual_weights(cfg, 1), max_batch=S)
enc = torch.empty(S, 512, device=dev)
res['vit_fwd'] = dict(ms=timeit(lambda: vit.forward(patches, S, out=enc)))
genc = torch.randn(S, 512, device=dev)
res['vit_bwd'] = dict(ms=timeit(lambda: vit.backward(genc, S, out=gpatch)))
F_img = 8817623040
res['vit_fwd']['tflops'] = S * F_img / res['vit_fwd']['ms'] / 1e9
res['vit_bwd']['tflops'] = S * F_img / res['vit_bwd']['ms'] / 1e9
tg = torch.randn(1, 512, device=dev)
res['loss'] = dict(ms=timeit(lambda: ops.sim_loss(enc, tg, [-1.0], 'mix')))
v = torch.zeros_like(params)
hy = torch.tensor(ops.adam_hyper(1, 0.05), device=dev)
res['adam'] = dict(ms=timeit(lambda: ops.adam_step(params, gp, None, v, None, hy)))
print(json.dumps(res, indent=1))
