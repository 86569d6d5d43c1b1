"""Times the depth-warp kernels (csrc/depthwarp.hip) at 1280x720: everything of depth.py:68-84 except the estimator."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphantasia_amd import depthwarp as DW

H, W = 720, 1280
dev = 'cuda'
img_t = torch.randn(1, 3, H, W, device=dev)
img = torch.rand(1, 3, H, W, device=dev)
dim = DW.estimator_size(H, W)
fake = torch.rand(1, 1, *dim, device=dev)
est = lambda image: fake                      # zero-cost stand-in: only the kernels around the estimator are timed


def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dep = DW.depth_map(img, est)
print('estimator input %dx%d' % tuple(dim))
print('triangle_blur+lerp      %7.1f us' % timed(lambda: DW.triangle_blur(img, 5, 2, mix=0.5)))
print('resize down             %7.1f us' % timed(lambda: DW.resize(img, dim)))
print('resize up (depth)       %7.1f us' % timed(lambda: DW.resize(fake, (H, W))))
print('grid_warp (max + 2 passes) %7.1f us' % timed(lambda: DW.grid_warp(img_t, dep.reshape(1, H, W), H, W, 0.3, [0.1, 0.0], 0.5)))
print('depth_transform without the estimator %7.1f us' % timed(lambda: DW.depth_transform(img_t, est, 0.3, 1.012, [0, 10.0], 1.5)))
