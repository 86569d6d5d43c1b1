"""Depth warp of the illustrip frame loop -- host-side mirror of the reference's depth/depth.py (same names, argument order
and meaning: `resize`, `grid_warp`, `depthwarp`, `InferDepthAny`) and of illustrip.py:115-128 `depth_transform`, on the HIP
kernels of csrc/depthwarp.hip (C ABI: aph_triangle_blur, aph_resize_bicubic, aph_flip_w, aph_grid_warp).

What is NOT here is the depth estimator: Depth-Anything-V2 (depth.py:20-32) is a third-party DINOv2 + DPT network whose
weights are not part of either repository.  `depthwarp` takes it as the `infer_any` callable, as the reference does;
`InferDepthAny` loads it through `transformers` from a LOCAL directory (there is no network here) and fails loudly when
none is given."""
import os

import torch

from . import ops
from .image import to_valid_rgb


def _chw(x):
    return x.detach().reshape(-1, x.shape[-2], x.shape[-1]).float().contiguous()


def triangle_blur(x, kernel_size=3, pow=1.0, mix=1.0, lib=None):
    """utils.py:137-147; with mix < 1: torch.lerp(x, triangle_blur(x), mix) in the same launch (depth.py:75)"""
    src = _chw(x)
    C, H, W = src.shape
    out = torch.empty_like(src)
    ops._L(lib, src).call('aph_triangle_blur', ops.ptr(src), C, H, W, int(kernel_size), float(pow), float(mix), ops.ptr(out), ops._stream(src))
    return out.reshape(x.shape)


def resize(img, size, lib=None):
    """depth.py:41-42: F.interpolate(img, size, mode='bicubic', align_corners=True)"""
    src = _chw(img)
    C, h, w = src.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty(C, H, W, dtype=torch.float32, device=src.device)
    ops._L(lib, src).call('aph_resize_bicubic', ops.ptr(src), C, h, w, ops.ptr(out), H, W, ops._stream(src))
    return out.reshape(*img.shape[:-2], H, W)


def flip_w(x, mul=None, lib=None):
    """torch.flip(x, [-1]) (times `mul` elementwise if given)"""
    src = _chw(x)
    C, H, W = src.shape
    m = _chw(mul) if mul is not None else None
    out = torch.empty_like(src)
    ops._L(lib, src).call('aph_flip_w', ops.ptr(src), ops.ptr(m), C, H, W, ops.ptr(out), ops._stream(src))
    return out.reshape(x.shape)


def grid_warp(img, dtensor, H, W, strength, centre, midpoint, dlens=0.05, lib=None):
    """depth.py:44-66.  img [1,C,H,W], dtensor [1,H,W] (or [H,W]), centre = (x, y) in [-1,1]"""
    src = _chw(img)
    C = src.shape[0]
    if tuple(src.shape[-2:]) != (H, W) or dtensor.numel() != H * W:
        raise ValueError('grid_warp: image %s / depth %s do not match (%d, %d)' % (tuple(img.shape), tuple(dtensor.shape), H, W))
    dep = dtensor.detach().reshape(H, W).float().contiguous()
    ws = torch.empty(C * H * W + 256, dtype=torch.float32, device=src.device)
    out = torch.empty_like(src)
    cx, cy = (float(c) for c in (centre.tolist() if torch.is_tensor(centre) else centre))
    ops._L(lib, src).call('aph_grid_warp', ops.ptr(src), ops.ptr(dep), C, H, W, float(strength), cx, cy, float(midpoint), float(dlens),
                          ops.ptr(ws), ops.ptr(out), ops._stream(src))
    return out.reshape(1, C, H, W)


def estimator_size(H, W, res=518):
    """depth.py:71-73"""
    dim = [res, int(res * W / H)] if H < W else [int(res * H / W), res]
    return [x - x % 14 for x in dim]


def depth_map(img, infer_any, res=518, lib=None):
    """depth.py:69-78: [1,3,H,W] in (0,1) -> depth [1,1,H,W]; two estimator calls (the second on the mirrored image)"""
    _, _, H, W = img.shape
    dim = estimator_size(H, W, res)
    image = resize(triangle_blur(img, 5, 2, mix=0.5, lib=lib), dim, lib=lib)
    d1 = infer_any(image)
    d2 = infer_any(flip_w(image, lib=lib))
    depth = flip_w(d2.reshape(1, 1, *dim), mul=d1.reshape(1, 1, *dim), lib=lib)
    return resize(depth, (H, W), lib=lib)


def depthwarp(img_t, img, infer_any, strength=0, centre=[0, 0], midpoint=0.5, save_path=None, save_num=0, dlens=0.05, res=518, lib=None):
    """depth.py:68-84 (argument order as upstream; `res` = its hard-coded 518 exposed for small tests)"""
    _, _, H, W = img.shape
    depth = depth_map(img, infer_any, res, lib=lib)
    if save_path is not None:
        from .utils import checkout                                   # depth.py:80-82 save_img: single channel -> grey JPEG
        checkout(depth.detach().reshape(1, H, W).expand(3, H, W).cpu(), os.path.join(save_path, '%05d.jpg' % save_num))
    return grid_warp(img_t, depth.reshape(1, H, W), H, W, strength, centre, midpoint, dlens, lib=lib)


def depth_transform(img_t, _deptha, depthX=0, scale=1., shift=[0, 0], colors=1, depth_dir=None, save_num=0, res=518, lib=None):
    """illustrip.py:115-128"""
    depthX = float(depthX)
    scale = float(scale[0]) if isinstance(scale, (list, tuple)) else float(scale)
    size = img_t.shape[-2:]
    dX = 100. * shift[0] / size[1]
    dY = 100. * shift[1] / size[0]
    dZ = 0.5 + 32. * (scale - 1)
    img = to_valid_rgb(lambda x: x, colors=colors)(img_t.detach().reshape(1, -1, *size))
    return depthwarp(img_t, img.detach(), _deptha, depthX, [dX, dY], dZ, save_path=depth_dir, save_num=save_num, res=res, lib=lib)


class InferDepthAny:
    """depth.py:20-32.  `path`: a local Depth-Anything-V2 checkpoint directory in Hugging Face format (or the environment
    variable APH_DEPTH_WEIGHTS); the estimator itself runs on PyTorch -- it is not part of the HIP path."""

    def __init__(self, modtype='B', device=None, path=None):
        path = path or os.environ.get('APH_DEPTH_WEIGHTS')
        if not path or not os.path.isdir(path):
            raise RuntimeError('InferDepthAny: no local Depth-Anything-V2 checkpoint (pass path= or set APH_DEPTH_WEIGHTS); this image has '
                               'no network access and the weights are not part of the repository')
        from transformers import AutoModelForDepthEstimation
        self.device = torch.device(device if device is not None else 'cuda')
        self.model = AutoModelForDepthEstimation.from_pretrained(path).to(self.device).eval()
        self.mean = torch.tensor([0.485, 0.456, 0.406], device=self.device).view(1, 3, 1, 1)
        self.std = torch.tensor([0.229, 0.224, 0.225], device=self.device).view(1, 3, 1, 1)

    @torch.no_grad()
    def __call__(self, image):
        image = (image - self.mean) / self.std
        depth = self.model(pixel_values=image).predicted_depth.unsqueeze(0)
        return (depth - depth.min()) / (depth.max() - depth.min())
