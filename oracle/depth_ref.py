"""TEST INFRASTRUCTURE -- CPU restatement of the reference's depth warp, fp32 torch.

Restates /root/reference/depth/depth.py:41-84 (`resize`, `grid_warp`, `depthwarp`), aphantasia/utils.py:137-147
(`triangle_blur`) and illustrip.py:115-128 (`depth_transform`).  The depth ESTIMATOR (Depth-Anything-V2, depth.py:20-32)
is a third-party network whose weights are not in this image: it is a caller-supplied callable here, exactly as it is an
argument (`infer_any`) of the reference's `depthwarp`.

Pinned: tests/golden/depthwarp_40x56.npz holds outputs of the reference's OWN grid_warp / depthwarp (imported in place
through oracle/shim.py:load_reference_depth, estimator = `toy_depth` below); tests/test_oracle.py checks this file
against it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


def triangle_blur(x, kernel_size=3, pow=1.0):
    """utils.py:137-147: separable triangle kernel ** pow, reflect padding"""
    padding = (kernel_size - 1) // 2
    b, c, h, w = x.shape
    k = torch.linspace(-1, 1, kernel_size + 2)[1:-1].abs().neg().add(1).reshape(1, 1, 1, kernel_size).pow(pow)
    k = k / k.sum()
    x = x.reshape(b * c, 1, h, w)
    x = F.pad(x, (padding, padding, padding, padding), mode='reflect')
    x = F.conv2d(x, k)
    x = F.conv2d(x, k.permute(0, 1, 3, 2))
    return x.reshape(b, c, h, w)


def resize(img, size):
    """depth.py:41-42"""
    return F.interpolate(img, size, mode='bicubic', align_corners=True).float()


def grid_warp(img, dtensor, H, W, strength, centre, midpoint, dlens=0.05):
    """depth.py:44-66.  img [1,C,H,W], dtensor [1,H,W], centre = (x, y) in [-1, 1]"""
    xx = torch.linspace(-1, 1, W)
    yy = torch.linspace(-1, 1, H)
    gy, gx = torch.meshgrid(yy, xx, indexing='ij')
    grid = torch.stack([gx, gy], dim=-1)
    d = torch.as_tensor(centre, dtype=torch.float32) - grid
    d_sum = dtensor[0]
    d_sum = d_sum - torch.max(d_sum) * midpoint
    grid_warped = grid + d * d_sum.unsqueeze(-1) * strength
    img = F.grid_sample(img, grid_warped.unsqueeze(0).float(), mode='bilinear', align_corners=True, padding_mode='reflection')
    lens = torch.sqrt((d ** 2).sum(dim=-1))
    grid_warped = grid + d * lens.unsqueeze(-1) * strength * dlens
    return F.grid_sample(img, grid_warped.unsqueeze(0).float(), mode='bilinear', align_corners=True, padding_mode='reflection')


def estimator_size(H, W, res=518):
    """depth.py:71-73: 518 on the lower dimension, both multiples of 14"""
    dim = [res, int(res * W / H)] if H < W else [int(res * H / W), res]
    return [x - x % 14 for x in dim]


def depth_map(img, infer, res=518):
    """depth.py:69-78: estimator input (blurred, resized), mirrored second estimate, product, resize back"""
    _, _, H, W = img.shape
    dim = estimator_size(H, W, res)
    image = resize(torch.lerp(img, triangle_blur(img, 5, 2), 0.5), dim)
    depth = infer(image)
    depth = depth * torch.flip(infer(torch.flip(image, [-1])), [-1])
    return resize(depth, (H, W))


def depthwarp(img_t, img, infer, strength=0, centre=(0, 0), midpoint=0.5, dlens=0.05, res=518):
    """depth.py:68-84 without the optional depth-map file output"""
    _, _, H, W = img.shape
    depth = depth_map(img, infer, res)
    return grid_warp(img_t, depth.squeeze(0), H, W, strength, centre, midpoint, dlens)


def to_valid_rgb_plain(x, colors=1.):
    """image.py:14-29 around an identity image function (illustrip.py:125-126): decorrelate + sigmoid, no std step"""
    from . import reference_path as R
    return R.to_rgb(x, R.colcorr_t(colors))


def depth_transform(img_t, infer, depthX=0., scale=1., shift=(0, 0), colors=1., res=518):
    """illustrip.py:115-128"""
    size = img_t.shape[-2:]
    dX = 100. * shift[0] / size[1]
    dY = 100. * shift[1] / size[0]
    dZ = 0.5 + 32. * (float(scale) - 1)
    img = to_valid_rgb_plain(img_t, colors)
    return depthwarp(img_t, img, infer, float(depthX), [dX, dY], dZ, res=res)


def toy_depth(image):
    """A deterministic stand-in for the estimator (any [1,3,h,w] -> [1,1,h,w] in [0,1] map will do): smooth, asymmetric
    under a horizontal flip, min-max normalised like InferDepthAny.__call__ (depth.py:27-32).  Used by the goldens, the
    tests, smoke and bench -- it says nothing about Depth-Anything."""
    _, _, h, w = image.shape
    yy = torch.linspace(0, 1, h, device=image.device).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w, device=image.device).view(1, 1, 1, w)
    lum = (image * torch.tensor([0.299, 0.587, 0.114], device=image.device).view(1, 3, 1, 1)).sum(1, keepdim=True)
    d = 0.6 * yy + 0.25 * torch.sin(5. * xx + 2. * yy) + 0.3 * lum + 0.1 * xx
    return (d - d.min()) / (d.max() - d.min())
