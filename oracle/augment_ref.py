"""TEST INFRASTRUCTURE -- CPU oracle, not part of the product path.

Restatement of `transforms_fast` (transforms.py:165-170):
    T.RandomPerspective(0.33, 0.2) -> T.RandomErasing(0.2)
    -> random_rotate_fast(range(-30,30) + 20*[0]) (transforms.py:73-83) -> normalize()

The arithmetic lives in torchvision (`torchvision>=0.8.2`, requirements.txt:10,
unpinned), which is absent from this image, so it is restated here from
torchvision's published tensor code path (transforms/_functional_tensor.py:
`_perspective_grid`, `_gen_affine_grid`, `_apply_grid_transform`; transforms.py:
`RandomPerspective.get_params`, `RandomErasing.get_params`;
functional.py: `_get_perspective_coeffs`, `_get_inverse_affine_matrix`).
**Parity unpinned**: no torchvision to run against here; the per-op maths is
cross-checked only against torch's own `grid_sample`.

Parameters are drawn in the reference's order on torch's global CPU generator
and numpy's global generator (transforms.py:75), so a seeded run reproduces
the same stream a seeded reference run with real torchvision would consume.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ROT_ANGLES = list(range(-30, 30)) + 20 * [0]        # transforms.py:168


def perspective_get_params(width, height, distortion_scale):
    hh, hw = height // 2, width // 2
    dw, dh = int(distortion_scale * hw), int(distortion_scale * hh)
    ri = lambda lo, hi: int(torch.randint(lo, hi, size=(1,)).item())
    topleft = [ri(0, dw + 1), ri(0, dh + 1)]
    topright = [ri(width - dw - 1, width), ri(0, dh + 1)]
    botright = [ri(width - dw - 1, width), ri(height - dh - 1, height)]
    botleft = [ri(0, dw + 1), ri(height - dh - 1, height)]
    start = [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]]
    return start, [topleft, topright, botright, botleft]


def perspective_coeffs(startpoints, endpoints):
    a = torch.zeros(8, 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]], dtype=torch.float64)
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]], dtype=torch.float64)
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    res = torch.linalg.lstsq(a, b, driver='gels').solution.to(torch.float32)
    return res.tolist()


def erase_get_params(img_h, img_w, scale=(0.02, 0.33), ratio=(0.3, 3.3)):
    area = img_h * img_w
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        erase_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        h = int(round(math.sqrt(erase_area * aspect)))
        w = int(round(math.sqrt(erase_area / aspect)))
        if not (h < img_h and w < img_w):
            continue
        i = int(torch.randint(0, img_h - h + 1, size=(1,)).item())
        j = int(torch.randint(0, img_w - w + 1, size=(1,)).item())
        return i, j, h, w
    return None   # torchvision returns the image unchanged


def draw_fast_params(size=224):
    """One cut's worth of transforms_fast draws, in upstream order.
    -> dict(persp=None|[8 coeffs], erase=None|(i,j,h,w), angle=float)"""
    out = dict(persp=None, erase=None, angle=0.0)
    if torch.rand(1) < 0.2:                                        # RandomPerspective(0.33, p=0.2)
        sp, ep = perspective_get_params(size, size, 0.33)
        out['persp'] = perspective_coeffs(sp, ep)
    if torch.rand(1) < 0.2:                                        # RandomErasing(p=0.2)
        out['erase'] = erase_get_params(size, size)
    out['angle'] = float(np.random.choice(ROT_ANGLES))             # transforms.py:75
    return out


def _apply_grid(img, grid):
    """_apply_grid_transform with bilinear + fill=0: sampled image times sampled ones-mask."""
    mask = torch.ones((img.shape[0], 1, img.shape[2], img.shape[3]), dtype=img.dtype)
    s = F.grid_sample(torch.cat((img, mask), dim=1), grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    return s[:, :-1] * s[:, -1:]


def perspective(img, coeffs):
    oh, ow = img.shape[-2:]
    theta1 = torch.tensor([[[coeffs[0], coeffs[1], coeffs[2]], [coeffs[3], coeffs[4], coeffs[5]]]], dtype=img.dtype)
    theta2 = torch.tensor([[[coeffs[6], coeffs[7], 1.0], [coeffs[6], coeffs[7], 1.0]]], dtype=img.dtype)
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(d, ow * 1.0 + d - 1.0, steps=ow))
    base[..., 1].copy_(torch.linspace(d, oh * 1.0 + d - 1.0, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rt1 = theta1.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=img.dtype)
    g1 = base.view(1, oh * ow, 3).bmm(rt1)
    g2 = base.view(1, oh * ow, 3).bmm(theta2.transpose(1, 2))
    grid = (g1 / g2 - 1.0).view(1, oh, ow, 2).expand(img.shape[0], oh, ow, 2)
    return _apply_grid(img, grid)


def rotate(img, angle):
    """T.functional.affine(img, angle, [0,0], 1, 0, fill=0, BILINEAR) + same-size center_crop."""
    oh, ow = img.shape[-2:]
    rot = math.radians(angle)
    matrix = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    theta = torch.tensor(matrix, dtype=img.dtype).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rt = theta.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=img.dtype)
    grid = base.view(1, oh * ow, 3).bmm(rt).view(1, oh, ow, 2).expand(img.shape[0], oh, ow, 2)
    return _apply_grid(img, grid)


def inverse_affine_matrix(angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix(center=[0,0], ..., inverted=True) as called by
    T.functional.affine on tensors (center is the image centre in the tensor code path).  shear: number or (sx, sy)."""
    if isinstance(shear, (int, float)):
        shear = [float(shear), 0.0]
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = float(translate[0]), float(translate[1])
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def affine(img, angle, translate, scale, shear):
    """T.functional.affine(img, angle, translate, scale, shear, fill=0, BILINEAR) on an NCHW tensor
    (illustrip.py:130-138 frame_transform; the same-size center_crop that follows is the identity)."""
    oh, ow = img.shape[-2:]
    theta = torch.tensor(inverse_affine_matrix(angle, translate, scale, shear), dtype=img.dtype).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rt = theta.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=img.dtype)
    grid = base.view(1, oh * ow, 3).bmm(rt).view(1, oh, ow, 2).expand(img.shape[0], oh, ow, 2)
    return _apply_grid(img, grid)


def erase(img, rect):
    i, j, h, w = rect
    img = img.clone()
    img[..., i:i + h, j:j + w] = 0.0
    return img


def apply_fast(cut, prm, normalize):
    """transforms_fast on one cut [1,3,s,s] with explicit parameters."""
    if prm['persp'] is not None:
        cut = perspective(cut, prm['persp'])
    if prm['erase'] is not None:
        cut = erase(cut, prm['erase'])
    cut = rotate(cut, prm['angle'])
    return normalize(cut)
