"""Host side of the augmentations (drop-in for aphantasia/transforms.py: `normalize`,
`transforms_fast`).

In the reference a transform is a closure applied to every cut inside slice_imgs' Python loop
(utils.py:250-251).  Here a transform is a *specification*: slice_imgs draws its random parameters
on the host -- in the reference's exact order on torch's / numpy's global generators, so a seeded
run consumes the same random stream -- and the HIP sampler applies all cuts in one launch
(csrc/sampler.hip).  `-tf custom` / `-tf elastic` (kornia-based, non-default) are not provided.
"""
import math

import numpy as np
import torch

from . import _ffi

ROT_ANGLES_FAST = list(range(-30, 30)) + 20 * [0]          # transforms.py:168


class Transform:
    """Marker object understood by aphantasia_amd.utils.slice_imgs."""
    geometric = False       # draws per-cut warp / erase parameters
    normalise = True        # CLIP mean/std (transforms.py:106)

    def draw(self, size):
        return None


class _Normalize(Transform):
    pass


class _Fast(Transform):
    """RandomPerspective(0.33, p=0.2) -> RandomErasing(p=0.2) -> random_rotate_fast -> normalize
    (transforms.py:165-170), parameters per torchvision's get_params."""
    geometric = True

    def draw(self, size):
        prm = dict(persp=None, erase=None, angle=0.0)
        if torch.rand(1) < 0.2:
            sp, ep = _perspective_params(size, size, 0.33)
            prm['persp'] = _perspective_coeffs(sp, ep)
        if torch.rand(1) < 0.2:
            prm['erase'] = _erase_params(size, size)
        prm['angle'] = float(np.random.choice(ROT_ANGLES_FAST))   # transforms.py:75
        return prm


def normalize():
    """transforms.py:102-109"""
    return _Normalize()


transforms_fast = _Fast()


def _perspective_params(width, height, distortion_scale):
    hh, hw = height // 2, width // 2
    dw, dh = int(distortion_scale * hw), int(distortion_scale * hh)

    def ri(lo, hi):
        return int(torch.randint(lo, hi, size=(1,)).item())
    tl = [ri(0, dw + 1), ri(0, dh + 1)]
    tr = [ri(width - dw - 1, width), ri(0, dh + 1)]
    br = [ri(width - dw - 1, width), ri(height - dh - 1, height)]
    bl = [ri(0, dw + 1), ri(height - dh - 1, height)]
    return [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]], [tl, tr, br, bl]


def _perspective_coeffs(startpoints, endpoints):
    """8 coefficients mapping output (endpoint) to input (startpoint) coordinates, least squares in fp64."""
    a = np.zeros((8, 8), dtype=np.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a[2 * i + 1] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    b = np.asarray(startpoints, dtype=np.float64).reshape(8)
    res = torch.linalg.lstsq(torch.from_numpy(a), torch.from_numpy(b), driver='gels').solution.to(torch.float32)
    return res.tolist()


def _erase_params(img_h, img_w, scale=(0.02, 0.33), ratio=(0.3, 3.3)):
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
    return None


def pack_aug(prms):
    """list of per-cut dicts (persp / erase / angle) -> f32 [S,16] table for aph_sample_fwd (host tensor)."""
    t = torch.zeros(len(prms), _ffi.APH_AUG_STRIDE, dtype=torch.float32)
    for s, p in enumerate(prms):
        if p.get('persp') is not None:
            t[s, 0:8] = torch.tensor(p['persp'], dtype=torch.float32)
            t[s, 8] = 1.0
        if p.get('erase') is not None:
            t[s, 9:13] = torch.tensor([float(v) for v in p['erase']])
        ang = p.get('angle')
        if ang is not None:
            rot = math.radians(float(ang))
            t[s, 13], t[s, 14], t[s, 15] = math.cos(rot), math.sin(rot), 1.0
    return t
