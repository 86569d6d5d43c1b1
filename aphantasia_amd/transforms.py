"""Host side of the augmentations (drop-in for aphantasia/transforms.py: `normalize`,
`transforms_fast`).

In the reference a transform is a closure applied to every cut inside slice_imgs' Python loop
(utils.py:250-251).  Here a transform is a *specification*: slice_imgs draws its random parameters
on the host -- in the reference's exact order on torch's / numpy's global generators, so a seeded
run consumes the same random stream -- and the HIP sampler applies all cuts in one launch
(csrc/sampler.hip).  `-tf custom` / `-tf elastic` (kornia-based, non-default) are not provided.
"""
import math
import os

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
        if _RU.uniform_().item() < 0.2:              # `torch.rand(1) < p` (fp32 compare; 0.2f > 0.2 and no fp32 lies between)
            prm['persp'] = _PendingPersp(*_perspective_params(size, size, 0.33))     # solved in one batch by finish_draws()
        if _RU.uniform_().item() < 0.2:
            prm['erase'] = _erase_params(size, size)
        prm['angle'] = float(ROT_ANGLES_FAST[np.random.randint(0, len(ROT_ANGLES_FAST))])   # == np.random.choice (same stream), transforms.py:75
        return prm


def normalize():
    """transforms.py:102-109"""
    return _Normalize()


transforms_fast = _Fast()


_EXACT_ZERO_ROT = os.environ.get('APH_EXACT_ZERO_ROTATION') == '1'
_RI = torch.empty(1, dtype=torch.int64)
_RU = torch.empty(1)
_LOG_RATIO = {}


def _perspective_params(width, height, distortion_scale):
    hh, hw = height // 2, width // 2
    dw, dh = int(distortion_scale * hw), int(distortion_scale * hh)

    def ri(lo, hi):                       # == torch.randint(lo, hi, (1,)) on the global generator, without the allocation
        return int(_RI.random_(lo, hi).item())
    tl = [ri(0, dw + 1), ri(0, dh + 1)]
    tr = [ri(width - dw - 1, width), ri(0, dh + 1)]
    br = [ri(width - dw - 1, width), ri(height - dh - 1, height)]
    bl = [ri(0, dw + 1), ri(height - dh - 1, height)]
    return [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]], [tl, tr, br, bl]


class _PendingPersp:
    """start / end points of a drawn perspective whose 8 coefficients are still to be solved (the solve consumes no random numbers)"""
    __slots__ = ('sp', 'ep')

    def __init__(self, sp, ep):
        self.sp, self.ep = sp, ep


def finish_draws(prms):
    """Solve the perspective systems of a list of per-cut parameter dicts in ONE batched fp64 solve (a seeded `--rng reference` run
    draws ~40 of them per step; one at a time they cost more host time than the GPU step takes)."""
    pend = [p for p in prms if isinstance(p.get('persp'), _PendingPersp)]
    if not pend:
        return prms
    e = np.asarray([p['persp'].ep for p in pend], dtype=np.float64)          # [n,4,2]
    st = np.asarray([p['persp'].sp for p in pend], dtype=np.float64)
    a = np.zeros((len(pend), 8, 8), dtype=np.float64)
    a[:, 0::2, 0:2] = e
    a[:, 0::2, 2] = 1.0
    a[:, 0::2, 6:8] = -st[:, :, 0:1] * e
    a[:, 1::2, 3:5] = e
    a[:, 1::2, 5] = 1.0
    a[:, 1::2, 6:8] = -st[:, :, 1:2] * e
    sol = np.linalg.solve(a, st.reshape(len(pend), 8, 1))[..., 0].astype(np.float32)
    for p, c in zip(pend, sol):
        p['persp'] = c.tolist()
    return prms


def _perspective_coeffs(startpoints, endpoints):
    """8 coefficients mapping output (endpoint) to input (startpoint) coordinates, least squares in fp64."""
    e = np.asarray(endpoints, dtype=np.float64)             # p1 rows
    st = np.asarray(startpoints, dtype=np.float64)          # p2 rows
    a = np.zeros((8, 8), dtype=np.float64)
    a[0::2, 0:2] = e
    a[0::2, 2] = 1.0
    a[0::2, 6:8] = -st[:, 0:1] * e
    a[1::2, 3:5] = e
    a[1::2, 5] = 1.0
    a[1::2, 6:8] = -st[:, 1:2] * e
    b = st.reshape(8)
    # torchvision solves this square full-rank system with lstsq(gels) in fp64 and casts to fp32; a direct fp64 solve
    # agrees to ~1e-14 before the cast
    return np.linalg.solve(a, b).astype(np.float32).tolist()


def _erase_params(img_h, img_w, scale=(0.02, 0.33), ratio=(0.3, 3.3)):
    area = img_h * img_w
    lr = _LOG_RATIO.get(ratio)
    if lr is None:
        t = torch.log(torch.tensor(ratio))                   # fp32 log as torchvision computes it; the bounds reach uniform_ as doubles
        lr = _LOG_RATIO[ratio] = (t[0].item(), t[1].item())
    for _ in range(10):
        erase_area = area * _RU.uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(_RU.uniform_(lr[0], lr[1])).item()
        h = int(round(math.sqrt(erase_area * aspect)))
        w = int(round(math.sqrt(erase_area / aspect)))
        if not (h < img_h and w < img_w):
            continue
        i = int(_RI.random_(0, img_h - h + 1).item())
        j = int(_RI.random_(0, img_w - w + 1).item())
        return i, j, h, w
    return None


def pack_aug(prms):
    """list of per-cut dicts (persp / erase / angle) -> f32 [S,16] table for aph_sample_fwd (host tensor)."""
    finish_draws(prms)
    t = np.zeros((len(prms), _ffi.APH_AUG_STRIDE), dtype=np.float32)
    for s, p in enumerate(prms):
        if p.get('persp') is not None:
            t[s, 0:8] = p['persp']
            t[s, 8] = 1.0
        if p.get('erase') is not None:
            t[s, 9:13] = p['erase']
        ang = p.get('angle')
        if ang is not None:
            rot = math.radians(float(ang))
            # 0 degrees (21 of the 80 choices of random_rotate_fast, transforms.py:168): upstream still resamples the cut through an
            # identity grid (torchvision's affine has no shortcut), which moves values by <= 2e-5 through the grid's fp32 rounding.
            # The sampler copies instead (has_rotation = 0): a documented deviation far inside the parity tolerance that saves a
            # quarter of the rotation kernels' work.  APH_EXACT_ZERO_ROTATION=1 restores the resampling.
            has_rot = 1.0 if (float(ang) != 0.0 or _EXACT_ZERO_ROT) else 0.0
            t[s, 13], t[s, 14], t[s, 15] = math.cos(rot), math.sin(rot), has_rot
    return torch.from_numpy(t)


# ----------------------------------------------------------------------------- bulk (vectorised) draws
def draw_fast_bulk(S, size, rng):
    """transforms_fast parameters for S cuts at once from a numpy Generator -> packed f32 [S,16] table.
    Same distributions as the per-cut draws above (torchvision get_params), NOT the reference's random stream."""
    t = np.zeros((S, _ffi.APH_AUG_STRIDE), dtype=np.float32)
    hw = size // 2
    dw = int(0.33 * hw)
    # RandomPerspective(0.33, p=0.2)
    idx = np.nonzero(rng.random(S) < 0.2)[0]
    if idx.size:
        n = idx.size
        lo = rng.integers(0, dw + 1, size=(n, 4))                     # near-edge offsets
        hi = rng.integers(size - dw - 1, size, size=(n, 4))           # far-edge coordinates
        end = np.stack([np.stack([lo[:, 0], lo[:, 1]], 1), np.stack([hi[:, 0], lo[:, 2]], 1),
                        np.stack([hi[:, 1], hi[:, 2]], 1), np.stack([lo[:, 3], hi[:, 3]], 1)], 1).astype(np.float64)   # tl,tr,br,bl
        start = np.array([[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]], dtype=np.float64)
        a = np.zeros((n, 8, 8))
        for i in range(4):
            p1, p2 = end[:, i], start[i]
            a[:, 2 * i, 0], a[:, 2 * i, 1], a[:, 2 * i, 2] = p1[:, 0], p1[:, 1], 1
            a[:, 2 * i, 6], a[:, 2 * i, 7] = -p2[0] * p1[:, 0], -p2[0] * p1[:, 1]
            a[:, 2 * i + 1, 3], a[:, 2 * i + 1, 4], a[:, 2 * i + 1, 5] = p1[:, 0], p1[:, 1], 1
            a[:, 2 * i + 1, 6], a[:, 2 * i + 1, 7] = -p2[1] * p1[:, 0], -p2[1] * p1[:, 1]
        b = np.broadcast_to(start.reshape(8), (n, 8))
        t[idx, 0:8] = np.linalg.solve(a, b[..., None])[..., 0].astype(np.float32)
        t[idx, 8] = 1.0
    # RandomErasing(p=0.2): up to 10 tries of (area in [.02,.33], log-uniform aspect in [.3,3.3])
    idx = np.nonzero(rng.random(S) < 0.2)[0]
    if idx.size:
        n = idx.size
        area = size * size * rng.uniform(0.02, 0.33, size=(n, 10))
        asp = np.exp(rng.uniform(math.log(0.3), math.log(3.3), size=(n, 10)))
        eh = np.rint(np.sqrt(area * asp)).astype(np.int64)
        ew = np.rint(np.sqrt(area / asp)).astype(np.int64)
        ok = (eh < size) & (ew < size)
        first = np.argmax(ok, axis=1)
        any_ok = ok.any(axis=1)
        eh, ew = eh[np.arange(n), first], ew[np.arange(n), first]
        ei = np.floor(rng.random(n) * (size - eh + 1)).astype(np.int64)
        ej = np.floor(rng.random(n) * (size - ew + 1)).astype(np.int64)
        sel = idx[any_ok]
        t[sel, 9], t[sel, 10], t[sel, 11], t[sel, 12] = ei[any_ok], ej[any_ok], eh[any_ok], ew[any_ok]
    # random_rotate_fast
    ang = np.asarray(ROT_ANGLES_FAST, dtype=np.float64)[rng.integers(0, len(ROT_ANGLES_FAST), size=S)]
    rot = np.radians(ang)
    t[:, 13], t[:, 14] = np.cos(rot), np.sin(rot)
    t[:, 15] = 1.0 if _EXACT_ZERO_ROT else (ang != 0.0)          # 0 degrees: copy instead of an identity resampling (see pack_aug)
    return t


# ----------------------------------------------------------------------------- illustrip's per-frame warp
def inverse_affine_matrix(angle, translate, scale, shear):
    """torchvision's _get_inverse_affine_matrix(center=[0, 0], angle, translate, scale, shear) -- the matrix
    T.functional.affine builds for tensors (origin at the image centre).  shear: degrees or (sx, sy)."""
    if isinstance(shear, (int, float)):
        shear = [float(shear), 0.0]
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = float(translate[0]), float(translate[1])
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [v / scale for v in (d, -b, 0.0, -c, a, 0.0)]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def frame_transform(img, size, angle, shift, scale, shear, lib=None):
    """illustrip.py:130-138: T.functional.affine(img, angle, tuple(shift), scale, shear, fill=0, BILINEAR) followed by
    center_crop(size) (which also zero-pads on torch >= 1.8).  img: [1,C,H,W] f32 on the GPU; one HIP launch."""
    from . import ops
    x = img.detach().reshape(-1, img.shape[-2], img.shape[-1]).float().contiguous()
    C, H, W = x.shape
    out = torch.empty_like(x)
    L = ops._L(lib, x)
    L.call('aph_frame_affine', ops.ptr(x), C, H, W, _ffi.floats(inverse_affine_matrix(angle, shift, scale, shear)), ops.ptr(out), ops._stream(x))
    out = out.reshape(1, C, H, W)
    th, tw = int(size[0]), int(size[1])
    if (th, tw) != (H, W):                      # torchvision center_crop: pad symmetric zeros if smaller, then crop
        ph, pw = max(th - H, 0), max(tw - W, 0)
        if ph or pw:
            out = torch.nn.functional.pad(out, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        H2, W2 = out.shape[-2:]
        top, left = int(round((H2 - th) / 2.0)), int(round((W2 - tw) / 2.0))
        out = out[..., top:top + th, left:left + tw].contiguous()
    return out
