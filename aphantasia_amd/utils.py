"""Drop-in for the hot-path parts of aphantasia/utils.py: slice_imgs (utils.py:218-254), sim_func
(utils.py:276-295), pad_up_to (utils.py:178-190), plus the small file helpers clip_fft.py imports.

slice_imgs keeps the reference's semantics -- including its random draws, made on the HOST from
torch's global CPU generator in the reference's order so that a seeded run reproduces the same crop
table -- but all `count` cuts are produced by ONE batched HIP launch instead of a Python loop of
slice + F.interpolate + transform calls.
"""
import os

import numpy as np
import torch

from . import _ffi, ops
from .transforms import Transform, pack_aug


# ----------------------------------------------------------------------------- crop table
def draw_crop_params(count, size, h, w, align='uniform', macro=0., transform=None):
    """The random draws of slice_imgs (utils.py:222-228, 243-247) for one image of size h x w.
    Returns (int32 ndarray [count,3] rows (csize, offx, offy), list of per-cut augment dicts or None).
    Arithmetic is fp32 and truncating exactly like the reference's 0-dim tensor expressions
    (`0.9*sz_max[i]` is python float x int64 0-dim tensor -> fp32; `torch.rand(1) < macro` compares in fp32)."""
    return draw_crop_params_multi(count, size, [(h, w)], align, macro, transform)[0]


def draw_crop_params_bulk(count, size, h, w, align, macro, transform, rng):
    """Vectorised draws with the same distributions and fp32/truncation arithmetic as draw_crop_params but from a
    numpy Generator (NOT the reference's random stream): for throughput, ~0.1 ms instead of ms per step.
    -> (int32 [count,3], packed f32 [count,16] augment table or None)"""
    f32 = np.float32
    r_size = rng.random(count, dtype=f32)
    if align == 'central':
        r_x = np.clip(rng.standard_normal(count, dtype=f32) * f32(0.2) + f32(0.5), 0., 1.).astype(f32)
        r_y = np.clip(rng.standard_normal(count, dtype=f32) * f32(0.2) + f32(0.5), 0., 1.).astype(f32)
    else:
        r_x, r_y = rng.random(count, dtype=f32), rng.random(count, dtype=f32)
    sz_max = min(h, w)
    ph, pw = h, w
    if 'over' in align:
        ph, pw = (2 * h, 2 * w) if align == 'overmax' else (int(1.5 * h), int(1.5 * w))
    big_min = f32(0.9) * f32(sz_max)
    is_macro = rng.random(count, dtype=f32) < f32(macro)
    cs_macro = (r_size * (f32(sz_max) - big_min) + big_min).astype(np.int32)
    cs_plain = (r_size * f32(sz_max - size) + f32(size)).astype(np.int32)
    csize = np.where(is_macro, cs_macro, cs_plain).astype(np.int32)
    table = np.empty((count, 3), dtype=np.int32)
    table[:, 0] = csize
    table[:, 1] = (r_x * (pw - csize).astype(f32)).astype(np.int32)
    table[:, 2] = (r_y * (ph - csize).astype(f32)).astype(np.int32)
    aug = None
    if isinstance(transform, Transform) and transform.geometric:
        from .transforms import draw_fast_bulk
        aug = draw_fast_bulk(count, size, rng)
    return table, aug


class _Slice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, geom, table, aug, out_mode):
        rgb = img.reshape(3, geom.H, geom.W).contiguous().float()
        tmp = ops.sample_ws(geom, aug is not None, rgb.device)
        out = ops.sample_fwd(geom, rgb, table, aug, tmp, None, out_mode)
        ctx.geom, ctx.out_mode, ctx.tmp = geom, out_mode, tmp
        ctx.save_for_backward(table, aug if aug is not None else table)
        ctx.has_aug = aug is not None
        ctx.shape = img.shape
        return out

    @staticmethod
    def backward(ctx, g):
        table, aug = ctx.saved_tensors
        d = ops.sample_bwd(ctx.geom, g.contiguous().float(), table, aug if ctx.has_aug else None, ctx.tmp, None, ctx.out_mode)
        return d.reshape(ctx.shape), None, None, None, None


_RU1 = torch.empty(1)


def draw_crop_params_multi(count, size, hw_list, align='uniform', macro=0., transform=None):
    """slice_imgs' draws for a LIST of images (utils.py:222-228 once, then utils.py:238-253 per image): the size / offset
    vectors are drawn once and shared by every image; the per-cut macro draw and the transform's draws repeat per image, in
    image order.  -> list of (table, augs), one per image."""
    rnd_size = torch.rand(count).numpy()
    if align == 'central':
        rnd_offx = torch.clip(torch.randn(count) * 0.2 + 0.5, 0., 1.).numpy()
        rnd_offy = torch.clip(torch.randn(count) * 0.2 + 0.5, 0., 1.).numpy()
    else:
        rnd_offx = torch.rand(count).numpy()
        rnd_offy = torch.rand(count).numpy()
    f32 = np.float32
    geometric = isinstance(transform, Transform) and transform.geometric
    out = []
    for (h, w) in hw_list:
        sz_max = min(h, w)
        ph, pw = h, w
        if 'over' in align:
            ph, pw = (2 * h, 2 * w) if align == 'overmax' else (int(1.5 * h), int(1.5 * w))
        big_min = f32(0.9) * f32(sz_max)
        table = np.empty((count, 3), dtype=np.int32)
        augs = [] if geometric else None
        for c in range(count):
            if f32(_RU1.uniform_().item()) < f32(macro):          # == torch.rand(1) on the global generator, without the allocation
                csize = int(rnd_size[c] * (f32(sz_max) - big_min) + big_min)
            else:
                csize = int(rnd_size[c] * f32(sz_max - size) + f32(size))
            table[c, 0] = csize
            table[c, 1] = int(rnd_offx[c] * f32(pw - csize))
            table[c, 2] = int(rnd_offy[c] * f32(ph - csize))
            if geometric:
                augs.append(transform.draw(size))
        if geometric:
            from .transforms import finish_draws
            finish_draws(augs)
        out.append((table, augs))
    return out


def slice_imgs(imgs, count, size=224, transform=None, align='uniform', macro=0., patch=32):
    """utils.py:218-254.  imgs: list of [1,3,H,W] CUDA tensors -> list of [count,3,size,size] tensors.
    transform: None, aphantasia_amd.transforms.normalize(), transforms_fast (fused in the HIP sampler),
    or any other callable (applied per cut on the un-normalised crops, like upstream)."""
    for img in imgs:
        if img.dim() != 4 or img.shape[0] != 1 or img.shape[1] != 3:
            raise ValueError('slice_imgs expects [1,3,H,W] images, got %s' % (tuple(img.shape),))
    fused = transform is None or isinstance(transform, Transform)
    # the reference draws the size / offset vectors once for all images and the per-cut draws image by image (utils.py:222-253)
    draws = draw_crop_params_multi(count, size, [tuple(img.shape[2:]) for img in imgs], align, macro, transform if fused else None)
    sliced = []
    for img, (table, augs) in zip(imgs, draws):
        h, w = img.shape[2:]
        geom = ops.make_geom(h, w, count, size, patch if size % patch == 0 else 1, align)
        dev = img.device
        tb = torch.from_numpy(table).to(dev)
        aug = pack_aug(augs).to(dev) if augs is not None else None
        if fused:
            mode = _ffi.APH_OUT_NCHW_RAW if transform is None or not transform.normalise else _ffi.APH_OUT_NCHW_NORM
            sliced.append(_Slice.apply(img, geom, tb, aug, mode))
        else:
            raw = _Slice.apply(img, geom, tb, None, _ffi.APH_OUT_NCHW_RAW)
            sliced.append(torch.cat([transform(raw[c:c + 1]) for c in range(count)], 0))
    return sliced


def _wrapped_index(n, before, after, mirror, device):
    """source index of every padded position -before .. n + after - 1: periodic (i mod n), or mirrored with the edge sample repeated"""
    i = torch.arange(-before, n + after, device=device)
    if not mirror:
        return i.remainder(n)
    m = i.remainder(2 * n)
    return torch.where(m < n, m, 2 * n - 1 - m)


def tile_pad(xt, padding, symm=False):
    """Periodic (or, with symm, mirrored) extension of the last two axes by padding = (left, right, top, bottom) -- the result upstream's
    helper of the same name produces (utils.py:152-176), built here as two 1-D index vectors and two index_select gathers.  The fused
    sampler never materialises this image: it wraps its source coordinates instead (csrc/sampler.hip)."""
    left, right, top, bottom = padding
    h, w = xt.shape[-2:]
    rows = _wrapped_index(h, top, bottom, symm is True, xt.device)
    cols = _wrapped_index(w, left, right, symm is True, xt.device)
    return xt.index_select(-2, rows).index_select(-1, cols)


def pad_up_to(x, size, type='centr'):
    """x [N,C,h,w] extended to size = (H, W): the extra rows / columns split evenly around the image ('centr') or all appended after it
    ('side'), periodic unless the type says 'symm' (utils.py:178-190)"""
    (H, W), (h, w) = size, x.shape[2:]
    if (h, w) == (H, W):
        return x
    kind = type.lower()

    def split(extra):
        return (0, extra) if 'side' in kind else (extra // 2, extra - extra // 2)
    return tile_pad(x, split(W - w) + split(H - h), symm='symm' in kind)


# ----------------------------------------------------------------------------- loss
class _SimLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, target, sim_type):
        if target.shape[0] == 1:
            loss, genc = ops.sim_loss(enc.contiguous().float(), target.contiguous().float(), [1.0], sim_type)
        else:       # pairwise form: row s of `target` against cut s
            loss, genc = ops.sim_loss(enc.contiguous().float(), None, [1.0], sim_type, per_sample=target.float()[None].contiguous())
        ctx.save_for_backward(genc)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        genc, = ctx.saved_tensors
        return genc * g, None, None


def sim_func(v1, v2, type=None):
    """utils.py:276-295.  The fused HIP kernel covers the call pattern of the optimisation loop
    (v1 = one target embedding [1,D] without grad, v2 = the cuts' embeddings [S,D]); other argument
    patterns (e.g. --enforce's pairwise form) are composed from torch ops on the GPU."""
    if type is not None and 'spher' in type and 'mix' not in type:
        a = torch.nn.functional.normalize(v1, dim=-1)
        b = torch.nn.functional.normalize(v2, dim=-1)
        return (a - b).norm(dim=-1).div(2).arcsin().pow(2).mul(2)
    if v1.dim() == 2 and v2.dim() == 2 and v1.shape[0] in (1, v2.shape[0]) and not v1.requires_grad and v2.is_cuda \
            and not (v1.shape[0] != 1 and type and 'dot' in type):
        return _SimLoss.apply(v2, v1.to(v2.device), type)
    if v2.dim() == 2 and v2.shape[0] == 1 and v1.dim() == 2 and not v2.requires_grad and v1.is_cuda and not (type and 'dot' in type):
        return _SimLoss.apply(v1, v2.to(v1.device), type)
    F = torch.nn.functional
    if type is not None and 'mix' in type:
        coss = torch.cosine_similarity(v1, v2, dim=-1).mean()
        a, b = F.normalize(v1, dim=-1), F.normalize(v2, dim=-1)
        return coss - 0.25 * torch.abs((a - b).norm(dim=-1).div(2).arcsin().pow(2).mul(2)).mean()
    if type is not None and 'ang' in type:
        return 1 - torch.acos(torch.cosine_similarity(v1, v2, dim=-1)).mean() / np.pi
    if type is not None and 'dot' in type:
        dot = (v1 * v2).sum()
        return dot * (dot / (1e-6 + torch.sqrt(torch.sum(v2 ** 2))))
    return torch.cosine_similarity(v1, v2, dim=-1).mean()


# ----------------------------------------------------------------------------- small host helpers
def old_torch():
    return False


def basename(file):
    return os.path.splitext(os.path.basename(file))[0]


def file_list(path, ext=None):
    files = [os.path.join(path, f) for f in os.listdir(path)]
    if ext is not None:
        exts = ext if isinstance(ext, list) else [ext]
        files = [f for f in files if os.path.splitext(f.lower())[1][1:] in exts]
    return sorted(f for f in files if os.path.isfile(f))


def img_list(path):
    return file_list(path, ['jpg', 'jpeg', 'png', 'ppm', 'tif'])


def img_read(path):
    from PIL import Image
    img = np.asarray(Image.open(path))
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=-1)
    return img[:, :, :3]


def txt_clean(txt):
    return txt.translate(str.maketrans(dict.fromkeys(list("\n',.—|!?/:;\\"), ""))).replace(' ', '_').replace('"', '')


def checkout(img, fname=None, verbose=False):
    """utils.py:94-100: CHW float image in [0,1] -> uint8 JPEG on disk"""
    from PIL import Image
    arr = np.transpose(img.detach().cpu().numpy() if torch.is_tensor(img) else np.array(img), (1, 2, 0))
    arr = np.clip(arr * 255, 0, 255).astype(np.uint8)
    if fname is not None:
        Image.fromarray(arr).save(fname, quality=95)
    return arr
