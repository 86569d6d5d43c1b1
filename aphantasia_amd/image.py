"""Drop-in for aphantasia/image.py's image parameterisers, computed by the HIP library.

Public surface (same names, argument meaning and return values as the reference):
    to_valid_rgb(image_f, colors=1., decorrelate=True)                 image.py:14-29
    fft_image(shape, sd=0.01, decay_power=1.0, resume=None)            image.py:152-177
    pixel_image(shape, resume=None, sd=1.)                             image.py:98-119
Parameters stay torch leaf tensors owned by Python (so torch.save / any torch.optim keep working);
`image_f()` returns a tensor that is differentiable w.r.t. them through torch.autograd, with the
forward and backward running as HIP kernels (csrc/synth.hip).
"""
import os

import numpy as np
import torch

from . import ops

_DEV = 'cuda'


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('aphantasia_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path')
    return torch.device(_DEV)


def colcorr_t(colors=1.0):
    """image.py:15-19 -> float32 [3,3] = normalised colour matrix, transposed"""
    m = torch.tensor([[0.26, 0.09, 0.02], [0.27, 0.00, -0.05], [0.27, -0.09, 0.03]])
    m = m / torch.tensor([colors, 1., 1.])
    m = m / m.norm(dim=0).max()
    return m.T.contiguous()


def rfft2d_freqs(h, w):
    """image.py:122-128"""
    fy = np.fft.fftfreq(h)[:, None]
    w2 = (w + 1) // 2 if w % 2 == 1 else w // 2 + 1
    fx = np.fft.fftfreq(w)[:w2]
    return np.sqrt(fx * fx + fy * fy)


def fft_scale(h, w, decay_power):
    """image.py:158-162 -> float32 [h, w//2+1] on the host"""
    freqs = rfft2d_freqs(h, w)
    scale = 1. / np.maximum(freqs, 4. / max(h, w)) ** decay_power
    scale *= np.sqrt(h * w)
    return torch.tensor(scale).float()


class _SynthFFT(torch.autograd.Function):
    """rgb (or the normalised pre-rgb image) from the spectrum: K1-K4 fused (csrc/synth.hip)."""

    @staticmethod
    def forward(ctx, params, gen, shift, contrast, cc, decorrelate, to_rgb):
        p = params.reshape(3, gen.h, gen.wc, 2)
        raw, out = ops.synth_fft_fwd(gen.plan, p.contiguous(), gen.scale, shift, contrast,
                                     cc if to_rgb else None, decorrelate and to_rgb)
        if not to_rgb:
            # image_f() without to_valid_rgb: image * contrast / std  (image.py:174)
            stats = torch.empty(2, dtype=torch.float32, device=raw.device)
            gen.plan.lib.call('aph_synth_stats', gen.plan.handle, ops.ptr(stats), ops._stream(raw))
            out = raw * (contrast / stats[1])
            ctx.plain = True
            ctx.save_for_backward(raw, stats)
            ctx.contrast = contrast
            ctx.gen = gen
            return out.unsqueeze(0)
        stats = torch.empty(2, dtype=torch.float32, device=raw.device)
        gen.plan.lib.call('aph_synth_stats', gen.plan.handle, ops.ptr(stats), ops._stream(raw))
        ctx.plain = False
        ctx.gen, ctx.contrast, ctx.cc, ctx.decorrelate = gen, contrast, cc, decorrelate
        ctx.save_for_backward(raw, out, stats)
        return out.unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        gen = ctx.gen
        g = g.reshape(3, gen.h, gen.w).contiguous().float()
        if ctx.plain:
            raw, stats = ctx.saved_tensors
            # adjoint of x*c/std through the identity colour path: rgb' = 1 is emulated by feeding the
            # spatial adjoint pieces directly: d raw = c/s g - c sum(g x)/(s^3 (N-1)) (x - mean)
            n = raw.numel()
            s, mu, c = stats[1], stats[0], ctx.contrast
            sgx = (g.double() * raw.double()).sum().float()
            draw = (c / s) * g - (c * sgx / (s ** 3 * (n - 1))) * (raw - mu)
            ones = torch.full_like(raw, 0.5)      # sigmoid'(0)*4 = 1: feed rgb = 0.5 with 4x gradient, identity colour
            gen.plan.lib.call('aph_synth_set_stats', gen.plan.handle, ops.ptr(torch.stack([mu * 0, s * 0 + 1]).contiguous()),
                              ops._stream(raw))
            grad = ops.synth_fft_bwd(gen.plan, (draw * 4.0).contiguous(), ones, raw * 0, gen.scale, 1.0, None, False)
            return grad.reshape(gen.param_shape), None, None, None, None, None, None
        raw, rgb, stats = ctx.saved_tensors
        gen.plan.lib.call('aph_synth_set_stats', gen.plan.handle, ops.ptr(stats), ops._stream(raw))
        grad = ops.synth_fft_bwd(gen.plan, g, rgb, raw, gen.scale, ctx.contrast, ctx.cc, ctx.decorrelate)
        return grad.reshape(gen.param_shape), None, None, None, None, None, None


class FFTImage:
    """The `image_f` closure of fft_image (image.py:164-175) as a callable object, so that
    to_valid_rgb can fuse the colour/sigmoid stage into the same HIP launches."""

    def __init__(self, params, h, w, decay_power):
        self.params = params
        self.h, self.w, self.wc = h, w, w // 2 + 1
        self.param_shape = tuple(params.shape)
        self.scale = fft_scale(h, w, decay_power).to(params.device).contiguous()
        self.plan = ops.SynthPlan(3, h, w)

    def _shift(self, shift):
        if shift is None:
            return None
        s = torch.as_tensor(shift, dtype=torch.float32, device=self.params.device)
        return s.expand(1, 1, self.h, self.wc, 1).reshape(self.h, self.wc).contiguous()

    def __call__(self, shift=None, contrast=1., *noargs, **nokwargs):
        return _SynthFFT.apply(self.params, self, self._shift(shift), float(contrast), None, False, False)

    def rgb(self, cc, decorrelate, shift=None, contrast=1., *noargs, **nokwargs):
        return _SynthFFT.apply(self.params, self, self._shift(shift), float(contrast), cc, decorrelate, True)


# ---- resume-from-image helpers (host, one-off; image.py:179-220) -------------------------------------------------
_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def inv_sigmoid(x):
    """image.py:179-183"""
    eps = 1.e-12
    x = torch.clamp(x.double(), eps, 1 - eps)
    return torch.log(x / (1 - x)).float()


def un_rgb(image, colors=1.):
    """image.py:185-197: inverse of to_valid_rgb's colour mixing, applied to the CLIP-normalised image (the reference's
    'experimental' choice; inv_sigmoid is commented out upstream).  numpy uint8 HWC or a float NCHW tensor in."""
    cc_inv = torch.linalg.inv(colcorr_t(colors))
    if not isinstance(image, torch.Tensor):
        image = torch.Tensor(np.array(image)).permute(2, 0, 1).unsqueeze(0) / 255.
    mean = torch.tensor(_CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(_CLIP_STD).view(1, 3, 1, 1)
    image = (image.cpu().float() - mean) / std                      # transforms.normalize()
    return torch.einsum('nchw,cd->ndhw', image, cc_inv)


def un_spectrum(spectrum, decay_power):
    """image.py:199-206 (note the 1/max(w,h) floor here vs 4/max(w,h) in fft_image)"""
    h = spectrum.shape[2]
    w = (spectrum.shape[3] - 1) * 2
    freqs = rfft2d_freqs(h, w)
    scale = 1.0 / np.maximum(freqs, 1.0 / max(w, h)) ** decay_power
    scale *= np.sqrt(w * h)
    scale = torch.tensor(scale).float()[None, None, ..., None]
    return spectrum / scale


def img2fft(img_in, decay=1., colors=1.):
    """image.py:208-220 -> spectrum [1,3,h,w//2+1,2] (host tensor)"""
    image_t = un_rgb(img_in, colors=colors)
    h, w = image_t.shape[2], image_t.shape[3]
    with torch.no_grad():
        spectrum = torch.view_as_real(torch.fft.rfftn(image_t, s=(h, w), dim=[2, 3], norm='ortho'))
        spectrum = un_spectrum(spectrum, decay_power=decay)
        spectrum = spectrum * 500000.                               # [sic], image.py:219
    return spectrum


def resume_fft(resume=None, shape=None, decay=None, colors=1.6, sd=0.01):
    """image.py:130-150.  Random init draws on torch's CPU generator exactly as the reference does
    (0.01*randn on the host, then moved to the device).  `.pt` snapshots and image files (img2fft) resume as upstream."""
    size = None
    if resume is None:
        params_shape = [*shape[:3], shape[3] // 2 + 1, 2]
        params = 0.01 * torch.randn(*params_shape)
    elif isinstance(resume, str):
        if not os.path.isfile(resume):
            print(' Snapshot not found:', resume)
            exit()
        if os.path.splitext(resume)[1].lower()[1:] in ['jpg', 'png', 'tif', 'bmp']:
            from .utils import img_read
            img_in = img_read(resume)
            params = img2fft(img_in, decay, colors)
            size = img_in.shape[:2]
        else:
            params = torch.load(resume)
            if isinstance(params, list):
                params = params[0]
        params = params.detach().float() * sd
    else:
        if isinstance(resume, list):
            resume = resume[0]
        params = resume
    return params.to(_device()).contiguous(), size


def fft_image(shape, sd=0.01, decay_power=1.0, resume=None):
    """image.py:152-177 -> ([spectrum_real_imag_t], image_f, size)"""
    params, size = resume_fft(resume, shape, decay_power, sd=sd)
    spectrum_real_imag_t = params.requires_grad_(True)
    if size is not None:
        shape[2:] = size
    h, w = list(shape[2:])
    if list(params.shape) != [1, 3, h, w // 2 + 1, 2]:
        raise ValueError('spectrum shape %s does not match image shape %s' % (list(params.shape), shape))
    return [spectrum_real_imag_t], FFTImage(spectrum_real_imag_t, h, w, decay_power), size


class _SpatialRGB(torch.autograd.Function):
    """to_valid_rgb over a spatial-domain image: x*contrast/std (or /fixed_div) -> colour -> sigmoid."""

    @staticmethod
    def forward(ctx, image, plan, contrast, fixed_div, cc, decorrelate):
        raw = image.reshape(3, plan.H, plan.W).contiguous().float()
        rgb = ops.synth_spatial_fwd(plan, raw, contrast, fixed_div, cc, decorrelate)
        stats = torch.empty(2, dtype=torch.float32, device=raw.device)
        plan.lib.call('aph_synth_stats', plan.handle, ops.ptr(stats), ops._stream(raw))
        ctx.plan, ctx.args = plan, (contrast, fixed_div, cc, decorrelate)
        ctx.save_for_backward(raw, rgb, stats)
        ctx.shape = image.shape
        return rgb.unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        raw, rgb, stats = ctx.saved_tensors
        plan = ctx.plan
        contrast, fixed_div, cc, decorrelate = ctx.args
        plan.lib.call('aph_synth_set_stats', plan.handle, ops.ptr(stats), ops._stream(raw))
        d = ops.synth_spatial_bwd(plan, g.reshape(3, plan.H, plan.W).contiguous().float(), rgb, raw, contrast, fixed_div, cc, decorrelate)
        return d.reshape(ctx.shape), None, None, None, None, None


class PixelImage:
    """pixel_image's closure (image.py:114-118)."""

    def __init__(self, image_t):
        self.image_t = image_t
        self.plan = ops.SynthPlan(3, image_t.shape[2], image_t.shape[3])

    def __call__(self, shift=None, contrast=1., fixcontrast=False):
        if fixcontrast is True:
            return self.image_t * contrast / 3.3
        return self.image_t * contrast / self.image_t.std()

    def rgb(self, cc, decorrelate, shift=None, contrast=1., fixcontrast=False):
        return _SpatialRGB.apply(self.image_t, self.plan, float(contrast), 3.3 if fixcontrast is True else 0.0, cc, decorrelate)


def pixel_image(shape, resume=None, sd=1., *noargs, **nokwargs):
    """image.py:98-119 -> ([image_t], image_f, size)"""
    size = None
    if resume is None:
        image_t = torch.randn(*shape) * sd
    elif isinstance(resume, str):
        if not os.path.isfile(resume):
            print(' Image not found:', resume)
            exit()
        from .utils import img_read
        img_in = img_read(resume)
        image_t = 3.3 * un_rgb(img_in, colors=2.)
        size = img_in.shape[:2]
        print(resume, size)
    else:
        if isinstance(resume, list):
            resume = resume[0]
        image_t = resume
    image_t = image_t.to(_device()).float().contiguous().requires_grad_(True)
    return [image_t], PixelImage(image_t), size


class _SynthDWT(torch.autograd.Function):
    """dwt_image.inner (+ to_valid_rgb when `cc` is given): inverse DWT levels + K3/K4 (csrc/dwt.hip, synth.hip)"""

    @staticmethod
    def forward(ctx, gen, contrast, cc, decorrelate, to_rgb, *Ys):
        flat = gen.flat_for(Ys)
        raw = gen.synth.forward(flat)
        if to_rgb:
            out = ops.synth_spatial_fwd(gen.plan, raw, contrast, 0.0, cc, decorrelate)
        else:
            out = ops.synth_spatial_fwd(gen.plan, raw, contrast, 0.0, None, False)      # still needs the std
        stats = torch.empty(2, dtype=torch.float32, device=raw.device)
        gen.plan.lib.call('aph_synth_stats', gen.plan.handle, ops.ptr(stats), ops._stream(raw))
        ctx.gen, ctx.args, ctx.to_rgb = gen, (contrast, cc, decorrelate), to_rgb
        if not to_rgb:
            out = raw * (contrast / stats[1])
        ctx.save_for_backward(raw.clone(), out, stats)
        return out.unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        raw, out, stats = ctx.saved_tensors
        gen = ctx.gen
        contrast, cc, decorrelate = ctx.args
        g = g.reshape(raw.shape).contiguous().float()
        if ctx.to_rgb:
            gen.plan.lib.call('aph_synth_set_stats', gen.plan.handle, ops.ptr(stats), ops._stream(raw))
            d_raw = ops.synth_spatial_bwd(gen.plan, g, out, raw, contrast, 0.0, cc, decorrelate)
        else:
            n, s, mu = raw.numel(), stats[1], stats[0]
            sgx = (g.double() * raw.double()).sum().float()
            d_raw = (contrast / s) * g - (contrast * sgx / (s ** 3 * (n - 1))) * (raw - mu)
        grad = torch.empty(gen.synth.numel, dtype=torch.float32, device=raw.device)
        gen.synth.backward(d_raw.contiguous(), grad)
        return (None, None, None, None, None, *[v.clone() for v in gen.synth.views(grad)])


class DWTImage:
    """dwt_image's closure (image.py:64-69).  `Ys` are leaf views into one flat coefficient buffer."""

    def __init__(self, h, w, wave, sharp, device):
        from .dwt import DWTSynth
        self.synth = DWTSynth(h, w, wave, sharp, device)
        self.flat = torch.empty(self.synth.numel, dtype=torch.float32, device=device)
        self.Ys = None
        self.plan = ops.SynthPlan(3, self.synth.H, self.synth.W)

    def flat_for(self, Ys):
        """the flat buffer if Ys are (still) our views, else a packed copy"""
        views = self.synth.views(self.flat)
        if all(y.data_ptr() == v.data_ptr() for y, v in zip(Ys, views)):
            return self.flat
        return torch.cat([y.detach().reshape(-1).float() for y in Ys]).contiguous()

    def __call__(self, shift=None, contrast=1.):
        return _SynthDWT.apply(self, float(contrast), None, False, False, *self.Ys)

    def rgb(self, cc, decorrelate, shift=None, contrast=1.):
        return _SynthDWT.apply(self, float(contrast), cc, decorrelate, True, *self.Ys)


def img2dwt(img_in, wave='coif2', sharp=0.3, colors=1.):
    """image.py:82-94: un_rgb -> DWTForward(J = max level, wave, 'symmetric') -> detail levels divided by dwt_scale.
    Returns the `Ys` list [Yl, Yh finest first ...] (host tensors)."""
    from .dwt import dwt_forward_host, dwt_scale_from_sizes
    image_t = un_rgb(img_in, colors=colors)
    yl, yh = dwt_forward_host(image_t, wave)
    scale = dwt_scale_from_sizes([tuple(y.shape[-2:]) for y in yh], sharp)
    return [yl] + [y / scale[i] for i, y in enumerate(yh)]


def dwt_image(shape, wave='coif2', sharp=0.3, colors=1., resume=None):
    """image.py:61-71 -> (Ys, image_f, size).  Random init draws randn per tensor on the CPU generator in the
    reference's order (Yl, then detail levels finest first; image.py:41-42).  `resume`: list of tensors, a .pt file, or an
    image file (img2dwt; the image then sets the size)."""
    h, w = shape[2:]
    size = None
    if isinstance(resume, str):
        if not os.path.isfile(resume):
            print(' Snapshot not found:', resume)
            exit()
        if os.path.splitext(resume)[1].lower()[1:] in ['jpg', 'png', 'tif', 'bmp']:      # image.py:43-50: the image sets the size
            from .utils import img_read
            img_in = img_read(resume)
            resume = img2dwt(img_in, wave, sharp, colors)
            size = img_in.shape[:2]
            h, w = int(size[0]), int(size[1])
        else:
            resume = torch.load(resume)
    gen = DWTImage(h, w, wave, sharp, _device())
    views = gen.synth.views(gen.flat)
    if resume is None:
        init = [torch.randn(*v.shape) for v in views]
    else:
        init = [y.detach().float() for y in resume]
        if [tuple(y.shape) for y in init] != [tuple(v.shape) for v in views]:
            raise ValueError('snapshot coefficient shapes do not match a %dx%d %s transform' % (w, h, wave))
    for v, y in zip(views, init):
        v.copy_(y)
    gen.Ys = [v.requires_grad_(True) for v in views]
    return gen.Ys, gen, size


def to_valid_rgb(image_f, colors=1., decorrelate=True):
    """image.py:14-29: returns inner(*args, **kwargs) -> sigmoid(colour-decorrelated image) in (0,1)."""
    cc = colcorr_t(colors).flatten().tolist()
    generic_plan = {}

    def inner(*args, **kwargs):
        if hasattr(image_f, 'rgb'):                 # FFTImage / PixelImage / DWTImage: fully fused HIP path
            return image_f.rgb(cc, decorrelate, *args, **kwargs)
        image = image_f(*args, **kwargs)            # any other callable: colour + sigmoid kernels on its output
        key = tuple(image.shape[2:])
        if key not in generic_plan:
            generic_plan[key] = ops.SynthPlan(3, key[0], key[1])
        return _SpatialRGB.apply(image, generic_plan[key], 1.0, 1.0, cc, decorrelate)
    return inner
