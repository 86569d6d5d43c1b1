"""TEST INFRASTRUCTURE -- CPU oracle, not part of the product path.

Restatement of the reference's DWT parameteriser (aphantasia/image.py:33-80: init_dwt, dwt_image, dwt_scale).
Its arithmetic lives in pytorch_wavelets (`git+https://github.com/fbcotter/pytorch_wavelets`,
requirements.txt:17, unpinned) on top of PyWavelets (>=1.1.1, requirements.txt:16), both absent from the
main interpreter.  Restated from pytorch_wavelets' published DWTInverse / lowlevel.SFB2D / sfb1d
(mode='symmetric'): per level, coarsest first, drop the last row/col of the running low band if it is one
larger than the level's detail bands, then
    lo = sfb1d(ll, LH, dim=H);  hi = sfb1d(HL, HH, dim=H);  ll = sfb1d(lo, hi, dim=W)
    sfb1d(a, b) = conv_transpose(a, rec_lo, stride 2, padding L-2) + conv_transpose(b, rec_hi, ...)
Pinned numerically against PyWavelets 1.1.1 `waverec2(..., 'symmetric')` run out-of-process through
oracle/pywt_dump.py (tests/test_oracle_dwt.py; subbands (LH, HL, HH) <-> (cH, cV, cD)).
"""
import math

import torch
import torch.nn.functional as F

from aphantasia_amd.wavelet_filters import REC_LO


def filters(wave):
    g0 = torch.tensor(REC_LO[wave], dtype=torch.float64)
    L = g0.numel()
    g1 = torch.tensor([(-1) ** k * REC_LO[wave][L - 1 - k] for k in range(L)], dtype=torch.float64)
    return g0, g1


def max_level(h, w):
    """pywt.WaveletPacket2D(zeros(h,w), 'db1', 'symmetric').maxlevel == dwt_max_level(min(h,w), 2)  (image.py:35-36)"""
    return int(math.floor(math.log2(min(h, w))))


def coeff_shapes(h, w, wave):
    """DWTForward(J, wave, 'symmetric') output sizes: n -> floor((n + L - 1) / 2) per level (finest first)."""
    L = len(REC_LO[wave])
    J = max_level(h, w)
    sizes = []
    for _ in range(J):
        h, w = (h + L - 1) // 2, (w + L - 1) // 2
        sizes.append((h, w))
    return J, sizes


def init_params(shape, wave):
    """image.py:40-42: randn (std 1) for Yl then every Yh level, finest first, on the CPU generator."""
    J, sizes = coeff_shapes(shape[2], shape[3], wave)
    Ys = [torch.randn(shape[0], shape[1], *sizes[-1])]
    Ys += [torch.randn(shape[0], shape[1], 3, *s) for s in sizes]
    return Ys


def dwt_scale(Ys, sharp):
    """image.py:73-80"""
    h0, w0 = Ys[1].shape[3:5]
    return [((h0 * w0) / (Ys[i + 1].shape[3] * Ys[i + 1].shape[4])) ** (1. - sharp) for i in range(len(Ys) - 1)]


def sfb1d(lo, hi, g0, g1, dim):
    C = lo.shape[1]
    L = g0.numel()
    shape = [1, 1, 1, 1]
    shape[dim] = L
    k0 = g0.to(lo.dtype).reshape(shape).repeat(C, 1, 1, 1)
    k1 = g1.to(lo.dtype).reshape(shape).repeat(C, 1, 1, 1)
    s = (2, 1) if dim == 2 else (1, 2)
    pad = (L - 2, 0) if dim == 2 else (0, L - 2)
    return F.conv_transpose2d(lo, k0, stride=s, padding=pad, groups=C) + F.conv_transpose2d(hi, k1, stride=s, padding=pad, groups=C)


def idwt(yl, yh, wave):
    """DWTInverse(wave, 'symmetric')((yl, yh)), yh finest first"""
    g0, g1 = filters(wave)
    ll = yl
    for h in yh[::-1]:
        if ll.shape[-2] > h.shape[-2]:
            ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]:
            ll = ll[..., :-1]
        lh, hl, hh = torch.unbind(h, dim=2)
        lo = sfb1d(ll, lh, g0, g1, 2)
        hi = sfb1d(hl, hh, g0, g1, 2)
        ll = sfb1d(lo, hi, g0, g1, 3)
    return ll


def dwt_image_raw(Ys, wave, sharp=0.3):
    """image.py:67: ifm((Ys[0], [Ys[i+1] * scale[i]]))"""
    scale = dwt_scale(Ys, sharp)
    return idwt(Ys[0], [Ys[i + 1] * float(scale[i]) for i in range(len(Ys) - 1)], wave)


def synth_dwt(Ys, wave, cc_t, sharp=0.3, contrast=1.0):
    from oracle.reference_path import std_normalise, to_rgb
    return to_rgb(std_normalise(dwt_image_raw(Ys, wave, sharp), contrast), cc_t)
