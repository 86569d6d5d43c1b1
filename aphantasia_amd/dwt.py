"""Wavelet (DWT) image parameteriser: host side of `dwt_image` (aphantasia/image.py:33-80).

All coefficient tensors live in ONE flat fp32 buffer (Yl, then every detail level, finest first -- the order
of the reference's `Ys` list); the per-tensor views handed to Python share that storage, so the fused engine
can run Adam and the multi-GPU all-reduce on the flat buffer while `torch.save(Ys)` / torch.optim keep working.
The inverse transform is one C-ABI call (csrc/dwt.hip): one launch per level, coarsest first, the single-tile coarse tail in one launch.
"""
import ctypes
import math

import torch

from . import _ffi, ops
from .wavelet_filters import REC_LO


def max_level(h, w):
    """pywt.WaveletPacket2D(zeros(h,w), 'db1', 'symmetric').maxlevel (image.py:35-36) = floor(log2(min(h,w)))"""
    return int(math.floor(math.log2(min(h, w))))


def coeff_shapes(h, w, wave):
    """DWTForward(J, wave, 'symmetric') coefficient sizes, finest first: n -> (n + L - 1) // 2"""
    if wave not in REC_LO:
        raise ValueError('unknown wavelet %r (available: %s)' % (wave, ', '.join(sorted(REC_LO))))
    L = len(REC_LO[wave])
    J = max_level(h, w)
    sizes = []
    for _ in range(J):
        h, w = (h + L - 1) // 2, (w + L - 1) // 2
        sizes.append((h, w))
    return J, sizes


def dwt_scale_from_sizes(sizes, sharp):
    """image.py:73-80"""
    h0, w0 = sizes[0]
    return [((h0 * w0) / (h * w)) ** (1. - sharp) for (h, w) in sizes]


def dwt_forward_host(x, wave, J=None):
    """pytorch_wavelets.DWTForward(J, wave, mode='symmetric') on the host (one-off: resuming the wavelet parameters from an
    image, image.py:82-94).  x: [N,C,H,W] tensor -> (yl [N,C,h,w], [yh_j [N,C,3,h_j,w_j]] finest first, bands (LH, HL, HH) =
    pywt's (cH, cV, cD)).  Per axis: half-sample symmetric extension, correlation with the reversed decomposition filters
    (dec_lo = rec_lo[::-1]; dec_hi = QMF), keep every second sample: c[i] = sum_j dec[j] x_ext[2 i + 1 - j]."""
    import numpy as np
    rec_lo = np.asarray(REC_LO[wave], dtype=np.float64)
    L = len(rec_lo)
    dec_lo = rec_lo[::-1].copy()
    rec_hi = np.array([(-1) ** k * rec_lo[L - 1 - k] for k in range(L)])
    dec_hi = rec_hi[::-1].copy()

    def afb(a, axis):
        a = np.moveaxis(a, axis, -1)
        n = a.shape[-1]
        out = (n + L - 1) // 2
        ext = np.pad(a, [(0, 0)] * (a.ndim - 1) + [(L - 1, L - 1)], mode='symmetric')
        idx = (2 * np.arange(out)[:, None] + 1 - np.arange(L)[None, :]) + (L - 1)          # [out, L] positions in ext
        g = ext[..., idx]                                                                  # [..., out, L]
        lo, hi = (g * dec_lo).sum(-1), (g * dec_hi).sum(-1)
        return np.moveaxis(lo, -1, axis), np.moveaxis(hi, -1, axis)

    a = x.detach().cpu().double().numpy()
    J = max_level(a.shape[-2], a.shape[-1]) if J is None else J
    yh = []
    for _ in range(J):
        lo_w, hi_w = afb(a, 3)                      # along width
        ll, lh = afb(lo_w, 2)                       # then along height: (low-w, low-h), (low-w, high-h) = cH
        hl, hh = afb(hi_w, 2)                       # (high-w, low-h) = cV, (high-w, high-h) = cD
        yh.append(torch.from_numpy(np.stack([lh, hl, hh], axis=2)).float())
        a = ll
    return torch.from_numpy(a).float(), yh


class DWTSynth:
    """Coefficient storage + the level-by-level inverse transform and its adjoint."""

    def __init__(self, h, w, wave, sharp, device, lib=None, C=3):
        self.lib = lib if lib is not None else _ffi.lib()
        self.C, self.wave, self.sharp = C, wave, sharp
        self.J, self.sizes = coeff_shapes(h, w, wave)
        self.L = len(REC_LO[wave])
        g0 = torch.tensor(REC_LO[wave], dtype=torch.float64)
        g1 = torch.tensor([(-1) ** k * REC_LO[wave][self.L - 1 - k] for k in range(self.L)], dtype=torch.float64)
        self.g0, self.g1 = g0.float().to(device), g1.float().to(device)
        self.scale = dwt_scale_from_sizes(self.sizes, sharp)
        # flat layout: [Yl | Yh_0 (finest) | Yh_1 | ...]
        hJ, wJ = self.sizes[-1]
        self.shapes = [(1, C, hJ, wJ)] + [(1, C, 3, hh, ww) for (hh, ww) in self.sizes]
        self.offsets, n = [], 0
        for s in self.shapes:
            self.offsets.append(n)
            n += math.prod(s)
        self.numel = n
        # running low band after each level, coarsest -> finest; out size of level j = 2 n_j - L + 2
        self.out_sizes = [(2 * hh - self.L + 2, 2 * ww - self.L + 2) for (hh, ww) in self.sizes]
        self.H, self.W = self.out_sizes[0]
        self.bufs = [torch.empty(C, *s, dtype=torch.float32, device=device) for s in self.out_sizes]
        self.gbufs = [torch.empty(C, *s, dtype=torch.float32, device=device) for s in self.out_sizes]

    def views(self, flat):
        return [flat[o:o + math.prod(s)].view(s) for o, s in zip(self.offsets, self.shapes)]

    def _level_arrays(self):
        """host-side size / gain arrays of the all-levels calls (built once)"""
        if getattr(self, '_lv', None) is None:
            J = self.J
            IntJ, FltJ, PtrJ = ctypes.c_int * J, ctypes.c_float * J, ctypes.c_void_p * J
            self._lv = dict(hs=IntJ(*[s[0] for s in self.sizes]), ws=IntJ(*[s[1] for s in self.sizes]),
                            sc=FltJ(*[float(v) for v in self.scale]), bufs=PtrJ(*[b.data_ptr() for b in self.bufs]),
                            gbufs=PtrJ(*[b.data_ptr() for b in self.gbufs]), PtrJ=PtrJ)
        return self._lv

    def forward(self, flat):
        """flat coefficient buffer -> raw image [C, H, W] (the tensor is owned by this object): every level in one C-ABI call"""
        ys = self.views(flat)
        lv = self._level_arrays()
        highs = lv['PtrJ'](*[y.data_ptr() for y in ys[1:]])
        self.lib.call('aph_idwt_fwd', ops.ptr(ys[0]), highs, lv['hs'], lv['ws'], lv['sc'], self.J, self.C, ops.ptr(self.g0), ops.ptr(self.g1),
                      self.L, lv['bufs'], ops._stream(flat))
        return self.bufs[0]

    def forward_per_level(self, flat):
        """the same transform as one aph_idwt_level_fwd call per level (tests and per-level timing: tools/exp/dwt_levels.py)"""
        ys = self.views(flat)
        st = ops._stream(flat)
        ll, llh, llw = ys[0], self.sizes[-1][0], self.sizes[-1][1]
        for j in range(self.J - 1, -1, -1):
            hh, ww = self.sizes[j]
            self.lib.call('aph_idwt_level_fwd', ops.ptr(ll), llh, llw, ops.ptr(ys[1 + j]), hh, ww, self.C, ops.ptr(self.g0),
                          ops.ptr(self.g1), self.L, float(self.scale[j]), ops.ptr(self.bufs[j]), st)
            ll, (llh, llw) = self.bufs[j], self.out_sizes[j]
        return self.bufs[0]

    def backward_per_level(self, d_raw, grad_flat):
        gs = self.views(grad_flat)
        st = ops._stream(grad_flat)
        g = d_raw
        for j in range(self.J):
            hh, ww = self.sizes[j]
            if j + 1 < self.J:
                dst, (llh, llw) = self.gbufs[j + 1], self.out_sizes[j + 1]
            else:
                dst, (llh, llw) = gs[0], self.sizes[-1]
            self.lib.call('aph_idwt_level_bwd', ops.ptr(g), hh, ww, self.C, ops.ptr(self.g0), ops.ptr(self.g1), self.L,
                          float(self.scale[j]), ops.ptr(dst), llh, llw, ops.ptr(gs[1 + j]), st)
            g = dst
        return grad_flat

    def backward(self, d_raw, grad_flat):
        """d_raw [C,H,W] -> gradient w.r.t. every coefficient, written into grad_flat (same layout as the params)"""
        gs = self.views(grad_flat)
        lv = self._level_arrays()
        ghighs = lv['PtrJ'](*[g.data_ptr() for g in gs[1:]])
        if not d_raw.is_contiguous() or tuple(d_raw.shape[-2:]) != (self.H, self.W):
            raise ValueError('d_raw must be a contiguous [C,%d,%d] tensor' % (self.H, self.W))
        self.lib.call('aph_idwt_bwd', ops.ptr(d_raw), lv['hs'], lv['ws'], lv['sc'], self.J, self.C, ops.ptr(self.g0), ops.ptr(self.g1), self.L,
                      lv['gbufs'], ops.ptr(gs[0]), ghighs, ops._stream(grad_flat))
        return grad_flat
