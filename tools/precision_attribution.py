"""CPU only (no GPU minutes).  WHICH f16 quantisation of the HIP ViT path produces the stress-weight loss-curve divergence?

The fp32 torch-CPU oracle (oracle/clip_vit_ref.py) is re-run with ONE tensor class at a time rounded to f16 exactly where the HIP
path holds it in f16 (csrc/vit.hip: the MFMA operands), everything else fp32:

  forward values (rounded in the forward; the backward then differentiates at the rounded values, as the HIP backward does with its
  stored f16 activations):
      patches   the sampler's f16 cuts (A operand of the patch-embed GEMM)
      w_patch w_qkv w_o w_fc1 w_fc2   the f16 weight copies (forward and dgrad use the same rounded values)
      h1 h2     LayerNorm outputs (A operands of the QKV / fc1 GEMMs)
      qkv       the QKV GEMM's f16 output (operands of QK^T and PV)
      p         softmax probabilities (A operand of PV)
      att       attention output (A operand of the out-projection)
      gact      QuickGELU output (A operand of fc2)
      dgelu     the stored f16 GELU derivative (backward only: multiplies d gact)
  gradient stream (identity in the forward, rounded in the backward after multiplication by the loss scale):
      dx16      f16 copies of the fp32 residual gradient (A operand of the fc2 / out-proj dgrad GEMMs)
      du        d u (output of the fc2 dgrad GEMM x GELU', A operand of the fc1 dgrad)
      dh        d h (outputs of the fc1 / QKV dgrad GEMMs, inputs of the LayerNorm backward)
      datt      d att (output of the out-proj dgrad, operand of the attention backward)
      ds        d scores inside the attention backward (operand of dQ / dK)
      dqkv      d qkv (attention backward output, A operand of the QKV dgrad)
      dx0       d x0 (ln_pre backward output, A operand of the patch-embed dgrad)
  all_fwd / all_bwd / all = the unions.

Per class: (1) ONE step at the loss-curve fixture's first crop table: |d loss| and the spectrum-gradient error max|g - g32| / max|g32|
(and its RMS relative to the gradient's RMS); (2) optionally the free-running curve against tests/golden/loss_curve_<name>.npz.

    python tools/precision_attribution.py [--name c2_s32_stress] [--steps 0|N] [--classes a,b,...] [--split cls1,cls2]

--split: the named classes are carried as hi + lo f16 pairs (value rounded to f16 plus the f16-rounded remainder = ~22 bits), i.e.
what `aph_vit_set_precise` would buy for that operand: they are left OUT of the `all` set, to show what remains.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_path as R                                  # noqa: E402
from oracle.make_loss_curves import CONFIGS, seed_all, weights_of       # noqa: E402

LOSS_SCALE = 4096.0
FWD = ['patches', 'w_patch', 'w_qkv', 'w_o', 'w_fc1', 'w_fc2', 'h1', 'h2', 'qkv', 'p', 'att', 'gact', 'dgelu']
EXTRA = ['h1qk', 'h1v']      # sub-classes of h1, not part of the union sets
BWD = ['dx16', 'du', 'dh', 'datt', 'ds', 'dqkv', 'dx0']


def r16(t):
    return t.half().float()


def r16x2(t):          # hi + lo pair of f16 values
    hi = t.half().float()
    return hi + (t - hi).half().float()


class _QF(torch.autograd.Function):          # value rounded in the forward, gradient untouched
    @staticmethod
    def forward(ctx, x):
        return r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _QB(torch.autograd.Function):          # identity in the forward, (loss-scaled) gradient rounded in the backward
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return r16(g * LOSS_SCALE) / LOSS_SCALE


class _QGelu(torch.autograd.Function):       # QuickGELU whose backward multiplies by the f16-STORED derivative (csrc/vit_gemm.h EpiGelu)
    @staticmethod
    def forward(ctx, u, round_d):
        s = torch.sigmoid(1.702 * u)
        g = u * s
        d = s + 1.702 * (g - g * s)
        ctx.save_for_backward(r16(d) if round_d else d)
        return g

    @staticmethod
    def backward(ctx, gg):
        (d,) = ctx.saved_tensors
        return gg * d, None


def make_encoder(w, cfg, q):
    """encode_image with the tensor classes in the set `q` rounded to f16 (see the module docstring)"""
    qf = lambda name, t: _QF.apply(t) if name in q else t
    qb = lambda name, t: _QB.apply(t) if name in q else t
    wq = {k: v for k, v in w.items()}
    for cls, key in (('w_patch', 'conv1.weight'),):
        if cls in q:
            wq[key] = r16(w[key])
    for i in range(cfg['layers']):
        pre = 'transformer.resblocks.%d.' % i
        for cls, key in (('w_qkv', 'attn.in_proj_weight'), ('w_o', 'attn.out_proj.weight'), ('w_fc1', 'mlp.c_fc.weight'), ('w_fc2', 'mlp.c_proj.weight')):
            if cls in q:
                wq[pre + key] = r16(w[pre + key])

    def enc(x):
        width, heads, layers, p = cfg['width'], cfg['heads'], cfg['layers'], cfg['patch_size']
        B = x.shape[0]
        x = qf('patches', x)
        x = F.conv2d(x, wq['conv1.weight'], stride=p)
        x = x.reshape(B, width, -1).permute(0, 2, 1)
        x = qb('dx0', x)                                                  # d x0 (patch rows) is the dgrad GEMM's f16 operand
        cls = wq['class_embedding'].to(x.dtype).expand(B, 1, width)
        x = torch.cat([cls, x], dim=1) + wq['positional_embedding']
        x = F.layer_norm(x, (width,), wq['ln_pre.weight'], wq['ln_pre.bias'], 1e-5)
        T = x.shape[1]
        hd = width // heads
        for i in range(layers):
            pre = 'transformer.resblocks.%d.' % i
            h = F.layer_norm(x, (width,), wq[pre + 'ln_1.weight'], wq[pre + 'ln_1.bias'], 1e-5)
            if 'h1qk' in q or 'h1v' in q:          # [r5] the f16 rounding of h1 seen by the Q / K columns only, or by the V columns only
                hr, Wi, bi = _QF.apply(h), wq[pre + 'attn.in_proj_weight'], wq[pre + 'attn.in_proj_bias']
                qkv = torch.cat([F.linear(hr if 'h1qk' in q else h, Wi[:2 * width], bi[:2 * width]),
                                 F.linear(hr if 'h1v' in q else h, Wi[2 * width:], bi[2 * width:])], dim=-1)
            else:
                h = qb('dh', qf('h1', h))
                qkv = F.linear(h, wq[pre + 'attn.in_proj_weight'], wq[pre + 'attn.in_proj_bias'])
            qkv = qb('dqkv', qf('qkv', qkv))
            qq, k, v = qkv.split(width, dim=-1)
            qq = qq.reshape(B, T, heads, hd).transpose(1, 2) * (hd ** -0.5)
            k = k.reshape(B, T, heads, hd).transpose(1, 2)
            v = v.reshape(B, T, heads, hd).transpose(1, 2)
            s = qb('ds', qq @ k.transpose(-1, -2))
            a = qf('p', torch.softmax(s, dim=-1)) @ v
            a = a.transpose(1, 2).reshape(B, T, width)
            a = qb('datt', qf('att', a))
            o = F.linear(a, wq[pre + 'attn.out_proj.weight'], wq[pre + 'attn.out_proj.bias'])
            x = x + qb('dx16', o)                                         # the branch's gradient = the f16 copy of the residual gradient
            h = F.layer_norm(x, (width,), wq[pre + 'ln_2.weight'], wq[pre + 'ln_2.bias'], 1e-5)
            h = qb('dh', qf('h2', h))
            u = F.linear(h, wq[pre + 'mlp.c_fc.weight'], wq[pre + 'mlp.c_fc.bias'])
            u = qb('du', u)
            g = _QGelu.apply(u, 'dgelu' in q)
            g = qf('gact', g)
            o = F.linear(g, wq[pre + 'mlp.c_proj.weight'], wq[pre + 'mlp.c_proj.bias'])
            x = x + qb('dx16', o)
        x = F.layer_norm(x[:, 0, :], (width,), wq['ln_post.weight'], wq['ln_post.bias'], 1e-5)
        return x @ wq['proj']
    return enc


def new_run(c, wts, cfg, q):
    seed_all(0)
    p0 = R.fft_params_init([1, 3, c['h'], c['w']])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    return R.ReferenceRun(c['h'], c['w'], make_encoder(wts, cfg, q), [(target, 1.0)], params=p0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='c2_s32_stress')
    ap.add_argument('--steps', type=int, default=0, help='free-running steps against the fixture (0 = single-step table only)')
    ap.add_argument('--classes', default=None)
    ap.add_argument('--split', default='', help='classes carried as hi + lo f16 pairs: removed from the union sets')
    ap.add_argument('--cuts', type=int, default=None)
    a = ap.parse_args()
    c = dict(CONFIGS[a.name])
    if a.cuts:
        c['S'] = a.cuts
    cfg, wts = weights_of(c['weights'])
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'loss_curve_%s.npz' % a.name))['loss']
    split = set(s for s in a.split.split(',') if s)
    sets = [(n, {n}) for n in FWD + BWD + EXTRA] + [('all_fwd', set(FWD)), ('all_bwd', set(BWD)), ('all', set(FWD + BWD)),
            ('all_split', set(FWD + BWD) - {'patches', 'h1'}), ('all_split_qk_only', (set(FWD + BWD) - {'patches', 'h1'}) | {'h1v'})]      # [r5] the shipped split mode; the split on the Q / K columns only
    if a.classes:
        want = a.classes.split(',')
        sets = [(n, s) for n, s in sets if n in want]
    sets = [(n if not (split and len(s) > 1) else n + ' minus ' + '+'.join(sorted(split)), s - split if len(s) > 1 else s) for n, s in sets]
    # fp32 baseline of step 0
    seed_all(9)
    table0 = R.draw_crop_table(c['S'], 224, c['h'], c['w'], 'uniform', 0.4)
    base = new_run(c, wts, cfg, set())
    l32 = base.loss(table0)
    base.opt.zero_grad(); l32.backward()
    g32 = base.params.grad.detach().clone()
    gmax, grms = g32.abs().max().item(), g32.pow(2).mean().sqrt().item()
    print('# %s: %dx%d, %d cuts, %s weights; fp32 step-0 loss %.7f, max|g| %.3e, rms g %.3e; torch %s, %d threads'
          % (a.name, c['w'], c['h'], c['S'], c['weights'], float(l32), gmax, grms, torch.__version__, torch.get_num_threads()), flush=True)
    print('%-22s %12s %14s %14s %s' % ('f16 class', '|d loss| s0', 'max|dg|/max|g|', 'rms dg/rms g', ('curve over %d steps: max |d loss|, first step past 1e-3' % a.steps) if a.steps else ''), flush=True)
    for name, q in sets:
        t0 = time.time()
        run = new_run(c, wts, cfg, q)
        seed_all(9)
        worst, first, line = 0.0, None, ''
        for i in range(max(a.steps, 1)):
            table = R.draw_crop_table(c['S'], 224, c['h'], c['w'], 'uniform', 0.4)
            loss = run.loss(table)
            run.opt.zero_grad(); loss.backward()
            if i == 0:
                g = run.params.grad.detach()
                dl0 = abs(float(loss.detach()) - float(l32))
                emax = (g - g32).abs().max().item() / gmax
                erms = (g - g32).pow(2).mean().sqrt().item() / grms
            if not a.steps:
                break
            run.opt.step(); run.i += 1
            d = abs(float(loss.detach()) - fx[i])
            worst = max(worst, d)
            if d > 1e-3 and first is None:
                first = i
        tail = ('%.2e  first %s' % (worst, first)) if a.steps else ''
        print('%-22s %12.2e %14.2e %14.2e %s   (%.0f s)' % (name, dl0, emax, erms, tail, time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
