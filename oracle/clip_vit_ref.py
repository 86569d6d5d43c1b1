"""TEST INFRASTRUCTURE -- CPU oracle, not part of the product path.

fp32 torch restatement of `model.encode_image` (call site clip_fft.py:254) for
the ViT CLIP models.  The arithmetic lives in the third-party package
openai/CLIP (`pip install git+https://github.com/openai/CLIP.git`,
README.md:31, unpinned master), which is NOT vendored under /root/reference.
Restated from its published `clip/model.py` structure:

  conv1 (patch x patch, stride patch, no bias) -> [B, g*g, width]
  -> prepend class_embedding -> + positional_embedding -> ln_pre
  -> layers x { x += MHA(ln_1(x)); x += c_proj(QuickGELU(c_fc(ln_2(x)))) }
  -> ln_post(x[:, 0]) @ proj

Weights are held as a flat dict in the OpenAI checkpoint key layout
(`visual.` prefix stripped).  Pinned against HF transformers'
CLIPVisionModelWithProjection through `to_hf_state_dict` (tests/test_oracle.py).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


def vit_config(name):
    name = name.replace('/', '-').replace('ViT-', '').upper()
    if name in ('B-32', 'B32'):
        return dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512)
    if name in ('B-16', 'B16'):
        return dict(input_resolution=224, patch_size=16, width=768, layers=12, heads=12, output_dim=512)
    raise ValueError('unsupported CLIP visual model: %s' % name)


def encode_image(w, x, cfg):
    """x [B,3,R,R] fp32 (already CLIP-normalised) -> [B, output_dim]."""
    width, heads, layers, p = cfg['width'], cfg['heads'], cfg['layers'], cfg['patch_size']
    B = x.shape[0]
    x = F.conv2d(x, w['conv1.weight'], stride=p)                       # [B,width,g,g]
    x = x.reshape(B, width, -1).permute(0, 2, 1)                        # [B,g*g,width]
    cls = w['class_embedding'].to(x.dtype).expand(B, 1, width)
    x = torch.cat([cls, x], dim=1) + w['positional_embedding']
    x = F.layer_norm(x, (width,), w['ln_pre.weight'], w['ln_pre.bias'], 1e-5)
    T = x.shape[1]
    hd = width // heads
    for i in range(layers):
        pre = 'transformer.resblocks.%d.' % i
        h = F.layer_norm(x, (width,), w[pre + 'ln_1.weight'], w[pre + 'ln_1.bias'], 1e-5)
        qkv = F.linear(h, w[pre + 'attn.in_proj_weight'], w[pre + 'attn.in_proj_bias'])
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(B, T, heads, hd).transpose(1, 2) * (hd ** -0.5)
        k = k.reshape(B, T, heads, hd).transpose(1, 2)
        v = v.reshape(B, T, heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v          # [B,heads,T,hd]
        a = a.transpose(1, 2).reshape(B, T, width)
        x = x + F.linear(a, w[pre + 'attn.out_proj.weight'], w[pre + 'attn.out_proj.bias'])
        h = F.layer_norm(x, (width,), w[pre + 'ln_2.weight'], w[pre + 'ln_2.bias'], 1e-5)
        u = F.linear(h, w[pre + 'mlp.c_fc.weight'], w[pre + 'mlp.c_fc.bias'])
        g = u * torch.sigmoid(1.702 * u)                                # QuickGELU
        x = x + F.linear(g, w[pre + 'mlp.c_proj.weight'], w[pre + 'mlp.c_proj.bias'])
    x = F.layer_norm(x[:, 0, :], (width,), w['ln_post.weight'], w['ln_post.bias'], 1e-5)
    return x @ w['proj']


def to_hf_state_dict(w, cfg):
    """OpenAI visual key layout -> HF CLIPVisionModelWithProjection state dict."""
    width, layers = cfg['width'], cfg['layers']
    sd = OrderedDict()
    e = 'vision_model.embeddings.'
    sd[e + 'class_embedding'] = w['class_embedding']
    sd[e + 'patch_embedding.weight'] = w['conv1.weight']
    sd[e + 'position_embedding.weight'] = w['positional_embedding']
    sd['vision_model.pre_layrnorm.weight'] = w['ln_pre.weight']
    sd['vision_model.pre_layrnorm.bias'] = w['ln_pre.bias']
    for i in range(layers):
        s = 'transformer.resblocks.%d.' % i
        d = 'vision_model.encoder.layers.%d.' % i
        wq, wk, wv = w[s + 'attn.in_proj_weight'].split(width, dim=0)
        bq, bk, bv = w[s + 'attn.in_proj_bias'].split(width, dim=0)
        for n, ww, bb in (('q', wq, bq), ('k', wk, bk), ('v', wv, bv)):
            sd[d + 'self_attn.%s_proj.weight' % n] = ww
            sd[d + 'self_attn.%s_proj.bias' % n] = bb
        sd[d + 'self_attn.out_proj.weight'] = w[s + 'attn.out_proj.weight']
        sd[d + 'self_attn.out_proj.bias'] = w[s + 'attn.out_proj.bias']
        for a, b in (('ln_1', 'layer_norm1'), ('ln_2', 'layer_norm2'), ('mlp.c_fc', 'mlp.fc1'), ('mlp.c_proj', 'mlp.fc2')):
            sd[d + b + '.weight'] = w[s + a + '.weight']
            sd[d + b + '.bias'] = w[s + a + '.bias']
    sd['vision_model.post_layernorm.weight'] = w['ln_post.weight']
    sd['vision_model.post_layernorm.bias'] = w['ln_post.bias']
    sd['visual_projection.weight'] = w['proj'].T.contiguous()
    return sd


def hf_model(w, cfg):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    c = CLIPVisionConfig(hidden_size=cfg['width'], intermediate_size=4 * cfg['width'],
                         projection_dim=cfg['output_dim'], num_hidden_layers=cfg['layers'],
                         num_attention_heads=cfg['heads'], image_size=cfg['input_resolution'],
                         patch_size=cfg['patch_size'], hidden_act='quick_gelu', layer_norm_eps=1e-5)
    m = CLIPVisionModelWithProjection(c).float().eval()
    missing = m.load_state_dict(to_hf_state_dict(w, cfg), strict=False)
    bad = [k for k in missing.missing_keys if 'position_ids' not in k]
    assert not bad and not missing.unexpected_keys, (bad, missing.unexpected_keys)
    return m
