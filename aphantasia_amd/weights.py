"""CLIP visual-tower weights: OpenAI checkpoint layout, loader and seeded synthetic init.

The reference gets its weights from `clip.load(name)` (clip_fft.py:119), which
downloads OpenAI's TorchScript archives.  There is no network on the build/bench
machines, so benchmarks and parity tests use seeded synthetic weights with the
same shapes and the same key layout ("Weights").
"""
import math

import torch

VIT_CONFIGS = {
    'ViT-B/32': dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512),
    'ViT-B/16': dict(input_resolution=224, patch_size=16, width=768, layers=12, heads=12, output_dim=512),
}


def visual_config(name):
    if name not in VIT_CONFIGS:
        raise ValueError("only the ViT CLIP models are supported by the HIP path (%s); got %s"
                         % (', '.join(VIT_CONFIGS), name))
    return dict(VIT_CONFIGS[name])


def synthetic_visual_weights(cfg, seed=1):
    """Random weights following openai/CLIP's own constructor initialisation
    (class/positional embeddings and proj ~ width**-0.5 * randn; conv/linear PyTorch
    defaults; MHA in_proj xavier-uniform, zero biases).  fp32, OpenAI key layout
    without the `visual.` prefix."""
    g = torch.Generator().manual_seed(seed)
    width, layers, p = cfg['width'], cfg['layers'], cfg['patch_size']
    T = (cfg['input_resolution'] // p) ** 2 + 1
    scale = width ** -0.5

    def randn(*s):
        return torch.randn(*s, generator=g)

    def uniform(shape, bound):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    w = {}
    fan_in = 3 * p * p
    w['conv1.weight'] = uniform((width, 3, p, p), 1.0 / math.sqrt(fan_in))
    w['class_embedding'] = scale * randn(width)
    w['positional_embedding'] = scale * randn(T, width)
    w['ln_pre.weight'] = torch.ones(width)
    w['ln_pre.bias'] = torch.zeros(width)
    for i in range(layers):
        pre = 'transformer.resblocks.%d.' % i
        w[pre + 'attn.in_proj_weight'] = uniform((3 * width, width), math.sqrt(6.0 / (4 * width)))
        w[pre + 'attn.in_proj_bias'] = torch.zeros(3 * width)
        w[pre + 'attn.out_proj.weight'] = uniform((width, width), 1.0 / math.sqrt(width))
        w[pre + 'attn.out_proj.bias'] = torch.zeros(width)
        w[pre + 'ln_1.weight'] = torch.ones(width)
        w[pre + 'ln_1.bias'] = torch.zeros(width)
        w[pre + 'mlp.c_fc.weight'] = uniform((4 * width, width), 1.0 / math.sqrt(width))
        w[pre + 'mlp.c_fc.bias'] = uniform((4 * width,), 1.0 / math.sqrt(width))
        w[pre + 'mlp.c_proj.weight'] = uniform((width, 4 * width), 1.0 / math.sqrt(4 * width))
        w[pre + 'mlp.c_proj.bias'] = uniform((width,), 1.0 / math.sqrt(4 * width))
        w[pre + 'ln_2.weight'] = torch.ones(width)
        w[pre + 'ln_2.bias'] = torch.zeros(width)
    w['ln_post.weight'] = torch.ones(width)
    w['ln_post.bias'] = torch.zeros(width)
    w['proj'] = scale * randn(width, cfg['output_dim'])
    return f16_representable(w)


def f16_representable(w):
    """OpenAI's archives store every tensor in fp16 ("Weights"); `clip.load` on a CPU up-casts those VALUES to fp32
    (`model.float()`), the GPU path runs them as they are.  A synthetic stand-in is therefore rounded to f16-representable values too:
    the fp32 CPU oracle and the f16 MFMA operands of the HIP path then see the SAME weights, as they would with a real checkpoint
    (round 4: before, the f16 rounding of random fp32 weights was an error source of its own in every parity number -- the largest
    single one in profiles/r04_precision_attribution.txt)."""
    return {k: v.half().float() for k, v in w.items()}


def load_openai_checkpoint(path):
    """Reads an OpenAI CLIP archive (`ViT-B-32.pt`, TorchScript) or a plain state-dict
    file and returns (visual weights fp32 in the layout above, cfg, full state dict)."""
    try:
        sd = torch.jit.load(path, map_location='cpu').state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location='cpu')
        if hasattr(sd, 'state_dict'):
            sd = sd.state_dict()
    if 'visual.proj' not in sd or 'visual.class_embedding' not in sd:
        raise ValueError('%s is not a ViT CLIP checkpoint (no visual.proj)' % path)
    vis = {k[len('visual.'):]: v.float() for k, v in sd.items() if k.startswith('visual.')}
    width = vis['conv1.weight'].shape[0]
    p = vis['conv1.weight'].shape[-1]
    layers = len({k.split('.')[2] for k in vis if k.startswith('transformer.resblocks.')})
    grid = round((vis['positional_embedding'].shape[0] - 1) ** 0.5)
    cfg = dict(input_resolution=grid * p, patch_size=p, width=width, layers=layers,
               heads=width // 64, output_dim=vis['proj'].shape[1])
    return vis, cfg, sd


def stress_visual_weights(cfg, seed=1):
    """Synthetic weights with the activation statistics real CLIP towers are known for, which the unit-gain
    initialisation above never produces: LayerNorm gains spread over [0.2, 10] with non-zero biases (the linear
    that follows is scaled down column-wise, so the block's output keeps order-1 magnitude while the fp16
    intermediate spans a wide range), a few residual-stream channels carrying 50-100x the typical magnitude
    ("massive activations": written by the class/positional embeddings and by every c_proj bias), and non-zero
    attention biases.  Same key layout as synthetic_visual_weights.  Used by the stress parity tests and by
    anyone without a checkpoint who wants realistic dynamic range; APH_CLIP_CHECKPOINT (a real OpenAI archive)
    takes precedence in those tests when it is set."""
    w = synthetic_visual_weights(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    width, layers = cfg['width'], cfg['layers']
    outliers = torch.randperm(width, generator=g)[:3]

    def gains():
        gn = torch.exp(torch.empty(width).uniform_(math.log(0.2), math.log(10.0), generator=g))
        gn[outliers] = torch.empty(3).uniform_(0.05, 0.2, generator=g)      # real towers damp the massive channels in the LN gain
        return gn

    w['class_embedding'][outliers] += torch.tensor([60.0, -80.0, 100.0]) * width ** -0.5 * 8
    w['positional_embedding'][:, outliers] += (torch.randn(w['positional_embedding'].shape[0], 3, generator=g) * 0.5 + 2.0) * torch.tensor([1.0, -1.0, 1.0])
    for name in ('ln_pre',):
        w[name + '.weight'] = torch.exp(torch.empty(width).uniform_(math.log(0.5), math.log(2.0), generator=g))
        w[name + '.bias'] = 0.1 * torch.randn(width, generator=g)
    for i in range(layers):
        pre = 'transformer.resblocks.%d.' % i
        for ln, lin in (('ln_1', 'attn.in_proj_weight'), ('ln_2', 'mlp.c_fc.weight')):
            gn = gains()
            w[pre + ln + '.weight'] = gn
            w[pre + ln + '.bias'] = 0.3 * torch.randn(width, generator=g) * gn
            w[pre + lin] = w[pre + lin] / gn[None, :]                       # wide fp16 range in h, order-1 products
        w[pre + 'attn.in_proj_weight'][:2 * width] *= 2.5                    # peaky softmax (logits up to ~15) instead of near-uniform attention
        w[pre + 'attn.in_proj_bias'] = 0.2 * torch.randn(3 * width, generator=g)
        w[pre + 'attn.out_proj.bias'] = 0.05 * torch.randn(width, generator=g)
        # massive residual channels: every MLP adds a large constant to three channels
        w[pre + 'mlp.c_proj.bias'][outliers] += torch.tensor([6.0, -8.0, 10.0]) * (1.0 if i < layers // 2 else 0.2)
    gn = torch.exp(torch.empty(width).uniform_(math.log(0.5), math.log(4.0), generator=g))
    w['ln_post.weight'] = gn
    w['ln_post.bias'] = 0.1 * torch.randn(width, generator=g)
    w['proj'] = w['proj'] / gn[:, None] * 2.0
    return f16_representable(w)
