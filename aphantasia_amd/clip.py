"""CLIP image tower for the optimisation loop: the piece of `clip.load(...)` / `model.encode_image`
that clip_fft.py:119,254 uses, running on the HIP ViT (csrc/vit.hip).

    model, _ = load('ViT-B/32', weights='/path/ViT-B-32.pt')     # OpenAI checkpoint if you have one
    model, _ = load('ViT-B/32')                                  # seeded synthetic weights (benchmarks/tests)
    enc = model.encode_image(cuts)                               # [S,3,224,224] normalised -> [S,512], autograd-aware

The text tower / tokenizer (encode_text, clip.tokenize) run once per prompt, off the hot path; they
are taken from the `clip` package when it is importable together with a real checkpoint, otherwise
text prompts are mapped to seeded synthetic target embeddings (loud warning).
"""
import hashlib
import warnings

import torch

from . import ops
from .weights import load_openai_checkpoint, synthetic_visual_weights, visual_config

LOSS_SCALE = 4096.0     # static scale of the fp16 backward chain (precision table)


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, visual):
        S = x.shape[0]
        patches = ops.patchify(x.contiguous().float(), visual.patch_size)
        enc = visual._forward_patches(patches, S)
        ctx.visual, ctx.S = visual, S
        ctx.gen = visual._generation
        ctx.save_for_backward(patches)
        ctx.shape = x.shape
        return enc

    @staticmethod
    def backward(ctx, g):
        patches, = ctx.saved_tensors
        v = ctx.visual
        if v._generation != ctx.gen:          # another forward ran since: rebuild this node's activations
            v._forward_patches(patches, ctx.S)
            ctx.gen = v._generation
        gp = v.handle.backward((g.float() * LOSS_SCALE).contiguous(), ctx.S, out_scale=1.0 / LOSS_SCALE)
        return ops.unpatchify(gp, ctx.S, ctx.shape[2], v.patch_size), None


class VisualTransformer:
    """Stands in for clip.model.VisionTransformer: `.input_resolution`, `.patch_size`, callable."""

    def __init__(self, cfg, weights, max_batch=256, lib=None):
        self.lib = lib
        self.cfg = dict(cfg)
        self.input_resolution = cfg['input_resolution']
        self.patch_size = cfg['patch_size']
        self.output_dim = cfg['output_dim']
        self.weights = weights
        self.handle = ops.VitHandle(cfg, weights, max_batch, lib=lib)
        self._generation = 0

    def ensure_batch(self, S):
        if S > self.handle.max_batch:
            self.handle = ops.VitHandle(self.cfg, self.weights, S, lib=self.lib)

    def _forward_patches(self, patches, S, out=None, hilo=False):
        self.ensure_batch(S)
        self._generation += 1
        return self.handle.forward(patches, S, out, hilo=hilo)

    def __call__(self, x):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_resolution or x.shape[3] != self.input_resolution:
            raise ValueError('encode_image expects [S,3,%d,%d], got %s' % (self.input_resolution, self.input_resolution, tuple(x.shape)))
        return _Encode.apply(x, self)


class CLIPModel:
    def __init__(self, name, cfg, weights, full_state=None, max_batch=256, lib=None):
        self.name = name
        self.visual = VisualTransformer(cfg, weights, max_batch, lib=lib)
        self._full_state = full_state
        self.synthetic = full_state is None

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        raise NotImplementedError(
            'encode_text needs the OpenAI text tower + BPE vocabulary (the `clip` package and a real checkpoint); '
            'use aphantasia_amd.clip.text_embedding(model, prompt), which falls back to a seeded synthetic embedding')

    def float(self):
        return self

    def eval(self):
        return self

    def cuda(self):
        return self


def load(name='ViT-B/32', device='cuda', jit=False, weights=None, seed=1, max_batch=256):
    """-> (model, None).  weights: path to an OpenAI CLIP checkpoint; None = seeded synthetic weights."""
    if weights is not None:
        vis, cfg, full = load_openai_checkpoint(weights)
        want = visual_config(name)
        if (cfg['patch_size'], cfg['width'], cfg['layers']) != (want['patch_size'], want['width'], want['layers']):
            raise ValueError('%s does not hold a %s visual tower' % (weights, name))
        return CLIPModel(name, cfg, vis, full, max_batch), None
    cfg = visual_config(name)
    warnings.warn('aphantasia_amd.clip.load(%r): no checkpoint given -> seeded SYNTHETIC weights (seed %d); '
                  'images will not be meaningful, use --clip-weights for real runs' % (name, seed))
    return CLIPModel(name, cfg, synthetic_visual_weights(cfg, seed), None, max_batch), None


def text_embedding(model, text, device='cuda'):
    """Target embedding for a prompt ([1, output_dim], detached).  Real text tower when available,
    else a deterministic synthetic vector derived from the prompt (so runs are reproducible)."""
    if not model.synthetic:
        # a real checkpoint: the target must come from its text tower -- a random stand-in would silently optimise towards
        # nothing while the output still looks like a valid run
        try:
            import clip as openai_clip          # the reference's dependency (openai/CLIP); tokenizer + text tower
        except ImportError as e:
            raise RuntimeError('text prompts with a real CLIP checkpoint need the `clip` package (openai/CLIP: BPE vocabulary + text '
                               'tower); it is not importable here.  Pass image prompts (-i) or install it.') from e
        if getattr(model, '_text_model', None) is None:
            import copy
            model._text_model = openai_clip.model.build_model(copy.copy(model._full_state)).float().to(device).eval()
        with torch.no_grad():
            return model._text_model.encode_text(openai_clip.tokenize(text).to(device)).detach().clone().float()
    h = int.from_bytes(hashlib.sha256(text.encode()).digest()[:4], 'little')
    g = torch.Generator().manual_seed(h)
    return torch.randn(1, model.visual.output_dim, generator=g).to(device)
