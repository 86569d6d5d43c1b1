"""illustrip's continuous mode on the fused engine: the per-frame loop of illustrip.py `process()` (illustrip.py:367-470).

    per frame:  MOTION   -- warp the current picture by the frame's (scale, shift, angle, shear) and re-create the parameters
                            from it (illustrip.py:381-409): RGB: frame_transform(pixels); FFT: irfftn -> frame_transform -> rfftn
                OPTIMISER -- a fresh torch.optim instance per frame (illustrip.py:411-423; `--smooth` reloads the previous state)
                STEPS    -- opt_step optimisation steps (illustrip.py:426-470), `-dm` alternates the two CLIP models
                SAVE     -- image_f(contrast) (illustrip.py:478-480)

Everything device-side runs through the C ABI: aph_frame_affine, aph_irfft2 / aph_rfft2, and the engine's fused step.
Nothing is allocated per frame: Engine.reset_params copies the warped picture into the existing leaf and zeroes the Adam
moments in place, so captured hipGraphs stay valid across frames.

Depth (illustrip.py:387-388,396-397,404-405): with `depth` > 0 the picture goes through depth_transform (depthwarp.py: blur,
bicubic resize, the caller's depth ESTIMATOR, aph_grid_warp) before the affine.  The estimator (Depth-Anything-V2 upstream)
is the `depth_fn` callable -- its weights are not part of this repository.

Not here (SURVEY.md section 2 out of scope): the text-file / multi-line prompt scheduling and `latent_anima` motion curves of the illustrip.py command line (pass per-frame motion
values to `frame()` yourself), LPIPS.
"""
import torch

from . import ops, transforms


class FrameLoop:
    def __init__(self, engine, gen='RGB', opt_step=1, smooth=False, engine2=None, dualmod=None, depth=0.0, depth_fn=None, colors=1.0,
                 depth_dir=None, depth_res=518):
        """engine: aphantasia_amd.engine.Engine with param_kind 'pixel' (gen RGB: rgb_priors=True as illustrip.py:438-440) or
        'fft' (gen FFT); engine2 + dualmod: the ViT-B/16 engine sharing its parameters / optimiser state (illustrip.py:372)."""
        self.eng, self.eng2, self.dualmod = engine, engine2, dualmod
        self.gen = gen.upper()
        if self.gen not in ('RGB', 'FFT'):
            raise ValueError("gen must be 'RGB' or 'FFT'")
        if (self.gen == 'RGB') != (engine.kind == 'pixel'):
            raise ValueError('gen %s needs an engine with param_kind %s' % (self.gen, 'pixel' if self.gen == 'RGB' else 'fft'))
        self.opt_step = int(opt_step)
        self.smooth = bool(smooth) and self.gen == 'FFT'           # illustrip.py:97-100: RGB forces smooth off
        self.h, self.w = engine.h, engine.w
        self.frames = 0
        self.depth, self.depth_fn, self.colors, self.depth_dir, self.depth_res = float(depth), depth_fn, colors, depth_dir, depth_res
        if self.depth > 0 and depth_fn is None:
            raise ValueError('depth > 0 needs depth_fn: a callable [1,3,h,w] in (0,1) -> depth [1,1,h,w] (depthwarp.InferDepthAny upstream)')
        self._img = torch.empty(3, self.h, self.w, dtype=torch.float32, device=engine.dev) if self.gen == 'FFT' else None
        self._spec = torch.empty_like(engine.params) if self.gen == 'FFT' else None

    def reparameterise(self, scale, shift, angle, shear):
        """MOTION (illustrip.py:381-409) + the optimiser restart (:411-423)"""
        e = self.eng

        def deep(x):                                                                                         # illustrip.py:387-388 / :404-405
            if self.depth <= 0:
                return x
            from . import depthwarp
            return depthwarp.depth_transform(x, self.depth_fn, self.depth, scale, shift, self.colors, self.depth_dir, self.frames,
                                             res=self.depth_res, lib=e.lib)
        if self.gen == 'RGB':
            new = transforms.frame_transform(deep(e.params.detach().reshape(1, 3, self.h, self.w)), (self.h, self.w), angle, shift, scale, shear,
                                             lib=e.lib)
        else:
            ops.irfft2(e.plan, e.params.detach(), out=self._img, lib=e.lib)                                  # illustrip.py:401-403
            img = transforms.frame_transform(deep(self._img.reshape(1, 3, self.h, self.w)), (self.h, self.w), angle, shift, scale, shear, lib=e.lib)
            new = ops.rfft2(e.plan, img.reshape(3, self.h, self.w).contiguous(), out=self._spec, lib=e.lib)    # :407-408
        e.reset_params(new, keep_optimizer_state=self.smooth and self.frames > 0)

    def frame(self, scale=1.012, shift=(0, 10.0), angle=0.8, shear=0.4, contrast=None, lr=None, noise=0.0, consume_noise_draw=False):
        """One frame.  Defaults = illustrip's non-animated motion (`1 + a.scale`, `[0, a.shift]`, a.angle, a.shear with the CLI
        defaults, illustrip.py:70-73,381-384).  Returns the frame to save (device [3,h,w] in (0,1)) when `contrast` is given."""
        self.reparameterise(scale, shift, angle, shear)
        for ss in range(self.opt_step):
            ii = self.frames                     # `ii in dualmod_nums` (illustrip.py:372): every dualmod-th frame of the line
            e = self.eng2 if (self.eng2 is not None and self.dualmod and ii >= self.dualmod and ii % self.dualmod == 0) else self.eng
            sh = None
            if consume_noise_draw:
                # illustrip.py:429 draws torch.rand(1,1,H,W//2+1,1) on every step whenever a.noise > 0, also for --gen RGB where
                # pixel_image ignores it: a seeded `--rng reference` run must consume it to stay on the reference's crop / augment stream
                torch.rand(1, 1, self.h, self.w // 2 + 1, 1)
            if noise > 0 and self.gen == 'FFT':                                                             # illustrip.py:429
                sh = (noise * (torch.rand(self.h, self.w // 2 + 1) - 0.5)).to(e.dev).contiguous()
            e.step(lr=lr, shift=sh)
            if self.eng.expand > 0:
                # illustrip.py:459-463: `prev_enc = out_enc.detach()` after EVERY step; the term `a.expand * sim_func(prev_enc, out_enc)`
                # is part of the loss from the line's second frame on (`if ii > 0`), whichever CLIP model produced prev_enc
                nxt = ii if ss + 1 < self.opt_step else ii + 1          # frame index of the step that will read it
                for other in (self.eng, self.eng2):
                    if other is not None:
                        other.set_prev_enc(e.enc, active=nxt > 0)
        self.frames += 1
        if contrast is not None:
            return self.eng.synthesize(contrast)                                                             # illustrip.py:478
        return None
