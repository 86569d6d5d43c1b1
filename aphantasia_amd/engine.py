"""The fused optimisation step: clip_fft.py's train(i) (clip_fft.py:235-295) as one chain of C-ABI
calls on a HIP stream -- no autograd graph, no per-step allocation, no host synchronisation.

    synth (irfft2 .. sigmoid) -> sampler (S cuts, patch-major f16) -> ViT forward -> similarity loss
      -> ViT input-gradient -> sampler adjoint -> synth adjoint (rfft2) -> [all-reduce] -> Adam

The autograd-based drop-in API (image.py / utils.py / clip.py) calls the same C entry points; this
class is what clip_fft.py and bench.py run when the loss is the standard prompt/image similarity sum.

Multi-GPU: every rank holds the full parameters, draws the SAME crop table,
processes a contiguous share of the S cuts, and the spectrum gradients are summed with ONE
all-reduce (RCCL through torch.distributed) per step; the loss term of each rank is already divided
by the global S, so the sum of the partial gradients is the single-GPU gradient.
"""
import math

import os

import numpy as np
import torch

from . import _ffi, ops
from .clip import LOSS_SCALE
from .image import colcorr_t, fft_scale
from .transforms import Transform, pack_aug
from .utils import draw_crop_params, draw_crop_params_bulk


def shard_range(S, rank, world):
    """Contiguous share of S cuts for `rank` (sizes differ by at most one)."""
    base, rem = divmod(S, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Engine:
    def __init__(self, params, h, w, model, samples, targets, sim='mix', colors=1.8, decay=1.5,
                 lr=0.05, optimizer='adam_custom', align='uniform', macro=0.4, transform=None,
                 size=None, rank=0, world=1, process_group=None, comm=None, param_kind='fft', decorrelate=True, lib=None, state=None, dwt=None, rng='bulk', use_graph=True,
                 rgb_priors=None, fixcontrast=False, sharp=0.0, expand=0.0, enforce=0.0, grad_f16=False, loss_scale=None, reduce_always=False, aest=None,
                 precise=False, graph_allreduce=None):
        """params: the leaf tensor ([1,3,h,w//2+1,2] spectrum for 'fft', [1,3,h,w] for 'pixel', the flat
        coefficient buffer for 'dwt' with dwt = its aphantasia_amd.dwt.DWTSynth);
        model: aphantasia_amd.clip.CLIPModel; targets: list of (embedding [1,D] tensor, coef) with
        coef = sign*weight as at clip_fft.py:257-267.
        rng: 'bulk' = vectorised host draws from a numpy Generator seeded off torch's global generator (same
        distributions, ~0.1 ms/step); 'reference' = the reference's exact per-cut draw order on torch's / numpy's
        global generators (a seeded run then reproduces the reference's crop tables; costs milliseconds of Python).
        precise: the opt-in split-precision ViT forward (aph_vit_forward_hilo: the cuts and every block's first LayerNorm output as hi + lo
        f16 pairs -- the two roundings that dominate the gradient error on weights with realistic dynamic range; ~8 % slower at C2).
        graph_allreduce: with world > 1 and a direct RCCL `comm`, capture the whole step INCLUDING its all-reduce into one hipGraph.  OPT-IN
        (default: the environment's APH_MULTIRANK_GRAPH=1, else off -> multi-rank steps launch eagerly): until that path has executed on a
        real multi-GPU node a hang inside the capture or its collectives would take a run down without an exception to catch;
        bench.py tries it as the first rung of a parent-supervised ladder, which CAN end a hang."""
        self.params = params
        self.dev = params.device
        self.h, self.w = h, w
        self.kind = param_kind
        self.model = model
        self.visual = model.visual
        self.size = size or self.visual.input_resolution
        self.patch = self.visual.patch_size
        self.S = int(samples)
        if self.S < 1:
            raise ValueError('Engine: samples = %d; at least one cut is needed (upstream: torch.cat of an empty list, utils.py:253)' % self.S)
        self.rank, self.world, self.pg = rank, world, process_group
        # comm: aphantasia_amd.comm.Comm -- RCCL called directly through the C ABI (aph_allreduce_f32) on the step's own stream.
        # Without one the reduction goes through torch.distributed's group (the gloo / CPU test path).
        self.comm = comm
        self._reduce = world > 1 or (bool(reduce_always) and comm is not None)    # reduce_always: run the collective on a 1-rank communicator too (tests)
        self.lo, self.hi = shard_range(self.S, rank, world)
        self.S_loc = self.hi - self.lo
        self.sim = sim
        if ops._sim_key(sim) == 'dot':
            # dot_compare (utils.py:270-274) is dot^2 / (eps + |v2|) over ALL cuts at once: not a mean of per-cut terms, so a
            # rank's shard cannot form it, and with --enforce / --expand the reference takes the magnitude from the OTHER
            # operand (out_enc2 / prev_enc) while the fused kernel takes it from the current encodings
            if world > 1:
                raise NotImplementedError("sim 'dot' needs sums over all cuts; it is single-GPU only in the fused engine")
            if float(enforce) != 0 or float(expand) > 0:
                raise NotImplementedError("sim 'dot' with --enforce / --expand is not supported in the fused engine "
                                          "(use the drop-in autograd API: aphantasia_amd.utils.sim_func composes it from torch ops)")
        self.rng_mode = rng
        # hipGraph replay only saves host time (on the 128-core GPU box eager launches measure the same step time at every
        # shard size).  Round 1 saw NaN gradients from replays next to eager work + device synchronizes; the cause turned out to
        # be the graph's MEMSET nodes (captured hipMemsetAsync), which this runtime mis-orders -- the step now zero-fills with
        # kernels (csrc/aph_device.h zero_fill_async; repro: tools/exp/frame_debug3.py on the commit before).
        # Multi-rank: with a direct RCCL communicator AND graph_allreduce the whole step INCLUDING the all-reduce and Adam is one graph too
        # (a shard's step is ~230 launches of 5-15 us: eager launches there are host-bound) -- behind a ONE-TIME self-check at capture
        # (_capture: the replay must reproduce an eager step bit for bit on every rank, else the engine stays eager and says so).  Through
        # torch.distributed (the gloo / CPU test path) the collective cannot be a graph node: eager.
        if graph_allreduce is None:
            graph_allreduce = os.environ.get('APH_MULTIRANK_GRAPH', '0') == '1'
        self.graph_allreduce = bool(graph_allreduce)
        if world > 1 and (comm is None or not self.graph_allreduce):
            use_graph = False
        self.use_graph, self._graphs, self._calls, self._vit_handle = use_graph, None, 0, None
        self.rgb_priors = (0.45, 0.17) if rgb_priors is True else rgb_priors       # illustrip.py:439-440 targets
        self.fixcontrast = bool(fixcontrast)
        self.sharp, self.expand = float(sharp), float(expand)       # clip_fft.py:269-270, :276-280
        # static loss scale of the fp16 backward with back-off: an overflowed step (NaN / inf gradient) is skipped on the
        # device (aph_adam_step_guarded); the host looks at the skip counter every GUARD_EVERY steps and halves the scale
        self.loss_scale = float(LOSS_SCALE if loss_scale is None else loss_scale)
        self._guard_seen, self._guard_host, self._guard_ev = 0, None, None
        self._guard_floor, self._floor_host, self._floor_ev = 0, None, None      # skip count at the last reset_params (device snapshot)
        self.enforce = float(enforce)                               # clip_fft.py:271-275
        # aest = (weight [D] or [1,D], bias, strength): `loss -= 0.001 * strength * (enc @ w + b).mean()` (clip_fft.py:255-256,
        # utils.py:402-413: the LAION linear aesthetic predictor on the raw encodings)
        self.aest = None
        if aest is not None and float(aest[2]) != 0:
            self.aest = (aest[0].detach().reshape(-1).float().to(self.dev).contiguous(), float(aest[1]), float(aest[2]))
        self.np_rng = np.random.default_rng(int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) if rng == 'bulk' else None
        self.align, self.macro, self.transform = align, macro, transform
        self.cc = colcorr_t(colors).flatten().tolist()
        self.decorrelate = decorrelate
        self.lib = lib if lib is not None else _ffi.lib()
        self.dwt = dwt
        if param_kind == 'dwt':
            h, w = dwt.H, dwt.W                  # the synthesised image may be one row/col larger than requested
            self.h, self.w = h, w
        self.plan = ops.SynthPlan(3, h, w, lib=self.lib)
        self.scale = fft_scale(h, w, decay).to(self.dev).contiguous() if param_kind == 'fft' else None
        self.lr = lr
        name = optimizer.lower()
        self.beta1 = 0.9 if name in ('adam', 'adamw') else 0.0
        self.wd = 0.01 if name.startswith('adamw') else 0.0
        self.decoupled = name.startswith('adamw')
        self.amsgrad = name == 'adamw_custom'
        n = params.numel()
        f32 = dict(dtype=torch.float32, device=self.dev)
        if state is None:        # optimiser state; `state=other.state()` shares it (the --dualmod engines, clip_fft.py:243-252)
            state = dict(m=torch.zeros(n, **f32) if self.beta1 else None, v=torch.zeros(n, **f32),
                         vmax=torch.zeros(n, **f32) if self.amsgrad else None, step=[0])
        self._state = state
        self.m, self.v, self.vmax = state['m'], state['v'], state['vmax']
        self.set_targets(targets)
        # static buffers
        self.visual.ensure_batch(max(self.S_loc, 1))
        if precise:
            self.visual.handle.enable_hilo()          # allocates the K-repeated weight copies: here, not inside a captured step
        g = self.size // self.patch
        self.P, self.Kp = g * g, 3 * self.patch * self.patch
        D = self.visual.output_dim
        Sl = max(self.S_loc, 1)
        self.raw = torch.empty(3, h, w, **f32)
        self.rgb = torch.empty(3, h, w, **f32)
        self.precise = bool(precise)
        self._patch_mode = _ffi.APH_OUT_PATCH_F16_HILO if self.precise else _ffi.APH_OUT_PATCH_F16
        self.patches = torch.empty(Sl * self.P, (2 if self.precise else 1) * self.Kp, dtype=torch.float16, device=self.dev)
        # ViT input-gradient handed to the sampler adjoint: f32 (default, exact path) or f16 carrying the loss scale
        # (measured: -1.6 % step time at C2, 5e-4 relative rounding per gradient element)
        self.grad_f16 = bool(grad_f16)
        self.gpatch = torch.empty(Sl * self.P, self.Kp, dtype=torch.float16 if self.grad_f16 else torch.float32, device=self.dev)
        self.enc = torch.empty(Sl, D, **f32)
        self.genc = torch.empty(Sl, D, **f32)
        self.grgb = torch.empty(3, h, w, **f32)
        self.grad = torch.empty_like(params)
        self.loss = torch.zeros(1, **f32)
        self.prior_ws = torch.empty(int(self.lib.cdll.aph_rgb_priors_ws_bytes()) // 8, device=self.dev, dtype=torch.float64)
        self.ws = torch.empty(Sl * (len(self.coef) + 2), **f32)
        self.guard = torch.zeros(2, dtype=torch.int32, device=self.dev)      # [skipped-step count, scratch]
        self._own_stream = None
        self._stage, self._stage_i = None, 0       # pinned host ring for the per-step H2D refreshes (built lazily, GPU only)
        self.geom = ops.make_geom(h, w, Sl, self.size, self.patch, align)
        self.geometric = isinstance(transform, Transform) and transform.geometric
        # Per-step host inputs (Adam scalars | crop table | augment table [| second set for --enforce]) live in ONE device
        # buffer, refreshed by one H2D copy per step; the named tensors are views of it (16-byte aligned).
        parts = [('hyper', (8,), torch.float32), ('table', (Sl, 3), torch.int32)]
        if self.geometric:
            parts.append(('aug', (Sl, _ffi.APH_AUG_STRIDE), torch.float32))
        if self.enforce != 0:
            parts.append(('table2', (Sl, 3), torch.int32))
            if self.geometric:
                parts.append(('aug2', (Sl, _ffi.APH_AUG_STRIDE), torch.float32))
        self.aug = self.aug2 = self.table2 = None
        self._stepin_layout, words = [], 0
        for name, shape, dt in parts:
            n = int(np.prod(shape))
            self._stepin_layout.append((name, words, n, shape, dt))
            words += (n + 3) // 4 * 4
        self._stepin = torch.zeros(words, dtype=torch.int32, device=self.dev)
        for name, v in self._stepin_views(self._stepin).items():
            setattr(self, name, v)
        self.tmp = ops.sample_ws(self.geom, self.geometric, self.dev, self.lib)      # engine-owned: tap tables + augmentation scratch
        if self.enforce != 0:          # second, independently drawn set of cuts of the same image
            self.enc2, self.genc2, self.grgb2 = torch.empty_like(self.enc), torch.empty_like(self.genc), torch.empty_like(self.grgb)
            self.loss2 = torch.zeros(1, **f32)
            self.ws2 = torch.empty(Sl * 3, **f32)
            self.enf_coef = torch.tensor([-self.enforce], **f32)

    def state(self):
        return self._state

    @property
    def step_count(self):
        return self._state['step'][0]

    def set_targets(self, targets):
        """targets: list of (embedding, coef); an embedding of shape [1,D] is compared with every cut, one of shape
        [S,D] pairs row s with cut s (the reference-image term, clip_fft.py:216,267).  Broadcast ones go first."""
        bro = [(t, c) for t, c in targets if t.reshape(-1, t.shape[-1]).shape[0] == 1]
        per = [(t, c) for t, c in targets if t.reshape(-1, t.shape[-1]).shape[0] != 1]
        for t, _ in per:
            if t.shape[0] != self.S:
                raise ValueError('per-cut target has %d rows, expected %d' % (t.shape[0], self.S))
        if self.expand > 0:          # slot of the previous step's encodings (clip_fft.py:276-280); inactive (coef 0) until one exists
            per = per + [(torch.ones(self.S, bro[0][0].shape[-1] if bro else per[0][0].shape[-1]), 0.0)]
        self._graphs = None          # captured graphs hold the old target / coefficient pointers
        self.n_broadcast = len(bro)
        self.targets = torch.cat([t.reshape(-1, t.shape[-1]).float().to(self.dev) for t, _ in bro + per], 0).contiguous()
        self.coef = [float(c) for _, c in bro + per]
        self.dcoef = torch.tensor(self.coef, dtype=torch.float32, device=self.dev)
        self.hcoef = _ffi.floats(self.coef)
        if getattr(self, 'ws', None) is not None and self.ws.numel() < max(self.S_loc, 1) * (len(self.coef) + 2):
            self.ws = torch.empty(max(self.S_loc, 1) * (len(self.coef) + 2), dtype=torch.float32, device=self.dev)

    def reset_params(self, new_params, keep_optimizer_state=False):
        """illustrip's per-frame re-parameterisation (illustrip.py:390,411-423) without re-creating anything: the new frame's
        parameters are copied INTO the existing leaf (same shape), the hipGraphs stay valid, and the Adam state is zeroed like a
        fresh torch.optim instance unless `keep_optimizer_state` (`--smooth`: optimizer.load_state_dict(opt_state))."""
        if tuple(new_params.shape) != tuple(self.params.shape) and new_params.numel() != self.params.numel():
            raise ValueError('reset_params: shape %s does not match %s' % (tuple(new_params.shape), tuple(self.params.shape)))
        with torch.no_grad():
            self.params.copy_(new_params.reshape(self.params.shape))
            if not keep_optimizer_state:
                for t in (self.m, self.v, self.vmax):
                    if t is not None:
                        t.zero_()
                self._state['step'][0] = 0
                # skipped-step bookkeeping belongs to the optimiser instance that just ended: a skip counted on the device BEFORE this
                # point must not be subtracted from the new frame's step counter, one counted after it must.  The boundary is a
                # snapshot of the device counter taken HERE in stream order (copied back asynchronously, read in _check_overflow),
                # not a window of calls: a genuine overflow of the new frame right after the reset is still accounted for.
                if self.params.is_cuda:
                    if self._floor_host is None:
                        self._floor_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                    elif self._floor_ev is not None:
                        self._floor_ev.synchronize()          # (the previous snapshot has landed before its buffer is reused)
                        self._guard_floor = max(self._guard_floor, int(self._floor_host[0]))
                    self._floor_host.copy_(self.guard[:1], non_blocking=True)
                    self._floor_ev = torch.cuda.Event()
                    self._floor_ev.record()
                else:
                    self._guard_floor = int(self.guard[0])

    def set_prev_enc(self, enc_rows=None, active=True):
        """--expand: make the encodings of the step just taken (this rank's rows; default: this engine's own) the per-cut
        target of the next step with coefficient +expand (clip_fft.py:276-280: `loss += a.expand * sim_func(out_enc, prev_enc)`)."""
        if not self.expand > 0:
            return
        if ops._sim_key(self.sim) == 'ang':
            raise NotImplementedError("--expand with the 'ang' similarity is not supported in the fused engine")
        enc_rows = self.enc if enc_rows is None else enc_rows
        D = self.targets.shape[1]
        row0 = self.n_broadcast + (len(self.coef) - 1 - self.n_broadcast) * self.S + self.lo
        self.targets[row0:row0 + self.S_loc].copy_(enc_rows[:self.S_loc].reshape(-1, D))
        want = self.expand if active else 0.0      # (`active=False`: keep the encodings, leave the term out -- illustrip's `if ii > 0`)
        if self.coef[-1] != want:
            self.coef[-1] = want
            self.dcoef[-1:].fill_(want)
            self.hcoef = _ffi.floats(self.coef)

    # ------------------------------------------------------------------
    def draw(self):
        """Host-side random draws for one step (the reference's own order, utils.py:222-251)."""
        if self.rng_mode == 'bulk':
            return draw_crop_params_bulk(self.S, self.size, self.h, self.w, self.align, self.macro, self.transform, self.np_rng)
        return draw_crop_params(self.S, self.size, self.h, self.w, self.align, self.macro, self.transform)

    def synthesize(self, contrast=1.0, shift=None):
        """image_f(shift, contrast) under no_grad (clip_fft.py:239 / :299) -> rgb [3,h,w] (engine-owned buffer)."""
        L, st = self.lib, ops._stream(self.params)
        if self.kind == 'fft':
            L.call('aph_synth_fft_fwd', self.plan.handle, ops.ptr(self.params), ops.ptr(self.scale), ops.ptr(shift), float(contrast),
                   _ffi.floats(self.cc), int(self.decorrelate), ops.ptr(self.raw), ops.ptr(self.rgb), st)
        else:
            self.spatial = self.dwt.forward(self.params) if self.kind == 'dwt' else self.params
            fixed_div = 3.3 if (self.fixcontrast and self.kind == 'pixel') else 0.0         # image.py:114-116
            L.call('aph_synth_spatial_fwd', self.plan.handle, ops.ptr(self.spatial), float(contrast), fixed_div, _ffi.floats(self.cc),
                   int(self.decorrelate), ops.ptr(self.rgb), st)
        return self.rgb

    def _enqueue_grad(self, shift):
        """forward + backward up to the parameter gradient: C-ABI calls only (capturable into a hipGraph)"""
        fixed_div = 3.3 if (self.fixcontrast and self.kind == 'pixel') else 0.0
        L, st = self.lib, ops._stream(self.params)
        Sl = self.S_loc
        self.synthesize(1.0, shift)
        cc = _ffi.floats(self.cc)
        vit_scale, smp_scale, gmode = self._grad_modes()
        if Sl > 0:
            L.call('aph_sample_fwd', ctypes_byref(self.geom), ops.ptr(self.rgb), ops.ptr(self.table), ops.ptr(self.aug), ops.ptr(self.tmp),
                   ops.ptr(self.patches), self._patch_mode, st)
            self.visual._forward_patches(self.patches, Sl, self.enc, hilo=self.precise)
            L.call('aph_sim_loss', ops.ptr(self.enc), Sl, self.enc.shape[1], ops.ptr(self.targets), ops.ptr(self.dcoef), self.hcoef,
                   len(self.coef), self.n_broadcast, self.S, self.lo, _ffi.SIM_TYPES[ops._sim_key(self.sim)], float(self.S), self.loss_scale, ops.ptr(self.ws),
                   ops.ptr(self.loss), ops.ptr(self.genc), st)
            if self.aest is not None:
                L.call('aph_linear_head', ops.ptr(self.enc), Sl, self.enc.shape[1], ops.ptr(self.aest[0]), self.aest[1], -0.001 * self.aest[2],
                       float(self.S), self.loss_scale, ops.ptr(self.loss), ops.ptr(self.genc), st)
            if self.enforce != 0:
                self._enqueue_enforce(L, st, Sl)
            self.visual.handle.backward(self.genc, Sl, self.gpatch, vit_scale)
            L.call('aph_sample_bwd', ctypes_byref(self.geom), ops.ptr(self.gpatch), smp_scale, ops.ptr(self.table), ops.ptr(self.aug),
                   ops.ptr(self.tmp), ops.ptr(self.grgb), gmode, st)
            if self.enforce != 0:
                L.call('aph_axpy_f32', ops.ptr(self.grgb), ops.ptr(self.grgb2), 1.0, self.grgb.numel(), st)
                L.call('aph_axpy_f32', ops.ptr(self.loss), ops.ptr(self.loss2), 1.0, 1, st)
        else:
            self.grgb.zero_()
            self.loss.zero_()
        if self.rgb_priors is not None and self.rank == 0:       # illustrip.py:438-440; replicated term -> one rank adds it
            L.call('aph_rgb_priors', ops.ptr(self.rgb), self.h, self.w, float(self.rgb_priors[0]), float(self.rgb_priors[1]), 1.0,
                   ops.ptr(self.prior_ws), ops.ptr(self.loss), ops.ptr(self.grgb), st)
        if self.sharp != 0 and self.kind != 'dwt' and self.rank == 0:      # `a.sharp != 0 and a.dwt is not True` (clip_fft.py:269)
            L.call('aph_rgb_sharp', ops.ptr(self.rgb), self.h, self.w, -self.sharp, ops.ptr(self.prior_ws), ops.ptr(self.loss), ops.ptr(self.grgb), st)
        if self.kind == 'fft':
            L.call('aph_synth_fft_bwd', self.plan.handle, ops.ptr(self.grgb), 1.0, ops.ptr(self.rgb), ops.ptr(self.raw), ops.ptr(self.scale),
                   1.0, cc, int(self.decorrelate), ops.ptr(self.grad), st)
        elif self.kind == 'dwt':
            L.call('aph_synth_spatial_bwd', self.plan.handle, ops.ptr(self.grgb), 1.0, ops.ptr(self.rgb), ops.ptr(self.spatial), 1.0, 0.0, cc,
                   int(self.decorrelate), ops.ptr(self.raw), st)            # d raw (reuses the raw buffer)
            self.dwt.backward(self.raw, self.grad)
        else:
            L.call('aph_synth_spatial_bwd', self.plan.handle, ops.ptr(self.grgb), 1.0, ops.ptr(self.rgb), ops.ptr(self.params), 1.0, fixed_div, cc,
                   int(self.decorrelate), ops.ptr(self.grad), st)

    def _grad_modes(self):
        """(out_scale of the ViT backward, gscale of the sampler adjoint, gradient layout): the f16 patch gradient keeps the
        loss scale (range) and the sampler adjoint removes it"""
        if self.grad_f16:
            return 1.0, 1.0 / self.loss_scale, _ffi.APH_GRAD_PATCH_F16
        return 1.0 / self.loss_scale, 1.0, _ffi.APH_OUT_PATCH_F16

    def _enqueue_enforce(self, L, st, Sl):
        """--enforce (clip_fft.py:271-275): `loss -= a.enforce * sim_func(out_enc, out_enc2)` with out_enc2 from a second,
        independently drawn slice_imgs of the same image.  Both encodings carry gradient.  The ViT handle keeps the
        activations of ONE forward, so: forward the second set, take its backward, then recompute the first set's forward
        (its backward follows in the caller): 3 forwards + 2 backwards instead of a second 2.5 GB arena."""
        code = _ffi.SIM_TYPES[ops._sim_key(self.sim)]
        D = self.enc.shape[1]
        hc = _ffi.floats([-self.enforce])

        def pair_term(enc, other, loss, genc):      # value + d/d enc of -enforce * sim(enc[s], other[s]) (mean over the GLOBAL S cuts)
            L.call('aph_sim_loss', ops.ptr(enc), Sl, D, ops.ptr(other), ops.ptr(self.enf_coef), hc, 1, 0, Sl, 0, code, float(self.S),
                   self.loss_scale, ops.ptr(self.ws2), ops.ptr(loss), ops.ptr(genc), st)
        L.call('aph_sample_fwd', ctypes_byref(self.geom), ops.ptr(self.rgb), ops.ptr(self.table2), ops.ptr(self.aug2), ops.ptr(self.tmp),
               ops.ptr(self.patches), self._patch_mode, st)
        self.visual._forward_patches(self.patches, Sl, self.enc2, hilo=self.precise)
        pair_term(self.enc2, self.enc, self.loss2, self.genc2)
        vit_scale, smp_scale, gmode = self._grad_modes()
        self.visual.handle.backward(self.genc2, Sl, self.gpatch, vit_scale)
        L.call('aph_sample_bwd', ctypes_byref(self.geom), ops.ptr(self.gpatch), smp_scale, ops.ptr(self.table2), ops.ptr(self.aug2),
               ops.ptr(self.tmp), ops.ptr(self.grgb2), gmode, st)
        pair_term(self.enc, self.enc2, self.loss2, self.genc2)       # same value again; genc2 now = d/d enc (first set)
        L.call('aph_axpy_f32', ops.ptr(self.genc), ops.ptr(self.genc2), 1.0, self.genc.numel(), st)
        L.call('aph_sample_fwd', ctypes_byref(self.geom), ops.ptr(self.rgb), ops.ptr(self.table), ops.ptr(self.aug), ops.ptr(self.tmp),
               ops.ptr(self.patches), self._patch_mode, st)
        self.visual._forward_patches(self.patches, Sl, self.enc, hilo=self.precise)

    def _stepin_views(self, flat):
        return {name: flat[o:o + n].view(dt).view(shape) for name, o, n, shape, dt in self._stepin_layout}

    def _upload(self, hy, table, augs, table2, augs2):
        """Refresh the device-side per-step inputs (Adam scalars, crop table, augment table).  On the GPU the values go
        through a ring of PINNED host buffers owned by the engine: the async copies never read from a temporary, and the
        host may run ahead of the device by up to the ring depth."""
        Sl = self.S_loc
        rows = lambda a: torch.from_numpy(np.ascontiguousarray(a[self.lo:self.hi]))
        packed = lambda a: rows(a) if isinstance(a, np.ndarray) else pack_aug(a[self.lo:self.hi])
        items = [('hyper', torch.tensor(hy, dtype=torch.float32))]
        if Sl > 0:
            items.append(('table', rows(table)))
            if self.geometric:
                items.append(('aug', packed(augs)))
            if self.enforce != 0:
                items.append(('table2', rows(table2)))
                if self.geometric:
                    items.append(('aug2', packed(augs2)))
        if not self.params.is_cuda:
            for name, src in items:
                getattr(self, name).copy_(src)
            return
        if self._stage is None:
            self._stage = []
            for _ in range(4):
                flat = torch.zeros(self._stepin.shape, dtype=torch.int32).pin_memory()
                self._stage.append(dict(flat=flat, views=self._stepin_views(flat), ev=None))
        slot = self._stage[self._stage_i % len(self._stage)]
        self._stage_i += 1
        if slot['ev'] is not None:
            slot['ev'].synchronize()            # the copy issued from this slot four steps ago has been consumed
        for name, src in items:
            slot['views'][name].copy_(src.reshape(slot['views'][name].shape))
        self._stepin.copy_(slot['flat'], non_blocking=True)
        slot['ev'] = torch.cuda.Event()
        slot['ev'].record()

    GUARD_EVERY = 16

    def _enqueue_adam(self):
        self.lib.call('aph_adam_step_guarded', ops.ptr(self.params), ops.ptr(self.grad), ops.ptr(self.m), ops.ptr(self.v), ops.ptr(self.vmax),
                      ops.ptr(self.hyper), int(self.decoupled), self.params.numel(), ops.ptr(self.guard), ops._stream(self.params))

    def _check_overflow(self):
        """Every GUARD_EVERY steps: look at the skipped-step counter copied back GUARD_EVERY steps ago (no stall), then
        start the next asynchronous copy.  A moved counter halves the loss scale (graphs are re-captured with it)."""
        if self._calls % self.GUARD_EVERY:
            return
        if not self.params.is_cuda:
            count = int(self.guard[0])
        else:
            count = self._guard_seen
            if self._guard_ev is not None:
                self._guard_ev.synchronize()
                count = int(self._guard_host[0])
            if self._guard_host is None:
                self._guard_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._guard_host.copy_(self.guard[:1], non_blocking=True)
            self._guard_ev = torch.cuda.Event()
            self._guard_ev.record()
        if count > self._guard_seen:
            # a skipped step is no optimiser step: torch.optim's state['step'] would not have advanced either (not carried across a
            # reset_params: those skips belonged to the previous frame's optimiser)
            if self._floor_ev is not None:
                self._floor_ev.synchronize()
                self._guard_floor, self._floor_ev = max(self._guard_floor, int(self._floor_host[0])), None
            mine = count - max(self._guard_seen, self._guard_floor)          # skips of the CURRENT optimiser instance among the new ones
            if mine > 0:
                self._state['step'][0] = max(self._state['step'][0] - mine, 0)
            self._guard_seen = count
            self.loss_scale = max(self.loss_scale * 0.5, 1.0)
            self._graphs = None
            print(' fp16 overflow in the backward pass: %d step(s) skipped so far, loss scale -> %g' % (count, self.loss_scale), flush=True)

    @staticmethod
    def _same_bits(a, b):
        return torch.equal(a, b)

    def _capture(self):
        """Record the step's ~280 launches into hipGraphs (replayed per step: the host then costs three tiny H2D copies
        and one or two graph launches).  Everything the kernels read that changes per step -- crop table, augment table,
        Adam scalars -- lives in fixed device buffers refreshed before the replay."""
        torch.cuda.synchronize()
        self._vit_handle = self.visual.handle
        g1 = torch.cuda.CUDAGraph()
        with_comm = self._reduce and self.comm is not None
        captured = True
        try:
            # with a collective in the graph, only THIS thread's calls are policed during capture: a collective library's helper threads
            # may call the runtime meanwhile (the default 'global' mode would fail the capture, or their call, for that)
            with torch.cuda.graph(g1, **({'capture_error_mode': 'thread_local'} if with_comm else {})):
                self._enqueue_grad(None)
                if self._reduce:
                    self._all_reduce()         # (only with a direct RCCL comm: the collective is a node of the step's graph)
                self._enqueue_adam()
        except Exception as e:                 # e.g. a collective library that refuses stream capture: stay eager, loudly
            if not with_comm:
                raise
            print(' rank %d: capturing the step with its all-reduce into a hipGraph failed (%s): steps stay eager' % (self.rank, e), flush=True)
            captured = False
        if with_comm and not getattr(self, '_graph_checked', False):
            # the ranks agree FIRST on whether every one of them holds a graph (a rank that left here alone would leave the others waiting in
            # the check's collectives): all or none
            try:
                flag = torch.tensor([1.0 if captured else 0.0], device=self.dev)
                self.comm.all_reduce_(flag, ops._stream(flag))
                torch.cuda.synchronize()
                everyone = int(round(float(flag))) == self.world
            except Exception as e:
                print(' rank %d: agreeing on the captured step failed (%s): steps stay eager' % (self.rank, e), flush=True)
                everyone = False
            if not everyone:
                if captured:
                    print(' rank %d: another rank could not capture its step: steps stay eager on every rank' % self.rank, flush=True)
                self._graph_checked = True
                self.use_graph, self._graphs = False, None
                return
        elif not captured:
            self.use_graph, self._graphs = False, None
            return
        if self._reduce and self.comm is not None and not getattr(self, '_graph_checked', False):
            # one-time self-check: ONE eager step and ONE replay from the same state and the same (already uploaded) step inputs must
            # leave the same bits in the parameters, on every rank (a one-rank communicator runs it too: that is how a one-GPU box
            # covers this code, tests/test_gpu_comm.py)
            state = [t for t in (self.params, self.m, self.v, self.vmax, self.guard) if t is not None]
            snap = [t.detach().clone() for t in state]

            def restore():
                with torch.no_grad():
                    for t, c in zip(state, snap):
                        t.copy_(c)
            try:
                self._enqueue_grad(None)
                self._all_reduce()
                self._enqueue_adam()
                ref = self.params.detach().clone()
                restore()
                g1.replay()
                same = self._same_bits(ref, self.params.detach())
                restore()
                flag = torch.tensor([1.0 if same else 0.0], device=self.dev)
                self.comm.all_reduce_(flag, ops._stream(flag))
                torch.cuda.synchronize()
            except Exception as e:             # never let the check itself take a run down: eager is always correct
                print(' rank %d: self-check of the captured multi-rank step failed to run (%s): steps stay eager' % (self.rank, e), flush=True)
                restore()
                self._graph_checked = True
                self.use_graph, self._graphs = False, None
                return
            self._graph_checked = True
            if int(round(float(flag))) != self.world:
                print(' rank %d: the captured step with its all-reduce did not reproduce an eager step bit for bit (this rank: %s): steps stay eager'
                      % (self.rank, 'same' if same else 'DIFFERENT'), flush=True)
                self.use_graph, self._graphs = False, None
                return
        self._graphs = (g1, None)

    def step(self, table=None, augs=None, lr=None, shift=None, tables2=None):
        """One train(i).  Returns the (device) loss tensor of THIS step -- do not .item() it every step."""
        if self.world > 1 and self.params.is_cuda:
            # Multi-rank: the step (and with it the RCCL all-reduce) runs on a stream of its own rather than on the legacy
            # default (NULL) stream, whose implicit synchronisation with every blocking stream is a bad neighbour for a
            # collective library's internal streams.  The caller's stream is fenced on entry and exit.
            if self._own_stream is None:
                self._own_stream = torch.cuda.Stream(device=self.dev)
            cur = torch.cuda.current_stream(self.dev)
            self._own_stream.wait_stream(cur)
            with torch.cuda.stream(self._own_stream):
                out = self._step(table, augs, lr, shift, tables2)
            cur.wait_stream(self._own_stream)
            return out
        return self._step(table, augs, lr, shift, tables2)

    def _step(self, table, augs, lr, shift, tables2):
        if table is None:
            table, augs = self.draw()
        table2 = augs2 = None
        if self.enforce != 0:
            table2, augs2 = self.draw() if tables2 is None else tables2          # the second slice_imgs of the same train(i)
        Sl = self.S_loc
        self._state['step'][0] += 1
        self._calls += 1
        lr = self.lr if lr is None else lr
        hy = ops.adam_hyper(self._state['step'][0], lr, self.beta1, 0.999, 1e-8, self.wd, 1.0)
        self._upload(hy, table, augs, table2, augs2)
        self._check_overflow()
        use_graph = self.use_graph and shift is None and self.params.is_cuda
        if self._graphs is not None and self._vit_handle is not self.visual.handle:
            self._graphs = None          # the ViT handle (and its activation arena) was re-created: the captured pointers are dead
        if use_graph and self._graphs is None and self._calls > 2:      # two eager steps first (one-time kernel attributes, allocator warm-up)
            self._capture()
        if use_graph and self._graphs is not None:
            self._graphs[0].replay()
            self.visual._generation += 1
            return self.loss
        self._enqueue_grad(shift)
        if self._reduce:
            self._all_reduce()
        self._enqueue_adam()
        return self.loss

    def _all_reduce(self):
        """the one collective of the step: sum of the partial parameter gradients over the ranks"""
        if self.comm is not None:
            self.comm.all_reduce_(self.grad, ops._stream(self.grad))
        else:
            import torch.distributed as dist
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.pg)

    def global_loss(self):
        """The step's loss summed over ranks (host float; synchronises)."""
        if self.world > 1:
            t = self.loss.clone()
            if self.sim and 'ang' in str(self.sim) and self.rank != 0:
                t -= sum(self.coef)          # the constant of 'ang' is counted once
            if self.comm is not None:
                # same stream as the step's collective: operations of one communicator must not overlap
                st = self._own_stream if self._own_stream is not None else torch.cuda.current_stream(self.dev)
                st.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(st):
                    self.comm.all_reduce_(t)
                torch.cuda.current_stream(self.dev).wait_stream(st)
            else:
                import torch.distributed as dist
                dist.all_reduce(t, group=self.pg)
            return float(t)
        return float(self.loss)


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)
