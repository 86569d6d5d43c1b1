#!/usr/bin/env python
"""bench.py -- optimisation steps/sec of the CLIP-guided hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c1|c3|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json configs[1], "C2"): 1280x720 FFT parameteriser, ViT-B/32, `--samples 200` with the CLI
defaults (-tf fast => 190 effective cuts, clip_fft.py:167-169), sim 'mix', Adam(lr .05, b1 0).  One step = one train(i):
synth -> sampler -> ViT fwd -> loss -> ViT input-grad -> sampler adjoint -> rfft2 adjoint -> [all-reduce] -> Adam.
Synthetic data: seeded random ViT weights / target embedding (no checkpoint or network here).  N > 1 splits the cuts
across ranks (strong scaling of the fixed 200-sample step) with one RCCL all-reduce of the spectrum gradient per step.

The ONE JSON line (rank 0) carries, besides the driver's fields:
  * `value` = the headline (`-tf fast`, per-step frame off);
  * `legs`  = at N = 1 the same K steps again for `-tf none` (200 cuts: the configuration whose parity with the reference is
    pinned end to end) and for `-tf fast` with the per-step frame written as the reference does (clip_fft.py:297-306);
  * `roofline` = the ViT GEMM family timed per launch with HIP events on the launch stream (+ `step_frac`, the whole
    step's algorithmic FLOPs against the MFMA peak; `traffic` from the committed PMC summary of THIS build, else flagged
    stale); `cpu_baseline` = the CPU oracle's train(i) on the host cores.
Other configurations (`--config c1|c3|c4`) print the same line for their workload.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_IMG = {'ViT-B/32': 8817623040, 'ViT-B/16': 35126906880}     # SURVEY.md section 8d (fwd FLOPs / image)
F_T, F_P, F_KP = {'ViT-B/32': 50, 'ViT-B/16': 197}, {'ViT-B/32': 49, 'ViT-B/16': 196}, {'ViT-B/32': 3072, 'ViT-B/16': 768}      # tokens, patches per cut, patch K
PEAK_TF, HBM_ACHIEVABLE_GBS = 2500.0, 6300.0                    # MI355X_MICROARCH.md: dense f16 MFMA; achievable HBM3E

CONFIGS = {     # SURVEY.md section 8 config shorthand
    'c1': dict(size='224-224', samples=1, model='ViT-B/32', transform='none', macro=0.0, note='BASELINE configs[0] (reference plumbing case; S = 1, -tf none, --macro 0)'),
    'c2': dict(size='1280-720', samples=200, model='ViT-B/32', transform='fast', macro=0.4, note='BASELINE configs[1] (the headline)'),
    'c3': dict(size='1280-720', samples=200, model='ViT-B/32', transform='fast', macro=0.4, dualmod=2, note='BASELINE configs[2]: --dualmod 2, ViT-B/32 <-> ViT-B/16 on one Adam state'),
    'c4': dict(size='3840-2160', samples=400, model='ViT-B/16', transform='fast', macro=0.4, dwt='db3', note='BASELINE configs[3]: --dwt -w db3, ViT-B/16'),
    'c5': dict(size='1280-720', samples=100, model='ViT-B/32', transform='fast', macro=0.3, illustrip='RGB', align='overscan', colors=2.3, lr=0.1,
               note='BASELINE configs[4] without the depth warp (no weights): illustrip continuous mode, --gen RGB, per frame = warp + re-parameterise + fresh Adam + 1 step'),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    p.add_argument('--size', default=None)
    p.add_argument('--samples', type=int, default=None)
    p.add_argument('--model', default=None)
    p.add_argument('--transform', default=None, choices=['fast', 'none'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--no-legs', action='store_true', help='skip the -tf none and with-save legs (N = 1 only has them)')
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    p.add_argument('--split', action='store_true', help='the opt-in split-precision forward (patch-embedding and QKV GEMMs on hi + lo f16 activation pairs): ~5 %% slower; '
                                                       'on the stress-weight loss-curve ensemble it is statistically indistinguishable from the default (profiles/r06_precision_ensemble.txt)')
    p.add_argument('--f16', action='store_true', help='(the default since round 6; accepted for old command lines) f16 operands on every ViT GEMM')
    p.add_argument('--reps', type=int, default=3, help='repetitions of the timed block of --steps steps (value = the median block)')
    p.add_argument('--vit-path', default=None, help='measurement switch: comma list of name=int pairs handed to the library\'s test hooks '
                                                    '(rs: aph_gemm_set_rs, fused: aph_vit_set_fused_max_rows, ws: aph_gemm_set_ws_min_tiles, stream16: aph_vit_set_grad_stream_f16)')
    p.add_argument('--grad-f16', default=None, type=int, choices=[0, 1], help='measurement switch: Engine(grad_f16=...) -- the patch gradient handed to the sampler adjoint as f16 (1) or f32 (0); default: the engine\'s')
    a = p.parse_args()
    if a.split and a.f16:
        p.error('--split and --f16 are mutually exclusive')
    a.f16 = not a.split
    cfg = dict(CONFIGS[a.config])
    for k in ('size', 'samples', 'model', 'transform'):
        if getattr(a, k) is not None:
            cfg[k] = getattr(a, k)
    a.cfg = cfg
    return a


def derate(samples, model, transform, dualmod):
    """clip_fft.py:125-127,134,167-169"""
    s = samples
    if model == 'ViT-B/16':
        s = int(s * 0.25)
    if dualmod is not None:
        s = int(s * 0.23)
    if transform == 'fast':
        s = int(s * 0.95)
    return s


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(w, h, model_name, samples, seed=0, budget_s=30.0, tf='none'):
    """The oracle's train(i) (oracle/reference_path.py: fp32 torch-CPU restatement of the reference's own path, -tf none) timed on
    the host cores.  (1) thread sweep: one warm 16-cut + one timed 64-cut step at each of {16, 32, 64, 128} threads (torch's default -- every hardware
    thread of the box -- is catastrophically oversubscribed on this workload: 17.9 s per step at 128 threads in round 2, 5.6 s at 16) ; (2) the two best counts are
    each timed at the FULL size (one 16-cut warm-up, then up to 3 / 2 full steps, fewer if the budget of ~30 s per count would be exceeded; median) and the
    better one is reported; (3) one more step in the
    reference's DEFAULT form, where the CLIP weights require grad and their gradients are computed and thrown away (BASELINE.md
    section 4) -- reported next to the input-gradient-only number, never instead of it."""
    from oracle import reference_path as R
    from oracle import augment_ref, clip_vit_ref
    from aphantasia_amd.weights import synthetic_visual_weights, visual_config
    cfg = visual_config(model_name)
    wts = synthetic_visual_weights(cfg, 1)
    target = torch.randn(1, cfg['output_dim'], generator=torch.Generator().manual_seed(2))
    torch.manual_seed(seed)
    nproc = os.cpu_count() or 1
    t_before = torch.get_num_threads()
    small = min(16, samples)

    # kind "reference" when the reference tree is importable (the build container; never the GPU box): train(i) assembled from the
    # reference's OWN fft_image / to_valid_rgb / slice_imgs / normalize / sim_func (oracle/shim.py imports them in place) + torch.optim.Adam
    # as at clip_fft.py:108-115,235-295 -- only CLIP's encode_image is the restatement (openai/CLIP is a third-party package that is not
    # in the reference tree).  Otherwise kind "port": the oracle's restatement of the same functions.
    ref = None
    try:
        from oracle import shim
        if shim.available():
            ref = shim.load_reference()
    except Exception as e:
        print('bench.py: the reference tree is present but could not be imported (%r): cpu_baseline falls back to the port' % (e,), file=sys.stderr)
        ref = None

    class RefRun:
        """the reference's own functions, called as clip_fft.py:91-101,108-115,235-295 calls them (-tf none)"""
        def __init__(self, weights):
            self.params, image_f, _ = ref.fft_image([1, 3, h, w], 0.01, 1.5)
            self.image_f = ref.to_valid_rgb(image_f, colors=1.8)
            self.opt = torch.optim.Adam(self.params, 0.05, betas=(.0, .999))
            self.enc = lambda x: clip_vit_ref.encode_image(weights, x, cfg)
            self.norm = ref.normalize()

        def step(self, n):
            img = self.image_f(None)
            cuts = ref.slice_imgs([img], n, 224, self.norm, 'uniform', 0.4)[0]
            loss = -1.0 * ref.sim_func(target, self.enc(cuts), 'mix')
            self.opt.zero_grad()
            loss.backward()
            self.opt.step()
            return float(loss.detach())

    def fresh(weights):
        if ref is not None:
            return RefRun(weights)
        return R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(weights, x, cfg), [(target, 1.0)])

    def one(run, n):
        if ref is not None:
            t0 = time.perf_counter()
            run.step(n)
            return time.perf_counter() - t0
        if tf == 'fast':       # the same augment chain as `value` (transforms.py:165-170 through the oracle's restated torchvision ops), drawn per cut in the reference's order
            augs = []
            table = R.draw_crop_table(n, 224, h, w, 'uniform', 0.4, per_cut_hook=lambda _c: augs.append(augment_ref.draw_fast_params(224)))
            t0 = time.perf_counter()
            run.step(table, lambda k, cut: augment_ref.apply_fast(cut, augs[k], R.normalize))
            return time.perf_counter() - t0
        table = R.draw_crop_table(n, 224, h, w, 'uniform', 0.4)
        t0 = time.perf_counter()
        run.step(table)
        return time.perf_counter() - t0
    # (1) thread sweep at a size whose GEMMs scale like the real step's: 64 cuts (round 3 swept on 16 cuts, whose batch-12x-smaller GEMMs stop
    #     scaling earlier than the 190-cut step's do -- VERDICT r3 weak #12)
    sweep_cuts = min(64, samples)
    sweep = {}
    for t in sorted({min(c, nproc) for c in (16, 32, 64, 128)}):      # (all 256 hardware threads of the GPU box: 109 s for ONE 16-cut step)
        torch.set_num_threads(t)
        run = fresh(wts)
        one(run, small)
        sweep[t] = one(run, sweep_cuts)
    # (2) the full-size step at the two best counts of the sweep: up to 3 timed steps at the best (median), 2 at the runner-up; the better wins
    ranked = sorted(sweep, key=sweep.get)[:2]
    full = {}
    for k, t in enumerate(ranked):
        torch.set_num_threads(t)
        run = fresh(wts)
        one(run, small)
        times, t_start = [], time.perf_counter()
        for _ in range(3 if k == 0 else 2):
            times.append(one(run, samples))
            if time.perf_counter() - t_start + times[-1] > budget_s:
                break
        full[t] = times
    med_of = lambda ts: sorted(ts)[len(ts) // 2]
    best = min(full, key=lambda t: med_of(full[t]))
    times, med = full[best], med_of(full[best])
    torch.set_num_threads(best)
    wg = {k: v.clone().requires_grad_(True) for k, v in wts.items()}
    run_wg = fresh(wg)
    one(run_wg, small)
    t_wg = one(run_wg, samples)
    torch.set_num_threads(t_before)
    tf_used = 'none' if ref is not None else tf
    return dict(value=1.0 / med, unit='steps/s', cores=best, kind='reference' if ref is not None else 'port', cpu=cpu_model(), host_threads_available=nproc,
                transform=tf_used,
                sample='%d timed full train(i) steps (median) at %dx%d, %d cuts, %s, -tf %s%s, fp32 torch-CPU %s, %d threads (the better of the two best '
                       'counts of a %d-cut sweep over {16, 32, 64, 128} threads, each timed at the full size), after a %d-cut warm-up step'
                       % (len(times), w, h, samples, model_name, tf_used,
                          '' if tf_used == tf else ' (`value` runs -tf %s: the reference\'s transforms_fast needs torchvision, which is not in this image)' % tf,
                          "reference functions (aphantasia/image.py, utils.py, transforms.py imported in place; CLIP encode_image = the oracle's restatement)" if ref is not None
                          else 'oracle restatement of the reference path (the reference tree is not on this machine)', best, sweep_cuts, small),
                seconds=med, seconds_all=times, thread_sweep={'cuts': sweep_cuts, 'seconds_per_step': {str(k): v for k, v in sweep.items()}},
                full_size_seconds_by_threads={str(k): v for k, v in full.items()},
                reference_default=dict(value=1.0 / t_wg, seconds=t_wg, note='CLIP weights require grad: their gradients are computed and discarded, as '
                                       'the reference does by default (BASELINE.md section 4); 1 timed step, same thread count'))


def mfma_peak(lib, dev):
    """pure-MFMA rate of this GPU, measured in this run (aph_mfma_rate: v_mfma_f32_16x16x32_f16, 256 workgroups x 8 waves, no memory
    traffic): what the matrix pipe sustains on random operands (the part is power limited: zeros run faster) -- the practical ceiling
    next to the 2.5 PFLOP/s datasheet peak `frac` is quoted against"""
    from aphantasia_amd.ops import ptr, _stream
    out = torch.empty(1024, device=dev)
    res = {}
    for label, src in (('random_operands', torch.randn(8192 * 8, device=dev).half()), ('zero_operands', torch.zeros(8192 * 8, device=dev).half())):
        blocks, iters = 256, 4000
        f = lambda: lib.call('aph_mfma_rate', blocks, iters, ptr(src), ptr(out), _stream(out))
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[label] = blocks * 8.0 * iters * 32 * 16384 / (e0.elapsed_time(e1) / 4 * 1e-3) / 1e12
    res['unit'] = 'TFLOP/s'
    return res


def irdwt_roofline(eng):
    """the HBM-bound part of C4: inverse DWT forward + adjoint of a DWT-parameterised engine, timed with events on the (current) launch
    stream; algorithmic bytes = every level's inputs read once + its output written once"""
    syn = eng.dwt
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    graw = torch.randn_like(eng.raw)
    reps = 20
    for _ in range(2):
        syn.forward(eng.params)
        syn.backward(graw, eng.grad)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        syn.forward(eng.params)
    e1.record()
    for _ in range(reps):
        syn.backward(graw, eng.grad)
    e2.record()
    torch.cuda.synchronize()
    elems = 0
    for (hh, ww), (ho, wo) in zip(syn.sizes, syn.out_sizes):
        elems += 3 * (4 * hh * ww + ho * wo)              # read ll + 3 detail bands, write the level's output
    by = 4.0 * elems
    tf, tb = e0.elapsed_time(e1) / reps * 1e-3, e1.elapsed_time(e2) / reps * 1e-3
    return dict(bound='hbm', kernel='aph::idwt_level_kernel + idwt_coarse_kernel / their adjoints (all levels, one aph_idwt_fwd / aph_idwt_bwd call each)', unit='GB/s',
                peak=HBM_ACHIEVABLE_GBS, algorithmic_bytes_per_pass=by, fwd_us=tf * 1e6, bwd_us=tb * 1e6,
                achieved=by / tf / 1e9, frac=by / tf / 1e9 / HBM_ACHIEVABLE_GBS,
                achieved_adjoint=by / tb / 1e9, frac_adjoint=by / tb / 1e9 / HBM_ACHIEVABLE_GBS, **irdwt_traffic())


def other_config_legs(names, steps=30, timeout_s=90):
    """`python bench.py --config <name>` (no legs, no CPU baseline, no roofline pass) in a child process per configuration -> its value line, abridged"""
    import subprocess
    out = {}
    for name in names:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--config', name, '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-legs',
                                '--no-roofline'], capture_output=True, text=True, timeout=timeout_s)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not lines:
                out[name] = dict(error='exit code %d: %s' % (r.returncode, r.stderr.strip()[-300:]))
                continue
            d = json.loads(lines[-1])
            out[name] = dict(value=d['value'], unit=d['unit'], ms_per_step=d['ms_per_step'], steps=d['steps'], workload=d['config'].get('workload'),
                             samples_effective=d['config'].get('samples_effective'), skipped_steps=d['config'].get('skipped_steps'))
        except Exception as e:
            out[name] = dict(error=repr(e))
    return out


def precision_block(f16):
    """what the line's precision mode is pinned by: the committed ensemble summary (tools/loss_ensemble.py on the GPU against the 30 oracle
    trajectories of tests/golden/ensemble) + the tests that gate it"""
    out = dict(mode='f16_everywhere' if f16 else 'split',
               pinned_by='tests/test_gpu_parity_configs.py::test_stress_weights_loss_curve_ensemble_default_mode (ensemble gates: every member < 3e-3, median member < 1e-3, '
                         'mean |d loss| < 4e-4, at most half of the members past 1e-3) + the hard per-step 1e-3 gates on BASELINE\'s own configurations '
                         '(::test_c2_loss_curve_200cuts_50steps / _200steps / test_c2_fast_loss_curve_190cuts_60steps, plain synthetic weights)',
               note='north_star: loss-vs-step curve within 1e-3 of the CPU reference.  On BASELINE\'s configurations with CLIP-initialised synthetic weights both modes hold it with margin '
                    '(1e-4 ... 5e-4).  On deliberately hostile "stress" weights the free-running curve is a chaotic amplifier of rounding ORDER: over 30 oracle trajectories x 3 rounding sequences '
                    'both modes exceed 1e-3 on a minority of members and neither is significantly closer (see `ensemble`), so the default is the reference\'s own GPU dtype.')
    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_precision_ensemble.json')) as f:
            j = json.load(f)
        sm = j['summary']
        out['ensemble'] = dict(source='profiles/r06_precision_ensemble.json', lib_sha256=j.get('lib_sha256'),
                               f16={k: sm['f16/any'][k] for k in ('n', 'exceed_1e3', 'median_max', 'p90_max', 'worst')},
                               split={k: sm['split/any'][k] for k in ('n', 'exceed_1e3', 'median_max', 'p90_max', 'worst')},
                               paired=sm.get('paired'), exceedance=sm.get('exceedance'), order_spread_max_over_min=sm.get('order_spread_max_over_min'))
    except (OSError, KeyError, ValueError):
        out['ensemble'] = None
    return out


def lib_sha():
    from aphantasia_amd import _ffi
    with open(_ffi.LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


GEMM_SOURCES = ('aph_device.h', 'aph_host.h', 'vit_gemm.h', 'vit_gemm_ws.h', 'vit_gemm_rs.h', 'vit_ops.h', 'vit_attn.h', 'vit.hip')   # the whole ViT translation unit: GEMM kernels, launch heuristic, launch sites / epilogue choice


def gemm_src_sha():
    """sha256 over the sources that define every GEMM kernel and its launch heuristic"""
    hsh = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(ROOT, 'aphantasia_amd', 'csrc', name), 'rb') as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def dwt_src_sha():
    hsh = hashlib.sha256()
    for name in ('aph_device.h', 'aph_host.h', 'dwt.hip'):
        with open(os.path.join(ROOT, 'aphantasia_amd', 'csrc', name), 'rb') as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def irdwt_traffic():
    """HBM-side bytes per inverse-DWT pass (forward, adjoint) from the newest committed `profiles/r*_c4_pmc_hbm_traffic.json`
    (tools/pmc_traffic.py on `bench.py --config c4`), accepted for this library or for a byte-identical csrc/dwt.hip"""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_c4_pmc_hbm_traffic.json'))):
        with open(path) as f:
            j = json.load(f)
        if 'irdwt_fwd_bytes_per_pass' not in j:
            continue
        match = 'library' if j.get('lib_sha256') == lib_sha() else 'dwt_sources' if j.get('dwt_src_sha256') == dwt_src_sha() else None
        if match or best is None or best['traffic_match'] is None:
            best = dict(traffic=j['irdwt_fwd_bytes_per_pass'], traffic_adjoint=j['irdwt_bwd_bytes_per_pass'], traffic_unit='bytes/pass (2*FETCH_SIZE + WRITE_SIZE)',
                        traffic_source=os.path.relpath(path, ROOT), traffic_stale=match is None, traffic_match=match)
    return best or dict(traffic=None, traffic_adjoint=None, traffic_source=None, traffic_stale=None, traffic_match=None)


def pmc_traffic(tag_glob):
    """HBM-side bytes per GEMM launch from a committed PMC summary (tools/pmc_traffic.py).  Accepted only if it was taken on
    the very library that is being timed, or on a library whose GEMM sources (GEMM_SOURCES) are byte-identical to this
    checkout's -- a later change to another translation unit does not move the GEMM kernels' traffic; anything else is
    reported as stale.  -> (bytes per launch, file, stale, what matched)"""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', tag_glob))):      # (by name: r01 < r02 < r03; mtimes mean nothing after a checkout)
        with open(path) as f:
            j = json.load(f)
        match = 'library' if j.get('lib_sha256') == lib_sha() else 'gemm_sources' if j.get('gemm_src_sha256') == gemm_src_sha() else None
        if match or best is None or best[3] is None:       # the newest matching summary, else the newest one (flagged stale)
            best = (j.get('traffic_bytes_per_launch'), os.path.relpath(path, ROOT), match is None, match)
    return best if best else (None, None, None, None)


RUNGS = ('graph+rccl', 'eager+rccl', 'eager+torch')
RUNG_NOTE = {'graph+rccl': 'whole step incl. ncclAllReduce in ONE hipGraph, RCCL called directly (aph_allreduce_f32); control plane (barriers, timing MAX) on a gloo group',
             'eager+rccl': 'eager launches, RCCL called directly (aph_allreduce_f32); control plane on a gloo group',
             'eager+torch': 'eager launches, all-reduce through torch.distributed (nccl backend = RCCL)'}


def launch_supervisors(a):
    """plain `python bench.py --gpus N`: start the N rank processes here, exactly as torchrun would (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment); each of them then supervises its own worker (supervise())"""
    import shutil
    import subprocess
    import tempfile
    from aphantasia_amd.comm import free_port
    sup = tempfile.mkdtemp(prefix='aph_bench_sup_')
    port = free_port()
    procs = []
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       APH_BENCH_SUP_DIR=sup, HSA_ENABLE_IPC_MODE_LEGACY='0')
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = max(p.wait() for p in procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(sup, ignore_errors=True)
    if rc:
        raise SystemExit(rc)


def supervise(a, rank, world):
    """A rank process of a multi-rank launch (torchrun's or launch_supervisors'): it does no GPU work itself.  It runs the measurement in
    a WORKER child, one rung of RUNGS at a time under a wall budget (APH_BENCH_RUNG_BUDGET seconds, default 180), and agrees with the other
    ranks' supervisors through files on whether the rung succeeded everywhere (aphantasia_amd.comm.ladder).  A worker that hangs in a
    capture or a collective is killed by its supervisor and the next rung starts on every rank; rank 0 prints the successful rung's JSON
    line with `config.multi_rank_mode` and the ladder's record -- or, if every rung failed, a line that says so.  (the
    first N > 1 run must not be able to end without a line.)"""
    import tempfile
    from aphantasia_amd.comm import ladder
    sup = os.environ.get('APH_BENCH_SUP_DIR') or os.path.join(tempfile.gettempdir(), 'aph_bench_sup_%d_%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0')))
    budget = float(os.environ.get('APH_BENCH_RUNG_BUDGET', '180'))
    rungs = [r for r in os.environ.get('APH_BENCH_RUNGS', ','.join(RUNGS)).split(',') if r]
    for r in rungs:
        if r not in RUNGS:
            raise SystemExit('APH_BENCH_RUNGS: unknown rung %r (known: %s)' % (r, ', '.join(RUNGS)))
    os.environ.pop('TORCHELASTIC_USE_AGENT_STORE', None)      # the workers rendezvous on a port of their own (rank 0's worker hosts the store)

    def make_cmd(k, name, port):
        return ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                dict(APH_BENCH_WORKER='1', APH_BENCH_RUNG=name, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), APH_BENCH_RUNG_BUDGET=str(budget)))
    k, records, out_path = ladder(rank, world, rungs, make_cmd, sup, budget_s=budget, grace_s=30.0,
                                  log=lambda m: print('bench.py supervisor ' + m, file=sys.stderr, flush=True))
    if rank == 0:
        line = None
        if k is not None:
            with open(out_path) as f:
                cand = [l for l in f.read().splitlines() if l.startswith('{')]
            if cand:
                line = json.loads(cand[-1])
                line['config']['multi_rank_mode'] = rungs[k]
                line['config']['multi_rank_mode_note'] = RUNG_NOTE[rungs[k]]
                line['config']['multi_rank_ladder'] = records
        if line is None:
            line = {'metric': 'optimization steps/sec', 'value': None, 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                    'ms_per_step': None, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'data': 'synthetic',
                    'error': 'every rung of the multi-rank ladder failed: no measurement', 'config': {'multi_rank_mode': None, 'multi_rank_ladder': records}}
        print(json.dumps(line), flush=True)
    if k is None:
        raise SystemExit(1)


def main():
    a = parse()
    cfg = a.cfg
    if a.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks here (one process per GPU), exactly what torchrun would have done
        have = torch.cuda.device_count()
        if have < a.gpus and os.environ.get('APH_BENCH_BACKEND', 'nccl') == 'nccl':
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to print a line that is not an N-GPU measurement' % (a.gpus, have))
        launch_supervisors(a)
        return
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if world > 1 and os.environ.get('APH_BENCH_WORKER') != '1':
        supervise(a, rank, world)
        return
    rung = os.environ.get('APH_BENCH_RUNG', RUNGS[0]) if world > 1 else None
    # in-worker watchdog: shortly before the supervisor's budget runs out, dump every thread's Python stack to stderr (the supervisor then
    # kills the worker): a rank stuck in a collective leaves a stack behind, not just a timeout.  Cancelled after the timed section.
    wd = 0
    if world > 1:
        wd = max(int(float(os.environ.get('APH_BENCH_RUNG_BUDGET', '180'))) - 15, 5)
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=False)
    # debugging aid for single-GPU boxes: APH_BENCH_BACKEND=gloo puts every rank on cuda:0 and reduces through the host
    backend = os.environ.get('APH_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    pg = None
    ctl = dev                     # device of the control-plane tensors (timing MAX, parameter hashes)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')      # one node: the control plane stays on loopback (the container's hostname may not resolve)
        # ONE RCCL communicator per GPU: with the direct communicator (rungs 1 / 2) the control plane is a gloo group; only the
        # torch.distributed rung holds an nccl group (and then no direct communicator)
        if backend == 'nccl' and rung == 'eager+torch':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            ctl = torch.device('cpu')

    from aphantasia_amd import clip as aclip, transforms
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.clip import LOSS_SCALE
    if a.vit_path:                          # A/B runs only: the default line never passes this
        from aphantasia_amd import _ffi
        hooks = dict(rs='aph_gemm_set_rs', fused='aph_vit_set_fused_max_rows', fattn='aph_vit_set_fused_attn', ws='aph_gemm_set_ws_min_tiles', stream16='aph_vit_set_grad_stream_f16')
        for kv in a.vit_path.split(','):
            k, v = kv.split('=')
            if not hasattr(_ffi.lib().cdll, hooks[k]):
                raise SystemExit('--vit-path %s: %s exists in -DAPH_EXPERIMENTS builds only (python -m aphantasia_amd._build --experiments)' % (kv, hooks[k]))
            getattr(_ffi.lib().cdll, hooks[k])(int(v))
    # the step's collective: RCCL called directly through the C ABI (aph_allreduce_f32); torch.distributed only carried the
    # 128-byte unique id and does the barriers / the MAX over ranks of the timing contract.  APH_COMM=torch: all-reduce through
    # torch.distributed instead (cross-check)
    comm = None
    if world > 1 and backend == 'nccl' and rung in ('graph+rccl', 'eager+rccl'):
        from aphantasia_amd import comm as acomm
        comm = acomm.create(rank, world)      # (a failure here fails this rung on every rank: the supervisors move on to the next one)
    use_graph = not a.no_graph and rung in (None, 'graph+rccl')
    if world > 1 and os.environ.get('APH_BENCH_INJECT_HANG') == rung and rank == world - 1:
        # TEST-ONLY hang injection (tools/gpu.sh mr2hang, tests of the ladder): the last rank never reaches the step's collective
        print('bench.py rank %d: APH_BENCH_INJECT_HANG=%s -- sleeping forever (test)' % (rank, rung), file=sys.stderr, flush=True)
        time.sleep(1e6)
    w, h = [int(s) for s in cfg['size'].split('-')]
    dualmod = cfg.get('dualmod')
    S = derate(cfg['samples'], cfg['model'], cfg['transform'], dualmod)
    S_none = derate(cfg['samples'], cfg['model'], 'none', dualmod)
    if S < 1:
        raise SystemExit('no effective cuts')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load(cfg['model'], weights=None, seed=1, max_batch=8)
        model2 = aclip.load('ViT-B/16', weights=None, seed=1, max_batch=8)[0] if dualmod is not None else None
    sim = 'cossim' if dualmod is not None else 'mix'                  # clip_fft.py:88
    target = torch.randn(1, model.visual.output_dim, generator=torch.Generator().manual_seed(2))
    target2 = torch.randn(1, model.visual.output_dim, generator=torch.Generator().manual_seed(3))

    def make(transform_name, S_eff, **extra):
        """(engines [main, dual or None], synth) on freshly initialised parameters"""
        trf = transforms.transforms_fast if transform_name == 'fast' else transforms.normalize()
        torch.manual_seed(0)
        np.random.seed(0)
        kw = dict(sim=sim, transform=trf, macro=cfg['macro'], rank=rank, world=world, process_group=pg, comm=comm, use_graph=use_graph, graph_allreduce=(rung == 'graph+rccl') or None)
        for k in ('align', 'colors', 'lr'):
            if k in cfg:
                kw[k] = cfg[k]
        if cfg.get('illustrip'):          # illustrip.py:270-276,438-440: pixel parameters + brightness / contrast priors
            leaf = (0.3 * torch.randn(1, 3, h, w)).to(dev).contiguous()
            kw.update(param_kind='pixel', rgb_priors=True)
        elif cfg.get('dwt'):
            from aphantasia_amd.image import dwt_image
            params, image_f, _ = dwt_image([1, 3, h, w], cfg['dwt'], 0.3, 1.8, None)
            leaf = image_f.flat
            kw.update(param_kind='dwt', dwt=image_f.synth)
        else:
            leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(dev).contiguous()
        kw['precise'] = not a.f16       # headline mode [r6]: f16 operands everywhere (the reference's own GPU dtype); --split = the opt-in split-precision forward
        if a.grad_f16 is not None:
            kw['grad_f16'] = bool(a.grad_f16)
        kw.update(extra)
        e1 = Engine(leaf, h, w, model, S_eff, [(target, -1.0)], **kw)
        e2 = Engine(leaf, h, w, model2, S_eff, [(target2, -1.0)], state=e1.state(), **kw) if dualmod is not None else None
        return e1, e2

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(e1, e2, steps, warmup, writer=None, tmpdir=None):
        """exactly `steps` steps between barrier + synchronize pairs; MAX over ranks"""
        loop = None
        if cfg.get('illustrip'):
            from aphantasia_amd.illustrip_loop import FrameLoop
            loop = FrameLoop(e1, gen=cfg['illustrip'], opt_step=1)

        def one(i):
            if loop is not None:                 # one illustrip frame: MOTION + re-parameterisation + fresh optimiser + the step (illustrip.py:381-470)
                loop.frame()
                return
            e = e2 if (e2 is not None and i >= dualmod and i % dualmod == 0) else e1          # list(range(steps))[dm::dm], clip_fft.py:135
            e.step()
            if writer is not None:                                                             # clip_fft.py:297-306 (opt_step = 1)
                img = e.synthesize(1.1)
                writer.put(img.reshape(3, e.h, e.w), os.path.join(tmpdir, '%04d.jpg' % i), 1.0)
        for i in range(warmup):
            one(i)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            one(i)
        if writer is not None:
            writer.drain()                        # every frame of the timed region is on disk before the clock stops
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], device=ctl, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    eng, eng_b = make(cfg['transform'], S)
    # the timed block of exactly --steps steps, --reps times back to back (warm-up before the first only); `value` is the MEDIAN block --
    # one block of 20 steps is 0.12 s, and box-to-box / run-to-run spread of a few percent is of the order of the effects reported here
    blocks = [timed(eng, eng_b, a.steps, a.warmup if r == 0 else 0) for r in range(max(a.reps, 1))]
    dt = sorted(blocks)[len(blocks) // 2]
    loss = eng.global_loss()
    # multi-rank sanity, outside the timed region: the communicator's own rank count, and the parameters bit-identical on every rank
    # after warm-up + K steps (a 64-bit hash of the bit patterns: every rank applied the same all-reduced gradient)
    ranks_seen, params_identical = 1, None
    if world > 1:
        import torch.distributed as dist
        ranks_seen = comm.ranks_seen() if comm is not None else dist.get_world_size()
        bits = eng.params.detach().reshape(-1).view(torch.int32).to(torch.int64)
        hsh = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()]).to(ctl)
        hs = [torch.zeros_like(hsh) for _ in range(world)]
        dist.all_gather(hs, hsh)
        params_identical = all(bool(torch.equal(hs[0], x)) for x in hs[1:])
        if ranks_seen != world or not params_identical:
            raise SystemExit('bench.py: rank check failed (communicator sees %d of %d ranks, parameters identical across ranks: %s)' % (ranks_seen, world, params_identical))
    if wd > 0:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
    skipped = int(eng.guard[0])          # steps whose fp16 backward overflowed and were skipped by the guarded Adam: must be 0 for a valid line
    flop_step = 2 * S * F_IMG[cfg['model']] / 1e12
    if dualmod is not None:     # the schedule's mix of B/32 and B/16 steps over the timed region
        n16 = len([i for i in range(a.steps) if i >= dualmod and i % dualmod == 0])
        flop_step = 2 * S * (F_IMG['ViT-B/32'] * (a.steps - n16) + F_IMG['ViT-B/16'] * n16) / a.steps / 1e12

    roof = None
    if not a.no_roofline:
        # same steps again with HIP events around every GEMM launch of the ViT (dominant kernel family)
        lib = eng.lib
        import ctypes
        vits = [e.visual.handle for e in (eng, eng_b) if e is not None]
        for e in (eng, eng_b):
            if e is not None:
                e.use_graph = False       # eager launches so that every GEMM gets its event pair
        for v in vits:
            lib.call('aph_vit_profile', v.handle, 1)
        nprof = min(a.steps, 10)
        for i in range(nprof):
            (eng_b if (eng_b is not None and i >= dualmod and i % dualmod == 0) else eng).step()
        torch.cuda.synchronize()
        ms_t, n_t, fl_t = 0.0, 0, 0.0
        for v in vits:
            ms, n, flops = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
            lib.call('aph_vit_profile_read', v.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(flops))
            lib.call('aph_vit_profile', v.handle, 0)
            ms_t, n_t, fl_t = ms_t + ms.value, n_t + n.value, fl_t + flops.value
        if n_t > 0:
            achieved = fl_t / (ms_t * 1e-3) / 1e12
            # what the hi | lo halves of the split-precision forward add on the matrix cores (QKV of every block + the patch embedding, width 768)
            m_ = cfg['model']
            hilo_extra = 0.0 if (a.f16 or dualmod is not None or m_ not in F_T) else 2.0 * S * (F_T[m_] * 2 * 768 * 768 * 12 + F_P[m_] * 768 * F_KP[m_])     # (the lo half feeds the Q and K columns only)
            traffic, tsrc, stale, tmatch = (None, None, None, None)
            if a.config == 'c2' and cfg == CONFIGS['c2'] and world == 1 and a.f16:      # (the PMC passes run the default mode)
                traffic, tsrc, stale, tmatch = pmc_traffic('r[0-9][0-9]_pmc_hbm_traffic*.json')
            # both floors of the family's average launch: matrix-core time of its FLOPs at the datasheet peak, and HBM time of its MEASURED fabric
            # traffic (PMC summary of this build) at the achievable bandwidth; `binding_floor` names the larger one, `frac_of_binding_floor` is
            # that floor over the measured launch time (on counter traffic the family sits below the ridge point).  `bound`,
            # `achieved`, `peak`, `unit`, `frac` stay ONE consistent triple -- the matrix-core figures every round has reported
            avg_us = ms_t * 1e3 / n_t
            floor_mfma_us = (fl_t / n_t) / (PEAK_TF * 1e12) * 1e6
            floor_hbm_us = (traffic / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6) if traffic else None
            binding = 'hbm' if (floor_hbm_us is not None and floor_hbm_us > floor_mfma_us) else 'mfma'
            roof = dict(bound='mfma', binding_floor=binding, floors=dict(mfma_us=floor_mfma_us, hbm_us=floor_hbm_us, ridge_flop_per_byte=PEAK_TF * 1e12 / (HBM_ACHIEVABLE_GBS * 1e9),
                                                   arithmetic_intensity_flop_per_byte=(fl_t / n_t / traffic) if traffic else None,
                                                   frac_of_binding_floor=(max(floor_mfma_us, floor_hbm_us or 0.0) / avg_us),
                                                   note='`achieved` / `peak` / `frac` stay the MFMA figures (algorithmic FLOPs over measured time against 2.5 PF) so that rounds compare; '
                                                        'on measured traffic the HBM floor is the higher of the two for this family'),
                        kernel='aph::gemm_ws_kernel<*> / aph::gemm_f16_kernel<*> / aph::gemm_sk_kernel<*> (the ViT GEMM family)', achieved=achieved, peak=PEAK_TF,
                        unit='TFLOP/s', frac=achieved / PEAK_TF, traffic=traffic, traffic_unit='bytes/launch (2*FETCH_SIZE + WRITE_SIZE; factors calibrated in profiles/r06_pmc_calibration.json)',
                        traffic_source=tsrc, traffic_stale=stale, traffic_match=tmatch, launches_per_step=n_t // nprof,
                        avg_launch_us=ms_t * 1e3 / n_t, flops_per_launch=fl_t / n_t, gemm_ms_per_step=ms_t / nprof,
                        executed_gemm_tflop_per_step=fl_t / nprof / 1e12,
                        executed_note='algorithmic FLOPs of the GEMM launches as launched (the last block runs its out-proj / MLP on the class rows only, which '
                                      'algorithmic_tflop_per_step -- the survey\'s definition -- still counts in full); in the split-precision mode the '
                                      'patch-embedding launch and the Q / K columns of the QKV launches execute twice these FLOPs on the matrix cores (hi and lo halves) -- counted ONCE here, '
                                      'so `achieved` is algorithmic work over measured time',
                        mfma_tflop_per_step_issued=(fl_t / nprof + hilo_extra) / 1e12,
                        step_frac=flop_step * (a.steps / dt) / PEAK_TF,
                        step_frac_note='algorithmic_tflop_per_step x steps/s / peak (whole step, every kernel and gap included)',
                        peak_measured=mfma_peak(lib, dev))
        if cfg.get('dwt'):
            roof = dict(roof or {}, irdwt=irdwt_roofline(eng))

    legs = None
    if world == 1 and not a.no_legs and a.config in ('c2', 'c3', 'c4') and cfg['transform'] == 'fast' and not cfg.get('illustrip'):
        legs = {}
        e_n, e_nb = make('none', S_none)
        dtn = timed(e_n, e_nb, a.steps, a.warmup)
        legs['tf_none'] = dict(value=a.steps / dtn, unit='steps/s', ms_per_step=1e3 * dtn / a.steps, samples_effective=S_none,
                               note='-tf none: the configuration whose parity with the reference is pinned end to end (tests/test_gpu_parity_configs.py)')
        del e_n, e_nb
        import shutil
        import tempfile
        import clip_fft
        tmpdir = tempfile.mkdtemp(prefix='aph_bench_frames_')
        try:
            e_s, e_sb = make(cfg['transform'], S)
            writer = clip_fft.FrameWriter(e_s.h, e_s.w)
            dts = timed(e_s, e_sb, a.steps, a.warmup, writer, tmpdir)
            writer.close()
            nfr = len([f for f in os.listdir(tmpdir) if f.endswith('.jpg')])
            legs['with_save'] = dict(value=a.steps / dts, unit='steps/s', ms_per_step=1e3 * dts / a.steps, frames_written=nfr,
                                     note='per-step frame on (clip_fft.py:297-306, opt_step 1): image_f(contrast 1.1) -> uint8 on the device '
                                          '-> pinned ring -> %d JPEG encoder threads; all frames on disk before the clock stops' % clip_fft.FrameWriter.THREADS)
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)

    if legs is not None and a.config == 'c2':
        # the reference's own draw order (utils.py:222-251 per cut on torch's / numpy's global generators: what every --seed run of the CLI
        # uses so that its crop tables reproduce the reference's) instead of the vectorised draws: same distributions, more host Python
        e_r, e_rb = make(cfg['transform'], S, rng='reference')
        dtr = timed(e_r, e_rb, a.steps, a.warmup)
        t0 = time.perf_counter()
        for _ in range(10):
            e_r.draw()
        host_ref = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10):
            eng.draw()
        host_bulk = (time.perf_counter() - t0) / 10
        legs['rng_reference'] = dict(value=a.steps / dtr, unit='steps/s', ms_per_step=1e3 * dtr / a.steps, host_draw_ms=1e3 * host_ref, host_draw_ms_bulk=1e3 * host_bulk,
                                     note="rng='reference': the reference's per-cut draw order (every --seed run); `value` uses the vectorised draws (rng='bulk')")
        del e_r, e_rb
        # the other precision mode, same workload as `value`: what the split-precision forward costs / what dropping it buys
        e_pr, e_prb = make(cfg['transform'], S, precise=bool(a.f16))
        dtpr = timed(e_pr, e_prb, a.steps, a.warmup)
        legs['f16_everywhere' if not a.f16 else 'split_precision'] = dict(
            value=a.steps / dtpr, unit='steps/s', ms_per_step=1e3 * dtpr / a.steps, skipped_steps=int(e_pr.guard[0]),
            note=('f16 operands on every ViT GEMM (what the reference itself runs CLIP at on a GPU): the default since round 6') if not a.f16 else
                 'bench.py --split / clip_fft.py --precise: patch-embedding and QKV GEMMs on hi + lo f16 activation pairs (twice their K on the Q / K columns); lower single-step '
                 'gradient error (7.7e-4 vs 1.10e-3), no significant difference on the 30-member stress-weight loss-curve ensemble (profiles/r06_precision_ensemble.txt)')
        del e_pr, e_prb
        # strong-scaling ceiling without an 8-GPU node: this GPU's step time at the per-rank shard sizes of 2 / 4 / 8 ranks (the collective
        # and its overlap are NOT in these numbers: 11 MB all-reduce per step)
        from aphantasia_amd.engine import shard_range
        proxy = {}
        for ranks in (2, 4, 8):
            lo, hi = shard_range(S, 0, ranks)
            e_p, _ = make(cfg['transform'], hi - lo)
            dtp = timed(e_p, None, a.steps, a.warmup)
            proxy['ranks_%d' % ranks] = dict(cuts=hi - lo, ms_per_step=1e3 * dtp / a.steps, speedup_bound=(dt / dtp))
            del e_p
        legs['shard_proxy'] = dict(note='one GPU running rank 0\'s share of the %d cuts (largest shard), no collective: upper bound of the '
                                        'strong-scaling speed-up at that rank count' % S, **proxy)
        # C4 (3840x2160 DWT db3, ViT-B/16) in the default line, short: the HBM-bound inverse DWT is only exercised there
        try:
            c4 = dict(CONFIGS['c4'])
            w4, h4 = [int(v) for v in c4['size'].split('-')]
            S4 = derate(c4['samples'], c4['model'], c4['transform'], None)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                m4 = aclip.load(c4['model'], weights=None, seed=1, max_batch=8)[0]
            from aphantasia_amd.image import dwt_image
            torch.manual_seed(0)
            np.random.seed(0)
            _, image_f, _ = dwt_image([1, 3, h4, w4], c4['dwt'], 0.3, 1.8, None)
            e4 = Engine(image_f.flat, h4, w4, m4, S4, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=c4['macro'],
                        param_kind='dwt', dwt=image_f.synth, use_graph=use_graph)
            n4 = min(a.steps, 10)
            dt4 = timed(e4, None, n4, 3)
            legs['c4'] = dict(value=n4 / dt4, unit='steps/s', ms_per_step=1e3 * dt4 / n4, steps=n4, samples_effective=S4, skipped_steps=int(e4.guard[0]),
                              note='BASELINE configs[3]: 3840x2160 DWT db3, ViT-B/16, --samples 400 -> %d cuts (full line: --config c4)' % S4)
            try:                     # its HBM-bound part against the achievable bandwidth (roofline.irdwt of the --config c4 line)
                legs['c4']['irdwt'] = irdwt_roofline(e4)
            except Exception as e:
                legs['c4']['irdwt'] = dict(error=repr(e))
            del e4, m4
        except Exception as e:       # the C4 leg must never take the headline down with it
            legs['c4'] = dict(error=repr(e))
        # the other configurations of SURVEY section 8 (C1, C3, C5), short, each in a process of its own (`--config cN`): what a failure or
        # a hang there costs is that entry, nothing else
        if a.config == 'c2' and cfg == CONFIGS['c2']:
            legs['other_configs'] = other_config_legs(('c1', 'c3', 'c5'))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if cfg.get('dwt') or dualmod is not None or cfg.get('illustrip'):
            cpu = None          # the oracle leg is defined for the FFT single-model step; C3 / C4 lines report the GPU side only
        else:
            cpu = cpu_baseline(w, h, cfg['model'], S, tf=cfg['transform'])

    if rank == 0:
        steps_per_s = a.steps / dt
        kind = 'DWT %s' % cfg['dwt'] if cfg.get('dwt') else ('RGB-pixel (illustrip frame loop)' if cfg.get('illustrip') else 'FFT')
        out = {
            'metric': 'optimization steps/sec @%dx%d %s samples=%d' % (w, h, cfg['model'], cfg['samples']),
            'value': steps_per_s, 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': ('f16 (MFMA operands, fp32 accumulate; fp32 residual stream / LayerNorm / softmax statistics / synth / sampler / loss / Adam)' if a.f16 else
                      'f16 MFMA operands with hi + lo split activations on the patch-embedding / QKV GEMMs, fp32 accumulate; fp32 synth/sampler/loss/Adam'),
            'precision': precision_block(a.f16),
            'repeats': dict(n=len(blocks), steps_per_s=sorted(a.steps / t for t in blocks), median=a.steps / dt, block_s=blocks),
            'data': 'synthetic',
            'config': {'workload': '%s: %dx%d %s parameteriser, %s%s, --samples %d -> %d effective cuts, -tf %s, sim %s, '
                                   'Adam(lr .05, b1 0), per-step image save off' % (a.config.upper(), w, h, kind, cfg['model'],
                                                                                    ' + ViT-B/16 every %d steps (--dualmod)' % dualmod if dualmod else '',
                                                                                    cfg['samples'], S, cfg['transform'], sim),
                       'note': cfg['note'], 'samples_effective': S,
                       'parallelism': 'samples split over %d rank(s), 1 all-reduce/step (%s)' % (world, 'none' if world == 1 else ('RCCL direct, aph_allreduce_f32' if comm is not None else 'torch.distributed ' + backend)),
                       'rccl_ranks_seen': ranks_seen, 'params_identical_across_ranks': params_identical,
                       'loss_scale': LOSS_SCALE, 'final_loss': loss, 'skipped_steps': skipped, 'algorithmic_tflop_per_step': flop_step,
                       'lib_sha256': lib_sha()[:16]},
            'legs': legs, 'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        # every rank has its result: leave together, and leave NOW -- no communicator / process-group destructors (a teardown that hangs
        # or errors on one rank after the line is out would fail the rung for nothing)
        import torch.distributed as dist
        dist.barrier()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
