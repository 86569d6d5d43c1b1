#!/usr/bin/env python
"""bench.py -- optimisation steps/sec of the CLIP-guided hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 1280x720 FFT parameteriser, ViT-B/32, `--samples 200` with the
CLI defaults (-tf fast => 190 effective cuts, clip_fft.py:167-169), sim 'mix', Adam(lr .05, b1 0).
One step = one train(i): synth -> sampler -> ViT fwd -> loss -> ViT input-grad -> sampler adjoint ->
rfft2 adjoint -> [all-reduce] -> Adam.  Synthetic data: seeded random ViT weights / target embedding
(no checkpoint or network here).  N > 1 splits the cuts across ranks (strong scaling of the fixed
200-sample step) with one RCCL all-reduce of the spectrum gradient per step.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
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


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--size', default='1280-720')
    p.add_argument('--samples', type=int, default=200)
    p.add_argument('--model', default='ViT-B/32')
    p.add_argument('--transform', default='fast', choices=['fast', 'none'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    return p.parse_args()


def cpu_baseline(w, h, model_name, samples, seed=0):
    """The oracle's train(i) (oracle/reference_path.py, fp32 torch-CPU restatement of the reference's
    own path, -tf none) timed on the host cores: one warm-up step on 4 cuts, one timed full step."""
    from oracle import reference_path as R
    from oracle import clip_vit_ref
    from aphantasia_amd.weights import synthetic_visual_weights, visual_config
    cfg = visual_config(model_name)
    wts = synthetic_visual_weights(cfg, 1)
    target = torch.randn(1, cfg['output_dim'], generator=torch.Generator().manual_seed(2))
    torch.manual_seed(seed)
    run = R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)])
    run.step(R.draw_crop_table(4, 224, h, w, 'uniform', 0.4))
    table = R.draw_crop_table(samples, 224, h, w, 'uniform', 0.4)
    t0 = time.perf_counter()
    run.step(table)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit='steps/s', cores=torch.get_num_threads(), kind='port',
                sample='1 full train(i) at %dx%d, %d cuts, %s, fp32 torch-CPU oracle (-tf none), after a 4-cut warm-up step'
                       % (w, h, samples, model_name), seconds=dt)


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    # debugging aid for single-GPU boxes: APH_BENCH_BACKEND=gloo puts every rank on cuda:0 and reduces through the host
    backend = os.environ.get('APH_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from aphantasia_amd import clip as aclip, transforms
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.clip import LOSS_SCALE
    w, h = [int(s) for s in a.size.split('-')]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load(a.model, weights=None, seed=1, max_batch=8)
    S = a.samples
    if a.model == 'ViT-B/16':
        S = int(S * 0.25)                                   # clip_fft.py:125-127
    trf = transforms.normalize()
    if a.transform == 'fast':
        S = int(S * 0.95)                                   # clip_fft.py:167-169
        trf = transforms.transforms_fast
    torch.manual_seed(0)
    np.random.seed(0)
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(dev).contiguous()
    target = torch.randn(1, model.visual.output_dim, generator=torch.Generator().manual_seed(2))
    eng = Engine(params, h, w, model, S, [(target, -1.0)], sim='mix', transform=trf, macro=0.4,
                 rank=rank, world=world, process_group=pg, use_graph=not a.no_graph)

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = eng.global_loss()

    roof = None
    if not a.no_roofline:
        # same steps again with HIP events around every GEMM launch of the ViT (dominant kernel family)
        lib = eng.lib
        h_ = eng.visual.handle
        eng.use_graph = False             # eager launches so that every GEMM gets its event pair
        lib.call('aph_vit_profile', h_.handle, 1)
        for _ in range(min(a.steps, 10)):
            eng.step()
        torch.cuda.synchronize()
        import ctypes
        ms, n = ctypes.c_double(), ctypes.c_longlong()
        flops = ctypes.c_double()
        lib.call('aph_vit_profile_read', h_.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(flops))
        lib.call('aph_vit_profile', h_.handle, 0)
        if n.value > 0:
            achieved = flops.value / (ms.value * 1e-3) / 1e12
            # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (a PMC pass cannot run inside this process);
            # the committed summary of the latest pass is quoted when the workload is the one it was taken on
            traffic, tsrc = None, None
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic_*.json')))
            if pm and (w, h, a.model, a.samples, a.transform) == (1280, 720, 'ViT-B/32', 200, 'fast') and world == 1:
                with open(pm[-1]) as f:
                    traffic = json.load(f)['traffic_bytes_per_launch']
                tsrc = os.path.relpath(pm[-1], ROOT)
            roof = dict(bound='mfma', kernel='aph::gemm_f16_kernel<*> / aph::gemm8_f16_kernel<*>', achieved=achieved, peak=2500.0,
                        unit='TFLOP/s', frac=achieved / 2500.0, traffic=traffic, traffic_unit='bytes/launch (2*FETCH_SIZE + WRITE_SIZE)',
                        traffic_source=tsrc, launches_per_step=n.value // min(a.steps, 10),
                        avg_launch_us=ms.value * 1e3 / n.value, flops_per_launch=flops.value / n.value,
                        peak_measured_random_operands=1800.0)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(w, h, a.model, S)

    if rank == 0:
        steps_per_s = a.steps / dt
        out = {
            'metric': 'optimization steps/sec @%dx%d %s samples=%d' % (w, h, a.model, a.samples),
            'value': steps_per_s, 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f16 (MFMA operands, fp32 accumulate; fp32 synth/sampler/loss/Adam)', 'data': 'synthetic',
            'config': {'workload': '%dx%d FFT parameteriser, %s, --samples %d -> %d effective cuts, -tf %s, sim mix, '
                                   'Adam(lr .05, b1 0), per-step image save off' % (w, h, a.model, a.samples, S, a.transform),
                       'samples_effective': S, 'parallelism': 'samples split over %d rank(s), 1 all-reduce/step' % world,
                       'loss_scale': LOSS_SCALE, 'final_loss': loss,
                       'algorithmic_tflop_per_step': 2 * S * F_IMG[a.model] / 1e12},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
