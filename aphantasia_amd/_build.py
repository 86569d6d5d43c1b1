"""Builds libaphantasia_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, 'libaphantasia_hip.so')
SOURCES = ['api.hip', 'synth.hip', 'dwt.hip', 'sampler.hip', 'loss_adam.hip', 'vit.hip', 'comm.hip', 'depthwarp.hip']


MARK = os.path.join(HERE, 'build', '.experiments')      # present: the library on disk is an A/B build (-DAPH_EXPERIMENTS), never to be shipped


def _stale():
    if not os.path.isfile(LIB) or os.path.isfile(MARK):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, 'include', 'aphantasia_hip.h'), os.path.join(ROOT, 'include', 'aphantasia_hip_test.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, experiments=False):
    """experiments: -DAPH_EXPERIMENTS (the measured-and-not-adopted kernels and their hooks: A/B runs only -- never the library that is shipped or benchmarked)"""
    if not force and not experiments and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DAPH_EXPERIMENTS'] if experiments else []) + ['-I', CSRC, '-I', os.path.join(ROOT, 'include'),
               '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
        if verbose and out.strip():
            print(out.decode(errors='replace'))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    if experiments:
        open(MARK, 'w').write('libaphantasia_hip.so was built with -DAPH_EXPERIMENTS\n')
    elif os.path.isfile(MARK):
        os.remove(MARK)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv or '--experiments' in sys.argv, experiments='--experiments' in sys.argv)
    print(LIB)
