"""TEST INFRASTRUCTURE: builds tests/emu/build/libaphantasia_emu.so -- the product's kernel sources
compiled for the host against the SIMT interpreter in hip_emu.h (see that file).  CPU tests only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'aphantasia_amd', 'csrc')
OUT = os.path.join(HERE, 'build', 'libaphantasia_emu.so')
SOURCES = ['api.hip', 'synth.hip', 'dwt.hip', 'sampler.hip', 'loss_adam.hip', 'vit.hip', 'comm.hip', 'depthwarp.hip']
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, 'hip_emu.h'),
                                                                  os.path.join(ROOT, 'include', 'aphantasia_hip.h')]
    if os.path.isfile(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src + '.emu.o')
        cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-DAPH_EMU', '-DAPH_EXPERIMENTS', '-Wno-unused-value', '-I', HERE, '-I', CSRC,
               '-I', os.path.join(ROOT, 'include'), '-c', os.path.join(CSRC, src), '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError('emu build failed on %s:\n%s' % (src, out.decode(errors='replace')))
    subprocess.check_call([CLANG, '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build())
