"""Are the GEMM kernels of this checkout the ones an earlier revision compiled to?  Compiles csrc/vit.hip of both trees to gfx950 assembly
(device only) and compares every kernel whose name contains `gemm` instruction for instruction (labels normalised, comments dropped).
Backs `roofline.traffic_match: "gemm_sources"` in bench.py: the PMC traffic summary of profiles/ was taken on the library of <rev>.
    python tools/gemm_isa_diff.py e6595d7"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev = sys.argv[1] if len(sys.argv) > 1 else 'e6595d7'


def asm(tree, out):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(tree, 'aphantasia_amd', 'csrc'), '-I', os.path.join(tree, 'include'),
                           '-S', '--cuda-device-only', '-o', out, os.path.join(tree, 'aphantasia_amd', 'csrc', 'vit.hip')], stderr=subprocess.DEVNULL)
    text = open(out).read()
    kern = {}
    for m in re.finditer(r'^(_ZN3aph\w+):.*?s_endpgm', text, re.S | re.M):
        kern[m.group(1)] = re.sub(r';.*', '', re.sub(r'\.LBB\d+_\d+', 'L', m.group(0)))
    return kern


with tempfile.TemporaryDirectory() as tmp:
    old_tree = os.path.join(tmp, 'old')
    os.makedirs(old_tree)
    subprocess.check_call('git -C %s archive %s aphantasia_amd/csrc include | tar -x -C %s' % (ROOT, rev, old_tree), shell=True)
    a, b = asm(old_tree, os.path.join(tmp, 'a.s')), asm(ROOT, os.path.join(tmp, 'b.s'))
gem = [k for k in a if 'gemm' in k]
diff = [k for k in gem if b.get(k) != a[k]]
print('%d GEMM kernel instantiations at %s: %d identical in this checkout, %d different; %d new GEMM instantiations here'
      % (len(gem), rev, len(gem) - len(diff), len(diff), len([k for k in b if 'gemm' in k and k not in a])))
for k in diff[:10]:
    print('  differs:', k[:140])
sys.exit(1 if diff else 0)
