"""One GEMM shape, a few launches (for rocprofv3 --pmc runs)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
M, N, K = [int(v) for v in sys.argv[1:4]]
A = torch.randn(M, K, device='cuda').half()
B = torch.randn(N, K, device='cuda').half()
C = torch.empty(M, N, device='cuda')
for _ in range(5):
    L.call('aph_gemm_f16', ptr(A), ptr(B), M, N, K, ptr(C), _stream(A))
torch.cuda.synchronize()
