// Experiment (not product): how fast can a CU fill LDS through global_load_lds_dwordx4, vs. through VGPRs + ds_write_b128,
// as a function of how many CUs are active?  Pattern = the GEMM's A/B tile loads (8 rows x 128 B per instruction).
#include "../../aphantasia_amd/csrc/aph_device.h"
using namespace aph;

template <int MODE>   // 0: LDS-DMA, 1: global_load -> ds_write_b128
__global__ __launch_bounds__(512) void dma_kernel(const half_t* __restrict__ A, int lda, int rows, int nk, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane >> 3, pc = lane & 7;
  const half_t* g[6];
  for (int k = 0; k < 6; ++k) {
    const int row = ((blockIdx.x * 48 + wave * 6 + k) * 8 + lrow) % rows;
    g[k] = A + (size_t)row * lda + pc * 8;
  }
  float acc = 0.f;
  for (int kt = 0; kt < nk; ++kt) {
    half_t* st = lds + (kt % 3) * (384 * 64);
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) glds16(g[k] + kt * 64, st + (wave * 6 + k) * 8 * 64);
      if (kt >= 2) wait_vm_barrier<12>(); 
    } else {
      half8 v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const half8*>(g[k] + kt * 64);
#pragma unroll
      for (int k = 0; k < 6; ++k) *reinterpret_cast<half8*>(st + (wave * 6 + k) * 8 * 64 + lane * 8) = v[k];
      __syncthreads();
    }
  }
  wait_vm_barrier<0>();
  acc += (float)lds[threadIdx.x];
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

extern "C" int dma_rate(int mode, const void* A, int lda, int rows, int nk, int blocks, float* out, void* stream) {
  const int smem = 3 * 384 * 64 * 2;
  if (mode == 0) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(dma_kernel<0>, dim3(blocks), dim3(512), smem, (hipStream_t)stream, (const half_t*)A, lda, rows, nk, out); }
  else { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(dma_kernel<1>, dim3(blocks), dim3(512), smem, (hipStream_t)stream, (const half_t*)A, lda, rows, nk, out); }
  return (int)hipGetLastError();
}
