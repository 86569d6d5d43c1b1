// Experiment: pure-MFMA throughput vs number of active workgroups and operand data (constant vs random), no memory traffic in the loop.
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void mfma_only(const half8* __restrict__ src, float* out, int iters) {
  f32x4 acc[NACC][4];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  half8 a[NACC], b[4];
  for (int i = 0; i < NACC; ++i) a[i] = src[(threadIdx.x * 16 + i) & 8191];
  for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x * 16 + 8 + i) & 8191];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  f32x4 s = acc[0][0];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];
}
extern "C" int mfma_rate(int nacc, int blocks, int iters, const void* src, float* out, void* st) {
  if (nacc == 4) hipLaunchKernelGGL(mfma_only<4>, dim3(blocks), dim3(512), 0, (hipStream_t)st, (const half8*)src, out, iters);
  else hipLaunchKernelGGL(mfma_only<8>, dim3(blocks), dim3(512), 0, (hipStream_t)st, (const half8*)src, out, iters);
  return (int)hipGetLastError();
}
