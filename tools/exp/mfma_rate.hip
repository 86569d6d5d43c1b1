// Experiment: pure-MFMA throughput vs number of active workgroups, operand data (constant vs random) and MFMA shape
// (16x16x32 vs 32x32x16), no memory traffic in the loop.  128 accumulator registers per wave in both shapes
// (the phased GEMM's budget), 2 waves per SIMD.
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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
// 32x32x16: NA x 2 accumulator tiles of 16 registers (NA = 4: 128 registers = a 128x64 wave tile)
template <int NA>
__global__ __launch_bounds__(512) void mfma32_only(const half8* __restrict__ src, float* out, int iters) {
  f32x16 acc[NA][2];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  half8 a[NA], b[2];
  for (int i = 0; i < NA; ++i) a[i] = src[(threadIdx.x * 16 + i) & 8191];
  for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x * 16 + 8 + i) & 8191];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  f32x16 s = acc[0][0];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];
}
extern "C" int mfma_rate(int nacc, int blocks, int iters, const void* src, float* out, void* st) {
  if (nacc == 4) hipLaunchKernelGGL(mfma_only<4>, dim3(blocks), dim3(512), 0, (hipStream_t)st, (const half8*)src, out, iters);
  else if (nacc == 8) hipLaunchKernelGGL(mfma_only<8>, dim3(blocks), dim3(512), 0, (hipStream_t)st, (const half8*)src, out, iters);
  else if (nacc == 32) hipLaunchKernelGGL(mfma32_only<4>, dim3(blocks), dim3(512), 0, (hipStream_t)st, (const half8*)src, out, iters);
  else return -1;
  return (int)hipGetLastError();
}
