// Device-side common definitions for the gfx950 kernels.
// Product builds: hipcc --offload-arch=gfx950 (this header pulls in the HIP runtime).
// The only other consumer is the test-only SIMT interpreter under tests/emu (APH_EMU),
// which executes these same kernel sources on the build container's CPU for logic checks.
#pragma once

#ifdef APH_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define APH_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
#define APH_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// kernels that ask for more than 64 KiB of dynamic LDS must opt in once per device: checked, remembered per (kernel, device)
// (api.hip); throws std::runtime_error on failure, which the C ABI's APH_CATCH turns into an error code
void aph_allow_smem(const void* kernel, int bytes);
#define APH_ALLOW_SMEM(kern, bytes) aph_allow_smem(reinterpret_cast<const void*>(kern), (int)(bytes))
#endif

#include <stdint.h>

namespace aph {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// ---- wave / block reductions -------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    T o = __shfl_xor(v, m);
    v = o > v ? o : v;
  }
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64).  `red` = LDS scratch of >= 16 T.
// Every thread gets the total.  Deterministic (fixed tree).
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // protect `red` reuse across consecutive calls
  if (lane == 0) red[w] = v;
  __syncthreads();
  T t = 0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// ---- matrix cores --------------------------------------------------------------
// v_mfma_f32_16x16x32_f16:  D[16x16] += A[16x32] * B[32x16]
//   A operand: lane l holds A[i = l & 15][k = (l >> 4) * 8 + j], j = 0..7
//   B operand: lane l holds B[k = (l >> 4) * 8 + j][n = l & 15]
//   C/D:       lane l, reg r holds D[i = (l >> 4) * 4 + r][n = l & 15]
__device__ __forceinline__ f32x4 mfma_16x16x32_f16(half8 a, half8 b, f32x4 c) {
#ifdef APH_EMU
  struct Slot { half8 a, b; };
  const int lane = emu::lane_id();
  Slot s{a, b};
  memcpy(emu::wave_slot(lane), &s, sizeof(s));
  emu::wave_barrier();
  const int n = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      Slot sa, sb;
      memcpy(&sa, emu::wave_slot(i + 16 * (k / 8)), sizeof(Slot));
      memcpy(&sb, emu::wave_slot(n + 16 * (k / 8)), sizeof(Slot));
      acc += (float)sa.a[k % 8] * (float)sb.b[k % 8];
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_32x32x16_f16:  D[32x32] += A[32x16] * B[16x32]   (half the instructions of 16x16x32 for the same flops;
// the same LDS bytes per flop at equal wave tile, a higher sustained rate -- MI355X_MICROARCH.md, matrix cores)
//   A operand: lane l holds A[i = l & 31][k = (l >> 5) * 8 + j], j = 0..7
//   B operand: lane l holds B[k = (l >> 5) * 8 + j][n = l & 31]
//   C/D:       lane l, reg r holds D[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][n = l & 31]
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(half8 a, half8 b, f32x16 c) {
#ifdef APH_EMU
  struct Slot { half8 a, b; };
  const int lane = emu::lane_id();
  Slot s{a, b};
  memcpy(emu::wave_slot(lane), &s, sizeof(s));
  emu::wave_barrier();
  const int n = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      Slot sa, sb;
      memcpy(&sa, emu::wave_slot(i + 32 * (k / 8)), sizeof(Slot));
      memcpy(&sb, emu::wave_slot(n + 32 * (k / 8)), sizeof(Slot));
      acc += (float)sa.a[k % 8] * (float)sb.b[k % 8];
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// global_load_lds_dwordx4: each lane copies 16 bytes from its own global address straight into LDS at
// (wave-uniform base) + lane * 16 -- asynchronous, tracked by vmcnt, no VGPR staging.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef APH_EMU
  memcpy(static_cast<char*>(lds_wave_base) + emu::lane_id() * 16, gsrc, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// "at most N vector-memory ops of this wave outstanding, then workgroup barrier" without the full drain that
// __syncthreads() implies.  The "memory" clobber keeps the compiler's LDS/global accesses on their side.
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
#ifdef APH_EMU
  __syncthreads();
#else
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// Lanes of one wave exchanging data through LDS: the hardware runs a wave's LDS instructions in order, so no
// s_barrier is needed -- this only pins the compiler's order (and lets the host interpreter, whose lanes are
// separate fibers, rendezvous).
__device__ __forceinline__ void wave_lds_fence() {
#ifdef APH_EMU
  emu::wave_barrier();
#else
  __builtin_amdgcn_wave_barrier();
#endif
}

// a value that is the same in every lane of the wave (e.g. the wave index), moved to a scalar register
__device__ __forceinline__ int wave_uniform(int v) {
#ifdef APH_EMU
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

// all of this wave's LDS reads have returned
__device__ __forceinline__ void wait_lgkm0() {
#ifndef APH_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// v_dot2_f32_f16: c + a.x*b.x + a.y*b.y, fp32 accumulate
__device__ __forceinline__ float dot2_f16(half2 a, half2 b, float c) {
#ifdef APH_EMU
  return c + (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1];
#else
  return __builtin_amdgcn_fdot2(a, b, c, false);
#endif
}

// 1 / x by v_rcp_f32 (1 ulp)
__device__ __forceinline__ float fast_rcp(float x) {
#ifdef APH_EMU
  return 1.0f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Zero-fill as a KERNEL (16 bytes per lane, grid-stride).  The step's launch sequence is captured into a hipGraph; with
// hipMemsetAsync the graph held memset nodes, and replays of such a graph next to eager work on the legacy default stream
// were measured to corrupt the backward pass on this runtime (tools/exp/frame_debug3.py).  A graph of kernel nodes only
// does not.  n16 = number of 16-byte units (buffers here are 256-byte aligned and sized).
static __global__ void zero16_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static __global__ void zero4_kernel(int* __restrict__ p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 0; }
static inline void zero_fill_async(void* p, size_t bytes, hipStream_t st) {       // bytes % 16 == 0
  const size_t n16 = bytes / 16;
  unsigned grid = (unsigned)((n16 + 255) / 256);
  grid = grid > 2048u ? 2048u : (grid ? grid : 1u);
  APH_LAUNCH(zero16_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n16);
}

}  // namespace aph
