// Experiment (not product): per-CU load rates on gfx950 for the access patterns a small-M GEMM can stream its operands with.
//   mode 0: global_load_lds_dwordx4, 8 rows x 128 B per instruction (the ring kernels' pattern)
//   mode 1: global_load_lds_dwordx4, 16 rows x 64 B per instruction (vit_gemm_rs.h)
//   mode 2: global_load_dwordx4 -> VGPR, 1 KiB contiguous per instruction (fragment-major packed weights)
//   mode 3: global_load_dwordx4 -> VGPR, MFMA fragment pattern straight from a row-major matrix (16 rows x 4 x 16 B)
//   mode 4: global_load_dwordx4 -> VGPR, 8 rows x 128 B per instruction
// U = instructions in flight per wave; every wave streams `iters` KiB from its workgroup's region (regions are shared by the workgroups with
// the same blockIdx % G), row pitch `pitch` bytes for the row patterns.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE, int U>
__global__ __launch_bounds__(1024) void rate_kernel(const char* __restrict__ src, size_t region, int G, int pitch, int iters, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (size_t)(blockIdx.x % G) * region;
  // the region is a [rows][pitch] matrix; a wave's i-th KiB: rows / columns by pattern
  // region = a power of two; the row pitch is fixed at 2048 B so that all index arithmetic is shifts and masks (no address-VALU bottleneck)
  constexpr unsigned PITCH = 2048;
  (void)pitch;
  const unsigned rmask = (unsigned)region - 1u;
  u32x4 acc = {0u, 0u, 0u, 0u};
  const unsigned lane_off = MODE == 2 ? lane * 16 : (MODE == 0 || MODE == 4) ? (lane >> 3) * PITCH + (lane & 7) * 16
                          : MODE == 1 ? (lane >> 2) * PITCH + (lane & 3) * 16 : (lane & 15) * PITCH + (lane >> 4) * 16;
  const unsigned t0 = (unsigned)wave * (unsigned)iters;
  auto addr = [&](int i) -> const char* {
    const unsigned t = t0 + (unsigned)i;          // this wave's i-th KiB, wrapped into the region
    unsigned o;
    if (MODE == 2) o = t * 1024u;
    else if (MODE == 0 || MODE == 4) o = (t >> 4) * (8u * PITCH) + (t & 15u) * 128u;      // 8 rows x 128 B: 16 tiles per row block
    else o = (t >> 5) * (16u * PITCH) + (t & 31u) * 64u;                                  // 16 rows x 64 B: 32 tiles per row block
    return base + ((o & rmask) + lane_off);
  };
  if (MODE <= 1) {
    char* ring = smem + wave * (U * 1024);
    for (int i = 0; i < U && i < iters; ++i) glds16(addr(i), ring + i * 1024);
    int slot = 0;
    for (int i = U; i < iters; ++i) {
      wait_vm<U - 1>();
      acc[0] += *reinterpret_cast<const unsigned*>(ring + slot * 1024 + lane * 16);
      glds16(addr(i), ring + slot * 1024);
      slot = slot == U - 1 ? 0 : slot + 1;
    }
    wait_vm<0>();
    acc[1] += *reinterpret_cast<const unsigned*>(ring + lane * 16);
  } else {
    u32x4 v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = *reinterpret_cast<const u32x4*>(addr(i));
    for (int i0 = U; i0 + U <= iters; i0 += U) {
#pragma unroll
      for (int d = 0; d < U; ++d) {
        acc ^= v[d];
        v[d] = *reinterpret_cast<const u32x4*>(addr(i0 + d));
      }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) acc ^= v[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}

template <int MODE, int U>
static int launch(const void* src, size_t region, int G, int pitch, int iters, int blocks, int threads, unsigned* out, hipStream_t st) {
  const int smem = MODE <= 1 ? (threads / 64) * U * 1024 : 0;
  if (smem > 160 * 1024) return -2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<MODE, U>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL((rate_kernel<MODE, U>), dim3(blocks), dim3(threads), smem, st, (const char*)src, region, G, pitch, iters, out);
  return (int)hipGetLastError();
}

extern "C" int load_rate(int mode, int U, const void* src, size_t region, int G, int pitch, int iters, int blocks, int threads, unsigned* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define CASE(M, UU) if (mode == M && U == UU) return launch<M, UU>(src, region, G, pitch, iters, blocks, threads, out, st);
  CASE(0, 4) CASE(0, 8) CASE(0, 16) CASE(1, 4) CASE(1, 8) CASE(1, 16)
  CASE(2, 4) CASE(2, 8) CASE(2, 16) CASE(3, 4) CASE(3, 8) CASE(3, 16) CASE(4, 4) CASE(4, 8) CASE(4, 16)
  return -1;
}
