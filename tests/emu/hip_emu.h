// TEST INFRASTRUCTURE ONLY -- never part of the product build.
//
// A tiny host-side SIMT interpreter used by `pytest -m "not gpu"` to execute the
// *unmodified* HIP kernel sources of aphantasia_amd/csrc on a machine without a GPU
// (the build container has none), so that index arithmetic, LDS staging, barriers and
// wave-collective logic are checked against the oracle before GPU minutes are spent.
// One OS thread; every GPU thread of a workgroup is a ucontext fiber; workgroups run
// one after another.  `__syncthreads()` and the wave collectives are cooperative
// barriers.  MFMA is emulated from the documented gfx950 fragment layouts
// (cdna_hip_programming.md section 3) -- the real layouts are verified on hardware by
// the `-m gpu` tests, this file only mirrors them.
//
// The product library (libaphantasia_hip.so) is built by hipcc from the same sources
// with the real <hip/hip_runtime.h>; nothing under tests/emu is linked into it.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return 0; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; ++r) memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return 0;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }

namespace emu {

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  uint3_emu tid;
};

struct Block {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  int cur = 0;
  int nthreads = 0;
  int alive = 0;
  // block barrier
  int bar_count = 0;
  unsigned bar_gen = 0;
  // wave barriers / exchange scratch
  int wave_count[16] = {0};
  unsigned wave_gen[16] = {0};
  int wave_alive[16] = {0};
  alignas(16) unsigned char wave_scratch[16][64][64];  // per wave, per lane, 64 bytes
  uint3_emu bid, bdim, gdim;
  const std::function<void()>* body = nullptr;
  char* dyn_smem = nullptr;
};

inline Block*& B() { static Block* b = nullptr; return b; }

constexpr size_t kStack = 256 * 1024;

inline void yield() { Block* b = B(); swapcontext(&b->fibers[b->cur].ctx, &b->sched); }

inline void fiber_main() {
  Block* b = B();
  (*b->body)();
  Fiber& f = b->fibers[b->cur];
  f.done = true;
  b->alive--;
  int w = b->cur / 64;
  b->wave_alive[w]--;
  // a thread that exits releases barriers the others may be waiting at
  if (b->alive > 0 && b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; }
  if (b->wave_alive[w] > 0 && b->wave_count[w] == b->wave_alive[w]) { b->wave_count[w] = 0; b->wave_gen[w]++; }
  swapcontext(&f.ctx, &b->sched);
}

inline void block_barrier() {
  Block* b = B();
  unsigned g = b->bar_gen;
  if (++b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; return; }
  while (b->bar_gen == g) yield();
}

inline void wave_barrier() {
  Block* b = B();
  int w = b->cur / 64;
  unsigned g = b->wave_gen[w];
  if (++b->wave_count[w] == b->wave_alive[w]) { b->wave_count[w] = 0; b->wave_gen[w]++; return; }
  while (b->wave_gen[w] == g) yield();
}

inline int lane_id() { return B()->cur & 63; }
inline int wave_id() { return B()->cur >> 6; }
inline unsigned char* wave_slot(int lane) { Block* b = B(); return b->wave_scratch[b->cur >> 6][lane]; }

inline void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  static Block* blk = new Block();
  Block* b = blk;
  B() = b;
  int nt = block.x * block.y * block.z;
  if (nt > 1024 || nt <= 0) { fprintf(stderr, "emu: bad block size %d\n", nt); abort(); }
  if ((int)b->fibers.size() < nt) {
    size_t old = b->fibers.size();
    b->fibers.resize(nt);
    for (size_t i = old; i < (size_t)nt; ++i) b->fibers[i].stack = (char*)malloc(kStack);
  }
  std::vector<char> dyn(smem + 64);
  b->dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
  b->body = &body;
  b->nthreads = nt;
  b->bdim = {block.x, block.y, block.z};
  b->gdim = {grid.x, grid.y, grid.z};
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        b->bid = {bx, by, bz};
        b->alive = nt;
        b->bar_count = 0;
        for (int w = 0; w < 16; ++w) {
          b->wave_count[w] = 0;
          int lo = w * 64, hi = lo + 64;
          b->wave_alive[w] = nt <= lo ? 0 : (nt < hi ? nt - lo : 64);
        }
        for (int t = 0; t < nt; ++t) {
          Fiber& f = b->fibers[t];
          f.done = false;
          f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, (void (*)())fiber_main, 0);
        }
        while (b->alive > 0) {
          for (int t = 0; t < nt; ++t) {
            if (b->fibers[t].done) continue;
            b->cur = t;
            swapcontext(&b->sched, &b->fibers[t].ctx);
          }
        }
      }
}

}  // namespace emu

#define threadIdx (emu::B()->fibers[emu::B()->cur].tid)
#define blockIdx (emu::B()->bid)
#define blockDim (emu::B()->bdim)
#define gridDim (emu::B()->gdim)

inline void __syncthreads() { emu::block_barrier(); }

template <typename T>
inline T __shfl(T v, int src) {
  static_assert(sizeof(T) <= 64, "");
  memcpy(emu::wave_slot(emu::lane_id()), &v, sizeof(T));
  emu::wave_barrier();
  T r;
  memcpy(&r, emu::wave_slot(src & 63), sizeof(T));
  emu::wave_barrier();
  return r;
}
template <typename T> inline T __shfl_xor(T v, int m) { return __shfl(v, emu::lane_id() ^ m); }
template <typename T> inline T __shfl_down(T v, int d) { int l = emu::lane_id(); return __shfl(v, l + d < 64 ? l + d : l); }
template <typename T> inline T __shfl_up(T v, int d) { int l = emu::lane_id(); return __shfl(v, l - d >= 0 ? l - d : l); }

inline unsigned long long __ballot(int pred) {
  unsigned char v = pred ? 1 : 0;
  memcpy(emu::wave_slot(emu::lane_id()), &v, 1);
  emu::wave_barrier();
  unsigned long long m = 0;
  const int base = emu::wave_id() * 64, nt = emu::B()->nthreads;
  for (int l = 0; l < 64 && base + l < nt; ++l) if (*emu::wave_slot(l)) m |= 1ull << l;
  emu::wave_barrier();
  return m;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline void __threadfence() {}
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }

#define APH_LAUNCH(kern, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define APH_DYN_SMEM(name) char* name = emu::B()->dyn_smem
#define APH_ALLOW_SMEM(kern, bytes) ((void)0)
inline float __logf(float x) { return logf(x); }
