// Matrix-core attention for short sequences (T <= 64 tokens: CLIP ViT-B/32 has T = 50), head dim 64.
// One workgroup (4 waves) per (image, head); wave w owns the 16-row tile w.  fp16 operands, fp32
// accumulation and softmax statistics.
//
// Layout trick (cdna_hip_programming.md T12, "make the reduction axis lane-local"): scores are formed
// TRANSPOSED, S^T = K Q^T, so that in the MFMA C/D layout every lane owns ONE query (column l & 15)
// and 4 consecutive keys per 16-key tile; the row softmax is then lane-local (+2 shuffles), and the
// exponentiated tile pairs are ALREADY a valid B-operand fragment for the next product if the other
// operand is stored with its reduction index permuted the same way:
//     slot(n) = (n >> 5) * 32 + ((n >> 2) & 3) * 8 + ((n >> 4) & 1) * 4 + (n & 3)
// so V / K / Q / dO are staged into LDS once as transposed, slot-permuted images [d][slot] and no
// probability matrix ever moves between lanes or through LDS.
#pragma once
#include "aph_device.h"
#include "vit_gemm.h"

namespace aph {

constexpr int AT_T = 64;                       // padded sequence length
constexpr int AT_RB = 56;                      // row-major LDS tiles of the one-tile backward keep 56 rows when T <= 56 (ViT-B/32: T = 50)
__device__ __forceinline__ int slot_of(int n) { return ((n >> 5) << 5) + (((n >> 2) & 3) << 3) + (((n >> 4) & 1) << 2) + (n & 3); }

// [64 rows][64 halfs] tile, 16-byte chunks XOR-swizzled exactly like the GEMM tiles (conflict-free b128 reads)
__device__ __forceinline__ int at_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// Stage one 64x64 tile (T valid rows of 64 halfs, row stride ld in global) into LDS as a swizzled row-major image
// and/or as its transposed slot-permuted image.  64 work items cover the tile: item = (d-chunk c, slot-chunk sc),
// sc fastest.  An item loads the 8 rows whose slots form chunk sc (16 bytes each, rows >= T read as zero),
// transposes the 8x8 block in registers and writes 8 + 8 full 16-byte chunks -- no scattered 2-byte LDS stores, no
// separate zero fill, and the 8 lanes of a store group hit 8 different 16-byte slots of one 128-byte row.
__device__ __forceinline__ void at_stage_load(const half_t* __restrict__ src, int ld, int T, int item, half8 (&rows)[8]) {
  const int c = item >> 3, sc = item & 7;
  const int nbase = ((sc >> 2) << 5) + ((sc & 3) << 2);          // rows n(sc, i) = nbase + (i >> 2) * 16 + (i & 3)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = nbase + ((i >> 2) << 4) + (i & 3);
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    rows[i] = n < T ? *reinterpret_cast<const half8*>(src + (size_t)n * ld + c * 8) : z;
  }
}
__device__ __forceinline__ void at_stage_store(const half8 (&rows)[8], int item, half_t* rowmajor, half_t* transposed, int row_limit = 64) {
  const int c = item >> 3, sc = item & 7;
  const int nbase = ((sc >> 2) << 5) + ((sc & 3) << 2);
  if (rowmajor) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = nbase + ((i >> 2) << 4) + (i & 3);
      if (n < row_limit) *reinterpret_cast<half8*>(rowmajor + at_off(n, c)) = rows[i];
    }
  }
  if (transposed) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      half8 col;
#pragma unroll
      for (int i = 0; i < 8; ++i) col[i] = rows[i][e];
      *reinterpret_cast<half8*>(transposed + at_off(c * 8 + e, sc)) = col;
    }
  }
}
__device__ __forceinline__ void at_stage_item(const half_t* __restrict__ src, int ld, int T, int item, half_t* rowmajor,
                                              half_t* transposed, int row_limit = 64) {
  half8 rows[8];
  at_stage_load(src, ld, T, item, rows);
  at_stage_store(rows, item, rowmajor, transposed, row_limit);
}

__device__ __forceinline__ half8 at_frag(const half_t* tile, int row, int chunk) {
  return *reinterpret_cast<const half8*>(tile + at_off(row, chunk));
}
// fragment of a row-major tile that only holds rows < limit (the rest are zero by construction)
__device__ __forceinline__ half8 at_frag_rows(const half_t* tile, int row, int chunk, int limit) {
  const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return row < limit ? *reinterpret_cast<const half8*>(tile + at_off(row, chunk)) : z;
}

__device__ __forceinline__ half8 pack8(const f32x4& a, const f32x4& b) {
  half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  return h;
}

// qkv [M,3D] f16 -> att [M,D] f16, lse [S*heads*T] f32 (log-sum-exp of the scaled scores)
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ att,
                                                           float* __restrict__ lse, int T, int heads) {
  __shared__ __attribute__((aligned(16))) half_t lds[3 * 64 * 64];
  half_t* Qs = lds;
  half_t* Ks = lds + 64 * 64;
  half_t* Vt = lds + 2 * 64 * 64;
  const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
  const int D = heads * 64, ld = 3 * D;
  const half_t* base = qkv + (size_t)s * T * ld + h * 64;
  {
    const int which = threadIdx.x >> 6, item = threadIdx.x & 63;
    if (which == 0) at_stage_item(base, ld, T, item, Qs, nullptr);
    else if (which == 1) at_stage_item(base + D, ld, T, item, Ks, nullptr);
    else if (which == 2) at_stage_item(base + 2 * D, ld, T, item, nullptr, Vt);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, it = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  if (it * 16 >= T) return;
  f32x4 st[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kd = 0; kd < 2; ++kd) {
    const half8 qf = at_frag(Qs, it * 16 + c16, kd * 4 + g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) st[jt] = mfma_16x16x32_f16(at_frag(Ks, jt * 16 + c16, kd * 4 + g), qf, st[jt]);
  }
  // lane: query i = it*16 + c16, keys j = jt*16 + g*4 + r
  float mx = -1e30f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = jt * 16 + g * 4 + r < T;
      st[jt][r] = ok ? st[jt][r] * 0.125f : -1e30f;
      mx = fmaxf(mx, st[jt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = jt * 16 + g * 4 + r < T ? __expf(st[jt][r] - mx) : 0.f;
      st[jt][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const half8 p0 = pack8(st[0], st[1]), p1 = pack8(st[2], st[3]);
  const float inv = 1.0f / l;
  const int i = it * 16 + c16;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = mfma_16x16x32_f16(at_frag(Vt, dt * 16 + c16, g), p0, o);
    o = mfma_16x16x32_f16(at_frag(Vt, dt * 16 + c16, 4 + g), p1, o);
    if (i < T) store_h4(att + ((size_t)s * T + i) * D + h * 64 + dt * 16 + g * 4, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  }
  if (g == 0 && i < T) lse[((size_t)s * heads + h) * T + i] = mx + __logf(l);
}

// backward: (qkv, att, lse, datt) -> dqkv [M,3D] f16
// PERSISTENT: the grid is a few workgroups per CU (3 fit its LDS), each walks the (cut, head) items with the grid stride.  The
// operand rows of item i+1 are loaded into registers right after item i's have been written to LDS, so they are in flight
// during item i's matrix products and stores: the kernel is HBM-bound (117 MB per launch at C2) and one item per workgroup
// left the memory pipe idle during every compute phase (3.3 TB/s).
template <int RB>            // rows kept per row-major LDS tile: 56 (T <= 56: 3 workgroups per CU) or 64
__global__ __launch_bounds__(256, RB == 56 ? 3 : 2) void attn_bwd_mfma_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ att,
                                                           const half_t* __restrict__ datt, const float* __restrict__ lse,
                                                           half_t* __restrict__ dqkv, int T, int heads, int items) {
  constexpr int RT = RB * 64;                    // halfs per row-major tile
  __shared__ __attribute__((aligned(16))) half_t lds[4 * RT + 3 * 4096 + 256];
  half_t* Qs = lds;
  half_t* Ks = lds + 1 * RT;
  half_t* Vs = lds + 2 * RT;
  half_t* Os = lds + 3 * RT;      // dO
  half_t* Qt = lds + 4 * RT;
  half_t* Kt = Qt + 4096;
  half_t* Ot = Kt + 4096;         // dO transposed
  float* Ls = reinterpret_cast<float*>(Ot + 4096);
  float* Ds = Ls + 64;
  const int D = heads * 64, ld = 3 * D;
  const int which = wave_uniform(threadIdx.x >> 6), sitem = threadIdx.x & 63;       // staging role: wave `which` stages Q / K / V / dO
  const int dr = threadIdx.x >> 2, dpart = threadIdx.x & 3;           // D_i = dO_i . O_i : 4 threads per row
  // prefetch registers of one item
  half8 rows[8], ox[2], oy[2];
  float lse_r = 0.f;
  auto fetch = [&](int item) {
    const int s = item / heads, h = item - s * heads;
    const half_t* base = qkv + (size_t)s * T * ld + h * 64;
    const half_t* dob = datt + (size_t)s * T * D + h * 64;
    const half_t* ob = att + (size_t)s * T * D + h * 64;
    at_stage_load(which == 3 ? dob : base + which * D, which == 3 ? D : ld, T, sitem, rows);
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ox[c] = dr < T ? *reinterpret_cast<const half8*>(ob + (size_t)dr * D + dpart * 16 + c * 8) : z;
      oy[c] = dr < T ? *reinterpret_cast<const half8*>(dob + (size_t)dr * D + dpart * 16 + c * 8) : z;
    }
    lse_r = (dpart == 0 && dr < T) ? lse[((size_t)s * heads + h) * T + dr] : 0.f;
  };
  int item = blockIdx.x;
  fetch(item);
  for (;;) {
    const int s = item / heads, h = item - s * heads;
    // wave `which` = Q, K, V, dO: row-major tile `which`; transposed image for Q (0), K (1), dO (2) -- V has none
    at_stage_store(rows, sitem, lds + which * RT, which == 2 ? nullptr : lds + 4 * RT + (which == 3 ? 2 : which) * 4096, RB);
    {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)ox[c][e] * (float)oy[c][e];
      a += __shfl_xor(a, 1);
      a += __shfl_xor(a, 2);
      if (dpart == 0) {
        Ds[dr] = a;
        Ls[dr] = lse_r;
      }
    }
    __syncthreads();
    const int next = item + gridDim.x;
    if (next < items) fetch(next);               // in flight during the products and stores below
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  half_t* dbase = dqkv + (size_t)s * T * ld + h * 64;
  if (w * 16 < T) {
    // ---- phase A: wave = query tile.  S^T = K Q^T, dP^T = V dO^T (lane: query i, keys jt*16 + g*4 + r)
    const int it = w, i = it * 16 + c16;
    f32x4 st[4], dp[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { st[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      const half8 qf = at_frag_rows(Qs, i, kd * 4 + g, RB), of = at_frag_rows(Os, i, kd * 4 + g, RB);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        st[jt] = mfma_16x16x32_f16(at_frag_rows(Ks, jt * 16 + c16, kd * 4 + g, RB), qf, st[jt]);
        dp[jt] = mfma_16x16x32_f16(at_frag_rows(Vs, jt * 16 + c16, kd * 4 + g, RB), of, dp[jt]);
      }
    }
    const float Li = Ls[i], Di = Ds[i];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = jt * 16 + g * 4 + r < T ? __expf(st[jt][r] * 0.125f - Li) : 0.f;
        st[jt][r] = p * (dp[jt][r] - Di) * 0.125f;      // dS^T
      }
    const half8 d0 = pack8(st[0], st[1]), d1 = pack8(st[2], st[3]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = mfma_16x16x32_f16(at_frag(Kt, dt * 16 + c16, g), d0, o);
      o = mfma_16x16x32_f16(at_frag(Kt, dt * 16 + c16, 4 + g), d1, o);
      if (i < T) store_h4(dbase + (size_t)i * ld + dt * 16 + g * 4, o[0], o[1], o[2], o[3]);
    }
    // ---- phase B: wave = key tile.  S = Q K^T, dP = dO V^T (lane: key j, queries it*16 + g*4 + r)
    const int jt = w, j = jt * 16 + c16;
    f32x4 sq[4], dq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { sq[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      const half8 kf = at_frag_rows(Ks, j, kd * 4 + g, RB), vf = at_frag_rows(Vs, j, kd * 4 + g, RB);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sq[t] = mfma_16x16x32_f16(at_frag_rows(Qs, t * 16 + c16, kd * 4 + g, RB), kf, sq[t]);
        dq[t] = mfma_16x16x32_f16(at_frag_rows(Os, t * 16 + c16, kd * 4 + g, RB), vf, dq[t]);
      }
    }
    f32x4 pp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + t * 16 + g * 4);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(Ds + t * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = t * 16 + g * 4 + r < T ? __expf(sq[t][r] * 0.125f - L4[r]) : 0.f;
        pp[t][r] = p;
        sq[t][r] = p * (dq[t][r] - D4[r]) * 0.125f;     // dS
      }
    }
    const half8 p0 = pack8(pp[0], pp[1]), p1 = pack8(pp[2], pp[3]);
    const half8 s0 = pack8(sq[0], sq[1]), s1 = pack8(sq[2], sq[3]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
      ov = mfma_16x16x32_f16(at_frag(Ot, dt * 16 + c16, g), p0, ov);
      ov = mfma_16x16x32_f16(at_frag(Ot, dt * 16 + c16, 4 + g), p1, ov);
      ok = mfma_16x16x32_f16(at_frag(Qt, dt * 16 + c16, g), s0, ok);
      ok = mfma_16x16x32_f16(at_frag(Qt, dt * 16 + c16, 4 + g), s1, ok);
      if (j < T) {
        store_h4(dbase + (size_t)j * ld + D + dt * 16 + g * 4, ok[0], ok[1], ok[2], ok[3]);
        store_h4(dbase + (size_t)j * ld + 2 * D + dt * 16 + g * 4, ov[0], ov[1], ov[2], ov[3]);
      }
    }
  }
    if (next >= items) break;
    item = next;
    __syncthreads();                             // every wave is done with this item's LDS image
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Same scheme for 64 < T <= 256 tokens (ViT-B/16: T = 197), NB = ceil(T / 64) blocks of 64 tokens, 8 waves.
// Row-major operand images are [NB*64][64] (at_off on the global row), transposed slot-permuted images are NB tiles
// of [64 d][64 slots]; probabilities / dS of a 32-token block are one B fragment, so the second product streams over
// 32-token blocks.  The forward keeps a query tile's whole score row in registers (<= 16 tiles) -- exact softmax, no
// online rescaling.  The backward is two kernels so that each fits LDS: dQ with K, V, K^T resident (query tiles
// stationary per wave, Q / dO fragments straight from global) and dK/dV with Q, dO, Q^T, dO^T resident (key tiles
// stationary); both recompute P = exp(S/8 - lse) per 32-block from the saved log-sum-exp.
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void atg_stage(const half_t* __restrict__ src, int ld, int T, half_t* rowmajor, half_t* transposed, int idx) {
  const int blk = idx >> 6, item = idx & 63;
  at_stage_item(src + (size_t)blk * 64 * ld, ld, T - blk * 64, item, rowmajor ? rowmajor + blk * 4096 : nullptr,
                transposed ? transposed + blk * 4096 : nullptr);
}

__device__ __forceinline__ half8 ld_frag_global(const half_t* __restrict__ p, bool ok) {
  const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? *reinterpret_cast<const half8*>(p) : z;
}

template <int NB>
__global__ __launch_bounds__(512) void attn_fwd_mfma_g_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ att,
                                                             float* __restrict__ lse, int T, int heads) {
  APH_DYN_SMEM(smem);
  half_t* Ks = reinterpret_cast<half_t*>(smem);
  half_t* Vt = Ks + NB * 4096;
  const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
  const int D = heads * 64, ld = 3 * D;
  const half_t* base = qkv + (size_t)s * T * ld + h * 64;
  for (int idx = threadIdx.x; idx < 2 * NB * 64; idx += 512) {
    if (idx < NB * 64) atg_stage<NB>(base + D, ld, T, Ks, nullptr, idx);
    else atg_stage<NB>(base + 2 * D, ld, T, nullptr, Vt, idx - NB * 64);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  for (int it = wave; it * 16 < T; it += 8) {
    const int i = it * 16 + c16;
    half8 qf[2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) qf[kd] = ld_frag_global(base + (size_t)i * ld + kd * 32 + g * 8, i < T);
    f32x4 st[4 * NB];
#pragma unroll
    for (int jt = 0; jt < 4 * NB; ++jt) {
      st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (jt * 16 < T) {
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) st[jt] = mfma_16x16x32_f16(at_frag(Ks, jt * 16 + c16, kd * 4 + g), qf[kd], st[jt]);
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int jt = 0; jt < 4 * NB; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = jt * 16 + g * 4 + r < T;
        st[jt][r] = ok ? st[jt][r] * 0.125f : -1e30f;
        mx = fmaxf(mx, st[jt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4 * NB; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = jt * 16 + g * 4 + r < T ? __expf(st[jt][r] - mx) : 0.f;
        st[jt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2 * NB; ++kb) {
      if (kb * 32 < T) {
        const half8 pf = pack8(st[2 * kb], st[2 * kb + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_16x16x32_f16(at_frag(Vt + (kb >> 1) * 4096, dt * 16 + c16, (kb & 1) * 4 + g), pf, o[dt]);
      }
    }
    if (i < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        store_h4(att + ((size_t)s * T + i) * D + h * 64 + dt * 16 + g * 4, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      if (g == 0) lse[((size_t)s * heads + h) * T + i] = mx + __logf(l);
    }
  }
}

// dQ (query tiles stationary) + the row dots delta_i = dO_i . O_i, written for the dK/dV kernel
template <int NB>
__global__ __launch_bounds__(512) void attn_bwd_dq_g_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ att,
                                                           const half_t* __restrict__ datt, const float* __restrict__ lse,
                                                           float* __restrict__ delta, half_t* __restrict__ dqkv, int T, int heads) {
  APH_DYN_SMEM(smem);
  half_t* Ks = reinterpret_cast<half_t*>(smem);
  half_t* Vs = Ks + NB * 4096;
  half_t* Kt = Vs + NB * 4096;
  const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
  const int D = heads * 64, ld = 3 * D;
  const half_t* base = qkv + (size_t)s * T * ld + h * 64;
  const half_t* dob = datt + (size_t)s * T * D + h * 64;
  const half_t* ob = att + (size_t)s * T * D + h * 64;
  for (int idx = threadIdx.x; idx < 2 * NB * 64; idx += 512) {
    if (idx < NB * 64) atg_stage<NB>(base + D, ld, T, Ks, Kt, idx);
    else atg_stage<NB>(base + 2 * D, ld, T, Vs, nullptr, idx - NB * 64);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  half_t* dbase = dqkv + (size_t)s * T * ld + h * 64;
  for (int it = wave; it * 16 < T; it += 8) {
    const int i = it * 16 + c16;
    const bool live = i < T;
    half8 qf[2], of[2];
    float Di = 0.f;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      qf[kd] = ld_frag_global(base + (size_t)i * ld + kd * 32 + g * 8, live);
      of[kd] = ld_frag_global(dob + (size_t)i * D + kd * 32 + g * 8, live);
      const half8 ov = ld_frag_global(ob + (size_t)i * D + kd * 32 + g * 8, live);
#pragma unroll
      for (int e = 0; e < 8; ++e) Di += (float)ov[e] * (float)of[kd][e];
    }
    Di += __shfl_xor(Di, 16);
    Di += __shfl_xor(Di, 32);
    const float Li = live ? lse[((size_t)s * heads + h) * T + i] : 0.f;
    if (live && g == 0) delta[((size_t)s * heads + h) * T + i] = Di;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2 * NB; ++kb) {
      if (kb * 32 < T) {
        f32x4 st[2], dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          st[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int jr = (2 * kb + u) * 16 + c16;
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            st[u] = mfma_16x16x32_f16(at_frag(Ks, jr, kd * 4 + g), qf[kd], st[u]);
            dp[u] = mfma_16x16x32_f16(at_frag(Vs, jr, kd * 4 + g), of[kd], dp[u]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = (2 * kb + u) * 16 + g * 4 + r < T ? __expf(st[u][r] * 0.125f - Li) : 0.f;
            st[u][r] = p * (dp[u][r] - Di) * 0.125f;      // dS^T
          }
        }
        const half8 df = pack8(st[0], st[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_16x16x32_f16(at_frag(Kt + (kb >> 1) * 4096, dt * 16 + c16, (kb & 1) * 4 + g), df, o[dt]);
      }
    }
    if (live) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_h4(dbase + (size_t)i * ld + dt * 16 + g * 4, o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
    }
  }
}

// dK, dV (key tiles stationary)
template <int NB>
__global__ __launch_bounds__(512) void attn_bwd_dkv_g_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ datt,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            half_t* __restrict__ dqkv, int T, int heads) {
  APH_DYN_SMEM(smem);
  half_t* Qs = reinterpret_cast<half_t*>(smem);
  half_t* Os = Qs + NB * 4096;
  half_t* Qt = Os + NB * 4096;
  half_t* Ot = Qt + NB * 4096;
  float* Ls = reinterpret_cast<float*>(Ot + NB * 4096);
  float* Ds = Ls + NB * 64;
  const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
  const int D = heads * 64, ld = 3 * D;
  const half_t* base = qkv + (size_t)s * T * ld + h * 64;
  const half_t* dob = datt + (size_t)s * T * D + h * 64;
  for (int idx = threadIdx.x; idx < 2 * NB * 64; idx += 512) {
    if (idx < NB * 64) atg_stage<NB>(base, ld, T, Qs, Qt, idx);
    else atg_stage<NB>(dob, D, T, Os, Ot, idx - NB * 64);
  }
  for (int r = threadIdx.x; r < NB * 64; r += 512) {
    Ls[r] = r < T ? lse[((size_t)s * heads + h) * T + r] : 0.f;
    Ds[r] = r < T ? delta[((size_t)s * heads + h) * T + r] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  half_t* dbase = dqkv + (size_t)s * T * ld + h * 64;
  for (int jt = wave; jt * 16 < T; jt += 8) {
    const int j = jt * 16 + c16;
    const bool live = j < T;
    half8 kf[2], vf[2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      kf[kd] = ld_frag_global(base + (size_t)j * ld + D + kd * 32 + g * 8, live);
      vf[kd] = ld_frag_global(base + (size_t)j * ld + 2 * D + kd * 32 + g * 8, live);
    }
    f32x4 ok[4], ov[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { ok[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qb = 0; qb < 2 * NB; ++qb) {
      if (qb * 32 < T) {
        f32x4 sq[2], dq[2], pp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          sq[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int t16 = (2 * qb + u) * 16;
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            sq[u] = mfma_16x16x32_f16(at_frag(Qs, t16 + c16, kd * 4 + g), kf[kd], sq[u]);
            dq[u] = mfma_16x16x32_f16(at_frag(Os, t16 + c16, kd * 4 + g), vf[kd], dq[u]);
          }
          const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + t16 + g * 4);
          const f32x4 D4 = *reinterpret_cast<const f32x4*>(Ds + t16 + g * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = t16 + g * 4 + r < T ? __expf(sq[u][r] * 0.125f - L4[r]) : 0.f;
            pp[u][r] = p;
            sq[u][r] = p * (dq[u][r] - D4[r]) * 0.125f;     // dS
          }
        }
        const half8 pf = pack8(pp[0], pp[1]), sf = pack8(sq[0], sq[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          ov[dt] = mfma_16x16x32_f16(at_frag(Ot + (qb >> 1) * 4096, dt * 16 + c16, (qb & 1) * 4 + g), pf, ov[dt]);
          ok[dt] = mfma_16x16x32_f16(at_frag(Qt + (qb >> 1) * 4096, dt * 16 + c16, (qb & 1) * 4 + g), sf, ok[dt]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_h4(dbase + (size_t)j * ld + D + dt * 16 + g * 4, ok[dt][0], ok[dt][1], ok[dt][2], ok[dt][3]);
        store_h4(dbase + (size_t)j * ld + 2 * D + dt * 16 + g * 4, ov[dt][0], ov[dt][1], ov[dt][2], ov[dt][3]);
      }
    }
  }
}

}  // namespace aph
