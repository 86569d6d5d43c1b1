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
// [r6] the TRANSPOSED images (V^T, K^T, Q^T, dO^T: [64 d][64 slots]) swizzle by row >> 2 instead: since the output products read their rows in
// the order d = 4 c16 + nt (whole output rows per store, at_fwd_tiles), sixteen lanes hit rows 4 apart -- (row >> 1) & 7 would take only four
// values there (4-way bank conflicts: the blocked forward went from 46 to 66 us), (row >> 2) & 7 takes eight (2-way, what consecutive rows
// had under at_off)
__device__ __forceinline__ int at_off_t(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 7)) << 3); }

// Stage one 64x64 tile (T valid rows of 64 halfs, row stride ld in global) into LDS as a swizzled row-major image
// and/or as its transposed slot-permuted image.  64 work items cover the tile: item = (d-chunk c, slot-chunk sc),
// sc fastest.  An item loads the 8 rows whose slots form chunk sc (16 bytes each, rows >= T read as zero),
// transposes the 8x8 block in registers and writes 8 + 8 full 16-byte chunks -- no scattered 2-byte LDS stores, no
// separate zero fill, and the 8 lanes of a store group hit 8 different 16-byte slots of one 128-byte row.
__device__ __forceinline__ void at_stage_load(const half_t* __restrict__ src, int ld, int T, int item, half8 (&rows)[8]) {
  const int c = item >> 3, sc = item & 7;
  const int nbase = ((sc >> 2) << 5) + ((sc & 3) << 2);          // rows n(sc, i) = nbase + (i >> 2) * 16 + (i & 3)
  // branch-free: rows >= T read the last valid row (T >= 1) and are zeroed by a select -- every lane issues all eight loads, so the
  // compiler's vmcnt bookkeeping stays exact and no exec-mask branch sits between the loads
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = nbase + ((i >> 2) << 4) + (i & 3);
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const half8 v = *reinterpret_cast<const half8*>(src + (size_t)(n < T ? n : T - 1) * ld + c * 8);
    rows[i] = n < T ? v : z;
  }
}
__device__ __forceinline__ void at_stage_store(const half8 (&rows)[8], int item, half_t* rowmajor, half_t* transposed, int row_limit = 64,
                                               bool tswz = true) {      // tswz: the transposed image uses at_off_t (false: at_off, for readers that take its rows in natural order)
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
      *reinterpret_cast<half8*>(transposed + (tswz ? at_off_t(c * 8 + e, sc) : at_off(c * 8 + e, sc))) = col;
    }
  }
}
__device__ __forceinline__ void at_stage_item(const half_t* __restrict__ src, int ld, int T, int item, half_t* rowmajor,
                                              half_t* transposed, int row_limit = 64, bool tswz = true) {
  half8 rows[8];
  at_stage_load(src, ld, T, item, rows);
  at_stage_store(rows, item, rowmajor, transposed, row_limit, tswz);
}

__device__ __forceinline__ half8 at_frag(const half_t* tile, int row, int chunk) {
  return *reinterpret_cast<const half8*>(tile + at_off(row, chunk));
}
__device__ __forceinline__ half8 at_frag_t(const half_t* tile, int row, int chunk) {      // fragment of a transposed image (at_off_t)
  return *reinterpret_cast<const half8*>(tile + at_off_t(row, chunk));
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

// ---- softmax arithmetic, trimmed [r4] ---------------------------------------------------------------------------------------------
// At T = 197 a lane holds 52 scores of its query and the blocked kernels are bound by the VALU work on them, not by memory or the matrix
// pipe (persistent prefetching changed nothing: note further down).  Per score the round-3 code spent ~15 issue slots: scale, compare +
// select (mask), max, subtract, multiply by log2 e, exp, compare + select again, add, convert.  Now:
//   * the 1/8 scale and log2 e are ONE constant folded into the exponent's fma:  p = 2^(s * kSL - m * kSL)  (max taken on the raw scores);
//   * only the LAST key tile of a row can hold keys >= T: the mask is applied there behind a wave-uniform branch, every full tile skips it,
//     and a masked score of -1e30 needs no second select after the exponential (2^-huge = 0).
constexpr float kSL = 0.125f * 1.4426950408889634f;      // scores / sqrt(64), in log2 units
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float fast_exp2(float x) {
#ifdef APH_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
// raw scores of one 16-key tile starting at key t16 (this lane: keys t16 + g4 + r, r = 0..3): keys >= T -> -1e30 (a tile that reaches T
// only: the test is wave-uniform), running max
__device__ __forceinline__ void at_mask_max(f32x4& st, int t16, int g4, int T, float& mx) {
  if (t16 + 15 >= T) {
#pragma unroll
    for (int r = 0; r < 4; ++r) st[r] = t16 + g4 + r < T ? st[r] : -1e30f;
  }
  mx = fmaxf(fmaxf(mx, fmaxf(st[0], st[1])), fmaxf(st[2], st[3]));
}
// Backward kernels: p = exp(s / 8 - L) recomputed from the raw scores, L2 = L * log2 e.  NO mask: the operand images are zero beyond row
// T (staging zero-fills), so a key / query >= T contributes through a zero K^T / Q^T / dO^T row whatever p is -- as long as p and dS stay
// finite in f16, which the clamp of the exponent at 0 guarantees (p <= 1; for a valid pair s / 8 <= L holds anyway, up to rounding).
// One v_min instead of a compare + select per score, and no branch between the MFMAs (a wave-uniform "last tile only" branch inside the
// unrolled product loops cost more than the selects it saved: 151 against 144 us at C4's shape).
__device__ __forceinline__ float at_p(float s_raw, float L2) { return fast_exp2(fminf(fmaf(s_raw, kSL, -L2), 0.f)); }
// dS of a key / query >= T is -p D_i / 8 with p up to 1 (no mask above): finite in fp32, but with a large loss scale it can leave the f16
// range, and inf x 0 against the zero operand rows would be NaN where the masked form gave an exact 0.  One v_med3 keeps every dS inside
// the f16 range before the conversion (a valid dS that large has overflowed anyway: the guarded Adam step skips the step).
__device__ __forceinline__ float at_ds(float p, float dp, float Di) {
  const float v = (p * 0.125f) * (dp - Di);
#ifdef APH_EMU
  return fminf(fmaxf(v, -65504.f), 65504.f);
#else
  return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
#endif
}

// One (cut, head) attention forward from LDS tiles: Qs, Ks row-major [64][64] (at_off), Vt the transposed slot-permuted image of V.
// Wave `it` owns query tile it (16 rows); att_row0 = the head's 64 columns of the cut's first row in att [.., D]; lse_row0 likewise.
__device__ __forceinline__ void at_fwd_tiles(const half_t* Qs, const half_t* Ks, const half_t* Vt, int T, int it, int lane, half_t* att_row0,
                                             int D, float* lse_row0) {
  const int c16 = lane & 15, g = lane >> 4;
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
  for (int jt = 0; jt < 4; ++jt) at_mask_max(st[jt], jt * 16, g * 4, T, mx);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mxs = mx * kSL;
  float l = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = fast_exp2(fmaf(st[jt][r], kSL, -mxs));       // (masked keys: 2^-huge = 0)
      st[jt][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const half8 p0 = pack8(st[0], st[1]), p1 = pack8(st[2], st[3]);
  const float inv = 1.0f / l;
  const int i = it * 16 + c16;
  // [r6] O = P V with the PROBABILITIES as the A fragment (rows = queries) and the V image's rows taken in the order d = 4 c16 + nt as B:
  // lane (c16, g) then holds, for each of its four queries g*4 + r, the four CONSECUTIVE columns 4 c16 .. 4 c16 + 3, and the sixteen lanes
  // c16 = 0..15 hold that row's 64 columns in order -- a store instruction writes four whole 128-byte rows from adjacent lanes (the same
  // output layout the wave-specialised GEMM's epilogue uses).  The form it replaces (V^T as A: one query per lane, 4 of its columns per
  // product) wrote 32 bytes of sixteen different rows per instruction; tools/exp/attn_ablate.py: the backward's partial-line stores cost
  // 13.5 us of its 27.8.  Same products, same k order: the sums are unchanged.
  f32x4 o[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    o[nt] = mfma_16x16x32_f16(p0, at_frag_t(Vt, 4 * c16 + nt, g), o[nt]);
    o[nt] = mfma_16x16x32_f16(p1, at_frag_t(Vt, 4 * c16 + nt, 4 + g), o[nt]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = it * 16 + g * 4 + r;
    const float iq = __shfl(inv, g * 4 + r);          // 1 / l of query q lives in the lanes with c16 = g*4 + r
    if (q < T) store_h4(att_row0 + (size_t)q * D + 4 * c16, o[0][r] * iq, o[1][r] * iq, o[2][r] * iq, o[3][r] * iq);
  }
  if (g == 0 && i < T) lse_row0[i] = mx * 0.125f + __logf(l);
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
  at_fwd_tiles(Qs, Ks, Vt, T, threadIdx.x >> 6, threadIdx.x & 63, att + (size_t)s * T * D + h * 64, D, lse + ((size_t)s * heads + h) * T);
}

// backward: (qkv, lse, datt) -> dqkv [M,3D] f16   (`att` is not read: see D_i below)
// PERSISTENT: the grid is a few workgroups per CU (3 fit its LDS), each walks the (cut, head) items with the grid stride.  The
// operand rows of item i+1 are loaded into registers right after item i's have been written to LDS, so they are in flight
// during item i's matrix products and stores: the kernel is HBM-bound and one item per workgroup left the memory pipe idle
// during every compute phase (3.3 TB/s).
// [r4] The softmax-gradient row term D_i = dO_i . O_i is formed as  sum_j P_ij dP_ij  (O = P V and dP = dO V^T, so the two are the
// same sum) from the probabilities and dP phase A already holds in registers, instead of from a second read of dO plus a read of O:
// 15 MB less HBM traffic per launch at C2 (117 -> 102 MB) and 16 fewer prefetch registers per lane.  The registers are what mattered:
// at 170 VGPRs under a 168-register bound the compiler spilled one address pair, and its reload -- `scratch_load` + `s_waitcnt vmcnt(0)`,
// vmcnt being one in-order counter -- sat right behind the prefetch loads of the next item: every wave waited for its prefetch to LAND
// before starting the products the prefetch was meant to hide behind (found in the ISA; round 2's 35.7 -> 30.7 us was what survived).
// ABL (measurement only, -DAPH_EXPERIMENTS builds, tools/exp/attn_ablate.py -- WRONG results): 1 = no products (the stores write zeros: loads +
// staging + stores, the memory skeleton), 2 = no stores (loads + staging + products), 3 = loads + staging only
template <int RB, int ABL = 0>            // rows kept per row-major LDS tile: 56 (T <= 56: 3 workgroups per CU) or 64
__global__ __launch_bounds__(256, RB == 56 ? 3 : 2) void attn_bwd_mfma_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ datt,
                                                           const float* __restrict__ lse, half_t* __restrict__ dqkv, int T, int heads, int items) {
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
  // prefetch registers of one item
  half8 rows[8];
  float lse_r = 0.f;
  auto fetch = [&](int item) {
    const int s = item / heads, h = item - s * heads;
    const half_t* base = qkv + (size_t)s * T * ld + h * 64;
    const half_t* dob = datt + (size_t)s * T * D + h * 64;
    at_stage_load(which == 3 ? dob : base + which * D, which == 3 ? D : ld, T, sitem, rows);
    lse_r = (which == 0 && sitem < T) ? lse[(unsigned)(s * heads + h) * (unsigned)T + (unsigned)sitem] : 0.f;
  };
  int item = blockIdx.x;
  fetch(item);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  const bool active = w * 16 < T;
  for (;;) {
    const int s = item / heads, h = item - s * heads;
    // wave `which` = Q, K, V, dO: row-major tile `which`; transposed image for Q (0), K (1), dO (2) -- V has none
    at_stage_store(rows, sitem, lds + which * RT, which == 2 ? nullptr : lds + 4 * RT + (which == 3 ? 2 : which) * 4096, RB);
    if (which == 0) Ls[sitem] = lse_r * kLog2e;
    __syncthreads();
    const int next = item + gridDim.x;
    if (next < items) fetch(next);               // in flight during the products and stores below
    half_t* dbase = dqkv + (size_t)s * T * ld + h * 64;
    if (ABL & 1) {          // ablation: the three output tiles as zeros (same store pattern), or nothing at all
      if (active && !(ABL & 2)) {
        const int i = w * 16 + c16;
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            if (i < T) store_h4(dbase + (size_t)i * ld + part * D + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
      }
      if (next >= items) break;
      item = next;
      __syncthreads();
      continue;
    }
    if (active) {
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
      const float Li2 = Ls[i];                     // lse * log2 e
      float Di = 0.f;                              // D_i = sum_j P_ij dP_ij (= dO_i . O_i): this lane's 16 keys, then the 4 key groups g
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[jt][r] = at_p(st[jt][r], Li2);         // (keys >= T: dP = 0 there -- V rows are zero -- so D_i does not see them)
          Di += st[jt][r] * dp[jt][r];
        }
      }
      Di += __shfl_xor(Di, 16);
      Di += __shfl_xor(Di, 32);
      if (g == 0) Ds[i] = Di;                      // for phase B (key tiles need the D of every query)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[jt][r] = at_ds(st[jt][r], dp[jt][r], Di);      // dS^T
      const half8 d0 = pack8(st[0], st[1]), d1 = pack8(st[2], st[3]);
      // [r6] dQ = dS K with dS as the A fragment and K^T's rows in the order d = 4 c16 + nt as B: whole 128-byte rows per store (at_fwd_tiles)
      f32x4 o[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        o[nt] = mfma_16x16x32_f16(d0, at_frag_t(Kt, 4 * c16 + nt, g), o[nt]);
        o[nt] = mfma_16x16x32_f16(d1, at_frag_t(Kt, 4 * c16 + nt, 4 + g), o[nt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = it * 16 + g * 4 + r;
        if (!(ABL & 2) && q < T) store_h4(dbase + (size_t)q * ld + 4 * c16, o[0][r], o[1][r], o[2][r], o[3][r]);
        if ((ABL & 2) && o[0][r] == 1.2345e33f) store_h4(dbase, o[0][r], o[1][r], o[2][r], o[3][r]);        // (keeps the products alive)
      }
    } else if (lane < 16) {
      Ds[w * 16 + lane] = 0.f;                     // query tiles past T (their dO rows are zero)
    }
    __syncthreads();                               // D of every query tile is in LDS
    if (active) {
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
        const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + t * 16 + g * 4);      // (lse * log2 e)
        const f32x4 D4 = *reinterpret_cast<const f32x4*>(Ds + t * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (queries >= T: Q and dO rows are zero, L2 = D = 0 -> p = 1, dS = 0; their dO^T / Q^T rows are zero)
          const float p = at_p(sq[t][r], L4[r]);
          pp[t][r] = p;
          sq[t][r] = at_ds(p, dq[t][r], D4[r]);     // dS
        }
      }
      const half8 p0 = pack8(pp[0], pp[1]), p1 = pack8(pp[2], pp[3]);
      const half8 s0 = pack8(sq[0], sq[1]), s1 = pack8(sq[2], sq[3]);
      // [r6] dV = P^T dO, dK = dS^T Q with P^T / dS^T as the A fragments (rows = keys) and the dO^T / Q^T images' rows in the order d = 4 c16 + nt
      // as B: lane (c16, g) holds columns 4 c16 .. 4 c16 + 3 of keys g*4 + r -- whole rows per store instruction (at_fwd_tiles)
      f32x4 ov[4], ok[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        ov[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        ok[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        ov[nt] = mfma_16x16x32_f16(p0, at_frag_t(Ot, 4 * c16 + nt, g), ov[nt]);
        ov[nt] = mfma_16x16x32_f16(p1, at_frag_t(Ot, 4 * c16 + nt, 4 + g), ov[nt]);
        ok[nt] = mfma_16x16x32_f16(s0, at_frag_t(Qt, 4 * c16 + nt, g), ok[nt]);
        ok[nt] = mfma_16x16x32_f16(s1, at_frag_t(Qt, 4 * c16 + nt, 4 + g), ok[nt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = jt * 16 + g * 4 + r;
        if (!(ABL & 2) && kj < T) {
          store_h4(dbase + (size_t)kj * ld + D + 4 * c16, ok[0][r], ok[1][r], ok[2][r], ok[3][r]);
          store_h4(dbase + (size_t)kj * ld + 2 * D + 4 * c16, ov[0][r], ov[1][r], ov[2][r], ov[3][r]);
        }
        if ((ABL & 2) && ok[0][r] + ov[0][r] == 1.2345e33f) store_h4(dbase, ok[0][r], ok[1][r], ov[2][r], ov[3][r]);
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
// (Round 4, measured and rejected: these three kernels as PERSISTENT workgroups with the next item's operand rows and the next tile round's
// fragments prefetched into registers, like the one-tile backward above -- forward 70.0 us against 65.6, backward 137.9 against 143.9 at C4's
// shape (profiles/r04_attn_bench.txt).  Nothing to hide: at T = 197 a lane holds 52 scores of its query and the softmax arithmetic on them --
// scale, mask, max, subtract, exp, sum, convert: ~15 VALU slots per score, 2.6 k clocks per query tile next to 0.9 k of MFMA -- is what a
// workgroup spends its time on; the static item split only added a 5-against-4.45 rounding.  git show <this commit>~1 has the code.)
template <int NB>
__device__ __forceinline__ void atg_stage(const half_t* __restrict__ src, int ld, int T, half_t* rowmajor, half_t* transposed, int idx, bool tswz = true) {
  const int blk = idx >> 6, item = idx & 63;
  at_stage_item(src + (size_t)blk * 64 * ld, ld, T - blk * 64, item, rowmajor ? rowmajor + blk * 4096 : nullptr,
                transposed ? transposed + blk * 4096 : nullptr, 64, tswz);
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
    else atg_stage<NB>(base + 2 * D, ld, T, nullptr, Vt, idx - NB * 64, false);      // (this kernel reads V^T's rows in natural order: at_off)
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
      if (jt * 16 < T) at_mask_max(st[jt], jt * 16, g * 4, T, mx);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxs = mx * kSL;
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4 * NB; ++jt) {
      if (jt * 16 < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(fmaf(st[jt][r], kSL, -mxs));       // (masked keys: 2^-huge = 0)
          st[jt][r] = p;
          l += p;
        }
      }                                                                 // (tiles past T keep their zero: no MFMA ran on them)
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
    // ([r6] the whole-row store form of at_fwd_tiles -- probabilities as the A fragment, V^T rows in the order 4 c16 + dt -- was measured on this
    //  kernel too: 46 -> 65 us at C4's shape, with either LDS swizzle; the 15 MB of `att` are 13 % of this kernel's traffic and its time is the
    //  softmax arithmetic (see above), so the one-query-per-lane store stays here.)
    if (i < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        store_h4(att + ((size_t)s * T + i) * D + h * 64 + dt * 16 + g * 4, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      if (g == 0) lse[((size_t)s * heads + h) * T + i] = mx * 0.125f + __logf(l);
    }
  }
}

#ifdef APH_EXPERIMENTS       // the round-3 / round-4 backward as two kernels, each recomputing P and dS: superseded by attn_bwd_one_g_kernel below
                             // (131-139 -> 113-115 us at C4's shape, profiles/r05_attn_bwd_one.txt); kept compilable for A/B runs only
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
    if (idx < NB * 64) atg_stage<NB>(base + D, ld, T, Ks, Kt, idx, false);      // (natural row order below: at_off)
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
    const float Li2 = live ? lse[((size_t)s * heads + h) * T + i] * kLog2e : 0.f;
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
          for (int r = 0; r < 4; ++r) st[u][r] = at_ds(at_p(st[u][r], Li2), dp[u][r], Di);      // dS^T
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
    if (idx < NB * 64) atg_stage<NB>(base, ld, T, Qs, Qt, idx, false);      // (natural row order below: at_off)
    else atg_stage<NB>(dob, D, T, Os, Ot, idx - NB * 64, false);
  }
  for (int r = threadIdx.x; r < NB * 64; r += 512) {
    Ls[r] = r < T ? lse[((size_t)s * heads + h) * T + r] * kLog2e : 0.f;
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
            const float p = at_p(sq[u][r], L4[r]);       // L4 = lse * log2 e
            pp[u][r] = p;
            sq[u][r] = at_ds(p, dq[u][r], D4[r]);     // dS
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

#endif  // APH_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------------
// [r5] The blocked backward as ONE kernel: P = exp(S / 8 - lse) and dS are formed ONCE per (query, key).
// The two kernels above each recompute them because the two contractions want them in transposed register layouts: dK / dV contract over
// QUERIES (the probabilities must be a B fragment with lane = key), dQ contracts over KEYS (lane = query) -- and at T = 197 the softmax
// arithmetic on a lane's scores, not the matrix pipe, is what these kernels spend their time on (note above).  Here the key-stationary form
// is kept (wave w owns the key tiles w, w + 8: K / V fragments and the dK / dV accumulators stay in registers for the whole kernel) and dS
// takes the other layout THROUGH LDS: for one 64-query block at a time every wave writes its dS entries, f16, into a [64 queries][T key slots]
// image (the slot order of the K^T image), and after a barrier the eight waves form the block's dQ^T = K^T dS^T tiles from that image and the
// resident K^T -- 2 bytes written and read per score instead of a second exponential, two more MFMAs and a dozen VALU slots.
// The row term D_i = dO_i . O_i is formed while the block's Q / dO rows are staged (threads that are not staging).
// (Measured and rejected: requesting the NEXT block's rows into registers right behind the phase-1 barrier so that they land under the dQ
// phase -- 120.9 against 113.3 us at C4's shape, profiles/r05_attn_bwd_one.txt: the in-order vmcnt ties the dQ stores to those loads.)
// LDS: Q, dO row-major + transposed for ONE 64-query block (32 KiB), K^T (8 NB KiB), dS^T (8 NB KiB), lse / D: 96.5 KiB at NB = 4.
// Two barriers per query block.  Keys / queries past T: as above (zero operand rows, finite p and dS); the dS^T image starts zeroed, so the
// slots of key tiles nobody computes contribute nothing.
template <int NB>
__global__ __launch_bounds__(512) void attn_bwd_one_g_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ att,
                                                            const half_t* __restrict__ datt, const float* __restrict__ lse,
                                                            half_t* __restrict__ dqkv, int T, int heads) {
  APH_DYN_SMEM(smem);
  half_t* Qs = reinterpret_cast<half_t*>(smem);          // current query block, row-major
  half_t* Os = Qs + 4096;                                // dO of the block, row-major
  half_t* Qt = Os + 4096;                                // transposed slot-permuted images of the two
  half_t* Ot = Qt + 4096;
  half_t* Kt = Ot + 4096;                                // K^T, NB tiles of [64 d][64 key slots]
  half_t* dSb = Kt + NB * 4096;                          // dS^T of the block: NB tiles of [64 queries][64 key slots]
  float* Ls = reinterpret_cast<float*>(dSb + NB * 4096);
  float* Ds = Ls + 64;
  const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
  const int D = heads * 64, ld = 3 * D;
  const half_t* base = qkv + (size_t)s * T * ld + h * 64;
  const half_t* dob = datt + (size_t)s * T * D + h * 64;
  const half_t* ob = att + (size_t)s * T * D + h * 64;
  half_t* dbase = dqkv + (size_t)s * T * ld + h * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < NB * 64; idx += 512) atg_stage<NB>(base + D, ld, T, nullptr, Kt, idx);
  for (int i = tid; i < NB * 512; i += 512) reinterpret_cast<half8*>(dSb)[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int NJ = (4 * NB + 7) / 8;                   // key tiles per wave
  half8 kf[NJ][2], vf[NJ][2];
  f32x4 ok[NJ][4], ov[NJ][4];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = (wave + 8 * jj) * 16 + c16;
    const bool live = j < T;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      kf[jj][kd] = ld_frag_global(base + (size_t)(live ? j : 0) * ld + D + kd * 32 + g * 8, live);
      vf[jj][kd] = ld_frag_global(base + (size_t)(live ? j : 0) * ld + 2 * D + kd * 32 + g * 8, live);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { ok[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  for (int blk = 0; blk * 64 < T; ++blk) {
    const int q0 = blk * 64;
    // stage the block's Q / dO rows; the other threads form lse * log2 e and D_i = dO_i . O_i of its rows
    if (tid < 64) at_stage_item(base + (size_t)q0 * ld, ld, T - q0, tid, Qs, Qt);
    else if (tid < 128) at_stage_item(dob + (size_t)q0 * D, D, T - q0, tid - 64, Os, Ot);
    else if (tid < 384) {
      const int t = tid - 128, row = t >> 2, q4 = t & 3, i = q0 + row;
      const bool live = i < T;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const half8 a = ld_frag_global(ob + (size_t)(live ? i : 0) * D + q4 * 16 + c * 8, live);
        const half8 b = ld_frag_global(dob + (size_t)(live ? i : 0) * D + q4 * 16 + c * 8, live);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)b[e];
      }
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      if (q4 == 0) {
        Ds[row] = d;
        Ls[row] = live ? lse[((size_t)s * heads + h) * T + i] * kLog2e : 0.f;
      }
    }
    __syncthreads();                                     // the block's images are complete (and every wave has left the previous block's dQ phase)
    // ---- phase 1: this wave's key tiles against the block's queries -> dK, dV (registers), dS^T (LDS)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int jt = wave + 8 * jj;
      if (jt * 16 < T) {
        const int key = jt * 16 + c16, sl = slot_of(key & 63);
        half_t* dst = dSb + (key >> 6) * 4096 + (sl & 7);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          if (q0 + qb * 32 < T) {
            f32x4 sq[2], dq[2], pp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              sq[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
              const int t16 = qb * 32 + u * 16;
#pragma unroll
              for (int kd = 0; kd < 2; ++kd) {
                sq[u] = mfma_16x16x32_f16(at_frag(Qs, t16 + c16, kd * 4 + g), kf[jj][kd], sq[u]);
                dq[u] = mfma_16x16x32_f16(at_frag(Os, t16 + c16, kd * 4 + g), vf[jj][kd], dq[u]);
              }
              const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + t16 + g * 4);
              const f32x4 D4 = *reinterpret_cast<const f32x4*>(Ds + t16 + g * 4);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float p = at_p(sq[u][r], L4[r]);
                pp[u][r] = p;
                sq[u][r] = at_ds(p, dq[u][r], D4[r]);     // dS
              }
            }
            const half8 pf = pack8(pp[0], pp[1]), sf = pack8(sq[0], sq[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {      // [r6] P^T / dS^T as A (rows = keys), the dO^T / Q^T rows in the order d = 4 c16 + dt as B: whole dK / dV rows per store (at_fwd_tiles)
              ov[jj][dt] = mfma_16x16x32_f16(pf, at_frag_t(Ot, 4 * c16 + dt, qb * 4 + g), ov[jj][dt]);
              ok[jj][dt] = mfma_16x16x32_f16(sf, at_frag_t(Qt, 4 * c16 + dt, qb * 4 + g), ok[jj][dt]);
            }
            // dS of (query qb*32 + u*16 + 4 g + r, this lane's key) -> [query][key slot]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int r = 0; r < 4; ++r) dst[at_off(qb * 32 + u * 16 + 4 * g + r, sl >> 3)] = sf[u * 4 + r];
          }
        }
      }
    }
    __syncthreads();                                     // dS^T of the block is complete
    // ---- phase 2: dQ^T = K^T dS^T of the block: wave -> query tile wave & 3, d tiles 2 (wave >> 2) and + 1
    const int qt = wave & 3, dt0 = 2 * (wave >> 2);
    if (q0 + qt * 16 < T) {
      f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < 2 * NB; ++kb) {
        if (kb * 32 < T) {
          const half8 df = at_frag(dSb + (kb >> 1) * 4096, qt * 16 + c16, (kb & 1) * 4 + g);
#pragma unroll
          for (int e = 0; e < 2; ++e)          // [r6] dS as A (rows = queries), this wave's 32 columns of K^T in the order d = 16 dt0 + 2 c16 + e as B
            o[e] = mfma_16x16x32_f16(df, at_frag_t(Kt + (kb >> 1) * 4096, dt0 * 16 + 2 * c16 + e, (kb & 1) * 4 + g), o[e]);
        }
      }
      // lane (c16, g): columns 16 dt0 + 2 c16, + 1 of queries g*4 + r -- the sixteen lanes of a row write 64 contiguous bytes (a wave owns half of the
      // head's 64 columns; the other half is the wave 4 further on)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + qt * 16 + g * 4 + r;
        if (q < T) {
          const half2 hv = {(half_t)o[0][r], (half_t)o[1][r]};
          *reinterpret_cast<half2*>(dbase + (size_t)q * ld + dt0 * 16 + 2 * c16) = hv;
        }
      }
    }
  }
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = (wave + 8 * jj) * 16 + g * 4 + r;
      if (j < T) {
        store_h4(dbase + (size_t)j * ld + D + 4 * c16, ok[jj][0][r], ok[jj][1][r], ok[jj][2][r], ok[jj][3][r]);
        store_h4(dbase + (size_t)j * ld + 2 * D + 4 * c16, ov[jj][0][r], ov[jj][1][r], ov[jj][2][r], ov[jj][3][r]);
      }
    }
  }
}

}  // namespace aph
