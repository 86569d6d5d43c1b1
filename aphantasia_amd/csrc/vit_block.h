// Fused transformer-block kernels for short sequences (T <= 64 tokens per cut: CLIP ViT-B/32 has T = 50), gfx950 (round 5).
//
// At a shard of a few dozen cuts a ViT block is latency, not work: seven launches forward (LayerNorm, QKV, attention, out-proj, LayerNorm,
// fc1, fc2) of 5-14 us each for 1-5 us of arithmetic, every one paying a kernel boundary plus a 2-3 us cold start of its first loads
// (profiles/r05_gemm_rs_phase_trace.txt).  A cut's rows never meet another cut's inside a block, so the row-wise operators fold into the
// GEMM that consumes or produces them WITHOUT any cross-workgroup hand-over, as long as a workgroup holds whole rows of the A operand:
//   blk_qkv_attn_kernel   one workgroup per (cut, head): LayerNorm of the cut's rows straight into the resident A block (the arithmetic of
//                         ln_fwd_kernel, same sums in the same order), the head's 192 QKV columns on the A-resident GEMM of vit_gemm_rs.h,
//                         bias, the q / k / v tiles handed to the one-tile attention of vit_attn.h through LDS; writes qkv (saved for the
//                         backward), att and lse in the unfused layouts;
//   blk_ln_gemm_kernel    one workgroup per (64 rows, 256 columns): LayerNorm prologue + A-resident GEMM + any apply8 epilogue (fc1 +
//                         QuickGELU).
//   blk_lnbwd_gemm_kernel the backward counterpart: the LayerNorm input-gradient of the block ABOVE (ln_1) as the prologue of this block's fc2
//                         dgrad GEMM with the QuickGELU-derivative epilogue.
// Forward per block: 4-5 launches instead of 7, backward 6 instead of 7; the saved activations keep their layouts, so the two directions
// switch independently.
#pragma once
#include "vit_gemm_rs.h"
#include "vit_ops.h"
#include "vit_attn.h"

namespace aph {

__device__ __forceinline__ half4 blk_h4(const f32x4& v) { return half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in all of them: quad_perm x2, row_half_mirror, row_mirror -- four
// VALU instructions, no LDS round trip (__shfl_xor compiles to ds_bpermute_b32 here: ~100 clocks per step, six dependent steps per
// wave-wide sum, which made the first version of this prologue a 10 us chain: profiles/r05_kernel_stats_s26_fused_fwd_v1.csv)
template <int CTRL>
__device__ __forceinline__ float blk_dpp(float v) {
#ifdef APH_EMU
  return v;
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float row16_sum(float v) {
#ifdef APH_EMU
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
#else
  v += blk_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += blk_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += blk_dpp<0x141>(v);     // row_half_mirror
  v += blk_dpp<0x140>(v);     // row_mirror
  return v;
#endif
}
// LayerNorm gain / bias into LDS (dst: 2 D floats, gain first; bias may be null): the prologues read them from there -- as loop-invariant
// GLOBAL loads the compiler kept all 2 D / 16 quads of a lane in registers across the row batches (96 VGPRs at D = 768) and spilled
template <int NV>
__device__ __forceinline__ void blk_stage_affine(float* dst, const float* __restrict__ gamma, const float* __restrict__ beta) {
  constexpr int D = 256 * NV;
  for (int i = threadIdx.x * 4; i < D; i += 256 * 4) {
    *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(gamma + i);
    if (beta) *reinterpret_cast<f32x4*>(dst + D + i) = *reinterpret_cast<const f32x4*>(beta + i);
  }
}
// nothing is scheduled across this point (pins the order "all loads first": the scheduler otherwise sinks them to their uses to save registers)
__device__ __forceinline__ void blk_sched_fence() {
#ifndef APH_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// LayerNorm of 64 rows into the resident A block: rows r < nrows of x (row pitch xstride elements), the others as zeros.  Wave w takes the
// rows 16 w .. 16 w + 15, FOUR AT A TIME: a row belongs to the 16 lanes of a DPP row (lane >> 4), a lane holds the elements
// d = 64 i + 4 (lane & 15) + 0..3 of it (16 lanes x 16 bytes = 256 contiguous bytes per row and load).  All of the wave's rows are requested
// before the first is reduced (rows past nrows re-read the last valid one: no load sits behind a condition).  Element d of row r goes to
// k-step d >> 5, chunk (d >> 3) & 3 of the image (rs_swz placement).  gamma / beta: the LDS copies (blk_stage_affine).
template <int NV>
__device__ __forceinline__ void blk_ln_fill(char* a_img, const float* __restrict__ x, size_t xstride, int nrows, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int wave, int lane) {
  constexpr int D = 256 * NV, NE = D / 64;               // f32x4 pieces per lane and row
  const int c16 = lane & 15, sub = lane >> 4;
  // two batches of eight rows per wave: a batch's 2 NE loads per lane fly together (a wave addresses 256 arch VGPRs: sixteen rows at once
  // next to the weight prefetch made the compiler park registers in AGPRs behind vmcnt(0) waits -- one serialised round trip per load,
  // 20 us of prologue: profiles/r05_kernel_stats_s26_fused_v2.csv)
#pragma unroll 1
  for (int b0 = 0; b0 < 4; b0 += 2) {
    f32x4 v[2][NE];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 16 * wave + 4 * (b0 + p) + sub, rr = r < nrows ? r : nrows - 1;
#pragma unroll
      for (int i = 0; i < NE; ++i) v[p][i] = *reinterpret_cast<const f32x4*>(x + (size_t)rr * xstride + i * 64 + c16 * 4);
    }
    blk_sched_fence();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 16 * wave + 4 * (b0 + p) + sub;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i) s += (v[p][i][0] + v[p][i][1]) + (v[p][i][2] + v[p][i][3]);
      const float mean = row16_sum(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c = v[p][i][e] - mean; q += c * c; }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + kLnEps);
      const bool live = r < nrows;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 64 + c16 * 4), b = *reinterpret_cast<const f32x4*>(beta + i * 64 + c16 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = live ? (v[p][i][e] - mean) * rstd * g[e] + b[e] : 0.f;
        const int ks = 2 * i + (c16 >> 3), c = (c16 >> 1) & 3;
        *reinterpret_cast<half4*>(a_img + ks * 4096 + r * 64 + ((c ^ rs_swz(r)) << 4) + (c16 & 1) * 8) = blk_h4(o);
      }
    }
    blk_sched_fence();
  }
}

// ---- LayerNorm + QKV (one head) + attention ------------------------------------------------------------------------------------------
template <int NV>
struct BlkQKV {
  static constexpr int D = 256 * NV, NKS = 8 * NV, NT = 3;
  static constexpr int SMEM = NKS * 4096 + 4 * 2 * NT * 1024 + 2 * D * 4;      // resident A block + per-wave weight images + gain / bias (the attention tiles alias the A block)
};

template <int NV, int PD>
__global__ __launch_bounds__(256) void blk_qkv_attn_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const half_t* __restrict__ w_qkv, const float* __restrict__ b_qkv, half_t* __restrict__ qkv,
                                                           half_t* __restrict__ att, float* __restrict__ lse, int S, int T, int heads) {
  using C = BlkQKV<NV>;
  constexpr int D = C::D;
  APH_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int t = rs_tile_index(), h = t / S, s = t - h * S;          // heads slowest: the run of an XCD holds few heads' weight rows
  const RSLane L(lane);
  const char* Bb = reinterpret_cast<const char*>(w_qkv);
  unsigned woff[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)      // tile nt = section (q, k, v); tile row i = lrow is the head's column 16 wave + i of that section
    woff[nt] = ((unsigned)(nt * D + h * 64 + 16 * wave + L.lrow) * (unsigned)D + L.lpc * 8) * 2u;
  float* aff = reinterpret_cast<float*>(smem + C::NKS * 4096 + 4 * 2 * 3 * 1024);
  blk_stage_affine<NV>(aff, gamma, beta);                            // (requested first: its wait must not cover the weight prefetch)
  blk_sched_fence();
  ARStream<3, PD> W;
  W.template prefetch<C::NKS>(Bb, woff);                             // the first weight k-steps fly during the LayerNorm
  blk_sched_fence();
  __syncthreads();
  blk_ln_fill<NV>(smem, x + (size_t)s * T * D, D, T, aff, aff + D, wave, lane);
  __syncthreads();
  f32x4 acc[4][3];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  W.template run<C::NKS>(acc, smem, smem + C::NKS * 4096 + wave * (2 * 3 * 1024), Bb, woff, L);
  __syncthreads();                                                   // every wave is done with the A block: its space becomes the attention tiles
  half_t* Qs = reinterpret_cast<half_t*>(smem);
  half_t* Ks = Qs + 4096;
  half_t* Vt = Ks + 4096;
  // lane: tokens 16 mt + (lane & 15), head columns c .. c + 3 of section nt
  const int c = 16 * wave + 4 * (lane >> 4);
  f32x4 bias[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(b_qkv + nt * D + h * 64 + c);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int tok = 16 * mt + (lane & 15);
    const int sl = slot_of(tok);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const half4 hv = blk_h4(acc[mt][nt] + bias[nt]);
      if (tok < T) *reinterpret_cast<half4*>(qkv + ((size_t)s * T + tok) * (3 * D) + nt * D + h * 64 + c) = hv;
      if (nt == 0) *reinterpret_cast<half4*>(Qs + at_off(tok, c >> 3) + (c & 7)) = hv;
      else if (nt == 1) *reinterpret_cast<half4*>(Ks + at_off(tok, c >> 3) + (c & 7)) = hv;
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) Vt[at_off_t(c + r, sl >> 3) + (sl & 7)] = hv[r];
      }
    }
  }
  __syncthreads();
  at_fwd_tiles(Qs, Ks, Vt, T, wave, lane, att + (size_t)s * T * D + h * 64, D, lse + ((size_t)s * heads + h) * T);
}

template <int NV, int PD>
inline void launch_blk_qkv_attn(const float* x, const float* gamma, const float* beta, const half_t* w_qkv, const float* b_qkv, half_t* qkv, half_t* att,
                                float* lse, int S, int T, int heads, hipStream_t st) {
  using C = BlkQKV<NV>;
  APH_ALLOW_SMEM((blk_qkv_attn_kernel<NV, PD>), C::SMEM);
  APH_LAUNCH((blk_qkv_attn_kernel<NV, PD>), dim3(S * heads), dim3(256), C::SMEM, st, x, gamma, beta, w_qkv, b_qkv, qkv, att, lse, S, T, heads);
}

// ---- LayerNorm + wide GEMM (K = width) + apply8 epilogue -------------------------------------------------------------------------------
// rows: m0 + r of the compact row space 0 .. M - 1 (what the epilogue sees); x row m lives at x + m * xs * D (xs = T: class rows only)
template <int NV, int PD, class Epi>
__global__ __launch_bounds__(256) void blk_ln_gemm_kernel(const float* __restrict__ x, int xs, int M, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const half_t* __restrict__ Wt, int N, Epi epi) {
  constexpr int D = 256 * NV, NKS = 8 * NV, NT = 4;
  using C = GemmAR<NT>;
  APH_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  int tm, tn;
  ar_tile((M + C::BM - 1) / C::BM, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN + wave * 16 * NT;
  const RSLane L(lane);
  const char* Bb = reinterpret_cast<const char*>(Wt);
  unsigned woff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) woff[nt] = ((unsigned)(n0 + 4 * NT * (L.lrow >> 2) + 4 * nt + (L.lrow & 3)) * (unsigned)D + L.lpc * 8) * 2u;
  float* aff = reinterpret_cast<float*>(smem + C::smem(D));
  blk_stage_affine<NV>(aff, gamma, beta);
  blk_sched_fence();
  ARStream<NT, PD> W;
  W.template prefetch<NKS>(Bb, woff);
  blk_sched_fence();
  const int nrows = M - m0 < C::BM ? M - m0 : C::BM;
  __syncthreads();
  blk_ln_fill<NV>(smem, x + (size_t)m0 * xs * D, (size_t)xs * D, nrows, aff, aff + D, wave, lane);
  __syncthreads();
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  W.template run<NKS>(acc, smem, smem + NKS * 4096 + wave * (2 * C::WIMG), Bb, woff, L);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + 16 * mt + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) epi.apply8(m, n0 + 4 * NT * (lane >> 4) + 8 * j, acc[mt][2 * j], acc[mt][2 * j + 1]);
    }
  }
}

// ---- LayerNorm input-gradient + wide GEMM + apply8 epilogue (backward: ln_1 of the block above in front of this block's fc2 dgrad) ------
// Prologue: g = [res +] rstd (gamma dy - mean(gamma dy) - xhat mean(gamma dy xhat)) for 64 rows (the arithmetic of ln_bwd_kernel; x = the
// LayerNorm's input, statistics recomputed), f16(g) into the resident A block, and -- from the workgroups of column group 0 only -- g in
// fp32 to out32 (a buffer OTHER than res: the other column groups still read res).  res_T > 0: only the rows with row % res_T == 0 have a
// residual (see ln_bwd_kernel).  Rows in four batches of four per wave (x, dy and res of a batch in flight together: 30 registers per row).
template <int NV>
__device__ __forceinline__ void blk_lnbwd_fill(char* a_img, const half_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                               const float* __restrict__ res, float* __restrict__ out32, int m0, int nrows, int res_T, int wave, int lane) {
  constexpr int D = 256 * NV, NE = D / 64;
  const int c16 = lane & 15, sub = lane >> 4;
#pragma unroll 1
  for (int b0 = 0; b0 < 4; ++b0) {
    f32x4 xv[1][NE], rv[1][NE];
    half4 dv[1][NE];
#pragma unroll
    for (int p = 0; p < 1; ++p) {
      const int r = 16 * wave + 4 * (b0 + p) + sub, rr = r < nrows ? r : nrows - 1;
      const size_t row = (size_t)(m0 + rr) * D;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        xv[p][i] = *reinterpret_cast<const f32x4*>(x + row + i * 64 + c16 * 4);
        dv[p][i] = *reinterpret_cast<const half4*>(dy + row + i * 64 + c16 * 4);
        rv[p][i] = *reinterpret_cast<const f32x4*>(res + row + i * 64 + c16 * 4);
      }
    }
    blk_sched_fence();
#pragma unroll
    for (int p = 0; p < 1; ++p) {
      const int r = 16 * wave + 4 * (b0 + p) + sub;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i) s += (xv[p][i][0] + xv[p][i][1]) + (xv[p][i][2] + xv[p][i][3]);
      const float mean = row16_sum(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[p][i][e] -= mean; q += xv[p][i][e] * xv[p][i][e]; }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + kLnEps);
      f32x4 gd[NE];
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 64 + c16 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[p][i][e] *= rstd;                                   // xhat
          gd[i][e] = (float)dv[p][i][e] * g[e];
          sg += gd[i][e];
          sgx += gd[i][e] * xv[p][i][e];
        }
      }
      sg = row16_sum(sg) * (1.0f / D);
      sgx = row16_sum(sgx) * (1.0f / D);
      const bool live = r < nrows;
      const bool has_res = res_T == 0 || (m0 + r) % res_T == 0;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rstd * (gd[i][e] - sg - xv[p][i][e] * sgx);
          if (has_res) o[e] += rv[p][i][e];
          if (!live) o[e] = 0.f;
        }
        if (out32 && live) *reinterpret_cast<f32x4*>(out32 + (size_t)(m0 + r) * D + i * 64 + c16 * 4) = o;
        const int ks = 2 * i + (c16 >> 3), c = (c16 >> 1) & 3;
        *reinterpret_cast<half4*>(a_img + ks * 4096 + r * 64 + ((c ^ rs_swz(r)) << 4) + (c16 & 1) * 8) = blk_h4(o);
      }
    }
  }
}

template <int NV, int PD, class Epi>
__global__ __launch_bounds__(256) void blk_lnbwd_gemm_kernel(const half_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ res, float* __restrict__ out32, int res_T, int M,
                                                             const half_t* __restrict__ Wt, int N, Epi epi) {
  constexpr int D = 256 * NV, NKS = 8 * NV, NT = 4;
  using C = GemmAR<NT>;
  APH_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  int tm, tn;
  ar_tile((M + C::BM - 1) / C::BM, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN + wave * 16 * NT;
  const RSLane L(lane);
  const char* Bb = reinterpret_cast<const char*>(Wt);
  unsigned woff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) woff[nt] = ((unsigned)(n0 + 4 * NT * (L.lrow >> 2) + 4 * nt + (L.lrow & 3)) * (unsigned)D + L.lpc * 8) * 2u;
  float* aff = reinterpret_cast<float*>(smem + C::smem(D));
  blk_stage_affine<NV>(aff, gamma, nullptr);
  blk_sched_fence();
  ARStream<NT, PD> W;
  W.template prefetch<NKS>(Bb, woff);
  blk_sched_fence();
  const int nrows = M - m0 < C::BM ? M - m0 : C::BM;
  __syncthreads();
  blk_lnbwd_fill<NV>(smem, dy, x, aff, res, tn == 0 ? out32 : nullptr, m0, nrows, res_T, wave, lane);
  __syncthreads();
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  W.template run<NKS>(acc, smem, smem + NKS * 4096 + wave * (2 * C::WIMG), Bb, woff, L);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + 16 * mt + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) epi.apply8(m, n0 + 4 * NT * (lane >> 4) + 8 * j, acc[mt][2 * j], acc[mt][2 * j + 1]);
    }
  }
}

template <int NV, int PD, class Epi>
inline void launch_blk_lnbwd_gemm(const half_t* dy, const float* x, const float* gamma, const float* res, float* out32, int res_T, int M, const half_t* Wt,
                                  int N, Epi epi, hipStream_t st) {
  using C = GemmAR<4>;
  const int smem = C::smem(256 * NV) + 2 * 256 * NV * 4;
  APH_ALLOW_SMEM((blk_lnbwd_gemm_kernel<NV, PD, Epi>), smem);
  APH_LAUNCH((blk_lnbwd_gemm_kernel<NV, PD, Epi>), dim3((N / C::BN) * ((M + C::BM - 1) / C::BM)), dim3(256), smem, st, dy, x, gamma, res, out32, res_T, M,
             Wt, N, epi);
}

template <int NV, int PD, class Epi>
inline void launch_blk_ln_gemm(const float* x, int xs, int M, const float* gamma, const float* beta, const half_t* Wt, int N, Epi epi, hipStream_t st) {
  using C = GemmAR<4>;
  const int smem = C::smem(256 * NV) + 2 * 256 * NV * 4;
  APH_ALLOW_SMEM((blk_ln_gemm_kernel<NV, PD, Epi>), smem);
  APH_LAUNCH((blk_ln_gemm_kernel<NV, PD, Epi>), dim3((N / C::BN) * ((M + C::BM - 1) / C::BM)), dim3(256), smem, st, x, xs, M, gamma, beta, Wt, N, epi);
}

}  // namespace aph
