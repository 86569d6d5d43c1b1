// Inverse 2-D discrete wavelet transform and its adjoint.
//
// Replaces: pytorch_wavelets DWTInverse(wave, mode='symmetric') as called by aphantasia/image.py:36-38,67
// (dwt_image.inner), i.e. per level  lowlevel.SFB2D:
//     lo = sfb1d(ll, LH, dim H);  hi = sfb1d(HL, HH, dim H);  out = sfb1d(lo, hi, dim W)
//     sfb1d(a, b)[m] = full[m + L - 2],  full[2 i + k] += a[i] rec_lo[k] + b[i] rec_hi[k]     (length 2 n - L + 2)
// The per-level detail gain of dwt_scale (image.py:73-80) is folded in as `hscale`.
// HBM-bound (the finest level of a 4K image moves ~200 MB); each workgroup stages the input patch of its
// output tile in LDS once and every coefficient / pixel is read from HBM exactly once.  One launch per level for the
// levels that fill the chip; the coarse tail (every level whose output is a single tile) runs in ONE launch, a
// workgroup per channel walking the levels (aph_idwt_fwd / aph_idwt_bwd).
#include "aph_device.h"
#include "aph_host.h"

#include <cstdlib>

namespace aph {

constexpr int DW_TY = 32, DW_TX = 128, DW_NT = 256;    // forward: output tile (rows x cols) per workgroup of the per-level kernel
constexpr int DW_IY = 16, DW_IX = 64;                  // adjoint: coefficient tile per workgroup
constexpr int DC_TY = 80, DC_TX = 128, DC_NT = 1024;   // coarse-tail kernels: one tile holds a whole level (out <= 80 x 128)
constexpr int DC_IY = 40, DC_IX = 64;                  //   adjoint: coefficient tile <= 40 x 64
constexpr int DC_MAX_LEVELS = 12;
constexpr int DC_HS_MAX = 12 * DC_NT;                  // detail-band values of all coarse levels together (one prefetch batch: 12 per thread)

// Forward tile, separable in LDS: a TY x TX output tile of one channel by NT threads.
//   1. the four band patches [PY][PX] (PY = TY/2 + L/2 input rows, PX likewise; zero beyond h / w) are loaded once --
//      with a compile-time filter length EVERY load of the patch is issued before the first LDS write (one memory round
//      trip per tile; the row loop this replaces waited for memory once per 64 columns of a row, ten times per wave, and
//      a workgroup's lifetime was those waits: 3.3 TB/s on the finest 4K level);
//   2. horizontal pass, one thread per (patch row, output column PAIR): outputs 2j and 2j+1 read the same L/2 inputs
//      i = j + L/2 - 1 - t with the even / odd filter taps:  rlo = g0 * ll + g1 * HL,  rhi = g0 * LH + g1 * HH;
//   3. vertical pass: out = g0 * rlo + g1 * rhi; four columns of a row pair per thread (16-byte LDS reads and stores)
//      when the output rows are 16-byte aligned, else one column per thread.
// bll: [llh][llw] (only rows < h / cols < w are used -- the "unpad" of DWTInverse.forward), bh: [3][h][w] (LH, HL, HH),
// bout: [Ho][Wo], Ho = 2h-L+2, Wo = 2w-L+2 -- all of ONE channel.  H2T = L/2 at compile time (0: run-time length).
// No __restrict__: the coarse-tail kernel reads what its previous level wrote.
template <int H2T, int TY, int TX, int NT>
__device__ __forceinline__ void idwt_tile(char* smem, const float* bll, int llw, const float* bh, int h, int w, const float* g0,
                                          const float* g1, int L, float hscale, float* bout, int Ho, int Wo, int my0, int mx0) {
  const int H2 = H2T ? H2T : (L >> 1);
  const int PY = TY / 2 + H2, PX = TX / 2 + H2;
  float* f0 = reinterpret_cast<float*>(smem);  // rec_lo
  float* f1 = f0 + 2 * H2;                     // rec_hi
  float* pll = f1 + 2 * H2;                    // 4 band patches [PY][PX]
  float* plh = pll + PY * PX;
  float* phl = plh + PY * PX;
  float* phh = phl + PY * PX;
  float* rlo = phh + PY * PX;                  // [PY][TX]
  float* rhi = rlo + PY * TX;
  const int tid = threadIdx.x;
  // output m = 2 j + p (p = 0, 1) reads inputs i = j + H2 - 1 - t with tap k = p + 2 t;  the tile's first input is m0 / 2
  const int iy0 = my0 / 2, ix0 = mx0 / 2;
  if constexpr (H2T > 0) {
    constexpr int PYc = TY / 2 + H2T, PXc = TX / 2 + H2T, NQ = PYc * PXc, NIT = (NQ + NT - 1) / NT;
    float v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * NT + tid;
      const int py = q / PXc, px = q - py * PXc;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = q < NQ && iy < h && ix < w;
      const size_t o = ok ? (size_t)iy * w + ix : 0, oll = ok ? (size_t)iy * llw + ix : 0;      // (element 0 always exists)
      v[it][0] = bll[oll];
      v[it][1] = bh[o];
      v[it][2] = bh[(size_t)h * w + o];
      v[it][3] = bh[2 * (size_t)h * w + o];
    }
    for (int k = tid; k < 2 * H2; k += NT) { f0[k] = g0[k]; f1[k] = g1[k]; }      // (behind the patch loads: same round trip)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * NT + tid;
      const int py = q / PXc, px = q - py * PXc;
      const bool ok = iy0 + py < h && ix0 + px < w;
      if (q < NQ) {
        pll[q] = ok ? v[it][0] : 0.f;
        plh[q] = ok ? v[it][1] * hscale : 0.f;
        phl[q] = ok ? v[it][2] * hscale : 0.f;
        phh[q] = ok ? v[it][3] * hscale : 0.f;
      }
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < 2 * H2; k += NT) { f0[k] = g0[k]; f1[k] = g1[k]; }
    for (int py = wave; py < PY; py += NT / 64) {
      const int iy = iy0 + py;
      for (int px = lane; px < PX; px += 64) {
        const int ix = ix0 + px;
        const bool ok = iy < h && ix < w;
        const size_t o = (size_t)iy * w + ix;
        const int q = py * PX + px;
        pll[q] = ok ? bll[(size_t)iy * llw + ix] : 0.f;
        plh[q] = ok ? bh[o] * hscale : 0.f;
        phl[q] = ok ? bh[(size_t)h * w + o] * hscale : 0.f;
        phh[q] = ok ? bh[2 * (size_t)h * w + o] * hscale : 0.f;
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < PY * (TX / 2); q += NT) {
    const int py = q / (TX / 2), jl = q - py * (TX / 2);
    const int e0 = py * PX + jl + H2 - 1;
    float lo_e = 0.f, lo_o = 0.f, hi_e = 0.f, hi_o = 0.f;
#pragma unroll
    for (int t = 0; t < (H2T ? H2T : H2); ++t) {
      const float a_e = f0[2 * t], a_o = f0[2 * t + 1], b_e = f1[2 * t], b_o = f1[2 * t + 1];
      const float vll = pll[e0 - t], vlh = plh[e0 - t], vhl = phl[e0 - t], vhh = phh[e0 - t];
      lo_e += a_e * vll + b_e * vhl; lo_o += a_o * vll + b_o * vhl;
      hi_e += a_e * vlh + b_e * vhh; hi_o += a_o * vlh + b_o * vhh;
    }
    *reinterpret_cast<float2*>(rlo + py * TX + 2 * jl) = make_float2(lo_e, lo_o);
    *reinterpret_cast<float2*>(rhi + py * TX + 2 * jl) = make_float2(hi_e, hi_o);
  }
  __syncthreads();
  if (((Wo & 3) | (int)(reinterpret_cast<size_t>(bout) & 15)) == 0) {
    // (the finest level of a 4K image, Wo = 3840, is three quarters of the transform's bytes)
    for (int q = tid; q < (TY / 2) * (TX / 4); q += NT) {
      const int rl = q / (TX / 4), c4 = q - rl * (TX / 4), my = my0 + 2 * rl, mx = mx0 + 4 * c4;
      const int e0 = (rl + H2 - 1) * TX + 4 * c4;
      float4 o_e = make_float4(0.f, 0.f, 0.f, 0.f), o_o = o_e;
#pragma unroll
      for (int t = 0; t < (H2T ? H2T : H2); ++t) {
        const float4 lo = *reinterpret_cast<const float4*>(rlo + e0 - t * TX), hi = *reinterpret_cast<const float4*>(rhi + e0 - t * TX);
        const float a_e = f0[2 * t], a_o = f0[2 * t + 1], b_e = f1[2 * t], b_o = f1[2 * t + 1];
        o_e.x += a_e * lo.x + b_e * hi.x; o_e.y += a_e * lo.y + b_e * hi.y; o_e.z += a_e * lo.z + b_e * hi.z; o_e.w += a_e * lo.w + b_e * hi.w;
        o_o.x += a_o * lo.x + b_o * hi.x; o_o.y += a_o * lo.y + b_o * hi.y; o_o.z += a_o * lo.z + b_o * hi.z; o_o.w += a_o * lo.w + b_o * hi.w;
      }
      if (mx < Wo) {
        if (my < Ho) *reinterpret_cast<float4*>(bout + (size_t)my * Wo + mx) = o_e;
        if (my + 1 < Ho) *reinterpret_cast<float4*>(bout + (size_t)(my + 1) * Wo + mx) = o_o;
      }
    }
  } else {
    for (int q = tid; q < (TY / 2) * TX; q += NT) {
      const int rl = q / TX, tx_ = q - rl * TX, my = my0 + 2 * rl, mx = mx0 + tx_;
      const int e0 = (rl + H2 - 1) * TX + tx_;
      float o_e = 0.f, o_o = 0.f;
#pragma unroll
      for (int t = 0; t < (H2T ? H2T : H2); ++t) {
        const float lo = rlo[e0 - t * TX], hi = rhi[e0 - t * TX];
        o_e += f0[2 * t] * lo + f1[2 * t] * hi;
        o_o += f0[2 * t + 1] * lo + f1[2 * t + 1] * hi;
      }
      if (mx < Wo) {
        if (my < Ho) bout[(size_t)my * Wo + mx] = o_e;
        if (my + 1 < Ho) bout[(size_t)(my + 1) * Wo + mx] = o_o;
      }
    }
  }
}
template <int TY, int TX>
constexpr size_t idwt_tile_smem(int L) { return sizeof(float) * (2 * L + 4 * (TY / 2 + L / 2) * (TX / 2 + L / 2) + 2 * (TY / 2 + L / 2) * TX); }

// ll: [C][llh][llw], highs: [C][3][h][w], out: [C][Ho][Wo]; one DW_TY x DW_TX output tile of one channel per workgroup
template <int H2T>
__global__ __launch_bounds__(DW_NT) void idwt_level_kernel(const float* __restrict__ ll, int llh, int llw,
                                                           const float* __restrict__ highs, int h, int w, const float* __restrict__ g0,
                                                           const float* __restrict__ g1, int L, float hscale, float* __restrict__ out,
                                                           int Ho, int Wo) {
  APH_DYN_SMEM(smem);
  const int c = blockIdx.z;
  idwt_tile<H2T, DW_TY, DW_TX, DW_NT>(smem, ll + (size_t)c * llh * llw, llw, highs + (size_t)c * 3 * h * w, h, w, g0, g1, L, hscale,
                                      out + (size_t)c * Ho * Wo, Ho, Wo, blockIdx.y * DW_TY, blockIdx.x * DW_TX);
}

// Adjoint tile, separable in LDS: bdout [Ho][Wo] -> bdll [llh][llw] (extra unpadded row/col = 0) and bdh [3][h][w]
// (x hscale), one channel.  Coefficient i touches outputs m = 2 i + k - (L - 2), k in [0, L).  Per IY x IX tile:
//   1. dout patch [PY][PX], PY = 2 IY + L - 2 (all loads in flight together, as in idwt_tile);
//   2. horizontal: slo[y][ix] = sum_k g0[k] pd[y][2 ix + k], shi with g1 -> LDS [2][PY][IX];
//   3. vertical:   dll = sum_k g0[k] slo[2 iy + k], dLH = sum g1 slo, dHL = sum g0 shi, dHH = sum g1 shi.
template <int LT, int IY, int IX, int NT>
__device__ __forceinline__ void idwt_adjoint_tile(char* smem, const float* bd, int Ho, int Wo, int h, int w, const float* g0,
                                                  const float* g1, int L_, float hscale, float* bdll, int llh, int llw, float* bdh,
                                                  int iy0, int ix0) {
  const int L = LT ? LT : L_;
  const int PY = 2 * IY + L - 2, PX = 2 * IX + L - 2;
  float* f0 = reinterpret_cast<float*>(smem);
  float* f1 = f0 + L;
  float* pd = f1 + L;                          // [PY][PX]
  float* slo = pd + PY * PX;                   // [PY][IX]
  float* shi = slo + PY * IX;
  const int tid = threadIdx.x;
  const int my0 = 2 * iy0 - (L - 2), mx0 = 2 * ix0 - (L - 2);
  if constexpr (LT > 0) {
    constexpr int PYc = 2 * IY + LT - 2, PXc = 2 * IX + LT - 2, NQ = PYc * PXc, NIT = (NQ + NT - 1) / NT;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * NT + tid;
      const int py = q / PXc, px = q - py * PXc;
      const int my = my0 + py, mx = mx0 + px;
      const bool ok = q < NQ && my >= 0 && my < Ho && mx >= 0 && mx < Wo;
      v[it] = bd[ok ? (size_t)my * Wo + mx : 0];
    }
    for (int k = tid; k < L; k += NT) { f0[k] = g0[k]; f1[k] = g1[k]; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * NT + tid;
      const int py = q / PXc, px = q - py * PXc;
      const int my = my0 + py, mx = mx0 + px;
      if (q < NQ) pd[q] = (my >= 0 && my < Ho && mx >= 0 && mx < Wo) ? v[it] : 0.f;
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < L; k += NT) { f0[k] = g0[k]; f1[k] = g1[k]; }
    for (int py = wave; py < PY; py += NT / 64) {
      const int my = my0 + py;
      const bool rowok = my >= 0 && my < Ho;
      for (int px = lane; px < PX; px += 64) {
        const int mx = mx0 + px;
        pd[py * PX + px] = (rowok && mx >= 0 && mx < Wo) ? bd[(size_t)my * Wo + mx] : 0.f;
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < PY * IX; q += NT) {
    const int py = q / IX, ixl = q - py * IX;
    const float2* row = reinterpret_cast<const float2*>(pd + py * PX + 2 * ixl);      // 8-byte aligned: PX and 2 L are even
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int t = 0; t < (LT ? LT / 2 : L / 2); ++t) {
      const float2 v = row[t];
      a += f0[2 * t] * v.x; a += f0[2 * t + 1] * v.y;
      b += f1[2 * t] * v.x; b += f1[2 * t + 1] * v.y;
    }
    slo[q] = a;
    shi[q] = b;
  }
  __syncthreads();
  for (int q = tid; q < IY * IX; q += NT) {
    const int iyl = q / IX, ixl = q - iyl * IX, iy = iy0 + iyl, ix = ix0 + ixl;
    if (iy >= llh || ix >= llw) continue;
    float* ol = bdll + (size_t)iy * llw + ix;
    if (iy >= h || ix >= w) { *ol = 0.f; continue; }      // the row / column DWTInverse drops
    float all = 0.f, alh = 0.f, ahl = 0.f, ahh = 0.f;
#pragma unroll
    for (int k = 0; k < (LT ? LT : L); ++k) {
      const int e = (2 * iyl + k) * IX + ixl;
      const float lo = slo[e], hi = shi[e];
      all += f0[k] * lo; alh += f1[k] * lo;
      ahl += f0[k] * hi; ahh += f1[k] * hi;
    }
    *ol = all;
    float* oh = bdh + (size_t)iy * w + ix;
    oh[0] = alh * hscale;
    oh[(size_t)h * w] = ahl * hscale;
    oh[2 * (size_t)h * w] = ahh * hscale;
  }
}
template <int IY, int IX>
constexpr size_t idwt_adjoint_tile_smem(int L) { return sizeof(float) * (2 * L + (2 * IY + L - 2) * (2 * IX + L - 2) + 2 * (2 * IY + L - 2) * IX); }

template <int LT>
__global__ __launch_bounds__(DW_NT) void idwt_level_adjoint_kernel(const float* __restrict__ dout, int Ho, int Wo, int h, int w,
                                                                   const float* __restrict__ g0, const float* __restrict__ g1, int L_,
                                                                   float hscale, float* __restrict__ dll, int llh, int llw,
                                                                   float* __restrict__ dhighs) {
  APH_DYN_SMEM(smem);
  const int c = blockIdx.z;
  idwt_adjoint_tile<LT, DW_IY, DW_IX, DW_NT>(smem, dout + (size_t)c * Ho * Wo, Ho, Wo, h, w, g0, g1, L_, hscale, dll + (size_t)c * llh * llw,
                                             llh, llw, dhighs + (size_t)c * 3 * h * w, blockIdx.y * DW_IY, blockIdx.x * DW_IX);
}

// ---- coarse tail: the levels whose bands fit the LDS, one workgroup per channel walking them -----------------------------------
// (eight such levels under a 4K db3 image were eight launches of 3 workgroups each, 6 us apiece of pure latency; a first fused
// version that still went through global memory between levels -- load, fence, barrier -- cost 4.7 us per level, hardly less.)
// Here nothing but the first level's low band, every level's detail bands (all requested up front: one memory round trip for the
// whole tail) and the last level's output touches global memory: the running low band stays in LDS between levels.  Whole levels
// need no zero padding: output pair jl reads inputs jl .. jl + L/2 - 1 <= w - 1, rows likewise.  The sums are those of idwt_tile,
// term for term.
struct IdwtLevel {                // forward: ll, highs -> out.  adjoint: out = incoming gradient, ll / highs = gradients written
  float* ll; float* highs; float* out;
  int llh, llw, h, w;
  float hscale;
  int hoff;                       // forward: offset (floats) of this level's detail bands in the LDS prefetch area
  int hdelta;                     // forward: highs - (the first level's highs), in floats (the host checks that it fits)
};
struct IdwtLevels { int n; IdwtLevel lv[DC_MAX_LEVELS]; };      // in execution order

constexpr int DC_LL = DC_TY * DC_TX;                  // running low band / incoming gradient, floats
constexpr int DC_R = DC_IY * DC_TX;                   // forward: one horizontal-pass plane [h <= DC_IY][Wo <= DC_TX]
inline size_t idwt_coarse_smem(int L, int hsum) { return sizeof(float) * (2 * L + DC_LL + 2 * DC_R + hsum); }
inline size_t idwt_coarse_adjoint_smem(int L) { return sizeof(float) * (2 * L + DC_LL + DC_IY * DC_IX + 2 * (2 * DC_IY + L - 2) * DC_IX); }

template <int H2T>
__global__ __launch_bounds__(DC_NT) void idwt_coarse_kernel(IdwtLevels lv, const float* __restrict__ g0, const float* __restrict__ g1, int L) {
  APH_DYN_SMEM(smem);
  constexpr int H2 = H2T;
  float* f0 = reinterpret_cast<float*>(smem);
  float* f1 = f0 + 2 * H2;
  float* LL = f1 + 2 * H2;             // running low band, row pitch `pitch`
  float* rlo = LL + DC_LL;             // [h][Wo]
  float* rhi = rlo + DC_R;
  float* HS = rhi + DC_R;              // every level's (LH, HL, HH)
  const int c = blockIdx.x, tid = threadIdx.x;
  // everything the tail reads from global memory, requested before the first LDS write: the coarsest low band (<= DC_IY x DC_IX:
  // three values per thread) and the detail bands of all levels as ONE flat index space (HS is laid out in level order, so the
  // level of element q is the last one whose hoff <= q: a chain of selects over the level table, which sits in scalar registers)
  constexpr int NLL = (DC_IY * DC_IX + DC_NT - 1) / DC_NT, NB = DC_HS_MAX / DC_NT;
  float wll[NLL];
  {
    const IdwtLevel& l0 = lv.lv[0];
    const int n0 = l0.llh * l0.llw;
    const float* src = l0.ll + (size_t)c * n0;
#pragma unroll
    for (int u = 0; u < NLL; ++u) { const int q = u * DC_NT + tid; wll[u] = src[q < n0 ? q : 0]; }
  }
  const int hsum = lv.lv[lv.n - 1].hoff + 3 * lv.lv[lv.n - 1].h * lv.lv[lv.n - 1].w;      // <= DC_HS_MAX (idwt_coarse_count)
  {
    // element q of HS belongs to the last level whose hoff <= q; its address is (first level's highs) + q + d_k with
    // d_k = hdelta_k + c * 3 h_k w_k - hoff_k.  The level table is read at compile-time indices (scalar registers, loaded once):
    // per element one compare + select per level.  (A run-time loop over the table cost a dependent scalar load per level and
    // element: 10 us of the kernel's 22.)
    const float* h0 = lv.lv[0].highs;
    int dk[DC_MAX_LEVELS], ho[DC_MAX_LEVELS];       // (wave-uniform: scalar registers; an unused slot never matches)
#pragma unroll
    for (int k = 0; k < DC_MAX_LEVELS; ++k) {
      const bool on = k < lv.n;
      ho[k] = on ? lv.lv[k].hoff : 0x7fffffff;
      dk[k] = on ? lv.lv[k].hdelta + c * 3 * lv.lv[k].h * lv.lv[k].w - lv.lv[k].hoff : 0;
    }
    float v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int q = u * DC_NT + tid, qc = q < hsum ? q : 0;
      int d = dk[0];
#pragma unroll
      for (int k = 1; k < DC_MAX_LEVELS; ++k) d = qc >= ho[k] ? dk[k] : d;
      v[u] = h0[(ptrdiff_t)d + qc];
    }
    for (int k = tid; k < 2 * H2; k += DC_NT) { f0[k] = g0[k]; f1[k] = g1[k]; }
    const int n0 = lv.lv[0].llh * lv.lv[0].llw;
#pragma unroll
    for (int u = 0; u < NLL; ++u) { const int q = u * DC_NT + tid; if (q < n0) LL[q] = wll[u]; }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int q = u * DC_NT + tid;
      if (q < hsum) HS[q] = v[u];       // (unscaled: the level's gain is applied where the value is used)
    }
  }
  __syncthreads();
  int pitch = lv.lv[0].llw;
  for (int i = 0; i < lv.n; ++i) {
    const IdwtLevel& l = lv.lv[i];
    const int h = l.h, w = l.w, Ho = 2 * h - L + 2, Wo = 2 * w - L + 2, W2 = Wo >> 1;
    const float hsc = l.hscale;
    const float* Hlh = HS + l.hoff;
    const float* Hhl = Hlh + h * w;
    const float* Hhh = Hhl + h * w;
    for (int q = tid; q < h * W2; q += DC_NT) {
      const int py = q / W2, jl = q - py * W2;
      const int e0 = jl + H2 - 1;
      const float* rll = LL + py * pitch;
      const float* rlh = Hlh + py * w;
      const float* rhl = Hhl + py * w;
      const float* rhh = Hhh + py * w;
      float lo_e = 0.f, lo_o = 0.f, hi_e = 0.f, hi_o = 0.f;
#pragma unroll
      for (int t = 0; t < H2; ++t) {
        const float a_e = f0[2 * t], a_o = f0[2 * t + 1], b_e = f1[2 * t], b_o = f1[2 * t + 1];
        const float vll = rll[e0 - t], vlh = rlh[e0 - t] * hsc, vhl = rhl[e0 - t] * hsc, vhh = rhh[e0 - t] * hsc;
        lo_e += a_e * vll + b_e * vhl; lo_o += a_o * vll + b_o * vhl;
        hi_e += a_e * vlh + b_e * vhh; hi_o += a_o * vlh + b_o * vhh;
      }
      *reinterpret_cast<float2*>(rlo + py * Wo + 2 * jl) = make_float2(lo_e, lo_o);      // (8-byte aligned: Wo is even)
      *reinterpret_cast<float2*>(rhi + py * Wo + 2 * jl) = make_float2(hi_e, hi_o);
    }
    __syncthreads();
    // vertical pass: the level's output is the next level's low band (row pitch Wo); only the last one leaves the CU
    const bool last = i + 1 == lv.n;
    float* gout = l.out + (size_t)c * Ho * Wo;
    for (int q = tid; q < (Ho >> 1) * Wo; q += DC_NT) {
      const int rl = q / Wo, x = q - rl * Wo;
      const int e0 = (rl + H2 - 1) * Wo + x;
      float o_e = 0.f, o_o = 0.f;
#pragma unroll
      for (int t = 0; t < H2; ++t) {
        const float lo = rlo[e0 - t * Wo], hi = rhi[e0 - t * Wo];
        o_e += f0[2 * t] * lo + f1[2 * t] * hi;
        o_o += f0[2 * t + 1] * lo + f1[2 * t + 1] * hi;
      }
      LL[2 * rl * Wo + x] = o_e;
      LL[(2 * rl + 1) * Wo + x] = o_o;
      if (last) { gout[(size_t)2 * rl * Wo + x] = o_e; gout[(size_t)(2 * rl + 1) * Wo + x] = o_o; }
    }
    __syncthreads();
    pitch = Wo;
  }
}

// adjoint of the coarse tail, finest of its levels first: the incoming gradient is read from global memory once, each level's
// low-band gradient stays in LDS as the next level's input, the detail-band gradients (and the coarsest low band's) are stored.
template <int LT>
__global__ __launch_bounds__(DC_NT) void idwt_coarse_adjoint_kernel(IdwtLevels lv, const float* __restrict__ g0, const float* __restrict__ g1, int L_) {
  APH_DYN_SMEM(smem);
  constexpr int L = LT;
  float* f0 = reinterpret_cast<float*>(smem);
  float* f1 = f0 + L;
  float* DA = f1 + L;                  // the level's incoming gradient [Ho][Wo]
  float* DB = DA + DC_LL;              // its low-band gradient [llh][llw] = the next level's incoming gradient (the two swap)
  float* slo = DB + DC_IY * DC_IX;     // [PY = 2 llh + L - 2][llw]
  float* shi = slo + (2 * DC_IY + L - 2) * DC_IX;
  const int c = blockIdx.x, tid = threadIdx.x;
  {
    // the tail's only read from global memory: the first level's incoming gradient (<= DC_TY x DC_TX), all loads in flight together
    constexpr int NA = (DC_LL + DC_NT - 1) / DC_NT;
    const IdwtLevel& l0 = lv.lv[0];
    const int n0 = (2 * l0.h - L + 2) * (2 * l0.w - L + 2);
    const float* src = l0.out + (size_t)c * n0;
    float v[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int q = u * DC_NT + tid; v[u] = src[q < n0 ? q : 0]; }
    for (int k = tid; k < L; k += DC_NT) { f0[k] = g0[k]; f1[k] = g1[k]; }
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int q = u * DC_NT + tid; if (q < n0) DA[q] = v[u]; }
  }
  __syncthreads();
  for (int i = 0; i < lv.n; ++i) {
    const IdwtLevel& l = lv.lv[i];
    const int h = l.h, w = l.w, llh = l.llh, llw = l.llw, Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
    const int PY = 2 * llh + L - 2;
    for (int q = tid; q < PY * llw; q += DC_NT) {
      const int py = q / llw, ixl = q - py * llw;
      const int my = py - (L - 2), mxb = 2 * ixl - (L - 2);
      const bool rowok = my >= 0 && my < Ho;
      const float* row = DA + (rowok ? my : 0) * Wo;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < L / 2; ++t) {
        const int mx = mxb + 2 * t;
        const float vx = (rowok && mx >= 0 && mx < Wo) ? row[mx] : 0.f;
        const float vy = (rowok && mx + 1 >= 0 && mx + 1 < Wo) ? row[mx + 1] : 0.f;
        a += f0[2 * t] * vx; a += f0[2 * t + 1] * vy;
        b += f1[2 * t] * vx; b += f1[2 * t + 1] * vy;
      }
      slo[q] = a;
      shi[q] = b;
    }
    __syncthreads();
    const bool last = i + 1 == lv.n;
    float* gll = l.ll + (size_t)c * llh * llw;
    float* gh = l.highs + (size_t)c * 3 * h * w;
    for (int q = tid; q < llh * llw; q += DC_NT) {
      const int iyl = q / llw, ixl = q - iyl * llw;
      if (iyl >= h || ixl >= w) {       // the row / column DWTInverse drops
        DB[q] = 0.f;
        if (last) gll[q] = 0.f;
        continue;
      }
      float all = 0.f, alh = 0.f, ahl = 0.f, ahh = 0.f;
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const int e = (2 * iyl + k) * llw + ixl;
        const float lo = slo[e], hi = shi[e];
        all += f0[k] * lo; alh += f1[k] * lo;
        ahl += f0[k] * hi; ahh += f1[k] * hi;
      }
      DB[q] = all;
      if (last) gll[q] = all;
      float* oh = gh + (size_t)iyl * w + ixl;
      oh[0] = alh * l.hscale;
      oh[(size_t)h * w] = ahl * l.hscale;
      oh[2 * (size_t)h * w] = ahh * l.hscale;
    }
    __syncthreads();
    float* t_ = DA; DA = DB; DB = t_;
  }
}

template <int H2T>
void launch_idwt_fwd(dim3 grid, size_t smem, hipStream_t st, const float* d_ll, int ll_h, int ll_w, const float* d_highs, int h, int w,
                     const float* d_g0, const float* d_g1, int L, float hscale, float* d_out, int Ho, int Wo) {
  APH_ALLOW_SMEM(idwt_level_kernel<H2T>, 150 * 1024);
  APH_LAUNCH(idwt_level_kernel<H2T>, grid, dim3(DW_NT), smem, st, d_ll, ll_h, ll_w, d_highs, h, w, d_g0, d_g1, L, hscale, d_out, Ho, Wo);
}
template <int LT>
void launch_idwt_bwd(dim3 grid, size_t smem, hipStream_t st, const float* d_out_grad, int Ho, int Wo, int h, int w, const float* d_g0,
                     const float* d_g1, int L, float hscale, float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad) {
  APH_ALLOW_SMEM(idwt_level_adjoint_kernel<LT>, 150 * 1024);
  APH_LAUNCH(idwt_level_adjoint_kernel<LT>, grid, dim3(DW_NT), smem, st, d_out_grad, Ho, Wo, h, w, d_g0, d_g1, L, hscale, d_ll_grad, ll_h,
             ll_w, d_highs_grad);
}
template <int H2T>
void launch_idwt_coarse(const IdwtLevels& lv, int C, size_t smem, hipStream_t st, const float* d_g0, const float* d_g1, int L) {
  APH_ALLOW_SMEM(idwt_coarse_kernel<H2T>, 150 * 1024);
  APH_LAUNCH(idwt_coarse_kernel<H2T>, dim3(C), dim3(DC_NT), smem, st, lv, d_g0, d_g1, L);
}
template <int LT>
void launch_idwt_coarse_adjoint(const IdwtLevels& lv, int C, size_t smem, hipStream_t st, const float* d_g0, const float* d_g1, int L) {
  APH_ALLOW_SMEM(idwt_coarse_adjoint_kernel<LT>, 150 * 1024);
  APH_LAUNCH(idwt_coarse_adjoint_kernel<LT>, dim3(C), dim3(DC_NT), smem, st, lv, d_g0, d_g1, L);
}

// filter lengths the coarse-tail kernels are instantiated for (fully unrolled filter loops)
inline bool idwt_coarse_length(int L) { return L == 2 || L == 4 || L == 6 || L == 8; }
constexpr size_t kCoarseSmemMax = 150 * 1024;
// (-DAPH_EXPERIMENTS builds only: APH_IDWT_COARSE=0 in the environment = one launch per level throughout, for A/B runs)
inline bool idwt_coarse_enabled() {
#ifdef APH_EXPERIMENTS
  static const bool on = [] { const char* e = getenv("APH_IDWT_COARSE"); return !(e && e[0] == '0'); }();
  return on;
#else
  return true;
#endif
}

void idwt_level_fwd(const float* d_ll, int ll_h, int ll_w, const float* d_highs, int h, int w, int C, const float* d_g0,
                    const float* d_g1, int L, float hscale, float* d_out, hipStream_t st) {
  const int Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
  const size_t smem = idwt_tile_smem<DW_TY, DW_TX>(L);
  const dim3 grid((Wo + DW_TX - 1) / DW_TX, (Ho + DW_TY - 1) / DW_TY, C);
#define APH_IDWT_FWD(N) launch_idwt_fwd<N>(grid, smem, st, d_ll, ll_h, ll_w, d_highs, h, w, d_g0, d_g1, L, hscale, d_out, Ho, Wo)
  switch (L) {            // common orthogonal wavelets get fully unrolled filter loops
    case 2: APH_IDWT_FWD(1); break;
    case 4: APH_IDWT_FWD(2); break;
    case 6: APH_IDWT_FWD(3); break;
    case 8: APH_IDWT_FWD(4); break;
    case 12: APH_IDWT_FWD(6); break;
    default: APH_IDWT_FWD(0); break;
  }
#undef APH_IDWT_FWD
}
void idwt_level_bwd(const float* d_out_grad, int h, int w, int C, const float* d_g0, const float* d_g1, int L, float hscale,
                    float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad, hipStream_t st) {
  const int Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
  const size_t smem = idwt_adjoint_tile_smem<DW_IY, DW_IX>(L);
  const dim3 grid((ll_w + DW_IX - 1) / DW_IX, (ll_h + DW_IY - 1) / DW_IY, C);
#define APH_IDWT_BWD(N) launch_idwt_bwd<N>(grid, smem, st, d_out_grad, Ho, Wo, h, w, d_g0, d_g1, L, hscale, d_ll_grad, ll_h, ll_w, d_highs_grad)
  switch (L) {
    case 2: APH_IDWT_BWD(2); break;
    case 4: APH_IDWT_BWD(4); break;
    case 6: APH_IDWT_BWD(6); break;
    case 8: APH_IDWT_BWD(8); break;
    case 12: APH_IDWT_BWD(12); break;
    default: APH_IDWT_BWD(0); break;
  }
#undef APH_IDWT_BWD
}
void idwt_coarse_fwd(const IdwtLevels& lv, int C, const float* d_g0, const float* d_g1, int L, hipStream_t st) {
  const IdwtLevel& lastl = lv.lv[lv.n - 1];
  const size_t smem = idwt_coarse_smem(L, lastl.hoff + 3 * lastl.h * lastl.w);
  switch (L) {
    case 2: launch_idwt_coarse<1>(lv, C, smem, st, d_g0, d_g1, L); break;
    case 4: launch_idwt_coarse<2>(lv, C, smem, st, d_g0, d_g1, L); break;
    case 6: launch_idwt_coarse<3>(lv, C, smem, st, d_g0, d_g1, L); break;
    default: launch_idwt_coarse<4>(lv, C, smem, st, d_g0, d_g1, L); break;      // 8 (idwt_coarse_length)
  }
}
void idwt_coarse_bwd(const IdwtLevels& lv, int C, const float* d_g0, const float* d_g1, int L, hipStream_t st) {
  const size_t smem = idwt_coarse_adjoint_smem(L);
  switch (L) {
    case 2: launch_idwt_coarse_adjoint<2>(lv, C, smem, st, d_g0, d_g1, L); break;
    case 4: launch_idwt_coarse_adjoint<4>(lv, C, smem, st, d_g0, d_g1, L); break;
    case 6: launch_idwt_coarse_adjoint<6>(lv, C, smem, st, d_g0, d_g1, L); break;
    default: launch_idwt_coarse_adjoint<8>(lv, C, smem, st, d_g0, d_g1, L); break;
  }
}

// size of the running low band that enters level j: the coarsest band itself, else the output of level j + 1
inline void idwt_ll_size(const int* hs, const int* ws, int J, int L, int j, int* llh, int* llw) {
  if (j + 1 < J) { *llh = 2 * hs[j + 1] - L + 2; *llw = 2 * ws[j + 1] - L + 2; }
  else { *llh = hs[j]; *llw = ws[j]; }
}
int idwt_check_levels(const char* who, const int* hs, const int* ws, int J, int C, int L) {
  if (!hs || !ws || J < 1 || C < 1 || L < 2 || L > 64 || (L & 1)) return aph_fail(APH_ERR_ARG, "%s: bad argument (J=%d C=%d L=%d)", who, J, C, L);
  for (int j = 0; j < J; ++j) {
    int llh, llw;
    idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
    if (hs[j] < 1 || ws[j] < 1 || 2 * hs[j] - L + 2 < 1 || 2 * ws[j] - L + 2 < 1 || llh < hs[j] || llw < ws[j] || llh > hs[j] + 1 || llw > ws[j] + 1)
      return aph_fail(APH_ERR_ARG, "%s: level %d is %dx%d but its low band would be %dx%d (filter length %d)", who, j, hs[j], ws[j], llh, llw, L);
  }
  return APH_OK;
}
// number of levels, counted from the coarsest, that the coarse-tail kernel takes (0: none -- it needs at least two to pay)
inline int idwt_coarse_count(const int* hs, const int* ws, int J, int L) {
  if (!idwt_coarse_enabled() || !idwt_coarse_length(L)) return 0;
  int n = 0, hsum = 0;
  for (int j = J - 1; j >= 0 && n < DC_MAX_LEVELS; --j, ++n) {
    int llh, llw;
    idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
    if (2 * hs[j] - L + 2 > DC_TY || 2 * ws[j] - L + 2 > DC_TX || llh > DC_IY || llw > DC_IX) break;
    hsum += 3 * hs[j] * ws[j];
    if (hsum > DC_HS_MAX || idwt_coarse_smem(L, hsum) > kCoarseSmemMax) break;      // (the detail bands of all its levels: one prefetch batch, and they must fit the LDS)
  }
  return n >= 2 ? n : 0;
}

}  // namespace aph

using namespace aph;

extern "C" {

// one SFB2D level.  d_ll [C, ll_h, ll_w] (ll_h in {h, h+1}, ll_w in {w, w+1}), d_highs [C,3,h,w] (LH,HL,HH),
// d_g0/d_g1 = rec_lo / rec_hi taps on the device (L even, <= 64), hscale = dwt_scale gain of this level
// -> d_out [C, 2h-L+2, 2w-L+2]
int aph_idwt_level_fwd(const float* d_ll, int ll_h, int ll_w, const float* d_highs, int h, int w, int C, const float* d_g0,
                       const float* d_g1, int L, float hscale, float* d_out, void* stream_) {
  APH_TRY
  if (!d_ll || !d_highs || !d_g0 || !d_g1 || !d_out || h < 1 || w < 1 || C < 1 || L < 2 || L > 64 || (L & 1) || ll_h < h || ll_w < w)
    return aph_fail(APH_ERR_ARG, "aph_idwt_level_fwd: bad argument (h=%d w=%d L=%d ll=%dx%d)", h, w, L, ll_h, ll_w);
  if (2 * h - L + 2 < 1 || 2 * w - L + 2 < 1) return aph_fail(APH_ERR_ARG, "aph_idwt_level_fwd: level %dx%d too small for filter length %d", h, w, L);
  idwt_level_fwd(d_ll, ll_h, ll_w, d_highs, h, w, C, d_g0, d_g1, L, hscale, d_out, (hipStream_t)stream_);
  return aph_check_launch("aph_idwt_level_fwd");
  APH_CATCH
}

int aph_idwt_level_bwd(const float* d_out_grad, int h, int w, int C, const float* d_g0, const float* d_g1, int L, float hscale,
                       float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad, void* stream_) {
  APH_TRY
  if (!d_out_grad || !d_g0 || !d_g1 || !d_ll_grad || !d_highs_grad || h < 1 || w < 1 || C < 1 || L < 2 || L > 64 || (L & 1) || ll_h < h || ll_w < w)
    return aph_fail(APH_ERR_ARG, "aph_idwt_level_bwd: bad argument");
  idwt_level_bwd(d_out_grad, h, w, C, d_g0, d_g1, L, hscale, d_ll_grad, ll_h, ll_w, d_highs_grad, (hipStream_t)stream_);
  return aph_check_launch("aph_idwt_level_bwd");
  APH_CATCH
}

// Every level of DWTInverse in one call (image.py:36-38,67).  Host arrays of J entries, level 0 = finest: hs / ws = size of
// the level's detail bands, hscales = dwt_scale gains, d_highs[j] [C,3,hs[j],ws[j]] (device), d_bufs[j] [C, 2 hs[j]-L+2,
// 2 ws[j]-L+2] = scratch for the running low band after level j (caller-owned; d_bufs[0] receives the image, the others are
// only written where a level's output has to travel through global memory).  d_yl [C, hs[J-1], ws[J-1]].
int aph_idwt_fwd(const float* d_yl, const float* const* d_highs, const int* hs, const int* ws, const float* hscales, int J, int C,
                 const float* d_g0, const float* d_g1, int L, float* const* d_bufs, void* stream_) {
  APH_TRY
  if (!d_yl || !d_highs || !hscales || !d_g0 || !d_g1 || !d_bufs) return aph_fail(APH_ERR_ARG, "aph_idwt_fwd: null argument");
  if (const int rc = idwt_check_levels("aph_idwt_fwd", hs, ws, J, C, L)) return rc;
  for (int j = 0; j < J; ++j)
    if (!d_highs[j] || !d_bufs[j]) return aph_fail(APH_ERR_ARG, "aph_idwt_fwd: null buffer at level %d", j);
  hipStream_t st = (hipStream_t)stream_;
  const int nc = idwt_coarse_count(hs, ws, J, L);
  int j = J - 1;
  if (nc) {
    IdwtLevels lv;
    lv.n = nc;
    int hoff = 0;
    bool fits = true;
    for (int i = 0; i < nc; ++i, --j) {
      int llh, llw;
      idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
      const ptrdiff_t delta = d_highs[j] - d_highs[J - 1];      // (in floats; the kernel addresses every level's bands from the first one's pointer)
      if (delta > (ptrdiff_t)0x3fffffff || delta < -(ptrdiff_t)0x3fffffff) { fits = false; break; }
      lv.lv[i] = IdwtLevel{const_cast<float*>(j + 1 < J ? d_bufs[j + 1] : d_yl), const_cast<float*>(d_highs[j]), d_bufs[j], llh, llw, hs[j], ws[j], hscales[j], hoff,
                           (int)delta};
      hoff += 3 * hs[j] * ws[j];
    }
    if (fits) idwt_coarse_fwd(lv, C, d_g0, d_g1, L, st);
    else j = J - 1;                      // (detail bands in allocations too far apart for 32-bit offsets: one launch per level)
  }
  for (; j >= 0; --j) {
    int llh, llw;
    idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
    idwt_level_fwd(j + 1 < J ? d_bufs[j + 1] : d_yl, llh, llw, d_highs[j], hs[j], ws[j], C, d_g0, d_g1, L, hscales[j], d_bufs[j], st);
  }
  return aph_check_launch("aph_idwt_fwd");
  APH_CATCH
}

// adjoint of aph_idwt_fwd: d_img_grad [C, 2 hs[0]-L+2, 2 ws[0]-L+2] -> d_highs_grad[j] [C,3,hs[j],ws[j]] and d_yl_grad
// [C, hs[J-1], ws[J-1]].  d_gbufs[j], j >= 1: scratch of d_bufs[j]'s size (gradient of the running low band; d_gbufs[0] unused).
int aph_idwt_bwd(const float* d_img_grad, const int* hs, const int* ws, const float* hscales, int J, int C, const float* d_g0,
                 const float* d_g1, int L, float* const* d_gbufs, float* d_yl_grad, float* const* d_highs_grad, void* stream_) {
  APH_TRY
  if (!d_img_grad || !hscales || !d_g0 || !d_g1 || !d_gbufs || !d_yl_grad || !d_highs_grad) return aph_fail(APH_ERR_ARG, "aph_idwt_bwd: null argument");
  if (const int rc = idwt_check_levels("aph_idwt_bwd", hs, ws, J, C, L)) return rc;
  for (int j = 0; j < J; ++j)
    if (!d_highs_grad[j] || (j && !d_gbufs[j])) return aph_fail(APH_ERR_ARG, "aph_idwt_bwd: null buffer at level %d", j);
  hipStream_t st = (hipStream_t)stream_;
  const int nc = idwt_coarse_count(hs, ws, J, L);
  int j = 0;
  for (; j < J - nc; ++j) {
    int llh, llw;
    idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
    idwt_level_bwd(j ? d_gbufs[j] : d_img_grad, hs[j], ws[j], C, d_g0, d_g1, L, hscales[j], j + 1 < J ? d_gbufs[j + 1] : d_yl_grad, llh, llw,
                   d_highs_grad[j], st);
  }
  if (nc) {
    IdwtLevels lv;
    lv.n = nc;
    for (int i = 0; i < nc; ++i, ++j) {
      int llh, llw;
      idwt_ll_size(hs, ws, J, L, j, &llh, &llw);
      lv.lv[i] = IdwtLevel{j + 1 < J ? d_gbufs[j + 1] : d_yl_grad, d_highs_grad[j], const_cast<float*>(j ? d_gbufs[j] : d_img_grad), llh, llw, hs[j], ws[j],
                           hscales[j], 0, 0};
    }
    idwt_coarse_bwd(lv, C, d_g0, d_g1, L, st);
  }
  return aph_check_launch("aph_idwt_bwd");
  APH_CATCH
}

}  // extern "C"
