// Inverse 2-D discrete wavelet transform, one level per launch, and its adjoint.
//
// Replaces: pytorch_wavelets DWTInverse(wave, mode='symmetric') as called by aphantasia/image.py:36-38,67
// (dwt_image.inner), i.e. per level  lowlevel.SFB2D:
//     lo = sfb1d(ll, LH, dim H);  hi = sfb1d(HL, HH, dim H);  out = sfb1d(lo, hi, dim W)
//     sfb1d(a, b)[m] = full[m + L - 2],  full[2 i + k] += a[i] rec_lo[k] + b[i] rec_hi[k]     (length 2 n - L + 2)
// The per-level detail gain of dwt_scale (image.py:73-80) is folded in as `hscale`.
// HBM-bound (the finest level of a 4K image moves ~200 MB); each workgroup stages the input patch of its
// output tile in LDS once and every coefficient / pixel is read from HBM exactly once.
#include "aph_device.h"
#include "aph_host.h"

namespace aph {

constexpr int DW_TY = 32, DW_TX = 128;   // forward: output tile (rows x cols) per workgroup
constexpr int DW_IY = 16, DW_IX = 64;    // adjoint: coefficient tile per workgroup

// Forward, separable in LDS.  Per workgroup: a DW_TY x DW_TX output tile of one channel.
//   1. the four band patches [PY][PX] (PY = TY/2 + L/2 input rows, PX likewise; zero beyond h / w) are loaded once, each
//      wave streaming whole rows;
//   2. horizontal pass, one thread per (patch row, output column PAIR): outputs 2j and 2j+1 read the same L/2 inputs
//      i = j + L/2 - 1 - t with the even / odd filter taps:  rlo = g0 * ll + g1 * HL,  rhi = g0 * LH + g1 * HH;
//   3. vertical pass, one thread per (output row pair, column): out = g0 * rlo + g1 * rhi, written as whole 512-byte rows.
// ll: [C][llh][llw] (only rows < h / cols < w are used -- the "unpad" of DWTInverse.forward), highs: [C][3][h][w]
// (LH, HL, HH), out: [C][Ho][Wo], Ho = 2h-L+2, Wo = 2w-L+2.  H2T = L/2 at compile time (0: run-time length).
template <int H2T>
__global__ __launch_bounds__(256) void idwt_level_kernel(const float* __restrict__ ll, int llh, int llw,
                                                         const float* __restrict__ highs, int h, int w, const float* __restrict__ g0,
                                                         const float* __restrict__ g1, int L, float hscale, float* __restrict__ out,
                                                         int Ho, int Wo) {
  APH_DYN_SMEM(smem);
  const int H2 = H2T ? H2T : (L >> 1);
  const int PY = DW_TY / 2 + H2, PX = DW_TX / 2 + H2;
  float* f0 = reinterpret_cast<float*>(smem);  // rec_lo
  float* f1 = f0 + 2 * H2;                     // rec_hi
  float* pll = f1 + 2 * H2;                    // 4 band patches [PY][PX]
  float* plh = pll + PY * PX;
  float* phl = plh + PY * PX;
  float* phh = phl + PY * PX;
  float* rlo = phh + PY * PX;                  // [PY][TX]
  float* rhi = rlo + PY * DW_TX;
  const int c = blockIdx.z, my0 = blockIdx.y * DW_TY, mx0 = blockIdx.x * DW_TX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 2 * H2; k += blockDim.x) { f0[k] = g0[k]; f1[k] = g1[k]; }
  // output m = 2 j + p (p = 0, 1) reads inputs i = j + H2 - 1 - t with tap k = p + 2 t;  the tile's first input is m0 / 2
  const int iy0 = my0 / 2, ix0 = mx0 / 2;
  const float* bll = ll + (size_t)c * llh * llw;
  const float* bh = highs + (size_t)c * 3 * h * w;
  for (int py = wave; py < PY; py += 4) {
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
  __syncthreads();
  for (int q = threadIdx.x; q < PY * (DW_TX / 2); q += blockDim.x) {
    const int py = q / (DW_TX / 2), jl = q - py * (DW_TX / 2);
    const int e0 = py * PX + jl + H2 - 1;
    float lo_e = 0.f, lo_o = 0.f, hi_e = 0.f, hi_o = 0.f;
#pragma unroll
    for (int t = 0; t < (H2T ? H2T : H2); ++t) {
      const float a_e = f0[2 * t], a_o = f0[2 * t + 1], b_e = f1[2 * t], b_o = f1[2 * t + 1];
      const float vll = pll[e0 - t], vlh = plh[e0 - t], vhl = phl[e0 - t], vhh = phh[e0 - t];
      lo_e += a_e * vll + b_e * vhl; lo_o += a_o * vll + b_o * vhl;
      hi_e += a_e * vlh + b_e * vhh; hi_o += a_o * vlh + b_o * vhh;
    }
    *reinterpret_cast<float2*>(rlo + py * DW_TX + 2 * jl) = make_float2(lo_e, lo_o);
    *reinterpret_cast<float2*>(rhi + py * DW_TX + 2 * jl) = make_float2(hi_e, hi_o);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < (DW_TY / 2) * DW_TX; q += blockDim.x) {
    const int rl = q / DW_TX, tx_ = q - rl * DW_TX, my = my0 + 2 * rl, mx = mx0 + tx_;
    const int e0 = (rl + H2 - 1) * DW_TX + tx_;
    float o_e = 0.f, o_o = 0.f;
#pragma unroll
    for (int t = 0; t < (H2T ? H2T : H2); ++t) {
      const float lo = rlo[e0 - t * DW_TX], hi = rhi[e0 - t * DW_TX];
      o_e += f0[2 * t] * lo + f1[2 * t] * hi;
      o_o += f0[2 * t + 1] * lo + f1[2 * t + 1] * hi;
    }
    if (mx < Wo) {
      if (my < Ho) out[((size_t)c * Ho + my) * Wo + mx] = o_e;
      if (my + 1 < Ho) out[((size_t)c * Ho + my + 1) * Wo + mx] = o_o;
    }
  }
}

// Adjoint, separable in LDS: dout [C][Ho][Wo] -> dll [C][llh][llw] (extra unpadded row/col = 0) and dhighs [C][3][h][w]
// (x hscale).  Coefficient i touches outputs m = 2 i + k - (L - 2), k in [0, L).  Per workgroup a DW_IY x DW_IX tile:
//   1. dout patch [PY][PX], PY = 2 IY + L - 2, whole rows per wave;
//   2. horizontal: slo[y][ix] = sum_k g0[k] pd[y][2 ix + k], shi with g1 -> LDS [2][PY][IX];
//   3. vertical:   dll = sum_k g0[k] slo[2 iy + k], dLH = sum g1 slo, dHL = sum g0 shi, dHH = sum g1 shi.
template <int LT>
__global__ __launch_bounds__(256) void idwt_level_adjoint_kernel(const float* __restrict__ dout, int Ho, int Wo, int h, int w,
                                                                 const float* __restrict__ g0, const float* __restrict__ g1, int L_,
                                                                 float hscale, float* __restrict__ dll, int llh, int llw,
                                                                 float* __restrict__ dhighs) {
  APH_DYN_SMEM(smem);
  const int L = LT ? LT : L_;
  const int PY = 2 * DW_IY + L - 2, PX = 2 * DW_IX + L - 2;
  float* f0 = reinterpret_cast<float*>(smem);
  float* f1 = f0 + L;
  float* pd = f1 + L;                          // [PY][PX]
  float* slo = pd + PY * PX;                   // [PY][IX]
  float* shi = slo + PY * DW_IX;
  const int c = blockIdx.z, iy0 = blockIdx.y * DW_IY, ix0 = blockIdx.x * DW_IX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < L; k += blockDim.x) { f0[k] = g0[k]; f1[k] = g1[k]; }
  const int my0 = 2 * iy0 - (L - 2), mx0 = 2 * ix0 - (L - 2);
  const float* bd = dout + (size_t)c * Ho * Wo;
  for (int py = wave; py < PY; py += 4) {
    const int my = my0 + py;
    const bool rowok = my >= 0 && my < Ho;
    for (int px = lane; px < PX; px += 64) {
      const int mx = mx0 + px;
      pd[py * PX + px] = (rowok && mx >= 0 && mx < Wo) ? bd[(size_t)my * Wo + mx] : 0.f;
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < PY * DW_IX; q += blockDim.x) {
    const int py = q / DW_IX, ixl = q - py * DW_IX;
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
  for (int q = threadIdx.x; q < DW_IY * DW_IX; q += blockDim.x) {
    const int iyl = q / DW_IX, ixl = q - iyl * DW_IX, iy = iy0 + iyl, ix = ix0 + ixl;
    if (iy >= llh || ix >= llw) continue;
    float* ol = dll + (size_t)c * llh * llw + (size_t)iy * llw + ix;
    if (iy >= h || ix >= w) { *ol = 0.f; continue; }      // the row / column DWTInverse drops
    float all = 0.f, alh = 0.f, ahl = 0.f, ahh = 0.f;
#pragma unroll
    for (int k = 0; k < (LT ? LT : L); ++k) {
      const int e = (2 * iyl + k) * DW_IX + ixl;
      const float lo = slo[e], hi = shi[e];
      all += f0[k] * lo; alh += f1[k] * lo;
      ahl += f0[k] * hi; ahh += f1[k] * hi;
    }
    *ol = all;
    float* oh = dhighs + (size_t)c * 3 * h * w + (size_t)iy * w + ix;
    oh[0] = alh * hscale;
    oh[(size_t)h * w] = ahl * hscale;
    oh[2 * (size_t)h * w] = ahh * hscale;
  }
}

template <int H2T>
void launch_idwt_fwd(dim3 grid, size_t smem, hipStream_t st, const float* d_ll, int ll_h, int ll_w, const float* d_highs, int h, int w,
                     const float* d_g0, const float* d_g1, int L, float hscale, float* d_out, int Ho, int Wo) {
  APH_ALLOW_SMEM(idwt_level_kernel<H2T>, 150 * 1024);
  APH_LAUNCH(idwt_level_kernel<H2T>, grid, dim3(256), smem, st, d_ll, ll_h, ll_w, d_highs, h, w, d_g0, d_g1, L, hscale, d_out, Ho, Wo);
}
template <int LT>
void launch_idwt_bwd(dim3 grid, size_t smem, hipStream_t st, const float* d_out_grad, int Ho, int Wo, int h, int w, const float* d_g0,
                     const float* d_g1, int L, float hscale, float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad) {
  APH_ALLOW_SMEM(idwt_level_adjoint_kernel<LT>, 150 * 1024);
  APH_LAUNCH(idwt_level_adjoint_kernel<LT>, grid, dim3(256), smem, st, d_out_grad, Ho, Wo, h, w, d_g0, d_g1, L, hscale, d_ll_grad, ll_h,
             ll_w, d_highs_grad);
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
  const int Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
  if (Ho < 1 || Wo < 1) return aph_fail(APH_ERR_ARG, "aph_idwt_level_fwd: level %dx%d too small for filter length %d", h, w, L);
  const int PY = DW_TY / 2 + L / 2, PX = DW_TX / 2 + L / 2;
  const size_t smem = sizeof(float) * (2 * L + 4 * PY * PX + 2 * PY * DW_TX);
  const dim3 grid((Wo + DW_TX - 1) / DW_TX, (Ho + DW_TY - 1) / DW_TY, C);
  hipStream_t st = (hipStream_t)stream_;
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
  return aph_check_launch("aph_idwt_level_fwd");
  APH_CATCH
}

int aph_idwt_level_bwd(const float* d_out_grad, int h, int w, int C, const float* d_g0, const float* d_g1, int L, float hscale,
                       float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad, void* stream_) {
  APH_TRY
  if (!d_out_grad || !d_g0 || !d_g1 || !d_ll_grad || !d_highs_grad || h < 1 || w < 1 || C < 1 || L < 2 || L > 64 || (L & 1) || ll_h < h || ll_w < w)
    return aph_fail(APH_ERR_ARG, "aph_idwt_level_bwd: bad argument");
  const int Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
  const int PY = 2 * DW_IY + L - 2, PX = 2 * DW_IX + L - 2;
  const size_t smem = sizeof(float) * (2 * L + PY * PX + 2 * PY * DW_IX);
  const dim3 grid((ll_w + DW_IX - 1) / DW_IX, (ll_h + DW_IY - 1) / DW_IY, C);
  hipStream_t st = (hipStream_t)stream_;
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
  return aph_check_launch("aph_idwt_level_bwd");
  APH_CATCH
}

}  // extern "C"
