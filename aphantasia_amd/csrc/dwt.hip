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

constexpr int DW_TO = 32;      // forward: output tile edge
constexpr int DW_TI = 16;      // adjoint: coefficient tile edge

// out[c][my][mx] for a DW_TO x DW_TO tile.  ll: [C][llh][llw] (only rows < h / cols < w are used -- the "unpad"
// of DWTInverse.forward), highs: [C][3][h][w] (LH, HL, HH), out: [C][Ho][Wo], Ho = 2h-L+2, Wo = 2w-L+2.
__global__ __launch_bounds__(256) void idwt_level_kernel(const float* __restrict__ ll, int llh, int llw,
                                                         const float* __restrict__ highs, int h, int w, const float* __restrict__ g0,
                                                         const float* __restrict__ g1, int L, float hscale, float* __restrict__ out,
                                                         int Ho, int Wo) {
  APH_DYN_SMEM(smem);
  const int H2 = L >> 1;
  const int P = DW_TO / 2 + H2 + 1;            // patch edge (input positions needed by the tile)
  float* f0 = reinterpret_cast<float*>(smem);  // rec_lo
  float* f1 = f0 + L;                          // rec_hi
  float* pll = f1 + L;                         // 4 band patches [P][P]
  float* plh = pll + P * P;
  float* phl = plh + P * P;
  float* phh = phl + P * P;
  const int c = blockIdx.z, my0 = blockIdx.y * DW_TO, mx0 = blockIdx.x * DW_TO;
  for (int k = threadIdx.x; k < L; k += blockDim.x) { f0[k] = g0[k]; f1[k] = g1[k]; }
  // first input index touched by output m0: i >= ceil((m0 - 1) / 2)  (k <= L-1)
  const int iy0 = my0 > 0 ? my0 / 2 : 0, ix0 = mx0 > 0 ? mx0 / 2 : 0;
  const float* bll = ll + (size_t)c * llh * llw;
  const float* bh = highs + (size_t)c * 3 * h * w;
  for (int q = threadIdx.x; q < P * P; q += blockDim.x) {
    const int py = q / P, px = q - py * P, iy = iy0 + py, ix = ix0 + px;
    const bool ok = iy < h && ix < w;
    const size_t o = (size_t)iy * w + ix;
    pll[q] = ok ? bll[(size_t)iy * llw + ix] : 0.f;
    plh[q] = ok ? bh[o] * hscale : 0.f;
    phl[q] = ok ? bh[(size_t)h * w + o] * hscale : 0.f;
    phh[q] = ok ? bh[2 * (size_t)h * w + o] * hscale : 0.f;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < DW_TO * DW_TO; t += blockDim.x) {
    const int my = my0 + t / DW_TO, mx = mx0 + t % DW_TO;
    if (my >= Ho || mx >= Wo) continue;
    // 2 i + k = m + L - 2 :  k = kp + 2 t', i = (m + L - 2 - kp) / 2 - t'
    const int kpy = (my + L) & 1, kpx = (mx + L) & 1;
    const int iyb = (my + L - 2 - kpy) >> 1, ixb = (mx + L - 2 - kpx) >> 1;
    float acc = 0.f;
    for (int ty = 0; ty < H2; ++ty) {
      const int iy = iyb - ty;
      if (iy < 0 || iy >= h) continue;
      const int py = iy - iy0;
      float rlo = 0.f, rhi = 0.f;
      for (int tx = 0; tx < H2; ++tx) {
        const int ix = ixb - tx;
        if (ix < 0 || ix >= w) continue;
        const int q = py * P + (ix - ix0);
        const float a = f0[kpx + 2 * tx], b = f1[kpx + 2 * tx];
        rlo += a * pll[q] + b * phl[q];
        rhi += a * plh[q] + b * phh[q];
      }
      acc += f0[kpy + 2 * ty] * rlo + f1[kpy + 2 * ty] * rhi;
    }
    out[((size_t)c * Ho + my) * Wo + mx] = acc;
  }
}

// adjoint: dout [C][Ho][Wo] -> dll [C][llh][llw] (extra unpadded row/col = 0) and dhighs [C][3][h][w] (x hscale)
__global__ __launch_bounds__(256) void idwt_level_adjoint_kernel(const float* __restrict__ dout, int Ho, int Wo, int h, int w,
                                                                 const float* __restrict__ g0, const float* __restrict__ g1, int L,
                                                                 float hscale, float* __restrict__ dll, int llh, int llw,
                                                                 float* __restrict__ dhighs) {
  APH_DYN_SMEM(smem);
  const int P = 2 * DW_TI + L;                 // dout patch edge
  float* f0 = reinterpret_cast<float*>(smem);
  float* f1 = f0 + L;
  float* pd = f1 + L;                          // [P][P]
  const int c = blockIdx.z, iy0 = blockIdx.y * DW_TI, ix0 = blockIdx.x * DW_TI;
  for (int k = threadIdx.x; k < L; k += blockDim.x) { f0[k] = g0[k]; f1[k] = g1[k]; }
  // coefficient i touches outputs m = 2 i + k - (L - 2), k in [0, L)
  const int my0 = 2 * iy0 - (L - 2), mx0 = 2 * ix0 - (L - 2);
  const float* bd = dout + (size_t)c * Ho * Wo;
  for (int q = threadIdx.x; q < P * P; q += blockDim.x) {
    const int py = q / P, px = q - py * P, my = my0 + py, mx = mx0 + px;
    pd[q] = (my >= 0 && my < Ho && mx >= 0 && mx < Wo) ? bd[(size_t)my * Wo + mx] : 0.f;
  }
  __syncthreads();
  const int iy = iy0 + threadIdx.x / DW_TI, ix = ix0 + threadIdx.x % DW_TI;
  if (iy >= llh || ix >= llw) return;
  float* ol = dll + (size_t)c * llh * llw + (size_t)iy * llw + ix;
  if (iy >= h || ix >= w) { *ol = 0.f; return; }      // the row / column DWTInverse drops
  float all = 0.f, alh = 0.f, ahl = 0.f, ahh = 0.f;
  const int py0 = 2 * (iy - iy0), px0 = 2 * (ix - ix0);
  for (int ky = 0; ky < L; ++ky) {
    float slo = 0.f, shi = 0.f;
    const float* row = pd + (py0 + ky) * P + px0;
    for (int kx = 0; kx < L; ++kx) { slo += f0[kx] * row[kx]; shi += f1[kx] * row[kx]; }
    all += f0[ky] * slo; alh += f1[ky] * slo;
    ahl += f0[ky] * shi; ahh += f1[ky] * shi;
  }
  *ol = all;
  float* oh = dhighs + (size_t)c * 3 * h * w + (size_t)iy * w + ix;
  oh[0] = alh * hscale;
  oh[(size_t)h * w] = ahl * hscale;
  oh[2 * (size_t)h * w] = ahh * hscale;
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
  const int P = DW_TO / 2 + L / 2 + 1;
  const size_t smem = sizeof(float) * (2 * L + 4 * P * P);
  APH_LAUNCH(idwt_level_kernel, dim3((Wo + DW_TO - 1) / DW_TO, (Ho + DW_TO - 1) / DW_TO, C), dim3(256), smem, (hipStream_t)stream_, d_ll,
             ll_h, ll_w, d_highs, h, w, d_g0, d_g1, L, hscale, d_out, Ho, Wo);
  return aph_check_launch("aph_idwt_level_fwd");
  APH_CATCH
}

int aph_idwt_level_bwd(const float* d_out_grad, int h, int w, int C, const float* d_g0, const float* d_g1, int L, float hscale,
                       float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad, void* stream_) {
  APH_TRY
  if (!d_out_grad || !d_g0 || !d_g1 || !d_ll_grad || !d_highs_grad || h < 1 || w < 1 || C < 1 || L < 2 || L > 64 || (L & 1) || ll_h < h || ll_w < w)
    return aph_fail(APH_ERR_ARG, "aph_idwt_level_bwd: bad argument");
  const int Ho = 2 * h - L + 2, Wo = 2 * w - L + 2;
  const int P = 2 * DW_TI + L;
  const size_t smem = sizeof(float) * (2 * L + P * P);
  APH_LAUNCH(idwt_level_adjoint_kernel, dim3((ll_w + DW_TI - 1) / DW_TI, (ll_h + DW_TI - 1) / DW_TI, C), dim3(256), smem,
             (hipStream_t)stream_, d_out_grad, Ho, Wo, h, w, d_g0, d_g1, L, hscale, d_ll_grad, ll_h, ll_w, d_highs_grad);
  return aph_check_launch("aph_idwt_level_bwd");
  APH_CATCH
}

}  // extern "C"
