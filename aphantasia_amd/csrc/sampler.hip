// Sampler kernels (SURVEY.md K5-K10 and adjoints): random square crops of the rgb image,
// bicubic resize to size x size, optional torchvision-style geometric augmentation, CLIP
// normalisation, and layout conversion to the patch-embed GEMM operand.
//
// Replaces: aphantasia/utils.py:243-253 (the per-cut Python loop: slice + F.interpolate bicubic
// align_corners=True), utils.py:152-187 (pad_up_to / tile_pad wrap padding, folded in as modular
// addressing), transforms.py:102-109 (normalize), transforms.py:165-170 (transforms_fast:
// RandomPerspective -> RandomErasing -> rotate, torchvision grid_sample semantics).
//
// All random parameters are drawn on the host exactly as the reference draws them; kernels are
// RNG-free.  One launch covers all S cuts (the reference issues S x ~10 tiny launches).
//
// Adjoint of the crop/resize: deterministic GATHER over crops per image pixel (fixed summation
// order s = 0..S-1, no atomics) so a given crop table gives bitwise-reproducible gradients.
// Adjoint of the bilinear warps: fp32 atomics into a per-cut buffer (only the `-tf fast` path).
#include "aph_device.h"
#include "aph_host.h"

namespace aph {

__device__ __constant__ const float kClipMean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
__device__ __constant__ const float kClipStd[3] = {0.26862954f, 0.26130258f, 0.27577711f};

struct Geom {
  int H, W, Hp, Wp, py0, px0, S, size, patch;
};

// cubic convolution weights, A = -0.75 (ATen UpSampleBicubic get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
  const float A = -0.75f;
  const float x1 = t, x2 = 1.0f - t;
  w[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
}

__device__ __forceinline__ int wrap(int v, int n) {
  v %= n;
  return v < 0 ? v + n : v;
}

// patch-major element offset of pixel (c,i,j) of cut s
__device__ __forceinline__ size_t patch_index(int s, int c, int i, int j, int size, int p) {
  const int g = size / p;
  return ((size_t)s * g * g + (size_t)(i / p) * g + (j / p)) * (size_t)(3 * p * p) + (size_t)c * p * p + (i % p) * p + (j % p);
}

template <int OUT>
__device__ __forceinline__ void emit(void* out, int s, int c, int i, int j, int size, int patch, float v) {
  if (OUT == APH_OUT_NCHW_RAW) {
    reinterpret_cast<float*>(out)[(((size_t)s * 3 + c) * size + i) * size + j] = v;
  } else if (OUT == APH_OUT_NCHW_NORM) {
    reinterpret_cast<float*>(out)[(((size_t)s * 3 + c) * size + i) * size + j] = (v - kClipMean[c]) / kClipStd[c];
  } else {
    reinterpret_cast<half_t*>(out)[patch_index(s, c, i, j, size, patch)] = (half_t)((v - kClipMean[c]) / kClipStd[c]);
  }
}

// gradient w.r.t. the un-normalised cut pixel (c,i,j) of cut s, read from `gout` in layout OUT
template <int OUT>
__device__ __forceinline__ float fetch_grad(const float* __restrict__ gout, int s, int c, int i, int j, int size, int patch) {
  if (OUT == APH_OUT_NCHW_RAW) return gout[(((size_t)s * 3 + c) * size + i) * size + j];
  if (OUT == APH_OUT_NCHW_NORM) return gout[(((size_t)s * 3 + c) * size + i) * size + j] / kClipStd[c];
  return gout[patch_index(s, c, i, j, size, patch)] / kClipStd[c];
}

// ---------------------------------------------------------------------------------
// crop + bicubic resize  (utils.py:248-249)
// ---------------------------------------------------------------------------------
template <int OUT>
__global__ void crop_resize_kernel(const float* __restrict__ rgb, const int* __restrict__ table, void* __restrict__ out, Geom g) {
  const int s = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= g.size * g.size) return;
  const int i = pix / g.size, j = pix - i * g.size;
  const int cs = table[3 * s], ox = table[3 * s + 1], oy = table[3 * s + 2];
  // area_pixel_compute_scale(align_corners=True): (in-1)/(out-1), source index = scale*dst, all fp32
  const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
  const float sy = scale * (float)i, sx = scale * (float)j;
  const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
  float wy[4], wx[4];
  cubic_w(sy - (float)y0, wy);
  cubic_w(sx - (float)x0, wx);
  int ry[4], rx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int yy = y0 - 1 + k; yy = yy < 0 ? 0 : (yy > cs - 1 ? cs - 1 : yy);   // clamp inside the cut
    int xx = x0 - 1 + k; xx = xx < 0 ? 0 : (xx > cs - 1 ? cs - 1 : xx);
    ry[k] = wrap(oy + yy - g.py0, g.H);                                     // tile_pad wrap (utils.py:165-167)
    rx[k] = wrap(ox + xx - g.px0, g.W);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = rgb + (size_t)c * g.H * g.W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* row = pl + (size_t)ry[a] * g.W;
      const float r = row[rx[0]] * wx[0] + row[rx[1]] * wx[1] + row[rx[2]] * wx[2] + row[rx[3]] * wx[3];
      acc += r * wy[a];
    }
    emit<OUT>(out, s, c, i, j, g.size, g.patch, acc);
  }
}

// Adjoint of crop_resize over all cuts: one thread per source pixel, gather over cuts in fixed order.
// For the 1-D map dst i -> taps clamp(floor(scale*i) - 1 + k), the cuts' outputs that touch source
// position q are a contiguous i-range; it is bracketed conservatively and every candidate is
// re-derived with the forward's own fp32 arithmetic, so weights match the forward bit for bit.
template <int OUT>
__global__ void crop_resize_adjoint_kernel(const float* __restrict__ gout, float gscale, const int* __restrict__ table,
                                           float* __restrict__ grgb, Geom g) {
  __shared__ int tcs[256], tox[256], toy[256];
  const int x = blockIdx.x * 16 + (threadIdx.x & 15);
  const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
  const bool live = x < g.W && y < g.H;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
  for (int sb = 0; sb < g.S; sb += 256) {
    const int ns = g.S - sb < 256 ? g.S - sb : 256;
    __syncthreads();
    if ((int)threadIdx.x < ns) {
      tcs[threadIdx.x] = table[3 * (sb + threadIdx.x)];
      tox[threadIdx.x] = table[3 * (sb + threadIdx.x) + 1];
      toy[threadIdx.x] = table[3 * (sb + threadIdx.x) + 2];
    }
    __syncthreads();
    for (int q = 0; q < ns; ++q) {
      const int cs = tcs[q], ox = tox[q], oy = toy[q], s = sb + q;
      const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
      const float inv = scale > 0.f ? 1.0f / scale : 0.f;
      // every padded-frame alias (Y, X) of source pixel (y, x)
      for (int Y = wrap(y + g.py0, g.H); Y < g.Hp; Y += g.H) {
        // tile-level cull (block-uniform branch keeps the wave converged on the common skip)
        const int Yt0 = Y - (y - ty0), Yt1 = Yt0 + 15;
        if (Yt1 < oy || Yt0 >= oy + cs) continue;
        for (int X = wrap(x + g.px0, g.W); X < g.Wp; X += g.W) {
          const int Xt0 = X - (x - tx0), Xt1 = Xt0 + 15;
          if (Xt1 < ox || Xt0 >= ox + cs) continue;
          const int yc = Y - oy, xc = X - ox;
          if (!live || yc < 0 || yc >= cs || xc < 0 || xc >= cs) continue;
          int ilo, ihi, jlo, jhi;
          if (scale > 0.f) {
            ilo = (int)floorf((float)(yc - 2) * inv) - 1; ihi = (int)floorf((float)(yc + 2) * inv) + 1;
            jlo = (int)floorf((float)(xc - 2) * inv) - 1; jhi = (int)floorf((float)(xc + 2) * inv) + 1;
          } else { ilo = jlo = 0; ihi = jhi = g.size - 1; }
          ilo = ilo < 0 ? 0 : ilo; jlo = jlo < 0 ? 0 : jlo;
          ihi = ihi > g.size - 1 ? g.size - 1 : ihi; jhi = jhi > g.size - 1 ? g.size - 1 : jhi;
          for (int i = ilo; i <= ihi; ++i) {
            const float sy = scale * (float)i;
            const int y0 = (int)floorf(sy);
            float wv[4];
            cubic_w(sy - (float)y0, wv);
            float wy = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              int yy = y0 - 1 + k; yy = yy < 0 ? 0 : (yy > cs - 1 ? cs - 1 : yy);
              if (yy == yc) wy += wv[k];
            }
            if (wy == 0.f) continue;
            for (int j = jlo; j <= jhi; ++j) {
              const float sx = scale * (float)j;
              const int x0 = (int)floorf(sx);
              float wu[4];
              cubic_w(sx - (float)x0, wu);
              float wx = 0.f;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                int xx = x0 - 1 + k; xx = xx < 0 ? 0 : (xx > cs - 1 ? cs - 1 : xx);
                if (xx == xc) wx += wu[k];
              }
              if (wx == 0.f) continue;
              const float w = wy * wx;
              acc0 += w * fetch_grad<OUT>(gout, s, 0, i, j, g.size, g.patch);
              acc1 += w * fetch_grad<OUT>(gout, s, 1, i, j, g.size, g.patch);
              acc2 += w * fetch_grad<OUT>(gout, s, 2, i, j, g.size, g.patch);
            }
          }
        }
      }
    }
  }
  if (live) {
    const size_t HW = (size_t)g.H * g.W, o = (size_t)y * g.W + x;
    grgb[o] = acc0 * gscale;
    grgb[HW + o] = acc1 * gscale;
    grgb[2 * HW + o] = acc2 * gscale;
  }
}

// ---------------------------------------------------------------------------------
// torchvision-style warps (grid_sample bilinear, zeros padding, align_corners=False, ones-mask fill 0)
// ---------------------------------------------------------------------------------
struct Tap {
  int x0, y0;
  float wx0, wx1, wy0, wy1;   // weights of x0, x0+1, y0, y0+1
};

// normalised grid coordinate -> bilinear footprint (at::native grid_sampler_unnormalize, align_corners=False)
__device__ __forceinline__ Tap make_tap(float gx, float gy, int n) {
  const float ix = ((gx + 1.f) * (float)n - 1.f) * 0.5f;
  const float iy = ((gy + 1.f) * (float)n - 1.f) * 0.5f;
  Tap t;
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.wx1 = ix - fx; t.wx0 = (fx + 1.f) - ix;
  t.wy1 = iy - fy; t.wy0 = (fy + 1.f) - iy;
  return t;
}

// torchvision _perspective_grid: base grid linspace(0.5, n-0.5), theta1 / (0.5 n), theta2, g1/g2 - 1
__device__ __forceinline__ Tap persp_tap(const float* __restrict__ a, int i, int j, int n) {
  const float x = (float)j + 0.5f, y = (float)i + 0.5f, hn = 0.5f * (float)n;
  const float g1x = x * (a[0] / hn) + y * (a[1] / hn) + (a[2] / hn);
  const float g1y = x * (a[3] / hn) + y * (a[4] / hn) + (a[5] / hn);
  const float g2 = x * a[6] + y * a[7] + 1.0f;
  return make_tap(g1x / g2 - 1.0f, g1y / g2 - 1.0f, n);
}

// torchvision _gen_affine_grid with the inverse rotation matrix [cos, sin, 0; -sin, cos, 0]
__device__ __forceinline__ Tap rot_tap(float cs, float sn, int i, int j, int n) {
  const float x = -(float)n * 0.5f + 0.5f + (float)j, y = -(float)n * 0.5f + 0.5f + (float)i, hn = 0.5f * (float)n;
  const float gx = x * (cs / hn) + y * (sn / hn) + (0.0f / hn);
  const float gy = x * (-sn / hn) + y * (cs / hn) + (0.0f / hn);
  return make_tap(gx, gy, n);
}

__device__ __forceinline__ bool in_rect(const float* __restrict__ a, int y, int x) {
  const int eh = (int)a[11];
  if (eh <= 0) return false;
  const int ei = (int)a[9], ej = (int)a[10], ew = (int)a[12];
  return y >= ei && y < ei + eh && x >= ej && x < ej + ew;
}

// sampled value times sampled ones-mask (fill = 0); ERASE: source pixels inside the erase rectangle read as 0
template <bool ERASE>
__device__ __forceinline__ float warp_gather(const float* __restrict__ src, const Tap& t, int n, const float* __restrict__ a) {
  float v = 0.f, m = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = t.y0 + dy, xx = t.x0 + dx;
      if (yy < 0 || yy >= n || xx < 0 || xx >= n) continue;
      const float w = (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0);
      m += w;
      if (ERASE && in_rect(a, yy, xx)) continue;
      v += w * src[(size_t)yy * n + xx];
    }
  return v * m;
}

// stage 1: RandomPerspective for the cuts that drew it (A -> B); other cuts are skipped
__global__ void persp_kernel(const float* __restrict__ A, const float* __restrict__ aug, float* __restrict__ Bo, int n) {
  const int s = blockIdx.y;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  if (a[8] == 0.f) return;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n * n) return;
  const int i = pix / n, j = pix - i * n;
  const Tap t = persp_tap(a, i, j, n);
  for (int c = 0; c < 3; ++c) {
    const size_t pl = ((size_t)s * 3 + c) * n * n;
    Bo[pl + pix] = warp_gather<false>(A + pl, t, n, a);
  }
}

// stage 2: RandomErasing (read-side) + rotation + normalise + emit
template <int OUT>
__global__ void rotate_emit_kernel(const float* __restrict__ A, const float* __restrict__ Bi, const float* __restrict__ aug,
                                   void* __restrict__ out, int n, int patch) {
  const int s = blockIdx.y;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n * n) return;
  const int i = pix / n, j = pix - i * n;
  const float* src = a[8] != 0.f ? Bi : A;
  if (a[15] != 0.f) {
    const Tap t = rot_tap(a[13], a[14], i, j, n);
    for (int c = 0; c < 3; ++c)
      emit<OUT>(out, s, c, i, j, n, patch, warp_gather<true>(src + ((size_t)s * 3 + c) * n * n, t, n, a));
  } else {
    for (int c = 0; c < 3; ++c) {
      const float v = in_rect(a, i, j) ? 0.f : src[((size_t)s * 3 + c) * n * n + pix];
      emit<OUT>(out, s, c, i, j, n, patch, v);
    }
  }
}

template <bool ERASE>
__device__ __forceinline__ void warp_scatter(float* __restrict__ dst, const Tap& t, int n, const float* __restrict__ a, float gv) {
  float m = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = t.y0 + dy, xx = t.x0 + dx;
      if (yy < 0 || yy >= n || xx < 0 || xx >= n) continue;
      m += (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0);
    }
  if (m == 0.f) return;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = t.y0 + dy, xx = t.x0 + dx;
      if (yy < 0 || yy >= n || xx < 0 || xx >= n) continue;
      if (ERASE && in_rect(a, yy, xx)) continue;
      const float w = (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0);
      atomicAdd(dst + (size_t)yy * n + xx, w * m * gv);
    }
}

// adjoint of stage 2: gout (layout OUT) -> dC accumulated into dA (cuts without perspective) or dB (with)
template <int OUT>
__global__ void rotate_emit_adjoint_kernel(const float* __restrict__ gout, const float* __restrict__ aug,
                                           float* __restrict__ dA, float* __restrict__ dB, int n, int patch) {
  const int s = blockIdx.y;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n * n) return;
  const int i = pix / n, j = pix - i * n;
  float* dst = a[8] != 0.f ? dB : dA;
  if (a[15] != 0.f) {
    const Tap t = rot_tap(a[13], a[14], i, j, n);
    for (int c = 0; c < 3; ++c)
      warp_scatter<true>(dst + ((size_t)s * 3 + c) * n * n, t, n, a, fetch_grad<OUT>(gout, s, c, i, j, n, patch));
  } else if (!in_rect(a, i, j)) {
    for (int c = 0; c < 3; ++c)
      atomicAdd(dst + ((size_t)s * 3 + c) * n * n + pix, fetch_grad<OUT>(gout, s, c, i, j, n, patch));
  }
}

// adjoint of stage 1: dB -> dA for the cuts with perspective
__global__ void persp_adjoint_kernel(const float* __restrict__ dB, const float* __restrict__ aug, float* __restrict__ dA, int n) {
  const int s = blockIdx.y;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  if (a[8] == 0.f) return;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n * n) return;
  const int i = pix / n, j = pix - i * n;
  const Tap t = persp_tap(a, i, j, n);
  for (int c = 0; c < 3; ++c) {
    const size_t pl = ((size_t)s * 3 + c) * n * n;
    warp_scatter<false>(dA + pl, t, n, a, dB[pl + pix]);
  }
}

// ---------------------------------------------------------------------------------
// layout conversion for caller-made batches (model.encode_image(x) on an NCHW tensor)
// ---------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ x, half_t* __restrict__ out, int S, int R, int p) {
  const size_t n = (size_t)S * 3 * R * R;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx % R, i = (idx / R) % R, c = (idx / ((size_t)R * R)) % 3, s = idx / ((size_t)3 * R * R);
    out[patch_index(s, c, i, j, R, p)] = (half_t)x[idx];
  }
}
__global__ void unpatchify_kernel(const float* __restrict__ g, float* __restrict__ out, int S, int R, int p, float gscale) {
  const size_t n = (size_t)S * 3 * R * R;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx % R, i = (idx / R) % R, c = (idx / ((size_t)R * R)) % 3, s = idx / ((size_t)3 * R * R);
    out[idx] = g[patch_index(s, c, i, j, R, p)] * gscale;
  }
}

}  // namespace aph

using namespace aph;

static Geom to_geom(const aph_sample_geom* g) { return Geom{g->H, g->W, g->Hp, g->Wp, g->py0, g->px0, g->S, g->size, g->patch}; }

static int check_geom(const aph_sample_geom* g, int out_mode, const char* who) {
  if (!g) return aph_fail(APH_ERR_ARG, "%s: null geometry", who);
  if (g->S < 1 || g->size < 1 || g->H < 1 || g->W < 1 || g->Hp < g->H || g->Wp < g->W)
    return aph_fail(APH_ERR_ARG, "%s: bad geometry S=%d size=%d H=%d W=%d Hp=%d Wp=%d", who, g->S, g->size, g->H, g->W, g->Hp, g->Wp);
  if (out_mode < 0 || out_mode > 2) return aph_fail(APH_ERR_ARG, "%s: bad out_mode %d", who, out_mode);
  if (out_mode == APH_OUT_PATCH_F16 && (g->patch < 1 || g->size % g->patch))
    return aph_fail(APH_ERR_ARG, "%s: size %d not divisible by patch %d", who, g->size, g->patch);
  return APH_OK;
}

extern "C" {

int aph_sample_fwd(const aph_sample_geom* gg, const float* rgb, const int32_t* table, const float* aug, float* tmp,
                   void* out, int out_mode, void* stream_) {
  APH_TRY
  if (int e = check_geom(gg, out_mode, "aph_sample_fwd")) return e;
  if (!rgb || !table || !out || (aug && !tmp)) return aph_fail(APH_ERR_ARG, "aph_sample_fwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const Geom g = to_geom(gg);
  const int n = g.size;
  const dim3 grid((n * n + 255) / 256, g.S), block(256);
  if (!aug) {
    if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(crop_resize_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, rgb, (const int*)table, out, g);
    else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(crop_resize_kernel<APH_OUT_NCHW_NORM>, grid, block, 0, st, rgb, (const int*)table, out, g);
    else APH_LAUNCH(crop_resize_kernel<APH_OUT_PATCH_F16>, grid, block, 0, st, rgb, (const int*)table, out, g);
    return aph_check_launch("aph_sample_fwd");
  }
  float* A = tmp;
  float* Bv = tmp + (size_t)g.S * 3 * n * n;
  APH_LAUNCH(crop_resize_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, rgb, (const int*)table, (void*)A, g);
  APH_LAUNCH(persp_kernel, grid, block, 0, st, (const float*)A, aug, Bv, n);
  if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(rotate_emit_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(rotate_emit_kernel<APH_OUT_NCHW_NORM>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  else APH_LAUNCH(rotate_emit_kernel<APH_OUT_PATCH_F16>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  return aph_check_launch("aph_sample_fwd");
  APH_CATCH
}

int aph_sample_bwd(const aph_sample_geom* gg, const float* gout, float gscale, const int32_t* table, const float* aug,
                   float* tmp, float* grgb, int out_mode, void* stream_) {
  APH_TRY
  if (int e = check_geom(gg, out_mode, "aph_sample_bwd")) return e;
  if (!gout || !table || !grgb || (aug && !tmp)) return aph_fail(APH_ERR_ARG, "aph_sample_bwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const Geom g = to_geom(gg);
  const int n = g.size;
  const dim3 agrid((g.W + 15) / 16, (g.H + 15) / 16), block(256);
  if (!aug) {
    if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(crop_resize_adjoint_kernel<APH_OUT_NCHW_RAW>, agrid, block, 0, st, gout, gscale, (const int*)table, grgb, g);
    else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(crop_resize_adjoint_kernel<APH_OUT_NCHW_NORM>, agrid, block, 0, st, gout, gscale, (const int*)table, grgb, g);
    else APH_LAUNCH(crop_resize_adjoint_kernel<APH_OUT_PATCH_F16>, agrid, block, 0, st, gout, gscale, (const int*)table, grgb, g);
    return aph_check_launch("aph_sample_bwd");
  }
  const size_t per = (size_t)g.S * 3 * n * n;
  float* dA = tmp;
  float* dB = tmp + per;
  (void)hipMemsetAsync(tmp, 0, sizeof(float) * 2 * per, st);
  const dim3 grid((n * n + 255) / 256, g.S);
  if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_NCHW_NORM>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  else APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_PATCH_F16>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  APH_LAUNCH(persp_adjoint_kernel, grid, block, 0, st, (const float*)dB, aug, dA, n);
  APH_LAUNCH(crop_resize_adjoint_kernel<APH_OUT_NCHW_RAW>, agrid, block, 0, st, (const float*)dA, gscale, (const int*)table, grgb, g);
  return aph_check_launch("aph_sample_bwd");
  APH_CATCH
}

int aph_patchify_f16(const float* x, int S, int R, int patch, void* out, void* stream_) {
  APH_TRY
  if (!x || !out || S < 1 || R < 1 || patch < 1 || R % patch) return aph_fail(APH_ERR_ARG, "aph_patchify_f16: bad argument");
  APH_LAUNCH(patchify_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream_, x, (half_t*)out, S, R, patch);
  return aph_check_launch("aph_patchify_f16");
  APH_CATCH
}

int aph_unpatchify_f32(const float* g, int S, int R, int patch, float gscale, float* out, void* stream_) {
  APH_TRY
  if (!g || !out || S < 1 || R < 1 || patch < 1 || R % patch) return aph_fail(APH_ERR_ARG, "aph_unpatchify_f32: bad argument");
  APH_LAUNCH(unpatchify_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream_, g, out, S, R, patch, gscale);
  return aph_check_launch("aph_unpatchify_f32");
  APH_CATCH
}

}  // extern "C"
