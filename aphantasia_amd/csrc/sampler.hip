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
// Adjoint of the bilinear warps: gathers through the inverse maps (deterministic as well).
#include <cstring>
#include <cstdlib>

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

// four consecutive floats at 4-byte alignment: the compiler emits one global_load_dwordx4 (gfx950 handles the misalignment)
struct __attribute__((packed, aligned(4))) F4u { float v[4]; };

// "every ACTIVE lane of the wave satisfies p" (a speed choice only: the test interpreter decides per lane)
__device__ __forceinline__ bool wave_all(bool p) {
#ifdef APH_EMU
  return p;
#else
  return __builtin_amdgcn_ballot_w64(!p) == 0;
#endif
}

// v mod n for the wrap-tiled overscan frame (utils.py:165-167).  The padded frame is at most 2x the image (overmax), so
// v lies in [-n, 2n): one conditional correction instead of an integer division (there are eight of these per output
// pixel of the bicubic resize -- with `%` they were most of that kernel's instructions); the generic path is kept for safety.
__device__ __forceinline__ int wrap(int v, int n) {
  if (v < 0) v += n;
  else if (v >= n) v -= n;
  if (v < 0 || v >= n) { v %= n; if (v < 0) v += n; }
  return v;
}

// patch-major element offset of pixel (c,i,j) of cut s.  [r4] Inside a patch row the order is PIXEL-major, channel fastest:
//     k = ((i mod p) * p + (j mod p)) * 3 + c          (openai/CLIP's conv1.weight flattens as (c, i, j): aph_vit_set_weight permutes its
// columns once at load time -- the GEMM does not care in which order K is summed).  The three channels of a pixel are then 6 (f16) / 12 (f32)
// contiguous bytes: one access per bilinear tap / candidate in the warp adjoints instead of three 4-byte gathers at a 4 KiB stride (the kernels
// are bound by the L1's access rate), and one contiguous run per lane pair in the emit.  The patch side p is a power of two (checked on the
// host): shifts and masks instead of integer divisions in the per-pixel index arithmetic.
__device__ __forceinline__ size_t patch_index(int s, int c, int i, int j, int size, int p) {
  const int lp = __ffs(p) - 1, g = size >> lp;
  return ((size_t)s * g * g + (size_t)(i >> lp) * g + (j >> lp)) * (size_t)(3 << (2 * lp)) + (size_t)((((i & (p - 1)) << lp) + (j & (p - 1))) * 3 + c);
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

// backward-only layout: patch-major like APH_OUT_PATCH_F16 but the gradient elements are f16 (the ViT input-gradient
// written by aph_vit_backward_h, still carrying the loss scale)
template <int OUT>
__device__ __forceinline__ float gload(const void* __restrict__ g, size_t o) {
  if (OUT == APH_GRAD_PATCH_F16) return (float)reinterpret_cast<const half_t*>(g)[o];
  return reinterpret_cast<const float*>(g)[o];
}
template <int OUT>
struct is_patch { static constexpr bool v = OUT == APH_OUT_PATCH_F16 || OUT == APH_GRAD_PATCH_F16; };
// the three channels of one pixel of a patch-major gradient (contiguous: one 12-byte load for f32)
struct __attribute__((packed, aligned(4))) F3u { float v[3]; };
template <int OUT>
__device__ __forceinline__ void gload3(const void* __restrict__ g, size_t o, float q[3]) {
  if (OUT == APH_GRAD_PATCH_F16) {
    const half_t* h = reinterpret_cast<const half_t*>(g) + o;
    q[0] = (float)h[0]; q[1] = (float)h[1]; q[2] = (float)h[2];
  } else {
    const F3u t = *reinterpret_cast<const F3u*>(reinterpret_cast<const float*>(g) + o);
    q[0] = t.v[0]; q[1] = t.v[1]; q[2] = t.v[2];
  }
}
// internal layout of the per-cut scratch of the FORWARD augment chain (never crosses the C ABI): f32 [S][size][size][4] = (r, g, b, pad).
// Every bilinear tap of the perspective / rotation warps is then ONE 16-byte access for the three channels instead of three 4-byte
// ones in three planes (the warps are bound by L1 line accesses: crop + persp + rotate 311 -> 280 us at C2).  The ADJOINT chain keeps
// planar [S][3][size][size] scratch: its gathers are bound by L2 / fabric bytes, and the pad lane made it slower (566 -> 594 us).
constexpr int APH_SCRATCH_HWC4 = 8;
__device__ __forceinline__ size_t hwc4_index(int s, int i, int j, int size) { return (((size_t)s * size + i) * size + j) * 4; }

// gradient w.r.t. the un-normalised cut pixel (c,i,j) of cut s, read from `gout` in layout OUT
template <int OUT>
__device__ __forceinline__ float fetch_grad(const float* __restrict__ gout, int s, int c, int i, int j, int size, int patch) {
  if (OUT == APH_OUT_NCHW_RAW) return gout[(((size_t)s * 3 + c) * size + i) * size + j];
  if (OUT == APH_OUT_NCHW_NORM) return gout[(((size_t)s * 3 + c) * size + i) * size + j] / kClipStd[c];
  return gout[patch_index(s, c, i, j, size, patch)] / kClipStd[c];
}

// all three channels of cut pixel (i, j): one index computation (the patch-major index needs integer divisions)
template <int OUT>
__device__ __forceinline__ void fetch_grad3(const void* __restrict__ gout, int s, int i, int j, int size, int patch, float g[3]) {
  if (is_patch<OUT>::v) {
    const size_t o = patch_index(s, 0, i, j, size, patch);
    float q[3];
    gload3<OUT>(gout, o, q);
    g[0] = q[0] / kClipStd[0]; g[1] = q[1] / kClipStd[1]; g[2] = q[2] / kClipStd[2];
  } else {
    const size_t o = ((size_t)s * 3 * size + i) * size + j, nn = (size_t)size * size;
    g[0] = gload<OUT>(gout, o); g[1] = gload<OUT>(gout, o + nn); g[2] = gload<OUT>(gout, o + 2 * nn);
    if (OUT == APH_OUT_NCHW_NORM) { g[0] /= kClipStd[0]; g[1] /= kClipStd[1]; g[2] /= kClipStd[2]; }
  }
}
template <int OUT>
__device__ __forceinline__ void emit3(void* out, int s, int i, int j, int size, int patch, float v0, float v1, float v2) {
  if (OUT == APH_SCRATCH_HWC4) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + hwc4_index(s, i, j, size)) = f32x4{v0, v1, v2, 0.f};
  } else if (OUT == APH_OUT_PATCH_F16) {
    half_t* q = reinterpret_cast<half_t*>(out) + patch_index(s, 0, i, j, size, patch);
    q[0] = (half_t)((v0 - kClipMean[0]) / kClipStd[0]);
    q[1] = (half_t)((v1 - kClipMean[1]) / kClipStd[1]);
    q[2] = (half_t)((v2 - kClipMean[2]) / kClipStd[2]);
  } else if (OUT == APH_OUT_PATCH_F16_HILO) {
    // rows [hi (Kp) | lo (Kp)]: hi = f16(x), lo = f16(x - hi) -- the A operand of the split-precision patch embedding (aph_vit_forward_hilo)
    const int kp = 3 * patch * patch;
    const size_t o = patch_index(s, 0, i, j, size, patch);
    const size_t row = o / (size_t)kp;
    half_t* q = reinterpret_cast<half_t*>(out) + o + row * (size_t)kp;
    const float n0 = (v0 - kClipMean[0]) / kClipStd[0], n1 = (v1 - kClipMean[1]) / kClipStd[1], n2 = (v2 - kClipMean[2]) / kClipStd[2];
    const half_t h0 = (half_t)n0, h1 = (half_t)n1, h2 = (half_t)n2;
    q[0] = h0; q[1] = h1; q[2] = h2;
    q[kp] = (half_t)(n0 - (float)h0); q[kp + 1] = (half_t)(n1 - (float)h1); q[kp + 2] = (half_t)(n2 - (float)h2);
  } else {
    emit<OUT>(out, s, 0, i, j, size, patch, v0);
    emit<OUT>(out, s, 1, i, j, size, patch, v1);
    emit<OUT>(out, s, 2, i, j, size, patch, v2);
  }
}

// ---------------------------------------------------------------------------------
// crop + bicubic resize  (utils.py:248-249)
// ---------------------------------------------------------------------------------
struct CutBox { int cs, ox, oy; float scale; };
__device__ __forceinline__ CutBox load_cut(const int* __restrict__ table, int s, int size) {
  CutBox b;
  b.cs = table[3 * s]; b.ox = table[3 * s + 1]; b.oy = table[3 * s + 2];
  // area_pixel_compute_scale(align_corners=True): (in-1)/(out-1), source index = scale*dst, all fp32
  b.scale = size > 1 ? (float)(b.cs - 1) / (float)(size - 1) : 0.f;
  return b;
}

// bicubic value (all three channels) of resized-cut pixel (i, j): utils.py:248-249
__device__ __forceinline__ void bicubic3(const float* __restrict__ rgb, const Geom& g, const CutBox& b, int i, int j, float v[3]) {
  const float sy = b.scale * (float)i, sx = b.scale * (float)j;
  const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
  float wy[4], wx[4];
  cubic_w(sy - (float)y0, wy);
  cubic_w(sx - (float)x0, wx);
  int ry[4], rx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int yy = y0 - 1 + k; yy = yy < 0 ? 0 : (yy > b.cs - 1 ? b.cs - 1 : yy);   // clamp inside the cut
    int xx = x0 - 1 + k; xx = xx < 0 ? 0 : (xx > b.cs - 1 ? b.cs - 1 : xx);
    ry[k] = wrap(b.oy + yy - g.py0, g.H);                                       // tile_pad wrap (utils.py:165-167)
    rx[k] = wrap(b.ox + xx - g.px0, g.W);
  }
  // the four column taps are consecutive source pixels unless the clamp at the cut's edge or the wrap at the image's
  // edge intervenes: one 16-byte load per tap row (4-byte aligned) instead of four scalar gathers.  [r3] The choice is made per
  // WAVE: with a per-lane branch the compiler shared the first and last tap between the two paths and emitted dword + dwordx2 + dword
  // per tap row, each behind its own divergent branch (46 vector-memory instructions per wave and pixel; the kernel is bound by the
  // L1's access rate: TCP_TOTAL_CACHE_ACCESSES 150 M per launch at C2).
  if (wave_all(rx[3] == rx[0] + 3)) {
    F4u t[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a) t[c][a] = *reinterpret_cast<const F4u*>(rgb + ((size_t)c * g.H + ry[a]) * g.W + rx[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) acc += (t[c][a].v[0] * wx[0] + t[c][a].v[1] * wx[1] + t[c][a].v[2] * wx[2] + t[c][a].v[3] * wx[3]) * wy[a];
      v[c] = acc;
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = rgb + (size_t)c * g.H * g.W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* row = pl + (size_t)ry[a] * g.W;
      acc += (row[rx[0]] * wx[0] + row[rx[1]] * wx[1] + row[rx[2]] * wx[2] + row[rx[3]] * wx[3]) * wy[a];
    }
    v[c] = acc;
  }
}

// XCD-aware forward (speed only; any assignment is correct).  The image (11 MB at 720p) does not fit one XCD's 4 MB L2, and with
// the plain (x, y, cut) grid every XCD gathers from all of it: 487 MB of fabric fetch per launch for 11 MB of source.  Here the
// unit of work is (cut, group of 4 output rows); strip_list_kernel assigns every unit to the XCD that owns the 16-pixel image
// strip its source rows fall into (strips interleaved over the XCDs: strip t -> XCD t % 8, so every XCD sees centre and edge
// strips alike), and crop_resize_strips_kernel's workgroup b, which runs on XCD b % 8 (observed dispatch order), walks that
// XCD's list.  An XCD then touches ~1.4 / 8 of the image.
constexpr int kStripPx = 16;          // [r3] 32 -> 16: the 22.5 strips of a 720-row image left one XCD a third short of work (148 -> 145 us; 8: 143.5, 64: 170)
constexpr int kUnitRows = 4;              // output rows of one unit of work (8: 153 us against 145; 128-thread workgroups: 150)
constexpr int kStripSlots = 768;          // workgroups per XCD in crop_resize_strips_kernel

// lists: [8][cap] unit ids (cut * groups + row group), counts: [8]; one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void strip_list_kernel(const int* __restrict__ table, int* __restrict__ lists, int* __restrict__ counts, int cap, Geom g, int strip_px) {
  __shared__ int cnt[8];
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int groups = (g.size + kUnitRows - 1) / kUnitRows, units = g.S * groups;
  for (int u = threadIdx.x; u < units; u += blockDim.x) {
    const int s = u / groups, rg = u - s * groups;
    const CutBox b = load_cut(table, s, g.size);
    int i = rg * kUnitRows + kUnitRows / 2; i = i > g.size - 1 ? g.size - 1 : i;
    int yy = (int)floorf(b.scale * (float)i); yy = yy > b.cs - 1 ? b.cs - 1 : yy;
    const int yc = wrap(b.oy + yy - g.py0, g.H);
    const int xcd = (yc / strip_px) & 7;
    const int pos = atomicAdd(&cnt[xcd], 1);            // (order inside a list is irrelevant: units are independent)
    lists[xcd * cap + pos] = u;
  }
  __syncthreads();
  if (threadIdx.x < 8) counts[threadIdx.x] = cnt[threadIdx.x];
}

template <int OUT>
__global__ __launch_bounds__(256) void crop_resize_strips_kernel(const float* __restrict__ rgb, const int* __restrict__ table, void* __restrict__ out, Geom g,
                                                                 const int* __restrict__ lists, const int* __restrict__ counts, int cap) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int count = counts[xcd], groups = (g.size + kUnitRows - 1) / kUnitRows, n = g.size;
  for (int it = slot; it < count; it += nslot) {
    const int u = lists[xcd * cap + it];
    const int s = u / groups, rg = u - s * groups;
    const CutBox b = load_cut(table, s, n);
    for (int p = threadIdx.x; p < kUnitRows * n; p += blockDim.x) {
      const int di = p / n, j = p - di * n, i = rg * kUnitRows + di;
      if (i >= n) continue;
      float v[3];
      bicubic3(rgb, g, b, i, j, v);
      emit3<OUT>(out, s, i, j, n, g.patch, v[0], v[1], v[2]);
    }
  }
}

// Adjoint of crop_resize over all cuts -- deterministic gather, one 16x16 pixel tile per workgroup.
//   d rgb[y][x] = sum_s sum_{i,j} Wy_s[i][y - oy_s] Wx_s[j][x - ox_s] G_s[i][j]       (fixed order s = 0..S-1)
// 1. wave 0 culls the S cuts (x wrap-padding aliases) against the tile into an ordered LDS list;
// 2. per batch of 8 listed cuts, 256 threads build the 1-D tables: for each of the tile's 16 rows and 16
//    columns the (<= 4, for down-sampling cuts) output indices whose clamped cubic taps land on it, with the
//    forward's own fp32 weights and the separable gradient-layout offsets;
// 3. every pixel accumulates its <= 4x4 products per cut.  Work ~ the forward's 16 taps per output pixel.
// Up-sampling cuts (csize < size, only possible for images smaller than `size`) take the per-pixel generic path.
struct AdjEntry {
  int off[4];
  float w[4];
};

// separable offset parts of gradient element (i, j) in layout OUT (channel/cut base added by the caller)
template <int OUT>
__device__ __forceinline__ int grad_rowpart(int i, int size, int p) {
  if (is_patch<OUT>::v) { const int lp = __ffs(p) - 1; return (i >> lp) * (size >> lp) * (3 << (2 * lp)) + ((i & (p - 1)) << lp) * 3; }
  return i * size;
}
template <int OUT>
__device__ __forceinline__ int grad_colpart(int j, int /*size*/, int p) {
  if (is_patch<OUT>::v) { const int lp = __ffs(p) - 1; return (j >> lp) * (3 << (2 * lp)) + (j & (p - 1)) * 3; }
  return j;
}

// weight of output index i on crop-local source position q (sum over clamped taps; forward arithmetic)
__device__ __forceinline__ float tap_weight(float scale, int i, int cs, int q) {
  const float sy = scale * (float)i;
  const int y0 = (int)floorf(sy);
  float wv[4];
  cubic_w(sy - (float)y0, wv);
  float w = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int yy = y0 - 1 + k; yy = yy < 0 ? 0 : (yy > cs - 1 ? cs - 1 : yy);
    if (yy == q) w += wv[k];
  }
  return w;
}

// Per-cut 1-D tap tables, once per step: for every crop-local source position q of cut s and each axis, the (<= 4)
// output indices whose clamped cubic taps land on q, with the forward's own fp32 weights and the gradient-layout
// offsets.  tab[(s * 2 + axis) * maxcs + q]; entries of up-sampling cuts stay unused (generic path).
template <int OUT>
__global__ void tap_table_kernel(const int* __restrict__ table, AdjEntry* __restrict__ tab, int maxcs, Geom g) {
  const int s = blockIdx.z, isrow = blockIdx.y == 0;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int cs = table[3 * s];
  if (q >= cs || q >= maxcs) return;
  AdjEntry e;
#pragma unroll
  for (int a = 0; a < 4; ++a) { e.off[a] = 0; e.w[a] = 0.f; }
  const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
  if (scale >= 1.0f) {
    const float inv = 1.0f / scale;
    int lo = (int)floorf((float)(q - 2) * inv) - 1, hi = (int)floorf((float)(q + 2) * inv) + 1;
    lo = lo < 0 ? 0 : lo; hi = hi > g.size - 1 ? g.size - 1 : hi;
    int n = 0;
    for (int i = lo; i <= hi && n < 4; ++i) {
      const float w = tap_weight(scale, i, cs, q);
      if (w != 0.f || n > 0) {             // contiguous run starting at the first non-zero
        e.w[n] = w;
        e.off[n] = isrow ? grad_rowpart<OUT>(i, g.size, g.patch) : grad_colpart<OUT>(i, g.size, g.patch);
        ++n;
      }
    }
  }
  tab[((size_t)s * 2 + (isrow ? 0 : 1)) * maxcs + q] = e;
}

template <int OUT>
__global__ __launch_bounds__(256) void crop_resize_adjoint_kernel(const void* __restrict__ gout, float gscale,
                                                                  const int* __restrict__ table, float* __restrict__ grgb, Geom g,
                                                                  const AdjEntry* __restrict__ tab, int maxcs) {
  constexpr int MAXV = 1024, NB = 8;
  __shared__ int vlist[MAXV];
  __shared__ int vcount;
  __shared__ AdjEntry ent[NB][32];
  __shared__ int vinfo[NB][3];     // cut index, generic-path flag, longest run of outputs per source position
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  // XCD-aware tile order (speed only; any order is correct).  Workgroup b runs on XCD b % 8 (observed dispatch order).  XCD k takes
  // the tile rows k, k + 8, k + 16, ...: a gradient row of a cut lands on 1-3 image rows, so almost every gradient line is then
  // gathered by ONE XCD's L2 instead of all eight (the plain 2-D grid measured 816 MB of fabric fetch per launch for a 114 MB
  // gradient, L2 hit rate 0.29), while every XCD still sees the same mix of centre and edge rows (contiguous bands were slower:
  // edge bands are covered by half as many cuts).  486 -> 375 us at the headline size.  Workgroups beyond an XCD's share exit.
  const int ntx_ = (g.W + 15) / 16, nty_ = (g.H + 15) / 16;
  const int xcd_ = blockIdx.x & 7, idx_ = blockIdx.x >> 3;
  const int lrow_ = idx_ / ntx_, bx_ = idx_ - lrow_ * ntx_, by_ = lrow_ * 8 + xcd_;
  if (by_ >= nty_) return;
  const int x = bx_ * 16 + tx, y = by_ * 16 + ty;
  const bool live = x < g.W && y < g.H;
  const int nay = (g.Hp + g.H - 1) / g.H, nax = (g.Wp + g.W - 1) / g.W;     // aliases per axis (1 without overscan)
  const int nvirt = g.S * nay * nax;
  const int ty0 = by_ * 16, tx0 = bx_ * 16;
  const int cchan = is_patch<OUT>::v ? 1 : g.size * g.size;       // channel stride of the gradient layout (patch-major: channel fastest)
  const int ccut = is_patch<OUT>::v ? (g.size / g.patch) * (g.size / g.patch) * 3 * g.patch * g.patch : 3 * g.size * g.size;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  for (int vbase = 0; vbase < nvirt; vbase += MAXV) {
    // ---- 1. ordered compaction by wave 0
    __syncthreads();
    if (threadIdx.x < 64) {
      int count = 0;
      const int vend = nvirt - vbase < MAXV ? nvirt - vbase : MAXV;
      for (int v0 = 0; v0 < vend; v0 += 64) {
        const int v = vbase + v0 + threadIdx.x;
        bool hit = false;
        if (v0 + (int)threadIdx.x < vend) {
          const int s = v / (nay * nax), al = v - s * (nay * nax), ay = al / nax, ax = al - ay * nax;
          const int cs = table[3 * s], ox = table[3 * s + 1], oy = table[3 * s + 2];
          // alias coordinates of the tile's first/last row and column (a wrapping tile is not culled on that axis)
          // (the LAST LIVE row / column: a tile taller or wider than the whole image must not wrap its end into the middle of it --
          // images under 16 pixels on a side lost every cut that misses their first rows)
          const int ylast = ty0 + 15 < g.H - 1 ? ty0 + 15 : g.H - 1, xlast = tx0 + 15 < g.W - 1 ? tx0 + 15 : g.W - 1;
          const int Ya = wrap(ty0 + g.py0, g.H) + ay * g.H, Yb = wrap(ylast + g.py0, g.H) + ay * g.H;
          const int Xa = wrap(tx0 + g.px0, g.W) + ax * g.W, Xb = wrap(xlast + g.px0, g.W) + ax * g.W;
          const bool yhit = Yb < Ya ? true : (Yb >= oy && Ya < oy + cs);
          const bool xhit = Xb < Xa ? true : (Xb >= ox && Xa < ox + cs);
          hit = yhit && xhit;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) vlist[count + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = v;
        count += __popcll(m);
      }
      if (threadIdx.x == 0) vcount = count;
    }
    __syncthreads();
    const int nlist = vcount;
    for (int b0 = 0; b0 < nlist; b0 += NB) {
      // ---- 2. tables for up to NB cuts: thread -> (cut vb, row/col idx)
      {
        const int vb = threadIdx.x >> 5, idx = threadIdx.x & 31;
        AdjEntry e;
#pragma unroll
        for (int a = 0; a < 4; ++a) { e.off[a] = 0; e.w[a] = 0.f; }
        if (b0 + vb < nlist) {
          const int v = vlist[b0 + vb];
          const int s = v / (nay * nax), al = v - s * (nay * nax), ay = al / nax, ax = al - ay * nax;
          const int cs = table[3 * s], ox = table[3 * s + 1], oy = table[3 * s + 2];
          const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
          const bool generic = !(scale >= 1.0f);
          // a source position lies inside the 4-tap windows of at most floor(4 / scale) + 1 outputs per axis (<= 4 entries):
          // wave-uniform loop bounds instead of 4 x 4 mostly-zero products for the (common) strongly down-sampling cuts
          int run = (int)floorf(4.0f / (scale > 1.0f ? scale : 1.0f)) + 1;
          run = run > 4 ? 4 : run;
          if (idx == 0) { vinfo[vb][0] = v; vinfo[vb][1] = generic ? 1 : 0; vinfo[vb][2] = run; }
          if (!generic) {
            const bool isrow = idx < 16;
            const int q = isrow ? wrap(ty0 + idx + g.py0, g.H) + ay * g.H - oy : wrap(tx0 + (idx - 16) + g.px0, g.W) + ax * g.W - ox;
            const int lim = isrow ? g.Hp : g.Wp;
            const int absq = q + (isrow ? oy : ox);
            if (q >= 0 && q < cs && q < maxcs && absq < lim) e = tab[((size_t)s * 2 + (isrow ? 0 : 1)) * maxcs + q];
          }
        } else if (idx == 0) { vinfo[vb][0] = -1; vinfo[vb][1] = 0; vinfo[vb][2] = 0; }
        ent[vb][idx] = e;
      }
      __syncthreads();
      // ---- 3. accumulate
      for (int vb = 0; vb < NB; ++vb) {
        const int v = vinfo[vb][0];
        if (v < 0) break;
        const int s = v / (nay * nax);
        if (vinfo[vb][1]) {
          // generic per-pixel path (up-sampling cut)
          const int al = v - s * (nay * nax), ay = al / nax, ax = al - ay * nax;
          const int cs = table[3 * s], ox = table[3 * s + 1], oy = table[3 * s + 2];
          const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
          const int Y = wrap(y + g.py0, g.H) + ay * g.H, X = wrap(x + g.px0, g.W) + ax * g.W;
          const int yc = Y - oy, xc = X - ox;
          if (live && Y < g.Hp && X < g.Wp && yc >= 0 && yc < cs && xc >= 0 && xc < cs) {
            const size_t gb = (size_t)s * ccut;
            for (int i = 0; i < g.size; ++i) {
              const float wy = tap_weight(scale, i, cs, yc);
              if (wy == 0.f) continue;
              for (int j = 0; j < g.size; ++j) {
                const float wx = tap_weight(scale, j, cs, xc);
                if (wx == 0.f) continue;
                const int o = grad_rowpart<OUT>(i, g.size, g.patch) + grad_colpart<OUT>(j, g.size, g.patch);
                acc0 += wy * wx * gload<OUT>(gout, gb + o);
                acc1 += wy * wx * gload<OUT>(gout, gb + o + cchan);
                acc2 += wy * wx * gload<OUT>(gout, gb + o + 2 * cchan);
              }
            }
          }
          continue;
        }
        const AdjEntry re = ent[vb][ty], ce = ent[vb][16 + tx];
        if (re.w[0] == 0.f && re.w[1] == 0.f) continue;     // (a run starts with its first non-zero weight)
        const size_t gb = (size_t)s * ccut;
        const int run = vinfo[vb][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (a >= run) break;
          if (re.w[a] == 0.f) continue;
#pragma unroll
          for (int bq = 0; bq < 4; ++bq) {
            if (bq >= run) break;
            if (ce.w[bq] == 0.f) continue;
            const float w = re.w[a] * ce.w[bq];
            const int o = re.off[a] + ce.off[bq];
            acc0 += w * gload<OUT>(gout, gb + o);
            acc1 += w * gload<OUT>(gout, gb + o + cchan);
            acc2 += w * gload<OUT>(gout, gb + o + 2 * cchan);
          }
        }
      }
      __syncthreads();
    }
  }
  if (live) {
    const size_t HW = (size_t)g.H * g.W, o = (size_t)y * g.W + x;
    const float k0 = OUT == APH_OUT_NCHW_RAW ? gscale : gscale / kClipStd[0];
    const float k1 = OUT == APH_OUT_NCHW_RAW ? gscale : gscale / kClipStd[1];
    const float k2 = OUT == APH_OUT_NCHW_RAW ? gscale : gscale / kClipStd[2];
    grgb[o] = acc0 * k0;
    grgb[HW + o] = acc1 * k1;
    grgb[2 * HW + o] = acc2 * k2;
  }
}

// ---------------------------------------------------------------------------------
// [r3] Crop / resize adjoint, SEPARABLE and row-block stationary (frames without wrap padding: --align uniform / central).
//
//   d img[oy + q][ox + p] += sum_i Wy[i -> q] * ( sum_j Wx[j -> p] * G[i][j] )        per cut, Wy / Wx = the per-cut 1-D tap tables
//
// The gather kernel above visits every (pixel, covering cut) pair with up to 16 gathers x 3 channels and is bound by that per-pair
// skeleton (332 us at C2).  Here a workgroup owns RB image rows of ONE channel across the whole width, every thread owns CPT columns and
// keeps their RB accumulators in registers.  The covering cuts are walked in index order (deterministic, no atomics) in batches of NBC:
//   phase 0  the batch's cut boxes -> LDS; per (cut, four image rows) the union of the <= 8 gradient rows their taps come from, with
//            one weight per image row (QuadRow)
//   phase 1  column pass: U[b][j][4 qq .. 4 qq + 3] = sum_r W[r][.] * G_b[i_lo + r][j] for all `size` columns of each cut (coalesced
//            along j; a gradient row is read once per four image rows it feeds, the four results leave as one 16-byte LDS write)
//   phase 2  row pass: acc[q][x] += sum_b wx[b] * U[b][j_b][q], the RB rows of a column tap fetched as 16-byte LDS reads
// Up-sampling cuts (cs < size: never at 1280x720) take the per-pixel generic path of the gather kernel.
// ---------------------------------------------------------------------------------
struct __attribute__((aligned(16))) QuadRow {      // one gradient row of the union behind four consecutive image rows of a cut
  float w[4];        // its weight on each of the four image rows
  int off;           // gradient-layout row offset, -1 = unused
  int pad[3];
};
// inverse of grad_rowpart
template <int OUT>
__device__ __forceinline__ int grad_row_of_off(int off, int size, int p) {
  if (is_patch<OUT>::v) {
    const int lp = __ffs(p) - 1, rs = (size >> lp) * (3 << (2 * lp));
    const int ip = off / rs;
    return (ip << lp) + (off - ip * rs) / (3 << lp);
  }
  return off / size;
}
constexpr int ADJ_NBC = 12;         // cuts per batch (the launcher lowers it when LDS is short)
template <int OUT>
__device__ __forceinline__ int grad_col_of_off(int off, int p) {         // inverse of grad_colpart
  if (is_patch<OUT>::v) {
    const int lp = __ffs(p) - 1;
    const int jp = ((off >> (2 * lp)) * 43) >> 7;      // / 3 for values < 128 (at most size / patch = 7 .. 14 patch columns)
    return (jp << lp) + (((off - jp * (3 << (2 * lp))) * 43) >> 7);      // 3 (j mod p) < 128 as well (p <= 32)
  }
  return off;
}

template <int OUT, int RBQ, int CPT>
__global__ __launch_bounds__(768) void crop_adjoint_rows_kernel(const void* __restrict__ gout, float gscale, const int* __restrict__ table,
                                                                 float* __restrict__ grgb, Geom g, const AdjEntry* __restrict__ tab, int maxcs,
                                                                 int RB, int NBC, int dbg, int XW, int center_out) {
  // [r4] XW: columns per workgroup; blockIdx.z selects the column segment [x0, x0 + XW) (frames wider than 768 threads x 3 columns: the
  // 3840-wide C4 frame is two segments; a segment culls the cuts that do not reach it)
  constexpr int RBP = RBQ * 4, MAXV = 512;
  APH_DYN_SMEM(smem);
  float* U = reinterpret_cast<float*>(smem);                                   // [NBC][size][RBP]
  QuadRow* qtab2 = reinterpret_cast<QuadRow*>(U + (size_t)NBC * g.size * RBP); // [2][NBC][RBQ][8]: the <= 8 gradient rows behind four image rows
  int* binfo2 = reinterpret_cast<int*>(qtab2 + 2 * NBC * RBQ * 8);             // [2][NBC][4] = s (-1: none), cs, ox, oy (cs < 0: generic cut)
  int* vlist = binfo2 + 2 * NBC * 4;                                           // [MAXV]
  int* vbox = vlist + MAXV;                                                    // [MAXV][3] = cs, ox, oy of the listed cuts
  int* vcount = vbox + 3 * MAXV;
  const int tid = threadIdx.x, nthr = blockDim.x;
  // [r6] center_out: workgroup i of a (channel, segment) takes row block centre + i / 2 (i even) or centre - (i + 1) / 2 (i odd): with random crops the
  // middle rows of the frame are covered by the most cuts (1.2x the mean, 3.6x the edge blocks), and a grid with more workgroups than CUs
  // should start its longest items first
  const int nrb = gridDim.x, bi = blockIdx.x, rbi = center_out ? ((bi & 1) ? nrb / 2 - (bi + 1) / 2 : nrb / 2 + bi / 2) : bi;
  const int c = blockIdx.y, y0 = rbi * RB;
  const int x0 = blockIdx.z * XW, x1 = (x0 + XW < g.W ? x0 + XW : g.W);
  const int rows = g.H - y0 < RB ? g.H - y0 : RB;
  const int cchan = is_patch<OUT>::v ? 1 : g.size * g.size;       // channel stride of the gradient layout (patch-major: channel fastest)
  const int ccut = is_patch<OUT>::v ? (g.size / g.patch) * (g.size / g.patch) * 3 * g.patch * g.patch : 3 * g.size * g.size;
  f32x4 acc[CPT][RBQ];
#pragma unroll
  for (int i = 0; i < CPT; ++i)
#pragma unroll
    for (int k = 0; k < RBQ; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < NBC * g.size * RBP; i += nthr) U[i] = 0.f;              // (the padding rows q >= RB stay zero for good)
  for (int vbase = 0; vbase < g.S; vbase += MAXV) {
    // ---- ordered list of the cuts that touch this row block (wave 0, ballot compaction)
    __syncthreads();
    if (tid < 64) {
      int count = 0;
      const int vend = g.S - vbase < MAXV ? g.S - vbase : MAXV;
      for (int v0 = 0; v0 < vend; v0 += 64) {
        const int s = vbase + v0 + tid;
        bool hit = false;
        int cs = 0, ox = 0, oy = 0;
        if (v0 + tid < vend) {
          cs = table[3 * s]; ox = table[3 * s + 1]; oy = table[3 * s + 2];
          hit = oy < y0 + rows && oy + cs > y0 && ox < x1 && ox + cs > x0;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) {
          const int pos = count + __popcll(m & ((1ull << tid) - 1ull));
          vlist[pos] = s; vbox[3 * pos] = cs; vbox[3 * pos + 1] = ox; vbox[3 * pos + 2] = oy;
        }
        count += __popcll(m);
      }
      if (tid == 0) *vcount = count;
    }
    __syncthreads();
    const int nlist = *vcount;
    // ---- phase 0 (of batch b0, into table buffer `buf`): boxes of the batch, and per (cut, four image rows) the union of the gradient
    // rows their taps come from.  Image row y of a down-sampling cut (scale >= 1) is touched by output rows i with floor(i scale) in
    // [y - 2, y + 1] (clamped taps land on rows that are in that set anyway), so four consecutive image rows draw on i in
    // [(y - 2) / scale, (y + 5) / scale): at most 8 rows.  Thread (b, qq, r) merges the four per-row tap entries into row r of that
    // union: gradient offset + 4 weights.
    // It runs one batch AHEAD, on the last 256 threads during the row pass of the batch before: those threads own the fewest columns
    // (W = 1280 on 768 threads x 2 columns: the last four waves have one), so the two dependent table loads cost the batch nothing.
    constexpr int P0_THREADS = 256;
    const int p0_first = nthr - P0_THREADS;
    auto phase0 = [&](int b0, int buf) {
      if (tid < p0_first) return;
      QuadRow* qt = qtab2 + buf * NBC * RBQ * 8;
      int* bi = binfo2 + buf * NBC * 4;
      for (int pt = tid - p0_first; pt < NBC * RBQ * 8; pt += P0_THREADS) {
        const int b = pt / (RBQ * 8), qq = (pt >> 3) % RBQ, r = pt & 7;
        QuadRow qr;
        qr.off = -1; qr.w[0] = qr.w[1] = qr.w[2] = qr.w[3] = 0.f;
        if (b0 + b < nlist) {
          const int s = vlist[b0 + b];
          const int cs = vbox[3 * (b0 + b)], ox = vbox[3 * (b0 + b) + 1], oy = vbox[3 * (b0 + b) + 2];      // (kept by the list build: one dependent load less)
          const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
          const bool generic = !(scale >= 1.0f);
          if (qq == 0 && r == 0) { bi[4 * b] = s; bi[4 * b + 1] = generic ? -cs : cs; bi[4 * b + 2] = ox; bi[4 * b + 3] = oy; }
          if (!generic) {
            AdjEntry e[4];
            int i0[4], ilo = 1 << 30;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int q = 4 * qq + k, yc = y0 + q - oy;
              const bool live = q < rows && yc >= 0 && yc < cs && yc < maxcs;
              if (live) e[k] = tab[((size_t)s * 2) * maxcs + yc];
              i0[k] = 1 << 30;
              if (live && (e[k].w[0] != 0.f || e[k].w[1] != 0.f || e[k].w[2] != 0.f || e[k].w[3] != 0.f)) i0[k] = grad_row_of_off<OUT>(e[k].off[0], g.size, g.patch);
              else { e[k].w[0] = e[k].w[1] = e[k].w[2] = e[k].w[3] = 0.f; }
              ilo = i0[k] < ilo ? i0[k] : ilo;
            }
            if (ilo < (1 << 30)) {
              const int i = ilo + r;
              bool any = false;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                float w = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) w += (i0[k] + a == i) ? e[k].w[a] : 0.f;
                qr.w[k] = w;
                any = any || w != 0.f;
              }
              if (any && i < g.size) qr.off = grad_rowpart<OUT>(i, g.size, g.patch);
            }
          }
        } else if (qq == 0 && r == 0) bi[4 * b] = -1;
        qt[pt] = qr;
      }
    };
    phase0(0, 0);
    __syncthreads();
    for (int b0 = 0, cur = 0; b0 < nlist; b0 += NBC, cur ^= 1) {
      const QuadRow* qtab = qtab2 + cur * NBC * RBQ * 8;
      const int* binfo = binfo2 + cur * NBC * 4;
      // ---- phase 1: column pass into U[b][j][4 qq .. 4 qq + 3]: one wave per (cut, four rows), lanes across the cut's columns; the
      // gradient rows of the union are read once for the four image rows they feed, and the four results leave as one 16-byte LDS write
      // (the scalar writes of a per-row pass are 8-way bank conflicted under the 16-byte-aligned column pitch the row pass needs)
      const int nb = nlist - b0 < NBC ? nlist - b0 : NBC;
      if (!(dbg & 1)) {
        // one quad per wave and trip, its 32 gathers issued before the first is used: the pass is bound by memory latency
        const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6, nquad = nb * RBQ;
        constexpr int JT = 4;                                      // column trips of 64 lanes: size <= 256 (checked by the launcher)
        unsigned colj[JT];                                         // 32-bit lane offsets against a scalar row base: one address register per column trip
#pragma unroll
        for (int m = 0; m < JT; ++m) { const int j = lane + 64 * m; colj[m] = (unsigned)grad_colpart<OUT>(j < g.size ? j : 0, g.size, g.patch); }
        for (int pq = wv; pq < nquad; pq += nwv) {
          float v[JT][8];
          const int b = pq / RBQ, qq = pq - b * RBQ;
          const size_t gb = (size_t)wave_uniform(binfo[4 * b]) * ccut + (size_t)c * cchan;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int off = wave_uniform(qtab[pq * 8 + r].off);
            const size_t rowbase = gb + (size_t)(off >= 0 ? off : 0);
#pragma unroll
            for (int m = 0; m < JT; ++m) {
              float x = 0.f;
              if (off >= 0 && lane + 64 * m < g.size) {
                if (OUT == APH_GRAD_PATCH_F16) x = (float)(reinterpret_cast<const half_t*>(gout) + rowbase)[colj[m]];
                else x = (reinterpret_cast<const float*>(gout) + rowbase)[colj[m]];
              }
              v[m][r] = x;
            }
          }
          f32x4 u[JT];
#pragma unroll
          for (int m = 0; m < JT; ++m) u[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(qtab[pq * 8 + r].w);
#pragma unroll
            for (int m = 0; m < JT; ++m) u[m] += w * v[m][r];
          }
#pragma unroll
          for (int m = 0; m < JT; ++m) {
            const int j = lane + 64 * m;
            if (j < g.size) *reinterpret_cast<f32x4*>(U + ((size_t)b * g.size + j) * RBP + 4 * qq) = u[m];
          }
        }
      }
      __syncthreads();
      if (b0 + NBC < nlist) phase0(b0 + NBC, cur ^ 1);
      // ---- phase 2: row pass, cuts in list order
      // ([r5] measured and not adopted, profiles/r05_sampler_pipelined_ab.txt: the entries of groups of two cuts loaded two groups ahead into a
      // register ring -- unconditional clamped loads, partial vmcnt waits in the ISA -- 293.6 -> 301.5 us: this pass does not wait for L2; and the
      // taps of a (column, cut) read back to back without the per-tap zero-weight skips: 296.6 -> 310.4 us -- every skipped tap is three LDS reads)
      if (!(dbg & 2)) {
        // four cuts x CPT columns at a time: the first offset and the four weights of every column-tap entry (20 of its 32 bytes) are
        // loaded together, then accumulated per column in list order
#pragma unroll
        for (int bh = 0; bh < ADJ_NBC; bh += 4) {
          if (bh >= nb) break;
          f32x4 cw[CPT][4];
          int coff[CPT][4];
#pragma unroll
          for (int i = 0; i < CPT; ++i) {
            const int x = x0 + i * nthr + tid;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              const int b = bh + bb;
              coff[i][bb] = -1;
              cw[i][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (b < nb && x < x1) {
                const int s = binfo[4 * b], csx = binfo[4 * b + 1], p = x - binfo[4 * b + 2];
                if (csx > 0 && p >= 0 && p < csx && p < maxcs) {
                  const AdjEntry* ep = tab + ((size_t)s * 2 + 1) * maxcs + p;
                  coff[i][bb] = ep->off[0];
                  cw[i][bb] = *reinterpret_cast<const f32x4*>(ep->w);
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              if (coff[i][bb] < 0) continue;
              // the taps of an entry are CONSECUTIVE output columns (tap_table_kernel: a contiguous run from the first non-zero weight)
              const float* u0 = U + ((size_t)(bh + bb) * g.size + grad_col_of_off<OUT>(coff[i][bb], g.patch)) * RBP;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                if (cw[i][bb][t] == 0.f) continue;
                const float* up = u0 + t * RBP;
#pragma unroll
                for (int k = 0; k < RBQ; ++k) acc[i][k] += cw[i][bb][t] * *reinterpret_cast<const f32x4*>(up + 4 * k);
              }
            }
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
          const int x = x0 + i * nthr + tid;
          if (x >= x1) continue;
          // up-sampling cuts (cs < size; none at 1280x720): the per-pixel generic path of crop_resize_adjoint_kernel.  Kept out of the
          // unrolled loop above (a loop the compiler does not unroll would index ce[] at run time and move it to scratch memory); the
          // sum of such a cut is added after the batch's table-driven cuts -- a fixed order all the same.
          for (int b = 0; b < nb; ++b) {
            const int csx = binfo[4 * b + 1];
            if (csx >= 0) continue;
            const int s = binfo[4 * b], oy = binfo[4 * b + 3], cs = -csx, p = x - binfo[4 * b + 2];
            if (p < 0 || p >= cs) continue;
            const float scale = g.size > 1 ? (float)(cs - 1) / (float)(g.size - 1) : 0.f;
            const size_t gb = (size_t)s * ccut + (size_t)c * cchan;
#pragma unroll
            for (int q = 0; q < RBP; ++q) {            // (fully unrolled: a run-time index into acc would move it to scratch memory)
              const int yc = y0 + q - oy;
              if (q >= rows || yc < 0 || yc >= cs) continue;
              float sum = 0.f;
              for (int ii = 0; ii < g.size; ++ii) {
                const float wy = tap_weight(scale, ii, cs, yc);
                if (wy == 0.f) continue;
                for (int jj = 0; jj < g.size; ++jj) {
                  const float wx = tap_weight(scale, jj, cs, p);
                  if (wx == 0.f) continue;
                  sum += wy * wx * gload<OUT>(gout, gb + grad_rowpart<OUT>(ii, g.size, g.patch) + grad_colpart<OUT>(jj, g.size, g.patch));
                }
              }
              acc[i][q >> 2][q & 3] += sum;
            }
          }
        }
      }
      __syncthreads();
    }
  }
  const float kc = OUT == APH_OUT_NCHW_RAW ? gscale : gscale / kClipStd[c];
  const size_t HW = (size_t)g.H * g.W;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int x = x0 + i * nthr + tid;
    if (x >= x1) continue;
#pragma unroll
    for (int k = 0; k < RBQ; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 4 * k + r;
        if (q < rows) grgb[(size_t)c * HW + (size_t)(y0 + q) * g.W + x] = acc[i][k][r] * kc;
      }
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

// sampled value (three channels of one HWC4 cut image) times sampled ones-mask (fill = 0); ERASE: source pixels inside the
// erase rectangle read as 0
// warp_block_note [r3]: a workgroup of the four warp kernels covers 32 x 8 pixels (it was 64 x 4).  Under a rotation the taps of a
// 64 x 4 strip cross ~32 gradient rows and use a few pixels of every 128-byte line they touch, and the neighbouring strips that use
// the rest run on other XCDs: rotate_emit_adjoint measured 353 MB of L2 misses per launch for a 114 MB gradient.  A squarer tile
// shares fewer lines with its neighbours: augment adjoints 178-188 -> 154-162 us, forward chain 225 -> 217 us (16 x 16 measured the same).
template <bool ERASE>
__device__ __forceinline__ void warp_gather3(const float* __restrict__ src, const Tap& t, int n, const float* __restrict__ a, float v[3]) {
  // branch-free: out-of-range taps read a clamped address with weight 0, so the four 16-byte loads issue together
  float m = 0.f;
  v[0] = v[1] = v[2] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = t.y0 + dy, xx = t.x0 + dx;
      const bool in = yy >= 0 && yy < n && xx >= 0 && xx < n;
      const float w = in ? (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0) : 0.f;
      const int yc = yy < 0 ? 0 : (yy > n - 1 ? n - 1 : yy), xc = xx < 0 ? 0 : (xx > n - 1 ? n - 1 : xx);
      const f32x4 sv = *reinterpret_cast<const f32x4*>(src + ((size_t)yc * n + xc) * 4);
      m += w;
      const float we = (ERASE && in_rect(a, yc, xc)) ? 0.f : w;
      v[0] += we * sv[0]; v[1] += we * sv[1]; v[2] += we * sv[2];
    }
  v[0] *= m; v[1] *= m; v[2] *= m;
}

// stage 1: RandomPerspective for the cuts that drew it (A -> B, both HWC4); other cuts are skipped
__global__ void persp_kernel(const float* __restrict__ A, const float* __restrict__ aug, float* __restrict__ Bo, int n) {
  const int s = blockIdx.z;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  if (a[8] == 0.f) return;
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (i >= n || j >= n) return;
  const Tap t = persp_tap(a, i, j, n);
  float v[3];
  warp_gather3<false>(A + hwc4_index(s, 0, 0, n), t, n, a, v);
  *reinterpret_cast<f32x4*>(Bo + hwc4_index(s, i, j, n)) = f32x4{v[0], v[1], v[2], 0.f};
}

// stage 2: RandomErasing (read-side) + rotation + normalise + emit
template <int OUT>
__global__ void rotate_emit_kernel(const float* __restrict__ A, const float* __restrict__ Bi, const float* __restrict__ aug,
                                   void* __restrict__ out, int n, int patch) {
  const int s = blockIdx.z;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (i >= n || j >= n) return;
  const float* src = (a[8] != 0.f ? Bi : A) + hwc4_index(s, 0, 0, n);
  float v[3];
  if (a[15] != 0.f) {
    const Tap t = rot_tap(a[13], a[14], i, j, n);
    warp_gather3<true>(src, t, n, a, v);
  } else {
    const f32x4 q = *reinterpret_cast<const f32x4*>(src + ((size_t)i * n + j) * 4);
    const bool er = in_rect(a, i, j);
    v[0] = er ? 0.f : q[0]; v[1] = er ? 0.f : q[1]; v[2] = er ? 0.f : q[2];
  }
  emit3<OUT>(out, s, i, j, n, patch, v[0], v[1], v[2]);
}

// sum of the in-bounds bilinear weights (= the sampled ones-mask of torchvision's fill handling)
__device__ __forceinline__ float tap_mask(const Tap& t, int n) {
  float m = 0.f;
  if (t.y0 >= 0 && t.y0 < n) { if (t.x0 >= 0 && t.x0 < n) m += t.wx0 * t.wy0; if (t.x0 + 1 >= 0 && t.x0 + 1 < n) m += t.wx1 * t.wy0; }
  if (t.y0 + 1 >= 0 && t.y0 + 1 < n) { if (t.x0 >= 0 && t.x0 < n) m += t.wx0 * t.wy1; if (t.x0 + 1 >= 0 && t.x0 + 1 < n) m += t.wx1 * t.wy1; }
  return m;
}
// weight with which output pixel's footprint `t` reads source pixel (py, px); 0 if it does not
__device__ __forceinline__ float tap_hits(const Tap& t, int py, int px) {
  const int dy = py - t.y0, dx = px - t.x0;
  if (dy < 0 || dy > 1 || dx < 0 || dx > 1) return 0.f;
  return (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0);
}

// Adjoint of stage 2 as a GATHER (deterministic, no atomics): thread = source pixel p of the pre-rotation
// cut; the output pixels whose bilinear footprint contains p lie in the inverse-rotated 2x2 square around p.
// Each candidate's footprint is re-derived with the forward's own arithmetic.
template <int OUT>
__global__ void rotate_emit_adjoint_kernel(const void* __restrict__ gout, const float* __restrict__ aug,
                                           float* __restrict__ dA, float* __restrict__ dB, int n, int patch) {
  const int s = blockIdx.z;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (py >= n || px >= n) return;
  float* dst = a[8] != 0.f ? dB : dA;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (!in_rect(a, py, px)) {
    if (a[15] != 0.f) {
      const float cs = a[13], sn = a[14], c = 0.5f * (float)(n - 1);
      // forward: (ix, iy) = Rot (q - c) + c with Rot = [[cs, sn], [-sn, cs]]  ->  q = Rot^T (p - c) + c
      const float ux = (float)px - c, uy = (float)py - c;
      const float qx = cs * ux - sn * uy + c, qy = sn * ux + cs * uy + c;
      const float rad = fabsf(cs) + fabsf(sn) + 0.02f;
      int j0 = (int)ceilf(qx - rad), j1 = (int)floorf(qx + rad), i0 = (int)ceilf(qy - rad), i1 = (int)floorf(qy + rad);
      if (rad <= 1.45f) {
        // a rotation: the candidates fit a 3 x 3 box.  Branch-free: a miss gets weight 0 and a clamped address, the 27 gathers issue
        // together (one memory round trip instead of one per candidate).  [r3] The weight of candidate (i, j) is the TENT form of the
        // forward's bilinear footprint -- max(0, 1 - |ix - px|) * max(0, 1 - |iy - py|), with (ix, iy) from the forward's own grid
        // arithmetic (rot_tap / make_tap), its row and column terms computed once per box row / column -- times the sampled ones-mask
        // clamp(min(ix + 1, n - ix), 0, 1) * (same in y): identical to tap_hits * tap_mask up to one rounding of (1 - frac), at a third
        // of the instructions.  (Time unchanged: the kernel is bound by the L1's access rate -- 27 scalar gathers per pixel, about 46
        // cache accesses per gather instruction whatever the wave's pixel footprint, 64 x 1 and 16 x 4 measured alike; only a
        // channel-interleaved gradient layout would cut that.)
        const float fn = (float)n, hn = 0.5f * fn, ka = cs / hn, kb = sn / hn;
        const int lp = is_patch<OUT>::v ? __ffs(patch) - 1 : 0, pg = n >> lp;
        float gxj[3], gyj[3], gxi[3], gyi[3];
        size_t rowo[3], colo[3];
        bool iok[3], jok[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int i = i0 + t, j = j0 + t;
          iok[t] = i >= 0 && i <= n - 1 && i <= i1;
          jok[t] = j >= 0 && j <= n - 1 && j <= j1;
          const int ic = i < 0 ? 0 : (i > n - 1 ? n - 1 : i), jc = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
          const float x = -fn * 0.5f + 0.5f + (float)jc, y = -fn * 0.5f + 0.5f + (float)ic;
          gxj[t] = x * ka; gyj[t] = x * -kb;
          gxi[t] = y * kb; gyi[t] = y * ka;
          colo[t] = is_patch<OUT>::v ? (size_t)(jc >> lp) * (size_t)(3 << (2 * lp)) + (size_t)(jc & (patch - 1)) * 3 : (size_t)jc;
          rowo[t] = is_patch<OUT>::v ? (size_t)(ic >> lp) * pg * (size_t)(3 << (2 * lp)) + (size_t)((ic & (patch - 1)) << lp) * 3 : (size_t)ic * n;
        }
        const size_t base = is_patch<OUT>::v ? (size_t)s * pg * pg * (size_t)(3 << (2 * lp)) : (size_t)s * 3 * n * n;
        const float fpx = (float)px, fpy = (float)py;
        float wm[9];
        size_t off[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          const int a3 = d / 3, b3 = d % 3;
          const float gx = gxj[b3] + gxi[a3], gy = gyj[b3] + gyi[a3];
          const float ix = ((gx + 1.f) * fn - 1.f) * 0.5f, iy = ((gy + 1.f) * fn - 1.f) * 0.5f;
          const float wx = fmaxf(0.f, 1.f - fabsf(ix - fpx)), wy = fmaxf(0.f, 1.f - fabsf(iy - fpy));
          const float mx = fminf(fmaxf(fminf(ix + 1.f, fn - ix), 0.f), 1.f), my = fminf(fmaxf(fminf(iy + 1.f, fn - iy), 0.f), 1.f);
          wm[d] = (iok[a3] && jok[b3]) ? (wx * wy) * (mx * my) : 0.f;
          off[d] = base + rowo[a3] + colo[b3];
        }
        const size_t cstride = (size_t)n * n;          // (planar layouts; the patch-major layouts hold a pixel's channels contiguously)
        float gv[9][3];
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          if (is_patch<OUT>::v) gload3<OUT>(gout, off[d], gv[d]);
          else { gv[d][0] = gload<OUT>(gout, off[d]); gv[d][1] = gload<OUT>(gout, off[d] + cstride); gv[d][2] = gload<OUT>(gout, off[d] + 2 * cstride); }
        }
#pragma unroll
        for (int d = 0; d < 9; ++d) { g0 += wm[d] * gv[d][0]; g1 += wm[d] * gv[d][1]; g2 += wm[d] * gv[d][2]; }
        if (OUT != APH_OUT_NCHW_RAW) { g0 /= kClipStd[0]; g1 /= kClipStd[1]; g2 /= kClipStd[2]; }
      } else {
        j0 = j0 < 0 ? 0 : j0; i0 = i0 < 0 ? 0 : i0; j1 = j1 > n - 1 ? n - 1 : j1; i1 = i1 > n - 1 ? n - 1 : i1;
        for (int i = i0; i <= i1; ++i)
          for (int j = j0; j <= j1; ++j) {
            const Tap t = rot_tap(cs, sn, i, j, n);
            const float w = tap_hits(t, py, px);
            if (w == 0.f) continue;
            const float wmm = w * tap_mask(t, n);
            float gq[3];
            fetch_grad3<OUT>(gout, s, i, j, n, patch, gq);
            g0 += wmm * gq[0]; g1 += wmm * gq[1]; g2 += wmm * gq[2];
          }
      }
    } else {
      float gq[3];
      fetch_grad3<OUT>(gout, s, py, px, n, patch, gq);
      g0 = gq[0]; g1 = gq[1]; g2 = gq[2];
    }
  }
  const size_t pl = (size_t)s * 3 * n * n, pix = (size_t)py * n + px;
  dst[pl + pix] = g0;
  dst[pl + (size_t)n * n + pix] = g1;
  dst[pl + 2 * (size_t)n * n + pix] = g2;
}

// Adjoint of stage 1 (perspective) as a gather: dB -> dA (in place of the cut's slot in dA).  Candidates =
// bounding box of the inverse homography applied to the 2x2 square around p.
__global__ void persp_adjoint_kernel(const float* __restrict__ dB, const float* __restrict__ aug, float* __restrict__ dA, int n) {
  const int s = blockIdx.z;
  const float* a = aug + (size_t)s * APH_AUG_STRIDE;
  if (a[8] == 0.f) return;
  const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (py >= n || px >= n) return;
    // forward: (u, v) = H (x, y), x = j + .5, y = i + .5, source index = (u - .5, v - .5);  adj(H) maps back
  const float m00 = a[4] - a[5] * a[7], m01 = a[2] * a[7] - a[1], m02 = a[1] * a[5] - a[2] * a[4];
  const float m10 = a[5] * a[6] - a[3], m11 = a[0] - a[2] * a[6], m12 = a[2] * a[3] - a[0] * a[5];
  const float m20 = a[3] * a[7] - a[4] * a[6], m21 = a[1] * a[6] - a[0] * a[7], m22 = a[0] * a[4] - a[1] * a[3];
  float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float u = (float)px + 0.5f + ((k & 1) ? 1.01f : -1.01f), v = (float)py + 0.5f + ((k & 2) ? 1.01f : -1.01f);
    const float d = m20 * u + m21 * v + m22;
    const float xx = (m00 * u + m01 * v + m02) / d - 0.5f, yy = (m10 * u + m11 * v + m12) / d - 0.5f;
    xmin = fminf(xmin, xx); xmax = fmaxf(xmax, xx); ymin = fminf(ymin, yy); ymax = fmaxf(ymax, yy);
  }
  int j0 = (int)ceilf(xmin - 0.05f), j1 = (int)floorf(xmax + 0.05f), i0 = (int)ceilf(ymin - 0.05f), i1 = (int)floorf(ymax + 0.05f);
  j0 = j0 < 0 ? 0 : j0; i0 = i0 < 0 ? 0 : i0; j1 = j1 > n - 1 ? n - 1 : j1; i1 = i1 > n - 1 ? n - 1 : i1;
  if (!(xmax - xmin < 64.f && ymax - ymin < 64.f)) { j0 = 0; i0 = 0; j1 = n - 1; i1 = n - 1; }   // degenerate map: exhaustive
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  const size_t pl = (size_t)s * 3 * n * n, nn = (size_t)n * n;
  for (int i = i0; i <= i1; ++i)
    for (int j = j0; j <= j1; ++j) {
      const Tap t = persp_tap(a, i, j, n);
      const float w = tap_hits(t, py, px);
      if (w == 0.f) continue;
      const float wm = w * tap_mask(t, n);
      const size_t o = pl + (size_t)i * n + j;
      g0 += wm * dB[o];
      g1 += wm * dB[o + nn];
      g2 += wm * dB[o + 2 * nn];
    }
  const size_t pix = (size_t)py * n + px;
  dA[pl + pix] = g0;
  dA[pl + nn + pix] = g1;
  dA[pl + 2 * nn + pix] = g2;
}

// ---------------------------------------------------------------------------------
// illustrip's frame_transform (illustrip.py:130-138): T.functional.affine(img, angle, shift, scale, shear, fill=0,
// BILINEAR) of a whole [C,H,W] image, once per frame.  m = the 2x3 INVERSE affine matrix (host, torchvision's
// _get_inverse_affine_matrix); grid = [x, y, 1] . (m^T / (0.5 W, 0.5 H)) over the centred base grid, bilinear, zeros
// padding, ones-mask fill -- the same sampler arithmetic as the per-cut rotation above.
// ---------------------------------------------------------------------------------
struct Affine6 { float m[6]; };

__global__ void frame_affine_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, Affine6 a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float bx = -(float)W * 0.5f + 0.5f + (float)x, by = -(float)H * 0.5f + 0.5f + (float)y;
  const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
  const float gx = bx * (a.m[0] / hw) + by * (a.m[1] / hw) + (a.m[2] / hw);
  const float gy = bx * (a.m[3] / hh) + by * (a.m[4] / hh) + (a.m[5] / hh);
  const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  float w[4];
  int off[4];
  float mask = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
    const bool in = xx >= 0 && xx < W && yy >= 0 && yy < H;
    w[k] = in ? ((k & 1) ? wx1 : wx0) * ((k >> 1) ? wy1 : wy0) : 0.f;
    off[k] = in ? yy * W + xx : 0;
    mask += w[k];
  }
  for (int c = 0; c < C; ++c) {
    const float* pl = src + (size_t)c * H * W;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += w[k] * pl[off[k]];
    dst[((size_t)c * H + y) * W + x] = v * mask;
  }
}

// ---------------------------------------------------------------------------------
// layout conversion for caller-made batches (model.encode_image(x) on an NCHW tensor)
// ---------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ x, half_t* __restrict__ out, int S, int R, int p, int hilo) {
  const size_t n = (size_t)S * 3 * R * R;
  const size_t kp = (size_t)3 * p * p;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx % R, i = (idx / R) % R, c = (idx / ((size_t)R * R)) % 3, s = idx / ((size_t)3 * R * R);
    const size_t o = patch_index(s, c, i, j, R, p);
    const half_t h = (half_t)x[idx];
    if (!hilo) { out[o] = h; continue; }
    const size_t q = o + (o / kp) * kp;                       // rows [hi | lo]
    out[q] = h;
    out[q + kp] = (half_t)(x[idx] - (float)h);
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

static int check_geom(const aph_sample_geom* g, int out_mode, const char* who, int max_mode = 2) {
  if (!g) return aph_fail(APH_ERR_ARG, "%s: null geometry", who);
  if (out_mode == APH_OUT_PATCH_F16_HILO && max_mode == 2) max_mode = APH_OUT_PATCH_F16_HILO;      // forward only: the split-precision patch rows
  if (g->S < 1 || g->size < 1 || g->H < 1 || g->W < 1 || g->Hp < g->H || g->Wp < g->W)
    return aph_fail(APH_ERR_ARG, "%s: bad geometry S=%d size=%d H=%d W=%d Hp=%d Wp=%d", who, g->S, g->size, g->H, g->W, g->Hp, g->Wp);
  if (out_mode < 0 || out_mode > max_mode || (out_mode == APH_GRAD_PATCH_F16 && max_mode != APH_GRAD_PATCH_F16))
    return aph_fail(APH_ERR_ARG, "%s: bad out_mode %d", who, out_mode);
  if (out_mode >= APH_OUT_PATCH_F16 && (g->patch < 1 || g->size % g->patch || (g->patch & (g->patch - 1))))
    return aph_fail(APH_ERR_ARG, "%s: size %d not divisible by patch %d, or patch not a power of two", who, g->size, g->patch);
  return APH_OK;
}

// Workspace of one sampler call (caller-owned, aph_sample_ws_bytes): [per-cut 1-D tap tables of the crop adjoint |
// cut scratch A | cut scratch B], the scratch planes only with geometric augmentation.  Nothing is allocated or freed
// in a launch path, so a captured hipGraph never holds a pointer the library could invalidate.
namespace {
size_t tab_bytes(const Geom& g) {
  const size_t maxcs = (size_t)(g.Hp < g.Wp ? g.Hp : g.Wp);    // a cut fits the (padded) image; csize <= min(H, W) upstream (utils.py:231,245)
  return ((size_t)g.S * 2 * maxcs * sizeof(AdjEntry) + 255) & ~(size_t)255;
}
size_t scratch_floats(const Geom& g) { return (size_t)g.S * 4 * g.size * g.size; }     // HWC4 in the forward; the adjoint uses 3/4 of it, planar
// per-XCD unit lists of the forward: [8][cap] + [8] ints
int strip_cap(const Geom& g) { return g.S * ((g.size + kUnitRows - 1) / kUnitRows); }
size_t strip_bytes(const Geom& g) { return ((size_t)(8 * (size_t)strip_cap(g) + 8) * sizeof(int) + 255) & ~(size_t)255; }

template <int OUT>
void launch_crop_resize(const float* rgb, const int* table, void* out, const Geom& g, void* ws, hipStream_t st) {
  // ws: [tap tables | strip lists | ...]
  int* lists = reinterpret_cast<int*>(static_cast<char*>(ws) + tab_bytes(g));
  int* counts = lists + 8 * (size_t)strip_cap(g);
  APH_LAUNCH(strip_list_kernel, dim3(1), dim3(1024), 0, st, table, lists, counts, strip_cap(g), g, kStripPx);
  APH_LAUNCH(crop_resize_strips_kernel<OUT>, dim3(8 * kStripSlots), dim3(256), 0, st, rgb, table, out, g, (const int*)lists, (const int*)counts, strip_cap(g));
}

// 1: always the round-2 gather kernel (aph_crop_adjoint_set_gather: A/B runs and the equivalence tests)
inline int& crop_adjoint_gather() {
  static int v = 0;
  return v;
}

// launch shape of the separable crop adjoint: 0 / -1 = automatic (aph_crop_adjoint_set_shape: the sweep of tools/exp/crop_adjoint_sweep.py)
struct CropAdjointShape { int rb = 0, cpt = 0, nbc = 0, nseg = 0, order = -1; };
inline CropAdjointShape& crop_adjoint_shape() {
  static CropAdjointShape v;
  return v;
}
inline int gemm_like_cu_count() {          // CUs of the current device (256 on MI355X); the interpreter build says 3 so that the centre-out order is exercised
#ifdef APH_EMU
  return 3;
#else
  static const int n = [] { int dev = 0, cu = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev); return cu > 0 ? cu : 256; }();
  return n;
#endif
}

template <int OUT>
int launch_crop_adjoint(const void* gout, float gscale, const int* table, float* grgb, const Geom& g, AdjEntry* tab, hipStream_t st) {
  const int maxcs = g.Hp < g.Wp ? g.Hp : g.Wp;
  APH_LAUNCH(tap_table_kernel<OUT>, dim3((maxcs + 127) / 128, 2, g.S), dim3(128), 0, st, table, tab, maxcs, g);
  // [r3] frames without wrap padding: the separable row-block kernel (APH_CROP_ADJOINT=gather keeps the round-2 gather kernel for A/B runs)
  if (!crop_adjoint_gather() && g.Hp == g.H && g.Wp == g.W && g.py0 == 0 && g.px0 == 0 && g.W <= 4 * 2304 && g.size <= 256) {
    // column segments of at most 2304 (768 threads x 3 columns); [r4] wider frames (C4: 3840) take several segments per row block
    int nseg = (g.W + 2303) / 2304;
    const CropAdjointShape& ov = crop_adjoint_shape();
    if (ov.nseg > 0) nseg = ov.nseg;
    const int xw = ((g.W + nseg - 1) / nseg + 3) & ~3;
    int cpt = xw <= 1536 ? 2 : 3;                             // columns per thread, at most 768 threads (three waves per SIMD: 168 VGPRs)
    if (ov.cpt > 0) cpt = ov.cpt;
    int nthr = (((xw + cpt - 1) / cpt) + 63) / 64 * 64;
    nthr = nthr < 256 ? 256 : nthr;
    if (nthr > 768) return aph_fail(APH_ERR_ARG, "crop adjoint: %d columns per segment need more than 768 threads x %d columns", xw, cpt);
    // rows per workgroup: about one workgroup per CU over rows x 3 channels (85 row blocks), 12 or 16 accumulator rows per column.
    // [r6] launch-shape sweep (tools/exp/crop_adjoint_sweep.py, profiles/r06_crop_adjoint_sweep.txt): at 1280x720 / 190 cuts the automatic
    // 9 rows x 2 columns x 1 segment (240 workgroups) is the fastest of 180 shapes (293 us with the tap tables; every finer split of rows or
    // columns loses: the pass pays per (column, cut) entry, and more rows per workgroup amortise it); at 3840x2160 / 95 cuts 16 rows x
    // 3 columns x 2 segments takes 495 us against 581 for the 12 rows segmented frames used to be held at (the 168-VGPR concern of round 4
    // did not materialise: no scratch in the ISA)
    int rb = (g.H + 84) / 85;
    rb = rb < 4 ? 4 : (rb > 16 ? 16 : rb);
    if (ov.rb > 0) rb = ov.rb;
    int rbq = rb <= 12 ? 3 : 4;
#if defined(APH_EXPERIMENTS) && !defined(APH_EMU)
    if (ov.rb > 0) rbq = (rb + 3) / 4;
#endif
    const int rbp = rbq * 4;
    int nbc = ov.nbc > 0 && ov.nbc <= ADJ_NBC ? ov.nbc : ADJ_NBC;
    auto lds = [&](int n) { return (size_t)n * g.size * rbp * 4 + 2 * (size_t)n * rbq * 8 * sizeof(QuadRow) + 2 * (size_t)n * 16 + 512 * 16 + 16; };
    while (nbc > 1 && lds(nbc) > 150 * 1024) --nbc;
    if (lds(nbc) <= 150 * 1024) {
      const dim3 rgrid((g.H + rb - 1) / rb, 3, nseg);
#ifdef APH_EXPERIMENTS      /* ablation hook (1 = no column pass, 2 = no row pass: WRONG gradients) -- only in a -DAPH_EXPERIMENTS build, never in the product library */
      static const int dbg = [] { const char* e = getenv("APH_SAMPLER_DBG"); return e ? atoi(e) : 0; }();
#else
      constexpr int dbg = 0;
#endif
      const size_t smem = lds(nbc);
      const int center_out = ov.order >= 0 ? ov.order : ((int)(rgrid.x * 3 * nseg) > gemm_like_cu_count() ? 1 : 0);
#define APH_ADJ_ROWS(RBQ, CPT)                                                                                                              \
  do {                                                                                                                                       \
    APH_ALLOW_SMEM((crop_adjoint_rows_kernel<OUT, RBQ, CPT>), 150 * 1024);                                                                   \
    APH_LAUNCH((crop_adjoint_rows_kernel<OUT, RBQ, CPT>), rgrid, dim3(nthr), smem, st, gout, gscale, table, grgb, g, (const AdjEntry*)tab, maxcs, rb, nbc, dbg, xw, center_out); \
  } while (0)
#if defined(APH_EXPERIMENTS) && !defined(APH_EMU)       /* every (rows, columns per thread) shape, for the launch-shape sweep of tools/exp/crop_adjoint_sweep.py (GPU only: the interpreter build skips them) */
      if (rbq == 1 && cpt == 1) APH_ADJ_ROWS(1, 1);
      else if (rbq == 1 && cpt == 2) APH_ADJ_ROWS(1, 2);
      else if (rbq == 1) APH_ADJ_ROWS(1, 3);
      else if (rbq == 2 && cpt == 1) APH_ADJ_ROWS(2, 1);
      else if (rbq == 2 && cpt == 2) APH_ADJ_ROWS(2, 2);
      else if (rbq == 2) APH_ADJ_ROWS(2, 3);
      else if (rbq == 3 && cpt == 1) APH_ADJ_ROWS(3, 1);
      else if (rbq == 4 && cpt == 1) APH_ADJ_ROWS(4, 1);
      else if (rbq == 5 && cpt == 2) APH_ADJ_ROWS(5, 2);
      else if (rbq == 5) APH_ADJ_ROWS(5, 3);
      else if (rbq == 6 && cpt == 2) APH_ADJ_ROWS(6, 2);
      else if (rbq >= 6) APH_ADJ_ROWS(6, 3);
      else
#endif
      if (rbq == 3 && cpt == 2) APH_ADJ_ROWS(3, 2);
      else if (rbq == 3) APH_ADJ_ROWS(3, 3);
      else if (cpt == 2) APH_ADJ_ROWS(4, 2);
      else APH_ADJ_ROWS(4, 3);
#undef APH_ADJ_ROWS
      return APH_OK;
    }
  }
  const dim3 agrid(8 * (((g.H + 15) / 16 + 7) / 8) * ((g.W + 15) / 16));        // 8 XCD shares of ceil(tile rows / 8) rows each (see the kernel's tile order)
  APH_LAUNCH(crop_resize_adjoint_kernel<OUT>, agrid, dim3(256), 0, st, gout, gscale, table, grgb, g, (const AdjEntry*)tab, maxcs);
  return APH_OK;
}
}  // namespace

extern "C" {

// test / measurement hook: launch shape of the separable crop adjoint (rows per workgroup, columns per thread, cuts per batch, column
// segments, row-block order 0 = top-down / 1 = centre-out); 0 (order: -1) = automatic.  rb beyond the shipped kernel shapes needs a -DAPH_EXPERIMENTS build.
int aph_crop_adjoint_set_shape(int rb, int cpt, int nbc, int nseg, int order) {
  CropAdjointShape& v = crop_adjoint_shape();
  v.rb = rb; v.cpt = cpt; v.nbc = nbc; v.nseg = nseg; v.order = order;
  return APH_OK;
}
// test / measurement hook: 1 = the crop adjoint always runs the gather kernel, 0 = automatic.  Returns the previous value.
int aph_crop_adjoint_set_gather(int on) {
  const int prev = crop_adjoint_gather();
  crop_adjoint_gather() = on ? 1 : 0;
  return prev;
}

size_t aph_sample_ws_bytes(const aph_sample_geom* gg, int with_aug) {
  if (!gg || gg->S < 1 || gg->size < 1 || gg->Hp < 1 || gg->Wp < 1) return 0;
  const Geom g = to_geom(gg);
  return tab_bytes(g) + strip_bytes(g) + (with_aug ? 2 * scratch_floats(g) * sizeof(float) : 0);
}

int aph_sample_fwd(const aph_sample_geom* gg, const float* rgb, const int32_t* table, const float* aug, void* ws,
                   void* out, int out_mode, void* stream_) {
  APH_TRY
  if (int e = check_geom(gg, out_mode, "aph_sample_fwd")) return e;
  if (!rgb || !table || !out || !ws) return aph_fail(APH_ERR_ARG, "aph_sample_fwd: null argument (the workspace of aph_sample_ws_bytes is required)");
  hipStream_t st = (hipStream_t)stream_;
  const Geom g = to_geom(gg);
  const int n = g.size;
  const dim3 grid((n + 31) / 32, (n + 7) / 8, g.S), block(256);       // thread = (column, row) of a cut: 32 x 8 pixels per workgroup (warp_block_note)
  if (!aug) {
    if (out_mode == APH_OUT_NCHW_RAW) launch_crop_resize<APH_OUT_NCHW_RAW>(rgb, (const int*)table, out, g, ws, st);
    else if (out_mode == APH_OUT_NCHW_NORM) launch_crop_resize<APH_OUT_NCHW_NORM>(rgb, (const int*)table, out, g, ws, st);
    else if (out_mode == APH_OUT_PATCH_F16) launch_crop_resize<APH_OUT_PATCH_F16>(rgb, (const int*)table, out, g, ws, st);
    else launch_crop_resize<APH_OUT_PATCH_F16_HILO>(rgb, (const int*)table, out, g, ws, st);
    return aph_check_launch("aph_sample_fwd");
  }
  float* A = reinterpret_cast<float*>(static_cast<char*>(ws) + tab_bytes(g) + strip_bytes(g));
  float* Bv = A + scratch_floats(g);
  // resized cut -> A; RandomPerspective for the cuts that drew it A -> B; RandomErasing + rotation + normalise (A or B) -> out
  launch_crop_resize<APH_SCRATCH_HWC4>(rgb, (const int*)table, (void*)A, g, ws, st);
  APH_LAUNCH(persp_kernel, grid, block, 0, st, (const float*)A, aug, Bv, n);
  if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(rotate_emit_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(rotate_emit_kernel<APH_OUT_NCHW_NORM>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  else if (out_mode == APH_OUT_PATCH_F16) APH_LAUNCH(rotate_emit_kernel<APH_OUT_PATCH_F16>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  else APH_LAUNCH(rotate_emit_kernel<APH_OUT_PATCH_F16_HILO>, grid, block, 0, st, (const float*)A, (const float*)Bv, aug, out, n, g.patch);
  return aph_check_launch("aph_sample_fwd");
  APH_CATCH
}

int aph_sample_bwd(const aph_sample_geom* gg, const void* gout, float gscale, const int32_t* table, const float* aug,
                   void* ws, float* grgb, int out_mode, void* stream_) {
  APH_TRY
  if (int e = check_geom(gg, out_mode, "aph_sample_bwd", APH_GRAD_PATCH_F16)) return e;
  if (!gout || !table || !grgb || !ws) return aph_fail(APH_ERR_ARG, "aph_sample_bwd: null argument (the workspace of aph_sample_ws_bytes is required)");
  hipStream_t st = (hipStream_t)stream_;
  const Geom g = to_geom(gg);
  const int n = g.size;
  const dim3 block(256);
  AdjEntry* tab = static_cast<AdjEntry*>(ws);
  if (!aug) {
    int rc;
    if (out_mode == APH_OUT_NCHW_RAW) rc = launch_crop_adjoint<APH_OUT_NCHW_RAW>(gout, gscale, (const int*)table, grgb, g, tab, st);
    else if (out_mode == APH_OUT_NCHW_NORM) rc = launch_crop_adjoint<APH_OUT_NCHW_NORM>(gout, gscale, (const int*)table, grgb, g, tab, st);
    else if (out_mode == APH_OUT_PATCH_F16) rc = launch_crop_adjoint<APH_OUT_PATCH_F16>(gout, gscale, (const int*)table, grgb, g, tab, st);
    else rc = launch_crop_adjoint<APH_GRAD_PATCH_F16>(gout, gscale, (const int*)table, grgb, g, tab, st);
    if (rc) return rc;
    return aph_check_launch("aph_sample_bwd");
  }
  float* dA = reinterpret_cast<float*>(static_cast<char*>(ws) + tab_bytes(g) + strip_bytes(g));
  float* dB = dA + scratch_floats(g);
  const dim3 grid((n + 31) / 32, (n + 7) / 8, g.S);
  if (out_mode == APH_OUT_NCHW_RAW) APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_NCHW_RAW>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  else if (out_mode == APH_OUT_NCHW_NORM) APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_NCHW_NORM>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  else if (out_mode == APH_OUT_PATCH_F16) APH_LAUNCH(rotate_emit_adjoint_kernel<APH_OUT_PATCH_F16>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  else APH_LAUNCH(rotate_emit_adjoint_kernel<APH_GRAD_PATCH_F16>, grid, block, 0, st, gout, aug, dA, dB, n, g.patch);
  APH_LAUNCH(persp_adjoint_kernel, grid, block, 0, st, (const float*)dB, aug, dA, n);
  if (int rc = launch_crop_adjoint<APH_OUT_NCHW_RAW>((const void*)dA, gscale, (const int*)table, grgb, g, tab, st)) return rc;
  return aph_check_launch("aph_sample_bwd");
  APH_CATCH
}

int aph_patchify_f16(const float* x, int S, int R, int patch, void* out, void* stream_) {
  APH_TRY
  if (!x || !out || S < 1 || R < 1 || patch < 1 || R % patch) return aph_fail(APH_ERR_ARG, "aph_patchify_f16: bad argument");
  APH_LAUNCH(patchify_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream_, x, (half_t*)out, S, R, patch, 0);
  return aph_check_launch("aph_patchify_f16");
  APH_CATCH
}

// the same into the split-precision rows [hi | lo] of aph_vit_forward_hilo: out f16 [S*(R/patch)^2, 2 * 3*patch*patch]
int aph_patchify_f16_hilo(const float* x, int S, int R, int patch, void* out, void* stream_) {
  APH_TRY
  if (!x || !out || S < 1 || R < 1 || patch < 1 || R % patch) return aph_fail(APH_ERR_ARG, "aph_patchify_f16_hilo: bad argument");
  APH_LAUNCH(patchify_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream_, x, (half_t*)out, S, R, patch, 1);
  return aph_check_launch("aph_patchify_f16_hilo");
  APH_CATCH
}

int aph_unpatchify_f32(const float* g, int S, int R, int patch, float gscale, float* out, void* stream_) {
  APH_TRY
  if (!g || !out || S < 1 || R < 1 || patch < 1 || R % patch) return aph_fail(APH_ERR_ARG, "aph_unpatchify_f32: bad argument");
  APH_LAUNCH(unpatchify_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream_, g, out, S, R, patch, gscale);
  return aph_check_launch("aph_unpatchify_f32");
  APH_CATCH
}

// illustrip.py:130-138 frame_transform: d_dst [C,H,W] = affine warp of d_src [C,H,W] (d_dst != d_src).
// inv_matrix6: row-major 2x3 inverse affine matrix as torchvision's _get_inverse_affine_matrix returns it.
int aph_frame_affine(const float* d_src, int C, int H, int W, const float* inv_matrix6, float* d_dst, void* stream_) {
  APH_TRY
  if (!d_src || !d_dst || !inv_matrix6 || d_src == d_dst || C < 1 || H < 1 || W < 1) return aph_fail(APH_ERR_ARG, "aph_frame_affine: bad argument");
  Affine6 a;
  for (int i = 0; i < 6; ++i) a.m[i] = inv_matrix6[i];
  APH_LAUNCH(frame_affine_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream_, d_src, d_dst, C, H, W, a);
  return aph_check_launch("aph_frame_affine");
  APH_CATCH
}

}  // extern "C"
