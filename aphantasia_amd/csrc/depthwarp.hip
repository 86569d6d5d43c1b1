// Depth warp of the illustrip frame loop: everything of the reference's
// depth/depth.py:41-84 (`resize`, `grid_warp`, `depthwarp`) and aphantasia/utils.py:137-147 (`triangle_blur`) EXCEPT the
// depth estimator itself (Depth-Anything-V2, depth.py:20-32: a third-party network, handed in by the caller as a depth
// map).  Once per frame on one [3,H,W] image: HBM-trivial gathers (11 MB at 720p); one thread per output pixel, the
// channel loop innermost so the coordinate arithmetic is done once.
//
// The arithmetic follows ATen's CPU kernels operation by operation (same formulas, fp32, no FMA contraction in the
// coordinate chain) because a bilinear sample of a noisy image moves by the image gradient times the coordinate error:
//   * linspace(-1, 1, n)[i]          = i < n/2 ? -1 + step*i : 1 - step*(n-1-i), step = 2/(n-1)      (RangeFactories)
//   * grid_sample(align_corners=True, padding_mode='reflection'): x = (g+1)/2*(n-1); reflect over [0, n-1]; clip; 4 taps
//   * interpolate(mode='bicubic', align_corners=True): src = dst*(in-1)/(out-1); taps floor-1..floor+2 index-clamped,
//     A = -0.75
#include "aph_device.h"
#include "aph_host.h"

#pragma clang fp contract(off)

namespace aph {

__device__ __forceinline__ float linspace_pm1(int i, int n) {
  if (n == 1) return -1.0f;
  const float step = 2.0f / (float)(n - 1);
  return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// ATen grid_sampler_compute_source_index, align_corners = true, reflection padding
__device__ __forceinline__ float warp_source_index(float g, int n) {
  float x = ((g + 1.0f) / 2.0f) * (float)(n - 1);
  const float span = (float)(n - 1);             // reflect_coordinates(x, 0, 2 (n-1))
  if (n == 1) return 0.0f;
  x = fabsf(x);
  const float extra = fmodf(x, span);
  const int flips = (int)floorf(x / span);
  x = (flips & 1) ? span - extra : extra;
  return fminf(fmaxf(x, 0.0f), (float)(n - 1));  // clip_coordinates
}

struct Taps4 { int i00, i01, i10, i11; float nw, ne, sw, se; };

__device__ __forceinline__ Taps4 bilinear_taps(float gx, float gy, int H, int W) {
  const float x = warp_source_index(gx, W), y = warp_source_index(gy, H);
  const float xw = floorf(x), yn = floorf(y);
  const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
  const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
  Taps4 t;
  t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
  // after reflection + clip (x0, y0) is inside; the +1 neighbours may be one past the edge: zero weight there (ATen masks them)
  const bool vx = x1 < W, vy = y1 < H;
  t.i00 = y0 * W + x0;
  t.i01 = vx ? y0 * W + x1 : -1;
  t.i10 = vy ? y1 * W + x0 : -1;
  t.i11 = (vx && vy) ? y1 * W + x1 : -1;
  return t;
}

__device__ __forceinline__ float sample4(const float* __restrict__ p, const Taps4& t) {
  float v = p[t.i00] * t.nw;
  if (t.i01 >= 0) v += p[t.i01] * t.ne;
  if (t.i10 >= 0) v += p[t.i10] * t.sw;
  if (t.i11 >= 0) v += p[t.i11] * t.se;
  return v;
}

// depth.py:44-66 grid_warp, one of its two passes:
//   PASS 0: grid + (centre - grid) * (depth - max(depth) * midpoint) * strength
//   PASS 1: grid + (centre - grid) * |centre - grid| * strength * dlens
template <int PASS>
__global__ void grid_warp_kernel(const float* __restrict__ src, const float* __restrict__ depth, const float* __restrict__ dmax,
                                 float* __restrict__ dst, int C, int H, int W, float strength, float cx, float cy, float midpoint,
                                 float dlens) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float gx = linspace_pm1(x, W), gy = linspace_pm1(y, H);
  const float dx = cx - gx, dy = cy - gy;
  float wx, wy;
  if (PASS == 0) {
    const float ds = depth[(size_t)y * W + x] - dmax[0] * midpoint;
    wx = gx + dx * ds * strength;
    wy = gy + dy * ds * strength;
  } else {
    const float lens = sqrtf(dx * dx + dy * dy);
    wx = gx + dx * lens * strength * dlens;
    wy = gy + dy * lens * strength * dlens;
  }
  const Taps4 t = bilinear_taps(wx, wy, H, W);
  const size_t plane = (size_t)H * W;
  for (int c = 0; c < C; ++c) dst[c * plane + (size_t)y * W + x] = sample4(src + c * plane, t);
}

// max over n floats: workgroup b -> out[b] (grid-stride); run again on the partials with one workgroup.  Deterministic.
__global__ void max_kernel(const float* __restrict__ p, size_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float m = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, p[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    out[blockIdx.x] = m;
  }
}

// utils.py:137-147 triangle_blur (horizontal pass, then vertical pass, reflect padding) followed by
// torch.lerp(src, blur, mix) (depth.py:75 uses mix = 0.5; mix = 1 is the plain blur)
struct BlurTaps { float w[9]; int k; };

__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void triangle_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, BlurTaps taps, float mix) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float* p = src + (size_t)blockIdx.z * H * W;
  const int pad = (taps.k - 1) / 2;
  float acc = 0.f;
  for (int j = 0; j < taps.k; ++j) {
    const float* row = p + (size_t)reflect_index(y + j - pad, H) * W;
    float h = 0.f;
    for (int i = 0; i < taps.k; ++i) h += row[reflect_index(x + i - pad, W)] * taps.w[i];
    acc += h * taps.w[j];
  }
  const float a = p[(size_t)y * W + x];
  const float diff = acc - a;
  dst[(size_t)blockIdx.z * H * W + (size_t)y * W + x] = mix < 0.5f ? a + mix * diff : acc - diff * (1.0f - mix);     // at::lerp
}

// F.interpolate(mode='bicubic', align_corners=True) (depth.py:41-42)
__device__ __forceinline__ void cubic_taps(int o, int n_in, float scale, int (&idx)[4], float (&w)[4]) {
  const float real = scale * (float)o;
  const float fl = floorf(real);
  int i0 = (int)fl;
  i0 = i0 < n_in - 1 ? i0 : n_in - 1;                 // guard_index_and_lambda
  float t = real - (float)i0;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float A = -0.75f;
  const float x1 = t, x2 = 1.0f - t;
  w[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 - 1 + k;
    idx[k] = i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i);
  }
}

__global__ void resize_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  int iy[4], ix[4];
  float wy[4], wx[4];
  cubic_taps(y, h, sy, iy, wy);
  cubic_taps(x, w, sx, ix, wx);
  const float* p = src + (size_t)blockIdx.z * h * w;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* row = p + (size_t)iy[j] * w;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r += row[ix[i]] * wx[i];
    acc += r * wy[j];
  }
  dst[(size_t)blockIdx.z * H * W + (size_t)y * W + x] = acc;
}

// dst = mul ? mul * flip_w(src) : flip_w(src)      (depth.py:77: depth * flip(infer(flip(image))))
__global__ void flip_w_kernel(const float* __restrict__ src, const float* __restrict__ mul, float* __restrict__ dst, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t o = (size_t)blockIdx.z * H * W + (size_t)y * W;
  const float v = src[o + (W - 1 - x)];
  dst[o + x] = mul ? mul[o + x] * v : v;
}

}  // namespace aph

using namespace aph;

extern "C" {

int aph_triangle_blur(const float* d_src, int C, int H, int W, int kernel_size, float power, float mix, float* d_dst, void* stream_) {
  APH_TRY
  if (!d_src || !d_dst || d_src == d_dst || C < 1 || H < 1 || W < 1 || kernel_size < 1 || kernel_size > 9 || !(kernel_size & 1) ||
      (kernel_size - 1) / 2 >= H || (kernel_size - 1) / 2 >= W)
    return aph_fail(APH_ERR_ARG, "aph_triangle_blur: bad argument (odd kernel_size <= 9, smaller than the image)");
  BlurTaps t;
  t.k = kernel_size;
  // torch.linspace(-1, 1, k + 2)[1:-1].abs().neg().add(1).pow(power), normalised -- in fp32 like the reference
  float sum = 0.f;
  const int n = kernel_size + 2;
  const float step = 2.0f / (float)(n - 1);
  for (int i = 0; i < kernel_size; ++i) {
    const int j = i + 1;
    const float v = j < n / 2 ? -1.0f + step * (float)j : 1.0f - step * (float)(n - 1 - j);
    t.w[i] = powf(1.0f - fabsf(v), power);
    sum += t.w[i];
  }
  for (int i = 0; i < kernel_size; ++i) t.w[i] /= sum;
  APH_LAUNCH(triangle_blur_kernel, dim3((W + 255) / 256, H, C), dim3(256), 0, (hipStream_t)stream_, d_src, d_dst, H, W, t, mix);
  return aph_check_launch("aph_triangle_blur");
  APH_CATCH
}

int aph_resize_bicubic(const float* d_src, int C, int h, int w, float* d_dst, int H, int W, void* stream_) {
  APH_TRY
  if (!d_src || !d_dst || d_src == d_dst || C < 1 || h < 1 || w < 1 || H < 1 || W < 1) return aph_fail(APH_ERR_ARG, "aph_resize_bicubic: bad argument");
  APH_LAUNCH(resize_bicubic_kernel, dim3((W + 255) / 256, H, C), dim3(256), 0, (hipStream_t)stream_, d_src, d_dst, h, w, H, W);
  return aph_check_launch("aph_resize_bicubic");
  APH_CATCH
}

int aph_flip_w(const float* d_src, const float* d_mul, int C, int H, int W, float* d_dst, void* stream_) {
  APH_TRY
  if (!d_src || !d_dst || d_src == d_dst || C < 1 || H < 1 || W < 1) return aph_fail(APH_ERR_ARG, "aph_flip_w: bad argument");
  APH_LAUNCH(flip_w_kernel, dim3((W + 255) / 256, H, C), dim3(256), 0, (hipStream_t)stream_, d_src, d_mul, d_dst, H, W);
  return aph_check_launch("aph_flip_w");
  APH_CATCH
}

int aph_grid_warp(const float* d_img, const float* d_depth, int C, int H, int W, float strength, float centre_x, float centre_y,
                  float midpoint, float dlens, float* d_ws, float* d_out, void* stream_) {
  APH_TRY
  if (!d_img || !d_depth || !d_ws || !d_out || d_img == d_out || d_ws == d_out || d_ws == d_img || C < 1 || H < 1 || W < 1)
    return aph_fail(APH_ERR_ARG, "aph_grid_warp: bad argument");
  hipStream_t st = (hipStream_t)stream_;
  float* tmp = d_ws;
  float* dmax = d_ws + (size_t)C * H * W;          // [0] the maximum, [1..] per-workgroup partials
  const dim3 grid((W + 255) / 256, H), block(256);
  const size_t n = (size_t)H * W;
  const int parts = (int)((n + 4095) / 4096 < 240 ? (n + 4095) / 4096 : 240);
  APH_LAUNCH(max_kernel, dim3(parts), dim3(256), 0, st, d_depth, n, dmax + 1);
  APH_LAUNCH(max_kernel, dim3(1), dim3(256), 0, st, (const float*)(dmax + 1), (size_t)parts, dmax);
  APH_LAUNCH(grid_warp_kernel<0>, grid, block, 0, st, d_img, d_depth, (const float*)dmax, tmp, C, H, W, strength, centre_x, centre_y, midpoint, dlens);
  APH_LAUNCH(grid_warp_kernel<1>, grid, block, 0, st, (const float*)tmp, d_depth, (const float*)dmax, d_out, C, H, W, strength, centre_x, centre_y,
             midpoint, dlens);
  return aph_check_launch("aph_grid_warp");
  APH_CATCH
}

}  // extern "C"
