// Image parameteriser kernels (SURVEY.md K1-K4 and their adjoints).
//
//   spectrum params --x scale--> column C2C (length H) --> row C2R (length W) --> raw image
//   raw --(global unbiased std, contrast)--> 3x3 colour mix --> sigmoid --> rgb in (0,1)
//
// Replaces: aphantasia/image.py:164-175 (fft_image.inner), :21-28 (to_valid_rgb.inner),
//           :114-118 (pixel_image.inner).  All arithmetic fp32, reductions in fp64.
//
// FFT: Stockham autosort, mixed radix (2/3/4/5 specialised, any other prime <= 31 generic in registers, larger primes by direct sums),
// one workgroup per column tile / per row PAIR, whole sequence resident in LDS (ping-pong).
// The C2R / R2C row transforms process two real rows as one complex sequence
// (z = a + i b), which halves the work and keeps odd W legal.
// HBM-bound: 11 MB in, 11 MB intermediate (write + read), 11 MB out at 1280x720.
#include "aph_device.h"
#include "aph_host.h"

namespace aph {

struct Fft1D {
  int n, npass;
  int radix[14];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by SIGN * i
template <int SIGN>
__device__ __forceinline__ float2 cmuli(float2 a) {
  return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <int SIGN>
__device__ __forceinline__ float2 twiddle(const float2* __restrict__ tw, int t) {
  float2 w = tw[t];  // exp(+2 pi i t / N)
  if (SIGN < 0) w.y = -w.y;
  return w;
}

// One Stockham pass, radix R, over nseq sequences of length N held in LDS (src -> dst).
template <int SIGN, int R>
__device__ __forceinline__ void fft_pass(const float2* src, float2* dst, int N, int Ns, int nseq,
                                         const float2* __restrict__ tw) {
  const int nb = N / R;
  const int tscale = N / (Ns * R);
  for (int idx = threadIdx.x; idx < nb * nseq; idx += blockDim.x) {
    const int q = idx / nb, j = idx - q * nb;
    const int k = j % Ns;
    const float2* x = src + q * N;
    float2* y = dst + q * N + (j / Ns) * Ns * R + k;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = x[j + r * nb];
      if (r > 0) v[r] = cmul(v[r], twiddle<SIGN>(tw, k * r * tscale));
    }
    if (R == 2) {
      y[0] = cadd(v[0], v[1]);
      y[Ns] = csub(v[0], v[1]);
    } else if (R == 3) {
      const float2 s = cadd(v[1], v[2]);
      const float2 t = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
      float2 u = csub(v[1], v[2]);
      u = cmuli<SIGN>(make_float2(u.x * 0.86602540378443865f, u.y * 0.86602540378443865f));
      y[0] = cadd(v[0], s);
      y[Ns] = cadd(t, u);
      y[2 * Ns] = csub(t, u);
    } else if (R == 4) {
      const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
      const float2 a2 = cadd(v[1], v[3]), a3 = cmuli<SIGN>(csub(v[1], v[3]));
      y[0] = cadd(a0, a2);
      y[Ns] = cadd(a1, a3);
      y[2 * Ns] = csub(a0, a2);
      y[3 * Ns] = csub(a1, a3);
    } else {  // small odd radix: direct DFT with table roots of unity
#pragma unroll
      for (int qq = 0; qq < R; ++qq) {
        float2 acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], twiddle<SIGN>(tw, ((qq * r) % R) * nb)));
        y[qq * Ns] = acc;
      }
    }
  }
}

// generic prime radix (7..31): runtime R, values kept in a local array
template <int SIGN>
__device__ void fft_pass_generic(const float2* src, float2* dst, int N, int R, int Ns, int nseq,
                                 const float2* __restrict__ tw) {
  const int nb = N / R;
  const int tscale = N / (Ns * R);
  for (int idx = threadIdx.x; idx < nb * nseq; idx += blockDim.x) {
    const int q = idx / nb, j = idx - q * nb;
    const int k = j % Ns;
    const float2* x = src + q * N;
    float2* y = dst + q * N + (j / Ns) * Ns * R + k;
    float2 v[32];
    for (int r = 0; r < R; ++r) {
      v[r] = x[j + r * nb];
      if (r > 0) v[r] = cmul(v[r], twiddle<SIGN>(tw, k * r * tscale));
    }
    for (int qq = 0; qq < R; ++qq) {
      float2 acc = v[0];
      for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], twiddle<SIGN>(tw, ((qq * r) % R) * nb)));
      y[qq * Ns] = acc;
    }
  }
}

// any larger prime radix (37, 41, ... up to N itself): the R inputs of a butterfly do not fit registers, so a work item is ONE
// output: y[qq] = sum_r x[j + r nb] w^(k r tscale + (qq r mod R) nb), the two twiddles folded into one table index.  O(N R) per pass
// instead of O(N log N): sizes with a big prime factor (1366 = 2 x 683) are transformed correctly, just not fast -- torch.fft takes
// any size, and a --size the reference accepts must not be refused here.
template <int SIGN>
__device__ void fft_pass_large(const float2* src, float2* dst, int N, int R, int Ns, int nseq, const float2* __restrict__ tw) {
  const int nb = N / R;
  const int tscale = N / (Ns * R);
  for (int idx = threadIdx.x; idx < N * nseq; idx += blockDim.x) {
    const int q = idx / N, o = idx - q * N;
    const int qq = o / nb, j = o - qq * nb;
    const int k = j % Ns;
    const float2* x = src + q * N;
    // (the R-term sum is kept in fp64: in fp32 its rounding grows with sqrt(R) -- 6e-6 on the image at R = 1307, found by tools/gpu_fuzz.py)
    double ar = x[j].x, ai = x[j].y;
    // twiddle index of term r: r (k tscale + qq nb) mod N  (N = R nb, so (qq r mod R) nb == qq r nb mod N): one modular add per term
    const int step = (int)(((long long)k * tscale + (long long)qq * nb) % N);
    int t = 0;
    for (int r = 1; r < R; ++r) {
      t += step; if (t >= N) t -= N;
      const float2 pr = cmul(x[j + r * nb], twiddle<SIGN>(tw, t));
      ar += pr.x; ai += pr.y;
    }
    dst[q * N + (j / Ns) * Ns * R + k + qq * Ns] = make_float2((float)ar, (float)ai);
  }
}

// Full transform of nseq LDS-resident sequences; returns the buffer holding the result.
template <int SIGN>
__device__ float2* fft_lds(float2* a, float2* b, const Fft1D& plan, int nseq, const float2* __restrict__ tw) {
  int Ns = 1;
  const int N = plan.n;
  for (int p = 0; p < plan.npass; ++p) {
    const int R = plan.radix[p];
    switch (R) {
      case 2: fft_pass<SIGN, 2>(a, b, N, Ns, nseq, tw); break;
      case 3: fft_pass<SIGN, 3>(a, b, N, Ns, nseq, tw); break;
      case 4: fft_pass<SIGN, 4>(a, b, N, Ns, nseq, tw); break;
      case 5: fft_pass<SIGN, 5>(a, b, N, Ns, nseq, tw); break;
      default:
        if (R <= 31) fft_pass_generic<SIGN>(a, b, N, R, Ns, nseq, tw);
        else fft_pass_large<SIGN>(a, b, N, R, Ns, nseq, tw);
        break;
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
    Ns *= R;
  }
  return a;
}

constexpr int kLoadBatch = 8;      // items per thread whose global loads are in flight together in the passes' load phases

// ---------------------------------------------------------------------------------
// column pass, forward synthesis: tmp[c][y][kx] = sum_ky scale*params[c][ky][kx] e^{+2 pi i ky y/H}
// ---------------------------------------------------------------------------------
__global__ void fft_col_synth_kernel(const float2* __restrict__ params, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float2* __restrict__ tmp, Fft1D plan,
                                     const float2* __restrict__ tw, int C, int H, int Wc, int TC) {
  APH_DYN_SMEM(smem);
  float2* a = reinterpret_cast<float2*>(smem);
  float2* b = a + TC * H;
  const int g0 = blockIdx.x * TC, G = C * Wc;
  // kLoadBatch items per thread with all their loads issued before the first LDS write (clamped addresses, the value masked
  // afterwards): one memory round trip per batch instead of one per item
  for (int base = 0; base < TC * H; base += kLoadBatch * blockDim.x) {
    float2 v[kLoadBatch];
    float sc[kLoadBatch], sf[kLoadBatch];
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int idx = base + u * blockDim.x + threadIdx.x;
      const int t = idx % TC, ky = idx / TC, g = g0 + t;
      const bool ok = idx < TC * H && g < G;
      const int c = ok ? g / Wc : 0, kx = ok ? g - c * Wc : 0, kyc = ok ? ky : 0;
      v[u] = params[((size_t)c * H + kyc) * Wc + kx];
      sc[u] = scale ? scale[kyc * Wc + kx] : 1.0f;
      sf[u] = shift ? shift[kyc * Wc + kx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int idx = base + u * blockDim.x + threadIdx.x;
      const int t = idx % TC, ky = idx / TC, g = g0 + t;
      if (idx < TC * H) {
        float2 w = make_float2(0.f, 0.f);
        if (g < G) {
          const float s = sc[u];
          w = v[u];
          w.x *= s; w.y *= s;
          if (shift) { const float sh = s * sf[u]; w.x += sh; w.y += sh; }
        }
        a[t * H + ky] = w;
      }
    }
  }
  __syncthreads();
  const float2* r = fft_lds<+1>(a, b, plan, TC, tw);
  for (int idx = threadIdx.x; idx < TC * H; idx += blockDim.x) {
    const int t = idx % TC, y = idx / TC, g = g0 + t;
    if (g < G) {
      const int c = g / Wc, kx = g - c * Wc;
      tmp[((size_t)c * H + y) * Wc + kx] = r[t * H + y];
    }
  }
}

// column pass, adjoint: grad[c][ky][kx] = scale * sum_y tmp[c][y][kx] e^{-2 pi i ky y/H}
__global__ void fft_col_adjoint_kernel(const float2* __restrict__ tmp, const float* __restrict__ scale,
                                       float2* __restrict__ grad, Fft1D plan, const float2* __restrict__ tw,
                                       int C, int H, int Wc, int TC) {
  APH_DYN_SMEM(smem);
  float2* a = reinterpret_cast<float2*>(smem);
  float2* b = a + TC * H;
  const int g0 = blockIdx.x * TC, G = C * Wc;
  for (int base = 0; base < TC * H; base += kLoadBatch * blockDim.x) {      // (batched loads: see fft_col_synth_kernel)
    float2 v[kLoadBatch];
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int idx = base + u * blockDim.x + threadIdx.x;
      const int t = idx % TC, y = idx / TC, g = g0 + t;
      const bool ok = idx < TC * H && g < G;
      const int c = ok ? g / Wc : 0, kx = ok ? g - c * Wc : 0;
      v[u] = tmp[((size_t)c * H + (ok ? y : 0)) * Wc + kx];
    }
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int idx = base + u * blockDim.x + threadIdx.x;
      const int t = idx % TC, y = idx / TC, g = g0 + t;
      if (idx < TC * H) a[t * H + y] = g < G ? v[u] : make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  const float2* r = fft_lds<-1>(a, b, plan, TC, tw);
  for (int idx = threadIdx.x; idx < TC * H; idx += blockDim.x) {
    const int t = idx % TC, ky = idx / TC, g = g0 + t;
    if (g < G) {
      const int c = g / Wc, kx = g - c * Wc;
      const float s = scale ? scale[ky * Wc + kx] : 1.0f;
      float2 v = r[t * H + ky];
      grad[((size_t)c * H + ky) * Wc + kx] = make_float2(v.x * s, v.y * s);
    }
  }
}

// ---------------------------------------------------------------------------------
// row pass, forward synthesis (C2R): rows (2p, 2p+1) of plane c as one complex transform.
// Emits raw[c][y][x] and per-block fp64 partial (sum, sum of squares) for the global std.
// ---------------------------------------------------------------------------------
__global__ void fft_row_synth_kernel(const float2* __restrict__ tmp, float* __restrict__ raw,
                                     double* __restrict__ partials, Fft1D plan, const float2* __restrict__ tw,
                                     int H, int W, int Wc, float norm) {
  APH_DYN_SMEM(smem);
  float2* a = reinterpret_cast<float2*>(smem);
  float2* b = a + W;
  __shared__ double red[16];
  const int pairs = (H + 1) / 2;
  const int c = blockIdx.x / pairs, p = blockIdx.x - c * pairs;
  const int y0 = 2 * p, y1 = 2 * p + 1;
  const bool has1 = y1 < H;
  const float2* ra = tmp + ((size_t)c * H + y0) * Wc;
  const float2* rb = tmp + ((size_t)c * H + (has1 ? y1 : y0)) * Wc;
  for (int base = 0; base < Wc; base += kLoadBatch * blockDim.x) {      // (batched loads: see fft_col_synth_kernel)
    float2 Av[kLoadBatch], Bw[kLoadBatch];
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int k = base + u * blockDim.x + threadIdx.x, kc = k < Wc ? k : 0;
      Av[u] = ra[kc];
      Bw[u] = rb[kc];
    }
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int k = base + u * blockDim.x + threadIdx.x;
      if (k < Wc) {
        const float2 A = Av[u];
        const float2 Bv = has1 ? Bw[u] : make_float2(0.f, 0.f);
        const bool edge = (k == 0) || (2 * k == W);   // DC / Nyquist: imaginary part ignored (C2R)
        if (edge) {
          a[k] = make_float2(A.x, Bv.x);
        } else {
          a[k] = make_float2(A.x - Bv.y, A.y + Bv.x);
          a[W - k] = make_float2(A.x + Bv.y, Bv.x - A.y);
        }
      }
    }
  }
  __syncthreads();
  const float2* r = fft_lds<+1>(a, b, plan, 1, tw);
  double s1 = 0.0, s2 = 0.0;
  float* o0 = raw + ((size_t)c * H + y0) * W;
  float* o1 = raw + ((size_t)c * H + y1) * W;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const float2 z = r[x];
    const float v0 = z.x * norm, v1 = z.y * norm;
    o0[x] = v0;
    s1 += v0; s2 += (double)v0 * v0;
    if (has1) { o1[x] = v1; s1 += v1; s2 += (double)v1 * v1; }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s1; partials[2 * blockIdx.x + 1] = s2; }
}

// row pass, adjoint (R2C with interior columns doubled).  The std-normalisation adjoint is
// fused into the load:  d raw = A * dn + B * (raw - mean)   (bstats = {A, B, mean}).
__global__ void fft_row_adjoint_kernel(const float* __restrict__ dn, const float* __restrict__ raw,
                                       const float* __restrict__ bstats, float2* __restrict__ tmp, Fft1D plan,
                                       const float2* __restrict__ tw, int H, int W, int Wc, float norm, int plain) {
  APH_DYN_SMEM(smem);
  float2* a = reinterpret_cast<float2*>(smem);
  float2* b = a + W;
  const int pairs = (H + 1) / 2;
  const int c = blockIdx.x / pairs, p = blockIdx.x - c * pairs;
  const int y0 = 2 * p, y1 = 2 * p + 1;
  const bool has1 = y1 < H;
  // plain: the forward transform rfft2 itself (aph_rfft2) -- no normalisation adjoint, no doubling of interior columns
  const float A = plain ? 1.0f : bstats[0], Bc = plain ? 0.0f : bstats[1], mu = plain ? 0.0f : bstats[2];
  const size_t o0 = ((size_t)c * H + y0) * W, o1 = ((size_t)c * H + y1) * W;
  const size_t o1c = has1 ? o1 : o0;
  for (int base = 0; base < W; base += kLoadBatch * blockDim.x) {      // (batched loads: see fft_col_synth_kernel)
    float d0[kLoadBatch], r0[kLoadBatch], d1[kLoadBatch], r1[kLoadBatch];
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int x = base + u * blockDim.x + threadIdx.x, xc = x < W ? x : 0;
      d0[u] = dn[o0 + xc]; r0[u] = raw[o0 + xc];
      d1[u] = dn[o1c + xc]; r1[u] = raw[o1c + xc];
    }
#pragma unroll
    for (int u = 0; u < kLoadBatch; ++u) {
      const int x = base + u * blockDim.x + threadIdx.x;
      if (x < W) {
        const float g0 = A * d0[u] + Bc * (r0[u] - mu);
        const float g1 = has1 ? A * d1[u] + Bc * (r1[u] - mu) : 0.f;
        a[x] = make_float2(g0, g1);
      }
    }
  }
  __syncthreads();
  const float2* r = fft_lds<-1>(a, b, plan, 1, tw);
  float2* t0 = tmp + ((size_t)c * H + y0) * Wc;
  float2* t1 = tmp + ((size_t)c * H + y1) * Wc;
  for (int k = threadIdx.x; k < Wc; k += blockDim.x) {
    const float2 z = r[k];
    const float2 zc = r[k == 0 ? 0 : W - k];
    // Ga = (Z[k] + conj(Z[-k])) / 2 ; Gb = (Z[k] - conj(Z[-k])) / (2i)
    const float f = ((!plain && k >= 1 && k <= W - Wc) ? 1.0f : 0.5f) * norm;
    t0[k] = make_float2((z.x + zc.x) * f, (z.y - zc.y) * f);
    if (has1) t1[k] = make_float2((z.y + zc.y) * f, (zc.x - z.x) * f);
  }
}

// ---------------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------------
// generic partial (sum, sumsq) over a flat fp32 buffer -- for the pixel / DWT parameterisers
__global__ void stats_partial_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partials) {
  __shared__ double red[16];
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    s1 += v; s2 += (double)v * v;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s1; partials[2 * blockIdx.x + 1] = s2; }
}

// stats[0] = mean, stats[1] = unbiased std   (image.py:174 `image.std()`)
__global__ void stats_finalize_kernel(const double* __restrict__ partials, int nparts, double n, float* __restrict__ stats) {
  __shared__ double red[16];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { s1 += partials[2 * i]; s2 += partials[2 * i + 1]; }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    const double mean = s1 / n;
    double var = (s2 - s1 * mean) / (n - 1.0);
    if (var < 0) var = 0;
    stats[0] = (float)mean;
    stats[1] = (float)sqrt(var);
  }
}

// bstats = {A, B, mean} for  d raw = A * dn + B * (raw - mean):
//   y = c x / s ;  dL/dx_i = (c/s) g_i - c (sum_j g_j x_j) / (s^3 (N-1)) (x_i - mean)
// fixed_div > 0 selects pixel_image's `fixcontrast` branch (image.py:115-116): y = c x / fixed_div.
__global__ void bstats_finalize_kernel(const double* __restrict__ partials, int nparts, double n,
                                       const float* __restrict__ stats, float contrast, float fixed_div,
                                       float* __restrict__ bstats) {
  __shared__ double red[16];
  double sg = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) sg += partials[i];
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) {
    if (fixed_div > 0.f) {
      bstats[0] = contrast / fixed_div; bstats[1] = 0.f; bstats[2] = 0.f;
    } else {
      const double s = stats[1], c = contrast;
      bstats[0] = (float)(c / s);
      bstats[1] = (float)(-c * sg / (s * s * s * (n - 1.0)));
      bstats[2] = stats[0];
    }
  }
}

// ---------------------------------------------------------------------------------
// colour decorrelation + sigmoid (to_valid_rgb), and adjoint
// ---------------------------------------------------------------------------------
struct ColorMat { float m[9]; };  // colcorr_t[c][d], row-major

__global__ void rgb_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ stats, float contrast,
                               float fixed_div, ColorMat cc, int decorrelate, float* __restrict__ rgb, size_t HW) {
  const float k = fixed_div > 0.f ? contrast / fixed_div : contrast / stats[1];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) {
    const float n0 = raw[i] * k, n1 = raw[HW + i] * k, n2 = raw[2 * HW + i] * k;
    float z0 = n0, z1 = n1, z2 = n2;
    if (decorrelate) {
      z0 = n0 * cc.m[0] + n1 * cc.m[3] + n2 * cc.m[6];
      z1 = n0 * cc.m[1] + n1 * cc.m[4] + n2 * cc.m[7];
      z2 = n0 * cc.m[2] + n1 * cc.m[5] + n2 * cc.m[8];
    }
    rgb[i] = 1.0f / (1.0f + expf(-z0));
    rgb[HW + i] = 1.0f / (1.0f + expf(-z1));
    rgb[2 * HW + i] = 1.0f / (1.0f + expf(-z2));
  }
}

// dn[c] = sum_d cc[c][d] * drgb[d] * rgb[d] (1 - rgb[d]);  partial sums of dn * raw (fp64)
__global__ void rgb_bwd_kernel(const float* __restrict__ drgb, const float* __restrict__ rgb,
                               const float* __restrict__ raw, ColorMat cc, int decorrelate, float gscale,
                               float* __restrict__ dn, double* __restrict__ partials, size_t HW) {
  __shared__ double red[16];
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) {
    const float r0 = rgb[i], r1 = rgb[HW + i], r2 = rgb[2 * HW + i];
    const float d0 = drgb[i] * gscale * r0 * (1.f - r0), d1 = drgb[HW + i] * gscale * r1 * (1.f - r1),
                d2 = drgb[2 * HW + i] * gscale * r2 * (1.f - r2);
    float g0 = d0, g1 = d1, g2 = d2;
    if (decorrelate) {
      g0 = cc.m[0] * d0 + cc.m[1] * d1 + cc.m[2] * d2;
      g1 = cc.m[3] * d0 + cc.m[4] * d1 + cc.m[5] * d2;
      g2 = cc.m[6] * d0 + cc.m[7] * d1 + cc.m[8] * d2;
    }
    dn[i] = g0; dn[HW + i] = g1; dn[2 * HW + i] = g2;
    acc += (double)g0 * raw[i] + (double)g1 * raw[HW + i] + (double)g2 * raw[2 * HW + i];
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// elementwise std-normalisation adjoint (pixel / DWT parameterisers; the FFT path fuses it)
__global__ void norm_bwd_kernel(const float* __restrict__ dn, const float* __restrict__ raw,
                                const float* __restrict__ bstats, float* __restrict__ draw, size_t n) {
  const float A = bstats[0], Bc = bstats[1], mu = bstats[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    draw[i] = A * dn[i] + Bc * (raw[i] - mu);
}

// ---------------------------------------------------------------------------------
// illustrip's RGB priors (illustrip.py:438-440, `--gen RGB`):
//     loss += mean_c |mean_hw(rgb_c) - t_mean|  +  mean_c |std_hw(rgb_c) - t_std|        (unbiased std)
// Per-channel (sum, sumsq) in fp64 block partials, then value and gradient in one elementwise pass:
//     d/d rgb_c[p] = [sign(m_c - t_mean) / HW  +  sign(s_c - t_std) (rgb_c[p] - m_c) / ((HW - 1) s_c)] / 3
// ---------------------------------------------------------------------------------
constexpr int kPriorBlocks = 128;

__global__ void rgb_prior_partial_kernel(const float* __restrict__ rgb, size_t HW, double* __restrict__ partials) {
  __shared__ double red[16];
  const float* x = rgb + (size_t)blockIdx.y * HW;
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    s1 += v; s2 += (double)v * v;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    double* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    o[0] = s1; o[1] = s2;
  }
}

__device__ __forceinline__ void prior_channel_stats(const double* __restrict__ partials, int nb, int c, double n, double& mean, double& sd) {
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < nb; ++i) { s1 += partials[((size_t)c * nb + i) * 2]; s2 += partials[((size_t)c * nb + i) * 2 + 1]; }
  mean = s1 / n;
  double var = (s2 - s1 * mean) / (n - 1.0);
  sd = sqrt(var > 0 ? var : 0);
}

__global__ void rgb_prior_apply_kernel(const float* __restrict__ rgb, size_t HW, const double* __restrict__ partials, int nb,
                                       float t_mean, float t_std, float weight, float* __restrict__ loss, float* __restrict__ grgb) {
  __shared__ float ab[3];
  const int c = blockIdx.y;
  const double n = (double)HW;
  if (threadIdx.x == 0) {
    double m, sd;
    prior_channel_stats(partials, nb, c, n, m, sd);
    const double sm = m > t_mean ? 1.0 : (m < t_mean ? -1.0 : 0.0), ss = sd > t_std ? 1.0 : (sd < t_std ? -1.0 : 0.0);
    ab[0] = (float)(weight * sm / (3.0 * n));
    ab[1] = sd > 0 ? (float)(weight * ss / (3.0 * (n - 1.0) * sd)) : 0.f;
    ab[2] = (float)m;
    if (loss && blockIdx.x == 0 && c == 0) {          // one thread adds the value of all three channels (fixed order)
      double v = 0.0;
      for (int cc = 0; cc < 3; ++cc) {
        double m2, s2;
        prior_channel_stats(partials, nb, cc, n, m2, s2);
        v += fabs(m2 - t_mean) / 3.0 + fabs(s2 - t_std) / 3.0;
      }
      loss[0] += (float)(weight * v);
    }
  }
  __syncthreads();
  if (!grgb) return;
  const float a = ab[0], b = ab[1], mu = ab[2];
  const float* x = rgb + (size_t)c * HW;
  float* g = grgb + (size_t)c * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) g[i] += a + b * (x[i] - mu);
}

// ---------------------------------------------------------------------------------
// --sharp term (clip_fft.py:269-270): derivat(img, 'naiv') = 0.5 (mean |d/dx| + mean |d/dy|)  (utils.py:265-268)
// over rgb [3,H,W]; value and gradient (sub-gradient 0 at ties, as torch.abs) in two passes.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void rgb_sharp_partial_kernel(const float* __restrict__ rgb, int H, int W, double* __restrict__ partials) {
  __shared__ double red[16];
  const size_t n = (size_t)3 * H * W;
  double sx = 0.0, sy = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float v = rgb[i];
    if (x + 1 < W) sx += fabsf(rgb[i + 1] - v);
    if (y + 1 < H) sy += fabsf(rgb[i + W] - v);
  }
  sx = block_sum(sx, red);
  sy = block_sum(sy, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sx; partials[2 * blockIdx.x + 1] = sy; }
}

__global__ void rgb_sharp_apply_kernel(const float* __restrict__ rgb, int H, int W, const double* __restrict__ partials, int nb,
                                       float weight, float* __restrict__ loss, float* __restrict__ grgb) {
  const double nx = 3.0 * H * (W - 1), ny = 3.0 * (H - 1) * W;
  if (loss && blockIdx.x == 0 && threadIdx.x == 0) {
    double sx = 0.0, sy = 0.0;
    for (int i = 0; i < nb; ++i) { sx += partials[2 * i]; sy += partials[2 * i + 1]; }
    loss[0] += (float)(weight * 0.5 * (sx / nx + sy / ny));
  }
  if (!grgb) return;
  const float kx = (float)(0.5 * weight / nx), ky = (float)(0.5 * weight / ny);
  const size_t n = (size_t)3 * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float v = rgb[i];
    float gx = 0.f, gy = 0.f;
    if (x > 0) gx += sgnf(v - rgb[i - 1]);
    if (x + 1 < W) gx -= sgnf(rgb[i + 1] - v);
    if (y > 0) gy += sgnf(v - rgb[i - W]);
    if (y + 1 < H) gy -= sgnf(rgb[i + W] - v);
    grgb[i] += kx * gx + ky * gy;
  }
}

}  // namespace aph

// =====================================================================================
// host side: plan + C ABI
// =====================================================================================
using namespace aph;

struct aph_synth_plan {
  int C, H, W, Wc, TC, col_threads = 1024;   // column pass: 8 columns x 1024 threads measured best at 720p (95 vs 122 us per synth fwd+bwd)
  Fft1D ph, pw;
  float2* twH = nullptr;
  float2* twW = nullptr;
  float2* tmp = nullptr;      // [C][H][Wc] complex intermediate
  float* dn = nullptr;        // [C][H][W]
  double* partials = nullptr; // max(nrow_blocks, kElemBlocks) * 2
  float* stats = nullptr;     // mean, std
  float* bstats = nullptr;    // A, B, mean
  int nrow_blocks;
};

static const int kElemBlocks = 1024;

static bool factorize(int n, Fft1D* p) {
  p->n = n; p->npass = 0;
  int m = n;
  auto push = [&](int r) { if (p->npass >= 14) return false; p->radix[p->npass++] = r; return true; };
  while (m % 4 == 0) { if (!push(4)) return false; m /= 4; }
  while (m % 2 == 0) { if (!push(2)) return false; m /= 2; }
  for (int r = 3; (long long)r * r <= m || r <= 31; r += 2)
    while (m % r == 0) { if (!push(r)) return false; m /= r; }
  if (m > 1 && !push(m)) return false;             // the remaining factor is a prime > 31 (fft_pass_large)
  return true;
}

static float2* make_twiddles(int n) {
  std::vector<float2> h(n);
  for (int t = 0; t < n; ++t) {
    const double ang = 2.0 * M_PI * (double)t / (double)n;
    h[t] = make_float2((float)cos(ang), (float)sin(ang));
  }
  float2* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(float2) * n) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  return d;
}

extern "C" {

int aph_synth_plan_create(int C, int H, int W, aph_synth_plan** out) {
  APH_TRY
  if (!out || C < 1 || H < 2 || W < 2) return aph_fail(APH_ERR_ARG, "aph_synth_plan_create: bad shape C=%d H=%d W=%d", C, H, W);
  auto* p = new aph_synth_plan();
  p->C = C; p->H = H; p->W = W; p->Wc = W / 2 + 1;
  if (!factorize(H, &p->ph) || !factorize(W, &p->pw)) {
    delete p;
    return aph_fail(APH_ERR_UNSUPPORTED, "FFT size %dx%d has more than 14 prime factors", W, H);
  }
  if (W > 8192 || H > 8192) { delete p; return aph_fail(APH_ERR_UNSUPPORTED, "FFT dimension > 8192 not supported (%dx%d)", W, H); }
  int tc = (64 * 1024) / (H * 8);
  p->TC = tc < 1 ? 1 : (tc > 8 ? 8 : tc);
  p->twH = make_twiddles(H);
  p->twW = make_twiddles(W);
  p->nrow_blocks = C * ((H + 1) / 2);
  const int npart = p->nrow_blocks > kElemBlocks ? p->nrow_blocks : kElemBlocks;
  hipError_t e = hipMalloc((void**)&p->tmp, sizeof(float2) * (size_t)C * H * p->Wc);
  if (e == hipSuccess) e = hipMalloc((void**)&p->dn, sizeof(float) * (size_t)C * H * W);
  if (e == hipSuccess) e = hipMalloc((void**)&p->partials, sizeof(double) * 2 * npart);
  if (e == hipSuccess) e = hipMalloc((void**)&p->stats, sizeof(float) * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&p->bstats, sizeof(float) * 4);
  APH_ALLOW_SMEM(fft_col_synth_kernel, sizeof(float2) * 2 * p->TC * H);
  APH_ALLOW_SMEM(fft_col_adjoint_kernel, sizeof(float2) * 2 * p->TC * H);
  APH_ALLOW_SMEM(fft_row_synth_kernel, sizeof(float2) * 2 * W);
  APH_ALLOW_SMEM(fft_row_adjoint_kernel, sizeof(float2) * 2 * W);
  if (e != hipSuccess || !p->twH || !p->twW) { aph_synth_plan_destroy(p); return aph_fail(APH_ERR_HIP, "synth plan allocation failed"); }
  *out = p;
  return APH_OK;
  APH_CATCH
}

int aph_synth_plan_destroy(aph_synth_plan* p) {
  if (!p) return APH_OK;
  (void)hipFree(p->twH); (void)hipFree(p->twW); (void)hipFree(p->tmp); (void)hipFree(p->dn); (void)hipFree(p->partials);
  (void)hipFree(p->stats); (void)hipFree(p->bstats);
  delete p;
  return APH_OK;
}

static ColorMat to_cm(const float* cc) { ColorMat m; for (int i = 0; i < 9; ++i) m.m[i] = cc ? cc[i] : (i % 4 == 0 ? 1.f : 0.f); return m; }

// params [C,H,Wc,2] f32, scale [H,Wc] f32, shift [H,Wc] f32 or NULL  ->  raw [C,H,W], rgb [C,H,W]
// (image.py:164-175 + :24-28).  stats (mean, std of raw at contrast 1) stay in the plan for the adjoint.
int aph_synth_fft_fwd(aph_synth_plan* p, const float* params, const float* scale, const float* shift, float contrast,
                      const float* colcorr_t9, int decorrelate, float* raw, float* rgb, void* stream_) {
  APH_TRY
  if (!p || !params || !scale || !raw || !rgb) return aph_fail(APH_ERR_ARG, "aph_synth_fft_fwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const int G = p->C * p->Wc;
  APH_LAUNCH(fft_col_synth_kernel, dim3((G + p->TC - 1) / p->TC), dim3(p->col_threads), sizeof(float2) * 2 * p->TC * p->H, st,
             (const float2*)params, scale, shift, p->tmp, p->ph, (const float2*)p->twH, p->C, p->H, p->Wc, p->TC);
  const float norm = (float)(1.0 / sqrt((double)p->H * (double)p->W));
  APH_LAUNCH(fft_row_synth_kernel, dim3(p->nrow_blocks), dim3(256), sizeof(float2) * 2 * p->W, st,
             (const float2*)p->tmp, raw, p->partials, p->pw, (const float2*)p->twW, p->H, p->W, p->Wc, norm);
  APH_LAUNCH(stats_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)p->partials, p->nrow_blocks,
             (double)p->C * p->H * p->W, p->stats);
  const size_t HW = (size_t)p->H * p->W;
  APH_LAUNCH(rgb_fwd_kernel, dim3(kElemBlocks), dim3(256), 0, st, (const float*)raw, (const float*)p->stats, contrast, 0.f,
             to_cm(colcorr_t9), decorrelate, rgb, HW);
  return aph_check_launch("aph_synth_fft_fwd");
  APH_CATCH
}

// d_rgb [C,H,W] (times gscale) -> grad_params [C,H,Wc,2]   (adjoint of aph_synth_fft_fwd)
int aph_synth_fft_bwd(aph_synth_plan* p, const float* d_rgb, float gscale, const float* rgb, const float* raw,
                      const float* scale, float contrast, const float* colcorr_t9, int decorrelate, float* grad_params,
                      void* stream_) {
  APH_TRY
  if (!p || !d_rgb || !rgb || !raw || !scale || !grad_params) return aph_fail(APH_ERR_ARG, "aph_synth_fft_bwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const size_t HW = (size_t)p->H * p->W;
  APH_LAUNCH(rgb_bwd_kernel, dim3(kElemBlocks), dim3(256), 0, st, d_rgb, rgb, raw, to_cm(colcorr_t9), decorrelate, gscale,
             p->dn, p->partials, HW);
  APH_LAUNCH(bstats_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)p->partials, kElemBlocks,
             (double)p->C * p->H * p->W, (const float*)p->stats, contrast, 0.f, p->bstats);
  const float norm = (float)(1.0 / sqrt((double)p->H * (double)p->W));
  APH_LAUNCH(fft_row_adjoint_kernel, dim3(p->nrow_blocks), dim3(256), sizeof(float2) * 2 * p->W, st, (const float*)p->dn, raw,
             (const float*)p->bstats, p->tmp, p->pw, (const float2*)p->twW, p->H, p->W, p->Wc, norm, 0);
  const int G = p->C * p->Wc;
  APH_LAUNCH(fft_col_adjoint_kernel, dim3((G + p->TC - 1) / p->TC), dim3(p->col_threads), sizeof(float2) * 2 * p->TC * p->H, st,
             (const float2*)p->tmp, scale, (float2*)grad_params, p->ph, (const float2*)p->twH, p->C, p->H, p->Wc, p->TC);
  return aph_check_launch("aph_synth_fft_bwd");
  APH_CATCH
}

// Standalone transforms on the plan's geometry, torch.fft semantics with norm='ortho' (illustrip.py:401-409: the per-frame
// irfftn -> warp -> rfftn round trip of `--gen FFT`).
//   aph_irfft2: spectrum [C,H,Wc,2] -> image [C,H,W]   (= torch.fft.irfftn(view_as_complex(x), s=(H,W), norm='ortho'))
//   aph_rfft2:  image [C,H,W] -> spectrum [C,H,Wc,2]   (= view_as_real(torch.fft.rfftn(x, s=(H,W), dim=[-2,-1], norm='ortho')))
int aph_irfft2(aph_synth_plan* p, const float* spectrum, float* image, void* stream_) {
  APH_TRY
  if (!p || !spectrum || !image) return aph_fail(APH_ERR_ARG, "aph_irfft2: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const int G = p->C * p->Wc;
  APH_LAUNCH(fft_col_synth_kernel, dim3((G + p->TC - 1) / p->TC), dim3(p->col_threads), sizeof(float2) * 2 * p->TC * p->H, st,
             (const float2*)spectrum, (const float*)nullptr, (const float*)nullptr, p->tmp, p->ph, (const float2*)p->twH, p->C, p->H, p->Wc, p->TC);
  const float norm = (float)(1.0 / sqrt((double)p->H * (double)p->W));
  APH_LAUNCH(fft_row_synth_kernel, dim3(p->nrow_blocks), dim3(256), sizeof(float2) * 2 * p->W, st,
             (const float2*)p->tmp, image, p->partials, p->pw, (const float2*)p->twW, p->H, p->W, p->Wc, norm);
  return aph_check_launch("aph_irfft2");
  APH_CATCH
}

int aph_rfft2(aph_synth_plan* p, const float* image, float* spectrum, void* stream_) {
  APH_TRY
  if (!p || !image || !spectrum) return aph_fail(APH_ERR_ARG, "aph_rfft2: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const float norm = (float)(1.0 / sqrt((double)p->H * (double)p->W));
  APH_LAUNCH(fft_row_adjoint_kernel, dim3(p->nrow_blocks), dim3(256), sizeof(float2) * 2 * p->W, st, image, image,
             (const float*)nullptr, p->tmp, p->pw, (const float2*)p->twW, p->H, p->W, p->Wc, norm, 1);
  const int G = p->C * p->Wc;
  APH_LAUNCH(fft_col_adjoint_kernel, dim3((G + p->TC - 1) / p->TC), dim3(p->col_threads), sizeof(float2) * 2 * p->TC * p->H, st,
             (const float2*)p->tmp, (const float*)nullptr, (float2*)spectrum, p->ph, (const float2*)p->twH, p->C, p->H, p->Wc, p->TC);
  return aph_check_launch("aph_rfft2");
  APH_CATCH
}

// Spatial-domain parameterisers (pixel_image image.py:114-118; the DWT path after its inverse transform):
// raw [C,H,W] -> rgb.  fixed_div > 0: image * contrast / fixed_div (fixcontrast), else / global std.
int aph_synth_spatial_fwd(aph_synth_plan* p, const float* raw, float contrast, float fixed_div, const float* colcorr_t9,
                          int decorrelate, float* rgb, void* stream_) {
  APH_TRY
  if (!p || !raw || !rgb) return aph_fail(APH_ERR_ARG, "aph_synth_spatial_fwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const size_t n = (size_t)p->C * p->H * p->W;
  APH_LAUNCH(stats_partial_kernel, dim3(kElemBlocks), dim3(256), 0, st, raw, n, p->partials);
  APH_LAUNCH(stats_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)p->partials, kElemBlocks, (double)n, p->stats);
  APH_LAUNCH(rgb_fwd_kernel, dim3(kElemBlocks), dim3(256), 0, st, raw, (const float*)p->stats, contrast, fixed_div,
             to_cm(colcorr_t9), decorrelate, rgb, (size_t)p->H * p->W);
  return aph_check_launch("aph_synth_spatial_fwd");
  APH_CATCH
}

int aph_synth_spatial_bwd(aph_synth_plan* p, const float* d_rgb, float gscale, const float* rgb, const float* raw,
                          float contrast, float fixed_div, const float* colcorr_t9, int decorrelate, float* d_raw,
                          void* stream_) {
  APH_TRY
  if (!p || !d_rgb || !rgb || !raw || !d_raw) return aph_fail(APH_ERR_ARG, "aph_synth_spatial_bwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const size_t n = (size_t)p->C * p->H * p->W;
  APH_LAUNCH(rgb_bwd_kernel, dim3(kElemBlocks), dim3(256), 0, st, d_rgb, rgb, raw, to_cm(colcorr_t9), decorrelate, gscale,
             p->dn, p->partials, (size_t)p->H * p->W);
  APH_LAUNCH(bstats_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)p->partials, kElemBlocks, (double)n,
             (const float*)p->stats, contrast, fixed_div, p->bstats);
  APH_LAUNCH(norm_bwd_kernel, dim3(kElemBlocks), dim3(256), 0, st, (const float*)p->dn, raw, (const float*)p->bstats, d_raw, n);
  return aph_check_launch("aph_synth_spatial_bwd");
  APH_CATCH
}

// copies {mean, std} of the last forward to host-visible device memory `out2` (2 floats, device pointer)
int aph_synth_stats(aph_synth_plan* p, float* out2, void* stream_) {
  APH_TRY
  if (!p || !out2) return aph_fail(APH_ERR_ARG, "aph_synth_stats: null argument");
  if (hipMemcpyAsync(out2, p->stats, 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream_) != hipSuccess)
    return aph_fail(APH_ERR_HIP, "aph_synth_stats: copy failed");
  return APH_OK;
  APH_CATCH
}

// restores {mean, std} saved by aph_synth_stats (an autograd node whose forward was followed by other forwards)
int aph_synth_set_stats(aph_synth_plan* p, const float* in2, void* stream_) {
  APH_TRY
  if (!p || !in2) return aph_fail(APH_ERR_ARG, "aph_synth_set_stats: null argument");
  if (hipMemcpyAsync(p->stats, in2, 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream_) != hipSuccess)
    return aph_fail(APH_ERR_HIP, "aph_synth_set_stats: copy failed");
  return APH_OK;
  APH_CATCH
}

// illustrip.py:438-440 RGB priors on rgb [3,H,W]: adds weight * value to *d_loss (nullable) and weight * gradient to
// d_rgb_grad (nullable, accumulated in place).  d_ws: >= aph_rgb_priors_ws_bytes() bytes of device scratch.
size_t aph_rgb_priors_ws_bytes(void) { return sizeof(double) * 3 * kPriorBlocks * 2; }

int aph_rgb_priors(const float* d_rgb, int H, int W, float t_mean, float t_std, float weight, void* d_ws, float* d_loss,
                   float* d_rgb_grad, void* stream_) {
  APH_TRY
  if (!d_rgb || !d_ws || H < 1 || W < 1 || (size_t)H * W < 2) return aph_fail(APH_ERR_ARG, "aph_rgb_priors: bad argument");
  hipStream_t st = (hipStream_t)stream_;
  const size_t HW = (size_t)H * W;
  APH_LAUNCH(rgb_prior_partial_kernel, dim3(kPriorBlocks, 3), dim3(256), 0, st, d_rgb, HW, (double*)d_ws);
  APH_LAUNCH(rgb_prior_apply_kernel, dim3(kPriorBlocks, 3), dim3(256), 0, st, d_rgb, HW, (const double*)d_ws, kPriorBlocks, t_mean, t_std,
             weight, d_loss, d_rgb_grad);
  return aph_check_launch("aph_rgb_priors");
  APH_CATCH
}

// clip_fft.py:269-270: adds weight * derivat(rgb, 'naiv') to *d_loss (nullable) and its gradient into d_rgb_grad (nullable,
// accumulated); pass weight = -a.sharp.  d_ws as for aph_rgb_priors.
int aph_rgb_sharp(const float* d_rgb, int H, int W, float weight, void* d_ws, float* d_loss, float* d_rgb_grad, void* stream_) {
  APH_TRY
  if (!d_rgb || !d_ws || H < 2 || W < 2) return aph_fail(APH_ERR_ARG, "aph_rgb_sharp: bad argument");
  hipStream_t st = (hipStream_t)stream_;
  APH_LAUNCH(rgb_sharp_partial_kernel, dim3(kPriorBlocks), dim3(256), 0, st, d_rgb, H, W, (double*)d_ws);
  APH_LAUNCH(rgb_sharp_apply_kernel, dim3(1024), dim3(256), 0, st, d_rgb, H, W, (const double*)d_ws, kPriorBlocks, weight, d_loss, d_rgb_grad);
  return aph_check_launch("aph_rgb_sharp");
  APH_CATCH
}

}  // extern "C"
