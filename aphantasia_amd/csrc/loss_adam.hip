// Similarity loss (+ its gradient) and the fused Adam update.
//
// Replaces: aphantasia/utils.py:270-295 (sim_func / dot_compare) as assembled at clip_fft.py:257-267
//           (loss = sum_t coef_t * sim_func(target_t, out_enc, type)), and torch.optim.Adam/AdamW as
//           configured at clip_fft.py:108-115 (+ optimizer.step() at :295).
#include "aph_device.h"
#include "aph_host.h"

namespace aph {

enum { SIM_COS = 0, SIM_MIX = 1, SIM_ANG = 2, SIM_DOT = 3 };

// Target buffer = n_broadcast rows [D] (text prompts: one embedding against every cut) followed by
// (T - n_broadcast) blocks [s_total, D] of per-cut targets (the reference-image term pairs cut s with
// slice s of the input image, clip_fft.py:216,267); s_offset = first global cut of this rank's shard.
struct TargetLayout { int n_broadcast, s_total, s_offset; };
__device__ __forceinline__ const float* tgt_row(const float* tgt, int t, int s, int D, TargetLayout L) {
  if (t < L.n_broadcast) return tgt + (size_t)t * D;
  return tgt + ((size_t)L.n_broadcast + (size_t)(t - L.n_broadcast) * L.s_total + L.s_offset + s) * D;
}

// One workgroup per sample: per-target value terms and d loss / d enc[s].
//   enc [S,D] f32, tgt [T,D] f32, coef [T] f32 (sign*weight), denom = number of samples in the GLOBAL
//   mean (S, or the all-rank total when samples are sharded across ranks).
//   partial[s] = sum_t coef_t * (per-sample term) ; loss = sum_s partial[s] / denom   (not for SIM_DOT)
__global__ void sim_loss_kernel(const float* __restrict__ enc, const float* __restrict__ tgt, const float* __restrict__ coef,
                                int T, int D, int type, float denom, float gscale, float* __restrict__ partial,
                                float* __restrict__ genc, TargetLayout lay) {
  __shared__ float red[16];
  const int s = blockIdx.x;
  const float* e = enc + (size_t)s * D;
  float ee = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) ee += e[d] * e[d];
  ee = block_sum(ee, red);
  const float ne = sqrtf(ee);
  float total = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) genc[(size_t)s * D + d] = 0.f;
  for (int t = 0; t < T; ++t) {
    const float* v = tgt_row(tgt, t, s, D, lay);
    float dot = 0.f, vv = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) { dot += v[d] * e[d]; vv += v[d] * v[d]; }
    dot = block_sum(dot, red);
    vv = block_sum(vv, red);
    const float nv = sqrtf(vv);
    // torch.cosine_similarity: x.y / max(|x| |y|, eps), eps = 1e-8
    const float den = fmaxf(ne * nv, 1e-8f);
    const float c = dot / den;
    float val, dval_dc;
    if (type == SIM_MIX) {
      // spher = 2 asin(|v^ - e^| / 2)^2 with |v^ - e^|^2 = 2 - 2c  (utils.py:279-282)
      float dd = fmaxf(2.f - 2.f * c, 0.f);
      const float dist = sqrtf(dd);
      const float h = fminf(dist * 0.5f, 1.0f);
      const float a = asinf(h);
      val = c - 0.25f * (2.f * a * a);
      // d(0.5 a^2)/dc = a * da/dc,  da/dc = -1 / (2 dist sqrt(1 - dist^2/4))
      const float sq = sqrtf(fmaxf(1.f - h * h, 1e-12f));
      const float dadc = dist > 1e-6f ? -1.f / (2.f * dist * sq) : -0.5f;   // limit as dist -> 0: a ~ dist/2
      dval_dc = 1.f - a * dadc * 1.0f;
      dval_dc = 1.f - (a * dadc);   // val = c - 0.5 a^2
    } else if (type == SIM_ANG) {
      const float cc = fminf(fmaxf(c, -1.f), 1.f);
      val = -acosf(cc) * 0.31830988618379067f;          // the constant 1 is added on the host side
      dval_dc = 0.31830988618379067f / sqrtf(fmaxf(1.f - cc * cc, 1e-12f));
    } else {
      val = c;
      dval_dc = 1.f;
    }
    total += coef[t] * val;
    // dc/de = v / den - c * e / |e|^2
    const float k = coef[t] * dval_dc * gscale / denom;
    const float inv_ee = ee > 0.f ? 1.f / ee : 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) genc[(size_t)s * D + d] += k * (v[d] / den - c * e[d] * inv_ee);
  }
  if (threadIdx.x == 0) partial[s] = total;
}

// 'dot' similarity (dot_compare, utils.py:270-274): global over all samples:
//   dot = sum_{s,d} v[d] e[s,d]; mag = sqrt(sum e^2); loss_t = dot * dot / (1e-6 + mag)
// Pass 1: per-sample partial sums (dotsum per target, sumsq);  pass 2: gradient.
__global__ void sim_dot_partial_kernel(const float* __restrict__ enc, const float* __restrict__ tgt, int T, int D,
                                       float* __restrict__ part /*[S, T+1]*/, TargetLayout lay) {
  __shared__ float red[16];
  const int s = blockIdx.x;
  const float* e = enc + (size_t)s * D;
  float ee = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) ee += e[d] * e[d];
  ee = block_sum(ee, red);
  if (threadIdx.x == 0) part[(size_t)s * (T + 1) + T] = ee;
  for (int t = 0; t < T; ++t) {
    float dot = 0.f;
    const float* v = tgt_row(tgt, t, s, D, lay);
    for (int d = threadIdx.x; d < D; d += blockDim.x) dot += v[d] * e[d];
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) part[(size_t)s * (T + 1) + t] = dot;
  }
}
__global__ void sim_dot_grad_kernel(const float* __restrict__ enc, const float* __restrict__ tgt, const float* __restrict__ coef,
                                    const float* __restrict__ part, int S, int T, int D, float gscale,
                                    float* __restrict__ loss_out, float* __restrict__ genc, TargetLayout lay) {
  __shared__ float sums[64];
  // every block recomputes the global sums deterministically (S, T are small)
  if ((int)threadIdx.x <= T && threadIdx.x < 64) {
    double a = 0.0;
    for (int s = 0; s < S; ++s) a += part[(size_t)s * (T + 1) + threadIdx.x];
    sums[threadIdx.x] = (float)a;
  }
  __syncthreads();
  const float mag = sqrtf(sums[T]);
  const int s = blockIdx.x;
  float loss = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float g = 0.f;
    for (int t = 0; t < T; ++t) {
      const float dot = sums[t];
      // f = dot^2 / (1e-6 + mag): df/de = 2 dot v / (eps+mag) - dot^2 / (eps+mag)^2 * e / mag
      const float q = 1e-6f + mag;
      g += coef[t] * (2.f * dot * tgt_row(tgt, t, s, D, lay)[d] / q - dot * dot / (q * q) * (mag > 0.f ? enc[(size_t)s * D + d] / mag : 0.f));
    }
    genc[(size_t)s * D + d] = g * gscale;
  }
  if (s == 0 && threadIdx.x == 0) {
    for (int t = 0; t < T; ++t) loss += coef[t] * sums[t] * sums[t] / (1e-6f + mag);
    loss_out[0] = loss;
  }
}

// loss = base + sum_s partial[s] / denom  (single block)
__global__ void loss_reduce_kernel(const float* __restrict__ partial, int S, float denom, float base, float* __restrict__ loss_out) {
  __shared__ double red[16];
  double a = 0.0;
  for (int s = threadIdx.x; s < S; s += blockDim.x) a += partial[s];
  a = block_sum(a, red);
  if (threadIdx.x == 0) loss_out[0] = base + (float)(a / denom);
}

// ---------------------------------------------------------------------------------
// Adam / AdamW (torch.optim semantics, single tensor).  hyper = device floats so a captured graph
// can be replayed with new step / lr:  {lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrt(bias_corr2), gradscale}
// ---------------------------------------------------------------------------------
// guard[0] = running count of skipped (overflowed) steps, guard[1] = "this step's gradient has a non-finite element"
// Linear head on the encodings -- the aesthetic predictor of clip_fft.py:255-256 / utils.py:402-413
// (`loss -= 0.001 * a.aest * aest(out_enc).mean()`, aest = nn.Linear(D, 1)):
//   loss += coef * ( sum_s (w . enc_s + bias) ) / denom ,   genc[s][d] += gscale * coef * w[d] / denom
// One workgroup; wave w takes cuts w, w + NW, ...; the per-cut dots are summed in cut order (deterministic).
__global__ __launch_bounds__(256) void linear_head_kernel(const float* __restrict__ enc, int S, int D, const float* __restrict__ w,
                                                          float bias, float coef, float denom, float gscale,
                                                          float* __restrict__ loss, float* __restrict__ genc) {
  APH_DYN_SMEM(smem);
  float* dots = reinterpret_cast<float*>(smem);            // [S]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int s = wave; s < S; s += nw) {
    float a = 0.f;
    for (int d = lane; d < D; d += 64) a += w[d] * enc[(size_t)s * D + d];
    a = wave_sum(a);
    if (lane == 0) dots[s] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += dots[s] + bias;
    loss[0] += coef * t / denom;
  }
  if (genc) {
    const float k = gscale * coef / denom;
    for (int idx = threadIdx.x; idx < S * D; idx += blockDim.x) genc[idx] += k * w[idx % D];
  }
}

// y += alpha * x (the --enforce path adds the second cut set's image gradient / encoding gradient / loss term)
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

// The per-step frame (clip_fft.py:297-306 -> utils.checkout, utils.py:94-100): rgb f32 [3,H,W] in [0,1] -> uint8 [H,W,3],
//     np.clip(img ** gamma * 255, 0, 255).astype(np.uint8)         (truncation; gamma = 1: no pow)
// one pixel per thread and trip (the three planes are read coalesced; 2.7 MB out at 720p)
__global__ void rgb_to_u8_kernel(const float* __restrict__ rgb, unsigned char* __restrict__ out, size_t n, float gamma) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = rgb[(size_t)c * n + p];
      if (gamma != 1.0f) v = powf(v, gamma);
      v = fminf(fmaxf(v * 255.0f, 0.0f), 255.0f);
      out[p * 3 + c] = (unsigned char)v;
    }
  }
}

// (the gradient as 16-byte quads when its base is 16-byte aligned -- one quad per thread, every load of the launch in flight at
// once; the scalar loop this replaces made ten dependent round trips per thread: 7.4 us for 11 MB)
template <bool VEC>
__global__ void grad_guard_kernel(const float* __restrict__ g, size_t n, int* __restrict__ guard) {
  bool bad = false;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  if (VEC) {
    const size_t n4 = n >> 2;
    for (size_t i = t0; i < n4; i += nt) {
      const float4 x = reinterpret_cast<const float4*>(g)[i];
      bad |= !(fabsf(x.x) <= 3.0e38f) | !(fabsf(x.y) <= 3.0e38f) | !(fabsf(x.z) <= 3.0e38f) | !(fabsf(x.w) <= 3.0e38f);
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += nt) bad |= !(fabsf(g[i]) <= 3.0e38f);
  } else {
    for (size_t i = t0; i < n; i += nt) bad |= !(fabsf(g[i]) <= 3.0e38f);             // false for NaN and +-inf
  }
  if (bad) guard[1] = 1;                         // every writer stores the same value
}

struct AdamHyper { float lr, b1, b2, eps, wd, bc1, sbc2, gs; };
// one element of optimizer.step(): the same arithmetic whichever loop calls it
__device__ __forceinline__ void adam_one(float& pi, float g_raw, float* mi, float& vi, float* vmaxi, const AdamHyper& h, int decoupled) {
  float gi = g_raw * h.gs;
  if (h.wd != 0.f) {
    if (decoupled) pi *= (1.f - h.lr * h.wd);
    else gi += h.wd * pi;
  }
  float m1;
  if (mi) { m1 = h.b1 * *mi + (1.f - h.b1) * gi; *mi = m1; }
  else m1 = (1.f - h.b1) * gi;   // beta1 == 0 path keeps no first-moment buffer
  float v1 = h.b2 * vi + (1.f - h.b2) * gi * gi;
  vi = v1;
  if (vmaxi) { const float mx = fmaxf(*vmaxi, v1); *vmaxi = mx; v1 = mx; }
  const float denom = sqrtf(v1) / h.sbc2 + h.eps;
  pi = pi - (h.lr / h.bc1) * (m1 / denom);
}

// VEC: 16-byte quads (every buffer 16-byte aligned), grid sized for about one quad per thread: the 3-5 loads of a thread are
// in flight together and the launch has every byte of the update requested at once (the scalar grid-stride loop made ten
// dependent round trips per thread: 17 us for 55 MB at 1280x720)
template <bool VEC>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, const float* __restrict__ hyper, int decoupled, size_t n, int* __restrict__ guard) {
  if (guard && guard[1]) {                       // overflowed step: leave parameters and moments untouched
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) guard[0] += 1;
    return;
  }
  const AdamHyper h{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7]};
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  size_t tail0 = 0;
  if (VEC) {
    const size_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (size_t i = t0; i < n4; i += nt) {
      float4 pq = reinterpret_cast<float4*>(p)[i], vq = reinterpret_cast<float4*>(v)[i];
      const float4 gq = reinterpret_cast<const float4*>(g)[i];
      float4 mq = make_float4(0.f, 0.f, 0.f, 0.f), xq = mq;
      if (m) mq = reinterpret_cast<float4*>(m)[i];
      if (vmax) xq = reinterpret_cast<float4*>(vmax)[i];
      adam_one(pq.x, gq.x, m ? &mq.x : nullptr, vq.x, vmax ? &xq.x : nullptr, h, decoupled);
      adam_one(pq.y, gq.y, m ? &mq.y : nullptr, vq.y, vmax ? &xq.y : nullptr, h, decoupled);
      adam_one(pq.z, gq.z, m ? &mq.z : nullptr, vq.z, vmax ? &xq.z : nullptr, h, decoupled);
      adam_one(pq.w, gq.w, m ? &mq.w : nullptr, vq.w, vmax ? &xq.w : nullptr, h, decoupled);
      reinterpret_cast<float4*>(p)[i] = pq;
      reinterpret_cast<float4*>(v)[i] = vq;
      if (m) reinterpret_cast<float4*>(m)[i] = mq;
      if (vmax) reinterpret_cast<float4*>(vmax)[i] = xq;
    }
  }
  for (size_t i = tail0 + t0; i < n; i += nt) {
    float pi = p[i], vi = v[i];
    adam_one(pi, g[i], m ? m + i : nullptr, vi, vmax ? vmax + i : nullptr, h, decoupled);
    p[i] = pi;
    v[i] = vi;
  }
}

inline bool aligned16(const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; }
// guard scan + update, vectorised when every buffer allows it
void launch_adam(float* d_p, const float* d_g, float* d_m, float* d_v, float* d_vmax, const float* d_hyper, int decoupled_wd, size_t n,
                 int* d_guard, hipStream_t st) {
  const bool vec = n >= 4 && aligned16(d_p) && aligned16(d_g) && aligned16(d_m) && aligned16(d_v) && aligned16(d_vmax);
  if (vec) {
    const size_t wg = (n / 4 + 255) / 256;
    const unsigned grid = (unsigned)(wg > 8192 ? 8192 : wg);
    if (d_guard) APH_LAUNCH(grad_guard_kernel<true>, dim3(grid > 2048u ? 2048u : grid), dim3(256), 0, st, d_g, n, d_guard);
    APH_LAUNCH(adam_kernel<true>, dim3(grid), dim3(256), 0, st, d_p, d_g, d_m, d_v, d_vmax, d_hyper, decoupled_wd, n, d_guard);
  } else {
    if (d_guard) APH_LAUNCH(grad_guard_kernel<false>, dim3(512), dim3(256), 0, st, d_g, n, d_guard);
    APH_LAUNCH(adam_kernel<false>, dim3(1024), dim3(256), 0, st, d_p, d_g, d_m, d_v, d_vmax, d_hyper, decoupled_wd, n, d_guard);
  }
}

}  // namespace aph

using namespace aph;

extern "C" {

// sim_func(target_t, enc, type) summed over targets with coefficients, value and gradient in one pass.
// type: 0 cossim (None), 1 mix, 2 ang, 3 dot.  d_ws: f32 scratch of S*(T+2) elements.
// loss (device scalar) = sum_t coef_t * sim_t ;  d_genc = gscale * d loss / d enc.
// denom: sample count of the global mean (== S unless samples are sharded over ranks).
int aph_sim_loss(const float* d_enc, int S, int D, const float* d_targets, const float* d_coef, const float* h_coef, int T,
                 int n_broadcast, int s_total, int s_offset, int type, float denom, float gscale, float* d_ws, float* d_loss,
                 float* d_genc, void* stream_) {
  APH_TRY
  if (!d_enc || !d_targets || !d_coef || !d_ws || !d_loss || !d_genc || S < 1 || D < 1 || T < 1 || T > 62)
    return aph_fail(APH_ERR_ARG, "aph_sim_loss: bad argument (S=%d D=%d T=%d)", S, D, T);
  if (type < 0 || type > 3) return aph_fail(APH_ERR_ARG, "aph_sim_loss: unknown similarity type %d", type);
  if (n_broadcast < 0 || n_broadcast > T || (n_broadcast < T && (s_offset < 0 || s_offset + S > s_total)))
    return aph_fail(APH_ERR_ARG, "aph_sim_loss: bad target layout (n_broadcast=%d T=%d s_total=%d s_offset=%d S=%d)", n_broadcast, T, s_total, s_offset, S);
  const TargetLayout lay{n_broadcast, s_total, s_offset};
  hipStream_t st = (hipStream_t)stream_;
  if (type == SIM_DOT) {
    APH_LAUNCH(sim_dot_partial_kernel, dim3(S), dim3(256), 0, st, d_enc, d_targets, T, D, d_ws, lay);
    APH_LAUNCH(sim_dot_grad_kernel, dim3(S), dim3(256), 0, st, d_enc, d_targets, d_coef, (const float*)d_ws, S, T, D, gscale, d_loss, d_genc, lay);
    return aph_check_launch("aph_sim_loss");
  }
  float base = 0.f;
  if (type == SIM_ANG) {   // 1 - acos(.)mean/pi : the constant part
    if (!h_coef) return aph_fail(APH_ERR_ARG, "aph_sim_loss: 'ang' needs host coefficients");
    for (int t = 0; t < T; ++t) base += h_coef[t];
  }
  APH_LAUNCH(sim_loss_kernel, dim3(S), dim3(256), 0, st, d_enc, d_targets, d_coef, T, D, type, denom, gscale, d_ws, d_genc, lay);
  APH_LAUNCH(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)d_ws, S, denom, base, d_loss);
  return aph_check_launch("aph_sim_loss");
  APH_CATCH
}

// Adds a linear head's term to a loss / encoding-gradient pair produced by aph_sim_loss (same denom / gscale conventions):
// the aesthetic predictor, clip_fft.py:255-256 with coef = -0.001 * a.aest.  d_w [D] device, bias host scalar.
int aph_linear_head(const float* d_enc, int S, int D, const float* d_w, float bias, float coef, float denom, float gscale,
                    float* d_loss, float* d_genc, void* stream_) {
  APH_TRY
  if (!d_enc || !d_w || !d_loss || S < 1 || D < 1 || S > 16384 || !(denom > 0.f))
    return aph_fail(APH_ERR_ARG, "aph_linear_head: bad argument (S=%d D=%d)", S, D);
  APH_LAUNCH(linear_head_kernel, dim3(1), dim3(256), sizeof(float) * S, (hipStream_t)stream_, d_enc, S, D, d_w, bias, coef, denom, gscale, d_loss, d_genc);
  return aph_check_launch("aph_linear_head");
  APH_CATCH
}

// d_y[i] += alpha * d_x[i]: sums of partial results inside the step (clip_fft.py:271-275 --enforce: two cut sets contribute
// to one image gradient), kept on the C ABI so that no framework op runs on the path
int aph_axpy_f32(float* d_y, const float* d_x, float alpha, size_t n, void* stream_) {
  APH_TRY
  if (!d_y || !d_x) return aph_fail(APH_ERR_ARG, "aph_axpy_f32: null argument");
  if (n == 0) return APH_OK;
  unsigned grid = (unsigned)((n + 255) / 256);
  grid = grid > 2048u ? 2048u : grid;
  APH_LAUNCH(axpy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, d_y, d_x, alpha, n);
  return aph_check_launch("aph_axpy_f32");
  APH_CATCH
}

// The saved frame's conversion as ONE launch (it was four framework elementwise kernels on the step's stream):
// d_rgb f32 [3,H,W] -> d_out uint8 [H,W,3] = clip(rgb ** gamma * 255, 0, 255) truncated, as utils.checkout does on the host (utils.py:94-100).
int aph_rgb_to_u8(const float* d_rgb, int H, int W, float gamma, void* d_out, void* stream_) {
  APH_TRY
  if (!d_rgb || !d_out || H < 1 || W < 1 || !(gamma > 0.f)) return aph_fail(APH_ERR_ARG, "aph_rgb_to_u8: bad argument");
  const size_t n = (size_t)H * W;
  unsigned grid = (unsigned)((n + 255) / 256);
  grid = grid > 4096u ? 4096u : grid;
  APH_LAUNCH(rgb_to_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, d_rgb, (unsigned char*)d_out, n, gamma);
  return aph_check_launch("aph_rgb_to_u8");
  APH_CATCH
}

// One optimizer.step() on a flat f32 tensor.  d_m may be NULL when beta1 == 0; d_vmax NULL unless amsgrad.
// d_hyper: 8 device floats {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, sqrt(1-beta2^t), grad_scale}.
int aph_adam_step(float* d_p, const float* d_g, float* d_m, float* d_v, float* d_vmax, const float* d_hyper, int decoupled_wd,
                  size_t n, void* stream_) {
  APH_TRY
  if (!d_p || !d_g || !d_v || !d_hyper) return aph_fail(APH_ERR_ARG, "aph_adam_step: null argument");
  launch_adam(d_p, d_g, d_m, d_v, d_vmax, d_hyper, decoupled_wd, n, nullptr, (hipStream_t)stream_);
  return aph_check_launch("aph_adam_step");
  APH_CATCH
}

// The same update behind an overflow guard (mixed-precision training practice; the reference runs CLIP in fp16 without
// one): if the gradient holds a NaN / inf -- an fp16 overflow somewhere in the loss-scaled backward -- the step is skipped
// and d_guard[0] (int, running count of skipped steps) is incremented; d_guard[1] is scratch.  d_guard: 2 ints, zeroed once
// by the caller, who lowers its loss scale when the count moves.
int aph_adam_step_guarded(float* d_p, const float* d_g, float* d_m, float* d_v, float* d_vmax, const float* d_hyper, int decoupled_wd,
                          size_t n, int* d_guard, void* stream_) {
  APH_TRY
  if (!d_p || !d_g || !d_v || !d_hyper || !d_guard) return aph_fail(APH_ERR_ARG, "aph_adam_step_guarded: null argument");
  hipStream_t st = (hipStream_t)stream_;
  APH_LAUNCH(zero4_kernel, dim3(1), dim3(64), 0, st, d_guard + 1);
  launch_adam(d_p, d_g, d_m, d_v, d_vmax, d_hyper, decoupled_wd, n, d_guard, st);
  return aph_check_launch("aph_adam_step_guarded");
  APH_CATCH
}

}  // extern "C"
