// Non-GEMM kernels of the CLIP ViT forward / input-gradient backward:
// LayerNorm (fp32 statistics) -- multi-head attention lives in vit_attn.h --,
// class-token / positional embedding, ln_post + projection head.
// Follows openai/CLIP clip/model.py VisionTransformer / ResidualAttentionBlock (see
// oracle/clip_vit_ref.py for the restatement these kernels are tested against).
#pragma once
#include "aph_device.h"

namespace aph {

constexpr float kLnEps = 1e-5f;
constexpr int kHeadDim = 64;

// ---------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, row held in registers (D = 256 * NV).
// OUT_F16: h = LN(x) as f16 (GEMM operand);  else fp32 (ln_pre, feeding the residual stream).
// CLS_FILL: row t == 0 of every image is read as class_embedding + pos[0] (the patch-embed GEMM
//           only writes token rows 1..T-1) -- used by ln_pre.
// ---------------------------------------------------------------------------------
// gamma2 != NULL (fp32 output only): the NEXT LayerNorm of the same rows is applied to the row just produced, from registers:
//           out2 = f16 LN(out; gamma2, beta2) -- ln_pre followed by the first block's ln_1 as one kernel (the same arithmetic on
//           the same fp32 values as the two kernels it replaces: bit-identical).
template <int NV, bool OUT_F16, bool CLS_FILL>
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                              void* __restrict__ out, int M, int T, const float* __restrict__ cls, const float* __restrict__ pos,
                              float* __restrict__ x_fill, int xs, const float* __restrict__ gamma2 = nullptr,
                              const float* __restrict__ beta2 = nullptr, half_t* __restrict__ out2 = nullptr, int hilo = 0) {
  // hilo (f16 outputs only; the split-precision forward, aph_vit_forward_hilo): a row is written as [hi (D) | lo (D)] with
  // hi = f16(y), lo = f16(y - hi): the A operand of a GEMM over K = 2 D against the weights repeated along K
  constexpr int D = 256 * NV;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  f32x4 v[NV];
  const bool fill = CLS_FILL && (row % T == 0);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = i * 256 + lane * 4;
    if (fill) {
      v[i] = *reinterpret_cast<const f32x4*>(cls + d) + *reinterpret_cast<const f32x4*>(pos + d);
      *reinterpret_cast<f32x4*>(x_fill + (size_t)row * D + d) = v[i];
    } else {
      v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * xs * D + d);     // xs: input row stride (T = class rows only)
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float c = v[i][j] - mean; q += c * c; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + kLnEps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = i * 256 + lane * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + d), b = *reinterpret_cast<const f32x4*>(beta + d);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
    if (OUT_F16) {
      half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
      half_t* op = reinterpret_cast<half_t*>(out) + (size_t)row * (hilo ? 2 * D : D) + d;
      *reinterpret_cast<half4*>(op) = h;
      if (hilo) {
        half4 l = {(half_t)(o[0] - (float)h[0]), (half_t)(o[1] - (float)h[1]), (half_t)(o[2] - (float)h[2]), (half_t)(o[3] - (float)h[3])};
        *reinterpret_cast<half4*>(op + D) = l;
      }
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)row * D + d) = o;
      v[i] = o;
    }
  }
  if (!OUT_F16 && gamma2) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s2 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean2 = wave_sum(s2) * (1.0f / D);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float c = v[i][j] - mean2; q2 += c * c; }
    const float rstd2 = rsqrtf(wave_sum(q2) * (1.0f / D) + kLnEps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = i * 256 + lane * 4;
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma2 + d), b = *reinterpret_cast<const f32x4*>(beta2 + d);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean2) * rstd2 * g[j] + b[j];
      half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
      half_t* op = out2 + (size_t)row * (hilo ? 2 * D : D) + d;
      *reinterpret_cast<half4*>(op) = h;
      if (hilo) {
        half4 l = {(half_t)(o[0] - (float)h[0]), (half_t)(o[1] - (float)h[1]), (half_t)(o[2] - (float)h[2]), (half_t)(o[3] - (float)h[3])};
        *reinterpret_cast<half4*>(op + D) = l;
      }
    }
  }
}

// LayerNorm input-gradient:  dx = [res +] rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
// x = the LN input (statistics are recomputed from it), dy f16 or f32, res = incoming residual-stream
// gradient (fp32) or NULL.  Writes fp32 (out32, optional) and/or f16 (out16, optional).
// PATCH_ROWS: out16 row index is compacted s*T + t -> s*(T-1) + t-1 and class rows are dropped
// (feeds the patch-embedding dgrad GEMM).
// res_T > 0: only the rows with row % res_T == 0 have a residual (the last block saw class rows only, so the fp32 gradient
//           stream it hands to its ln_1 holds nothing else -- and need not be zero-filled first).
// x_b != NULL (not with PATCH_ROWS): the row just produced is at once the dy of the PREVIOUS LayerNorm over the same rows (input x_b,
//           gain gamma_b): that one's input-gradient goes to out16 in the PATCH_ROWS layout and nothing else is written -- the first
//           block's ln_1 backward followed by ln_pre's as one kernel (the same arithmetic on the same fp32 values: bit-identical).
template <int NV, bool DY_F16, bool PATCH_ROWS>
__global__ void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                              const void* res, float* __restrict__ out32, half_t* out16, int M, int T,
                              int xs, int res_T = 0, const float* __restrict__ x_b = nullptr, const float* __restrict__ gamma_b = nullptr, int res_f16 = 0) {
  // res_f16 [r6, measurement switch aph_vit_set_grad_stream_f16]: the incoming residual-stream gradient is the f16 copy the previous
  // LayerNorm backward wrote for its dgrad GEMM (`res` may then alias out16: a lane reads its elements before it writes them), and no
  // fp32 stream is kept (out32 = NULL): 73 instead of 117 MB per launch at 190 cuts -- at the price of one f16 rounding of the stream
  // per LayerNorm (24 of them); see DESIGN.md section 4 *Round 6* for what the loss-curve ensemble says about it
  constexpr int D = 256 * NV;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const size_t srow = (size_t)row * xs;       // x / res / outputs live at row stride xs (T = class rows only); dy is compact
  f32x4 v[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = i * 256 + lane * 4;
    v[i] = *reinterpret_cast<const f32x4*>(x + srow * D + d);
    f32x4 dyv;
    if (DY_F16) {
      const half4 h = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(dy) + (size_t)row * D + d);
      dyv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
      dyv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + (size_t)row * D + d);
    }
    g[i] = dyv * *reinterpret_cast<const f32x4*>(gamma + d);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[i][j] -= mean; q += v[i][j] * v[i][j]; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + kLnEps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[i][j] *= rstd; sg += g[i][j]; sgx += g[i][j] * v[i][j]; }
  sg = wave_sum(sg) * (1.0f / D);
  sgx = wave_sum(sgx) * (1.0f / D);
  size_t orow = srow;
  if (PATCH_ROWS) {
    const int s_ = row / T, t_ = row - s_ * T;
    if (t_ == 0) return;
    orow = s_ * (T - 1) + t_ - 1;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = i * 256 + lane * 4;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - sg - v[i][j] * sgx);
    if (res && (res_T == 0 || row % res_T == 0)) {
      if (res_f16) {
        const half4 h = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(res) + srow * D + d);
        o += f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        o += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(res) + srow * D + d);
      }
    }
    if (!PATCH_ROWS && x_b) { g[i] = o; continue; }
    if (out32) *reinterpret_cast<f32x4*>(out32 + srow * D + d) = o;
    if (out16) {
      half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
      *reinterpret_cast<half4*>(out16 + orow * D + d) = h;
    }
  }
  if (!PATCH_ROWS && x_b) {
    // second LayerNorm backward, dy = g[] (fp32), exactly ln_bwd_kernel<NV, false, true>(g, x_b, gamma_b, res = NULL)
    const int s_ = row / T, t_ = row - s_ * T;
    if (t_ == 0) return;
    const size_t prow = (size_t)s_ * (T - 1) + t_ - 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = i * 256 + lane * 4;
      v[i] = *reinterpret_cast<const f32x4*>(x_b + srow * D + d);
      g[i] = g[i] * *reinterpret_cast<const f32x4*>(gamma_b + d);
    }
    float sb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sb += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean_b = wave_sum(sb) * (1.0f / D);
    float qb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][j] -= mean_b; qb += v[i][j] * v[i][j]; }
    const float rstd_b = rsqrtf(wave_sum(qb) * (1.0f / D) + kLnEps);
    float sgb = 0.f, sgxb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][j] *= rstd_b; sgb += g[i][j]; sgxb += g[i][j] * v[i][j]; }
    sgb = wave_sum(sgb) * (1.0f / D);
    sgxb = wave_sum(sgxb) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = i * 256 + lane * 4;
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd_b * (g[i][j] - sgb - v[i][j] * sgxb);
      half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
      *reinterpret_cast<half4*>(out16 + prow * D + d) = h;
    }
  }
}

// ---------------------------------------------------------------------------------
// head: enc[s] = LN_post(x[s*T + 0]) @ proj        (proj [D, E] f32).  fp32 throughout.
// [r4] grid (ceil(S / 4), E/128), 1024 threads: a workgroup takes FOUR cuts over its 128 outputs, so the 393 KB slice of proj it streams is
// read once per four cuts (it was once per cut: 300 MB of L2 reads per launch at C2), and the D reduction is cut into EIGHT slices of
// threads (it was two): a thread's chain of dependent 16-load batches is 6 long instead of 24 -- the kernel is bound by that chain
// (23 us for 1.5 MB of weights), not by bytes.  The eight partial sums of an output are added in slice order (deterministic).
// ---------------------------------------------------------------------------------
constexpr int kHeadCuts = 4;
__global__ __launch_bounds__(1024) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ proj,
                                                       float* __restrict__ enc, int S, int T, int D, int E) {
  APH_DYN_SMEM(smem);
  float* y = reinterpret_cast<float*>(smem);     // [kHeadCuts][D]
  float* part = y + kHeadCuts * D;               // [kHeadCuts][8][128]
  const int s0 = blockIdx.x * kHeadCuts, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < kHeadCuts) {                         // LayerNorm of class row s0 + wave: one wave per row
    const int s = s0 + wave;
    if (s < S) {
      const float* row = x + (size_t)s * T * D;
      float a = 0.f;
      for (int d = lane; d < D; d += 64) a += row[d];
      const float mean = wave_sum(a) / D;
      float q = 0.f;
      for (int d = lane; d < D; d += 64) { const float c = row[d] - mean; q += c * c; }
      const float rstd = rsqrtf(wave_sum(q) / D + kLnEps);
      for (int d = lane; d < D; d += 64) y[wave * D + d] = (row[d] - mean) * rstd * gamma[d] + beta[d];
    } else {
      for (int d = lane; d < D; d += 64) y[wave * D + d] = 0.f;
    }
  }
  __syncthreads();
  const int el = threadIdx.x & 127, slice = threadIdx.x >> 7, e = blockIdx.y * 128 + el;
  const int dn = D >> 3, d0 = slice * dn;
  float acc[kHeadCuts];
#pragma unroll
  for (int c = 0; c < kHeadCuts; ++c) acc[c] = 0.f;
  if (e < E) {
    int d = d0;
    for (; d + 16 <= d0 + dn; d += 16) {           // 16 independent loads in flight per thread
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = proj[(size_t)(d + u) * E + e];
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < kHeadCuts; ++c) acc[c] += y[c * D + d + u] * w[u];
    }
    for (; d < d0 + dn; ++d) {
      const float w = proj[(size_t)d * E + e];
#pragma unroll
      for (int c = 0; c < kHeadCuts; ++c) acc[c] += y[c * D + d] * w;
    }
  }
#pragma unroll
  for (int c = 0; c < kHeadCuts; ++c) part[(c * 8 + slice) * 128 + el] = acc[c];
  __syncthreads();
  if (threadIdx.x < kHeadCuts * 128) {
    const int c = threadIdx.x >> 7, s = s0 + c;
    if (s < S && e < E) {
      float t = part[(c * 8) * 128 + el];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += part[(c * 8 + k) * 128 + el];
      enc[(size_t)s * E + blockIdx.y * 128 + el] = t;
    }
  }
}

// (Round 4, measured and rejected: this kernel with four cuts per workgroup -- thread = (quad of features, one of four slices of E), projT read
// as 16-byte quads once per four cuts, slice partials through LDS, one wave per cut for the LayerNorm backward: 52 us against 27 at C2 and at
// 24 cuts alike.  48 workgroups leave four fifths of the chip idle and the per-thread chain did not get shorter in time.)
// head backward: genc [S,E] -> the class-token rows of dx (fp32) and dx16; every other row of dx must have been zeroed
// by the caller (only the class token reaches the head; the other rows of dx16 are not read before they are rewritten).  projT = proj transposed [E, D].
// One workgroup per image, one thread per feature d (blockDim.x == D <= 1024).
__global__ __launch_bounds__(1024) void head_bwd_kernel(const float* __restrict__ genc, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ projT,
                                                       float* __restrict__ dx, half_t* __restrict__ dx16, int T, int D, int E) {
  APH_DYN_SMEM(smem);
  float* ge = reinterpret_cast<float*>(smem);   // [E]
  __shared__ float red[16];
  const int s = blockIdx.x, d = threadIdx.x;
  const float* row = x + (size_t)s * T * D;
  for (int e = threadIdx.x; e < E; e += blockDim.x) ge[e] = genc[(size_t)s * E + e];
  const float xv = row[d];
  const float mean = block_sum(xv, red) / D;                 // (block_sum's barriers also publish ge)
  const float c = xv - mean;
  const float rstd = rsqrtf(block_sum(c * c, red) / D + kLnEps);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int e = 0;
  for (; e + 32 <= E; e += 32) {                   // 32 independent loads in flight per thread (L2-latency bound loop: 16 batches at E = 512)
    float w[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) w[u] = projT[(size_t)(e + u) * D + d];
#pragma unroll
    for (int u = 0; u < 32; ++u) acc[u & 7] += ge[e + u] * w[u];
  }
  for (; e < E; ++e) acc[0] += ge[e] * projT[(size_t)e * D + d];
  const float g = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) * gamma[d];
  const float xh = c * rstd;
  const float sg = block_sum(g, red) / D;
  const float sgx = block_sum(g * xh, red) / D;
  const float v = rstd * (g - sg - xh * sgx);
  dx[(size_t)s * T * D + d] = v;
  dx16[(size_t)s * T * D + d] = (half_t)v;
}

}  // namespace aph
