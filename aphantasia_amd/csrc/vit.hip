// CLIP ViT visual tower: `model.encode_image(x)` forward and its INPUT-gradient backward
// (clip_fft.py:254 and the autograd pass of clip_fft.py:294 through it).  Weight gradients are
// never formed -- the optimisation only updates the image parameters.
//
// fp16 operands / fp32 accumulation on the matrix cores, fp32 residual stream, fp32 LayerNorm and
// softmax statistics; the backward chain carries a static loss scale (applied by the caller to
// d_enc, removed by `out_scale`) so fp16 gradient activations do not underflow.
#include <cstring>
#include <string>
#include <type_traits>

#include "aph_device.h"
#include "aph_host.h"
#include "vit_gemm.h"
#include "vit_gemm_ws.h"
#include "vit_gemm_rs.h"
#include "vit_ops.h"
#include "vit_attn.h"
#ifdef APH_EXPERIMENTS
#include "vit_block.h"
#endif

using namespace aph;

namespace {

struct Layer {
  half_t *w_qkv = nullptr, *w_qkvT = nullptr, *w_o = nullptr, *w_oT = nullptr;
  half_t* w_qkv2 = nullptr;            // [3D, 2D]: w_qkv repeated along K (split-precision forward)
  half_t *w_fc1 = nullptr, *w_fc1T = nullptr, *w_fc2 = nullptr, *w_fc2T = nullptr;
  float *b_qkv = nullptr, *b_o = nullptr, *b_fc1 = nullptr, *b_fc2 = nullptr;
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  // per-layer activations kept for the backward
  float *x_in = nullptr, *x_mid = nullptr, *lse = nullptr;
  half_t *qkv = nullptr, *att = nullptr, *u = nullptr;
};

}  // namespace

struct aph_vit {
  int res, patch, D, L, heads, E, T, P, Kp, max_batch;
  half_t *w_patch = nullptr, *w_patchT = nullptr, *w_patch2 = nullptr;      // w_patch2 [D, 2 Kp]: w_patch repeated along K
  float *cls = nullptr, *pos = nullptr, *ln_pre_g = nullptr, *ln_pre_b = nullptr, *ln_post_g = nullptr, *ln_post_b = nullptr;
  float *proj = nullptr, *projT = nullptr;
  std::vector<Layer> layers;
  float *x0 = nullptr, *x_last = nullptr;
  half_t *h = nullptr, *gact = nullptr;
  float *dx = nullptr, *dx2 = nullptr;      // fp32 residual-stream gradient (dx2: the second buffer of the fused backward's hand-over)
  float* delta = nullptr;              // attention backward row dots dO_i . O_i (T > 64 path)
  half_t *dx16 = nullptr, *du = nullptr, *dh = nullptr, *datt = nullptr, *dqkv = nullptr, *dx0_16 = nullptr;
  SplitKSpace sk;                      // split-K partials of the small-M GEMMs (per-rank shards, class-row GEMMs)
  char* arena = nullptr;
  size_t arena_bytes = 0;
  char* arena_hilo = nullptr;          // [r6] the K-repeated weight copies of the split-precision forward (w_patch2, w_qkv2): allocated by aph_vit_enable_hilo only
  size_t arena_hilo_bytes = 0;
  int n_set = 0;
  // optional per-launch timing of the GEMM family (bench.py roofline): HIP event pairs on the launch stream
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;     // pairs
  size_t prof_used = 0;
  double prof_flops = 0.0;
};

namespace {

struct Carver {
  char* base; size_t off = 0;
  template <typename T> T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n * sizeof(T) + 255) & ~(size_t)255;
    return p;
  }
};

// the second arena (aph_vit_enable_hilo): [N, 2 K] copies of the patch-embedding and QKV weights, every row twice along K -- the B operand of
// a GEMM over [hi | lo] activation rows.  85 MB at ViT-B/32 that the default (f16 everywhere) path never touches.
void carve_hilo(aph_vit* v, char* base, size_t* total) {
  Carver c{base};
  const size_t D = v->D, Kp = v->Kp;
  v->w_patch2 = c.take<half_t>(2 * D * Kp);
  for (auto& l : v->layers) l.w_qkv2 = c.take<half_t>(6 * D * D);
  *total = c.off;
}
// device [rows, cols] f16 -> device [rows, 2 cols]: every row twice along K
int repeat_rows_k(half_t* dst, const half_t* src, size_t rows, size_t cols) {
  const size_t w = cols * sizeof(half_t);
  if (hipMemcpy2D(dst, 2 * w, src, w, w, rows, hipMemcpyDeviceToDevice) != hipSuccess) return -1;
  if (hipMemcpy2D(reinterpret_cast<char*>(dst) + w, 2 * w, src, w, w, rows, hipMemcpyDeviceToDevice) != hipSuccess) return -1;
  return 0;
}

void carve(aph_vit* v, char* base, size_t* total) {
  Carver c{base};
  const size_t D = v->D, Mx = (size_t)v->max_batch * v->T, E = v->E, Kp = v->Kp, T = v->T;
  v->w_patch = c.take<half_t>(D * Kp); v->w_patchT = c.take<half_t>(D * Kp);
  v->cls = c.take<float>(D); v->pos = c.take<float>(T * D);
  v->ln_pre_g = c.take<float>(D); v->ln_pre_b = c.take<float>(D);
  v->ln_post_g = c.take<float>(D); v->ln_post_b = c.take<float>(D);
  v->proj = c.take<float>(D * E); v->projT = c.take<float>(D * E);
  for (auto& l : v->layers) {
    l.w_qkv = c.take<half_t>(3 * D * D); l.w_qkvT = c.take<half_t>(3 * D * D);
    l.w_o = c.take<half_t>(D * D); l.w_oT = c.take<half_t>(D * D);
    l.w_fc1 = c.take<half_t>(4 * D * D); l.w_fc1T = c.take<half_t>(4 * D * D);
    l.w_fc2 = c.take<half_t>(4 * D * D); l.w_fc2T = c.take<half_t>(4 * D * D);
    l.b_qkv = c.take<float>(3 * D); l.b_o = c.take<float>(D); l.b_fc1 = c.take<float>(4 * D); l.b_fc2 = c.take<float>(D);
    l.ln1_g = c.take<float>(D); l.ln1_b = c.take<float>(D); l.ln2_g = c.take<float>(D); l.ln2_b = c.take<float>(D);
    l.x_in = c.take<float>(Mx * D); l.x_mid = c.take<float>(Mx * D); l.lse = c.take<float>((size_t)v->max_batch * v->heads * T);
    l.qkv = c.take<half_t>(Mx * 3 * D); l.att = c.take<half_t>(Mx * D); l.u = c.take<half_t>(Mx * 4 * D);
  }
  v->x0 = c.take<float>(Mx * D); v->x_last = c.take<float>(Mx * D);
  v->h = c.take<half_t>(Mx * 2 * D); v->gact = c.take<half_t>(Mx * 4 * D);      // h: [hi | lo] rows in the split-precision forward
  v->dx = c.take<float>(Mx * D);
#ifdef APH_EXPERIMENTS       // scratch of measured-and-not-adopted paths only: the fused backward's hand-over buffer, the two-kernel attention backward's row dots
  v->dx2 = c.take<float>(Mx * D);
  v->delta = c.take<float>((size_t)v->max_batch * v->heads * T);
#endif
  v->dx16 = c.take<half_t>(Mx * D); v->du = c.take<half_t>(Mx * 4 * D); v->dh = c.take<half_t>(Mx * D);
  v->datt = c.take<half_t>(Mx * D); v->dqkv = c.take<half_t>(Mx * 3 * D); v->dx0_16 = c.take<half_t>(Mx * D);
  v->sk.ws_floats = (size_t)256 * GemmSmall::BM * GemmSmall::BN;       // choose_splits keeps tiles * splits <= 256
  v->sk.ws = c.take<float>(v->sk.ws_floats);
  *total = c.off;
}

// host fp32 [rows, cols] -> device fp16, optionally transposed
int upload_f16(half_t* dst, const float* src, size_t rows, size_t cols, bool transpose) {
  std::vector<half_t> tmp(rows * cols);
  if (!transpose) {
    for (size_t i = 0; i < rows * cols; ++i) tmp[i] = (half_t)src[i];
  } else {
    for (size_t r = 0; r < rows; ++r)
      for (size_t c = 0; c < cols; ++c) tmp[c * rows + r] = (half_t)src[r * cols + c];
  }
  return hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(half_t), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
// host fp32 [rows, cols] -> device fp16 [rows, 2 cols]: every row twice along K (the B operand of a hi | lo split GEMM)
int upload_f16_twice(half_t* dst, const float* src, size_t rows, size_t cols) {
  std::vector<half_t> tmp(rows * cols * 2);
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c) tmp[r * 2 * cols + c] = tmp[r * 2 * cols + cols + c] = (half_t)src[r * cols + c];
  return hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(half_t), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
int upload_f32(float* dst, const float* src, size_t rows, size_t cols, bool transpose) {
  if (!transpose) return hipMemcpy(dst, src, rows * cols * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
  std::vector<float> tmp(rows * cols);
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c) tmp[c * rows + r] = src[r * cols + c];
  return hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

// a launch of the GEMM family, with its HIP event pair when the profile is on (bench.py roofline); flops = its algorithmic FLOPs
template <class F>
void vtimed(aph_vit* v, double flops, hipStream_t st, F&& launch) {
  if (!v->prof_on) { launch(); return; }
  if (v->prof_used + 2 > v->prof_ev.size()) {
    hipEvent_t a, b;
    APH_HIP(hipEventCreate(&a)); APH_HIP(hipEventCreate(&b));
    v->prof_ev.push_back(a); v->prof_ev.push_back(b);
  }
  APH_HIP(hipEventRecord(v->prof_ev[v->prof_used], st));
  launch();
  APH_HIP(hipEventRecord(v->prof_ev[v->prof_used + 1], st));
  v->prof_used += 2;
  v->prof_flops += flops;
}
// kdiv: the split-precision forward runs a GEMM over K = 2 x the algorithmic K (hi | lo operand against the weights repeated along K): the
// profile counts the ALGORITHMIC FLOPs (roofline.achieved is algorithmic work over measured time)
template <class Epi>
void vgemm(aph_vit* v, const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st, int kdiv = 1) {
  vtimed(v, 2.0 * M * N * K / kdiv, st, [&] { launch_gemm(A, lda, Bt, ldb, M, N, K, epi, st, &v->sk); });
}

#ifdef APH_EXPERIMENTS
// Fused block kernels (vit_block.h) for short sequences: used while the batch has at most this many token rows (0 = never).  Above it the
// one-launch-per-operator path with the wave-specialised GEMM is the faster one (measured crossover: DESIGN.md section 4).
#ifndef APH_VIT_FUSED_MAX_ROWS_DEFAULT
#define APH_VIT_FUSED_MAX_ROWS_DEFAULT 0
#endif
int g_fused_max_rows = APH_VIT_FUSED_MAX_ROWS_DEFAULT;
inline bool vit_fused(const aph_vit* v, int S) { return v->T <= AT_T && v->D <= 1024 && (long long)S * v->T <= g_fused_max_rows; }
// the (cut, head) kernel with the attention behind the QKV GEMM holds 120 KiB of LDS -- one workgroup per CU -- so it only pays while all
// S x heads workgroups run at once (288 of them on 256 CUs take two rounds: 33 against 21 us at 24 cuts); otherwise LayerNorm + QKV go
// through the flat-row kernel and the attention stays a launch of its own.  0 = never, 1 = automatic, 2 = always (tests).
int g_fused_attn = 1;
inline bool vit_fused_attn(const aph_vit* v, int S) { return g_fused_attn == 2 || (g_fused_attn == 1 && S * v->heads <= gemm_persistent_wgs()); }

void launch_qkv_attn(aph_vit* v, const Layer& l, int S, hipStream_t st) {
  const int nv = v->D / 256;
  auto go = [&](auto tag) {
    constexpr int NV = decltype(tag)::value;
    launch_blk_qkv_attn<NV, 4>(l.x_in, l.ln1_g, l.ln1_b, l.w_qkv, l.b_qkv, l.qkv, l.att, l.lse, S, v->T, v->heads, st);
  };
  switch (nv) {
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 3: go(std::integral_constant<int, 3>{}); break;
    default: go(std::integral_constant<int, 4>{}); break;
  }
}
template <class Epi>
void launch_lnbwd_gemm(aph_vit* v, const half_t* dy, const float* x, const float* g, const float* res, float* out32, int res_T, int M, const half_t* Wt,
                       int N, Epi epi, hipStream_t st) {
  switch (v->D / 256) {
    case 1: launch_blk_lnbwd_gemm<1, 4>(dy, x, g, res, out32, res_T, M, Wt, N, epi, st); break;
    case 2: launch_blk_lnbwd_gemm<2, 4>(dy, x, g, res, out32, res_T, M, Wt, N, epi, st); break;
    case 3: launch_blk_lnbwd_gemm<3, 4>(dy, x, g, res, out32, res_T, M, Wt, N, epi, st); break;
    default: launch_blk_lnbwd_gemm<4, 4>(dy, x, g, res, out32, res_T, M, Wt, N, epi, st); break;
  }
}
template <class Epi>
void launch_ln_gemm(aph_vit* v, const float* x, int xs, int M, const float* g, const float* b, const half_t* Wt, int N, Epi epi, hipStream_t st) {
  switch (v->D / 256) {
    case 1: launch_blk_ln_gemm<1, 4>(x, xs, M, g, b, Wt, N, epi, st); break;
    case 2: launch_blk_ln_gemm<2, 4>(x, xs, M, g, b, Wt, N, epi, st); break;
    case 3: launch_blk_ln_gemm<3, 4>(x, xs, M, g, b, Wt, N, epi, st); break;
    default: launch_blk_ln_gemm<4, 4>(x, xs, M, g, b, Wt, N, epi, st); break;
  }
}

#else
inline bool vit_fused(const aph_vit*, int) { return false; }
#endif

// g2 / b2 / out2: the next LayerNorm of the same rows fused behind this one (ln_fwd_kernel)
template <bool OUT_F16, bool CLS>
void launch_ln_fwd(int nv, const float* x, const float* g, const float* b, void* out, int M, int T, const float* cls,
                   const float* pos, float* x_fill, hipStream_t st, int xs = 1, const float* g2 = nullptr, const float* b2 = nullptr,
                   half_t* out2 = nullptr, int hilo = 0) {
  const dim3 grid((M + 3) / 4), block(256);
  switch (nv) {
    case 1: APH_LAUNCH((ln_fwd_kernel<1, OUT_F16, CLS>), grid, block, 0, st, x, g, b, out, M, T, cls, pos, x_fill, xs, g2, b2, out2, hilo); break;
    case 2: APH_LAUNCH((ln_fwd_kernel<2, OUT_F16, CLS>), grid, block, 0, st, x, g, b, out, M, T, cls, pos, x_fill, xs, g2, b2, out2, hilo); break;
    case 3: APH_LAUNCH((ln_fwd_kernel<3, OUT_F16, CLS>), grid, block, 0, st, x, g, b, out, M, T, cls, pos, x_fill, xs, g2, b2, out2, hilo); break;
    default: APH_LAUNCH((ln_fwd_kernel<4, OUT_F16, CLS>), grid, block, 0, st, x, g, b, out, M, T, cls, pos, x_fill, xs, g2, b2, out2, hilo); break;
  }
}
// res_T: residual on the rows with row % res_T == 0 only;  x_b / g_b: the previous LayerNorm's backward fused behind this one (ln_bwd_kernel)
template <bool DY_F16, bool PATCH>
void launch_ln_bwd(int nv, const void* dy, const float* x, const float* g, const void* res, float* out32, half_t* out16, int M,
                   int T, hipStream_t st, int xs = 1, int res_T = 0, const float* x_b = nullptr, const float* g_b = nullptr, int res_f16 = 0) {
  const dim3 grid((M + 3) / 4), block(256);
  switch (nv) {
    case 1: APH_LAUNCH((ln_bwd_kernel<1, DY_F16, PATCH>), grid, block, 0, st, dy, x, g, res, out32, out16, M, T, xs, res_T, x_b, g_b, res_f16); break;
    case 2: APH_LAUNCH((ln_bwd_kernel<2, DY_F16, PATCH>), grid, block, 0, st, dy, x, g, res, out32, out16, M, T, xs, res_T, x_b, g_b, res_f16); break;
    case 3: APH_LAUNCH((ln_bwd_kernel<3, DY_F16, PATCH>), grid, block, 0, st, dy, x, g, res, out32, out16, M, T, xs, res_T, x_b, g_b, res_f16); break;
    default: APH_LAUNCH((ln_bwd_kernel<4, DY_F16, PATCH>), grid, block, 0, st, dy, x, g, res, out32, out16, M, T, xs, res_T, x_b, g_b, res_f16); break;
  }
}
// LayerNorm pairs of the first block as one kernel each way, and no zero fill of the fp32 gradient stream (aph_vit_set_fuse_ln(0): the
// separate kernels -- bit-identical, kept for the equivalence test)
int g_fuse_ln = 1;
// [r6] measurement switch (aph_vit_set_grad_stream_f16): the backward's residual-stream gradient kept in f16 only (see ln_bwd_kernel res_f16)
int g_grad_stream_f16 = 0;

// attention launches: T <= 64 one-tile kernels, 64 < T <= 256 the blocked kernels (NB = ceil(T / 64))
struct AttnArgs {
  const half_t* qkv; half_t* att; float* lse;        // forward: qkv -> att, lse
  const half_t* datt; float* delta; half_t* dqkv;    // backward: (qkv, att, lse, datt) -> dqkv; delta [S*heads*T] scratch for T > 64
  int S, T, heads;
};
template <int NB>
void launch_attn_fwd_g(const AttnArgs& a, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * NB * 8192;
  APH_ALLOW_SMEM((attn_fwd_mfma_g_kernel<NB>), smem);
  APH_LAUNCH((attn_fwd_mfma_g_kernel<NB>), dim3(a.S * a.heads), dim3(512), smem, st, a.qkv, a.att, a.lse, a.T, a.heads);
}
// blocked backward (64 < T <= 256): one kernel, P and dS formed once (attn_bwd_one_g_kernel); -DAPH_EXPERIMENTS builds can switch back to
// the dQ + dK/dV kernel pair it superseded (aph_attn_set_bwd_one(0))
#ifdef APH_EXPERIMENTS
int g_attn_bwd_one = 1;
#endif
template <int NB>
void launch_attn_bwd_g(const AttnArgs& a, hipStream_t st) {
#ifdef APH_EXPERIMENTS
  if (g_attn_bwd_one)
#endif
  {
    constexpr size_t smem = (size_t)(4 + 2 * NB) * 8192 + 2 * 64 * sizeof(float);
    APH_ALLOW_SMEM((attn_bwd_one_g_kernel<NB>), smem);
    APH_LAUNCH((attn_bwd_one_g_kernel<NB>), dim3(a.S * a.heads), dim3(512), smem, st, a.qkv, (const half_t*)a.att, a.datt, (const float*)a.lse, a.dqkv,
               a.T, a.heads);
    return;
  }
#ifdef APH_EXPERIMENTS
  constexpr size_t smem_q = (size_t)3 * NB * 8192, smem_kv = (size_t)4 * NB * 8192 + 2 * NB * 64 * sizeof(float);
  APH_ALLOW_SMEM((attn_bwd_dq_g_kernel<NB>), smem_q);
  APH_ALLOW_SMEM((attn_bwd_dkv_g_kernel<NB>), smem_kv);
  APH_LAUNCH((attn_bwd_dq_g_kernel<NB>), dim3(a.S * a.heads), dim3(512), smem_q, st, a.qkv, (const half_t*)a.att, a.datt, (const float*)a.lse, a.delta,
             a.dqkv, a.T, a.heads);
  APH_LAUNCH((attn_bwd_dkv_g_kernel<NB>), dim3(a.S * a.heads), dim3(512), smem_kv, st, a.qkv, a.datt, (const float*)a.lse, (const float*)a.delta,
             a.dqkv, a.T, a.heads);
#endif
}
void launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  const int T = a.T;
  if (T <= AT_T) APH_LAUNCH(attn_fwd_mfma_kernel, dim3(a.S * a.heads), dim3(256), 0, st, a.qkv, a.att, a.lse, T, a.heads);
  else if (T <= 128) launch_attn_fwd_g<2>(a, st);
  else if (T <= 192) launch_attn_fwd_g<3>(a, st);
  else launch_attn_fwd_g<4>(a, st);
}
// workgroups of the persistent one-tile backward: 6 per CU (3 are resident at a time -- its LDS footprint; the second half starts as
// the first finishes, which evens out the tail: measured 34.6 us one item per workgroup, 32.8 us with 3 per CU, 30.7 us with 6,
// 31.8 with 8, 34.5 with 12 at C2, profiles/r02_attn_bwd_persistent.txt), never more than there are (cut, head) items
inline int attn_bwd_wgs(int items) {
#ifdef APH_EMU
  return items < 3 ? items : 3;                     // exercises the item loop under the interpreter
#else
  constexpr int per_cu = 6;
  thread_local int dev_cached = -1, ncu = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return items;
  if (dev != dev_cached) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) return items;
    dev_cached = dev;
  }
  const int w = ncu * per_cu;
  return items < w ? items : w;
#endif
}
#ifdef APH_EXPERIMENTS
int g_attn_ablate = 0;       // aph_attn_set_ablate: measurement variants of the T <= 56 backward (WRONG results)
#endif
void launch_attn_bwd(const AttnArgs& a, hipStream_t st) {
  const int T = a.T, items = a.S * a.heads;
  if (T <= AT_T)       // (the split dQ / dKdV kernels with NB = 1 were measured slower here: 7.62 vs 7.35 ms per C2 step)
  {
#ifdef APH_EXPERIMENTS
    if (T <= AT_RB && g_attn_ablate == 1) { APH_LAUNCH((attn_bwd_mfma_kernel<AT_RB, 1>), dim3(attn_bwd_wgs(items)), dim3(256), 0, st, a.qkv, a.datt, (const float*)a.lse, a.dqkv, T, a.heads, items); return; }
    if (T <= AT_RB && g_attn_ablate == 2) { APH_LAUNCH((attn_bwd_mfma_kernel<AT_RB, 2>), dim3(attn_bwd_wgs(items)), dim3(256), 0, st, a.qkv, a.datt, (const float*)a.lse, a.dqkv, T, a.heads, items); return; }
    if (T <= AT_RB && g_attn_ablate == 3) { APH_LAUNCH((attn_bwd_mfma_kernel<AT_RB, 3>), dim3(attn_bwd_wgs(items)), dim3(256), 0, st, a.qkv, a.datt, (const float*)a.lse, a.dqkv, T, a.heads, items); return; }
#endif
    if (T <= AT_RB)
      APH_LAUNCH(attn_bwd_mfma_kernel<AT_RB>, dim3(attn_bwd_wgs(items)), dim3(256), 0, st, a.qkv, a.datt, (const float*)a.lse, a.dqkv, T, a.heads, items);
    else
      APH_LAUNCH(attn_bwd_mfma_kernel<AT_T>, dim3(attn_bwd_wgs(items)), dim3(256), 0, st, a.qkv, a.datt, (const float*)a.lse, a.dqkv, T, a.heads, items);
  }
  else if (T <= 128) launch_attn_bwd_g<2>(a, st);
  else if (T <= 192) launch_attn_bwd_g<3>(a, st);
  else launch_attn_bwd_g<4>(a, st);
}
inline AttnArgs attn_args(aph_vit* v, const Layer& l, int S) {
  return AttnArgs{(const half_t*)l.qkv, l.att, l.lse, (const half_t*)v->datt, v->delta, v->dqkv, S, v->T, v->heads};
}

}  // namespace

extern "C" {

// cfg mirrors clip.model.VisionTransformer(input_resolution, patch_size, width, layers, heads, output_dim)
int aph_vit_create(int input_resolution, int patch_size, int width, int layers, int heads, int output_dim, int max_batch,
                   aph_vit** out) {
  APH_TRY
  if (!out) return aph_fail(APH_ERR_ARG, "aph_vit_create: null out");
  if (width % 256 || width > 1024 || heads * kHeadDim != width)
    return aph_fail(APH_ERR_UNSUPPORTED, "aph_vit_create: width %d / heads %d unsupported (need head dim 64, width in {256,512,768,1024})", width, heads);
  if (input_resolution % patch_size || (3 * patch_size * patch_size) % 128 || output_dim < 1 || layers < 1 || max_batch < 1)
    return aph_fail(APH_ERR_UNSUPPORTED, "aph_vit_create: resolution %d / patch %d unsupported", input_resolution, patch_size);
  auto* v = new aph_vit();
  v->res = input_resolution; v->patch = patch_size; v->D = width; v->L = layers; v->heads = heads; v->E = output_dim;
  const int g = input_resolution / patch_size;
  v->P = g * g; v->T = v->P + 1; v->Kp = 3 * patch_size * patch_size; v->max_batch = max_batch;
  if (v->T > 256) { delete v; return aph_fail(APH_ERR_UNSUPPORTED, "aph_vit_create: %d tokens per image not supported", v->T); }
  v->layers.resize(layers);
  size_t total = 0;
  carve(v, nullptr, &total);
  const hipError_t me = hipMalloc((void**)&v->arena, total);
  if (me != hipSuccess) { delete v; return aph_fail(APH_ERR_HIP, "aph_vit_create: cannot allocate %zu bytes (%s)", total, hipGetErrorString(me)); }
  v->arena_bytes = total;
  carve(v, v->arena, &total);
  *out = v;
  return APH_OK;
  APH_CATCH
}

int aph_vit_destroy(aph_vit* v) {
  if (!v) return APH_OK;
  for (hipEvent_t e : v->prof_ev) (void)hipEventDestroy(e);
  (void)hipFree(v->arena);
  if (v->arena_hilo) (void)hipFree(v->arena_hilo);
  delete v;
  return APH_OK;
}

// [r6] Allocates and fills the K-repeated weight copies aph_vit_forward_hilo multiplies [hi | lo] activation rows with (85 MB at ViT-B/32).
// Call once, after the weights are loaded and outside any stream capture (it allocates and copies synchronously); idempotent.  The default
// path (aph_vit_forward: f16 operands everywhere) never needs it -- round 5 carried these copies in every handle.
int aph_vit_enable_hilo(aph_vit* v) {
  APH_TRY
  if (!v) return aph_fail(APH_ERR_ARG, "aph_vit_enable_hilo: null handle");
  if (v->arena_hilo) return APH_OK;
  if (v->n_set < 8 + 12 * v->L) return aph_fail(APH_ERR_ARG, "aph_vit_enable_hilo: weights not fully loaded (%d tensors)", v->n_set);
  size_t total = 0;
  carve_hilo(v, nullptr, &total);
  char* base = nullptr;
  const hipError_t me = hipMalloc((void**)&base, total);
  if (me != hipSuccess) { carve_hilo(v, nullptr, &total); return aph_fail(APH_ERR_HIP, "aph_vit_enable_hilo: cannot allocate %zu bytes (%s)", total, hipGetErrorString(me)); }
  carve_hilo(v, base, &total);
  int rc = repeat_rows_k(v->w_patch2, v->w_patch, v->D, v->Kp);
  for (auto& l : v->layers) rc |= repeat_rows_k(l.w_qkv2, l.w_qkv, 3 * (size_t)v->D, v->D);
  if (rc || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(base);
    carve_hilo(v, nullptr, &total);          // back to null pointers
    return aph_fail(APH_ERR_HIP, "aph_vit_enable_hilo: device copy failed");
  }
  v->arena_hilo = base;
  v->arena_hilo_bytes = total;
  return APH_OK;
  APH_CATCH
}

size_t aph_vit_workspace_bytes(const aph_vit* v) { return v ? v->arena_bytes + v->arena_hilo_bytes : 0; }

// Upload one tensor by its OpenAI checkpoint key (without the `visual.` prefix), fp32 host data.
// e.g. "conv1.weight", "transformer.resblocks.3.attn.in_proj_weight", "proj".
int aph_vit_set_weight(aph_vit* v, const char* name, const float* data, size_t count) {
  APH_TRY
  if (!v || !name || !data) return aph_fail(APH_ERR_ARG, "aph_vit_set_weight: null argument");
  const size_t D = v->D, E = v->E, Kp = v->Kp, T = v->T;
  const std::string n(name);
  auto need = [&](size_t want) { return count == want ? 0 : aph_fail(APH_ERR_ARG, "aph_vit_set_weight(%s): %zu elements, expected %zu", name, count, want); };
  int rc = 0;
  if (n == "conv1.weight") {
    if ((rc = need(D * Kp))) return rc;
    // [D, 3, p, p] (openai/CLIP) -> K order of the sampler's patch rows: pixel-major, channel fastest (sampler.hip patch_index)
    const size_t pp = (size_t)v->patch * v->patch;
    std::vector<float> perm(D * Kp);
    for (size_t d = 0; d < D; ++d)
      for (size_t c = 0; c < 3; ++c)
        for (size_t q = 0; q < pp; ++q) perm[d * Kp + q * 3 + c] = data[d * Kp + c * pp + q];
    rc = upload_f16(v->w_patch, perm.data(), D, Kp, false) | upload_f16(v->w_patchT, perm.data(), D, Kp, true) | (v->w_patch2 ? upload_f16_twice(v->w_patch2, perm.data(), D, Kp) : 0);
  }
  else if (n == "class_embedding") { if ((rc = need(D))) return rc; rc = upload_f32(v->cls, data, 1, D, false); }
  else if (n == "positional_embedding") { if ((rc = need(T * D))) return rc; rc = upload_f32(v->pos, data, T, D, false); }
  else if (n == "ln_pre.weight") { if ((rc = need(D))) return rc; rc = upload_f32(v->ln_pre_g, data, 1, D, false); }
  else if (n == "ln_pre.bias") { if ((rc = need(D))) return rc; rc = upload_f32(v->ln_pre_b, data, 1, D, false); }
  else if (n == "ln_post.weight") { if ((rc = need(D))) return rc; rc = upload_f32(v->ln_post_g, data, 1, D, false); }
  else if (n == "ln_post.bias") { if ((rc = need(D))) return rc; rc = upload_f32(v->ln_post_b, data, 1, D, false); }
  else if (n == "proj") { if ((rc = need(D * E))) return rc; rc = upload_f32(v->proj, data, D, E, false) | upload_f32(v->projT, data, D, E, true); }
  else if (n.rfind("transformer.resblocks.", 0) == 0) {
    const size_t p0 = strlen("transformer.resblocks.");
    const size_t dot = n.find('.', p0);
    if (dot == std::string::npos) return aph_fail(APH_ERR_ARG, "aph_vit_set_weight: bad key %s", name);
    const int li = atoi(n.substr(p0, dot - p0).c_str());
    if (li < 0 || li >= v->L) return aph_fail(APH_ERR_ARG, "aph_vit_set_weight: layer %d out of range", li);
    Layer& l = v->layers[li];
    const std::string k = n.substr(dot + 1);
    if (k == "attn.in_proj_weight") { if ((rc = need(3 * D * D))) return rc; rc = upload_f16(l.w_qkv, data, 3 * D, D, false) | upload_f16(l.w_qkvT, data, 3 * D, D, true) | (l.w_qkv2 ? upload_f16_twice(l.w_qkv2, data, 3 * D, D) : 0); }
    else if (k == "attn.in_proj_bias") { if ((rc = need(3 * D))) return rc; rc = upload_f32(l.b_qkv, data, 1, 3 * D, false); }
    else if (k == "attn.out_proj.weight") { if ((rc = need(D * D))) return rc; rc = upload_f16(l.w_o, data, D, D, false) | upload_f16(l.w_oT, data, D, D, true); }
    else if (k == "attn.out_proj.bias") { if ((rc = need(D))) return rc; rc = upload_f32(l.b_o, data, 1, D, false); }
    else if (k == "ln_1.weight") { if ((rc = need(D))) return rc; rc = upload_f32(l.ln1_g, data, 1, D, false); }
    else if (k == "ln_1.bias") { if ((rc = need(D))) return rc; rc = upload_f32(l.ln1_b, data, 1, D, false); }
    else if (k == "ln_2.weight") { if ((rc = need(D))) return rc; rc = upload_f32(l.ln2_g, data, 1, D, false); }
    else if (k == "ln_2.bias") { if ((rc = need(D))) return rc; rc = upload_f32(l.ln2_b, data, 1, D, false); }
    else if (k == "mlp.c_fc.weight") { if ((rc = need(4 * D * D))) return rc; rc = upload_f16(l.w_fc1, data, 4 * D, D, false) | upload_f16(l.w_fc1T, data, 4 * D, D, true); }
    else if (k == "mlp.c_fc.bias") { if ((rc = need(4 * D))) return rc; rc = upload_f32(l.b_fc1, data, 1, 4 * D, false); }
    else if (k == "mlp.c_proj.weight") { if ((rc = need(4 * D * D))) return rc; rc = upload_f16(l.w_fc2, data, D, 4 * D, false) | upload_f16(l.w_fc2T, data, D, 4 * D, true); }
    else if (k == "mlp.c_proj.bias") { if ((rc = need(D))) return rc; rc = upload_f32(l.b_fc2, data, 1, D, false); }
    else return aph_fail(APH_ERR_ARG, "aph_vit_set_weight: unknown key %s", name);
  } else return aph_fail(APH_ERR_ARG, "aph_vit_set_weight: unknown key %s", name);
  if (rc) return aph_fail(APH_ERR_HIP, "aph_vit_set_weight(%s): upload failed", name);
  v->n_set++;
  return APH_OK;
  APH_CATCH
}

// encode_image: d_patches f16 [S*P, 3*patch*patch] (patch-major, CLIP-normalised) -> d_enc f32 [S, output_dim]
// hilo: the SPLIT-PRECISION forward -- d_patches rows are [hi (Kp) | lo (Kp)] (APH_OUT_PATCH_F16_HILO) and the first LayerNorm of every
// block writes [hi (D) | lo (D)]: the patch-embedding and the QKV GEMMs then run over K = 2 Kp / 2 D against the weights repeated along K,
// i.e. with ~22 operand bits on the two activations whose f16 rounding dominates the input-gradient error on weights with realistic
// dynamic range (profiles/r04_precision_attribution.txt).  Everything else, the backward included, is unchanged.
static int vit_forward_impl(aph_vit* v, const void* d_patches, int S, float* d_enc, bool hilo, void* stream_) {
  if (!v || !d_patches || !d_enc) return aph_fail(APH_ERR_ARG, "aph_vit_forward: null argument");
  if (S < 1 || S > v->max_batch) return aph_fail(APH_ERR_ARG, "aph_vit_forward: batch %d outside 1..%d", S, v->max_batch);
  if (v->n_set < 8 + 12 * v->L) return aph_fail(APH_ERR_ARG, "aph_vit_forward: weights not fully loaded (%d tensors)", v->n_set);
  if (hilo && !v->arena_hilo)
    return aph_fail(APH_ERR_ARG, "aph_vit_forward_hilo: call aph_vit_enable_hilo(vit) once after loading the weights (the split-precision forward's "
                    "K-repeated weight copies are not allocated by default)");
  hipStream_t st = (hipStream_t)stream_;
  const int D = v->D, T = v->T, M = S * T, nv = D / 256;
  const int kx = hilo ? 2 : 1;
  v->sk.small_batch = M <= 128;   // see gemm_rs_mode(): the split-K small-M kernel only when the whole batch is small
  vgemm(v, (const half_t*)d_patches, kx * v->Kp, hilo ? v->w_patch2 : v->w_patch, kx * v->Kp, S * v->P, D, kx * v->Kp, EpiPatchEmbed{v->x0, v->pos, D, v->P, T}, st, kx);
  const bool fuse = g_fuse_ln != 0;
  const bool blk = !hilo && vit_fused(v, S);          // fused block kernels: LayerNorm inside the QKV / fc1 launches, attention behind the QKV GEMM
  launch_ln_fwd<false, true>(nv, v->x0, v->ln_pre_g, v->ln_pre_b, v->layers[0].x_in, M, T, v->cls, v->pos, v->x0, st, 1,
                             (fuse && !blk) ? v->layers[0].ln1_g : nullptr, (fuse && !blk) ? v->layers[0].ln1_b : nullptr, (fuse && !blk) ? v->h : nullptr, hilo ? 1 : 0);
  for (int li = 0; li < v->L; ++li) {
    Layer& l = v->layers[li];
    float* x_next = li + 1 < v->L ? v->layers[li + 1].x_in : v->x_last;
#ifdef APH_EXPERIMENTS
    if (blk && vit_fused_attn(v, S)) {
      vtimed(v, 2.0 * M * 3 * D * D + 4.0 * S * v->heads * T * T * 64, st, [&] { launch_qkv_attn(v, l, S, st); });
    } else if (blk) {
      vtimed(v, 2.0 * M * 3 * D * D, st, [&] { launch_ln_gemm(v, l.x_in, 1, M, l.ln1_g, l.ln1_b, l.w_qkv, 3 * D, EpiF16{l.qkv, 3 * D, l.b_qkv}, st); });
      launch_attn_fwd(attn_args(v, l, S), st);
    } else
#endif
    {
      if (!(fuse && li == 0)) launch_ln_fwd<true, false>(nv, l.x_in, l.ln1_g, l.ln1_b, v->h, M, T, nullptr, nullptr, nullptr, st, 1, nullptr, nullptr, nullptr, hilo ? 1 : 0);
      if (hilo) {
        // [r5] the lo half reaches the Q and K columns only; the V columns are summed over the hi half of the [hi | lo] rows alone (the first D of
        // the 2 D columns of A and of [W | W]).  What the lo half repairs is the cancellation in h . W on weights whose residual stream carries
        // large common offsets: through Q and K that error is amplified by the softmax, through V it enters the block linearly next to the f16
        // rounding V is stored with anyway (CPU model, tools/precision_attribution.py: single-step gradient error 5.2e-4 through Q / K, 1.8e-4
        // through V; on the GPU the single-step errors of the two forms are equal, profiles/r05_split_qk_only_ab.txt).  At full batch this is ONE
        // launch of the wave-specialised kernel with two k-loop lengths (vit_gemm_ws.h).  Batches below that kernel's threshold are launch-bound,
        // not MFMA-bound: they keep the plain launch over [hi | lo] on all 3 D columns (V a little more exact than it needs to be; a second
        // launch per block would cost more than the shorter sums save: C1 726 -> 690 steps/s when it was tried).
        const EpiF16 eq{l.qkv, 3 * D, l.b_qkv};
        if (D % 128 == 0 && gemm_takes_ws(M, 2 * D, 3 * D, 2 * D)) {
          vtimed(v, 2.0 * M * 3 * D * D, st, [&] { launch_gemm_ws(v->h, 2 * D, l.w_qkv2, 2 * D, M, 3 * D, 2 * D, eq, st, nullptr, D, D); });
        } else {
          vgemm(v, v->h, 2 * D, l.w_qkv2, 2 * D, M, 3 * D, 2 * D, eq, st, 2);      // (launch-bound sizes: one launch, the lo half on every column)
        }
      } else {
        vgemm(v, v->h, D, l.w_qkv, D, M, 3 * D, D, EpiF16{l.qkv, 3 * D, l.b_qkv}, st);
      }
      launch_attn_fwd(attn_args(v, l, S), st);
    }
    // Only the class token leaves the last block (VisionTransformer.forward: ln_post(x[:, 0, :])), so everything after
    // its attention runs on the S class rows alone: the same buffers addressed with a row pitch of T rows.
    const bool cls_only = li + 1 == v->L;
    const int Mr = cls_only ? S : M, rs = cls_only ? T : 1;
    vgemm(v, l.att, rs * D, l.w_o, D, Mr, D, D, EpiResidual{l.x_mid, l.x_in, rs * D, l.b_o}, st);
#ifdef APH_EXPERIMENTS
    if (blk) {
      vtimed(v, 2.0 * Mr * 4 * D * D, st, [&] { launch_ln_gemm(v, l.x_mid, rs, Mr, l.ln2_g, l.ln2_b, l.w_fc1, 4 * D, EpiGelu{l.u, v->gact, 4 * D, l.b_fc1}, st); });
    } else
#endif
    {
      launch_ln_fwd<true, false>(nv, l.x_mid, l.ln2_g, l.ln2_b, v->h, Mr, T, nullptr, nullptr, nullptr, st, rs);
      vgemm(v, v->h, D, l.w_fc1, D, Mr, 4 * D, D, EpiGelu{l.u, v->gact, 4 * D, l.b_fc1}, st);
    }
    vgemm(v, v->gact, 4 * D, l.w_fc2, 4 * D, Mr, D, 4 * D, EpiResidual{x_next, l.x_mid, rs * D, l.b_fc2}, st);
  }
  APH_ALLOW_SMEM(head_fwd_kernel, sizeof(float) * kHeadCuts * (D + 8 * 128));
  APH_LAUNCH(head_fwd_kernel, dim3((S + kHeadCuts - 1) / kHeadCuts, (v->E + 127) / 128), dim3(1024), sizeof(float) * kHeadCuts * (D + 8 * 128), st,
             (const float*)v->x_last, (const float*)v->ln_post_g, (const float*)v->ln_post_b, (const float*)v->proj, d_enc, S, T, D, v->E);
  return aph_check_launch("aph_vit_forward");
}
int aph_vit_forward(aph_vit* v, const void* d_patches, int S, float* d_enc, void* stream_) {
  APH_TRY
  return vit_forward_impl(v, d_patches, S, d_enc, false, stream_);
  APH_CATCH
}
// d_patches_hilo f16 [S*P, 2 * 3*patch*patch] (APH_OUT_PATCH_F16_HILO rows [hi | lo])
int aph_vit_forward_hilo(aph_vit* v, const void* d_patches_hilo, int S, float* d_enc, void* stream_) {
  APH_TRY
  return vit_forward_impl(v, d_patches_hilo, S, d_enc, true, stream_);
  APH_CATCH
}

// input-gradient of the last aph_vit_forward: d_genc f32 [S, output_dim] (already multiplied by the caller's
// loss scale) -> d_patch_grad f32 [S*P, 3*patch*patch] multiplied by out_scale (pass 1/loss_scale).
static int vit_backward_impl(aph_vit* v, const float* d_genc, int S, void* d_patch_grad, bool grad_f16, float out_scale, void* stream_) {
  if (!v || !d_genc || !d_patch_grad) return aph_fail(APH_ERR_ARG, "aph_vit_backward: null argument");
  if (S < 1 || S > v->max_batch) return aph_fail(APH_ERR_ARG, "aph_vit_backward: batch %d outside 1..%d", S, v->max_batch);
  hipStream_t st = (hipStream_t)stream_;
  const int D = v->D, T = v->T, M = S * T, nv = D / 256;
  v->sk.small_batch = M <= 128;   // see gemm_rs_mode(): the split-K small-M kernel only when the whole batch is small
  // only the class rows carry gradient out of the head: the fp32 stream starts from zero; dx16 needs no clearing -- the
  // last block reads and writes its class rows only (row pitch T), and its ln_1 backward rewrites every row
  const bool fuse = g_fuse_ln != 0;      // (then the last block's ln_1 backward takes its residual from the class rows only: no fill)
  const int s16 = (g_grad_stream_f16 != 0 && fuse && !vit_fused(v, S)) ? 1 : 0;      // f16-only gradient stream (measurement switch; needs the fused LayerNorm pairs' row conventions)
  if (!fuse) zero_fill_async(v->dx, sizeof(float) * (size_t)M * D, st);            // (a kernel node, not a memset node: see zero_fill_async)
  APH_LAUNCH(head_bwd_kernel, dim3(S), dim3(D), sizeof(float) * v->E, st, d_genc, (const float*)v->x_last,
             (const float*)v->ln_post_g, (const float*)v->projT, v->dx, v->dx16, T, D, v->E);
  // fused backward (vit_block.h, -DAPH_EXPERIMENTS builds): a block's closing ln_1 backward is not launched; it runs as the prologue of the NEXT (lower) block's fc2
  // dgrad -- `pending` carries it over: dy = v->dh, LayerNorm input = the upper block's x_in, residual = v->dx (only the rows % res_T == 0)
  const bool blk = vit_fused(v, S);
  bool pending = false;
  int pending_res_T = 0;
  for (int li = v->L - 1; li >= 0; --li) {
    Layer& l = v->layers[li];
    const bool cls_only = li + 1 == v->L;          // see aph_vit_forward: the last block's MLP / out-proj saw class rows only
    const int Mr = cls_only ? S : M, rs = cls_only ? T : 1;
    if (cls_only) zero_fill_async(v->datt, sizeof(half_t) * (size_t)M * D, st);   // no gradient into the other rows' attention output
    const float* res2 = v->dx;                      // residual of this block's ln_2 backward
#ifdef APH_EXPERIMENTS
    if (pending) {
      const Layer& up = v->layers[li + 1];
      vtimed(v, 2.0 * M * 4 * D * D, st, [&] {
        launch_lnbwd_gemm(v, v->dh, up.x_in, up.ln1_g, v->dx, v->dx2, pending_res_T, M, l.w_fc2T, 4 * D, EpiGeluBwd{v->du, l.u, 4 * D}, st);
      });
      res2 = v->dx2;
      pending = false;
    } else
#endif
    {
      vgemm(v, v->dx16, rs * D, l.w_fc2T, D, Mr, 4 * D, D, EpiGeluBwd{v->du, l.u, 4 * D}, st);
    }
    vgemm(v, v->du, 4 * D, l.w_fc1T, 4 * D, Mr, D, 4 * D, EpiF16{v->dh, D, nullptr}, st);
    if (s16) launch_ln_bwd<true, false>(nv, v->dh, l.x_mid, l.ln2_g, v->dx16, nullptr, v->dx16, Mr, T, st, rs, 0, nullptr, nullptr, 1);
    else launch_ln_bwd<true, false>(nv, v->dh, l.x_mid, l.ln2_g, res2, v->dx, v->dx16, Mr, T, st, rs);
    vgemm(v, v->dx16, rs * D, l.w_oT, D, Mr, D, D, EpiF16{v->datt, rs * D, nullptr}, st);
    launch_attn_bwd(attn_args(v, l, S), st);
    vgemm(v, v->dqkv, 3 * D, l.w_qkvT, 3 * D, M, D, 3 * D, EpiF16{v->dh, D, nullptr}, st);
    if (blk && li > 0) {      // deferred into block li - 1's first launch
      pending = true;
      pending_res_T = (fuse && cls_only) ? T : 0;
    }
    else if (fuse && li == 0)      // ln_1 backward and ln_pre backward as one kernel: writes the patch rows of dx0_16 only
      launch_ln_bwd<true, false>(nv, v->dh, l.x_in, l.ln1_g, s16 ? (const void*)v->dx16 : (const void*)v->dx, nullptr, v->dx0_16, M, T, st, 1, cls_only ? T : 0, v->x0, v->ln_pre_g, s16);
    else if (s16)
      launch_ln_bwd<true, false>(nv, v->dh, l.x_in, l.ln1_g, v->dx16, nullptr, v->dx16, M, T, st, 1, cls_only ? T : 0, nullptr, nullptr, 1);
    else
      launch_ln_bwd<true, false>(nv, v->dh, l.x_in, l.ln1_g, v->dx, v->dx, v->dx16, M, T, st, 1, (fuse && cls_only) ? T : 0);
  }
  if (!fuse) launch_ln_bwd<false, true>(nv, v->dx, v->x0, v->ln_pre_g, nullptr, nullptr, v->dx0_16, M, T, st);
  if (grad_f16) vgemm(v, v->dx0_16, D, v->w_patchT, D, S * v->P, v->Kp, D, EpiF16Scale{(half_t*)d_patch_grad, v->Kp, out_scale}, st);
  else vgemm(v, v->dx0_16, D, v->w_patchT, D, S * v->P, v->Kp, D, EpiF32{(float*)d_patch_grad, v->Kp, out_scale}, st);
  return aph_check_launch("aph_vit_backward");
}

int aph_vit_backward(aph_vit* v, const float* d_genc, int S, float* d_patch_grad, float out_scale, void* stream_) {
  APH_TRY
  return vit_backward_impl(v, d_genc, S, d_patch_grad, false, out_scale, stream_);
  APH_CATCH
}
int aph_vit_backward_h(aph_vit* v, const float* d_genc, int S, void* d_patch_grad_f16, float out_scale, void* stream_) {
  APH_TRY
  return vit_backward_impl(v, d_genc, S, d_patch_grad_f16, true, out_scale, stream_);
  APH_CATCH
}

// GEMM-family timing for bench.py: enable, run steps, then read {sum of launch durations [ms], launches, flops}
int aph_vit_profile(aph_vit* v, int on) {
  APH_TRY
  if (!v) return aph_fail(APH_ERR_ARG, "aph_vit_profile: null handle");
  v->prof_on = on != 0;
  v->prof_used = 0;
  v->prof_flops = 0.0;
  return APH_OK;
  APH_CATCH
}
int aph_vit_profile_read(aph_vit* v, double* ms_total, long long* launches, double* flops) {
  APH_TRY
  if (!v || !ms_total || !launches || !flops) return aph_fail(APH_ERR_ARG, "aph_vit_profile_read: null argument");
  double total = 0.0;
  for (size_t i = 0; i + 1 < v->prof_used; i += 2) {
    if (hipEventSynchronize(v->prof_ev[i + 1]) != hipSuccess) return aph_fail(APH_ERR_HIP, "aph_vit_profile_read: event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, v->prof_ev[i], v->prof_ev[i + 1]) != hipSuccess) return aph_fail(APH_ERR_HIP, "aph_vit_profile_read: elapsed failed");
    total += ms;
  }
  *ms_total = total; *launches = (long long)(v->prof_used / 2); *flops = v->prof_flops;
  return APH_OK;
  APH_CATCH
}

// MFMA shape of every GEMM main loop launched from now on: 0 = 16x16x32 (default), 1 = 32x32x16.  Returns the previous value.
// LayerNorm fusions of the first / last block on (1, default) or off (0).  Returns the previous value.
int aph_vit_set_fuse_ln(int on) {
  const int prev = g_fuse_ln;
  g_fuse_ln = on ? 1 : 0;
  return prev;
}

// [r6] 1 = the backward keeps its residual-stream gradient in f16 only (each LayerNorm backward reads the f16 copy its predecessor wrote and
// writes no fp32 stream); 0 (default) = fp32 stream.  Returns the previous value.  A measurement switch: see DESIGN.md section 4 *Round 6*.
int aph_vit_set_grad_stream_f16(int on) {
  const int prev = g_grad_stream_f16;
  g_grad_stream_f16 = on ? 1 : 0;
  return prev;
}

#ifdef APH_EXPERIMENTS
// measurement variants of the T <= 56 attention backward (WRONG results): 0 = the kernel, 1 = no products (zeros stored), 2 = no stores, 3 = loads + staging only
int aph_attn_set_ablate(int mode) {
  const int prev = g_attn_ablate;
  g_attn_ablate = mode < 0 || mode > 3 ? 0 : mode;
  return prev;
}
// largest batch (token rows S * T) that runs the fused block kernels of vit_block.h (0 = never).  Returns the previous value.
int aph_vit_set_fused_max_rows(int rows) {
  const int prev = g_fused_max_rows;
  g_fused_max_rows = rows < 0 ? 0 : rows;
  return prev;
}

// the (cut, head) LayerNorm + QKV + attention kernel inside the fused forward: 0 = never, 1 = while S x heads workgroups fit the chip in one
// round (default), 2 = always.  Returns the previous value.
int aph_vit_set_fused_attn(int mode) {
  const int prev = g_fused_attn;
  g_fused_attn = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return prev;
}

#endif  // APH_EXPERIMENTS

int aph_gemm_set_mfma32(int on) {
  const int prev = gemm_mfma32();
  gemm_mfma32() = on ? 1 : 0;
  return prev;
}

// tile-count threshold from which launch_gemm picks the wave-specialised persistent kernel (0 = never).  Returns the previous value.
int aph_gemm_set_ws_min_tiles(int tiles) {
  const int prev = gemm_ws_min_tiles();
  gemm_ws_min_tiles() = tiles < 0 ? 0 : tiles;
  return prev;
}

// small-M GEMMs (below the wave-specialised kernel's threshold) on the register-staged kernels of vit_gemm_rs.h (1, default) or on the shared-ring
// tile configurations of vit_gemm.h (0).  Returns the previous value.
#ifdef APH_EXPERIMENTS
// blocked attention backward (64 < tokens <= 256): 1 = one kernel (default), 0 = the dQ + dK/dV pair.  Returns the previous value.
int aph_attn_set_bwd_one(int on) {
  const int prev = g_attn_bwd_one;
  g_attn_bwd_one = on ? 1 : 0;
  return prev;
}
#endif
int aph_gemm_set_rs(int mode) {
  const int prev = gemm_rs_mode();
  gemm_rs_mode() = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return prev;
}
// row panels per tile-order group of the wave-specialised GEMM (vit_gemm_ws.h `coords`): 0 = automatic, k > 0 = force.  Returns the previous value.
int aph_gemm_set_ws_pgroup(int g) {
  const int prev = gemm_ws_pgroup_override();
  gemm_ws_pgroup_override() = g < 0 ? 0 : g;
  return prev;
}

// Measurement hook (bench.py roofline.peak_measured): a pure v_mfma_f32_16x16x32_f16 loop, 128 accumulator registers per wave, 8 waves
// per workgroup, no memory traffic inside the loop; operands from d_src (>= 8192 x 16 bytes: zeros run faster than random data -- the part
// is power limited).  FLOPs of one launch = blocks * 8 waves * iters * 32 MFMAs * 16384.
namespace {
__global__ __launch_bounds__(512) void mfma_rate_kernel(const half8* __restrict__ src, float* out, int iters) {
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(threadIdx.x * 16 + i) & 8191];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x * 16 + 8 + i) & 8191];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32_f16(b[j], a[i], acc[i][j]);
  }
  f32x4 s = acc[0][0];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];
}
}  // namespace
int aph_mfma_rate(int blocks, int iters, const void* d_src, float* d_out, void* stream_) {
  APH_TRY
  if (blocks < 1 || iters < 1 || !d_src || !d_out) return aph_fail(APH_ERR_ARG, "aph_mfma_rate: bad argument");
  APH_LAUNCH(mfma_rate_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream_, (const half8*)d_src, d_out, iters);
  return aph_check_launch("aph_mfma_rate");
  APH_CATCH
}

// Measurement hook: the wave-specialised GEMM with one of the ViT's real epilogues and per-tile shader-clock stamps.
// epi_kind 0 = f16 + bias (QKV), 1 = QuickGELU (two f16 outputs: d_out, d_out2), 2 = f32 residual (d_out f32 in/out pitch N, d_bias),
// 3 = no store.  d_trace: gridDim x 16 tiles x 4 uint64 {first k-tile done, main loop done, epilogue issued, -} of consumer wave 0, or null.
int aph_gemm_ws_probe(const void* d_A, const void* d_Bt, int M, int N, int K, void* d_out, void* d_out2, const float* d_bias, int epi_kind,
                      unsigned long long* d_trace, void* stream_) {
  APH_TRY
  if (!d_A || !d_Bt || !d_out || M < 1 || N % 128 || K % GEMM_BK || N > 4096 || !gemm8_addressable(M, K, N, K))
    return aph_fail(APH_ERR_ARG, "aph_gemm_ws_probe: bad shape");
  const half_t* A = (const half_t*)d_A;
  const half_t* B = (const half_t*)d_Bt;
  hipStream_t st = (hipStream_t)stream_;
  if (epi_kind == 0) launch_gemm_ws(A, K, B, K, M, N, K, EpiF16{(half_t*)d_out, N, d_bias}, st, d_trace);
  else if (epi_kind == 1 && d_out2 && d_bias) launch_gemm_ws(A, K, B, K, M, N, K, EpiGelu{(half_t*)d_out2, (half_t*)d_out, N, d_bias}, st, d_trace);
  else if (epi_kind == 2 && d_bias) launch_gemm_ws(A, K, B, K, M, N, K, EpiResidual{(float*)d_out, (const float*)d_out, N, d_bias}, st, d_trace);
  else if (epi_kind == 3) launch_gemm_ws(A, K, B, K, M, N, K, EpiNoStore{(float*)d_out, N}, st, d_trace);
  else return aph_fail(APH_ERR_ARG, "aph_gemm_ws_probe: bad epilogue kind / missing buffer");
  return aph_check_launch("aph_gemm_ws_probe");
  APH_CATCH
}

// Measurement hook: the register-staged small-M kernels with an f16 output and per-phase stamps of the chip-wide 100 MHz clock.
// kind 0 = split-K 64x64 (stamps: entry, first fragments read, main loop done, past the barrier, end), 1 = A-resident 64x256 (entry, fill
// issued, fill barrier passed, main loop done, end).  d_trace: (workgroups x 8) uint64 or NULL.
int aph_gemm_rs_probe(const void* d_A, const void* d_Bt, int M, int N, int K, void* d_out, int kind, unsigned long long* d_trace, void* stream_) {
  APH_TRY
#ifdef APH_EXPERIMENTS
  const bool fits = kind == 0 ? gemm_sk_fits(N, K) : gemm_ar_fits(N, K);
#else
  const bool fits = gemm_sk_fits(N, K);
#endif
  if (!d_A || !d_Bt || !d_out || M < 1 || !gemm8_addressable(M, K, N, K) || !fits) return aph_fail(APH_ERR_ARG, "aph_gemm_rs_probe: bad shape");
  const EpiF16 epi{(half_t*)d_out, N, nullptr};
#ifndef APH_EXPERIMENTS
  if (kind != 0) return aph_fail(APH_ERR_UNSUPPORTED, "aph_gemm_rs_probe: kinds 1 / 2 (A-resident kernels) exist in -DAPH_EXPERIMENTS builds only");
  launch_gemm_sk<4>((const half_t*)d_A, K, (const half_t*)d_Bt, K, M, N, K, epi, (hipStream_t)stream_, d_trace);
  return aph_check_launch("aph_gemm_rs_probe");
#else
  if (kind == 2) {        // d_Bt = the fragment-major image written by aph_gemm_pack_frag (experiment: K = 768 only)
    if (K != 768) return aph_fail(APH_ERR_ARG, "aph_gemm_rs_probe: kind 2 is instantiated for K = 768");
    using C = GemmAR<4>;
    APH_ALLOW_SMEM((gemm_arp_kernel<4, 24, 8, EpiF16>), C::smem(768));
    APH_LAUNCH((gemm_arp_kernel<4, 24, 8, EpiF16>), dim3((N / C::BN) * ((M + C::BM - 1) / C::BM)), dim3(256), C::smem(768), (hipStream_t)stream_,
               (const half_t*)d_A, K, (const half_t*)d_Bt, M, N, epi, d_trace);
    return aph_check_launch("aph_gemm_rs_probe");
  }
  if (kind == 0) launch_gemm_sk<4>((const half_t*)d_A, K, (const half_t*)d_Bt, K, M, N, K, epi, (hipStream_t)stream_, d_trace);
  else launch_gemm_ar<4, 8>((const half_t*)d_A, K, (const half_t*)d_Bt, K, M, N, K, epi, (hipStream_t)stream_, d_trace);
  return aph_check_launch("aph_gemm_rs_probe");
#endif
  APH_CATCH
}

#ifdef APH_EXPERIMENTS
// Bt [N, K] f16 -> fragment-major image for the A-resident kernel's 256-column groups (experiment hook)
int aph_gemm_pack_frag(const void* d_Bt, int N, int K, void* d_out, void* stream_) {
  APH_TRY
  if (!d_Bt || !d_out || N % 256 || K % 32) return aph_fail(APH_ERR_ARG, "aph_gemm_pack_frag: bad shape");
  const size_t total = (size_t)(N / 16) * (K / 32) * 64;
  APH_LAUNCH((pack_frag_kernel<4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (const half_t*)d_Bt, (half_t*)d_out, N, K);
  return aph_check_launch("aph_gemm_pack_frag");
  APH_CATCH
}

#endif

// the attention kernels alone (unit tests, micro-benchmarks): mode 0 = forward (qkv -> att, lse), 1 = backward
// ((qkv, att, lse, datt) -> dqkv).  qkv / dqkv [S*T, 3*heads*64] f16, att / datt [S*T, heads*64] f16, lse [S*heads*T] f32,
// d_delta: S*heads*T floats of scratch, needed by the backward when T > 64.
int aph_attn_test(const void* d_qkv, void* d_att, float* d_lse, const void* d_datt, float* d_delta, void* d_dqkv, int S, int T, int heads,
                  int mode, void* stream_) {
  APH_TRY
  if (!d_qkv || !d_att || !d_lse || S < 1 || T < 1 || T > 256 || heads < 1 || (mode != 0 && mode != 1) ||
      (mode == 1 && (!d_datt || !d_dqkv || (T > AT_T && !d_delta))))
    return aph_fail(APH_ERR_ARG, "aph_attn_test: bad argument");
  const AttnArgs a{(const half_t*)d_qkv, (half_t*)d_att, d_lse, (const half_t*)d_datt, d_delta, (half_t*)d_dqkv, S, T, heads};
  if (mode == 0) launch_attn_fwd(a, (hipStream_t)stream_);
  else launch_attn_bwd(a, (hipStream_t)stream_);
  return aph_check_launch("aph_attn_test");
  APH_CATCH
}

// plain C = A * Bt^T (f16 in, f32 out) -- exported for the GEMM unit tests and micro-benchmarks
int aph_gemm_f16(const void* d_A, const void* d_Bt, int M, int N, int K, float* d_C, void* stream_) {
  APH_TRY
  if (!d_A || !d_Bt || !d_C || M < 1 || N % 128 || K % GEMM_BK || N < 1 || K < 1)
    return aph_fail(APH_ERR_ARG, "aph_gemm_f16: need N %% 128 == 0 and K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
  launch_gemm((const half_t*)d_A, K, (const half_t*)d_Bt, K, M, N, K, EpiF32{d_C, N, 1.0f}, (hipStream_t)stream_);
  return aph_check_launch("aph_gemm_f16");
  APH_CATCH
}

// same with explicit leading dimensions (row pitches in elements) and tile configuration
// (0 = automatic, 1 = 64x64, 2 = 256x128, 4 = 256x256 phased [needs N % 256 == 0], 5 = 256x128 wave-specialised persistent, 8 / 9 = 64x64 split-K x2 / x4,
// 10 = 128x128 4-stage, 11 = 128x128 4 waves 2-stage (two workgroups per CU), 12 = 256x128 on 4 waves,
// 14 / 15 = 64x64 register-staged split-K (4 / 3 k-steps in flight per wave), 16 / 17 = 64x256 A-resident (8 / 4 k-steps in flight; N % 256 == 0, K <= 1024),
// 22 / 24 = 128x128 split-K x2 / x4) -- unit tests and tuning sweeps
#ifdef APH_EXPERIMENTS
static bool gemm_ar_addressable(int N, int K) { return gemm_ar_fits(N, K); }
#else
static bool gemm_ar_addressable(int, int) { return true; }      // (tile_cfg 16 / 17 are refused further down in product builds)
#endif
int aph_gemm_f16_ld(const void* d_A, int lda, const void* d_Bt, int ldb, int M, int N, int K, float* d_C, int tile_cfg, void* stream_) {
  APH_TRY
  const bool nostore = (tile_cfg & 0x100) != 0;
  tile_cfg &= 0xff;
  if (!d_A || !d_Bt || !d_C || M < 1 || N % 128 || K % GEMM_BK || N < 1 || K < 1 || lda < K || ldb < K || (lda & 7) || (ldb & 7) ||
      !(tile_cfg == 0 || tile_cfg == 1 || tile_cfg == 2 || tile_cfg == 4 || tile_cfg == 5 || (tile_cfg >= 8 && tile_cfg <= 12) || (tile_cfg >= 14 && tile_cfg <= 17) || tile_cfg == 22 || tile_cfg == 24) ||
      (tile_cfg >= 14 && tile_cfg <= 17 && !gemm8_addressable(M, lda, N, ldb)) || (tile_cfg >= 14 && tile_cfg <= 15 && !gemm_sk_fits(N, K)) || (tile_cfg >= 16 && tile_cfg <= 17 && !gemm_ar_addressable(N, K)) || (tile_cfg == 4 && (N % 256 || !gemm8_addressable(M, lda, N, ldb))) || (tile_cfg == 5 && (!gemm8_addressable(M, lda, N, ldb) || N > GemmWS::BIAS_MAX)))
    return aph_fail(APH_ERR_ARG, "aph_gemm_f16_ld: bad shape");
  const half_t* A = (const half_t*)d_A;
  const half_t* B = (const half_t*)d_Bt;
  const EpiF32 epi{d_C, N, 1.0f};
  hipStream_t st = (hipStream_t)stream_;
  if (nostore) {          // measurement only: the same main loops with the output stores compiled out of the taken path
    const EpiNoStore en{d_C, N};
    if (tile_cfg == 2) launch_gemm_cfg<GemmBig>(A, lda, B, ldb, M, N, K, en, st);
#ifdef APH_EXPERIMENTS
    else if (tile_cfg == 4) launch_gemm8(A, lda, B, ldb, M, N, K, en, st);
#endif
    else if (tile_cfg == 5) launch_gemm_ws_cfg<GemmWS>(A, lda, B, ldb, M, N, K, en, st, nullptr);
    else return aph_fail(APH_ERR_ARG, "aph_gemm_f16_ld: the no-store variant exists for tile_cfg 2 and 5 (4: -DAPH_EXPERIMENTS builds)");
    return aph_check_launch("aph_gemm_f16_ld");
  }
  if (tile_cfg == 1) launch_gemm_cfg<GemmSmall>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 2) launch_gemm_cfg<GemmBig>(A, lda, B, ldb, M, N, K, epi, st);
#ifdef APH_EXPERIMENTS
  else if (tile_cfg == 4) launch_gemm8(A, lda, B, ldb, M, N, K, epi, st);
#else
  else if (tile_cfg == 4) return aph_fail(APH_ERR_UNSUPPORTED, "aph_gemm_f16_ld: tile_cfg 4 (phased 256x256 kernel) exists in -DAPH_EXPERIMENTS builds only");
#endif
  else if (tile_cfg == 5) launch_gemm_ws_cfg<GemmWS>(A, lda, B, ldb, M, N, K, epi, st, nullptr);
  else if (tile_cfg == 8 || tile_cfg == 9 || tile_cfg == 22 || tile_cfg == 24) {                // split-K (2 / 4 ways) of the 64x64 configuration, private workspace
    static SplitKSpace sp;
    const int splits = (tile_cfg == 8 || tile_cfg == 22) ? 2 : 4;
    if (K / GEMM_BK < splits || (size_t)M * N > ((size_t)1 << 24)) return aph_fail(APH_ERR_ARG, "aph_gemm_f16_ld: shape not usable with split-K");
    if (!sp.ws) {
      sp.ws_floats = (size_t)4 << 24;
      if (hipMalloc((void**)&sp.ws, sp.ws_floats * sizeof(float)) != hipSuccess) return aph_fail(APH_ERR_HIP, "aph_gemm_f16_ld: split-K workspace");
    }
    if (tile_cfg >= 22) launch_gemm_splitk<GemmMidDeep8>(A, lda, B, ldb, M, N, K, epi, splits, sp, st);
    else launch_gemm_splitk<GemmSmall>(A, lda, B, ldb, M, N, K, epi, splits, sp, st);
  }
  else if (tile_cfg == 10) launch_gemm_cfg<GemmMidDeep8>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 11) launch_gemm_cfg<GemmPair>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 12) launch_gemm_cfg<GemmFat>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 14) launch_gemm_sk<4>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 15) launch_gemm_sk<3>(A, lda, B, ldb, M, N, K, epi, st);
#ifdef APH_EXPERIMENTS
  else if (tile_cfg == 16) launch_gemm_ar<4, 8>(A, lda, B, ldb, M, N, K, epi, st);
  else if (tile_cfg == 17) launch_gemm_ar<4, 4>(A, lda, B, ldb, M, N, K, epi, st);
#else
  else if (tile_cfg == 16 || tile_cfg == 17) return aph_fail(APH_ERR_UNSUPPORTED, "aph_gemm_f16_ld: tile_cfg 16 / 17 (A-resident kernel) exist in -DAPH_EXPERIMENTS builds only");
#endif
  else launch_gemm(A, lda, B, ldb, M, N, K, epi, st);
  return aph_check_launch("aph_gemm_f16_ld");
  APH_CATCH
}

}  // extern "C"
