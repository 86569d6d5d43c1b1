// fp16-in / fp32-accumulate MFMA GEMM for the CLIP ViT linears, gfx950.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        A: [M,K] f16 row-major, Bt: [N,K] f16 row-major
//
// Both operands are K-contiguous, so the forward (activations x weight^T) and the dgrad
// (d_out x weight, using a pre-transposed weight copy) run through the same kernel.
// 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_f16 tiles.
// Global -> registers -> LDS staging with the next tile's loads in flight during the MFMAs
// (T14 split), 16-byte XOR-swizzled LDS chunks so every ds_read_b128 fragment fetch is
// conflict-free (cdna_hip_programming.md section 5 / Guideline 4).
// Operands are swapped at the MFMA (weights as the A fragment) so each lane ends up holding
// 4 CONSECUTIVE output columns of one output row -> 16-byte epilogue accesses.
//
// Constraints: N % 128 == 0, K % 64 == 0 (true for every ViT-B linear); M is arbitrary
// (loads clamp the row, stores are predicated).
#pragma once
#include "aph_device.h"

namespace aph {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64;

// LDS tile: [128 rows][8 chunks of 8 halfs]; chunk c of row r lives at physical chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * GEMM_BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <class Epi>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt,
                                                       int ldb, int M, int N, int K, Epi epi) {
  __shared__ __attribute__((aligned(16))) half_t lds[2 * GEMM_BM * GEMM_BK];
  half_t* As = lds;
  half_t* Bs = lds + GEMM_BM * GEMM_BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (T1): workgroup b runs on XCD b % 8 (observed dispatch, speed only).  Give every
  // XCD one contiguous run of tiles, n-tiles fastest, so the column tiles that re-read the same 128-row A
  // panel share one L2 instead of pulling it through the fabric into eight.  Bijective for any grid size.
  int m0, n0;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int ntn = N / GEMM_BN;
    const int tm = tile / ntn;
    n0 = (tile - tm * ntn) * GEMM_BN;
    m0 = tm * GEMM_BM;
  }

  // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread
  const half_t* ga[4];
  const half_t* gb[4];
  int so[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = tid + 256 * r, row = q >> 3, c = q & 7;
    int am = m0 + row; am = am < M ? am : M - 1;
    ga[r] = A + (size_t)am * lda + c * 8;
    gb[r] = Bt + (size_t)(n0 + row) * ldb + c * 8;
    so[r] = lds_off(row, c);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  half8 ra[4], rb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ra[r] = *reinterpret_cast<const half8*>(ga[r]);
    rb[r] = *reinterpret_cast<const half8*>(gb[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *reinterpret_cast<half8*>(As + so[r]) = ra[r];
    *reinterpret_cast<half8*>(Bs + so[r]) = rb[r];
  }
  __syncthreads();

  const int nk = K / GEMM_BK;
  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      const int ko = (kt + 1) * GEMM_BK;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ra[r] = *reinterpret_cast<const half8*>(ga[r] + ko);
        rb[r] = *reinterpret_cast<const half8*>(gb[r] + ko);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *reinterpret_cast<const half8*>(As + lds_off(wm * 64 + t * 16 + frow, kk * 4 + fchunk));
        fb[t] = *reinterpret_cast<const half8*>(Bs + lds_off(wn * 64 + t * 16 + frow, kk * 4 + fchunk));
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_16x16x32_f16(fb[nt], fa[mt], acc[mt][nt]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<half8*>(As + so[r]) = ra[r];
        *reinterpret_cast<half8*>(Bs + so[r]) = rb[r];
      }
      __syncthreads();
    }
  }

  // lane (l) reg r of tile (mt, nt):  row m = .. + (l & 15), cols n = .. + (l >> 4) * 4 + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) epi(m, n0 + wn * 64 + nt * 16 + (lane >> 4) * 4, acc[mt][nt]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Large-M variant: 256x128x64 block tile, 8 waves (4x2, 64x64 each).
//  * operands stream straight into LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass)
//    through a 3-stage ring with COUNTED vmcnt waits and a raw s_barrier (one barrier per k-tile);
//  * the LDS -> register fragment loads are software-pipelined one k-step (32) ahead of the MFMAs, ACROSS the
//    tile barrier: while the matrix cores work on k-step s, the ds_reads of k-step s+1 are in flight, so the
//    LDS latency never sits between a barrier and the first MFMA of a tile.
// Arithmetic intensity of the tile is 85 flop per L2 byte (128x128: 64).  The LDS image written by the DMA is
// lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address and undone by the fragment reads
// (cdna_hip_programming.md rule 21).
// ---------------------------------------------------------------------------------------------
constexpr int GB_BM = 256, GB_BN = 128, GB_BK = 64;
constexpr int GB_STAGE = (GB_BM + GB_BN) * GB_BK;          // halfs per stage (48 KiB)
constexpr int GB_NSTAGE = 3;
constexpr int GB_SMEM = GB_NSTAGE * GB_STAGE * 2;          // bytes (144 KiB: one workgroup per CU)

struct GbFrags {
  half8 a[4], b[4];
};

__device__ __forceinline__ void gb_load_frags(GbFrags& f, const half_t* As, const half_t* Bs, int arow, int brow, int chunk) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f.a[t] = *reinterpret_cast<const half8*>(As + lds_off(arow + t * 16, chunk));
    f.b[t] = *reinterpret_cast<const half8*>(Bs + lds_off(brow + t * 16, chunk));
  }
}
__device__ __forceinline__ void gb_mma(f32x4 (&acc)[4][4], const GbFrags& f) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.b[nt], f.a[mt], acc[mt][nt]);
}

template <class Epi>
__global__ __launch_bounds__(512) void gemm_f16_big_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt,
                                                           int ldb, int M, int N, int K, Epi epi) {
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int ntn = N / GB_BN;
    const int tm = tile / ntn;
    n0 = (tile - tm * ntn) * GB_BN;
    m0 = tm * GB_BM;
  }
  // DMA assignment: one instruction = 64 lanes x 16 B = 8 tile rows.  A: 32 row groups (4 per wave), B: 16 (2 per wave).
  const half_t* ga[4];
  const half_t* gb[2];
  const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = (wave * 4 + k) * 8 + lrow;
    int am = m0 + row; am = am < M ? am : M - 1;
    ga[k] = A + (size_t)am * lda + ((pc ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = (wave * 2 + k) * 8 + lrow;
    gb[k] = Bt + (size_t)(n0 + row) * ldb + ((pc ^ ((row >> 1) & 7)) << 3);
  }
  auto issue = [&](int kt, int stage) {
    half_t* As = lds + stage * GB_STAGE;
    half_t* Bs = As + GB_BM * GB_BK;
    const int ko = kt * GB_BK;
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16(ga[k] + ko, As + (wave * 4 + k) * 8 * GB_BK);
#pragma unroll
    for (int k = 0; k < 2; ++k) glds16(gb[k] + ko, Bs + (wave * 2 + k) * 8 * GB_BK);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / GB_BK;
  const int arow = wm * 64 + (lane & 15), brow = wn * 64 + (lane & 15), fchunk = lane >> 4;
  // Ring protocol per tile kt (stage kt % 3):   own DMAs of tile kt retired (counted vmcnt: the 6 of tile kt+1
  // stay in flight) -> s_barrier (everyone's retired; everyone's fragment reads of tile kt-1 are complete, so its
  // stage may be refilled) -> issue tile kt+2 -> read fragments of tile kt.
  GbFrags f0, f1;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1) wait_vm_barrier<6>(); else wait_vm_barrier<0>();
  if (nk > 2) issue(2, 2);
  gb_load_frags(f0, lds, lds + GB_BM * GB_BK, arow, brow, fchunk);
  int st_cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* As = lds + st_cur * GB_STAGE;
    const half_t* Bs = As + GB_BM * GB_BK;
    gb_load_frags(f1, As, Bs, arow, brow, 4 + fchunk);     // k-step 1 of tile kt: in flight during the MFMAs below
    gb_mma(acc, f0);                                        // k-step 0 of tile kt
    st_cur = st_cur == GB_NSTAGE - 1 ? 0 : st_cur + 1;
    if (kt + 1 < nk) {
      wait_lgkm0();                                         // f1 has left LDS: stage st(kt) is dead for this wave
      if (kt + 2 < nk) wait_vm_barrier<6>(); else wait_vm_barrier<0>();
      // refill the stage just released.  The two waves that share a SIMD (w and w + 4) issue their DMA half a k-tile
      // apart, so one of them always has MFMAs to issue while the other sits in the vector-memory issue queue.
      if (wave < 4 && kt + 3 < nk) issue(kt + 3, st_cur == 0 ? GB_NSTAGE - 1 : st_cur - 1);
      const half_t* An = lds + st_cur * GB_STAGE;
      gb_load_frags(f0, An, An + GB_BM * GB_BK, arow, brow, fchunk);   // k-step 0 of tile kt+1: overlaps the MFMAs below
    }
    gb_mma(acc, f1);                                        // k-step 1 of tile kt
    if (wave >= 4 && kt + 3 < nk) issue(kt + 3, st_cur == 0 ? GB_NSTAGE - 1 : st_cur - 1);
  }
  // Epilogue through LDS: every wave parks its 64x64 fp32 accumulator tile in its own slice of the (now idle)
  // ring, then each lane picks up 8 CONSECUTIVE columns of one row -> 16/32-byte global accesses covering whole
  // 128-byte lines, and 4x fewer store instructions than storing straight from the MFMA C/D layout.
  __syncthreads();
  constexpr int CT_LD = 68;                                   // floats per staged row (pad: conflict-free b128 writes)
  float* ct = reinterpret_cast<float*>(smem) + wave * (64 * CT_LD);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *reinterpret_cast<f32x4*>(ct + (mt * 16 + (lane & 15)) * CT_LD + nt * 16 + (lane >> 4) * 4) = acc[mt][nt];
  wave_lds_fence();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(ct + r * CT_LD + c8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ct + r * CT_LD + c8 + 4);
    const int m = m0 + wm * 64 + r;
    if (m < M) epi.apply8(m, n0 + wn * 64 + c8, a, b);
  }
}

// ---- epilogues: operator() gets 4 consecutive columns n..n+3 of row m, apply8 gets 8 -------------------
__device__ __forceinline__ void store_h4(half_t* p, float a, float b, float c, float d) {
  half4 h = {(half_t)a, (half_t)b, (half_t)c, (half_t)d};
  *reinterpret_cast<half4*>(p) = h;
}
__device__ __forceinline__ void store_h8(half_t* p, const f32x4& a, const f32x4& b) {
  half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  *reinterpret_cast<half8*>(p) = h;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

struct EpiF16 {          // out = acc (+ bias)
  half_t* out; int ldo; const float* bias;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    if (bias) v += ld4(bias + n);
    store_h4(out + (size_t)m * ldo + n, v[0], v[1], v[2], v[3]);
  }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    if (bias) { a += ld4(bias + n); b += ld4(bias + n + 4); }
    store_h8(out + (size_t)m * ldo + n, a, b);
  }
};

struct EpiF32 {          // out = acc * scale  (fp32)
  float* out; int ldo; float scale;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const { st4(out + (size_t)m * ldo + n, v * scale); }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    st4(out + (size_t)m * ldo + n, a * scale);
    st4(out + (size_t)m * ldo + n + 4, b * scale);
  }
};

struct EpiResidual {     // out = res + acc + bias   (fp32 residual stream)
  float* out; const float* res; int ldo; const float* bias;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    st4(out + (size_t)m * ldo + n, ld4(res + (size_t)m * ldo + n) + v + ld4(bias + n));
  }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    (*this)(m, n, a);
    (*this)(m, n + 4, b);
  }
};

// QuickGELU g = u sigmoid(1.702 u) with u = acc + bias.  Besides g the forward stores the derivative
// dg/du = s (1 + 1.702 u (1 - s)) (same sigmoid s, f16), so the backward's epilogue is one multiply.
__device__ __forceinline__ void quick_gelu4(const f32x4& u, f32x4& g, f32x4& dg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = 1.0f / (1.0f + __expf(-1.702f * u[i]));
    g[i] = u[i] * s;
    dg[i] = s * (1.0f + 1.702f * u[i] * (1.0f - s));
  }
}
struct EpiGelu {
  half_t* dg; half_t* g; int ldo; const float* bias;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    v += ld4(bias + n);
    f32x4 gg, dd;
    quick_gelu4(v, gg, dd);
    store_h4(g + (size_t)m * ldo + n, gg[0], gg[1], gg[2], gg[3]);
    store_h4(dg + (size_t)m * ldo + n, dd[0], dd[1], dd[2], dd[3]);
  }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    a += ld4(bias + n); b += ld4(bias + n + 4);
    f32x4 ga, da, gb, db;
    quick_gelu4(a, ga, da);
    quick_gelu4(b, gb, db);
    store_h8(g + (size_t)m * ldo + n, ga, gb);
    store_h8(dg + (size_t)m * ldo + n, da, db);
  }
};

struct EpiGeluBwd {      // du = acc * dg/du (stored by the forward)
  half_t* out; const half_t* dg; int ldo;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    const half4 d = *reinterpret_cast<const half4*>(dg + (size_t)m * ldo + n);
    store_h4(out + (size_t)m * ldo + n, v[0] * (float)d[0], v[1] * (float)d[1], v[2] * (float)d[2], v[3] * (float)d[3]);
  }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    const half8 d = *reinterpret_cast<const half8*>(dg + (size_t)m * ldo + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] *= (float)d[i]; b[i] *= (float)d[4 + i]; }
    store_h8(out + (size_t)m * ldo + n, a, b);
  }
};

struct EpiPatchEmbed {   // token row s*T + 1 + p  <-  patch row s*P + p ;  + positional embedding
  float* x0; const float* pos; int D, P, T;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    const int s = m / P, p = m - s * P;
    st4(x0 + ((size_t)s * T + 1 + p) * D + n, v + ld4(pos + (size_t)(1 + p) * D + n));
  }
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    (*this)(m, n, a);
    (*this)(m, n + 4, b);
  }
};

template <class Epi>
inline void launch_gemm(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  if (M >= 2048) {
    static bool once = (APH_ALLOW_SMEM(gemm_f16_big_kernel<Epi>, GB_SMEM), true);
    (void)once;
    APH_LAUNCH(gemm_f16_big_kernel<Epi>, dim3((N / GB_BN) * ((M + GB_BM - 1) / GB_BM)), dim3(512), GB_SMEM, st, A, lda, Bt, ldb, M, N, K, epi);
    return;
  }
  APH_LAUNCH(gemm_f16_kernel<Epi>, dim3((N / GEMM_BN) * ((M + GEMM_BM - 1) / GEMM_BM)), dim3(256), 0, st, A, lda, Bt, ldb, M, N, K, epi);
}

}  // namespace aph
