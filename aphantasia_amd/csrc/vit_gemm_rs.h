// Small-M fp16 GEMM for the CLIP ViT linears (round 5), gfx950: every WAVE streams its own share of K.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        same contract as vit_gemm.h
//
// What the ring kernels of vit_gemm.h do at M ~ 1200 rows (a 24-cut shard: one of eight ranks; also C1 / C3 / C5): 228 tiles of 64 x 64 on
// 256 CUs is ONE 4-wave workgroup per CU, and its k loop is a chain of dependent steps -- counted DMA wait, s_barrier, fragment reads,
// eight MFMAs per wave -- with nothing else resident on the CU to fill the waits: 0.29-0.37 us per k-tile, 21 % of the matrix pipe of the
// CUs that are busy at all (profiles/r02_gemm_shapes_small_m.txt: fc2 17.9 us, dqkv 14.4 us for 5.7 / 4.2 GFLOP).
// Here a 64 x BN output tile is still one workgroup of four waves, but the waves split K instead of the tile:
//   * wave w owns the k-tiles kt = w, w + 4, ... (whole 128-byte lines of every operand row) and accumulates the FULL 64 x BN tile over
//     them: 16 + 4 NT fragment reads feed 16 NT MFMAs per k-tile, the ratio of the 256 x 128 kernel's wave tile;
//   * each wave has a PRIVATE ring of LDS slots (one k-step of 32: 64 x 32 of A + BN x 32 of Bt, 8 KiB at BN = 64) that only its own DMA
//     (global_load_lds_dwordx4) writes and only its own ds_read_b128 read: the covering counted vmcnt of the issuing wave is all the
//     ordering there is -- NO s_barrier in the main loop, the four waves drift freely and hide each other's waits;
//   * every byte of the A panel and of the weight rows is fetched exactly once per workgroup, as in the shared-ring kernels;
//   * at the end the four fp32 partial tiles meet in LDS (over the now idle rings), wave w sums row tile w in the fixed order
//     0, 1, 2, 3 (bitwise reproducible) and applies the epilogue to 4 NT consecutive columns per lane (apply8 of vit_gemm.h's epilogues).
// A k-step row is 64 bytes: a DMA instruction covers 16 rows x 4 pieces of 16 bytes; piece p of row r holds chunk p ^ ((-(r >> 2)) & 3),
// which makes the four lane groups of a ds_read_b128 fragment fetch (lanes {0-3, 12-15, 20-27}, ...) hit 16 different 16-byte bank groups.
// MFMA operands are swapped (weights as A fragment) with the weight rows of a 16-row tile permuted through the DMA source address so that
// lane l ends up with 4 NT CONSECUTIVE columns of token row l & 15.
#pragma once
#include "vit_gemm.h"

namespace aph {

template <int NT_, int NST_>
struct GemmRSCfg {
  static constexpr int NT = NT_, NST = NST_;                  // 16-column tiles per workgroup; ring slots per wave
  static constexpr int BM = 64, BN = 16 * NT, BK = 32, NWAVE = 4, NTHREAD = 256;
  static constexpr int SLOT = (BM + BN) * BK * 2;             // bytes per slot
  static constexpr int QT = BM / 16, QW = BN / 16, QPS = QT + QW;      // DMA instructions per slot (16 rows of 64 bytes each)
  static constexpr int RING = NST * SLOT;                     // bytes per wave
  static constexpr int PART = NWAVE * BM * BN * 4;            // four fp32 partial tiles
  static constexpr int SMEM = NWAVE * RING > PART ? NWAVE * RING : PART;
  static_assert(NT % 2 == 0 && NST >= 3 && (NST - 2) * QPS <= 63 && SMEM <= 160 * 1024, "ring depth / vmcnt range / LDS budget");
};
using GemmRS = GemmRSCfg<4, 5>;          // 64 x 64 tiles, 5 x 8 KiB per wave = all 160 KiB: three k-steps in flight per wave behind the one being read
using GemmRS4 = GemmRSCfg<4, 4>;         // ... 128 KiB
using GemmRSWide = GemmRSCfg<8, 3>;      // 64 x 128 tiles (wide outputs), 3 x 12 KiB per wave

__device__ __forceinline__ int rs_swz(int row) { return (0 - (row >> 2)) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef APH_EMU
  emu::wave_barrier();        // (the interpreter's lanes are fibers: every lane's copy has run once all of them are here)
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// this wave's fragment reads have returned: their slot may be refilled
__device__ __forceinline__ void rs_reads_done() {
#ifdef APH_EMU
  emu::wave_barrier();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// all but the newest `ahead` slots of this wave's DMA have landed
template <class C>
__device__ __forceinline__ void rs_wait(int ahead) {
  if (ahead <= 0) wait_vm<0>();
  else if (ahead == 1) wait_vm<C::QPS>();
  else if (ahead == 2) wait_vm<(2 * C::QPS <= 63 ? 2 * C::QPS : 63)>();
  else wait_vm<(3 * C::QPS <= 63 ? 3 * C::QPS : 63)>();
}

template <class C>
struct RSFrags {
  half8 t[4], w[C::NT];
};

template <class C, class Epi>
__global__ __launch_bounds__(256, 1) void gemm_rs_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt, int ldb, int M, int N,
                                                         int K, Epi epi) {
  APH_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  // XCD-aware tile order (as gemm_f16_kernel): every XCD gets one contiguous run of tiles, n-tiles fastest
  int m0, n0;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int ntn = N / C::BN;
    const int tm = tile / ntn;
    n0 = (tile - tm * ntn) * C::BN;
    m0 = tm * C::BM;
  }
  char* ring = smem + wave * C::RING;
  // DMA sources: instruction q covers 16 rows (lane >> 2) x 4 pieces (lane & 3)
  const int drow = lane >> 2, dpc = (lane & 3) ^ rs_swz(drow);      // logical chunk this lane fetches ((16 q + drow) >> 2 & 3 == drow >> 2)
  unsigned offT[C::QT], offW[C::QW];
#pragma unroll
  for (int q = 0; q < C::QT; ++q) {
    int am = m0 + 16 * q + drow;
    am = am < M ? am : M - 1;
    offT[q] = ((unsigned)am * (unsigned)lda + dpc * 8) * 2u;
  }
#pragma unroll
  for (int q = 0; q < C::QW; ++q) {
    // tile q, tile row i = drow is weight row 4 NT (i >> 2) + 4 q + (i & 3) of the workgroup's BN rows
    const int wr = n0 + 4 * C::NT * (drow >> 2) + 4 * q + (drow & 3);
    offW[q] = ((unsigned)wr * (unsigned)ldb + dpc * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bt);
  const int nk = K / GEMM_BK;
  const int ns = wave < nk ? 2 * ((nk - wave + 3) >> 2) : 0;        // k-steps of this wave: two per owned k-tile
  auto issue = [&](int s, int slot) {
    const size_t kb = (size_t)(wave + 4 * (s >> 1)) * 128 + (s & 1) * 64;
    char* dst = ring + slot * C::SLOT;
#pragma unroll
    for (int q = 0; q < C::QT; ++q) glds16(Ab + offT[q] + kb, dst + q * 1024);
#pragma unroll
    for (int q = 0; q < C::QW; ++q) glds16(Bb + offW[q] + kb, dst + C::BM * 64 + q * 1024);
  };
  const int frow = lane & 15, fpc = ((lane >> 4) ^ rs_swz(frow)) * 16;
  auto read = [&](RSFrags<C>& f, int slot) {
    const char* src = ring + slot * C::SLOT + frow * 64 + fpc;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) f.t[mt] = *reinterpret_cast<const half8*>(src + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) f.w[nt] = *reinterpret_cast<const half8*>(src + C::BM * 64 + nt * 1024);
  };
  f32x4 acc[4][C::NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const RSFrags<C>& f) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.w[nt], f.t[mt], acc[mt][nt]);
  };
  if (ns > 0) {
    RSFrags<C> f0, f1;
    const int pre = ns < C::NST - 1 ? ns : C::NST - 1;
    for (int s = 0; s < pre; ++s) issue(s, s);
    rs_wait<C>(pre - 1);                                             // k-step 0 has landed
    read(f0, 0);
    int sl_read = 1, sl_issue = C::NST - 1;                          // slot of k-step s + 1; slot of k-step s + NST - 1 (= the one k-step s - 1 vacated)
    for (int s = 0; s < ns; s += 2) {
      // even k-step s: fragments in f0
      rs_reads_done();
      if (s + C::NST - 1 < ns) { issue(s + C::NST - 1, sl_issue); sl_issue = sl_issue == C::NST - 1 ? 0 : sl_issue + 1; }
      {
        const int last = s + C::NST - 1 < ns ? s + C::NST - 1 : ns - 1;
        rs_wait<C>(last - (s + 1));                                   // k-step s + 1 has landed (ns is even: it exists)
      }
      read(f1, sl_read); sl_read = sl_read == C::NST - 1 ? 0 : sl_read + 1;
      mma(f0);
      // odd k-step s + 1: fragments in f1
      rs_reads_done();
      if (s + C::NST < ns) { issue(s + C::NST, sl_issue); sl_issue = sl_issue == C::NST - 1 ? 0 : sl_issue + 1; }
      if (s + 2 < ns) {
        const int last = s + C::NST < ns ? s + C::NST : ns - 1;
        rs_wait<C>(last - (s + 2));
        read(f0, sl_read); sl_read = sl_read == C::NST - 1 ? 0 : sl_read + 1;
      }
      mma(f1);
    }
  }
  // the four partial tiles meet in LDS: [source wave][row tile][column tile][lane] as f32x4
  __syncthreads();                                                   // every wave is done with its ring
  f32x4* part = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) part[((wave * 4 + mt) * C::NT + nt) * 64 + lane] = acc[mt][nt];
  __syncthreads();
  f32x4 sum[C::NT];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    sum[nt] = part[((0 * 4 + wave) * C::NT + nt) * 64 + lane];
#pragma unroll
    for (int src = 1; src < 4; ++src) sum[nt] += part[((src * 4 + wave) * C::NT + nt) * 64 + lane];
  }
  // lane: token row m0 + 16 wave + (lane & 15), columns n0 + 4 NT (lane >> 4) + 4 nt + r
  const int m = m0 + 16 * wave + (lane & 15);
  if (m < M) {
#pragma unroll
    for (int j = 0; j < C::NT / 2; ++j) epi.apply8(m, n0 + 4 * C::NT * (lane >> 4) + 8 * j, sum[2 * j], sum[2 * j + 1]);
  }
}

template <class C, class Epi>
inline void launch_gemm_rs(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  const dim3 grid((N / C::BN) * ((M + C::BM - 1) / C::BM));
  APH_ALLOW_SMEM((gemm_rs_kernel<C, Epi>), C::SMEM);
  APH_LAUNCH((gemm_rs_kernel<C, Epi>), grid, dim3(C::NTHREAD), C::SMEM, st, A, lda, Bt, ldb, M, N, K, epi);
}

// tile choice (tools/gemm_shapes_bench.py sweep): 64 x 128 tiles once they fill the chip about twice over, 64 x 64 below
inline int& gemm_rs_wide_min_tiles() {
  static int v = 512;
  return v;
}
template <class Epi>
inline void launch_gemm_rs_auto(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  const int wide_tiles = (N / GemmRSWide::BN) * ((M + GemmRSWide::BM - 1) / GemmRSWide::BM);
  if (N % GemmRSWide::BN == 0 && wide_tiles >= gemm_rs_wide_min_tiles()) launch_gemm_rs<GemmRSWide>(A, lda, Bt, ldb, M, N, K, epi, st);
  else launch_gemm_rs<GemmRS>(A, lda, Bt, ldb, M, N, K, epi, st);
}

}  // namespace aph
