// fp16-in / fp32-accumulate MFMA GEMM for the CLIP ViT linears, gfx950.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        A: [M,K] f16 row-major, Bt: [N,K] f16 row-major
//
// Both operands are K-contiguous, so the forward (activations x weight^T) and the dgrad (d_out x weight, using a
// pre-transposed weight copy) run through the same kernels.  Tile configurations (launch_gemm picks one per shape):
//   * 256x256x64 "phased" kernel (gemm8_f16_kernel): 8 waves (2x4 of 128x64), 2 x 64 KiB stages, four phases per k-tile,
//     the two wave groups one barrier apart -- shapes with >= 400 such tiles: fc1 / fc2^T / patch-embed dgrad (N = 3072) at full
//     batch (456 tiles).  QKV (N = 2304: 342 tiles = 1.34 rounds of 256 CUs) is NOT among them;
//   * 256x128x64, 8 waves (4x2 of 64x64), 3-stage ring, 144 KiB LDS -- QKV and every N = 768 GEMM at full batch;
//   * 128x128x64, 8 waves (4x2 of 32x64), 4-stage ring, 128 KiB LDS -- half-batch shards, wide outputs of small shards;
//   *  64x 64x64, 4 waves (2x2 of 32x32), 4-stage ring,  64 KiB LDS (2 workgroups per CU) -- small M; two-pass split-K
//     when only a handful of tiles exist (the class-row GEMMs of the last block).
// Common structure (cdna_hip_programming.md section 5):
//   * operands stream straight into LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass)
//     with COUNTED vmcnt waits and a raw s_barrier, 2-3 k-tiles in flight;
//   * the LDS image written by the DMA is lane-linear, so the XOR swizzle that makes every ds_read_b128 fragment
//     fetch conflict-free (0 SQ_LDS_BANK_CONFLICT measured) is applied to the per-lane SOURCE address (rule 21);
//   * ring kernels: LDS -> register fragment loads software-pipelined one k-step (32) ahead of the MFMAs, across the
//     barrier; the two waves that share a SIMD issue their DMA half a k-tile apart;
//   * operands are swapped at the MFMA (weights as the A fragment) and the accumulators leave through LDS, so every
//     lane stores 8 consecutive columns of one row: whole 128-byte lines, 16/32-byte accesses;
//   * XCD-aware tile order (T1): the column tiles that re-read one A panel run on the same XCD / L2.
// Constraints: N % 128 == 0 (N % 256 for the phased kernel), K % 64 == 0 (true for every ViT-B linear); M is arbitrary
// (loads clamp the row, stores are predicated).
#pragma once
#include "aph_device.h"
#include <cstdlib>
#include <type_traits>

namespace aph {

constexpr int GEMM_BK = 64;

// LDS tile: [rows][8 chunks of 8 halfs]; chunk c of row r lives at physical chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * GEMM_BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int WM_, int WN_, int TM_, int TN_, int NSTAGE_>
struct GemmCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, NSTAGE = NSTAGE_;
  static constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  static constexpr int NWAVE = WM * WN, NTHREAD = NWAVE * 64;
  static constexpr int STAGE = (BM + BN) * GEMM_BK;                 // halfs per stage
  static constexpr int GA = BM / 8 / NWAVE, GB = BN / 8 / NWAVE;    // DMA instructions (8 tile rows each) per wave per tile
  static constexpr int GPT = GA + GB;
  static constexpr int CT_LD = TN * 16 + 4;                         // staged accumulator row pitch (floats)
  static constexpr int SMEM = NSTAGE * STAGE * 2;                   // bytes
  // accumulator rows (in 16-row MFMA tiles) a wave stages per epilogue pass: all of them if the ring is big enough
  static constexpr int EP_MT = (NWAVE * TM * 16 * CT_LD * 4 <= SMEM) ? TM : 2;
  static_assert(BM % (8 * NWAVE) == 0 && BN % (8 * NWAVE) == 0, "DMA row groups must divide over the waves");
  static_assert(NWAVE * EP_MT * 16 * CT_LD * 4 <= SMEM && TM % EP_MT == 0, "epilogue staging must fit in the ring");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
};
using GemmBig = GemmCfg<4, 2, 4, 4, 3>;      // 256 x 128, 512 threads, 144 KiB:  85 flop per L2 byte
using GemmMidDeep8 = GemmCfg<4, 2, 2, 4, 4>; // 128 x 128, 512 threads (32x64 per wave), 128 KiB: 3 k-tiles in flight
using GemmSmall = GemmCfg<2, 2, 2, 2, 4>;    //  64 x  64, 256 threads,  64 KiB
// (Round 5, measured and rejected: the same 64 x 64 tile on EIGHT waves of 16 x 32 -- GemmCfg<4, 2, 1, 2, 4> -- on the reading that a cold stream
// fills a CU faster from more waves: equal to the four-wave configuration on every shape at M = 300 ... 4750 and in the step at 5 ... 95 cuts,
// profiles/r05_gemm_small8_{shapes,steps}.txt: the ring's DMA count per workgroup is what it is, whoever issues it.)
using GemmFat = GemmCfg<2, 2, 8, 4, 3>;      // 256 x 128, 256 threads: FOUR waves of 128x64 (one per SIMD): half the LDS fragment bytes per flop of GemmBig (experiment)
using GemmPair = GemmCfg<2, 2, 4, 4, 2>;     // 128 x 128, 256 threads (64x64 per wave), 64 KiB: TWO workgroups per CU, one in its epilogue while the other computes (experiment)

template <class C>
struct GemmFrags {
  half8 a[C::TM], b[C::TN];
};

template <class C>
__device__ __forceinline__ void gemm_load_frags(GemmFrags<C>& f, const half_t* As, const half_t* Bs, int arow, int brow, int chunk) {
#pragma unroll
  for (int t = 0; t < C::TM; ++t) f.a[t] = *reinterpret_cast<const half8*>(As + lds_off(arow + t * 16, chunk));
#pragma unroll
  for (int t = 0; t < C::TN; ++t) f.b[t] = *reinterpret_cast<const half8*>(Bs + lds_off(brow + t * 16, chunk));
}
template <class C>
__device__ __forceinline__ void gemm_mma(f32x4 (&acc)[C::TM][C::TN], const GemmFrags<C>& f) {
#pragma unroll
  for (int mt = 0; mt < C::TM; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::TN; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.b[nt], f.a[mt], acc[mt][nt]);
}

// "the DMAs of all but the newest `ahead` tiles of this wave have landed", then workgroup barrier
template <class C>
__device__ __forceinline__ void ring_wait(int ahead) {
  if (ahead <= 0) wait_vm_barrier<0>();
  else if (ahead == 1) wait_vm_barrier<C::GPT>();
  else wait_vm_barrier<2 * C::GPT>();
}

__device__ __forceinline__ void mfma_prio(int on) {
#ifndef APH_EMU
  if (on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); }
  else { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
#endif
}

// Split-K (latency-bound shapes: small M, long K): `splits` workgroups share one output tile, each running a contiguous
// share of the k-tiles and writing its fp32 partial tile to ws[split][M][N]; a second small kernel sums the partials in
// split order 0..splits-1 (fixed order: bitwise reproducible) and applies the epilogue.  The kernel boundary is the
// synchronisation: an in-kernel ticket scheme needs an agent-scope release per workgroup, which on this part writes back
// the whole L2 (measured 0.2 us per workgroup, serialised: 100-1000 us per launch).
struct SplitK {
  float* ws;          // [splits][M][N] floats
  int splits;
};

// 32x32x16 fragments of the same LDS image: row tile t (32 rows), k16 sub-step u of k32-step ks -> 16-byte chunk 4 ks + 2 u + (lane >> 5)
template <class C>
struct GemmFrags32 {
  half8 a[C::TM / 2][2], b[C::TN / 2][2];
};
template <class C>
__device__ __forceinline__ void gemm_load_frags32(GemmFrags32<C>& f, const half_t* As, const half_t* Bs, int arow, int brow, int chunk0) {
#pragma unroll
  for (int t = 0; t < C::TM / 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) f.a[t][u] = *reinterpret_cast<const half8*>(As + lds_off(arow + t * 32, chunk0 + 2 * u));
#pragma unroll
  for (int t = 0; t < C::TN / 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) f.b[t][u] = *reinterpret_cast<const half8*>(Bs + lds_off(brow + t * 32, chunk0 + 2 * u));
}
template <class C>
__device__ __forceinline__ void gemm_mma32(f32x16 (&acc)[C::TM / 2][C::TN / 2], const GemmFrags32<C>& f) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int mt = 0; mt < C::TM / 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::TN / 2; ++nt) acc[mt][nt] = mfma_32x32x16_f16(f.b[nt][u], f.a[mt][u], acc[mt][nt]);
}

// M32: the main loop runs on v_mfma_f32_32x32x16_f16 (half the MFMA instructions per k-tile; same LDS image, same DMA
// schedule, same epilogue reads) instead of v_mfma_f32_16x16x32_f16
template <class C, class Epi, bool SK = false, bool M32 = false>
__global__ __launch_bounds__(C::NTHREAD) void gemm_f16_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt,
                                                              int ldb, int M, int N, int K, Epi epi, SplitK sk) {
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch, speed only).  Every XCD gets one contiguous
  // run of tiles, n-tiles fastest.  Bijective for any grid size.
  int m0, n0, split = 0;
  {
    int nwg = gridDim.x, b = blockIdx.x;
    if (SK) { split = b % sk.splits; b /= sk.splits; nwg /= sk.splits; }
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int ntn = N / C::BN;
    const int tm = tile / ntn;
    n0 = (tile - tm * ntn) * C::BN;
    m0 = tm * C::BM;
  }
  // DMA assignment: one instruction = 64 lanes x 16 B = 8 tile rows of 128 B.
  const half_t* ga[C::GA];
  const half_t* gb[C::GB];
  const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int k = 0; k < C::GA; ++k) {
    const int row = (wave * C::GA + k) * 8 + lrow;
    int am = m0 + row; am = am < M ? am : M - 1;
    ga[k] = A + (size_t)am * lda + ((pc ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int k = 0; k < C::GB; ++k) {
    const int row = (wave * C::GB + k) * 8 + lrow;
    gb[k] = Bt + (size_t)(n0 + row) * ldb + ((pc ^ ((row >> 1) & 7)) << 3);
  }
  int kbeg = 0, nk = K / GEMM_BK;
  if (SK) {
    kbeg = (int)((long long)nk * split / sk.splits);
    nk = (int)((long long)nk * (split + 1) / sk.splits) - kbeg;
  }
  auto issue = [&](int kt, int stage) {
    half_t* As = lds + stage * C::STAGE;
    half_t* Bs = As + C::BM * GEMM_BK;
    const int ko = (kbeg + kt) * GEMM_BK;
#pragma unroll
    for (int k = 0; k < C::GA; ++k) glds16(ga[k] + ko, As + (wave * C::GA + k) * 8 * GEMM_BK);
#pragma unroll
    for (int k = 0; k < C::GB; ++k) glds16(gb[k] + ko, Bs + (wave * C::GB + k) * 8 * GEMM_BK);
  };

  static_assert(!M32 || (C::TM % 2 == 0 && C::TN % 2 == 0 && C::EP_MT == C::TM), "32x32 fragments need even tile counts and a one-pass epilogue");
  f32x4 acc[M32 ? 1 : C::TM][M32 ? 1 : C::TN];
  f32x16 acc32[M32 ? C::TM / 2 : 1][M32 ? C::TN / 2 : 1];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < C::TM / 2; ++i)
#pragma unroll
      for (int j = 0; j < C::TN / 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int frow = M32 ? (lane & 31) : (lane & 15);
  const int arow = wm * C::TM * 16 + frow, brow = wn * C::TN * 16 + frow, fchunk = M32 ? (lane >> 5) : (lane >> 4);
  // Ring protocol, tile t lives in stage t % NSTAGE.  Per tile kt: own DMAs of tile kt retired (counted vmcnt: newer
  // tiles stay in flight) -> s_barrier (everyone's retired; everyone's fragment reads of tile kt-1 are complete, so
  // its stage may be refilled) -> issue tile kt-1+NSTAGE -> read fragments of tile kt.
  GemmFrags<C> f0, f1;
  GemmFrags32<C> g0, g1;
  auto load0 = [&](const half_t* As_) {                 // k32-step 0 of a tile
    if constexpr (M32) gemm_load_frags32<C>(g0, As_, As_ + C::BM * GEMM_BK, arow, brow, fchunk);
    else gemm_load_frags<C>(f0, As_, As_ + C::BM * GEMM_BK, arow, brow, fchunk);
  };
  auto load1 = [&](const half_t* As_) {                 // k32-step 1
    if constexpr (M32) gemm_load_frags32<C>(g1, As_, As_ + C::BM * GEMM_BK, arow, brow, 4 + fchunk);
    else gemm_load_frags<C>(f1, As_, As_ + C::BM * GEMM_BK, arow, brow, 4 + fchunk);
  };
  // (s_setprio around these MFMA clusters was measured: no change on any ViT shape -- the two waves of a SIMD already alternate)
  auto mma0 = [&]() { if constexpr (M32) gemm_mma32<C>(acc32, g0); else gemm_mma<C>(acc, f0); };
  auto mma1 = [&]() { if constexpr (M32) gemm_mma32<C>(acc32, g1); else gemm_mma<C>(acc, f1); };
#pragma unroll
  for (int t = 0; t < C::NSTAGE - 1; ++t)
    if (t < nk) issue(t, t);
  ring_wait<C>((nk - 1 < C::NSTAGE - 2) ? nk - 1 : C::NSTAGE - 2);
  if (C::NSTAGE - 1 < nk) issue(C::NSTAGE - 1, C::NSTAGE - 1);
  load0(lds);
  int st_cur = 0;
  const bool late = wave >= C::NWAVE / 2;      // the SIMD partner of wave w - NWAVE/2: issues its DMA half a k-tile later
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* As = lds + st_cur * C::STAGE;
    const int st_free = st_cur;
    st_cur = st_cur == C::NSTAGE - 1 ? 0 : st_cur + 1;
    const int rem = nk - 2 - kt;
    load1(As);                                                                 // k-step 1 of tile kt: in flight during the MFMAs
    mma0();                                                                    // k-step 0 of tile kt
    if (kt + 1 < nk) {
      wait_lgkm0();                                                            // f1 has left LDS: stage st_free is dead for this wave
      ring_wait<C>(rem < C::NSTAGE - 2 ? rem : C::NSTAGE - 2);
      if (!late && kt + C::NSTAGE < nk) issue(kt + C::NSTAGE, st_free);
      load0(lds + st_cur * C::STAGE);                                          // k-step 0 of tile kt+1: overlaps the MFMAs below
    }
    mma1();                                                                    // k-step 1 of tile kt
    if (late && kt + 1 < nk && kt + C::NSTAGE < nk) issue(kt + C::NSTAGE, st_free);
  }
  // Epilogue through LDS: every wave parks its fp32 accumulator tile in its own slice of the (now idle) ring, then each
  // lane picks up 8 CONSECUTIVE columns of one row.
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem) + wave * (C::EP_MT * 16 * C::CT_LD);
  constexpr int CPR = C::TN * 2;                 // 8-column chunks per row
  constexpr int RPI = 64 / CPR;                  // rows covered by one pass of the wave
#pragma unroll
  for (int p0 = 0; p0 < C::TM; p0 += C::EP_MT) {
    wave_lds_fence();
    if constexpr (M32) {       // D[n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][m = lane & 31]: four runs of 4 consecutive columns per lane
#pragma unroll
      for (int mt = 0; mt < C::TM / 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::TN / 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(ct + (mt * 32 + (lane & 31)) * C::CT_LD + nt * 32 + 8 * g + 4 * (lane >> 5)) =
                f32x4{acc32[mt][nt][4 * g], acc32[mt][nt][4 * g + 1], acc32[mt][nt][4 * g + 2], acc32[mt][nt][4 * g + 3]};
    } else {
#pragma unroll
      for (int mt = 0; mt < C::EP_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::TN; ++nt)
          *reinterpret_cast<f32x4*>(ct + (mt * 16 + (lane & 15)) * C::CT_LD + nt * 16 + (lane >> 4) * 4) = acc[p0 + mt][nt];
    }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < C::EP_MT * 16 / RPI; ++it) {
      const int r = it * RPI + lane / CPR, c8 = (lane % CPR) * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(ct + r * C::CT_LD + c8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(ct + r * C::CT_LD + c8 + 4);
      const int m = m0 + (wm * C::TM + p0) * 16 + r;
      if (m < M) {
        if (SK) {
          float* o = sk.ws + ((size_t)split * M + m) * N + n0 + wn * C::TN * 16 + c8;
          *reinterpret_cast<f32x4*>(o) = a;
          *reinterpret_cast<f32x4*>(o + 4) = b;
        } else {
          epi.apply8(m, n0 + wn * C::TN * 16 + c8, a, b);
        }
      }
    }
  }
}

// second pass of split-K: thread = 8 consecutive columns of one row
template <class Epi>
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, Epi epi) {
  const int n8 = N >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * n8) return;
  const int m = (int)(idx / n8), n = (int)(idx - (size_t)m * n8) * 8;
  const float* p = ws + (size_t)m * N + n;
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  for (int sp = 1; sp < splits; ++sp) {
    p += (size_t)M * N;
    a += *reinterpret_cast<const f32x4*>(p);
    b += *reinterpret_cast<const f32x4*>(p + 4);
  }
  epi.apply8(m, n, a, b);
}

#ifdef APH_EXPERIMENTS       // a measured-and-superseded kernel family: no BASELINE configuration launches it (the wave-specialised kernel takes every
                             // shape it was tuned for); kept compilable for A/B runs only
// ---- phased 256x256x64 kernel (cdna_hip_programming.md section 5.5, "8-phase" schedule) ----------------------
// 8 waves as 2 (M) x 4 (N), 128x64 of C per wave, double-buffered 64 KiB stages.  A k-tile is worked off in FOUR
// phases, one C quadrant of the wave (64x32, 16 MFMAs over K = 64) each, in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0)
// so that consecutive phases share one operand's fragments: 24 ds_read_b128 feed 64 MFMAs.  A phase is
//     { fragment reads + 2 DMA instructions | s_barrier | 16 MFMAs at raised priority | s_barrier }
// and the waves of the second M half run ONE BARRIER BEHIND the first (`if (wr) s_barrier` up front): the two waves
// that share a SIMD alternate, one in its MFMA segment while the other issues LDS reads and DMA.
// The next k-tile streams in as four half-tiles (A0, B0, B1, A1: 16 KiB = 2 DMA instructions per wave each), one per
// phase, each >= 3 phases ahead of its first read; counted vmcnt(4) keeps the newest two in flight.
// Hazards.  RAW: the half-tile read in phase p+1 was issued in phase p-2; every wave retires its share (vmcnt(4):
// only the issues of phases p-1 and p may be outstanding) before the barrier that precedes those reads -- for the
// leading group that is the phase's second barrier, for the trailing group (one barrier behind) its first.
// WAR: a stage is refilled during the k-tile after the one it served, >= 2 phases after the trailing group's last
// read of the half-tile being replaced (B0 is kept in registers for phase 4, so phase 4 reads nothing).
struct Gemm8 {
  static constexpr int BM = 256, BN = 256, NTHREAD = 512, NWAVE = 8;
  static constexpr int STAGE = (BM + BN) * GEMM_BK;      // halfs
  static constexpr int SMEM = 2 * STAGE * 2;             // bytes (128 KiB)
  static constexpr int CT_LD = 64 + 4, EP_MT = 2;
};

__device__ __forceinline__ void phase_barrier(bool wait, bool last) {
  // `wait`: this wave's DMA shares for the next phase's reads must have landed (see RAW above)
  if (!wait) wait_vm_barrier<63>();
  else if (last) wait_vm_barrier<0>();
  else wait_vm_barrier<4>();
}

template <class Epi, bool M32 = false>
__global__ __launch_bounds__(512) void gemm8_f16_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt, int ldb,
                                                        int M, int N, int K, Epi epi, int ntiles) {
  using C = Gemm8;
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // PERSISTENT TILE LOOP.  The grid is min(ntiles, CUs) workgroups (one per CU: 128 KiB of LDS); workgroup b runs on XCD
  // b % 8 (observed dispatch, speed only).  XCD x owns one contiguous run of tiles (n-tiles fastest, so the tiles in flight
  // on one L2 share A panels); its workgroups walk that run with stride = workgroups on the XCD.  With gridDim.x == ntiles
  // this is the one-tile-per-workgroup order of the ring kernels.  Bijective for any grid size.
  int tile, tile_end, tile_step;
  {
    const int nwg = gridDim.x, b = blockIdx.x, G = nwg < 8 ? nwg : 8;      // G = 8 on the device whenever ntiles >= 8
    const int q = ntiles / G, r = ntiles - q * G, xcd = b % G;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile_end = start + q + (xcd < r ? 1 : 0);
    tile_step = (nwg - xcd + G - 1) / G;
    tile = start + b / G;
  }
  const int ntn = N / C::BN;
  // DMA shares: half-tile X, instruction i of this wave covers 8 consecutive tile rows starting at row0(X, i)
  const int lrow = lane >> 3, pc = lane & 7;
  // operand addresses = wave-uniform base (SGPR pair: matrix + k offset) + a 32-bit per-lane byte offset (the instruction's
  // saddr form): 6 VGPRs instead of 6 pointers, and no vector address arithmetic in the main loop
  unsigned oA[2][2];            // [half][i]
  unsigned oB[2];               // [i], half 0; half 1 = + 32 rows (folded into the scalar base)
  int m0 = 0, n0 = 0;
  auto setup = [&](int t) {     // operand offsets of tile t
    const int tm = t / ntn;
    n0 = (t - tm * ntn) * C::BN;
    m0 = tm * C::BM;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q0 = (wave * 2 + i) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = (q0 >> 6) * 128 + h * 64 + (q0 & 63) + lrow;
        int am = m0 + row; am = am < M ? am : M - 1;
        oA[h][i] = ((unsigned)am * (unsigned)lda + ((pc ^ ((row >> 1) & 7)) << 3)) * 2u;
      }
      const int brow = (q0 >> 5) * 64 + (q0 & 31) + lrow;
      oB[i] = ((unsigned)(n0 + brow) * (unsigned)ldb + ((pc ^ ((brow >> 1) & 7)) << 3)) * 2u;
    }
  };
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bt);
  const size_t bhalf = (size_t)32 * ldb * 2;
  auto issue_a = [&](int h, int kt, half_t* stage) {
    const char* base_k = Ab + (size_t)kt * (GEMM_BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(base_k + oA[h][i], stage + (((wave * 2 + i) >> 3) * 128 + h * 64 + (((wave * 2 + i) * 8) & 63)) * GEMM_BK);
  };
  auto issue_b = [&](int h, int kt, half_t* stage) {
    const char* base_k = Bb + (size_t)kt * (GEMM_BK * 2) + (h ? bhalf : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(base_k + oB[i], stage + C::BM * GEMM_BK + (((wave * 2 + i) >> 2) * 64 + h * 32 + (((wave * 2 + i) * 8) & 31)) * GEMM_BK);
  };

  // accumulators and fragments: 16x16x32 -> acc[8][4] f32x4, A [row tile 16][k32 step], B [half][col tile 16][k32 step];
  //                             32x32x16 -> acc32[4][2] f32x16, A [row tile 32][k16 step], B [half][k16 step] (one 32-col tile per half)
  f32x4 acc[M32 ? 1 : 8][M32 ? 1 : 4];
  f32x16 acc32[M32 ? 4 : 1][M32 ? 2 : 1];
  half8 fa[4][2], fb[2][2][2];
  half8 ga[2][4], gb[2][4];
  const int frow = M32 ? (lane & 31) : (lane & 15);
  const int arow = wr * 128 + frow, brow = wc * 64 + frow, fchunk = M32 ? (lane >> 5) : (lane >> 4);
  auto read_a = [&](int h, const half_t* stage) {
    if constexpr (M32) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) ga[t][u] = *reinterpret_cast<const half8*>(stage + lds_off(arow + h * 64 + t * 32, 2 * u + fchunk));
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[t][ks] = *reinterpret_cast<const half8*>(stage + lds_off(arow + h * 64 + t * 16, ks * 4 + fchunk));
    }
  };
  auto read_b = [&](int h, const half_t* stage) {
    if constexpr (M32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) gb[h][u] = *reinterpret_cast<const half8*>(stage + C::BM * GEMM_BK + lds_off(brow + h * 32, 2 * u + fchunk));
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          fb[h][t][ks] = *reinterpret_cast<const half8*>(stage + C::BM * GEMM_BK + lds_off(brow + h * 32 + t * 16, ks * 4 + fchunk));
    }
  };
  auto quadrant = [&](int ah, int bh) {
    mfma_prio(1);
    if constexpr (M32) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc32[ah * 2 + mt][bh] = mfma_32x32x16_f16(gb[bh][u], ga[mt][u], acc32[ah * 2 + mt][bh]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[ah * 4 + mt][bh * 2 + nt] = mfma_16x16x32_f16(fb[bh][nt][ks], fa[mt][ks], acc[ah * 4 + mt][bh * 2 + nt]);
    }
    mfma_prio(0);
  };

  const int nk = K / GEMM_BK;
  const bool lead = wr == 0;
  int base = 0;                                  // k-tile kt of the current tile lives in stage (base + kt) & 1
  setup(tile);
  issue_a(0, 0, lds); issue_b(0, 0, lds); issue_b(1, 0, lds); issue_a(1, 0, lds);
  for (;;) {
    if constexpr (M32) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // k-tile 0 has landed (issued above, or under the previous tile's epilogue); the previous tile's staging reads and
    // stores are behind every wave
    wait_vm_barrier<0>();
    if (!lead) wait_vm_barrier<63>();            // trailing group: one barrier behind from here on
    for (int kt = 0; kt < nk; ++kt) {
      half_t* cur = lds + ((base + kt) & 1) * C::STAGE;
      half_t* nxt = lds + ((base + kt + 1) & 1) * C::STAGE;
      const bool more = kt + 1 < nk;
      // phase 1: quadrant (A0, B0)
      read_a(0, cur); read_b(0, cur);
      if (more) issue_a(0, kt + 1, nxt);
      phase_barrier(!lead, !more);
      quadrant(0, 0);
      phase_barrier(lead, !more);
      // phase 2: (A0, B1)
      read_b(1, cur);
      if (more) issue_b(0, kt + 1, nxt);
      phase_barrier(!lead, !more);
      quadrant(0, 1);
      phase_barrier(lead, !more);
      // phase 3: (A1, B1)
      read_a(1, cur);
      if (more) issue_b(1, kt + 1, nxt);
      phase_barrier(false, false);
      quadrant(1, 1);
      phase_barrier(false, false);
      // phase 4: (A1, B0) -- nothing to read
      if (more) issue_a(1, kt + 1, nxt);
      phase_barrier(!lead, false);
      quadrant(1, 0);
      phase_barrier(lead, false);
    }
    if (lead) wait_vm_barrier<63>();             // balance the trailing group's extra barrier
    __syncthreads();                             // both stages are dead
    // Next tile's k-tile 0 streams in UNDER this tile's epilogue: it goes to the stage that served k-tile nk-2; the accumulators
    // are staged in the other one (k-tile nk-1's), so the two never meet.  The epilogue's stores drain while the next main
    // loop runs out of L2.
    base = (base + nk) & 1;
    const int em0 = m0, en0 = n0;
    const int next = tile + tile_step;
    const bool has_next = next < tile_end;       // workgroup-uniform
    if (has_next) {
      setup(next);
      half_t* st0 = lds + base * C::STAGE;
      issue_a(0, 0, st0); issue_b(0, 0, st0); issue_b(1, 0, st0); issue_a(1, 0, st0);
    }
    if constexpr (M32) {
      // one tile per workgroup only (launch_gemm8): padded staging from the start of the ring
      float* ct = reinterpret_cast<float*>(smem) + wave * (C::EP_MT * 16 * C::CT_LD);
#pragma unroll
      for (int p0 = 0; p0 < 8; p0 += C::EP_MT) {
        wave_lds_fence();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(ct + (lane & 31) * C::CT_LD + nt * 32 + 8 * g + 4 * (lane >> 5)) =
                f32x4{acc32[p0 / 2][nt][4 * g], acc32[p0 / 2][nt][4 * g + 1], acc32[p0 / 2][nt][4 * g + 2], acc32[p0 / 2][nt][4 * g + 3]};
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < C::EP_MT * 2; ++it) {
          const int r = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
          const f32x4 a = *reinterpret_cast<const f32x4*>(ct + r * C::CT_LD + c8);
          const f32x4 b = *reinterpret_cast<const f32x4*>(ct + r * C::CT_LD + c8 + 4);
          const int m = em0 + (wr * 8 + p0) * 16 + r;
          if (m < M) epi.apply8(m, en0 + wc * 64 + c8, a, b);
        }
      }
    } else {
      // 8 KiB per wave inside the dead stage: 32 rows x 16 chunks of 16 bytes, chunk c of row r at c ^ (r & 15) (the
      // 16 lanes of one accumulator column group hit 16 different chunks; a row's 8 readers too)
      float* ct = reinterpret_cast<float*>(lds + (base ^ 1) * C::STAGE) + wave * (C::EP_MT * 16 * 64);
#pragma unroll
      for (int p0 = 0; p0 < 8; p0 += C::EP_MT) {
        wave_lds_fence();
#pragma unroll
        for (int mt = 0; mt < C::EP_MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int r = mt * 16 + (lane & 15);
            *reinterpret_cast<f32x4*>(ct + r * 64 + (((nt * 4 + (lane >> 4)) ^ (r & 15)) << 2)) = acc[p0 + mt][nt];
          }
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < C::EP_MT * 2; ++it) {
          const int r = it * 8 + (lane >> 3), j = (lane & 7) * 2;
          const f32x4 a = *reinterpret_cast<const f32x4*>(ct + r * 64 + ((j ^ (r & 15)) << 2));
          const f32x4 b = *reinterpret_cast<const f32x4*>(ct + r * 64 + (((j + 1) ^ (r & 15)) << 2));
          const int m = em0 + (wr * 8 + p0) * 16 + r;
          if (m < M) epi.apply8(m, en0 + wc * 64 + j * 4, a, b);
        }
      }
    }
    if (!has_next) break;
    tile = next;
  }
}

#endif  // APH_EXPERIMENTS

// ---- epilogues: apply8 gets 8 consecutive columns n..n+7 of row m -------------------------------------------
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
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    if (bias) { a += ld4(bias + n); b += ld4(bias + n + 4); }
    store_h8(out + (size_t)m * ldo + n, a, b);
  }
};

struct EpiF32 {          // out = acc * scale  (fp32)
  float* out; int ldo; float scale;
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    st4(out + (size_t)m * ldo + n, a * scale);
    st4(out + (size_t)m * ldo + n + 4, b * scale);
  }
};

struct EpiNoStore {       // measurement hook: keeps the accumulators live, stores only on a value that never occurs
  float* out; int ldo;
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    if (a[0] == 1.2345678e33f) { st4(out + (size_t)m * ldo + n, a); st4(out + (size_t)m * ldo + n + 4, b); }
  }
};

struct EpiF16Scale {     // out = acc * scale (f16)
  half_t* out; int ldo; float scale;
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const { store_h8(out + (size_t)m * ldo + n, a * scale, b * scale); }
};

struct EpiResidual {     // out = res + acc + bias   (fp32 residual stream)
  float* out; const float* res; int ldo; const float* bias;
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    const size_t o = (size_t)m * ldo + n;
    st4(out + o, ld4(res + o) + a + ld4(bias + n));
    st4(out + o + 4, ld4(res + o + 4) + b + ld4(bias + n + 4));
  }
};

// QuickGELU g = u sigmoid(1.702 u) with u = acc + bias.  Besides g the forward stores the derivative
// dg/du = s (1 + 1.702 u (1 - s)) (same sigmoid s, f16), so the backward's epilogue is one multiply.
__device__ __forceinline__ void quick_gelu4(const f32x4& u, f32x4& g, f32x4& dg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = fast_rcp(1.0f + __expf(-1.702f * u[i]));      // v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the outputs are f16
    g[i] = u[i] * s;
    dg[i] = s + 1.702f * (g[i] - g[i] * s);                        // = s (1 + 1.702 u (1 - s))
  }
}
struct EpiGelu {
  half_t* dg; half_t* g; int ldo; const float* bias;
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
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    const half8 d = *reinterpret_cast<const half8*>(dg + (size_t)m * ldo + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] *= (float)d[i]; b[i] *= (float)d[4 + i]; }
    store_h8(out + (size_t)m * ldo + n, a, b);
  }
};

struct EpiPatchEmbed {   // token row s*T + 1 + p  <-  patch row s*P + p ;  + positional embedding
  float* x0; const float* pos; int D, P, T;
  __device__ __forceinline__ void apply8(int m, int n, f32x4 a, f32x4 b) const {
    const int s = m / P, p = m - s * P;
    const float* pe = pos + (size_t)(1 + p) * D + n;
    float* o = x0 + ((size_t)s * T + 1 + p) * D + n;
    st4(o, a + ld4(pe));
    st4(o + 4, b + ld4(pe + 4));
  }
};

#ifndef APH_GEMM_WS_MIN_TILES_DEFAULT
#define APH_GEMM_WS_MIN_TILES_DEFAULT 160
#endif
// MFMA shape of the ring kernels' main loops: 0 = v_mfma_f32_16x16x32_f16 (what the product runs: measured faster on MI355X with real
// operands), 1 = v_mfma_f32_32x32x16_f16 (aph_gemm_set_mfma32(): kept for the layout tests and A/B measurements).
inline int& gemm_mfma32() {
  static int v = 0;
  return v;
}

template <class C, class Epi>
inline void launch_gemm_cfg(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  const dim3 grid((N / C::BN) * ((M + C::BM - 1) / C::BM));
  // the 32x32x16 variant (a measured alternative, not what the product runs) is instantiated for the GEMM test entry's fp32 epilogue only
  constexpr bool M32OK = C::TM % 2 == 0 && C::TN % 2 == 0 && C::EP_MT == C::TM && std::is_same<Epi, EpiF32>::value;
  if constexpr (M32OK) {
    if (gemm_mfma32()) {
      APH_ALLOW_SMEM((gemm_f16_kernel<C, Epi, false, true>), C::SMEM);
      APH_LAUNCH((gemm_f16_kernel<C, Epi, false, true>), grid, dim3(C::NTHREAD), C::SMEM, st, A, lda, Bt, ldb, M, N, K, epi, SplitK{nullptr, 1});
      return;
    }
  }
  {
    APH_ALLOW_SMEM((gemm_f16_kernel<C, Epi, false, false>), C::SMEM);
    APH_LAUNCH((gemm_f16_kernel<C, Epi, false, false>), grid, dim3(C::NTHREAD), C::SMEM, st, A, lda, Bt, ldb, M, N, K, epi, SplitK{nullptr, 1});
  }
}

// split-K workspace of one caller (the ViT handle owns one; the partials of a launch are consumed by the reduce kernel
// that follows it on the same stream)
struct SplitKSpace {
  float* ws = nullptr;
  size_t ws_floats = 0;
  // the WHOLE batch of the ViT call these launches belong to is at most 128 token rows (one or two cuts): written by aph_vit_forward /
  // aph_vit_backward into their own handle's space, read by launch_gemm -- per call, not process-wide (two engines on two host threads
  // no longer race on it).  Stand-alone GEMM calls (the test entries, no space) count as small batches.
  bool small_batch = true;
};

// how many ways to split K for the 64x64 configuration: only when the tiles alone occupy a small part of the chip (the
// class-row GEMMs: M = cuts) and the k loop is long; every share keeps >= 6 k-tiles.  At M ~ 1200 (228 tiles) the
// 64x64 tiles are bound by L2-miss bandwidth on the weight panel, and splitting K was measured to lose (18 -> 21 us).
// (Round 4, profiles/r04_small_m_deep_ring.txt: an EIGHT-stage ring for the 64x64 configuration -- seven k-tiles in flight per CU, one workgroup
// per CU -- was the queued experiment for M = 1200: slower on every ViT shape, 19.8 vs 13.5 us on QKV, 19.7 vs 17.9 on fc2: these k loops are
// not short of requests in flight; two co-resident 4-stage workgroups hide more than one deep ring.  Removed.)
inline int choose_splits(int M, int N, int K, const SplitKSpace* sp) {
  if (!sp || !sp->ws) return 1;
  const int tiles = (N / GemmSmall::BN) * ((M + GemmSmall::BM - 1) / GemmSmall::BM), nk = K / GEMM_BK;
  int splits = 1;
  while (splits < 4 && tiles * (splits * 2) <= 256 && nk / (splits * 2) >= 6) splits *= 2;
  if ((size_t)splits * M * N > sp->ws_floats) return 1;
  return splits;
}

template <class C, class Epi>
inline void launch_gemm_splitk(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, int splits,
                               const SplitKSpace& sp, hipStream_t st) {
  const int tiles = (N / C::BN) * ((M + C::BM - 1) / C::BM);
  if constexpr (std::is_same<Epi, EpiF32>::value) {
    if (gemm_mfma32()) {
      APH_ALLOW_SMEM((gemm_f16_kernel<C, Epi, true, true>), C::SMEM);
      APH_LAUNCH((gemm_f16_kernel<C, Epi, true, true>), dim3(tiles * splits), dim3(C::NTHREAD), C::SMEM, st, A, lda, Bt, ldb, M, N, K, epi,
                 SplitK{sp.ws, splits});
      const size_t work0 = (size_t)M * (N / 8);
      APH_LAUNCH((splitk_reduce_kernel<Epi>), dim3((unsigned)((work0 + 255) / 256)), dim3(256), 0, st, (const float*)sp.ws, splits, M, N, epi);
      return;
    }
  }
  {
    APH_ALLOW_SMEM((gemm_f16_kernel<C, Epi, true, false>), C::SMEM);
    APH_LAUNCH((gemm_f16_kernel<C, Epi, true, false>), dim3(tiles * splits), dim3(C::NTHREAD), C::SMEM, st, A, lda, Bt, ldb, M, N, K, epi,
               SplitK{sp.ws, splits});
  }
  const size_t work = (size_t)M * (N / 8);
  APH_LAUNCH((splitk_reduce_kernel<Epi>), dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, (const float*)sp.ws, splits, M, N, epi);
}

// workgroups of a persistent launch: one per CU of the current device
inline int gemm_persistent_wgs() {
#ifdef APH_EMU
  return 3;                                        // exercises the tile loop (and its remainders) under the interpreter
#else
  thread_local int dev_cached = -1, ncu = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1 << 30;
  if (dev != dev_cached) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) return 1 << 30;
    dev_cached = dev;
  }
  return ncu;
#endif
}

// the phased kernel addresses its operands with 32-bit byte offsets from the matrix base
inline bool gemm8_addressable(int M, int lda, int N, int ldb) {
  return (size_t)M * lda * 2 < ((size_t)1 << 32) && (size_t)N * ldb * 2 < ((size_t)1 << 32);
}

#ifdef APH_EXPERIMENTS
template <class Epi>
inline void launch_gemm8(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  const int ntiles = (N / Gemm8::BN) * ((M + Gemm8::BM - 1) / Gemm8::BM);
  if (gemm_mfma32()) {
    APH_ALLOW_SMEM((gemm8_f16_kernel<Epi, true>), Gemm8::SMEM);
    APH_LAUNCH((gemm8_f16_kernel<Epi, true>), dim3(ntiles), dim3(Gemm8::NTHREAD), Gemm8::SMEM, st, A, lda, Bt, ldb, M, N, K, epi, ntiles);
  } else {
    const int wgs = gemm_persistent_wgs();
    APH_ALLOW_SMEM((gemm8_f16_kernel<Epi, false>), Gemm8::SMEM);
    APH_LAUNCH((gemm8_f16_kernel<Epi, false>), dim3(ntiles < wgs ? ntiles : wgs), dim3(Gemm8::NTHREAD), Gemm8::SMEM, st, A, lda, Bt, ldb, M, N, K,
               epi, ntiles);
  }
}

#endif  // APH_EXPERIMENTS

// tile choice, from the measured sweep over the ViT-B shapes at 1/2/4/8-rank shard sizes (tools/exp/tune_table.py):
//   256x256 phased   wide outputs with >= 400 such tiles (N = 3072 at full batch: 456; QKV's 342 tiles take the 256x128 path --
//                    forcing every GEMM through 256x128 instead measured 136.1 vs 139.0 steps/s, profiles/r02_ab_mfma32.txt)
//   256x128          >= 160 tiles
//   128x128, 8 waves, 4-stage ring   >= 160 such tiles (half-batch shards with N = 768, wide outputs of small shards)
//   64x64 (2 workgroups per CU)      everything smaller; split-K when only a handful of tiles exist
template <class Epi>
inline void launch_gemm_ws(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                           unsigned long long* trace = nullptr, int n_short = 0, int k_short = 0);   // vit_gemm_ws.h
// the wave-specialised persistent kernel takes every shape with at least this many 256x128 tiles (aph_gemm_set_ws_min_tiles(); 0 = never)
inline int& gemm_ws_min_tiles() {
  static int v = APH_GEMM_WS_MIN_TILES_DEFAULT;
  return v;
}

// The register-staged kernels of vit_gemm_rs.h.  gemm_rs_mode(): 1 (default) = the split-K kernel for GEMMs of at most 128 rows over
// K <= 1024 WHEN THE WHOLE BATCH IS THAT SMALL (SplitKSpace::small_batch, set by the ViT entry points from cuts x tokens: one or two cuts,
// C1): one launch with an ordered in-kernel reduction instead of a split-K launch plus its reduce launch (C1 577 -> 726 steps/s).  Over
// K = 3072 the two-pass split-K stays: its 48 workgroups pull the cold weight matrix through four times as many CUs (11.3 against 20 us,
// profiles/r05_kernel_stats_s26_fused_v4.csv).  The class-row GEMMs of a LARGER batch's last block (M = cuts) stay on the two-pass
// kernels as well: 3-4 us each faster on this kernel at 24 cuts, but the stress-weight loss curve is a chaotic amplifier of rounding ORDER
// (DESIGN.md section 4 *Precision*): three summation orders of those few GEMMs gave 2.6e-4 / 7.6e-4 / 1.04e-3 in the split-precision mode
// (profiles/r05_stress_rs_ab.txt, r05_stress_rs_dense.txt; against fp64 this kernel is the more accurate of the two), and the order with
// the margin under north_star's 1e-3 is worth more than 0.7 % of a shard's step;
// 2 = every shape the register-staged kernels address (the A/B switch of bench.py --vit-path rs and the test hook); 0 = off.
inline int& gemm_rs_mode() {
  static int v = 1;
  return v;
}

// does launch_gemm hand this shape to the wave-specialised persistent kernel?
inline bool gemm_takes_ws(int M, int lda, int N, int ldb) {
  const int big_tiles = (N / GemmBig::BN) * ((M + GemmBig::BM - 1) / GemmBig::BM);
  return gemm_ws_min_tiles() > 0 && big_tiles >= gemm_ws_min_tiles() && N <= 4096 /* GemmWS::BIAS_MAX */ && gemm8_addressable(M, lda, N, ldb) && !gemm_mfma32();
}
template <class Epi>
inline void launch_gemm(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                        const SplitKSpace* sp = nullptr) {
  const int big_tiles = (N / GemmBig::BN) * ((M + GemmBig::BM - 1) / GemmBig::BM);
  if (gemm_takes_ws(M, lda, N, ldb)) {
    launch_gemm_ws(A, lda, Bt, ldb, M, N, K, epi, st);
    return;
  }
  if (gemm_rs_mode() && (gemm_rs_mode() > 1 || ((sp ? sp->small_batch : true) && M <= 128 && K <= 1024)) && gemm8_addressable(M, lda, N, ldb) && !gemm_mfma32() &&
      launch_gemm_rs_auto(A, lda, Bt, ldb, M, N, K, epi, st, gemm_rs_mode() > 1))
    return;
  const int mid_tiles = (N / GemmMidDeep8::BN) * ((M + GemmMidDeep8::BM - 1) / GemmMidDeep8::BM);
  if (big_tiles >= 160) launch_gemm_cfg<GemmBig>(A, lda, Bt, ldb, M, N, K, epi, st);
  else if (mid_tiles >= 160) launch_gemm_cfg<GemmMidDeep8>(A, lda, Bt, ldb, M, N, K, epi, st);
  else {
    const int splits = choose_splits(M, N, K, sp);
    if (splits > 1) launch_gemm_splitk<GemmSmall>(A, lda, Bt, ldb, M, N, K, epi, splits, *sp, st);
    else launch_gemm_cfg<GemmSmall>(A, lda, Bt, ldb, M, N, K, epi, st);
  }
}

}  // namespace aph
