// Flag-synchronised variant of the wave-specialised persistent GEMM (round 4), gfx950.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        same contract, tiles, ring, LDS image and epilogues as vit_gemm_ws.h
//
// What round 3's per-tile trace left: the main loop of gemm_ws_kernel runs at 1320 shader clocks per k-tile
// against the 1024 its MFMAs need (two consumer waves per SIMD x 32 MFMAs x 16 clocks) -- one ten-wave s_barrier per k-tile: the eight
// consumers arrive together, the matrix pipes drain while the barrier resolves, and because the barrier keeps every wave in the same
// k-tile, all eight reach their epilogue (conversion, bias, GELU, stores: 1.5-2.8 us with the matrix pipe idle) together as well.
// Here NO workgroup barrier is executed after the prologue.  Producers and consumers hand k-tile units over through counters in LDS:
//   ready[p]  (producer wave p):  number of units whose DMA share of wave p has landed   (written after a counted vmcnt wait)
//   done[c]   (consumer wave c):  number of units whose fragments wave c has finished reading (written after lgkmcnt(0))
// A consumer reads unit u + 1 once min(ready) >= u + 2; it asks for the counters together with its last fragment reads of unit u, so
// the answer is normally there when the MFMAs of unit u have been issued (no stall).  A producer refills the stage of unit u with unit
// u + NSTAGE once min(done) >= u + 1, polling with s_sleep.  Waves therefore drift against each other by up to the ring depth: the two
// consumer waves of a SIMD fall out of phase (one in its MFMAs while the other waits for LDS), and a wave in its epilogue leaves the
// matrix pipe to its SIMD partner, which is up to NSTAGE - 1 units further on.
// LDS ordering assumptions (the same the barrier kernels rest on): a wave's DS instructions execute in order; the LDS write of a
// global_load_lds is complete when vmcnt has counted it.  The hand-over counters are only ever written by one wave each.
#pragma once
#include "vit_gemm_ws.h"

namespace aph {

__device__ __forceinline__ void wsf_pause() {
#ifdef APH_EMU
  emu::yield();
#else
  __builtin_amdgcn_s_sleep(1);
#endif
}
template <int N>
__device__ __forceinline__ void wsf_wait_vm() {           // at most N vector-memory operations of this wave outstanding
#ifndef APH_EMU
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wsf_compiler_fence() {
#ifndef APH_EMU
  asm volatile("" ::: "memory");
#endif
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// counters in LDS: volatile dword accesses (ds_read_b32 / ds_write_b32), never cached in registers
__device__ __forceinline__ unsigned wsf_ld(const unsigned* p) { return *reinterpret_cast<const volatile unsigned*>(p); }
__device__ __forceinline__ void wsf_st(unsigned* p, unsigned v) { *reinterpret_cast<volatile unsigned*>(p) = v; }

template <int AHEAD_UNIT>
__device__ __forceinline__ void wsf_wait_landed(int ahead) {   // all but the newest `ahead` units of this producer wave's DMA have landed
  if (ahead <= 0) wsf_wait_vm<0>();
  else if (ahead == 1) wsf_wait_vm<AHEAD_UNIT>();
  else wsf_wait_vm<(2 * AHEAD_UNIT <= 63 ? 2 * AHEAD_UNIT : 63)>();
}

template <class C, class Epi>
__global__ __launch_bounds__(C::NTHREAD, C::MIN_WAVES_PER_SIMD) void gemm_wsf_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt,
                                                                   int ldb, int M, int N, int K, Epi epi, int ntiles) {
  using F = GemmBig;
  static_assert(C::NPROD == 2 && C::NCONS == 8, "counter layout below: two producers, eight consumers");
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  int tile, tile_end, tile_step;
  {
    const int nwg = gridDim.x, b = blockIdx.x, G = nwg < 8 ? nwg : 8;
    const int q = ntiles / G, r = ntiles - q * G, xcd = b % G;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile_end = start + q + (xcd < r ? 1 : 0);
    tile_step = (nwg - xcd + G - 1) / G;
    tile = start + b / G;
  }
  if (tile >= tile_end) return;
  const int nk = K / GEMM_BK, ntn = N / C::BN;
  const int U = ((tile_end - tile + tile_step - 1) / tile_step) * nk;
  float* lbias = reinterpret_cast<float*>(smem + C::SMEM);
  unsigned* flags = reinterpret_cast<unsigned*>(smem + C::SMEM_TOTAL);       // [0..1] ready, [4..11] done (16-byte aligned groups)
  unsigned* ready = flags;
  unsigned* done = flags + 4;
  if (const float* gb = ws_bias(epi)) {
    for (int i = tid * 4; i < N; i += C::NTHREAD * 4) *reinterpret_cast<f32x4*>(lbias + i) = ld4(gb + i);
  }
  if (tid < 16) wsf_st(flags + tid, 0u);
  __syncthreads();                               // the only workgroup barrier: bias vector and zeroed counters are visible

  if (wave >= C::NCONS) {
    // ---------------------------------------------------------------- producer
    const int p = wave - C::NCONS, lrow = lane >> 3, pc = lane & 7;
    unsigned off[C::QPW];
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(Bt);
    auto setup = [&](int t) {
      const int tm = t / ntn, n0 = (t - tm * ntn) * C::BN, m0 = tm * C::BM;
#pragma unroll
      for (int i = 0; i < C::QPW; ++i) {
        if (i < C::QAW) {
          const int row = (i * C::NPROD + p) * 8 + lrow;
          int am = m0 + row; am = am < M ? am : M - 1;
          off[i] = ((unsigned)am * (unsigned)lda + ((pc ^ ((row >> 1) & 7)) << 3)) * 2u;
        } else {
          const int v = ((i - C::QAW) * C::NPROD + p) * 8 + lrow;
          off[i] = ((unsigned)(n0 + gemm_ws_brow(v)) * (unsigned)ldb + ((pc ^ ((v >> 1) & 7)) << 3)) * 2u;
        }
      }
    };
    auto issue = [&](int kt, int stage) {
      half_t* As = lds + stage * C::STAGE;
      half_t* Bs = As + C::BM * GEMM_BK;
      const char* ak = Ab + (size_t)kt * (GEMM_BK * 2);
      const char* bk = Bb + (size_t)kt * (GEMM_BK * 2);
#pragma unroll
      for (int i = 0; i < C::QPW; ++i) {
        if (i < C::QAW) glds16(ak + off[i], As + (i * C::NPROD + p) * 8 * GEMM_BK);
        else glds16(bk + off[i], Bs + ((i - C::QAW) * C::NPROD + p) * 8 * GEMM_BK);
      }
    };
    int it = tile, ikt = 0, ist = 0, issued = 0;
    setup(it);
    auto issue_next = [&]() {
      issue(ikt, ist);
      ist = ist == C::NSTAGE - 1 ? 0 : ist + 1;
      ++issued;
      if (++ikt == nk) {
        ikt = 0;
        it += tile_step;
        if (issued < U) setup(it);
      }
    };
    auto publish = [&](int landed) {              // this wave's share of units 0 .. landed - 1 is in LDS
      wave_lds_fence();
      if (lane == 0) wsf_st(ready + p, (unsigned)landed);
      wsf_compiler_fence();
    };
    auto wait_done = [&](int need) {              // every consumer has finished reading units 0 .. need - 1
      for (;;) {
        const u32x4 a = *reinterpret_cast<const volatile u32x4*>(done), b = *reinterpret_cast<const volatile u32x4*>(done + 4);
        unsigned m = a.x < a.y ? a.x : a.y, m2 = a.z < a.w ? a.z : a.w, m3 = b.x < b.y ? b.x : b.y, m4 = b.z < b.w ? b.z : b.w;
        m = m < m2 ? m : m2; m3 = m3 < m4 ? m3 : m4; m = m < m3 ? m : m3;
        if (wave_uniform((int)m) >= need) break;
        wsf_pause();
      }
      wsf_compiler_fence();
    };
    const int pre = U < C::NSTAGE ? U : C::NSTAGE;                 // every stage is free at the start
    for (int i = 0; i < pre; ++i) issue_next();
    wsf_wait_landed<C::QPW>(pre - 1);                              // unit 0 has landed
    publish(1);
    for (int u = 0; u < U; ++u) {
      const int target = u + 1 < U ? u + 1 : U - 1;                // highest unit that must have landed now
      if (target > 0 || u > 0) {
        wsf_wait_landed<C::QPW>(issued - 1 - target);
        publish(target + 1);
      }
      if (u + C::NSTAGE < U) {
        wait_done(u + 1);                                          // the stage of unit u is free
        issue_next();                                              // unit u + NSTAGE
      }
    }
    return;
  }

  // ------------------------------------------------------------------ consumer
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  f32x4 acc[F::TM][F::TN];
  const int frow = lane & 15, arow = wm * 64 + frow, brow = wn * 64 + frow, fchunk = lane >> 4;
  GemmFrags<F> f0, f1;
  int kt = 0, st = 0;
  auto ready_min = [&]() -> int {
    const u32x2 r = *reinterpret_cast<const volatile u32x2*>(ready);
    return (int)(r.x < r.y ? r.x : r.y);
  };
  auto wait_ready = [&](int need, int have) {     // both producers' shares of units 0 .. need - 1 have landed
    while (wave_uniform(have) < need) {
      wsf_pause();
      have = ready_min();
    }
    wsf_compiler_fence();
  };
  auto init_tile = [&](int t) {
    const int tm = t / ntn;
    const int m4 = tm * C::BM + wm * 64 + 4 * (lane >> 4), n4 = (t - tm * ntn) * C::BN + wn * 64 + 4 * (lane & 15);
#pragma unroll
    for (int mt = 0; mt < F::TM; ++mt) ws_init(epi, m4 + mt * 16, M, n4, acc[mt], lbias);
  };
  auto mma = [&](const GemmFrags<F>& f) {
#pragma unroll
    for (int mt = 0; mt < F::TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < F::TN; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.a[mt], f.b[nt], acc[mt][nt]);
  };
  init_tile(tile);
  wait_ready(1, ready_min());                                                    // unit 0 has landed
  gemm_load_frags<F>(f0, lds, lds + C::BM * GEMM_BK, arow, brow, fchunk);
  for (int u = 0; u < U; ++u) {
    const half_t* As = lds + st * C::STAGE;
    st = st == C::NSTAGE - 1 ? 0 : st + 1;
    gemm_load_frags<F>(f1, As, As + C::BM * GEMM_BK, arow, brow, 4 + fchunk);    // k-step 1 of unit u: in flight during the MFMAs
    int have = 0;
    if (u + 1 < U) have = ready_min();                                           // asked for behind the fragment reads, looked at after the MFMAs
    mma(f0);                                                                     // k-step 0 of unit u
    if (u + 1 < U) {
      wait_lgkm0();                                                              // f1 (and the counters) have left LDS: this wave is through with unit u's stage
      wave_lds_fence();
      if (lane == 0) wsf_st(done + wave, (unsigned)(u + 1));
      wait_ready(u + 2, have);                                                   // unit u + 1 has landed (normally known already)
      const half_t* An = lds + st * C::STAGE;
      gemm_load_frags<F>(f0, An, An + C::BM * GEMM_BK, arow, brow, fchunk);      // k-step 0 of unit u + 1: overlaps the MFMAs below
    }
    mma(f1);                                                                     // k-step 1 of unit u
    if (++kt == nk) {
      const int tm = tile / ntn;
      const int m4 = tm * C::BM + wm * 64 + 4 * (lane >> 4), n4 = (tile - tm * ntn) * C::BN + wn * 64 + 4 * (lane & 15);
      ws_tiles(epi, m4, M, n4, acc, lbias);
      kt = 0;
      tile += tile_step;
      if (u + 1 < U) init_tile(tile);
    }
  }
}

template <class C, class Epi>
inline void launch_gemm_wsf_cfg(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  const int ntiles = (N / C::BN) * ((M + C::BM - 1) / C::BM);
  const int cus = gemm8_persistent_wgs();
  const int wgs = cus > (1 << 20) ? cus : cus * C::WG_PER_CU;
  constexpr int smem = C::SMEM_TOTAL + 64;                 // + the hand-over counters
  static_assert(smem <= 160 * 1024, "ring + bias + counters must fit the CU's LDS");
  APH_ALLOW_SMEM((gemm_wsf_kernel<C, Epi>), smem);
  APH_LAUNCH((gemm_wsf_kernel<C, Epi>), dim3(ntiles < wgs ? ntiles : wgs), dim3(C::NTHREAD), smem, st, A, lda, Bt, ldb, M, N, K, epi, ntiles);
}

// 256 x 128 tiles, 8 consumers + 2 producers, 3 x 48 KiB ring, 12 KiB of bias (N <= 3072: every ViT-B linear) + 64 B of counters
using GemmWSF = GemmWSCfg<256, 2, 3, 1, 3072>;
template <class Epi>
inline void launch_gemm_wsf(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st) {
  launch_gemm_wsf_cfg<GemmWSF>(A, lda, Bt, ldb, M, N, K, epi, st);
}

}  // namespace aph
