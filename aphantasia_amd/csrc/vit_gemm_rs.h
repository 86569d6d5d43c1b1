// Small-M fp16 GEMMs for the CLIP ViT linears (round 5), gfx950: operands staged through REGISTERS, no barrier in the main loop.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        same contract as vit_gemm.h
//
// What the ring kernels of vit_gemm.h do at M ~ 1200 rows (a 24-cut shard: one of eight ranks; also C1 / C3 / C5): 228 tiles of 64 x 64 on
// 256 CUs is ONE 4-wave workgroup per CU, and (profiles/r05_load_rate.txt, tools/exp/load_rate.hip) four waves cannot feed a CU through
// global_load_lds_dwordx4: 70 GB/s per CU from L2-resident data whatever the number in flight (8 waves: 126, the 64 B/clk limit is 134),
// and their k loop is a chain of counted wait -> s_barrier -> fragment reads -> MFMAs with nothing else resident to fill the waits:
// 50-60 GB/s per CU in the event, fc2 17.9 us / dqkv 14.4 us for 5.7 / 4.2 GFLOP.  The same four waves reach 110-125 GB/s with plain
// global_load_dwordx4 into registers -- provided every QUAD of lanes reads 64 contiguous bytes (the MFMA fragment pattern straight from a
// row-major matrix, 16 rows x 16 bytes per quarter wave, is stuck at 37 GB/s at any occupancy).  So here:
//   * loads are global_load_dwordx4 of [16 rows][64 bytes] blocks (lane >> 2 = row, lane & 3 = 16-byte chunk), PD k-steps of 32 in flight
//     per wave in registers (the register file is the prefetch buffer: one wave per SIMD owns 512 VGPRs);
//   * a wave writes a landed k-step into its PRIVATE LDS image (ds_write_b128, chunk p of row r at piece p ^ ((-(r >> 2)) & 3): the four
//     lane groups of a ds_read_b128 fragment fetch then hit 16 different 16-byte bank groups) and reads its MFMA fragments back: LDS is
//     the transposer, and since only the owning wave touches an image, program order is all the synchronisation there is -- NO s_barrier,
//     NO hand-counted vmcnt (the compiler's own counted waits cover the register loads); the four waves drift freely;
//   * operands are swapped at the MFMA (weights as A fragment) and the weight rows of a 16-row tile are permuted through the load address so
//     that lane l ends up with 4 NT CONSECUTIVE columns of token row l & 15: the epilogues are vit_gemm.h's apply8.
// Two kernels:
//   gemm_sk_kernel  64 x 64 tile per workgroup, the four waves SPLIT K (wave w owns the k-tiles w, w + 4, ...: whole 128-byte lines of every
//                   operand row) and each accumulates the full tile; the four fp32 partials meet in LDS at the end and are summed in the
//                   fixed order 0, 1, 2, 3 (bitwise reproducible).  Every operand byte is fetched once per workgroup.  For the long-K,
//                   narrow-N shapes (out-proj, fc2, dfc1, dqkv, patch embedding).
//   gemm_ar_kernel  64 rows x 64 NT columns per workgroup, the A block (64 x K, K <= 1024) RESIDENT in LDS as K / 32 k-step images, filled
//                   once; wave w streams the weight rows of its own 16 NT columns.  Tiles run column-group-major so that an XCD's L2 holds
//                   the weight rows its workgroups share.  For the wide-N, K = width shapes (QKV, fc1, dfc2, patch-embedding dgrad); the
//                   fused block kernels of vit_block.h are this kernel with other prologues / epilogues.
#pragma once
#include "vit_gemm.h"

namespace aph {

// k-step image: [rows][64 bytes]; 16-byte piece p of row r holds chunk p ^ rs_swz(r)
__device__ __forceinline__ int rs_swz(int row) { return (0 - (row >> 2)) & 3; }

struct RSLane {
  int lrow, lpc;      // load role: row within a 16-row block, 16-byte chunk of the 64-byte k-step row (a quad of lanes = 64 contiguous bytes)
  int wpos;           // byte position of that piece within a 16-row (1 KiB) block of a k-step image
  int fpos;           // fragment read role: row lane & 15, chunk lane >> 4 -> byte position within a 16-row block
  __device__ __forceinline__ explicit RSLane(int lane) {
    lrow = lane >> 2; lpc = lane & 3;
    wpos = lrow * 64 + ((lpc ^ rs_swz(lrow)) << 4);
    const int frow = lane & 15;
    fpos = frow * 64 + (((lane >> 4) ^ rs_swz(frow)) << 4);
  }
};

__device__ __forceinline__ half8 ldg8(const char* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ void sts8(char* p, const half8& v) { *reinterpret_cast<half8*>(p) = v; }
__device__ __forceinline__ half8 lds8(const char* p) { return *reinterpret_cast<const half8*>(p); }

// XCD-aware tile run (as gemm_f16_kernel): workgroup b runs on XCD b % 8 (observed dispatch, speed only); every XCD gets one contiguous
// run of tile indices.  Bijective for any grid size.
__device__ __forceinline__ int rs_tile_index() {
  const int nwg = gridDim.x, b = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// measurement hook (aph_gemm_rs_probe): chip-wide 100 MHz clock stamps of wave 0 of every workgroup, slot k of trace[wg][8]
__device__ __forceinline__ void rs_stamp(unsigned long long* trace, int k) {
#ifndef APH_EMU
  if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- split-K kernel -----------------------------------------------------------------------------------------------------------------
struct GemmSK {
  static constexpr int BM = 64, BN = 64, NTHREAD = 256;
  static constexpr int IMG = (BM + BN) * 64;                 // bytes of one k-step image (both operands)
  static constexpr int SMEM = 4 * BM * BN * 4;               // four fp32 partial tiles (>= 4 waves x 2 images)
};

// NS = k-steps per wave = K / 128 (K % 256 == 0): a COMPILE-TIME count -- the loop is straight-line code, so every register load is
// unconditional and the compiler's in-order vmcnt bookkeeping is exact (with the loads behind run-time conditions it has to assume they
// were not issued, and every wait drains the whole prefetch queue: found in the ISA of the first version).
template <int NS, int PD, class Epi>
__global__ __launch_bounds__(256) void gemm_sk_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt, int ldb, int M, int N,
                                                      Epi epi, unsigned long long* __restrict__ trace) {
  using C = GemmSK;
  APH_DYN_SMEM(smem);
  rs_stamp(trace, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  int m0, n0;
  {
    const int tile = rs_tile_index(), ntn = N / C::BN, tm = tile / ntn;      // n-tiles fastest: the workgroups of an XCD share A panels
    n0 = (tile - tm * ntn) * C::BN;
    m0 = tm * C::BM;
  }
  char* ring = smem + wave * (2 * C::IMG);
  const RSLane L(lane);
  unsigned offT[4], offW[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int am = m0 + 16 * q + L.lrow;
    am = am < M ? am : M - 1;
    offT[q] = ((unsigned)am * (unsigned)lda + L.lpc * 8) * 2u;
    // tile q, tile row i = lrow is weight row 16 (i >> 2) + 4 q + (i & 3) of the workgroup's 64
    offW[q] = ((unsigned)(n0 + 16 * (L.lrow >> 2) + 4 * q + (L.lrow & 3)) * (unsigned)ldb + L.lpc * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bt);
  auto load = [&](int s, half8 (&r)[8]) {                           // k-step s of this wave: half s & 1 of its k-tile wave + 4 (s >> 1)
    const size_t kb = (size_t)(wave + 4 * (s >> 1)) * 128 + (s & 1) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r[q] = ldg8(Ab + offT[q] + kb); r[4 + q] = ldg8(Bb + offW[q] + kb); }
  };
  auto stage = [&](char* img, const half8 (&r)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) sts8(img + q * 1024 + L.wpos, r[q]);
  };
  struct Frags { half8 t[4], w[4]; };
  auto frags = [&](Frags& f, const char* img) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { f.t[q] = lds8(img + q * 1024 + L.fpos); f.w[q] = lds8(img + 4096 + q * 1024 + L.fpos); }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.w[nt], f.t[mt], acc[mt][nt]);
  };
  {
    half8 R[PD][8];
    Frags f[2];
#pragma unroll
    for (int d = 0; d < PD; ++d)
      if (d < NS) load(d, R[d]);
    stage(ring, R[0]);
    if (PD < NS) load(PD, R[0]);
    wave_lds_fence();
    frags(f[0], ring);
    rs_stamp(trace, 1);
    // rolled steady state + unrolled tail (see ARStream::run: straight-line code of this length misses the instruction cache in the step)
    constexpr int U = (PD % 2) ? 2 * PD : PD;
    constexpr int NMAIN = NS > PD + 1 ? ((NS - PD - 1) / U) * U : 0;
    auto step = [&](int s, int u, bool more, bool req) {
      if (more) {
        char* img = ring + ((u + 1) & 1) * C::IMG;
        wave_lds_fence();                                     // (the fragment reads of k-step s - 1 from this image are behind every lane)
        stage(img, R[(u + 1) % PD]);
        if (req) load(s + 1 + PD, R[(u + 1) % PD]);
        wave_lds_fence();
        frags(f[(u + 1) & 1], img);
      }
      mma(f[u & 1]);
    };
#pragma unroll 1
    for (int s0 = 0; s0 < NMAIN; s0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) step(s0 + u, u, true, true);
    }
#pragma unroll
    for (int s = NMAIN; s < NS; ++s) step(s, s % U, s + 1 < NS, s + 1 + PD < NS);
  }
  // the four partial tiles meet in LDS: [source wave][row tile][column tile][lane] as f32x4
  rs_stamp(trace, 2);
  __syncthreads();                                                   // every wave is done with its images
  rs_stamp(trace, 3);
  f32x4* part = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) part[((wave * 4 + mt) * 4 + nt) * 64 + lane] = acc[mt][nt];
  __syncthreads();
  f32x4 sum[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    sum[nt] = part[((0 * 4 + wave) * 4 + nt) * 64 + lane];
#pragma unroll
    for (int src = 1; src < 4; ++src) sum[nt] += part[((src * 4 + wave) * 4 + nt) * 64 + lane];
  }
  // lane: token row m0 + 16 wave + (lane & 15), columns n0 + 16 (lane >> 4) + 4 nt + r
  const int m = m0 + 16 * wave + (lane & 15);
  if (m < M) {
    epi.apply8(m, n0 + 16 * (lane >> 4), sum[0], sum[1]);
    epi.apply8(m, n0 + 16 * (lane >> 4) + 8, sum[2], sum[3]);
  }
  rs_stamp(trace, 4);
}

template <int NS, int PD, class Epi>
inline void launch_gemm_sk_n(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, Epi epi, hipStream_t st, unsigned long long* trace) {
  const dim3 grid((N / GemmSK::BN) * ((M + GemmSK::BM - 1) / GemmSK::BM));
  APH_ALLOW_SMEM((gemm_sk_kernel<NS, PD, Epi>), GemmSK::SMEM);
  APH_LAUNCH((gemm_sk_kernel<NS, PD, Epi>), grid, dim3(GemmSK::NTHREAD), GemmSK::SMEM, st, A, lda, Bt, ldb, M, N, epi, trace);
}
// the K values the kernel is instantiated for (k-steps per wave = K / 128): the linears of a ViT of width 768 (B/32, B/16) and of the
// width-256 test models; other shapes stay on the ring kernels
inline bool gemm_sk_fits(int N, int K) { return N % 64 == 0 && (K == 256 || K == 768 || K == 1024 || K == 2304 || K == 3072); }
template <int PD, class Epi>
inline void launch_gemm_sk(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                           unsigned long long* trace = nullptr) {
  switch (K) {
    case 256: launch_gemm_sk_n<2, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    case 768: launch_gemm_sk_n<6, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    case 1024: launch_gemm_sk_n<8, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    case 2304: launch_gemm_sk_n<18, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    default: launch_gemm_sk_n<24, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;       // 3072
  }
}

#ifdef APH_EXPERIMENTS       // measured slower than the ring kernels inside the step at every shard size (profiles/r05_gemm_rs_shapes.txt,
                             // r05_fused_v4_steps.txt): compiled for A/B builds and the CPU interpreter only, not into the product library
// ---- A-resident kernel ----------------------------------------------------------------------------------------------------------------
// shared pieces (also used by vit_block.h): the resident A block is nks k-step images of [64 rows][64 bytes]; behind it every wave has two
// private images of [16 NT weight rows][64 bytes].
template <int NT>
struct GemmAR {
  static constexpr int BM = 64, BN = 64 * NT, NTHREAD = 256, KMAX = 1024;
  static constexpr int WIMG = NT * 1024;                     // bytes of one weight k-step image of a wave
  static constexpr int smem(int K) { return (K / 32) * 4096 + 4 * 2 * WIMG; }
};

// Fill the resident A block from a row-major f16 matrix: rowoff = byte offsets of this lane's loads of the rows 16 q + lrow (clamped by
// the caller); wave w copies the k-steps w, w + 4, ...; six k-steps (24 loads) in flight per wave and batch.  NKS at compile time: for
// NKS % 4 == 0 no load sits behind a run-time condition.
template <int NKS>
__device__ __forceinline__ void ar_fill_copy(char* a_img, const char* Ab, const unsigned (&rowoff)[4], int wave, const RSLane& L) {
#pragma unroll
  for (int b0 = 0; b0 < NKS; b0 += 24) {
    half8 r[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (b0 + 4 * j + 3 < NKS || (b0 + 4 * j < NKS && b0 + 4 * j + wave < NKS)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[j][q] = ldg8(Ab + rowoff[q] + (size_t)(b0 + 4 * j + wave) * 64);
      }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (b0 + 4 * j + 3 < NKS || (b0 + 4 * j < NKS && b0 + 4 * j + wave < NKS)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sts8(a_img + (b0 + 4 * j + wave) * 4096 + q * 1024 + L.wpos, r[j][q]);
      }
  }
}

// The weight stream of one wave: acc[mt][nt] += A block x (its 16 NT weight rows)^T over nks k-steps.  woff[nt]: byte offset of this
// lane's load of tile nt (the weight row of tile row lrow, chunk lpc) from Bb.  prefetch() requests k-steps 0 .. PD - 1 so that they fly
// during the caller's prologue; run() does the rest.
template <int NT, int PD>
struct ARStream {
  half8 R[PD][NT];
  __device__ __forceinline__ void request(int s, int d, const char* Bb, const unsigned (&woff)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) R[d][nt] = ldg8(Bb + woff[nt] + (size_t)s * 64);
  }
  template <int NKS>
  __device__ __forceinline__ void prefetch(const char* Bb, const unsigned (&woff)[NT]) {
#pragma unroll
    for (int d = 0; d < PD; ++d)
      if (d < NKS) request(d, d, Bb, woff);
  }
  // NKS = K / 32 at COMPILE time: straight-line code, unconditional loads, exact vmcnt bookkeeping (see gemm_sk_kernel)
  template <int NKS>
  __device__ __forceinline__ void run(f32x4 (&acc)[4][NT], const char* a_img, char* wimg, const char* Bb, const unsigned (&woff)[NT], const RSLane& L) {
    struct Frags { half8 t[4], w[NT]; };
    Frags f[2];
    auto stage = [&](char* img, int d) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sts8(img + nt * 1024 + L.wpos, R[d][nt]);
    };
    auto frags = [&](Frags& fr, const char* img, int s) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fr.t[mt] = lds8(a_img + s * 4096 + mt * 1024 + L.fpos);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fr.w[nt] = lds8(img + nt * 1024 + L.fpos);
    };
    stage(wimg, 0);
    if (PD < NKS) request(PD, 0, Bb, woff);
    wave_lds_fence();
    frags(f[0], wimg, 0);
    // Steady state as a ROLLED loop of U k-steps per trip (every step of it requests a k-step: no load behind a condition, exact vmcnt
    // bookkeeping), the last steps unrolled with compile-time conditions.  Fully unrolled, the loop was ~14 KiB of straight-line code that a
    // wave runs through once: in the step, where three dozen kernels take turns, that is an instruction-cache miss stream per launch
    // (18-21 us in the step against 13 us back to back: profiles/r05_gemm_rs_shapes.txt, r05_kernel_stats_s26_fused_v3.csv).
    constexpr int U = (PD % 2) ? 2 * PD : PD;
    constexpr int NMAIN = NKS > PD + 1 ? ((NKS - PD - 1) / U) * U : 0;
    auto step = [&](int s, int u, bool more, bool req) {          // k-step s (register set / image parity by u = s mod U)
      if (more) {
        char* img = wimg + ((u + 1) & 1) * (NT * 1024);
        wave_lds_fence();
        stage(img, (u + 1) % PD);
        if (req) request(s + 1 + PD, (u + 1) % PD, Bb, woff);
        wave_lds_fence();
        frags(f[(u + 1) & 1], img, s + 1);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f[u & 1].w[nt], f[u & 1].t[mt], acc[mt][nt]);
    };
#pragma unroll 1
    for (int s0 = 0; s0 < NMAIN; s0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) step(s0 + u, u, true, true);
    }
#pragma unroll
    for (int s = NMAIN; s < NKS; ++s) step(s, s % U, s + 1 < NKS, s + 1 + PD < NKS);
  }
};

__device__ __forceinline__ void ar_tile(int ntm, int& tm, int& tn);
// The same stream from FRAGMENT-MAJOR packed weights: Bp [column group][wave][k-step][tile nt][lane][8 halfs], lane l = tile row l & 15,
// chunk l >> 4 -- a wave's load of one fragment is 1 KiB contiguous and lands in the registers the MFMA reads: no LDS staging of the weights.
template <int NT, int PD>
struct ARStreamP {
  half8 R[PD][NT];
  __device__ __forceinline__ void request(int s, int d, const char* Bw) {          // Bw: this wave's stream + lane * 16
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) R[d][nt] = ldg8(Bw + (size_t)(s * NT + nt) * 1024);
  }
  template <int NKS>
  __device__ __forceinline__ void prefetch(const char* Bw) {
#pragma unroll
    for (int d = 0; d < PD; ++d)
      if (d < NKS) request(d, d, Bw);
  }
  template <int NKS>
  __device__ __forceinline__ void run(f32x4 (&acc)[4][NT], const char* a_img, const char* Bw, const RSLane& L) {
    half8 t[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) t[0][mt] = lds8(a_img + mt * 1024 + L.fpos);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      if (s + 1 < NKS) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) t[(s + 1) & 1][mt] = lds8(a_img + (s + 1) * 4096 + mt * 1024 + L.fpos);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_16x16x32_f16(R[s % PD][nt], t[s & 1][mt], acc[mt][nt]);
      if (s + PD < NKS) request(s + PD, s % PD, Bw);
    }
  }
};

// Bt [N, K] row-major -> the packed image for column groups of 64 NT (one thread per 16-byte piece)
template <int NT>
__global__ void pack_frag_kernel(const half_t* __restrict__ Bt, half_t* __restrict__ Bp, int N, int K) {
  const int nks = K / 32;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)(N / 16) * nks * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  size_t r = idx >> 6;
  const int nt = (int)(r % NT); r /= NT;
  const int ks = (int)(r % nks); r /= nks;
  const int wave = (int)(r & 3), tn = (int)(r >> 2);
  const int i = lane & 15, row = tn * 64 * NT + wave * 16 * NT + 4 * NT * (i >> 2) + 4 * nt + (i & 3);
  *reinterpret_cast<half8*>(Bp + idx * 8) = *reinterpret_cast<const half8*>(Bt + (size_t)row * K + ks * 32 + (lane >> 4) * 8);
}

template <int NT, int NKS, int PD, class Epi>
__global__ __launch_bounds__(256) void gemm_arp_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bp, int M, int N, Epi epi,
                                                       unsigned long long* __restrict__ trace) {
  using C = GemmAR<NT>;
  APH_DYN_SMEM(smem);
  rs_stamp(trace, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  int tm, tn;
  ar_tile((M + C::BM - 1) / C::BM, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN + wave * 16 * NT;
  const RSLane L(lane);
  const char* Bw = reinterpret_cast<const char*>(Bp) + ((size_t)(tn * 4 + wave) * NKS * NT * 64 + lane) * 16;
  ARStreamP<NT, PD> W;
  W.template prefetch<NKS>(Bw);
  unsigned rowoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int am = m0 + 16 * q + L.lrow;
    am = am < M ? am : M - 1;
    rowoff[q] = ((unsigned)am * (unsigned)lda + L.lpc * 8) * 2u;
  }
  ar_fill_copy<NKS>(smem, reinterpret_cast<const char*>(A), rowoff, wave, L);
  rs_stamp(trace, 1);
  __syncthreads();
  rs_stamp(trace, 2);
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  W.template run<NKS>(acc, smem, Bw, L);
  rs_stamp(trace, 3);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + 16 * mt + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) epi.apply8(m, n0 + 4 * NT * (lane >> 4) + 8 * j, acc[mt][2 * j], acc[mt][2 * j + 1]);
    }
  }
  rs_stamp(trace, 4);
}

// tile order of the A-resident kernels: column groups slowest, so that the contiguous run of an XCD holds few column groups (their weight
// rows stay in its L2) and all row blocks of each
__device__ __forceinline__ void ar_tile(int ntm, int& tm, int& tn) {
  const int t = rs_tile_index();
  tn = t / ntm;
  tm = t - tn * ntm;
}

template <int NT, int NKS, int PD, class Epi>
__global__ __launch_bounds__(256) void gemm_ar_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt, int ldb, int M, int N,
                                                      Epi epi, unsigned long long* __restrict__ trace) {
  using C = GemmAR<NT>;
  APH_DYN_SMEM(smem);
  rs_stamp(trace, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  constexpr int nks = NKS;
  int tm, tn;
  ar_tile((M + C::BM - 1) / C::BM, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN + wave * 16 * NT;        // this wave's first column
  const RSLane L(lane);
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bt);
  unsigned woff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)      // tile nt, tile row i = lrow is weight row 4 NT (i >> 2) + 4 nt + (i & 3) of the wave's 16 NT
    woff[nt] = ((unsigned)(n0 + 4 * NT * (L.lrow >> 2) + 4 * nt + (L.lrow & 3)) * (unsigned)ldb + L.lpc * 8) * 2u;
  ARStream<NT, PD> W;
  W.template prefetch<NKS>(Bb, woff);                                         // the first weight k-steps fly during the fill
  unsigned rowoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int am = m0 + 16 * q + L.lrow;
    am = am < M ? am : M - 1;
    rowoff[q] = ((unsigned)am * (unsigned)lda + L.lpc * 8) * 2u;
  }
  ar_fill_copy<NKS>(smem, Ab, rowoff, wave, L);
  rs_stamp(trace, 1);
  __syncthreads();
  rs_stamp(trace, 2);
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  W.template run<NKS>(acc, smem, smem + nks * 4096 + wave * (2 * C::WIMG), Bb, woff, L);
  rs_stamp(trace, 3);
  // lane: token rows m0 + 16 mt + (lane & 15), columns n0 + 4 NT (lane >> 4) + 4 nt + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + 16 * mt + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) epi.apply8(m, n0 + 4 * NT * (lane >> 4) + 8 * j, acc[mt][2 * j], acc[mt][2 * j + 1]);
    }
  }
  rs_stamp(trace, 4);
}

template <int NT, int NKS, int PD, class Epi>
inline void launch_gemm_ar_n(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, Epi epi, hipStream_t st, unsigned long long* trace) {
  using C = GemmAR<NT>;
  const dim3 grid((N / C::BN) * ((M + C::BM - 1) / C::BM));
  APH_ALLOW_SMEM((gemm_ar_kernel<NT, NKS, PD, Epi>), C::smem(NKS * 32));
  APH_LAUNCH((gemm_ar_kernel<NT, NKS, PD, Epi>), grid, dim3(C::NTHREAD), C::smem(NKS * 32), st, A, lda, Bt, ldb, M, N, epi, trace);
}
// K = the width of a ViT (256 ... 1024): the block fits LDS
inline bool gemm_ar_fits(int N, int K) { return N % GemmAR<4>::BN == 0 && (K == 256 || K == 512 || K == 768 || K == 1024); }
template <int NT, int PD, class Epi>
inline void launch_gemm_ar(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                           unsigned long long* trace = nullptr) {
  switch (K) {
    case 256: launch_gemm_ar_n<NT, 8, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    case 512: launch_gemm_ar_n<NT, 16, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    case 768: launch_gemm_ar_n<NT, 24, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;
    default: launch_gemm_ar_n<NT, 32, PD>(A, lda, Bt, ldb, M, N, epi, st, trace); break;       // 1024
  }
}

#endif  // APH_EXPERIMENTS

// the split-K kernel if it is instantiated for this shape (false: the caller falls back to the ring kernels of vit_gemm.h); `wide`
// (-DAPH_EXPERIMENTS builds): the A-resident kernel for wide outputs over K = width
template <class Epi>
inline bool launch_gemm_rs_auto(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st, bool wide) {
#ifdef APH_EXPERIMENTS      // (the A-resident kernel behind every wide ViT GEMM: measured slower than the ring kernels in the step; A/B builds only)
  if (wide && N >= 4 * GemmAR<4>::BN && gemm_ar_fits(N, K)) { launch_gemm_ar<4, 8>(A, lda, Bt, ldb, M, N, K, epi, st); return true; }
#endif
  (void)wide;
  if (gemm_sk_fits(N, K)) launch_gemm_sk<4>(A, lda, Bt, ldb, M, N, K, epi, st);
  else return false;
  return true;
}

}  // namespace aph
