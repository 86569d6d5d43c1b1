// Wave-specialised persistent fp16 GEMM for the CLIP ViT linears at full batch (round 3), gfx950.
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]        same contract as vit_gemm.h
//
// What round 2 measured: a GEMM launch is main loops (HBM idle) followed by a store burst (MFMA
// idle), a K = 768 output tile spends ~40 % of its time outside the main loop (workgroup turnover, first-DMA latency,
// LDS-staged epilogue, store acknowledgements), and a persistent tile loop did not help because a wave's loads and stores
// retire through ONE in-order vmcnt: the first counted DMA wait of the next tile also waited for the epilogue's stores.
// This kernel separates the two kinds of memory traffic by WAVE:
//   * 2 PRODUCER waves do nothing but issue the operand DMA (global_load_lds_dwordx4) for a flat stream of k-tiles that
//     runs ACROSS output tiles (3-stage ring of 48 KiB: 256 x 64 of A + 128 x 64 of Bt); their vmcnt counts only DMA;
//   * 8 CONSUMER waves (4 x 2 of 64 x 64) read fragments, run the MFMAs and store their accumulators straight from
//     registers: their vmcnt counts only epilogue traffic and the main loop never waits on it, so the stores of tile i
//     drain while tile i + 1 is being multiplied, and the operands of tile i + 1 are already in LDS when tile i ends.
//   * No LDS staging in the epilogue: the rows of the Bt tile are PERMUTED on their way into LDS (the DMA source address
//     is per lane, so this is free) such that a lane's accumulator registers are four consecutive columns of four rows and the 16
//     lanes l & 15 hold a row's 64 columns in order: stores leave as whole rows from adjacent lanes (gemm_ws_brow).  The ring is
//     never idle, no workgroup barrier separates main loop and epilogue.
// One s_barrier per k-tile, shared by all ten waves (the protocol of the ring kernels in vit_gemm.h):
//   producers: DMA of unit u+1 landed (counted vmcnt, unit u+2 stays in flight) -> barrier -> issue unit u+3 into the
//              stage unit u just vacated;
//   consumers: own fragment reads of unit u complete (lgkmcnt) -> barrier -> read unit u+1.
// 10 waves = 3 per SIMD on two SIMDs: the kernel must stay within 168 VGPRs (launch bound 640).
#pragma once
#include "vit_gemm.h"

namespace aph {

template <int BM_, int NPROD_, int NSTAGE_, int WG_PER_CU_, int BIAS_MAX_, int FORCE_WAVES_ = 0>
struct GemmWSCfg {
  static constexpr int BM = BM_, BN = 128, WM = BM / 64, WN = 2, NCONS = WM * WN, NPROD = NPROD_, NWAVE = NCONS + NPROD, NTHREAD = NWAVE * 64;
  static constexpr int NSTAGE = NSTAGE_, WG_PER_CU = WG_PER_CU_;
  static constexpr int STAGE = (BM + BN) * GEMM_BK;      // halfs per stage
  static constexpr int SMEM = NSTAGE * STAGE * 2;        // bytes of the ring
  static constexpr int BIAS_MAX = BIAS_MAX_;             // floats of bias staged behind the ring
  static constexpr int SMEM_TOTAL = SMEM + BIAS_MAX * 4;
  static constexpr int QA = BM / 8, QB = BN / 8;         // DMA instructions (8 tile rows of 128 B each) per unit
  static constexpr int QPW = (QA + QB) / NPROD;          // per producer wave
  static constexpr int QAW = QA / NPROD;                 // of which the first QAW fetch A rows
  static constexpr int MIN_WAVES_PER_SIMD = FORCE_WAVES_ ? FORCE_WAVES_ : (WG_PER_CU * NWAVE + 3) / 4;
  static_assert(QA % NPROD == 0 && QB % NPROD == 0 && (NSTAGE - 1) * QPW <= 63, "the DMA in flight must fit the vmcnt range");
  static_assert(WG_PER_CU * SMEM_TOTAL <= 160 * 1024 && NSTAGE >= 2 && NSTAGE <= 3, "LDS budget / ring depth");
};
// 256 x 128 tiles, one workgroup per CU: 8 consumers + 2 producers, 3 x 48 KiB ring (+ 16 KiB bias)
using GemmWS = GemmWSCfg<256, 2, 3, 1, 4096>;
// (Measured and rejected, profiles/r03_gemm_ws_pair.txt: 128 x 128 tiles on TWO workgroups per CU -- GemmWSCfg<128, 1, 2, 2, 3072>, 4 consumers +
// 1 producer each (also 2 producers; also forced to 128 VGPRs), so that one workgroup's epilogue sits under the other's MFMAs.  76 KiB per
// workgroup only leaves a 2-stage ring, which exposes the DMA latency on every k-tile: a lone 128 x 128 workgroup needs 0.77 us per k-tile
// where the 256 x 128 one needs 0.84 for twice the work, and two of them per CU do not make up for it (QKV 55-68 us against 47-49; fc2, one
// tile per workgroup and every workgroup resident: 65 us against 42).  About a third of the second workgroups also started only when a first
// one had left (entry / exit stamps on the chip-wide clock), although the occupancy API reports 2 per CU: a 5-wave workgroup puts two waves on
// one SIMD, and when both workgroups' pairs land on the same SIMD, 4 waves x 144 VGPRs do not fit.  Second look: with one fragment buffer the
// kernel fits 128 VGPRs (four waves per SIMD) and every one of the 512 workgroups enters within 0.8 us -- and the pair is then EQUAL to the
// single 256 x 128 workgroup, not better (QKV f16 41.0 vs 38.9 us, fc1 + GELU 64.9 vs 63.3, fc2 52.9 vs 49.7): what the out-of-phase partner
// hides is given back by the 2-stage ring, the exposed fragment reads and the smaller tile.)

// LDS row v of the Bt tile (0..127) holds tile row perm(v): within each 64-row block (one consumer column group), fragment row
// i = v & 15 of column tile nt = (v >> 4) & 3 is weight row 4 i + nt.  The MFMAs take the TOKENS as their A fragment and the weights
// as B (vit_gemm.h does it the other way round for its LDS-staged epilogue), so lane l, register r of acc[mt][nt] is
//     row  16 mt + 4 (l >> 4) + r ,   column  4 (l & 15) + nt        of the wave's 64 x 64 tile:
// for a fixed (mt, r) a lane holds FOUR CONSECUTIVE columns (nt = 0..3) and the 16 lanes l & 15 = 0..15 hold the row's 64 columns in order.
// A store instruction therefore writes whole rows from ADJACENT lanes (16 lanes x 8 B = one 128-byte line of f16, x 16 B = two lines of
// f32; four rows per instruction), which is what the address coalescer wants: the first version of this epilogue kept the weights-as-A
// order, gave a lane 16 consecutive columns and exchanged pieces among lanes 16 apart (v_permlane16/32_swap) -- its stores were 64
// contiguous bytes per row, but from lanes 16 apart, and issued at a quarter of the store path's rate (15 B/clk per CU by the per-tile
// trace, profiles/r03_gemm_ws_trace.txt; the lane mapping of the swaps is probed by tools/exp/permlane_probe.hip).
__host__ __device__ __forceinline__ int gemm_ws_brow(int v) {
  const int vv = v & 63, nt = vv >> 4, i = vv & 15;
  return (v & ~63) + (i << 2) + nt;
}

__device__ __forceinline__ half4 pack_h4(float a, float b, float c, float d) { return half4{(half_t)a, (half_t)b, (half_t)c, (half_t)d}; }

// ---- epilogues of the wave-specialised kernel ------------------------------------------------------------------------
// ws_tile(epi, m4, M, n4, v, lb): v[nt][r] = C[m4 + r][n4 + nt], r = 0..3 (four rows: m4 = tile row base + 4 (l >> 4)), nt = 0..3 (four
// consecutive columns: n4 = wave column base + 4 (l & 15)); lb = this GEMM's bias vector in LDS (or null).  Rows >= M are not stored.
#define APH_WS_ROWS(r, m4, M) for (int r = 0; r < 4; ++r) if ((m4) + r < (M))
__device__ __forceinline__ void ws_tile(const EpiF16& e, int m4, int M, int n4, f32x4 (&v)[4], const float* lb) {
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (e.bias) b = *reinterpret_cast<const f32x4*>(lb + n4);
#pragma unroll
  APH_WS_ROWS(r, m4, M)
    *reinterpret_cast<half4*>(e.out + (size_t)(m4 + r) * e.ldo + n4) = pack_h4(v[0][r] + b[0], v[1][r] + b[1], v[2][r] + b[2], v[3][r] + b[3]);
}
__device__ __forceinline__ void ws_tile(const EpiF16Scale& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
#pragma unroll
  APH_WS_ROWS(r, m4, M)
    *reinterpret_cast<half4*>(e.out + (size_t)(m4 + r) * e.ldo + n4) = pack_h4(v[0][r] * e.scale, v[1][r] * e.scale, v[2][r] * e.scale, v[3][r] * e.scale);
}
__device__ __forceinline__ void ws_tile(const EpiF32& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
#pragma unroll
  APH_WS_ROWS(r, m4, M) st4(e.out + (size_t)(m4 + r) * e.ldo + n4, f32x4{v[0][r], v[1][r], v[2][r], v[3][r]} * e.scale);
}
__device__ __forceinline__ void ws_tile(const EpiNoStore& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
  const f32x4 t = v[0] + v[1] + v[2] + v[3];            // every accumulator stays live
  if (m4 < M && t[0] + t[1] + t[2] + t[3] == 1.2345678e33f) st4(e.out + (size_t)m4 * e.ldo + n4, t);
}
__device__ __forceinline__ void ws_tile(const EpiResidual& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
  // residual and bias are already in the accumulators (ws_init below): store only
#pragma unroll
  APH_WS_ROWS(r, m4, M) st4(e.out + (size_t)(m4 + r) * e.ldo + n4, f32x4{v[0][r], v[1][r], v[2][r], v[3][r]});
}
__device__ __forceinline__ void ws_tile(const EpiPatchEmbed& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
#pragma unroll
  APH_WS_ROWS(r, m4, M) {
    const int m = m4 + r, s = m / e.P, p = m - s * e.P;
    st4(e.x0 + ((size_t)s * e.T + 1 + p) * e.D + n4, f32x4{v[0][r], v[1][r], v[2][r], v[3][r]} + ld4(e.pos + (size_t)(1 + p) * e.D + n4));
  }
}
__device__ __forceinline__ void ws_tile(const EpiGelu& e, int m4, int M, int n4, f32x4 (&v)[4], const float* lb) {
  const f32x4 b = *reinterpret_cast<const f32x4*>(lb + n4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 gl, dg;
    quick_gelu4(f32x4{v[0][r] + b[0], v[1][r] + b[1], v[2][r] + b[2], v[3][r] + b[3]}, gl, dg);
    if (m4 + r < M) {
      *reinterpret_cast<half4*>(e.g + (size_t)(m4 + r) * e.ldo + n4) = pack_h4(gl[0], gl[1], gl[2], gl[3]);
      *reinterpret_cast<half4*>(e.dg + (size_t)(m4 + r) * e.ldo + n4) = pack_h4(dg[0], dg[1], dg[2], dg[3]);
    }
  }
}
__device__ __forceinline__ void ws_tile(const EpiGeluBwd& e, int m4, int M, int n4, f32x4 (&v)[4], const half4 (&d)[4]) {
#pragma unroll
  APH_WS_ROWS(r, m4, M)
    *reinterpret_cast<half4*>(e.out + (size_t)(m4 + r) * e.ldo + n4) =
        pack_h4(v[0][r] * (float)d[r][0], v[1][r] * (float)d[r][1], v[2][r] * (float)d[r][2], v[3][r] * (float)d[r][3]);
}
#undef APH_WS_ROWS
// the whole 64 x 64 wave tile: acc[mt] is the 16-row block at m4 + 16 mt
template <class E>
__device__ __forceinline__ void ws_tiles(const E& e, int m4, int M, int n4, f32x4 (&acc)[4][4], const float* lb) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ws_tile(e, m4 + mt * 16, M, n4, acc[mt], lb);
}
// QuickGELU backward reads dg/du: every load of the tile is issued before the first store (the compiler cannot prove that `out` and `dg`
// do not alias; interleaved, each row's load would be waited for between two stores: 88 instead of 70 us per launch)
__device__ __forceinline__ void ws_tiles(const EpiGeluBwd& e, int m4, int M, int n4, f32x4 (&acc)[4][4], const float*) {
  half4 d[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m4 + mt * 16 + r;
      d[mt][r] = *reinterpret_cast<const half4*>(e.dg + (size_t)(m < M ? m : M - 1) * e.ldo + n4);       // (rows past M: clamped read, never stored)
    }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ws_tile(e, m4 + mt * 16, M, n4, acc[mt], d[mt]);
}
// ws_init: the accumulators' starting value (same arguments as ws_tile).  The residual epilogue starts them from res + bias: the loads'
// latency hides behind the tile's first DMA wait instead of sitting between main loop and stores.
template <class E>
__device__ __forceinline__ void ws_init(const E&, int, int, int, f32x4 (&v)[4], const float*) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void ws_init(const EpiResidual& e, int m4, int M, int n4, f32x4 (&v)[4], const float*) {
  const f32x4 b = ld4(e.bias + n4);          // from global (3 KiB, L2-resident), not from the LDS copy: the first tile's init runs BEFORE the first barrier
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m4 + r < M ? m4 + r : M - 1;                      // (rows past M: clamped read, never stored)
    const f32x4 x = ld4(e.res + (size_t)m * e.ldo + n4) + b;
    v[0][r] = x[0]; v[1][r] = x[1]; v[2][r] = x[2]; v[3][r] = x[3];
  }
}
// the bias vector an epilogue wants staged in LDS (nullptr: none)
__device__ __forceinline__ const float* ws_bias(const EpiF16& e) { return e.bias; }
__device__ __forceinline__ const float* ws_bias(const EpiResidual& e) { return e.bias; }
__device__ __forceinline__ const float* ws_bias(const EpiGelu& e) { return e.bias; }
template <class E> __device__ __forceinline__ const float* ws_bias(const E&) { return nullptr; }

// chip-wide constant 100 MHz clock (measurement hook only): comparable across workgroups, unlike s_memtime
__device__ __forceinline__ unsigned long long ws_realtime() {
#ifdef APH_EMU
  return 0ull;
#else
  return __builtin_amdgcn_s_memrealtime();
#endif
}
// shader clock (measurement hook only)
__device__ __forceinline__ unsigned long long ws_clock() {
#ifdef APH_EMU
  return 0ull;
#else
  return __builtin_amdgcn_s_memtime();
#endif
}

template <int AHEAD_UNIT>
__device__ __forceinline__ void ws_wait_barrier(int ahead) {      // all but the newest `ahead` units of this producer wave's DMA have landed, then barrier
  if (ahead <= 0) wait_vm_barrier<0>();
  else if (ahead == 1) wait_vm_barrier<AHEAD_UNIT>();
  else wait_vm_barrier<(2 * AHEAD_UNIT <= 63 ? 2 * AHEAD_UNIT : 63)>();
}

template <class C, class Epi>
__global__ __launch_bounds__(C::NTHREAD, C::MIN_WAVES_PER_SIMD) void gemm_ws_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt,
                                                                  int ldb, int M, int N, int K, Epi epi, int ntiles, int pgroup, unsigned long long* __restrict__ trace,
                                                                  int ntn_short, int nk_short) {
  using F = GemmBig;                             // fragment shapes of a consumer wave: 4 x 4 MFMA tiles (64 x 64)
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  // persistent tile run (as gemm8_f16_kernel): XCD x owns one contiguous run of tiles, n-tiles fastest; its workgroups walk it.
  // (Measured and rejected: a RECTANGLE of the tile grid per XCD -- a quarter of the row panels x half of the column tiles -- so that an
  // XCD's share of the weights stays in its L2 instead of being re-fetched per row panel (284 MB of fabric traffic per fc1 launch for 136 MB
  // algorithmic).  Slower everywhere: fc1 58.3 vs 53.8 us, QKV 43.5 vs 42.0, and the single-round N = 768 shapes lose their one-tile-per-
  // workgroup balance (fc2 73.6 vs 44.4 us).  The re-fetches are served by the 256 MiB Infinity Cache and are not what bounds these launches.)
  // [r5] TWO tile spaces: the LONG column tiles (the first ntn - ntn_short of a row panel: k-loop over all of K) and the SHORT ones (the last
  // ntn_short: k-loop over the first nk_short k-tiles only -- the V columns of the split-precision QKV GEMM, which take the hi half of
  // [hi | lo] rows alone).  An XCD owns one contiguous run of each space and walks its long tiles first, then its short ones: a workgroup's
  // positions p, p + step, ... then hold its most expensive tiles first and the cheap ones last, which is what balances the walk (456 long +
  // 228 short tiles at C2: 228 workgroups take two long + one short = 60 k-tile rounds where the one-space walk gives 72 to some).
  // ntn_short = 0 is the plain GEMM: one space, the walk of rounds 3 / 4.
  const int nk = K / GEMM_BK, ntn = N / C::BN, ntm = ntiles / ntn, ntn_long = ntn - ntn_short;
  int pos, pos_end, tile_step, lstart, sstart, nlong;
  {
    const int nwg = gridDim.x, b = blockIdx.x, G = nwg < 8 ? nwg : 8, xcd = b % G;
    auto share = [&](int n, int& start, int& count) {
      const int q = n / G, r = n - q * G;
      start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      count = q + (xcd < r ? 1 : 0);
    };
    int nshort;
    share(ntm * ntn_long, lstart, nlong);
    share(ntm * ntn_short, sstart, nshort);
    pos_end = nlong + nshort;
    tile_step = (nwg - xcd + G - 1) / G;
    pos = b / G;
  }
  if (pos >= pos_end) return;                    // (workgroup-uniform; does not happen with gridDim.x <= ntiles)
  if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + 15) * 4 + 3] = ws_realtime();   // kernel entry, chip-wide 100 MHz clock
  int U = 0;                                     // k-tile units of this workgroup
  for (int q = pos; q < pos_end; q += tile_step) U += q < nlong ? nk : nk_short;
  // Tile order inside the XCD runs: groups of `pgroup` row panels; inside a group the tiles of ONE column tile across the group's panels
  // are consecutive, then the next column tile.  The 32 workgroups of an XCD work on 32 consecutive tiles, i.e. on pgroup row panels x
  // (32 / pgroup) column tiles: with pgroup = 4 that is 1.5 MB of A panels + 1.5 MB of weight rows -- it fits the XCD's 4 MiB L2, and a
  // weight tile is fetched from the fabric once per FOUR row panels instead of once per panel (n-fastest order, pgroup = 1: 32 consecutive
  // tiles = 1.3-1.8 panels x all of W, 3.5-4.7 MB at N = 2304 / 3072, which the L2 cannot hold next to the A panels: 284 MB of fabric
  // traffic per fc1 launch for 136 MB algorithmic, profiles/r03_pmc_hbm_traffic.json).  The runs and their lengths are unchanged, so the
  // balance of the persistent walk is too (what the per-XCD rectangle order of round 3 lost).
  auto coords = [&](int q, int& tm, int& tn) -> int {          // position q of this XCD's walk -> tile coordinates; returns the tile's k-tile count
    const bool lng = q < nlong;
    const int t = lng ? lstart + q : sstart + (q - nlong), cols = lng ? ntn_long : ntn_short;
    const int per = pgroup * cols, g = t / per, rem = t - g * per;
    const int left = ntm - g * pgroup, rg = left < pgroup ? left : pgroup;
    tn = rem / rg;
    tm = g * pgroup + (rem - tn * rg);
    if (!lng) tn += ntn_long;
    return lng ? nk : nk_short;
  };
  // the epilogue's bias vector goes to LDS behind the ring once (N <= 4096 floats); visible after the first barrier
  float* lbias = reinterpret_cast<float*>(smem + C::SMEM);
  if (const float* gb = ws_bias(epi)) {
    for (int i = tid * 4; i < N; i += C::NTHREAD * 4) *reinterpret_cast<f32x4*>(lbias + i) = ld4(gb + i);
    wait_lgkm0();
  }

  if (wave >= C::NCONS) {
    // ---------------------------------------------------------------- producer
    const int p = wave - C::NCONS, lrow = lane >> 3, pc = lane & 7;
    unsigned off[C::QPW];                        // per-lane byte offsets from the matrix bases, current issue tile
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(Bt);
    int nk_it = nk;                              // k-tiles of the tile being issued
    auto setup = [&](int t) {
      int tm, tn;
      nk_it = coords(t, tm, tn);
      const int n0 = tn * C::BN, m0 = tm * C::BM;
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
    int it = pos, ikt = 0, ist = 0, issued = 0;
    setup(it);
    auto issue_next = [&]() {
      issue(ikt, ist);
      ist = ist == C::NSTAGE - 1 ? 0 : ist + 1;
      ++issued;
      if (++ikt == nk_it) {
        ikt = 0;
        it += tile_step;
        if (issued < U) setup(it);
      }
    };
    const int pre = U < C::NSTAGE - 1 ? U : C::NSTAGE - 1;
    for (int i = 0; i < pre; ++i) issue_next();
    ws_wait_barrier<C::QPW>(pre - 1);                                                        // unit 0 has landed
    if (U > C::NSTAGE - 1) issue_next();
    for (int u = 0; u + 1 < U; ++u) {
      const int rem = U - u - 2;                                                             // units issued beyond u+1
      ws_wait_barrier<C::QPW>(rem < C::NSTAGE - 2 ? rem : C::NSTAGE - 2);                    // unit u+1 has landed; the stage of unit u is free
      if (u + C::NSTAGE < U) issue_next();
    }
    return;
  }

  // ------------------------------------------------------------------ consumer
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  f32x4 acc[F::TM][F::TN];
  const int frow = lane & 15, arow = wm * 64 + frow, brow = wn * 64 + frow, fchunk = lane >> 4;
  GemmFrags<F> f0, f1;
  int kt = 0, st = 0, nk_c = nk, ti = 0;          // k-tile inside the tile, ring stage, k-tiles of the current tile, tiles done
  auto init_tile = [&](int t) {
    int tm, tn;
    nk_c = coords(t, tm, tn);
    const int m4 = tm * C::BM + wm * 64 + 4 * (lane >> 4), n4 = tn * C::BN + wn * 64 + 4 * (lane & 15);
#pragma unroll
    for (int mt = 0; mt < F::TM; ++mt) ws_init(epi, m4 + mt * 16, M, n4, acc[mt], lbias);
  };
  auto mma = [&](const GemmFrags<F>& f) {                                    // tokens as the A fragment: D rows = token rows
#pragma unroll
    for (int mt = 0; mt < F::TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < F::TN; ++nt) acc[mt][nt] = mfma_16x16x32_f16(f.a[mt], f.b[nt], acc[mt][nt]);
  };
  init_tile(pos);                                                              // [r4] the residual epilogue's 16 loads per lane go out before the wait below:
  wait_vm_barrier<63>();                                                       // unit 0 has landed (and the bias vector is in LDS); vmcnt(63) does not wait for them
  gemm_load_frags<F>(f0, lds, lds + C::BM * GEMM_BK, arow, brow, fchunk);
  for (int u = 0; u < U; ++u) {
    const half_t* As = lds + st * C::STAGE;
    st = st == C::NSTAGE - 1 ? 0 : st + 1;
    gemm_load_frags<F>(f1, As, As + C::BM * GEMM_BK, arow, brow, 4 + fchunk);  // k-step 1 of unit u: in flight during the MFMAs
    mma(f0);                                                                   // k-step 0 of unit u
    if (u + 1 < U) {
      wait_lgkm0();                                                            // f1 has left LDS: the stage of unit u is dead for this wave
      wait_vm_barrier<63>();                                                   // (vmcnt(63): this wave's stores are never waited for here)
      const half_t* An = lds + st * C::STAGE;
      gemm_load_frags<F>(f0, An, An + C::BM * GEMM_BK, arow, brow, fchunk);    // k-step 0 of unit u+1: overlaps the MFMAs below
    }
    mma(f1);                                                                   // k-step 1 of unit u
    if (trace && kt == 0 && tid == 0 && ti < 14) trace[((size_t)blockIdx.x * 16 + ti) * 4 + 0] = ws_clock();      // first k-tile of a tile done
    if (++kt == nk_c) {
      if (trace && tid == 0 && ti < 14) trace[((size_t)blockIdx.x * 16 + ti) * 4 + 1] = ws_clock();      // (slots 14 / 15 hold the entry / exit stamps)                // main loop done
      // epilogue straight from the accumulators (layout: gemm_ws_brow above)
      int tm, tn;
      coords(pos, tm, tn);
      const int m4 = tm * C::BM + wm * 64 + 4 * (lane >> 4), n4 = tn * C::BN + wn * 64 + 4 * (lane & 15);
      ws_tiles(epi, m4, M, n4, acc, lbias);
      if (trace && tid == 0 && ti < 14) trace[((size_t)blockIdx.x * 16 + ti) * 4 + 2] = ws_clock();                // epilogue issued
      kt = 0;
      ++ti;
      pos += tile_step;
      if (u + 1 < U) init_tile(pos);
    }
  }
  if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + 14) * 4 + 3] = ws_realtime();     // consumer wave 0 done
}

// row panels per tile-order group (see `coords` in the kernel): 4 when more than one round of tiles shares the weights (wide outputs), 1 = the
// plain n-fastest order otherwise.  aph_gemm_set_ws_pgroup(): 0 = automatic, k > 0 = force k (A/B measurements, the tile-order tests)
inline int& gemm_ws_pgroup_override() {
  static int v = 0;
  return v;
}
inline int gemm_ws_panel_group(int ntn, int ntm) {
  int g = gemm_ws_pgroup_override();
  if (g <= 0) g = ntn >= 12 ? 4 : 1;
  return g < ntm ? g : (ntm > 0 ? ntm : 1);
}

template <class C, class Epi>
inline void launch_gemm_ws_cfg(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                               unsigned long long* trace, int n_short = 0, int k_short = 0) {
  const int ntiles = (N / C::BN) * ((M + C::BM - 1) / C::BM);
  const int cus = gemm_persistent_wgs();
  const int wgs = cus > (1 << 20) ? cus : cus * C::WG_PER_CU;
  APH_ALLOW_SMEM((gemm_ws_kernel<C, Epi>), C::SMEM_TOTAL);
  APH_LAUNCH((gemm_ws_kernel<C, Epi>), dim3(ntiles < wgs ? ntiles : wgs), dim3(C::NTHREAD), C::SMEM_TOTAL, st, A, lda, Bt, ldb, M, N, K, epi, ntiles,
             gemm_ws_panel_group(N / C::BN, (M + C::BM - 1) / C::BM), trace, n_short / C::BN, k_short / GEMM_BK);
}
// (Measured and rejected in round 4, profiles/r04_ab_gemm_flag_sync.txt: the same kernel with NO workgroup barrier in the main loop -- producers and
// consumers handing k-tile units over through counters in LDS (ready[producer] after a counted vmcnt wait, done[consumer] after lgkmcnt(0), the
// consumers asking for the counters behind their fragment reads, the producers polling with s_sleep), so that waves drift by up to the ring
// depth and a wave's epilogue sits under its SIMD partner's MFMAs.  Bit-identical results, 14-34 % SLOWER per shape (QKV 58.9 vs 43.8 us, fc2
// 55.2 vs 48.3), whole step 150.5 vs 162.8 steps/s: a stage is refilled only when the SLOWEST consumer has released it while the fastest one
// already wants the unit after next -- the skew the barrier forbids is paid for out of the 3-stage ring.  git show e138f33:aphantasia_amd/csrc/vit_gemm_wsf.h)
template <class Epi>
inline void launch_gemm_ws(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                           unsigned long long* trace, int n_short, int k_short) {
  // n_short: the LAST n_short columns (a multiple of 128) are summed over the first k_short of K only (a multiple of 64); 0 = a plain GEMM
  launch_gemm_ws_cfg<GemmWS>(A, lda, Bt, ldb, M, N, K, epi, st, trace, n_short, k_short);
}

}  // namespace aph
