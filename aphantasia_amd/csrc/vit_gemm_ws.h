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
//     is per lane, so this is free) such that the 4 x 4 accumulator registers a lane holds for one output row are 16
//     CONSECUTIVE columns: a lane stores 32 (f16) / 64 (f32) contiguous bytes per row, the four lanes of a row one or
//     two whole 128-byte lines.  The ring is never idle, no workgroup barrier separates main loop and epilogue.
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
// one had left (entry / exit stamps on the chip-wide clock), although the occupancy API reports 2 per CU.)

// LDS row v of the Bt tile (0..127) holds tile row perm(v): within each 64-row block (one consumer column group), fragment row
// i = v & 15 of column tile nt = (v >> 4) & 3 is weight row 16 (i >> 2) + 4 nt + (i & 3).  With the MFMA operands swapped
// (weights as the A fragment) lane l, register r of acc[mt][nt] is then column 16 (l >> 4) + 4 nt + r of row l & 15.
__host__ __device__ __forceinline__ int gemm_ws_brow(int v) {
  const int vv = v & 63, nt = vv >> 4, i = vv & 15;
  return (v & ~63) + ((i >> 2) << 4) + (nt << 2) + (i & 3);
}


// ---- register transposes among the four lanes that hold one output row (lanes j, j + 16, j + 32, j + 48) -------------
// v_permlane16_swap: rows (16 lanes) 1 and 3 of `a` trade places with rows 0 and 2 of `b`; v_permlane32_swap: the upper 32 lanes
// of `a` with the lower 32 lanes of `b` (gfx950; semantics probed on hardware by tools/exp/permlane_probe.hip).
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
#ifdef APH_EMU
  const int l = emu::lane_id();
  const unsigned b_prev = __shfl(b, (l - 16) & 63), a_next = __shfl(a, (l + 16) & 63);
  if ((l >> 4) & 1) a = b_prev; else b = a_next;
#else
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
#ifdef APH_EMU
  const int l = emu::lane_id();
  const unsigned b_prev = __shfl(b, (l - 32) & 63), a_next = __shfl(a, (l + 32) & 63);
  if (l >= 32) a = b_prev; else b = a_next;
#else
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool S32, class V>
__device__ __forceinline__ void swap_regs4(V& a, V& b) {       // V: any 16-byte register quad
  u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    unsigned p = x[c], q = y[c];
    if (S32) swap32(p, q); else swap16(p, q);
    x[c] = p; y[c] = q;
  }
  a = __builtin_bit_cast(V, x); b = __builtin_bit_cast(V, y);
}
// in: lane group g (= lane >> 4) holds pieces 4g .. 4g+3 of its row's 16 four-column pieces;  out: v[q] = piece 4q + g
// (store instruction q then writes 64 contiguous bytes per row).  Its own inverse.
__device__ __forceinline__ void xpose4(f32x4 (&v)[4]) {
  swap_regs4<false>(v[0], v[1]); swap_regs4<false>(v[2], v[3]);
  swap_regs4<true>(v[0], v[2]); swap_regs4<true>(v[1], v[3]);
}
// in: lane group g holds pieces 2g, 2g+1 of its row's 8 eight-column f16 pieces;  out: p_q = piece 4q + g.  xpose2_inv undoes it.
__device__ __forceinline__ void xpose2(half8& p0, half8& p1) { swap_regs4<false>(p0, p1); swap_regs4<true>(p0, p1); }
__device__ __forceinline__ void xpose2_inv(half8& p0, half8& p1) { swap_regs4<true>(p0, p1); swap_regs4<false>(p0, p1); }

__device__ __forceinline__ half8 pack_h8(const f32x4& a, const f32x4& b) {
  return half8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
}
__device__ __forceinline__ void st_h8(half_t* p, const half8& h) { *reinterpret_cast<half8*>(p) = h; }

// ---- epilogues of the wave-specialised kernel ------------------------------------------------------------------------
// ws_row(epi, m, ok, nw, g, v, lb): v = the 16 consecutive columns nw + 16 g .. + 15 of output row m held by this lane (ok = row
// exists; every lane of the wave must call, the transposes are wave collectives); lb = this GEMM's bias vector in LDS (or null).
// Stores leave as 16-byte pieces, 64 contiguous bytes per row and instruction.
__device__ __forceinline__ void ws_add_bias16(f32x4 (&v)[4], const float* lb, int n) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += *reinterpret_cast<const f32x4*>(lb + n + 4 * i);
}
__device__ __forceinline__ void ws_store_h16(half_t* row, int nw, int g, bool ok, const f32x4 (&v)[4]) {
  half8 p0 = pack_h8(v[0], v[1]), p1 = pack_h8(v[2], v[3]);
  xpose2(p0, p1);
  if (ok) { st_h8(row + nw + 8 * g, p0); st_h8(row + nw + 32 + 8 * g, p1); }
}
__device__ __forceinline__ void ws_row(const EpiF16& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float* lb) {
  if (e.bias) ws_add_bias16(v, lb, nw + 16 * g);
  ws_store_h16(e.out + (size_t)m * e.ldo, nw, g, ok, v);
}
__device__ __forceinline__ void ws_row(const EpiF16Scale& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] *= e.scale;
  ws_store_h16(e.out + (size_t)m * e.ldo, nw, g, ok, v);
}
__device__ __forceinline__ void ws_row(const EpiF32& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
  xpose4(v);
  if (ok) {
    float* o = e.out + (size_t)m * e.ldo + nw + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + 16 * q, v[q] * e.scale);
  }
}
__device__ __forceinline__ void ws_row(const EpiNoStore& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
  const f32x4 t = v[0] + v[1] + v[2] + v[3];            // every accumulator stays live
  if (ok && t[0] + t[1] + t[2] + t[3] == 1.2345678e33f) st4(e.out + (size_t)m * e.ldo + nw + 16 * g, t);
}
__device__ __forceinline__ void ws_row(const EpiResidual& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
  // residual and bias are already in the accumulators (ws_init below): transpose and store
  xpose4(v);
  if (ok) {
    float* o = e.out + (size_t)m * e.ldo + nw + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + 16 * q, v[q]);
  }
}
__device__ __forceinline__ void ws_row(const EpiPatchEmbed& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
  xpose4(v);
  if (ok) {
    const int s = m / e.P, p = m - s * e.P;
    const float* pe = e.pos + (size_t)(1 + p) * e.D + nw + 4 * g;
    float* o = e.x0 + ((size_t)s * e.T + 1 + p) * e.D + nw + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + 16 * q, v[q] + ld4(pe + 16 * q));
  }
}
__device__ __forceinline__ void ws_row(const EpiGelu& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float* lb) {
  ws_add_bias16(v, lb, nw + 16 * g);
  f32x4 gl[4], dg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) quick_gelu4(v[i], gl[i], dg[i]);
  ws_store_h16(e.g + (size_t)m * e.ldo, nw, g, ok, gl);
  ws_store_h16(e.dg + (size_t)m * e.ldo, nw, g, ok, dg);
}
__device__ __forceinline__ void ws_row(const EpiGeluBwd& e, int m, bool ok, int nw, int g, f32x4 (&v)[4], const float*) {
  // dg/du is read in the store pattern (64 contiguous bytes per row and instruction) and moved back to the accumulators' columns
  half8 d0 = {}, d1 = {};
  if (ok) {
    const half_t* d = e.dg + (size_t)m * e.ldo + nw + 8 * g;
    d0 = *reinterpret_cast<const half8*>(d);
    d1 = *reinterpret_cast<const half8*>(d + 32);
  }
  xpose2_inv(d0, d1);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[0][i] *= (float)d0[i]; v[1][i] *= (float)d0[4 + i]; v[2][i] *= (float)d1[i]; v[3][i] *= (float)d1[4 + i]; }
  ws_store_h16(e.out + (size_t)m * e.ldo, nw, g, ok, v);
}
// ws_init: the accumulators' starting value for output row m (same arguments as ws_row).  The residual epilogue starts them from
// res + bias, read in the store pattern (64 contiguous bytes per row and instruction) and transposed into the MFMA layout: the
// loads' latency hides behind the tile's first DMA wait instead of sitting between main loop and stores.
template <class E>
__device__ __forceinline__ void ws_init(const E&, int, int, int, int, f32x4 (&v)[4], const float*) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void ws_init(const EpiResidual& e, int m, int M, int nw, int g, f32x4 (&v)[4], const float* lb) {
  const float* r = e.res + (size_t)(m < M ? m : M - 1) * e.ldo + nw + 4 * g;       // (rows past M: clamped read, never stored)
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = ld4(r + 16 * q) + *reinterpret_cast<const f32x4*>(lb + nw + 4 * g + 16 * q);
  xpose4(v);
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
                                                                  int ldb, int M, int N, int K, Epi epi, int ntiles, unsigned long long* __restrict__ trace) {
  using F = GemmBig;                             // fragment shapes of a consumer wave: 4 x 4 MFMA tiles (64 x 64)
  APH_DYN_SMEM(smem);
  half_t* lds = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  // persistent tile run (as gemm8_f16_kernel): XCD x owns one contiguous run of tiles, n-tiles fastest; its workgroups walk it
  int tile, tile_end, tile_step;
  {
    const int nwg = gridDim.x, b = blockIdx.x, G = nwg < 8 ? nwg : 8;
    const int q = ntiles / G, r = ntiles - q * G, xcd = b % G;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile_end = start + q + (xcd < r ? 1 : 0);
    tile_step = (nwg - xcd + G - 1) / G;
    tile = start + b / G;
  }
  if (tile >= tile_end) return;                  // (workgroup-uniform; does not happen with gridDim.x <= ntiles)
  if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + 15) * 4 + 3] = ws_realtime();   // kernel entry, chip-wide 100 MHz clock
  const int nk = K / GEMM_BK, ntn = N / C::BN;
  const int U = ((tile_end - tile + tile_step - 1) / tile_step) * nk;       // k-tile units of this workgroup
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
  int kt = 0, st = 0;
  wait_vm_barrier<63>();                                                       // unit 0 has landed (and the bias vector is in LDS)
  auto init_tile = [&](int t) {
    const int tm = t / ntn;
    const int mrow = tm * C::BM + wm * 64 + (lane & 15), nw = (t - tm * ntn) * C::BN + wn * 64;
#pragma unroll
    for (int mt = 0; mt < F::TM; ++mt) ws_init(epi, mrow + mt * 16, M, nw, lane >> 4, acc[mt], lbias);
  };
  init_tile(tile);
  gemm_load_frags<F>(f0, lds, lds + C::BM * GEMM_BK, arow, brow, fchunk);
  for (int u = 0; u < U; ++u) {
    const half_t* As = lds + st * C::STAGE;
    st = st == C::NSTAGE - 1 ? 0 : st + 1;
    gemm_load_frags<F>(f1, As, As + C::BM * GEMM_BK, arow, brow, 4 + fchunk);  // k-step 1 of unit u: in flight during the MFMAs
    gemm_mma<F>(acc, f0);                                                      // k-step 0 of unit u
    if (u + 1 < U) {
      wait_lgkm0();                                                            // f1 has left LDS: the stage of unit u is dead for this wave
      wait_vm_barrier<63>();                                                   // (vmcnt(63): this wave's stores are never waited for here)
      const half_t* An = lds + st * C::STAGE;
      gemm_load_frags<F>(f0, An, An + C::BM * GEMM_BK, arow, brow, fchunk);    // k-step 0 of unit u+1: overlaps the MFMAs below
    }
    gemm_mma<F>(acc, f1);                                                      // k-step 1 of unit u
    if (trace && kt == 0 && tid == 0) trace[((size_t)blockIdx.x * 16 + (u / nk)) * 4 + 0] = ws_clock();      // first k-tile of a tile done
    if (++kt == nk) {
      if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + (u / nk)) * 4 + 1] = ws_clock();                // main loop done
      // epilogue straight from the accumulators: row l & 15 of each 16-row tile, 16 consecutive columns at 16 (l >> 4)
      const int tm = tile / ntn;
      const int mrow = tm * C::BM + wm * 64 + (lane & 15), nw = (tile - tm * ntn) * C::BN + wn * 64;
#pragma unroll
      for (int mt = 0; mt < F::TM; ++mt) {
        const int m = mrow + mt * 16;
        ws_row(epi, m, m < M, nw, lane >> 4, acc[mt], lbias);
      }
      if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + (u / nk)) * 4 + 2] = ws_clock();                // epilogue issued
      kt = 0;
      tile += tile_step;
      if (u + 1 < U) init_tile(tile);
    }
  }
  if (trace && tid == 0) trace[((size_t)blockIdx.x * 16 + 14) * 4 + 3] = ws_realtime();     // consumer wave 0 done
}

template <class C, class Epi>
inline void launch_gemm_ws_cfg(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                               unsigned long long* trace) {
  const int ntiles = (N / C::BN) * ((M + C::BM - 1) / C::BM);
  const int cus = gemm8_persistent_wgs();
  const int wgs = cus > (1 << 20) ? cus : cus * C::WG_PER_CU;
  APH_ALLOW_SMEM((gemm_ws_kernel<C, Epi>), C::SMEM_TOTAL);
#ifndef APH_EMU
  if (getenv("APH_WS_OCC")) {       // (experiment aid) what the runtime says about residency
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_ws_kernel<C, Epi>, C::NTHREAD, C::SMEM_TOTAL);
    fprintf(stderr, "gemm_ws_kernel<BM %d>: occupancy API says %d workgroup(s) per CU (threads %d, LDS %d)\n", C::BM, nb, C::NTHREAD, C::SMEM_TOTAL);
  }
#endif
  APH_LAUNCH((gemm_ws_kernel<C, Epi>), dim3(ntiles < wgs ? ntiles : wgs), dim3(C::NTHREAD), C::SMEM_TOTAL, st, A, lda, Bt, ldb, M, N, K, epi, ntiles, trace);
}
template <class Epi>
inline void launch_gemm_ws(const half_t* A, int lda, const half_t* Bt, int ldb, int M, int N, int K, Epi epi, hipStream_t st,
                           unsigned long long* trace) {
  launch_gemm_ws_cfg<GemmWS>(A, lda, Bt, ldb, M, N, K, epi, st, trace);
}

}  // namespace aph
