// Experiment for the small-M regime (per-rank shards of the multi-GPU path): the 64 x 64 ring GEMM of
// vit_gemm.h with an EIGHT-stage ring (128 KiB: seven k-tiles = 112 KiB of operands in flight per CU instead of three = 48 KiB).
// At M = 1200 (8 ranks) every GEMM of the step is 228 tiles of 64 x 64 -- one per CU -- whose k loop moves 16 KiB per k-tile
// from L2 and waits for it: with three tiles in flight a CU sustains about 48 KiB per L2 round trip, a third of the 110 GB/s
// per-CU DMA rate measured in tools/exp/dma_rate.  Not selected by the launch heuristic: reachable through the test hook only
// (aph_gemm_f16_ld tile_cfg 13) until it has been measured and race-tested on hardware -- the counted vmcnt waits below are
// what the host interpreter cannot check (it maps wait_vm_barrier to a plain barrier).
#pragma once
#include "vit_gemm.h"

namespace aph {

// GemmCfg<2, 2, 2, 2, 8> member for member (GemmCfg itself asserts NSTAGE <= 4: its ring_wait only counts up to two tiles ahead)
struct GemmSmallDeep {
  static constexpr int WM = 2, WN = 2, TM = 2, TN = 2, NSTAGE = 8;
  static constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  static constexpr int NWAVE = WM * WN, NTHREAD = NWAVE * 64;
  static constexpr int STAGE = (BM + BN) * GEMM_BK;
  static constexpr int GA = BM / 8 / NWAVE, GB = BN / 8 / NWAVE;
  static constexpr int GPT = GA + GB;
  static constexpr int CT_LD = TN * 16 + 4;
  static constexpr int SMEM = NSTAGE * STAGE * 2;
  static constexpr int EP_MT = TM;
  static_assert(NWAVE * EP_MT * 16 * CT_LD * 4 <= SMEM, "epilogue staging must fit in the ring");
  static_assert((NSTAGE - 2) * GPT < 64, "vmcnt is a 6-bit counter");
};

// "the DMAs of all but the newest `ahead` tiles of this wave have landed", then workgroup barrier: up to NSTAGE - 2 = 6 tiles ahead
template <>
__device__ __forceinline__ void ring_wait<GemmSmallDeep>(int ahead) {
  constexpr int G = GemmSmallDeep::GPT;
  switch (ahead < 0 ? 0 : ahead) {
    case 0: wait_vm_barrier<0>(); break;
    case 1: wait_vm_barrier<G>(); break;
    case 2: wait_vm_barrier<2 * G>(); break;
    case 3: wait_vm_barrier<3 * G>(); break;
    case 4: wait_vm_barrier<4 * G>(); break;
    case 5: wait_vm_barrier<5 * G>(); break;
    default: wait_vm_barrier<6 * G>(); break;
  }
}

}  // namespace aph
