// Experiment (not product): the 256x128x64 LDS-DMA GEMM mainloop with pieces switched off, to see which resource bounds it.
//   MODE 0 full | 1 no DMA after the prologue (compute + barriers only) | 2 no fragment reads / MFMA (DMA + barriers only)
//   MODE 3 MFMA only from registers (no LDS reads, no DMA in the loop)
//   MODE 4 full, DMA issue staggered: waves 4-7 (SIMD partners of 0-3) issue half a k-tile later
#include "../../aphantasia_amd/csrc/vit_gemm.h"
using namespace aph;

template <int MODE>
__global__ __launch_bounds__(512) void ablate_kernel(const half_t* __restrict__ A, int lda, const half_t* __restrict__ Bt, int ldb,
                                                     int M, int N, int K, float* __restrict__ out) {
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
  const half_t* ga[4];
  const half_t* gb[2];
  const int lrow = lane >> 3, pc = lane & 7;
  for (int k = 0; k < 4; ++k) {
    const int row = (wave * 4 + k) * 8 + lrow;
    int am = m0 + row; am = am < M ? am : M - 1;
    ga[k] = A + (size_t)am * lda + ((pc ^ ((row >> 1) & 7)) << 3);
  }
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
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / GB_BK;
  const int arow = wm * 64 + (lane & 15), brow = wn * 64 + (lane & 15), fchunk = lane >> 4;
  GbFrags f0, f1;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1) wait_vm_barrier<6>(); else wait_vm_barrier<0>();
  if (nk > 2) issue(2, 2);
  gb_load_frags(f0, lds, lds + GB_BM * GB_BK, arow, brow, fchunk);
  f1 = f0;
  int st_cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* As = lds + st_cur * GB_STAGE;
    const half_t* Bs = As + GB_BM * GB_BK;
    if (MODE != 2 && MODE != 3) gb_load_frags(f1, As, Bs, arow, brow, 4 + fchunk);
    if (MODE != 2) gb_mma(acc, f0);
    st_cur = st_cur == GB_NSTAGE - 1 ? 0 : st_cur + 1;
    if (kt + 1 < nk) {
      if (MODE != 3) {
        wait_lgkm0();
        if (MODE == 1) wait_vm_barrier<0>();
        else if (kt + 2 < nk) wait_vm_barrier<6>(); else wait_vm_barrier<0>();
        if (MODE != 1 && kt + 3 < nk && !(MODE == 4 && wave >= 4)) issue(kt + 3, st_cur == 0 ? GB_NSTAGE - 1 : st_cur - 1);
      }
      const half_t* An = lds + st_cur * GB_STAGE;
      if (MODE != 2 && MODE != 3) gb_load_frags(f0, An, An + GB_BM * GB_BK, arow, brow, fchunk);
    }
    if (MODE != 2) gb_mma(acc, f1);
    if (MODE == 4 && wave >= 4 && kt + 1 < nk && kt + 3 < nk) issue(kt + 3, st_cur == 0 ? GB_NSTAGE - 1 : st_cur - 1);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (MODE == 2) s += (float)lds[tid];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

extern "C" int ablate(int mode, const void* A, const void* Bt, int M, int N, int K, float* out, void* stream) {
  dim3 grid((N / GB_BN) * ((M + GB_BM - 1) / GB_BM)), block(512);
  hipStream_t st = (hipStream_t)stream;
#define GO(MD) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ablate_kernel<MD>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_SMEM); \
                 hipLaunchKernelGGL(ablate_kernel<MD>, grid, block, GB_SMEM, st, (const half_t*)A, K, (const half_t*)Bt, K, M, N, K, out); }
  if (mode == 0) GO(0) else if (mode == 1) GO(1) else if (mode == 2) GO(2) else if (mode == 3) GO(3) else GO(4)
  return (int)hipGetLastError();
}
