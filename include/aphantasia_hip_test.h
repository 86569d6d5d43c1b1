/* libaphantasia_hip.so -- test and measurement hooks.  NOT part of the drop-in boundary (include/aphantasia_hip.h):
 * nothing a reference-side binding needs is declared here.  Used by tests/ (GEMM core alone, every tile
 * configuration) and by bench.py's roofline leg (per-launch GEMM timing on the launch stream).
 */
#ifndef APHANTASIA_HIP_TEST_H
#define APHANTASIA_HIP_TEST_H

#include "aphantasia_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per-launch HIP-event timing of the ViT's GEMM launches (bench.py roofline): on/off, then read the sums */
int aph_vit_profile(aph_vit* vit, int on);
int aph_vit_profile_read(aph_vit* vit, double* ms_total, long long* launches, double* flops);
/* The ViT's attention kernels alone (head dim 64, T <= 256): mode 0 = forward (qkv -> att, lse), mode 1 = backward
 * ((qkv, att, lse, datt) -> dqkv).  qkv / dqkv [S*T, 3*heads*64] f16 (q | k | v column blocks), att / datt [S*T, heads*64] f16,
 * lse [S*heads*T] f32 (log-sum-exp of the scores / 8); d_delta: S*heads*T floats of scratch for the backward when T > 64 (only the two-kernel
 * backward of -DAPH_EXPERIMENTS builds writes it; still required for call compatibility). */
int aph_attn_test(const void* d_qkv, void* d_att, float* d_lse, const void* d_datt, float* d_delta, void* d_dqkv, int S, int T, int heads,
                  int mode, void* stream);
/* C[M,N] f32 = A[M,K] f16 * Bt[N,K]^T f16 (N % 128 == 0, K % 64 == 0): the ViT GEMM core with the automatic tile choice */
int aph_gemm_f16(const void* d_A, const void* d_Bt, int M, int N, int K, float* d_C, void* stream);
/* same with explicit row pitches (elements, multiples of 8) and an explicit tile configuration:
 *    0  automatic (the shape heuristic of launch_gemm)
 *    1  64x64, 4 waves            2  256x128, 8 waves, 3-stage ring       4  256x256 phased (N % 256 == 0; -DAPH_EXPERIMENTS builds only)
 *    5  256x128 wave-specialised persistent (2 DMA producer waves + 8 MFMA consumer waves, register epilogue: vit_gemm_ws.h)
 *    8 / 9   64x64 split-K x2 / x4                10  128x128, 8 waves, 4-stage ring
 *   11  128x128, 4 waves, 2-stage ring, two workgroups per CU (measured slower than 2 on every ViT shape: profiles/r02_gemm_shapes.txt)
 *   12  256x128 on four waves of 128x64, one per SIMD (measured slower than 2: same file)
 *   14 / 15  64x64 register-staged split-K: each of the four waves streams its own k-tiles (global_load_dwordx4 -> private LDS image ->
 *            fragments; 4 / 3 k-steps of 32 in flight), no barrier in the main loop, ordered 4-way sum at the end (vit_gemm_rs.h: the small-M
 *            default for narrow outputs)
 *   16 / 17  64x256 A-resident: the 64 x K block of A stays in LDS, every wave streams the weight rows of its 64 columns (8 / 4 k-steps in
 *            flight); needs N % 256 == 0 and K = 256 ... 1024.  -DAPH_EXPERIMENTS builds only (measured slower than 1 / 10 in the step)
 *   22 / 24  128x128 split-K x2 / x4
 * | 0x100 (with 2, 4 or 5 only): measurement variant whose epilogue keeps the accumulators live but never stores (upper bound of
 *   what overlapping the store phase could gain: tools/exp/gemm_nostore.py).
 * any other value is rejected (APH_ERR_ARG). */
int aph_gemm_f16_ld(const void* d_A, int lda, const void* d_Bt, int ldb, int M, int N, int K, float* d_C, int tile_cfg,
                    void* stream);

/* Crop / resize adjoint of aph_sample_bwd: 1 = always the per-pixel gather kernel (round 2), 0 = automatic (the separable row-block kernel
 * on frames without wrap padding).  Process-wide, returns the previous value.  (No environment variable changes which kernels the library
 * runs: every switch here is an explicit call.) */
int aph_crop_adjoint_set_gather(int on);
/* Launch shape of the separable crop adjoint: rows per workgroup (rb), columns per thread (cpt), cuts per batch (nbc), column segments (nseg),
 * row-block order (0 = top-down, 1 = centre-out); 0 (order: -1) = automatic.  Shapes other than the shipped ones need a -DAPH_EXPERIMENTS build
 * (tools/exp/crop_adjoint_sweep.py). */
int aph_crop_adjoint_set_shape(int rb, int cpt, int nbc, int nseg, int order);

/* MFMA shape of the main loops of the GEMM TEST ENTRIES (aph_gemm_f16, aph_gemm_f16_ld; the ViT's own GEMMs are compiled for the default
 * only) launched from now on: 0 = v_mfma_f32_16x16x32_f16 (default: measured
 * faster on MI355X with real operands -- the chip is power-limited there and the 32x32x16 form sustains less),
 * 1 = v_mfma_f32_32x32x16_f16.  Returns the previous setting.  For within-process A/B measurements and the unit tests. */
int aph_gemm_set_mfma32(int on);
/* the first block's LayerNorm pairs (ln_pre + ln_1 forward, ln_1 + ln_pre backward) as one kernel each and no zero fill of the
 * fp32 gradient stream: on (1, default) / off (0).  Bit-identical either way.
 * Returns the previous value.  Captured graphs keep the setting they were recorded with. */
int aph_vit_set_fuse_ln(int on);
/* [r6] 1 = the ViT backward keeps its residual-stream gradient in f16 only (every LayerNorm backward reads the f16 copy its predecessor wrote
 * for the dgrad GEMM and writes no fp32 stream: 73 instead of 117 MB per launch at 190 cuts); 0 (default) = fp32 stream.  Measurement switch for
 * the loss-curve ensemble (tools/loss_ensemble.py); returns the previous value.  Captured graphs keep the setting they were recorded with. */
int aph_vit_set_grad_stream_f16(int on);
/* Number of 256x128 output tiles from which the shape heuristic picks the wave-specialised persistent kernel (tile_cfg 5)
 * for the ViT's own GEMMs; 0 = never.  Process-wide, returns the previous value (A/B measurements, unit tests at small sizes). */
int aph_gemm_set_ws_min_tiles(int tiles);
/* Register-staged GEMMs (tile_cfg 14 / 16) inside the ViT: 1 (default) = the split-K kernel for GEMMs of at most 128 rows over K <= 1024
 * when the WHOLE batch of the ViT call is that small (cuts x tokens <= 128: one or two cuts) -- the class-row GEMMs of a larger batch's
 * last block stay on the two-pass split-K kernels; 2 = every shape below the wave-specialised kernel's threshold (A/B measurements),
 * 0 = never (the shared-ring tile configurations 1 / 2 / 10 and their two-pass split-K).  The stand-alone entries (aph_gemm_f16 with
 * tile_cfg 0) count as small batches.  Returns the previous value. */
int aph_gemm_set_rs(int mode);
/* Tile order of the wave-specialised GEMM inside an XCD's run: groups of g row panels, column tile by column tile inside a group
 * (0 = automatic: 4 for outputs of >= 12 column tiles, else 1 = n-fastest).  Returns the previous value. */
int aph_gemm_set_ws_pgroup(int g);
/* Pure-MFMA rate probe (bench.py `roofline.peak_measured`): `blocks` workgroups of 8 waves run `iters` x 32 v_mfma_f32_16x16x32_f16 on
 * operands read once from d_src (>= 128 KiB of f16; random data sustains less than zeros: the part is power limited), nothing stored unless a
 * never-true condition holds (d_out: 512 floats).  FLOPs per launch = blocks * 8 * iters * 32 * 16384. */
int aph_mfma_rate(int blocks, int iters, const void* d_src, float* d_out, void* stream);
/* The wave-specialised GEMM (tile_cfg 5) with one of the ViT's real epilogues and optional per-tile shader-clock stamps
 * (tools/exp/gemm_ws_trace.py).  A [M,K], Bt [N,K] f16 dense; epi_kind 0: d_out f16 [M,N] = acc + bias; 1: QuickGELU, d_out = g,
 * d_out2 = dg/du (both f16); 2: d_out f32 [M,N] += acc + bias (residual in place); 3: nothing stored.
 * d_trace: (workgroups x 16 x 4) uint64 or NULL. */
int aph_gemm_ws_probe(const void* d_A, const void* d_Bt, int M, int N, int K, void* d_out, void* d_out2, const float* d_bias, int epi_kind,
                      unsigned long long* d_trace, void* stream);

/* The register-staged small-M GEMMs (tile_cfg 14 / 16) with an f16 output and per-phase stamps of the chip-wide 100 MHz clock
 * (tools/exp/gemm_rs_trace.py): kind 0 = split-K (1 = A-resident, 2 = A-resident from fragment-major weights: -DAPH_EXPERIMENTS builds);
 * d_trace: (workgroups x 8) uint64 or NULL. */
int aph_gemm_rs_probe(const void* d_A, const void* d_Bt, int M, int N, int K, void* d_out, int kind, unsigned long long* d_trace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
