/* libaphantasia_hip.so -- hooks of code paths that were MEASURED AND NOT ADOPTED (csrc/vit_block.h, the A-resident GEMM of
 * csrc/vit_gemm_rs.h).  They exist only in a library built with -DAPH_EXPERIMENTS (`python -m aphantasia_amd._build --experiments`, and the
 * CPU interpreter build of tests/emu, which keeps the code covered); the product library does not export them and no product path calls them.
 */
#ifndef APHANTASIA_HIP_EXPERIMENTS_H
#define APHANTASIA_HIP_EXPERIMENTS_H

#include "aphantasia_hip_test.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Largest batch, in token rows (cuts x tokens per cut), whose forward runs the fused block kernels (LayerNorm inside the QKV / fc1
 * launches, attention behind the QKV GEMM: csrc/vit_block.h; sequences of at most 64 tokens only); 0 = never.  Returns the previous value. */
int aph_vit_set_fused_max_rows(int rows);
/* Inside the fused forward: the (cut, head) LayerNorm + QKV + attention kernel 0 = never (LayerNorm + QKV on the flat-row kernel, attention
 * as its own launch), 1 = while cuts x heads workgroups fit the chip in one round (default), 2 = always.  Returns the previous value. */
int aph_vit_set_fused_attn(int mode);
int aph_gemm_pack_frag(const void* d_Bt, int N, int K, void* d_out, void* stream);      /* experiment: fragment-major weights (probe kind 2) */
/* Attention backward for sequences of 65 ... 256 tokens (ViT-B/16): 1 = one kernel that forms the probabilities and dS once and hands dS to
 * the dQ contraction through LDS (default), 0 = the dQ kernel + dK/dV kernel pair (each recomputes them).  Returns the previous value. */
int aph_attn_set_bwd_one(int on);
/* [r6] measurement variants of the T <= 56 attention backward (WRONG results; tools/exp/attn_ablate.py): 0 = the kernel, 1 = no products (zeros
 * stored: loads + staging + stores), 2 = no stores (loads + staging + products), 3 = loads + staging only.  Returns the previous value. */
int aph_attn_set_ablate(int mode);

#ifdef __cplusplus
}
#endif
#endif
