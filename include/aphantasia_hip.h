/* libaphantasia_hip.so -- C ABI of the MI355X (gfx950) hot path of eps696/aphantasia.
 *
 * The reference has no FFI layer of its own: its per-step loop (clip_fft.py:235-306) calls
 * Python functions (aphantasia/image.py, aphantasia/utils.py, aphantasia/transforms.py,
 * openai/CLIP's model.encode_image, torch.optim.Adam).  Each entry point below replaces one of
 * those call sites; the ctypes stubs a maintainer would add are in INTEGRATION.md.
 *
 * Conventions: plain C types; every pointer named d_* / documented "device" is a HIP device
 * pointer (tensor.data_ptr()); `stream` is a hipStream_t passed as void* (0 = default stream);
 * all calls are asynchronous on that stream and never allocate per call; return value 0 = ok,
 * < 0 = error (message through aph_last_error(), thread-local).
 */
#ifndef APHANTASIA_HIP_H
#define APHANTASIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APH_OK 0
#define APH_ERR_ARG (-1)
#define APH_ERR_HIP (-2)
#define APH_ERR_UNSUPPORTED (-3)
#define APH_ERR_INTERNAL (-4)
#define APH_ERR_COMM (-5)      /* RCCL error */

int aph_version(void);
const char* aph_last_error(void);

/* ---- image parameteriser: aphantasia/image.py ------------------------------------------- */
typedef struct aph_synth_plan aph_synth_plan;

/* twiddles + workspace for a C x H x W image (fft_image closure state, image.py:152-162) */
int aph_synth_plan_create(int C, int H, int W, aph_synth_plan** out);
int aph_synth_plan_destroy(aph_synth_plan* plan);

/* to_valid_rgb(fft_image(...))(shift, contrast): image.py:164-175 + :21-28.
 * params [C,H,W/2+1,2] f32, scale [H,W/2+1] f32, shift [H,W/2+1] f32 or NULL, colcorr_t9 = 3x3
 * row-major colcorr_t (image.py:19) host pointer or NULL (identity).  Writes raw [C,H,W] (the
 * irfft2 output before std-normalisation) and rgb [C,H,W] in (0,1). */
int aph_synth_fft_fwd(aph_synth_plan* plan, const float* d_params, const float* d_scale, const float* d_shift,
                      float contrast, const float* colcorr_t9, int decorrelate, float* d_raw, float* d_rgb, void* stream);
/* adjoint: d_rgb_grad [C,H,W] (multiplied by gscale) -> d_grad_params [C,H,W/2+1,2].  Must follow the
 * aph_synth_fft_fwd that produced rgb/raw (the plan keeps mean/std). */
int aph_synth_fft_bwd(aph_synth_plan* plan, const float* d_rgb_grad, float gscale, const float* d_rgb, const float* d_raw,
                      const float* d_scale, float contrast, const float* colcorr_t9, int decorrelate,
                      float* d_grad_params, void* stream);
/* The two transforms alone, torch.fft semantics with norm='ortho' on the plan's geometry -- illustrip.py:401-409, the
 * per-frame `irfftn -> frame_transform -> rfftn` round trip of `--gen FFT`:
 *   aph_irfft2: d_spectrum [C,H,W/2+1,2] -> d_image [C,H,W]   == torch.fft.irfftn(view_as_complex(x), s=(H,W), norm='ortho')
 *   aph_rfft2:  d_image [C,H,W] -> d_spectrum [C,H,W/2+1,2]   == view_as_real(torch.fft.rfftn(x, s=(H,W), dim=[2,3], norm='ortho')) */
int aph_irfft2(aph_synth_plan* plan, const float* d_spectrum, float* d_image, void* stream);
int aph_rfft2(aph_synth_plan* plan, const float* d_image, float* d_spectrum, void* stream);
/* to_valid_rgb(pixel_image(...)) (image.py:114-118) and the post-inverse-DWT part of dwt_image (image.py:68):
 * raw -> raw*contrast/std (or /fixed_div when fixed_div > 0) -> colour mix -> sigmoid */
int aph_synth_spatial_fwd(aph_synth_plan* plan, const float* d_raw, float contrast, float fixed_div,
                          const float* colcorr_t9, int decorrelate, float* d_rgb, void* stream);
int aph_synth_spatial_bwd(aph_synth_plan* plan, const float* d_rgb_grad, float gscale, const float* d_rgb,
                          const float* d_raw, float contrast, float fixed_div, const float* colcorr_t9, int decorrelate,
                          float* d_raw_grad, void* stream);
/* {mean, std} of raw from the last forward -> d_out2 (device, 2 floats) */
int aph_synth_stats(aph_synth_plan* plan, float* d_out2, void* stream);
/* restore them (d_in2 device, 2 floats) before an adjoint whose forward was followed by other forwards */
int aph_synth_set_stats(aph_synth_plan* plan, const float* d_in2, void* stream);

/* illustrip.py:438-440 (`--gen RGB` priors): mean_c |mean_hw(rgb_c) - t_mean| + mean_c |std_hw(rgb_c) - t_std| (unbiased
 * std) over d_rgb [3,H,W].  Adds weight * value to *d_loss (nullable) and weight * gradient into d_rgb_grad (nullable,
 * accumulated).  d_ws: device scratch of aph_rgb_priors_ws_bytes() bytes. */
size_t aph_rgb_priors_ws_bytes(void);
int aph_rgb_priors(const float* d_rgb, int H, int W, float t_mean, float t_std, float weight, void* d_ws, float* d_loss,
                   float* d_rgb_grad, void* stream);

/* clip_fft.py:269-270 (`--sharp`): weight * derivat(rgb, 'naiv') (utils.py:265-268), value into *d_loss and gradient into
 * d_rgb_grad (both nullable, accumulated); the reference subtracts, so pass weight = -sharp.  d_ws as for aph_rgb_priors. */
int aph_rgb_sharp(const float* d_rgb, int H, int W, float weight, void* d_ws, float* d_loss, float* d_rgb_grad, void* stream);

/* ---- wavelet parameteriser: aphantasia/image.py:33-80 (dwt_image) over pytorch_wavelets.DWTInverse ------- */
/* One synthesis level (lowlevel.SFB2D, mode 'symmetric').  d_ll [C,ll_h,ll_w] running low band (ll_h in {h,h+1}:
 * the extra row/col DWTInverse.forward drops is ignored), d_highs [C,3,h,w] = (LH,HL,HH) of this level,
 * d_g0/d_g1 = rec_lo/rec_hi taps (device, L even <= 64), hscale = dwt_scale gain (image.py:73-80)
 * -> d_out [C, 2h-L+2, 2w-L+2]. */
int aph_idwt_level_fwd(const float* d_ll, int ll_h, int ll_w, const float* d_highs, int h, int w, int C,
                       const float* d_g0, const float* d_g1, int L, float hscale, float* d_out, void* stream);
/* adjoint: d_out_grad -> d_ll_grad [C,ll_h,ll_w] (dropped row/col = 0) and d_highs_grad [C,3,h,w] */
int aph_idwt_level_bwd(const float* d_out_grad, int h, int w, int C, const float* d_g0, const float* d_g1, int L,
                       float hscale, float* d_ll_grad, int ll_h, int ll_w, float* d_highs_grad, void* stream);

/* Every level of DWTInverse in one call (what dwt_image.inner runs per step, image.py:36-38,67).  Host arrays of J entries,
 * level 0 = finest: hs / ws = size of the level's detail bands, hscales = dwt_scale gains (image.py:73-80), d_highs[j]
 * [C,3,hs[j],ws[j]] (device pointers), d_bufs[j] [C, 2 hs[j]-L+2, 2 ws[j]-L+2] = caller-owned scratch for the running low band
 * after level j (d_bufs[0] receives the image; the others are written only where a level's output travels through global
 * memory).  d_yl [C, hs[J-1], ws[J-1]] = the coarsest low band.  The coarse levels -- output up to 80 x 128 -- run in one launch
 * with the running low band kept in LDS (filter lengths 2-8; APH_IDWT_COARSE=0: one launch per level throughout). */
int aph_idwt_fwd(const float* d_yl, const float* const* d_highs, const int* hs, const int* ws, const float* hscales, int J,
                 int C, const float* d_g0, const float* d_g1, int L, float* const* d_bufs, void* stream);
/* adjoint: d_img_grad [C, 2 hs[0]-L+2, 2 ws[0]-L+2] -> d_highs_grad[j] [C,3,hs[j],ws[j]] and d_yl_grad [C,hs[J-1],ws[J-1]];
 * d_gbufs[j] (j >= 1) = scratch of d_bufs[j]'s size (d_gbufs[0] is not used). */
int aph_idwt_bwd(const float* d_img_grad, const int* hs, const int* ws, const float* hscales, int J, int C,
                 const float* d_g0, const float* d_g1, int L, float* const* d_gbufs, float* d_yl_grad,
                 float* const* d_highs_grad, void* stream);

/* ---- sampler: aphantasia/utils.py:218-254 slice_imgs + transforms.py:102-109,165-170 ------- */
/* Geometry of one slice_imgs call.  (Hp,Wp,py0,px0) describe the wrap-tiled overscan frame of
 * pad_up_to/tile_pad (utils.py:152-187); Hp=H, Wp=W, py0=px0=0 when align has no 'over'. */
typedef struct aph_sample_geom {
  int H, W;          /* source image */
  int Hp, Wp;        /* padded frame the crop table refers to */
  int py0, px0;      /* top / left padding */
  int S;             /* number of cuts */
  int size;          /* output side (224) */
  int patch;         /* ViT patch size for the patch-major layouts (32 / 16) */
} aph_sample_geom;

#define APH_OUT_NCHW_RAW 0   /* f32 [S,3,size,size], no normalisation (transform=None) */
#define APH_OUT_NCHW_NORM 1  /* f32 [S,3,size,size], CLIP mean/std normalised (transforms.normalize) */
#define APH_OUT_PATCH_F16 2  /* f16 [S*(size/patch)^2, 3*patch*patch] normalised = patch-embed GEMM operand; row = patch (s, gy, gx), column
                              * k = (iy*patch + ix)*3 + c: pixel-major inside the patch, channel fastest (aph_vit_set_weight permutes conv1.weight to match) */
#define APH_OUT_PATCH_F16_HILO 4 /* aph_sample_fwd only: APH_OUT_PATCH_F16 rows written as [hi (3*patch^2) | lo (3*patch^2)], hi = f16(x), lo = f16(x - hi):
                              * the operand of the split-precision forward aph_vit_forward_hilo */
#define APH_GRAD_PATCH_F16 3 /* aph_sample_bwd only: gradient in the patch-major layout stored as f16 (aph_vit_backward_h) */

#define APH_AUG_STRIDE 16
/* per-cut augment row (f32 x 16): [0..7] perspective coeffs, [8] has_perspective, [9..12] erase i,j,h,w
 * (h=0: none), [13] cos(angle), [14] sin(angle), [15] has_rotation (transforms_fast always 1) */

/* bytes of the caller-owned device workspace d_ws of one forward/backward pair on geometry g (per-cut tap tables of the
 * crop adjoint, per-XCD work lists of the forward; with_aug != 0 adds the cut scratch of the geometric augmentations).  The library never allocates in a
 * launch path, so the calls can be captured into a hipGraph that stays valid as long as the caller's buffers do. */
size_t aph_sample_ws_bytes(const aph_sample_geom* g, int with_aug);
/* d_rgb [3,H,W] f32; d_table int32 [S,3] rows (csize, offx, offy) (utils.py:245-247);
 * d_aug f32 [S,16] or NULL (no geometric augmentation); d_ws: see aph_sample_ws_bytes (required).
 * out layout per out_mode. */
int aph_sample_fwd(const aph_sample_geom* g, const float* d_rgb, const int32_t* d_table, const float* d_aug,
                   void* d_ws, void* d_out, int out_mode, void* stream);
/* adjoint.  d_out_grad: f32 in the layout of out_mode (patch-major f32 for APH_OUT_PATCH_F16; patch-major f16 for
 * APH_GRAD_PATCH_F16), multiplied by gscale.  d_ws as above (required, overwritten).
 * Writes (does not accumulate) d_rgb_grad [3,H,W]. */
int aph_sample_bwd(const aph_sample_geom* g, const void* d_out_grad, float gscale, const int32_t* d_table,
                   const float* d_aug, void* d_ws, float* d_rgb_grad, int out_mode, void* stream);
/* illustrip.py:130-138 frame_transform = T.functional.affine(img, angle, shift, scale, shear, fill=0, BILINEAR) of a whole
 * [C,H,W] image (once per frame).  h_inv_matrix6: HOST pointer, row-major 2x3 inverse affine matrix (torchvision's
 * _get_inverse_affine_matrix with the image centre as origin).  d_dst must not alias d_src. */
int aph_frame_affine(const float* d_src, int C, int H, int W, const float* h_inv_matrix6, float* d_dst, void* stream);
/* ---- depth warp of the illustrip frame loop (/root/reference/depth/depth.py:41-84, illustrip.py:115-128) ----
 * Everything but the depth ESTIMATOR (Depth-Anything-V2, depth.py:20-32, a third-party network): the caller runs it (or
 * anything else) between aph_resize_bicubic and aph_flip_w and hands the depth map to aph_grid_warp.  All images f32,
 * [C,H,W] contiguous; destinations must not alias sources.
 *   aph_triangle_blur: d_dst = torch.lerp(src, triangle_blur(src, kernel_size, power), mix)   (utils.py:137-147; depth.py:75
 *                      uses kernel_size 5, power 2, mix 0.5; mix = 1 is the plain blur).  kernel_size odd, <= 9.
 *   aph_resize_bicubic: F.interpolate(src [C,h,w], (H,W), mode='bicubic', align_corners=True)          (depth.py:41-42)
 *   aph_flip_w: d_dst = flip(src, [-1]), times d_mul elementwise when d_mul != NULL                     (depth.py:77)
 *   aph_grid_warp: depth.py:44-66 -- two bilinear, reflection-padded, align_corners=True grid samples: the depth push
 *                  grid + (centre - grid) * (depth - max(depth) * midpoint) * strength, then the lens stretch
 *                  grid + (centre - grid) * |centre - grid| * strength * dlens.  d_depth [H,W]; centre in [-1,1] (x, y);
 *                  d_ws: C*H*W + 256 floats of scratch (intermediate image + the depth maximum and its partials). */
int aph_triangle_blur(const float* d_src, int C, int H, int W, int kernel_size, float power, float mix, float* d_dst, void* stream);
int aph_resize_bicubic(const float* d_src, int C, int h, int w, float* d_dst, int H, int W, void* stream);
int aph_flip_w(const float* d_src, const float* d_mul, int C, int H, int W, float* d_dst, void* stream);
int aph_grid_warp(const float* d_img, const float* d_depth, int C, int H, int W, float strength, float centre_x, float centre_y,
                  float midpoint, float dlens, float* d_ws, float* d_out, void* stream);
/* NCHW f32 [S,3,R,R] <-> patch-major (entry of model.encode_image for a caller-made batch) */
int aph_patchify_f16(const float* d_nchw, int S, int R, int patch, void* d_patches_f16, void* stream);
/* the same into the [hi | lo] rows of APH_OUT_PATCH_F16_HILO: d_patches_hilo f16 [S*(R/patch)^2, 2 * 3*patch^2] */
int aph_patchify_f16_hilo(const float* d_nchw, int S, int R, int patch, void* d_patches_hilo, void* stream);
int aph_unpatchify_f32(const float* d_patch_grad, int S, int R, int patch, float gscale, float* d_nchw_grad, void* stream);

/* ---- CLIP ViT visual tower: model.encode_image (clip_fft.py:254) + its input-gradient ----- */
typedef struct aph_vit aph_vit;
/* mirrors clip.model.VisionTransformer(input_resolution, patch_size, width, layers, heads, output_dim);
 * max_batch = largest number of cuts per call (activation arena is sized once, never re-allocated) */
int aph_vit_create(int input_resolution, int patch_size, int width, int layers, int heads, int output_dim,
                   int max_batch, aph_vit** out);
int aph_vit_destroy(aph_vit* vit);
size_t aph_vit_workspace_bytes(const aph_vit* vit);
/* one tensor by its OpenAI checkpoint key without the "visual." prefix (fp32 HOST data) */
int aph_vit_set_weight(aph_vit* vit, const char* name, const float* h_data, size_t count);
/* d_patches f16 [S*P, 3*patch^2] (APH_OUT_PATCH_F16 layout) -> d_enc f32 [S, output_dim] */
int aph_vit_forward(aph_vit* vit, const void* d_patches, int S, float* d_enc, void* stream);
/* SPLIT-PRECISION forward (opt-in: clip_fft.py --precise, bench.py --split; the default is aph_vit_forward -- f16 operands everywhere, the
 * reference's own GPU dtype -- since round 6): d_patches_hilo f16 [S*P, 2 * 3*patch^2] (APH_OUT_PATCH_F16_HILO rows [hi | lo]).  The
 * patch-embedding GEMM and every block's QKV GEMM take their activation operand as a hi + lo pair of f16 values (~22 bits; the GEMMs run over
 * twice the K against the weights repeated along K): the two f16 roundings that dominate the input-gradient error on weights with realistic
 * dynamic range (profiles/r04_precision_attribution.txt; single-step input-gradient error 7.7e-4 against 1.10e-3).  At full batch the lo half
 * feeds the Q and K columns of the QKV GEMM only -- the V column tiles run over the hi half, in the same launch (DESIGN.md section 4
 * *Precision* [r5]).  Same outputs / saved activations as aph_vit_forward; the backward (aph_vit_backward) is unchanged.  Costs ~5 % of a C2
 * step; over the 30-member stress-weight loss-curve ensemble its free-running curves are not significantly closer to the fp32 CPU reference
 * than the default's (profiles/r06_precision_ensemble.txt), which is why it is no longer the default.  Needs aph_vit_enable_hilo once. */
int aph_vit_forward_hilo(aph_vit* vit, const void* d_patches_hilo, int S, float* d_enc, void* stream);
/* [r6] Allocates and fills the K-repeated copies of the patch-embedding / QKV weights aph_vit_forward_hilo needs (85 MB at ViT-B/32; a handle
 * does not carry them by default).  Once per handle, after aph_vit_set_weight of every tensor, outside any stream capture; idempotent. */
int aph_vit_enable_hilo(aph_vit* vit);
/* d_genc f32 [S, output_dim] (times the caller's loss scale) -> d_patch_grad f32 [S*P, 3*patch^2] times out_scale */
int aph_vit_backward(aph_vit* vit, const float* d_genc, int S, float* d_patch_grad, float out_scale, void* stream);
/* same with the patch gradient stored as f16 (keep the loss scale in it: out_scale = 1, and undo it in aph_sample_bwd's
 * gscale with out_mode APH_GRAD_PATCH_F16): halves the bytes the sampler adjoint gathers */
int aph_vit_backward_h(aph_vit* vit, const float* d_genc, int S, void* d_patch_grad_f16, float out_scale, void* stream);
/* (test / measurement hooks -- GEMM core alone, per-launch GEMM timing -- are declared in aphantasia_hip_test.h) */

/* ---- loss: aphantasia/utils.py:270-295 sim_func, assembled as at clip_fft.py:257-267 -------- */
#define APH_SIM_COS 0   /* type None / 'cossim' */
#define APH_SIM_MIX 1   /* 'mix' (default) */
#define APH_SIM_ANG 2   /* 'ang' */
#define APH_SIM_DOT 3   /* 'dot' */
/* loss = sum_t coef_t * sim_func(target_t, enc, type), value -> d_loss (1 float) and
 * gscale * dloss/denc -> d_genc [S,D].  d_targets = n_broadcast rows [D] (one embedding vs every cut: text
 * prompts) followed by T-n_broadcast blocks [s_total,D] of per-cut targets (reference-image term,
 * clip_fft.py:216,267; s_offset = first global cut of this rank's shard), d_coef [T] (sign*weight), h_coef = host copy
 * (needed for 'ang'), denom = sample count of the global mean (S, or the all-rank total),
 * d_ws = f32 scratch of S*(T+2) elements. */
int aph_sim_loss(const float* d_enc, int S, int D, const float* d_targets, const float* d_coef, const float* h_coef,
                 int T, int n_broadcast, int s_total, int s_offset, int type, float denom, float gscale, float* d_ws,
                 float* d_loss, float* d_genc, void* stream);

/* the aesthetic predictor term, clip_fft.py:255-256 / utils.py:402-413 (`loss -= 0.001 * a.aest * aest(out_enc).mean()`,
 * aest = nn.Linear(D, 1)): adds coef * mean_s(w . enc_s + bias) to *d_loss and its gradient (times gscale) into d_genc
 * (nullable), both ACCUMULATED on top of what aph_sim_loss wrote; denom as there.  d_w [D] device, bias a host scalar;
 * pass coef = -0.001 * aest. */
int aph_linear_head(const float* d_enc, int S, int D, const float* d_w, float bias, float coef, float denom, float gscale,
                    float* d_loss, float* d_genc, void* stream);

/* d_y[i] += alpha * d_x[i] -- partial results summed inside a step (--enforce, clip_fft.py:271-275: two cut sets contribute to
 * one image gradient and one loss) */
int aph_axpy_f32(float* d_y, const float* d_x, float alpha, size_t n, void* stream);
/* the per-step frame (clip_fft.py:297-306 / utils.py:94-100 checkout): d_rgb f32 [3,H,W] in [0,1] -> d_out_u8 uint8 [H,W,3] =
 * clip(rgb ** gamma * 255, 0, 255) truncated; one launch on `stream`, the caller copies it to the host asynchronously */
int aph_rgb_to_u8(const float* d_rgb, int H, int W, float gamma, void* d_out_u8, void* stream);

/* ---- optimiser: torch.optim.Adam / AdamW as configured at clip_fft.py:108-115 --------------- */
/* d_hyper: 8 device floats {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, sqrt(1-beta2^t), grad_scale};
 * d_m may be NULL when beta1 == 0, d_vmax NULL unless amsgrad; decoupled_wd = AdamW */
int aph_adam_step(float* d_p, const float* d_g, float* d_m, float* d_v, float* d_vmax, const float* d_hyper,
                  int decoupled_wd, size_t n, void* stream);
/* the same update behind an overflow guard: a gradient with a NaN / inf element (fp16 overflow in the loss-scaled backward)
 * skips the step and increments d_guard[0] (running count); d_guard[1] is scratch.  d_guard: 2 ints zeroed once by the
 * caller, who lowers its loss scale when the count moves. */
int aph_adam_step_guarded(float* d_params, const float* d_grad, float* d_m, float* d_v, float* d_vmax, const float* d_hyper8,
                          int decoupled_wd, size_t n, int* d_guard2, void* stream);

/* ---- multi-GPU: the one collective of the step (the reference is single-GPU) ------------
 * Samples are split over the ranks of one node (one process per GPU); every rank forms its partial spectrum gradient
 * and ONE all-reduce (sum, f32) per step makes it the global gradient before the (replicated) Adam update.  RCCL is
 * called directly and bound at run time; torch.distributed is not on the data path.
 *   rank 0: aph_comm_unique_id(h_uid128) -> ship the 128 bytes to the other ranks (file, env, socket: the caller's choice);
 *   every rank, with its GPU current: aph_comm_init(rank, nranks, h_uid128, &comm);
 *   per step: aph_allreduce_f32(comm, d_buf, n, stream) -- in place, asynchronous on `stream`, capturable into a hipGraph. */
typedef struct aph_comm aph_comm;
int aph_comm_unique_id(void* h_uid128);
int aph_comm_init(int rank, int nranks, const void* h_uid128, aph_comm** out);
int aph_allreduce_f32(aph_comm* comm, float* d_buf, size_t n, void* stream);
/* the number of ranks the communicator itself reports (ncclCommCount) */
int aph_comm_ranks(aph_comm* comm, int* nranks);
int aph_comm_destroy(aph_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
