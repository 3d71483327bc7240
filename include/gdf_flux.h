/* gdf_flux.h — C ABI of the MMDiT (Flux) hot path of libgdf.so (BASELINE config 5).
 *
 * What it replaces in the reference (paths relative to /root/reference/feature):
 *   - FluxTransformer2DModel.forward                       diffusers/models/transformers/transformer_flux.py:414-603
 *     (FluxTransformerBlock :167-226, FluxSingleTransformerBlock :86-112, FluxAttnProcessor2_0
 *      diffusers/models/attention_processor.py:2266-2362, FeedForward diffusers/models/attention.py:1249-1258)
 *     — the denoiser call inside `self.pipe(image=..., prompt=..., strength=t/1000, guidance_scale=1)`,
 *     diffusion_feature.py:246-254
 *   - the side effects of every `feature_gatherer.gather(...)` in those files (FeatureStore.store,
 *     components/feature_extractor.py:31-76) with the flux id scheme of components/feature_extractor.py:98-123:
 *       vit-block{i}-{q,k,v,cross-map,self-map,attn-out,norm-out,ffn-inner,out}   i <  num_layers   (double blocks)
 *       vit-block{i}-{q,k,v,cross-map,self-map,attn-out,out}                      i >= num_layers   (single blocks)
 *     `cross-map` (B, heads, S_img, n_txt) / `self-map` (B, heads, S_img, S_img): softmax probabilities of the image queries
 *     (FluxAttnStoreProcessor, components/attention.py:404-527), contiguous fp16; they need n_txt % 8 == 0.
 *
 * Handles (gdf_model / gdf_plan) and every model / plan query, hook-info, workspace and timing function are the ones
 * of gdf.h; only creation and the forward call differ.  Same conventions: plain C, int status (0 = ok), caller owns
 * all device buffers, asynchronous on the stream passed in.  Arithmetic (gdf_flux_desc.compute_dtype): GDF_BF16 — what the
 * reference runs (torch.bfloat16, components/models.py:158-169): bf16 weights / activations / MFMA operands
 * (mfma_f32_*_bf16), fp32 accumulate, fp32 residual stream; or GDF_F16 — same rate, 3 more mantissa bits, but fp16 RANGE:
 * activations beyond +-65504 saturate (real FLUX.1-dev checkpoints reach that; synthetic weights do not).  Hooks are fp16
 * in both, like the reference's (components/feature_extractor.py:59-60), written with saturating casts.
 */
#ifndef GDF_FLUX_H
#define GDF_FLUX_H
#include <stddef.h>
#include <stdint.h>

#include "gdf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* FluxTransformer2DModel hyper-parameters (transformer/config.json of black-forest-labs/FLUX.1-dev,
 * components/models.py:150-169; registered at transformer_flux.py:247-259). */
typedef struct gdf_flux_desc {
  int in_channels;              /* 64  (packed 2x2 latent patches x 16 channels) */
  int num_layers;               /* 19  double (MMDiT) blocks */
  int num_single_layers;        /* 38  single blocks */
  int attention_head_dim;       /* 128 (must be 128: sum of axes_dims_rope) */
  int num_attention_heads;      /* 24 */
  int joint_attention_dim;      /* 4096 (T5 width) */
  int pooled_projection_dim;    /* 768  (CLIP pooled width) */
  int guidance_embeds;          /* 1 (FLUX.1-dev), 0 (schnell) */
  int axes_dims_rope[3];        /* 16, 56, 56 */
  int mlp_ratio;                /* 4 */
  int compute_dtype;            /* GDF_F16 (0) or GDF_BF16 (2): element type of weights, activations, inputs and `out`.
                                   GDF_BF16X2 (3): weights, inputs and `out` are bf16 as with GDF_BF16, but every activation that feeds an
                                   MFMA contraction is kept as a bf16 PAIR hi + lo (16 mantissa bits, bf16's range) and multiplied as
                                   [hi | lo] x [W | W] (K doubled, weights read twice), and the attention internals (q, k after
                                   RMSNorm + RoPE, v, P) are fp16: every hook within 1e-3 of the fp32 reference at full depth
                                   (bf16: 3.4e-3) without leaving bf16's range on the residual / MLP path; about 1.7x the time.
                                   GDF_F16S (5, round 5; the Python front end's 'auto' default): GDF_F16 made range-safe where the MMDiT has no
                                   a-priori bound.  The residual streams are fp32 (all modes); LayerNorm outputs are bounded by
                                   (1 + |scale|) sqrt(C) + |shift|, q / k are RMS-normalised, v and the attention output are hooked tensors
                                   the reference itself hands out as fp16; the ONE remaining operand class — the MLP hidden tensors
                                   gelu(ff_in(.)) (and with them the single blocks' [attn | mlp] operand rows) — is stored as fp16 of
                                   x * 2^-8 (range +-1.7e7, fp16 mantissa) and the factor is undone on the consuming GEMM's fp32 accumulators.
                                   fp16 operands carry 3 more mantissa bits than bf16: every hook within 1e-3 of the fp32 reference at full
                                   depth (<= 4.6e-4 measured; bf16 3.4e-3) at the bf16 mode's speed.  Weights are bf16 checkpoint values cast
                                   to fp16 (exact for 6.1e-5 <= |w| <= 65504; the Python loader checks the cast and falls back to
                                   GDF_BF16X2 when a matrix does not survive it).
                                   GDF_FP8MX (4), OPT-IN and LOWER precision than the reference's bf16 (BASELINE.json configs[4]: "optional fp8
                                   MFMA"): as GDF_BF16, but the large linears (QKV, MLP, attention / block output projections) multiply
                                   OCP e4m3 operands with v_mfma_scale_f32_16x16x128_f8f6f4 — activations quantised per token, weights per
                                   output channel, power-of-two (e8m0) scales applied to the fp32 accumulators; the residual stream, norms,
                                   attention internals and hooks are unchanged.  Never a default; error ~3-4e-2 per hook (stated in the tests). */
} gdf_flux_desc;

int gdf_flux_model_create(const gdf_flux_desc* desc, gdf_model** out);

/* Static op program for `batch` samples of img_h x img_w image tokens (packed latent grid: 64 x 64 at 1024^2) and
 * n_txt text tokens (512).  Hook tensors are logical (B, C, img_h, img_w) stored channels-last, as in gdf.h. */
int gdf_flux_plan_create(gdf_model* m, int batch, int img_h, int img_w, int n_txt, const char* const* hook_ids,
                         int n_hooks, const gdf_plan_opts* opts, gdf_plan** out);

/* One transformer forward.  "e16" = the model's compute_dtype (fp16 or bf16).  hidden_states (B, img_h*img_w, in_channels)
 * e16; encoder_hidden_states (B, n_txt, joint_attention_dim) e16; pooled_projections (B, pooled_projection_dim) e16; timestep (B) fp32 in
 * [0,1] (the model multiplies by 1000, transformer_flux.py:472); guidance (B) fp32 or NULL when !guidance_embeds;
 * img_ids (img_h*img_w, 3) fp32, txt_ids (n_txt, 3) fp32; out (B, img_h*img_w, in_channels) e16 or NULL with
 * early_exit; hook_out buffers are fp16 in either mode. */
int gdf_flux_forward(gdf_plan* p, const void* hidden_states, const void* encoder_hidden_states,
                     const void* pooled_projections, const float* timestep, const float* guidance,
                     const float* img_ids, const float* txt_ids, void* const* hook_out, void* out, void* workspace,
                     void* stream);

/* Per-op timing (diagnostics; synchronises). Same contract as gdf_plan_profile. */
int gdf_flux_plan_profile(gdf_plan* p, const void* hidden_states, const void* encoder_hidden_states,
                          const void* pooled_projections, const float* timestep, const float* guidance,
                          const float* img_ids, const float* txt_ids, void* const* hook_out, void* out, void* workspace,
                          void* stream, float* ms, const char** names, double* flops, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GDF_FLUX_H */
