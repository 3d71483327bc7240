/* gdf_pixart.h — C ABI of the PixArt (alpha / sigma) DiT hot path of libgdf.so ("PixArt-DiT" in
 * BASELINE.json's north_star).
 *
 * What it replaces in the reference (paths relative to /root/reference/feature):
 *   `noise_pred = self.pipe.transformer(latent_model_input, encoder_hidden_states=prompt_embeds,
 *        encoder_attention_mask=prompt_attention_mask, timestep=t, return_dict=False,
 *        added_cond_kwargs={'resolution': None, 'aspect_ratio': None})[0]`                diffusion_feature.py:466-474
 *   == Transformer2DModel.forward, patched inputs + ada_norm_single   diffusers/models/transformers/transformer_2d.py:404-475,
 *      :496-516, :540-575; BasicTransformerBlock.forward (ada_norm_single)  diffusers/models/attention.py:498-592;
 *      Attention + AttnProcessor2_0  diffusers/models/attention_processor.py:3244-3331; FeedForward  attention.py:1249-1258
 *   and every `gather` side effect with the DiT id scheme of components/feature_extractor.py:250-286:
 *      vit-block{i}-{self-q, self-k, self-v, self-map, cross-q, cross-map, ffn-inner, out}
 *      (cross-k / cross-v are dropped by the store, :38-39; self-map (B, heads, S, S) / cross-map (B, heads, S, n_txt) are the
 *      softmax probabilities of AttnStoreProcessor, components/attention.py, masked caption keys hold exact zeros)
 *
 * Handles, model / plan queries, hook info, workspace and timing functions are those of gdf.h.  Parameter names are the
 * `Transformer2DModel.state_dict()` names ("pos_embed.proj.*", "adaln_single.*", "caption_projection.*",
 * "transformer_blocks.N.*", "scale_shift_table", "proj_out.*"); the sincos positional table is computed, not loaded.
 * use_additional_conditions (PixArt-alpha-1024 micro-conditioning) is not supported: the reference itself calls with
 * resolution = aspect_ratio = None.
 */
#ifndef GDF_PIXART_H
#define GDF_PIXART_H
#include <stddef.h>
#include <stdint.h>

#include "gdf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdf_pixart_desc {
  int num_attention_heads;      /* 16 */
  int attention_head_dim;       /* 72 */
  int in_channels;              /* 4 */
  int out_channels;             /* 8 (learned sigma) */
  int num_layers;               /* 28 */
  int patch_size;               /* 2 */
  int sample_size;              /* 128 (1024 px) / 64 (512 px) */
  int caption_channels;         /* 4096 (T5) */
  int interpolation_scale;      /* 2 at sample_size 128, 1 at 64 */
} gdf_pixart_desc;

int gdf_pixart_model_create(const gdf_pixart_desc* desc, gdf_model** out);

/* Plan for `batch` latents of lat_h x lat_w (multiples of patch_size) and n_txt caption tokens. Hook tensors are logical
 * (B, C, lat_h/p, lat_w/p) stored channels-last. */
int gdf_pixart_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, int n_txt, const char* const* hook_ids, int n_hooks,
                           const gdf_plan_opts* opts, gdf_plan** out);

/* latents (B, in_channels, lat_h, lat_w) fp16 NCHW; timestep (B) fp32; encoder_hidden_states (B, n_txt, caption_channels)
 * fp16; text_lens (B) int32 = number of valid (unmasked, leading) caption tokens per sample, or NULL = all valid
 * (`encoder_attention_mask`, transformer_2d.py:397-399); out (B, out_channels, lat_h, lat_w) fp16 NCHW. */
int gdf_pixart_forward(gdf_plan* p, const void* latents, const float* timestep, const void* encoder_hidden_states,
                       const int* text_lens, void* const* hook_out, void* out, void* workspace, void* stream);

int gdf_pixart_plan_profile(gdf_plan* p, const void* latents, const float* timestep, const void* encoder_hidden_states,
                            const int* text_lens, void* const* hook_out, void* out, void* workspace, void* stream, float* ms,
                            const char** names, double* flops, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GDF_PIXART_H */
