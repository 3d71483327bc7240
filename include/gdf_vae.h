/* gdf_vae.h — C ABI of the step immediately before the hot path: VAE encode + latent sampling +
 * scheduler noise-add + scale_model_input, fused into one native call whose output is gdf_forward's `latents` input.
 *
 * What it replaces in the reference (paths relative to /root/reference/feature):
 *   `latents = self.pipe.prepare_latents(image, latent_timestep, 1, batch_size, prompt_embeds.dtype, device)`
 *       diffusion_feature.py:371-380  (StableDiffusion(XL)Img2ImgPipeline.prepare_latents: vae.encode(image)
 *       .latent_dist.sample() * vae.config.scaling_factor, then scheduler.add_noise(latents, noise, timestep))
 *   `latent_model_input = self.pipe.scheduler.scale_model_input(latent_model_input, t)`   diffusion_feature.py:405-406
 * AutoencoderKL is un-vendored diffusers; its encoder is built from ResnetBlock2D (diffusers/models/resnet.py:189-379),
 * Downsample2D(padding=0) (diffusers/models/downsampling.py:132-152) and a single-head Attention with GroupNorm and
 * residual (diffusers/models/attention_processor.py:50-297, 3244-3331), which ARE in the reference tree.
 *
 * Randomness stays with the caller (torch generators in the reference): `eps` is the posterior sample noise and
 * `noise` the scheduler noise; the scheduler's scalars for the chosen timestep are passed in:
 *     z = mean + exp(0.5 * clamp(logvar, -30, 20)) * eps          (eps == NULL: posterior mode)
 *     out = input_scale * (noise_a * scaling_factor * z + noise_b * noise)
 *   DDPM/PNDM (SD1.5): noise_a = sqrt(alphas_cumprod[t]), noise_b = sqrt(1 - alphas_cumprod[t]), input_scale = 1
 *   EulerDiscrete (SDXL): noise_a = 1, noise_b = sigma_t, input_scale = 1 / sqrt(sigma_t^2 + 1)
 * Handles and model functions (param names / set_param / weight bytes) are the ones of gdf.h.  Parameter names are the
 * `vae.state_dict()` names of the encoder half: "encoder.*" and "quant_conv.*".
 */
#ifndef GDF_VAE_H
#define GDF_VAE_H
#include <stddef.h>
#include <stdint.h>

#include "gdf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdf_vae_desc {
  int in_channels;                            /* 3 */
  int latent_channels;                        /* 4 */
  int n_levels;                               /* 4 */
  int block_out_channels[GDF_MAX_LEVELS];     /* 128, 256, 512, 512 */
  int layers_per_block;                       /* 2 */
  int use_quant_conv;                         /* 1 */
} gdf_vae_desc;

int gdf_vae_model_create(const gdf_vae_desc* desc, gdf_model** out);

/* Plan for `batch` images of img_h x img_w pixels (multiples of 8 * 2^(n_levels-1)); the library processes the batch in
 * sub-batches sized to its 32-bit buffer offsets (4 images at 1024^2) and sizes the workspace for one sub-batch. */
int gdf_vae_plan_create(gdf_model* m, int batch, int img_h, int img_w, gdf_plan** out);

/* image (B,3,H,W) fp16 NCHW in [-1,1]; eps / noise (B,L,H/f,W/f) fp16 NCHW or NULL; latents_out (B,L,H/f,W/f) fp16 NCHW,
 * f = 2^(n_levels-1) (8 for the SD / SDXL VAEs). */
int gdf_vae_encode(gdf_plan* p, const void* image, const void* eps, const void* noise, float scaling_factor,
                   float noise_a, float noise_b, float input_scale, void* latents_out, void* workspace, void* stream);

/* Per-op timing of one sub-batch pass (diagnostics; synchronises). Same contract as gdf_plan_profile. */
int gdf_vae_plan_profile(gdf_plan* p, const void* image, const void* eps, const void* noise, float scaling_factor,
                         float noise_a, float noise_b, float input_scale, void* latents_out, void* workspace, void* stream,
                         float* ms, const char** names, double* flops, int cap);

/* ---- `vae-out`: the optional last id of the reference's layer grammar (feature/diffusion_feature.py:60, :477-485) -------------
 *     latents    = self.pipe.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
 *     vae_output = self.pipe.vae.decode(latents / self.pipe.vae.config.scaling_factor, return_dict=False)[0]
 * One native call on the decoder half of the same AutoencoderKL (parameter names "post_quant_conv.*", "decoder.*"; the model is
 * created from the same gdf_vae_desc; Decoder / UpDecoderBlock2D are un-vendored diffusers, built from ResnetBlock2D(temb=None),
 * the single-head Attention and Upsample2D — nearest x2 + conv3x3, feature/diffusers/models/upsampling.py:142-195 — which ARE in
 * the reference tree).  The scheduler step of the first call after set_timesteps is linear in (latents, noise_pred) for the
 * schedulers the reference uses, so it travels as two scalars:
 *     z = (step_c_sample * latents + step_c_eps * noise_pred) * inv_scaling;      image = decode(z)
 *   PNDM (SD1.5), first step_plms call:  c_sample = sqrt(a_prev / a_t),  c_eps = -(a_prev - a_t) / (a_t sqrt(1 - a_prev) + sqrt(a_t (1 - a_t) a_prev))
 *   EulerDiscrete (SD2.1 / SDXL):        c_sample = 1,                   c_eps = sigma_next - sigma_t
 * noise_pred == NULL decodes step_c_sample * latents * inv_scaling (plain vae.decode).
 * latents / noise_pred (B,L,h,w) fp16 NCHW; image_out (B, h*f, w*f, 3) fp16 CHANNELS-LAST (the logical (B,3,H,W) tensor with strides
 * (H*W*3, 1, W*3, 3)); h, w multiples of 8, h*w <= 16384.  Sub-batching as in gdf_vae_plan_create. */
int gdf_vae_decoder_create(const gdf_vae_desc* desc, gdf_model** out);
int gdf_vae_decode_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, gdf_plan** out);
int gdf_vae_decode(gdf_plan* p, const void* latents, const void* noise_pred, float step_c_sample, float step_c_eps, float inv_scaling,
                   void* image_out, void* workspace, void* stream);
int gdf_vae_decode_plan_profile(gdf_plan* p, const void* latents, const void* noise_pred, float step_c_sample, float step_c_eps,
                                float inv_scaling, void* image_out, void* workspace, void* stream, float* ms, const char** names,
                                double* flops, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GDF_VAE_H */
