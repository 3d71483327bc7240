/* gdf.h — C ABI of the MI355X-native diffusion-feature hot path (libgdf.so).
 *
 * What it replaces in the reference (paths relative to /root/reference/feature):
 *   - the call `self.pipe.unet(latent_model_input, timestep=t, encoder_hidden_states=prompt_embeds,
 *     [added_cond_kwargs=...], return_dict=False)[0]`            diffusion_feature.py:445-465
 *     i.e. UNet2DConditionModel.forward                           diffusers/models/unet/unet_2d_condition.py:1040-1319
 *   - the side effects of every `feature_gatherer.gather(...)` call inside that forward
 *     (FeatureGatherer.gather -> FeatureStore.store)              components/feature_extractor.py:31-76,83-89
 *   - hook registration / id scheme (prepare_feature_extractor)   components/feature_extractor.py:92-288
 *
 * Conventions: plain C, no exceptions across the ABI, int status codes (0 = ok), caller owns every
 * device buffer, every launch is asynchronous on the hipStream_t passed in, one plan per host
 * thread, no hidden global state besides the per-thread last-error string.
 * All device pointers are HIP device pointers; activations are fp16 unless stated.
 */
#ifndef GDF_H
#define GDF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdf_model gdf_model;
typedef struct gdf_plan gdf_plan;

enum { GDF_OK = 0, GDF_ERR_ARG = 1, GDF_ERR_HIP = 2, GDF_ERR_STATE = 3, GDF_ERR_UNSUPPORTED = 4 };
enum { GDF_F16 = 0, GDF_F32 = 1, GDF_BF16 = 2, GDF_BF16X2 = 3, GDF_FP8MX = 4, GDF_F16S = 5 /* 3, 4, 5: gdf_flux_desc.compute_dtype only, see gdf_flux.h */ };

#define GDF_MAX_LEVELS 4

/* UNet2DConditionModel hyper-parameters (the `config.json` the reference downloads,
 * components/models.py:18-56; registered at unet_2d_condition.py:171-484). */
typedef struct gdf_arch_desc {
  int in_channels;                              /* 4 */
  int out_channels;                             /* 4 */
  int n_levels;                                 /* 4 (SD1.5) / 3 (SDXL) */
  int block_out_channels[GDF_MAX_LEVELS];       /* 320,640,1280,1280 / 320,640,1280 */
  int has_attn[GDF_MAX_LEVELS];                 /* CrossAttnDownBlock2D (1) or DownBlock2D (0) per level */
  int transformer_layers[GDF_MAX_LEVELS];       /* transformer_layers_per_block */
  int heads[GDF_MAX_LEVELS];                    /* attention heads per level */
  int layers_per_block;                         /* 2 */
  int cross_attention_dim;                      /* 768 / 2048 */
  int use_linear_projection;                    /* 0 / 1 (same arithmetic in NHWC; kept for weight shapes) */
  int time_embed_dim;                           /* 1280 */
  int addition_embed_text_time;                 /* 0 / 1 (SDXL `text_time`) */
  int addition_time_embed_dim;                  /* 256 */
  int add_in_dim;                               /* projection_class_embeddings_input_dim, 2816 */
} gdf_arch_desc;

/* Thread-local description of the last failure. */
const char* gdf_last_error(void);
/* Library/ABI version, bumped on any signature change. */
int gdf_abi_version(void);

/* ---- model: architecture + weights resident in HBM ------------------------------------------ */
int gdf_model_create(const gdf_arch_desc* arch, gdf_model** out);
void gdf_model_destroy(gdf_model* m);

/* Number of parameter tensors / i-th name in diffusers `state_dict()` naming
 * (e.g. "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight"). */
int gdf_model_param_count(const gdf_model* m);
const char* gdf_model_param_name(const gdf_model* m, int i);
/* shape in the diffusers layout (OIHW convs, [out,in] linears); returns ndim. */
int gdf_model_param_shape(const gdf_model* m, int i, int64_t shape[4]);

/* Copy one parameter from a DEVICE buffer in diffusers layout into the model's own MFMA-friendly
 * layout (OHWI convs, fused QKV / KV, interleaved GEGLU, stacked time projections). dtype GDF_F16|GDF_F32|GDF_BF16. */
int gdf_model_set_param(gdf_model* m, const char* name, const void* dev_ptr, int dtype, void* stream);
/* 1 when every parameter has been set. */
int gdf_model_ready(const gdf_model* m);
size_t gdf_model_weight_bytes(const gdf_model* m);
/* The model's device weight arena as ONE flat blob (already in the MFMA-friendly layout): what a data-parallel launch
 * broadcasts from rank 0 at init (RCCL, few large buckets) instead of re-reading and re-laying-out the checkpoint on every
 * rank; a receiving rank calls gdf_model_set_ready() after the blob has arrived.  Same architecture descriptor on both sides
 * => same layout. */
int gdf_model_weights(const gdf_model* m, void** dev_ptr, size_t* bytes);
int gdf_model_set_ready(gdf_model* m);

/* Every hook id this architecture can emit, in execution order (== the key order of the
 * reference's config_*_full.json dumps, extract_feature.py:103-110). */
int gdf_model_hook_count(const gdf_model* m);
const char* gdf_model_hook_name(const gdf_model* m, int i);

/* ---- plan: static op program for (batch, latent size, hook selection) ------------------------ */
typedef struct gdf_plan_opts {
  int stream_fp32;     /* 1: fp32 master copy of the residual stream (default), 0: fp16 only */
  int early_exit;      /* 1: stop after the last requested hook (opt-in; noise_pred is then not produced) */
  int reserved[6];     /* reserved[0] = shared_ctx: every sample uses ctx row-block 0 (one prompt repeated over the batch,
                          reference diffusion_feature.py:272): text K/V projections are computed once, not per sample.
                          reserved[1] = precise (opt-in, UNet plans, needs stream_fp32): every activation operand of an MFMA
                          contraction and every GroupNorm input is kept as a split fp16 pair hi + lo (22 mantissa bits) and
                          multiplied as [hi | lo] x [W | W] (K doubled, weights read twice) — removes the fp16-operand rounding
                          that bounds the default plans at ~1.0-1.3e-3 on `ffn-inner` / `unet-out`;
                          about twice the GEMM / conv time.  Attention internals stay fp16 (P always; q, k, v unless the qkv class below is split).
                          Round 4: reserved[1] = (class mask << 8) keeps only the listed operand CLASSES split (csrc/builder.h SP_*: stream 1, gnv 2,
                          ln_attn 4, attn_out 8, ln_ff 16, ff_inner 32, res 64, out 128, sampler 256, attn2_out 512, upsampler 1024,
                          qkv 2048 / xqkv 4096 = round 5: the self- / cross-attention q, k, v stored as pairs, the flash kernel contracts over both halves); 1 = all.
                          reserved[2] = cus: the plan will run on a stream restricted to this many CUs (gdf_stream_create_cu_mask):
                          tile selection, persistent grids and the XCD super-block order are sized for that partition. 0 = whole chip. */
} gdf_plan_opts;

int gdf_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, int n_ctx,
                    const char* const* hook_ids, int n_hooks, const gdf_plan_opts* opts, gdf_plan** out);
void gdf_plan_destroy(gdf_plan* p);
size_t gdf_plan_workspace_bytes(const gdf_plan* p);
int gdf_plan_num_ops(const gdf_plan* p);

/* Hook i of the plan (i indexes the ids ACCEPTED from hook_ids, in execution order; unknown ids are
 * silently ignored exactly like FeatureStore.store, feature_extractor.py:36).
 * A hook tensor is returned to the caller as logical (B, C, H, W) stored channels-last:
 * element (b,c,y,x) at  b*stride[0] + c*stride[1] + y*stride[2] + x*stride[3]  (in fp16 elements). */
typedef struct gdf_hook_info {
  const char* id;
  int64_t shape[4];     /* B, C, H, W   ('-map' hooks: B, heads, Q, K) */
  int64_t stride[4];
  size_t bytes;         /* size of the caller-provided buffer */
} gdf_hook_info;
int gdf_plan_hook_count(const gdf_plan* p);
int gdf_plan_hook_info(const gdf_plan* p, int i, gdf_hook_info* info);
/* 1: hook i is written by a separate coalesced copy op (`hook_store`, copy2d_kernel: read 2 B + write 2 B per element);
 * 0: its producing kernel stores it directly (GEMM / attention epilogue: `res-increment`, `*-map`, `cross-q`, `ffn-inner`, ...).
 * Measurement aid: the hook-write roofline counts only copied hooks against copy2d_kernel's time. */
int gdf_plan_hook_copied(const gdf_plan* p, int i);

/* One single-timestep denoiser forward.  latents (B,4,H,W) fp16 NCHW; timesteps (B) fp32;
 * ctx (B,n_ctx,cross_dim) fp16; add_text_embeds (B, add_in_dim - 6*addition_time_embed_dim) fp16 or NULL;
 * add_time_ids (B,6) fp32 or NULL; hook_out[i] = device buffer of gdf_hook_info.bytes;
 * noise_pred (B,H,W,4) fp16 channels-last or NULL; workspace >= gdf_plan_workspace_bytes. */
int gdf_forward(gdf_plan* p, const void* latents, const float* timesteps, const void* ctx,
                const void* add_text_embeds, const float* add_time_ids,
                void* const* hook_out, void* noise_pred, void* workspace, void* stream);

/* hipGraph replay: with enable != 0 every forward on a NON-default stream is served by one hipGraphLaunch of the plan's op
 * program, recorded once per distinct set of buffer addresses (workspace, inputs, hooks, outputs; LRU of 12) after one eager
 * warm-up forward.  The graph is BUILT with the graph API (kernel nodes in launch order, csrc/launch.h) — no stream is ever put into
 * capture mode, so other host threads may allocate, free, synchronise the device and build their own graphs meanwhile (round 6).
 * Calls on the legacy default stream and profiled calls run eagerly.
 * With live kernel timing on (gdf_plan_set_timing) the replayed graphs carry event-record nodes around the timed launches, one graph
 * per timing event set (LRU of 12), so the timing measures the replay itself.  Applies to UNet, Flux, PixArt and VAE plans alike. */
int gdf_plan_set_graph(gdf_plan* p, int enable);
int gdf_plan_graph_stats(const gdf_plan* p, long* captures, long* launches);
/* Forwards that had to run EAGERLY although graph replay was enabled, because graph construction / instantiation failed (each such call
 * retries; the first failure of a plan is also reported once on stderr).  0 on a healthy plan: a measurement that labels its
 * timed region "graph replay" must check this together with the launch count of gdf_plan_graph_stats. */
long gdf_plan_graph_failures(const gdf_plan* p);

/* Per-op timing of the last plan (diagnostics; synchronises the stream). Fills up to cap entries
 * with milliseconds per op, returns the op count. names[i] points into plan-owned storage. */
int gdf_plan_profile(gdf_plan* p, const void* latents, const float* timesteps, const void* ctx,
                     const void* add_text_embeds, const float* add_time_ids, void* const* hook_out,
                     void* noise_pred, void* workspace, void* stream,
                     float* ms, const char** names, double* flops, int cap);

/* Live per-kernel timing for roofline reporting: HIP events are recorded on `stream` around every op
 * whose kernel label (e.g. "gemm_kernel<dense,BN128>") matches; gdf_forward stays asynchronous.
 * gdf_plan_read_timing waits for the recorded events and returns the accumulated kernel time,
 * launch count and algorithmic FLOPs since gdf_plan_set_timing.  NULL label disables timing. */
int gdf_plan_num_kernel_labels(const gdf_plan* p);
/* kernel symbol (as rocprofv3 prints it, e.g. "gemm_kernel<0, 256, 320, 2, false>") that op i of the plan launches */
const char* gdf_plan_op_kernel(const gdf_plan* p, int i);
const char* gdf_plan_kernel_label(const gdf_plan* p, int i);
int gdf_plan_set_timing(gdf_plan* p, const char* kernel_label);
int gdf_plan_read_timing(gdf_plan* p, double* ms_total, long* launches, double* flops_total);
/* Time every `stride`-th launch of the label only (default 1 = every launch).  An event-record node costs ~3.5 us of GPU time
 * inside a replayed graph (766 of them: -2.4 % on the SDXL step), a stride of 8 keeps the measurement live and inside the timed
 * region at ~0.3 %.  gdf_plan_read_timing then returns the sums over the SAMPLED launches.  Call before gdf_plan_set_timing. */
int gdf_plan_set_timing_stride(gdf_plan* p, int stride);

/* ---- CU partitions: two or more forwards side by side on disjoint sets of compute units -----------------------------------
 * A GEMM / conv workgroup of this library owns its CU's whole register file, and all workgroups of a launch reach their HBM-bound
 * epilogues together, so on ONE stream MFMA-bound and HBM-bound phases alternate and never overlap.  Two
 * streams restricted to disjoint halves of the chip (hipExtStreamCreateWithCUMask) each run a half-batch plan built with
 * gdf_plan_opts.reserved[2] = CUs of the partition; their phases drift apart and one chain's epilogues / norm passes meet the other
 * chain's main loops.  mask: bit i of word i/32 enables CU i of the device's CU numbering; n_words = ceil(CUs / 32). */
int gdf_stream_create_cu_mask(const uint32_t* mask, int n_words, void** stream);
/* A private non-blocking stream for one plan (round 6).  The Python mirror used to take plan streams from torch's pool of 32, which hands
 * the same hipStream_t to every 32nd request: two LIVE plans of two host threads could share a stream (one capturing while the other
 * launches).  The reference has no equivalent (one default stream per process, feature/diffusion_feature.py:445-465).  Creation and
 * destruction are serialised with the library's other allocations (csrc/model.h). */
int gdf_stream_create(void** stream);
int gdf_stream_destroy(void* stream);
int gdf_device_cu_count(void);
/* Diagnostics: launches n_blocks single-wave workgroups on `stream`; workgroup b writes {XCC_ID, HW_ID} of the CU it ran on to
 * dev_out[2b], dev_out[2b+1] and then idles for ~spin * 64 * 64 cycles, so the records show which physical CUs the stream uses. */
int gdf_cu_census(uint32_t* dev_out, int n_blocks, int spin, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDF_H */
