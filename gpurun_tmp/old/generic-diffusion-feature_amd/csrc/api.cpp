// extern "C" surface of libgdf.so — see include/gdf.h for the contract and the reference call sites.
#include <cstring>

#include "model.h"

using namespace gdf;

struct gdf_model { Model* m; };
struct gdf_plan { Plan p; gdf_model* owner; };

extern "C" {

const char* gdf_last_error(void) { return last_error(); }
int gdf_abi_version(void) { return 1; }

int gdf_model_create(const gdf_arch_desc* arch, gdf_model** out) {
  if (!arch || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libgdf has no CPU fallback"); return GDF_ERR_HIP; }
  Model* m = model_create(*arch);
  if (!m) return GDF_ERR_ARG;
  *out = new gdf_model{m};
  return GDF_OK;
}
void gdf_model_destroy(gdf_model* m) { if (m) { model_destroy(m->m); delete m; } }
int gdf_model_param_count(const gdf_model* m) { return m ? (int)m->m->params.size() : 0; }
const char* gdf_model_param_name(const gdf_model* m, int i) {
  return (m && i >= 0 && i < (int)m->m->params.size()) ? m->m->params[i].name.c_str() : nullptr;
}
int gdf_model_param_shape(const gdf_model* m, int i, int64_t shape[4]) {
  if (!m || i < 0 || i >= (int)m->m->params.size()) return 0;
  const ParamRec& p = m->m->params[i];
  for (int k = 0; k < 4; ++k) shape[k] = k < p.ndim ? p.shape[k] : 1;
  return p.ndim;
}
int gdf_model_set_param(gdf_model* m, const char* name, const void* dev_ptr, int dtype, void* stream) {
  if (!m || !name || !dev_ptr) { set_error("null argument"); return GDF_ERR_ARG; }
  return model_set_param(m->m, name, dev_ptr, dtype, (hipStream_t)stream);
}
int gdf_model_ready(const gdf_model* m) { return m && m->m->n_set == (int)m->m->params.size(); }
int gdf_model_weights(const gdf_model* m, void** dev_ptr, size_t* bytes) {
  if (!m || !dev_ptr || !bytes) { set_error("null argument"); return GDF_ERR_ARG; }
  *dev_ptr = m->m->weights; *bytes = m->m->weight_bytes;
  return GDF_OK;
}
int gdf_model_set_ready(gdf_model* m) {
  if (!m) { set_error("null model"); return GDF_ERR_ARG; }
  for (auto& p : m->m->params) p.set = true;
  m->m->n_set = (int)m->m->params.size();
  return GDF_OK;
}
size_t gdf_model_weight_bytes(const gdf_model* m) { return m ? m->m->weight_bytes : 0; }
int gdf_model_hook_count(const gdf_model* m) { return m ? (int)m->m->hook_names.size() : 0; }
const char* gdf_model_hook_name(const gdf_model* m, int i) {
  return (m && i >= 0 && i < (int)m->m->hook_names.size()) ? m->m->hook_names[i].c_str() : nullptr;
}

int gdf_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, int n_ctx, const char* const* hook_ids, int n_hooks,
                    const gdf_plan_opts* opts, gdf_plan** out) {
  if (!m || !out || (n_hooks > 0 && !hook_ids)) { set_error("null argument"); return GDF_ERR_ARG; }
  if (m->m->kind != 0) { set_error("gdf_plan_create on a Flux model: use gdf_flux_plan_create"); return GDF_ERR_ARG; }
  gdf_plan_opts o{};
  o.stream_fp32 = 1;
  if (opts) o = *opts;
  gdf_plan* p = new gdf_plan();
  p->owner = m;
  p->p.model = m->m;
  const int rc = plan_build(*m->m, p->p, batch, lat_h, lat_w, n_ctx, hook_ids, n_hooks, o, false);
  if (rc != GDF_OK) { delete p; return rc; }
  *out = p;
  return GDF_OK;
}
void gdf_plan_destroy(gdf_plan* p) { delete p; }
size_t gdf_plan_workspace_bytes(const gdf_plan* p) { return p ? p->p.ws_bytes : 0; }
int gdf_plan_num_ops(const gdf_plan* p) { return p ? (int)p->p.ops.size() : 0; }
int gdf_plan_hook_count(const gdf_plan* p) { return p ? (int)p->p.hooks.size() : 0; }
int gdf_plan_hook_info(const gdf_plan* p, int i, gdf_hook_info* info) {
  if (!p || !info || i < 0 || i >= (int)p->p.hooks.size()) { set_error("bad hook index"); return GDF_ERR_ARG; }
  const HookSlot& h = p->p.hooks[i];
  info->id = h.id.c_str();
  for (int k = 0; k < 4; ++k) { info->shape[k] = h.shape[k]; info->stride[k] = h.stride[k]; }
  info->bytes = h.bytes;
  return GDF_OK;
}

int gdf_plan_hook_copied(const gdf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->p.hooks.size() && p->p.hooks[i].copied) ? 1 : 0;
}

// ---- PixArt DiT front end (include/gdf_pixart.h) ----
int gdf_pixart_model_create(const gdf_pixart_desc* desc, gdf_model** out) {
  if (!desc || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libgdf has no CPU fallback"); return GDF_ERR_HIP; }
  Model* m = pixart_model_create(*desc);
  if (!m) return GDF_ERR_ARG;
  *out = new gdf_model{m};
  return GDF_OK;
}
int gdf_pixart_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, int n_txt, const char* const* hook_ids, int n_hooks,
                           const gdf_plan_opts* opts, gdf_plan** out) {
  if (!m || !out || (n_hooks > 0 && !hook_ids)) { set_error("null argument"); return GDF_ERR_ARG; }
  gdf_plan_opts o{};
  o.stream_fp32 = 1;
  if (opts) o = *opts;
  gdf_plan* p = new gdf_plan();
  p->owner = m;
  p->p.model = m->m;
  const int rc = pixart_plan_build(*m->m, p->p, batch, lat_h, lat_w, n_txt, hook_ids, n_hooks, o, false);
  if (rc != GDF_OK) { delete p; return rc; }
  *out = p;
  return GDF_OK;
}
int gdf_pixart_forward(gdf_plan* p, const void* latents, const float* timestep, const void* encoder_hidden_states,
                       const int* text_lens, void* const* hook_out, void* out, void* workspace, void* stream) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return pixart_forward(p->p, *p->p.model, latents, timestep, encoder_hidden_states, text_lens, hook_out, out, workspace,
                        (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}
int gdf_pixart_plan_profile(gdf_plan* p, const void* latents, const float* timestep, const void* encoder_hidden_states,
                            const int* text_lens, void* const* hook_out, void* out, void* workspace, void* stream, float* ms,
                            const char** names, double* flops, int cap) {
  if (!p || !ms) { set_error("null argument"); return -1; }
  const int rc = pixart_forward(p->p, *p->p.model, latents, timestep, encoder_hidden_states, text_lens, hook_out, out, workspace,
                                (hipStream_t)stream, ms, names, flops, cap);
  if (rc != GDF_OK) return -1;
  return (int)p->p.ops.size();
}

// ---- VAE encoder front end (include/gdf_vae.h) ----
int gdf_vae_model_create(const gdf_vae_desc* desc, gdf_model** out) {
  if (!desc || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libgdf has no CPU fallback"); return GDF_ERR_HIP; }
  Model* m = vae_model_create(*desc);
  if (!m) return GDF_ERR_ARG;
  *out = new gdf_model{m};
  return GDF_OK;
}
int gdf_vae_plan_create(gdf_model* m, int batch, int img_h, int img_w, gdf_plan** out) {
  if (!m || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  gdf_plan* p = new gdf_plan();
  p->owner = m;
  p->p.model = m->m;
  const int rc = vae_plan_build(*m->m, p->p, batch, img_h, img_w, false);
  if (rc != GDF_OK) { delete p; return rc; }
  *out = p;
  return GDF_OK;
}
int gdf_vae_encode(gdf_plan* p, const void* image, const void* eps, const void* noise, float scaling_factor, float noise_a,
                   float noise_b, float input_scale, void* latents_out, void* workspace, void* stream) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return vae_encode(p->p, *p->p.model, image, eps, noise, scaling_factor, noise_a, noise_b, input_scale, latents_out, workspace,
                    (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}
int gdf_vae_plan_profile(gdf_plan* p, const void* image, const void* eps, const void* noise, float scaling_factor, float noise_a,
                         float noise_b, float input_scale, void* latents_out, void* workspace, void* stream, float* ms,
                         const char** names, double* flops, int cap) {
  if (!p || !ms) { set_error("null argument"); return -1; }
  const int rc = vae_encode(p->p, *p->p.model, image, eps, noise, scaling_factor, noise_a, noise_b, input_scale, latents_out,
                            workspace, (hipStream_t)stream, ms, names, flops, cap);
  if (rc != GDF_OK) return -1;
  return (int)p->p.ops.size();
}

// ---- VAE decoder: the optional `vae-out` feature (include/gdf_vae.h) ----
int gdf_vae_decoder_create(const gdf_vae_desc* desc, gdf_model** out) {
  if (!desc || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libgdf has no CPU fallback"); return GDF_ERR_HIP; }
  Model* m = vae_decoder_create(*desc);
  if (!m) return GDF_ERR_ARG;
  *out = new gdf_model{m};
  return GDF_OK;
}
int gdf_vae_decode_plan_create(gdf_model* m, int batch, int lat_h, int lat_w, gdf_plan** out) {
  if (!m || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  gdf_plan* p = new gdf_plan();
  p->owner = m;
  p->p.model = m->m;
  const int rc = vae_dec_plan_build(*m->m, p->p, batch, lat_h, lat_w, false);
  if (rc != GDF_OK) { delete p; return rc; }
  *out = p;
  return GDF_OK;
}
int gdf_vae_decode(gdf_plan* p, const void* latents, const void* noise_pred, float step_c_sample, float step_c_eps, float inv_scaling,
                   void* image_out, void* workspace, void* stream) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return vae_decode(p->p, *p->p.model, latents, noise_pred, step_c_sample, step_c_eps, inv_scaling, image_out, workspace,
                    (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}
int gdf_vae_decode_plan_profile(gdf_plan* p, const void* latents, const void* noise_pred, float step_c_sample, float step_c_eps,
                                float inv_scaling, void* image_out, void* workspace, void* stream, float* ms, const char** names,
                                double* flops, int cap) {
  if (!p || !ms) { set_error("null argument"); return -1; }
  const int rc = vae_decode(p->p, *p->p.model, latents, noise_pred, step_c_sample, step_c_eps, inv_scaling, image_out, workspace,
                            (hipStream_t)stream, ms, names, flops, cap);
  if (rc != GDF_OK) return -1;
  return (int)p->p.ops.size();
}

// ---- MMDiT / Flux front end (include/gdf_flux.h) ----
int gdf_flux_model_create(const gdf_flux_desc* desc, gdf_model** out) {
  if (!desc || !out) { set_error("null argument"); return GDF_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libgdf has no CPU fallback"); return GDF_ERR_HIP; }
  Model* m = flux_model_create(*desc);
  if (!m) return GDF_ERR_ARG;
  *out = new gdf_model{m};
  return GDF_OK;
}
int gdf_flux_plan_create(gdf_model* m, int batch, int img_h, int img_w, int n_txt, const char* const* hook_ids, int n_hooks,
                         const gdf_plan_opts* opts, gdf_plan** out) {
  if (!m || !out || (n_hooks > 0 && !hook_ids)) { set_error("null argument"); return GDF_ERR_ARG; }
  gdf_plan_opts o{};
  o.stream_fp32 = 1;
  if (opts) o = *opts;
  gdf_plan* p = new gdf_plan();
  p->owner = m;
  p->p.model = m->m;
  const int rc = flux_plan_build(*m->m, p->p, batch, img_h, img_w, n_txt, hook_ids, n_hooks, o, false);
  if (rc != GDF_OK) { delete p; return rc; }
  *out = p;
  return GDF_OK;
}
int gdf_flux_forward(gdf_plan* p, const void* hidden_states, const void* encoder_hidden_states, const void* pooled_projections,
                     const float* timestep, const float* guidance, const float* img_ids, const float* txt_ids,
                     void* const* hook_out, void* out, void* workspace, void* stream) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return flux_forward(p->p, *p->p.model, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance, img_ids,
                      txt_ids, hook_out, out, workspace, (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}
int gdf_flux_plan_profile(gdf_plan* p, const void* hidden_states, const void* encoder_hidden_states,
                          const void* pooled_projections, const float* timestep, const float* guidance, const float* img_ids,
                          const float* txt_ids, void* const* hook_out, void* out, void* workspace, void* stream, float* ms,
                          const char** names, double* flops, int cap) {
  if (!p || !ms) { set_error("null argument"); return -1; }
  const int rc = flux_forward(p->p, *p->p.model, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance,
                              img_ids, txt_ids, hook_out, out, workspace, (hipStream_t)stream, ms, names, flops, cap);
  if (rc != GDF_OK) return -1;
  return (int)p->p.ops.size();
}

int gdf_forward(gdf_plan* p, const void* latents, const float* timesteps, const void* ctx, const void* add_text_embeds,
                const float* add_time_ids, void* const* hook_out, void* noise_pred, void* workspace, void* stream) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return plan_forward(p->p, *p->p.model, latents, timesteps, ctx, add_text_embeds, add_time_ids, hook_out, noise_pred,
                      workspace, (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}

int gdf_plan_profile(gdf_plan* p, const void* latents, const float* timesteps, const void* ctx,
                     const void* add_text_embeds, const float* add_time_ids, void* const* hook_out, void* noise_pred,
                     void* workspace, void* stream, float* ms, const char** names, double* flops, int cap) {
  if (!p || !ms) { set_error("null argument"); return -1; }
  const int rc = plan_forward(p->p, *p->p.model, latents, timesteps, ctx, add_text_embeds, add_time_ids, hook_out,
                              noise_pred, workspace, (hipStream_t)stream, ms, names, flops, cap);
  if (rc != GDF_OK) return -1;
  return (int)p->p.ops.size();
}

const char* gdf_plan_op_kernel(const gdf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->p.ops.size()) ? p->p.labels[p->p.ops[i].label].c_str() : nullptr;
}
int gdf_plan_num_kernel_labels(const gdf_plan* p) { return p ? (int)p->p.labels.size() : 0; }
const char* gdf_plan_kernel_label(const gdf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->p.labels.size()) ? p->p.labels[i].c_str() : nullptr;
}
int gdf_plan_set_graph(gdf_plan* p, int enable) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  p->p.graph_mode = enable ? 1 : 0;
  return GDF_OK;
}
int gdf_plan_graph_stats(const gdf_plan* p, long* captures, long* launches) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  if (captures) *captures = p->p.graph_captures;
  if (launches) *launches = p->p.graph_launches;
  return GDF_OK;
}
long gdf_plan_graph_failures(const gdf_plan* p) { return p ? p->p.graph_capture_failures : -1; }
int gdf_plan_set_timing(gdf_plan* p, const char* kernel_label) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return plan_set_timing(p->p, kernel_label);
}
int gdf_plan_set_timing_stride(gdf_plan* p, int stride) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  if (stride < 1) { set_error("timing stride must be >= 1"); return GDF_ERR_ARG; }
  if (p->p.timing_label >= 0) { set_error("set the stride before gdf_plan_set_timing"); return GDF_ERR_STATE; }
  p->p.timing_stride = stride;
  return GDF_OK;
}
int gdf_plan_read_timing(gdf_plan* p, double* ms_total, long* launches, double* flops_total) {
  if (!p) { set_error("null plan"); return GDF_ERR_ARG; }
  return plan_read_timing(p->p, ms_total, launches, flops_total);
}


// ---- CU-partitioned streams (gdf.h) -------------------------------------------------------------------------------------------
int gdf_stream_create_cu_mask(const uint32_t* mask, int n_words, void** stream) {
  if (!mask || n_words < 1 || !stream) { set_error("gdf_stream_create_cu_mask: bad arguments"); return GDF_ERR_ARG; }
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask);
  if (e != hipSuccess) { set_error(std::string("hipExtStreamCreateWithCUMask: ") + hipGetErrorString(e)); return GDF_ERR_HIP; }
  *stream = (void*)s;
  return GDF_OK;
}
int gdf_stream_create(void** stream) {
  if (!stream) { set_error("gdf_stream_create: bad arguments"); return GDF_ERR_ARG; }
  hipStream_t s = nullptr;
  gdf::CaptureExclusive guard;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { set_error(std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e)); return GDF_ERR_HIP; }
  *stream = (void*)s;
  return GDF_OK;
}
int gdf_stream_destroy(void* stream) {
  if (!stream) return GDF_OK;
  gdf::CaptureExclusive guard;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? GDF_OK : GDF_ERR_HIP;
}
int gdf_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

}  // extern "C"

// one record per workgroup: (XCC_ID, HW_ID) of the CU it ran on — which physical CUs a CU-masked stream really uses
__global__ void gdf_cu_census_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
  }
  // keep the workgroup resident for a while so that the grid spreads over every CU the stream may use
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
extern "C" int gdf_cu_census(uint32_t* dev_out, int n_blocks, int spin, void* stream) {
  if (!dev_out || n_blocks < 1) { gdf::set_error("gdf_cu_census: bad arguments"); return GDF_ERR_ARG; }
  hipLaunchKernelGGL(gdf_cu_census_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, dev_out, spin);
  return hipGetLastError() == hipSuccess ? GDF_OK : GDF_ERR_HIP;
}
