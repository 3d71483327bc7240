// Internal host-side structures of libgdf.so (model = arch + weight arena; plan = static op program).
#pragma once
#include <functional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/gdf.h"
#include "../../include/gdf_flux.h"
#include "../../include/gdf_vae.h"
#include "../../include/gdf_pixart.h"
#include "kernels.h"

namespace gdf {

typedef gdf_arch_desc GdfArch;
typedef gdf_plan_opts PlanOpts;
static const size_t NPOS = (size_t)-1;

void set_error(const std::string& s);
const char* last_error();

// ---- weights (byte offsets into the model's device arena) ---------------------------------------
struct NormW { size_t g = 0, b = 0; int c = 0; };                               // fp32 gamma / beta
struct ConvW { size_t w = 0, b = 0; int cin = 0, cout = 0; };                    // fp16 [cout][9][cin], fp32 bias
struct LinW { size_t w = 0, b = NPOS; int n = 0, k = 0; bool has_bias = false; };  // fp16 [n][k], fp32 bias
struct ResnetW { NormW n1, n2; ConvW c1, c2; LinW sc; bool has_sc = false; int cin = 0, cout = 0, temb_off = 0;
                 bool has_temb = true; float eps = 1e-5f; };   // VAE resnets: temb_channels=None, eps 1e-6
struct BlockW { NormW ln1, ln2, ln3; LinW qkv, o1, q2, kv2, o2, ff1, ff2; int kv_group = 0, kv_index = 0; };
// text K/V projection weights of all transformer blocks with the same width live contiguously: one grouped GEMM per width
struct KvGroup { int C = 0, count = 0, next = 0; size_t base = 0, stride = 0; };
struct VitW { NormW gn; LinW pin, pout; std::vector<BlockW> blocks; int c = 0, heads = 0; };
struct LevelW { std::vector<ResnetW> res; std::vector<VitW> vit; std::vector<int> skip_c; bool has_sampler = false; ConvW sampler; };

enum ParamKind { PK_VEC, PK_VEC_OFF, PK_VEC_GEGLU, PK_CONV3, PK_CONV_IN, PK_ROWS, PK_ROWS_GEGLU, PK_ROWS_PADK };
struct ParamRec {
  std::string name; int ndim = 0; int64_t shape[4] = {0, 0, 0, 0};
  int kind = 0; size_t dst = 0; int a0 = 0, a1 = 0, a2 = 0; bool set = false;
};
struct TembReg { std::string name; int cout; int off; };

// ---- MMDiT (Flux) weights ------------------------------------------------------------------------
// mod_*: column offsets (floats) into the per-sample modulation table [B][mod_total] = Linear(silu(temb)) of every
// AdaLayerNormZero / ZeroSingle / Continuous, stacked into ONE [mod_total][C] matrix (a single launch per forward).
struct FluxDoubleW {
  int mod = 0, cmod = 0;                       // norm1.linear (6C: shift,scale,gate msa | shift,scale,gate mlp), norm1_context.linear
  LinW qkv, cqkv, o, co, ff1, ff2, cff1, cff2; // to_q|k|v fused [3C][C], add_q|k|v_proj fused, to_out.0, to_add_out, ff, ff_context
  size_t nq = 0, nk = 0, cnq = 0, cnk = 0;     // RMSNorm gains fp32 [D]: norm_q, norm_k, norm_added_q, norm_added_k
};
struct FluxSingleW {
  int mod = 0;                                 // norm.linear (3C: shift, scale, gate)
  LinW qkv, mlp, out;                          // to_q|k|v fused, proj_mlp, proj_out [C][C + hid]
  size_t nq = 0, nk = 0;
};
struct FluxW {
  gdf_flux_desc d{};
  int C = 0, hid = 0, D = 0, mod_total = 0, mod_out = 0;
  LinW x_emb, ctx_emb, t1, t2, g1, g2, p1, p2, mod_all, proj_out;
  std::vector<FluxDoubleW> dbl;
  std::vector<FluxSingleW> sgl;
};

// ---- PixArt DiT weights -----------------------------------------------------------------------------
struct PixartBlockW { LinW qkv, o1, q2, kv2, o2, ff1, ff2; int table = 0; };   // table: float offset of scale_shift_table (6C)
struct PixartW {
  gdf_pixart_desc d{};
  int C = 0, kpad = 0;
  LinW patch, t1, t2, ada, cap1, cap2, proj_out;
  size_t tables = 0;                           // fp32 [num_layers * 6C + 2C]: every block's scale_shift_table, then the final one
  std::vector<PixartBlockW> blocks;
};

// ---- VAE encoder weights ---------------------------------------------------------------------------
struct VaeW {
  gdf_vae_desc d{};
  ConvW conv_in, conv_out;
  std::vector<std::vector<ResnetW>> down;      // [level][layer]   (encoder)
  std::vector<ConvW> downsamplers;             // level < L-1
  ResnetW mid0, mid1;
  NormW attn_gn, norm_out;
  LinW q, k, v, o, quant;
  // decoder half (Model::kind 4): up blocks of layers_per_block + 1 resnets over the REVERSED channel list, Upsample2D on all but the
  // last block, post_quant_conv ([L][L], fp16 rows + fp32 bias)
  std::vector<std::vector<ResnetW>> up;
  std::vector<ConvW> upsamplers;
  LinW post_quant;
};

struct Model {
  int bf16 = 0;                                // 16-bit element type of weights / activations: 0 fp16, 1 bf16 (Flux only)
  // Flux 'fp8-mx' (GDF_FP8MX): the arena additionally holds, for every [n][k] linear at byte offset w, its fp8 (e4m3) copy at f8_off + w / 2 and
  // the per-output-channel power-of-two scales (float[n]) at sc_off + w / 16 (both written when the parameter is set)
  int fp8 = 0; size_t f8_off = 0, sc_off = 0;
  float hid_scale = 0.f;                       // Flux 'float16s' (GDF_F16S): the MLP hidden tensors (and the single blocks' [attn | mlp] operand rows) are stored
                                               // as fp16 of (x * hid_scale), a power of two < 1 — fp16 mantissa, +-1.7e7 range; undone on the consumer's accumulators
  int x2 = 0;                                  // Flux 'bfloat16x2' (GDF_BF16X2): bf16 weights, activation operands as bf16 hi + lo pairs, fp16 attention internals
  int kind = 0;                                // 0: UNet2DConditionModel, 1: FluxTransformer2DModel, 2: AutoencoderKL encoder, 3: PixArt DiT, 4: AutoencoderKL decoder
  FluxW flux;
  VaeW vae;
  PixartW pix;                                 // kind 3
  GdfArch arch{};
  void* weights = nullptr;
  size_t weight_bytes = 0;
  std::vector<ParamRec> params;
  std::unordered_map<std::string, int> index;
  int n_set = 0;
  ConvW conv_in, conv_out;
  LinW te1, te2, ae1, ae2, temb_all;
  std::vector<TembReg> temb_regs;
  int temb_total = 0;
  std::vector<LevelW> down, up;
  ResnetW mid_res0, mid_res1;
  VitW mid_vit;
  NormW norm_out;
  std::vector<std::string> hook_names;
  std::vector<KvGroup> kv_groups;
};

// ---- plan ----------------------------------------------------------------------------------------
// Flux reuses the slots: LAT = hidden_states, T = timestep, CTX = encoder_hidden_states, TXT = pooled_projections,
// TID = guidance, NOISE = output; IDS_IMG / IDS_TXT = img_ids / txt_ids
enum { BUF_WS = 0, BUF_WT, BUF_LAT, BUF_T, BUF_CTX, BUF_TXT, BUF_TID, BUF_NOISE, BUF_IDS_IMG, BUF_IDS_TXT, BUF_COUNT };
enum { BUF_HOOK0 = 1 << 16 };                   // Ref.buf = BUF_HOOK0 + slot: the caller's hook buffer `slot` (an op's output IS the hook)
struct Ref { int buf = BUF_WS; size_t off = 0; };
struct Bind {
  char* base[BUF_COUNT] = {nullptr};
  void* const* hooks = nullptr;
  float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // run-time scalars (VAE: scaling_factor, noise_a, noise_b, input_scale)
  void* p(const Ref& r) const { return (r.buf >= BUF_HOOK0 ? (char*)hooks[r.buf - BUF_HOOK0] : base[r.buf]) + r.off; }
  void* ws(size_t off) const { return base[BUF_WS] + off; }
  void* hook(int slot) const { return hooks[slot]; }
};
struct Op { const char* name; double flops; std::function<hipError_t(const Bind&, hipStream_t)> fn; int label = 0; };
struct HookSlot { std::string id; int64_t shape[4]; int64_t stride[4]; size_t bytes; bool copied = false; };   // copied: stored by a hook_store (copy2d) op, else by its producer's epilogue

// Stream-capture guard.  hipFree / hipMalloc / hipGraph(Exec)Destroy / hipStreamDestroy made by ANY host thread while another thread's stream is
// capturing can invalidate that capture on this runtime (relaxed mode notwithstanding), and an invalidated capture leaves the stream unusable
// ("operation failed due to a previous error during capture" from every later call on it; round 6: a model garbage-collected in one thread while
// another thread captured took seven later tests down through torch's 32-entry stream pool).  Captures hold the guard shared; every
// allocation / free the library makes outside a capture holds it exclusively for the duration of the HIP call.
struct CaptureShared { CaptureShared(); ~CaptureShared(); };
struct CaptureExclusive { CaptureExclusive(); ~CaptureExclusive(); };

struct Plan {
  const Model* model = nullptr;
  int chunk = 0;                              // VAE: images per pass (the plan is built for `chunk`, forward loops)
  int batch = 0, H = 0, W = 0, n_ctx = 0;    // Flux: H x W = packed-latent token grid, n_ctx = text tokens
  PlanOpts opts{};
  std::vector<Op> ops;
  std::vector<HookSlot> hooks;
  std::unordered_set<std::string> requested;
  bool want_maps = false;
  bool writes_noise = false;
  size_t ws_bytes = 0;
  std::vector<std::string> dry_ids;
  // live per-kernel timing (bench roofline): HIP events around every op of one kernel label
  std::vector<std::string> labels;
  int timing_label = -1;
  static const int EV_RING = 4;
  std::vector<hipEvent_t> ev[EV_RING];
  bool ev_used[EV_RING] = {false, false, false, false};
  int ev_next = 0;
  int timing_stride = 1;                      // time every timing_stride-th launch of the label
  double t_ms = 0, t_flops = 0; long t_launches = 0;
  // hipGraph replay of the op program (gdf_plan_set_graph): one captured + instantiated graph per distinct binding table
  // (the op program is static, only the caller's buffer addresses vary), small LRU
  // evset >= 0: the capture carries event-record nodes (hipEventRecordExternal) of timing event set `evset` around every launch of the
  // timed kernel label, so that gdf_plan_set_timing measures the graph REPLAY (the product path) and not an eager re-launch
  struct GraphEntry { Bind key; std::vector<void*> hook_ptrs; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; long stamp = 0; int evset = -1; int label = -1; };
  int graph_mode = 0;
  bool warmed = false;                        // first forward runs eagerly (lazy one-time kernel attribute setup)
  bool timed_graph_broken = false;            // a capture with event-record nodes failed once: timed forwards run eagerly
  long graph_clock = 0, graph_launches = 0, graph_captures = 0, graph_capture_failures = 0;
  std::vector<GraphEntry> graphs;
  ~Plan();
};
const char* kernel_label(const char* opname);
int plan_set_timing(Plan& P, const char* label);
int plan_read_timing(Plan& P, double* ms, long* launches, double* flops);

Model* model_create(const GdfArch& arch);
void model_destroy(Model* m);
int model_set_param(Model* m, const char* name, const void* src, int dtype, hipStream_t s);
int plan_build(const Model& m, Plan& P, int batch, int H, int W, int n_ctx, const char* const* ids, int n_ids,
               const PlanOpts& opts, bool dry);
int plan_forward(Plan& P, const Model& m, const void* lat, const float* t, const void* ctx, const void* txt,
                 const float* tid, void* const* hook_out, void* noise, void* ws, hipStream_t s, float* ms,
                 const char** names, double* flops, int cap);
// executes the op program against an already filled binding table (shared by the UNet and Flux front ends)
int plan_run(Plan& P, const Bind& b, hipStream_t s, float* ms, const char** names, double* flops, int cap);

// ---- VAE encoder front end (include/gdf_vae.h) ----
Model* vae_model_create(const gdf_vae_desc& d);
int vae_plan_build(const Model& m, Plan& P, int batch, int img_h, int img_w, bool dry);
int vae_encode(Plan& P, const Model& m, const void* image, const void* eps, const void* noise, float scaling, float noise_a,
               float noise_b, float in_scale, void* out, void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap);

// ---- VAE decoder (`vae-out`, include/gdf_vae.h) ----
Model* vae_decoder_create(const gdf_vae_desc& d);
int vae_dec_plan_build(const Model& m, Plan& P, int batch, int lat_h, int lat_w, bool dry);
int vae_decode(Plan& P, const Model& m, const void* latents, const void* noise_pred, float c_sample, float c_eps, float inv_scaling,
               void* image_out, void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap);

// ---- PixArt DiT front end (include/gdf_pixart.h) ----
Model* pixart_model_create(const gdf_pixart_desc& d);
int pixart_plan_build(const Model& m, Plan& P, int batch, int lat_h, int lat_w, int n_txt, const char* const* ids, int n_ids,
                      const PlanOpts& opts, bool dry);
int pixart_forward(Plan& P, const Model& m, const void* latents, const float* timestep, const void* enc, const int* text_lens,
                   void* const* hook_out, void* out, void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap);

Model* flux_model_create(const gdf_flux_desc& d);
int flux_plan_build(const Model& m, Plan& P, int batch, int img_h, int img_w, int n_txt, const char* const* ids, int n_ids,
                    const PlanOpts& opts, bool dry);
int flux_forward(Plan& P, const Model& m, const void* hidden, const void* enc, const void* pooled, const float* timestep,
                 const float* guidance, const float* img_ids, const float* txt_ids, void* const* hook_out, void* out,
                 void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap);

}  // namespace gdf
