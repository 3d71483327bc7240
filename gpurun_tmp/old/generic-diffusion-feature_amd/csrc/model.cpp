// Host side of libgdf.so: architecture walk, weight arena + re-layout, static plan builder
// (op program + workspace arena + hook table) and the forward executor.
//
// The op program restates, for the reference's single-timestep path, the orchestration of
//   UNet2DConditionModel.forward        /root/reference/feature/diffusers/models/unet/unet_2d_condition.py:1040-1319
//   (un-vendored diffusers==0.32.2 unet_2d_blocks wiring: CrossAttnDownBlock2D / DownBlock2D /
//    UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D / UpBlock2D)
//   ResnetBlock2D.forward               resnet.py:320-379
//   Transformer2DModel.forward          transformers/transformer_2d.py:327-530
//   BasicTransformerBlock.forward       attention.py:469-592
//   Attention + AttnProcessor2_0        attention_processor.py:3244-3331 (AttnStoreProcessor for '-map')
//   FeedForward / GEGLU                 attention.py:1249-1258
//   Downsample2D / Upsample2D           downsampling.py:132-152 / upsampling.py:142-195
// and of the hook id scheme in components/feature_extractor.py:126-249.
//
// Data layout in HBM: every activation is NHWC == token-major [B*H*W][C] fp16 with an explicit leading
// dimension, so (a) conv and transformer layers share one layout with no permutes, (b) the skip
// concatenations of the up path are free: producers store straight into channel slices of a
// pre-allocated concat buffer.  The residual stream additionally keeps an fp32 master copy
// (stream_fp32) that only the residual adds in GEMM epilogues read and write.
#include "builder.h"
#include <shared_mutex>

namespace gdf {

thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
const char* last_error() { return g_err.c_str(); }

// =====================================================================================================
// Model: walk the architecture once, lay out the weight arena, register diffusers parameter names
// =====================================================================================================
struct ModelBuilder : WeightBuilder {
  explicit ModelBuilder(Model& mm) : WeightBuilder(mm) {}
  ResnetW resnet(const std::string& p, int ci, int co) {
    ResnetW r; r.cin = ci; r.cout = co;
    r.n1 = norm(p + ".norm1", ci);
    r.c1 = conv3(p + ".conv1", co, ci);
    r.temb_off = m.temb_total;
    m.temb_regs.push_back({p + ".time_emb_proj", co, m.temb_total});
    m.temb_total += co;
    r.n2 = norm(p + ".norm2", co);
    r.c2 = conv3(p + ".conv2", co, co);
    r.has_sc = ci != co;
    if (r.has_sc) r.sc = lin(p + ".conv_shortcut", co, ci, true, true);
    return r;
  }

  VitW vit(const std::string& p, int c, int heads, int depth) {
    const GdfArch& a = m.arch;
    VitW v; v.c = c; v.heads = heads;
    v.gn = norm(p + ".norm", c);
    v.pin = lin(p + ".proj_in", c, c, true, !a.use_linear_projection);
    for (int i = 0; i < depth; ++i) {
      const std::string b = p + ".transformer_blocks." + std::to_string(i);
      BlockW w;
      w.ln1 = norm(b + ".norm1", c);
      w.qkv = lin_alloc(3 * c, c, false);
      lin_rows(b + ".attn1.to_q", w.qkv, c, 0, false, false);
      lin_rows(b + ".attn1.to_k", w.qkv, c, c, false, false);
      lin_rows(b + ".attn1.to_v", w.qkv, c, 2 * c, false, false);
      w.o1 = lin(b + ".attn1.to_out.0", c, c);
      w.ln2 = norm(b + ".norm2", c);
      w.q2 = lin(b + ".attn2.to_q", c, c, false);
      {  // slot inside the contiguous group of this width (grouped GEMM over all blocks at plan start)
        int gi = 0;
        for (; gi < (int)m.kv_groups.size(); ++gi) if (m.kv_groups[gi].C == c) break;
        KvGroup& g = m.kv_groups[gi];
        w.kv_group = gi; w.kv_index = g.next++;
        w.kv2.n = 2 * c; w.kv2.k = a.cross_attention_dim; w.kv2.has_bias = false; w.kv2.b = NPOS;
        w.kv2.w = g.base + (size_t)w.kv_index * g.stride;
      }
      lin_rows(b + ".attn2.to_k", w.kv2, c, 0, false, false);
      lin_rows(b + ".attn2.to_v", w.kv2, c, c, false, false);
      w.o2 = lin(b + ".attn2.to_out.0", c, c);
      w.ln3 = norm(b + ".norm3", c);
      w.ff1 = lin_alloc(8 * c, c, true);
      reg(b + ".ff.net.0.proj.weight", {8 * c, c}, PK_ROWS_GEGLU, w.ff1.w, 8 * c, c);
      reg(b + ".ff.net.0.proj.bias", {8 * c}, PK_VEC_GEGLU, w.ff1.b, 8 * c);
      w.ff2 = lin(b + ".ff.net.2", c, 4 * c);
      v.blocks.push_back(w);
    }
    v.pout = lin(p + ".proj_out", c, c, true, !a.use_linear_projection);
    return v;
  }

  void build() {
    const GdfArch& a = m.arch;
    const int L = a.n_levels, nl = a.layers_per_block, te = a.time_embed_dim;
    const int* boc = a.block_out_channels;
    // conv_in: [C0][16 taps][8 ch] (9 real taps, Cin real channels)
    m.conv_in.cin = a.in_channels; m.conv_in.cout = boc[0];
    m.conv_in.w = take((size_t)boc[0] * 128 * 2); m.conv_in.b = take(boc[0] * 4);
    reg("conv_in.weight", {boc[0], a.in_channels, 3, 3}, PK_CONV_IN, m.conv_in.w, boc[0], a.in_channels);
    reg("conv_in.bias", {boc[0]}, PK_VEC, m.conv_in.b);
    m.te1 = lin("time_embedding.linear_1", te, boc[0]);
    m.te2 = lin("time_embedding.linear_2", te, te);
    if (a.addition_embed_text_time) {
      m.ae1 = lin("add_embedding.linear_1", te, a.add_in_dim);
      m.ae2 = lin("add_embedding.linear_2", te, te);
    }
    // contiguous arenas for the cross-attention K/V projection weights, one per channel width
    for (int lv = 0; lv < L; ++lv) {
      if (!a.has_attn[lv] && lv != L - 1) continue;
      int cnt = 0;
      if (a.has_attn[lv]) cnt += (nl + (nl + 1)) * a.transformer_layers[lv];
      if (lv == L - 1) cnt += a.transformer_layers[lv];
      if (!cnt) continue;
      KvGroup* g = nullptr;
      for (auto& x : m.kv_groups) if (x.C == boc[lv]) g = &x;
      if (!g) { m.kv_groups.push_back(KvGroup{}); g = &m.kv_groups.back(); g->C = boc[lv]; }
      g->count += cnt;
    }
    for (auto& g : m.kv_groups) {
      g.stride = align_up((size_t)2 * g.C * a.cross_attention_dim * 2, 256);
      g.base = take(g.stride * g.count);
    }
    int ci = boc[0];
    for (int lv = 0; lv < L; ++lv) {
      LevelW lw;
      const int co = boc[lv];
      for (int r = 0; r < nl; ++r) {
        lw.res.push_back(resnet("down_blocks." + std::to_string(lv) + ".resnets." + std::to_string(r), ci, co));
        if (a.has_attn[lv])
          lw.vit.push_back(vit("down_blocks." + std::to_string(lv) + ".attentions." + std::to_string(r), co, a.heads[lv],
                               a.transformer_layers[lv]));
        ci = co;
      }
      lw.has_sampler = lv != L - 1;
      if (lw.has_sampler) lw.sampler = conv3("down_blocks." + std::to_string(lv) + ".downsamplers.0.conv", co, co);
      m.down.push_back(lw);
    }
    const int cm = boc[L - 1];
    m.mid_res0 = resnet("mid_block.resnets.0", cm, cm);
    m.mid_vit = vit("mid_block.attentions.0", cm, a.heads[L - 1], a.transformer_layers[L - 1]);
    m.mid_res1 = resnet("mid_block.resnets.1", cm, cm);
    int prev = boc[L - 1];
    for (int i = 0; i < L; ++i) {
      LevelW lw;
      const int lv = L - 1 - i;
      const int co = boc[lv];
      const int cin_skip = boc[std::max(lv - 1, 0)];
      for (int r = 0; r < nl + 1; ++r) {
        const int skip_c = (r == nl) ? cin_skip : co;
        const int in_c = (r == 0) ? prev : co;
        lw.res.push_back(resnet("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(r), in_c + skip_c, co));
        lw.skip_c.push_back(skip_c);
        if (a.has_attn[lv])
          lw.vit.push_back(vit("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(r), co, a.heads[lv],
                               a.transformer_layers[lv]));
      }
      lw.has_sampler = i != L - 1;
      if (lw.has_sampler) lw.sampler = conv3("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co);
      m.up.push_back(lw);
      prev = co;
    }
    m.norm_out = norm("conv_norm_out", boc[0]);
    m.conv_out = conv3("conv_out", a.out_channels, boc[0]);
    // stacked time_emb_proj: one [sum Cout][te] matrix -> a single launch per forward
    m.temb_all = lin_alloc(m.temb_total, te, true);
    for (auto& t : m.temb_regs) lin_rows(t.name, m.temb_all, t.cout, t.off, false, true);
    m.weight_bytes = cur;
  }
};

Model* model_create(const GdfArch& arch) {
  if (arch.n_levels < 2 || arch.n_levels > 4) { set_error("n_levels must be 2..4"); return nullptr; }
  for (int i = 0; i < arch.n_levels; ++i) {
    if (arch.block_out_channels[i] % 64) { set_error("block_out_channels must be multiples of 64"); return nullptr; }
    if (arch.has_attn[i]) {
      const int d = arch.block_out_channels[i] / std::max(1, arch.heads[i]);
      if (!(d == 32 || d == 40 || d == 64 || d == 80 || d == 160)) { set_error("unsupported head dim"); return nullptr; }
    }
  }
  if (arch.cross_attention_dim % 64 || arch.time_embed_dim % 8 || arch.in_channels > 8) {
    set_error("unsupported cross_attention_dim / time_embed_dim / in_channels"); return nullptr;
  }
  if (arch.addition_embed_text_time && (arch.add_in_dim % 8)) { set_error("add_in_dim % 8"); return nullptr; }
  Model* m = new Model();
  m->arch = arch;
  ModelBuilder b(*m);
  b.build();
  {
    CaptureExclusive g;
    if (hipMalloc(&m->weights, m->weight_bytes) != hipSuccess) {
      set_error("hipMalloc(weights) failed"); delete m; return nullptr;
    }
    hipMemset(m->weights, 0, m->weight_bytes);
  }
  // hook ids (dry plan walk)
  PlanOpts o{}; o.stream_fp32 = 1;
  Plan dry;
  plan_build(*m, dry, 1, 8 << (arch.n_levels - 1), 8 << (arch.n_levels - 1), 8, nullptr, 0, o, /*dry=*/true);
  m->hook_names = dry.dry_ids;
  return m;
}

static std::shared_mutex& capture_mx() { static std::shared_mutex mx; return mx; }
static bool capture_guard_on() { static const bool on = [] { const char* e = getenv("GDF_CAPTURE_GUARD"); return !e || atoi(e) != 0; }(); return on; }   // 0: diagnostics (tests/test_gpu_dist.py reproduces the invalidation)
CaptureShared::CaptureShared() { if (capture_guard_on()) capture_mx().lock_shared(); }
CaptureShared::~CaptureShared() { if (capture_guard_on()) capture_mx().unlock_shared(); }
CaptureExclusive::CaptureExclusive() { if (capture_guard_on()) capture_mx().lock(); }
CaptureExclusive::~CaptureExclusive() { if (capture_guard_on()) capture_mx().unlock(); }

void model_destroy(Model* m) {
  if (!m) return;
  if (m->weights) { CaptureExclusive g; hipFree(m->weights); }
  delete m;
}

// GEGLU projections are stored with rows interleaved [16 h | 16 gate] so the GEMM epilogue can gate in registers
static int geglu_group(int) { return 16; }

int model_set_param(Model* m, const char* name, const void* src, int dtype, hipStream_t s) {
  auto it = m->index.find(name);
  if (it == m->index.end()) { set_error(std::string("unknown parameter: ") + name); return GDF_ERR_ARG; }
  ParamRec& p = m->params[it->second];
  if (dtype != GDF_F16 && dtype != GDF_F32 && dtype != GDF_BF16) { set_error("dtype must be GDF_F16, GDF_F32 or GDF_BF16"); return GDF_ERR_ARG; }
  const int f32 = dtype;                       // source dtype code of the relayout kernels: 0 fp16, 1 fp32, 2 bf16
  char* base = (char*)m->weights;
  hipError_t e = hipSuccess;
  switch (p.kind) {
    case PK_VEC: e = launch_relayout_vec(src, f32, (float*)(base + p.dst), (int)p.shape[0], 0, 0, s); break;
    case PK_VEC_OFF: e = launch_relayout_vec(src, f32, (float*)(base + p.dst), p.a0, p.a1, 0, s); break;
    case PK_VEC_GEGLU: e = launch_relayout_vec(src, f32, (float*)(base + p.dst), p.a0, 0, geglu_group(p.a0), s); break;
    case PK_CONV3: e = launch_relayout_conv(src, f32, (half_t*)(base + p.dst), p.a0, p.a1, 9, p.a1, 9, s, 64); break;
    case PK_CONV_IN: e = launch_relayout_conv(src, f32, (half_t*)(base + p.dst), p.a0, p.a1, 9, 8, 16, s); break;
    case PK_ROWS:
      e = launch_relayout_rows(src, f32, (half_t*)(base + p.dst), p.a0, p.a1, p.a2, 0, s, m->bf16);
      if (e == hipSuccess && m->fp8 && (p.a1 % 128) == 0)       // 'fp8-mx' plans: the e4m3 copy + per-output-channel scales of these rows
        e = launch_quant_rows_fp8((const half_t*)(base + p.dst) + (size_t)p.a2 * p.a1, p.a1, p.a0, p.a1, m->bf16,
                                  (unsigned char*)(base + m->f8_off + p.dst / 2) + (size_t)p.a2 * p.a1, p.a1,
                                  (float*)(base + m->sc_off + p.dst / 16) + p.a2, s);
      break;
    case PK_ROWS_PADK: e = launch_relayout_rows_padk(src, f32, (half_t*)(base + p.dst), p.a0, p.a1, p.a2, s); break;
    case PK_ROWS_GEGLU: e = launch_relayout_rows(src, f32, (half_t*)(base + p.dst), p.a0, p.a1, 0, geglu_group(p.a0), s); break;
    default: set_error("bad param kind"); return GDF_ERR_STATE;
  }
  if (e != hipSuccess) { set_error(std::string("relayout launch failed: ") + hipGetErrorString(e)); return GDF_ERR_HIP; }
  if (!p.set) { p.set = true; m->n_set++; }
  return GDF_OK;
}

// =====================================================================================================
// Plan builder
// =====================================================================================================
namespace {


struct B : PlanBuilder {   // UNet op program
  B(const Model& mm, Plan& pp, bool d, const PlanOpts& o) : PlanBuilder(mm, pp, d, o) {}
  std::vector<std::pair<size_t, size_t>> kv_bufs;   // per KvGroup: (workspace offset, bytes per block) of the text K/V

  // ---- attention helper ---------------------------------------------------------------------------
  void attention(const char* name, Ref q, int ldq, Ref k, int ldk, Ref v, int ldv, Ref o, int ldo, int heads, int Sq,
                 int Sk, int D, int map_slot, int kv_rows_per_batch = -1, int o_lo = 0, int q_lo = 0, int kv_lo = 0) {
    if (kv_rows_per_batch < 0) kv_rows_per_batch = Sk;
    const int Bq = Bn;
    const double fl = 4.0 * (double)Bn * heads * Sq * Sk * D;
    op(name, fl, [=](const Bind& b, hipStream_t s) {
      AttnParams a{};
      a.q = (const half_t*)b.p(q); a.ldq = ldq; a.k = (const half_t*)b.p(k); a.ldk = ldk;
      a.v = (const half_t*)b.p(v); a.ldv = ldv; a.o = (half_t*)b.p(o); a.ldo = ldo;
      a.B = Bq; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D; a.scale = 1.0f / sqrtf((float)D);
      a.kv_bstride = kv_rows_per_batch; a.o_lo = o_lo; a.q_lo = q_lo; a.kv_lo = kv_lo;
      a.map = map_slot >= 0 ? (half_t*)b.hook(map_slot) : nullptr;
      return launch_attention(a, s);
    });
    if (map_slot >= 0) hook_done();
  }

  // ---- Transformer2DModel ---------------------------------------------------------------------------
  void vit(const std::string& id, const VitW& w, const Act& x, Act& y) {
    if (stop) return;
    const size_t n = rows(x);
    const int C = x.C, S = x.H * x.W, heads = w.heads, D = C / heads;
    const bool maps = P.want_maps;
    // GroupNorm(eps 1e-6) -> proj_in  (conv1x1 == linear in NHWC)
    // precise plans: every GEMM A operand below is a split image [hi | lo] (row width 2K, lo at +K): builder.h gemm(..., a_lo)
    // split operand classes (builder.h SP_*): s_x = 1 when the class is stored as pairs, p_x = its row-width factor
    const int s_gnv = spl(SP_GNV), s_lna = spl(SP_LN_ATTN), s_ao = spl(SP_ATTN_OUT), s_ao2 = spl(SP_ATTN2_OUT), s_lnf = spl(SP_LN_FF), s_inn = spl(SP_FF_INNER);
    const int p_gnv = 1 + s_gnv, p_lna = 1 + s_lna, p_ao = 1 + s_ao, p_ao2 = 1 + s_ao2, p_lnf = 1 + s_lnf, p_inn = 1 + s_inn;
    const size_t gn = groupnorm(x, w.gn, 1e-6f, false, SP_GNV);
    Act tok = new_act(C, x.H, x.W, true);
    {
      Epi e; e.bias = wt(w.pin.b); e.has_bias = true; out_to(e, tok, /*need_shadow=*/w.blocks.empty());
      gemm("proj_in", ws(gn), C * p_gnv, n, w.pin, C, C, 0, e, s_gnv * C);
    }
    untmp(gn, img_bytes(n, C, SP_GNV));
    for (size_t bi = 0; bi < w.blocks.size() && !stop; ++bi) {
      const BlockW& bw = w.blocks[bi];
      const std::string bid = id + "-block" + std::to_string(bi);
      const size_t nb = n * C * 2;            // a plain fp16 [n][C] tensor (q, hooks)
      const size_t nb_lna = img_bytes(n, C, SP_LN_ATTN), nb_ao = img_bytes(n, C, SP_ATTN_OUT), nb_ao2 = img_bytes(n, C, SP_ATTN2_OUT), nb_lnf = img_bytes(n, C, SP_LN_FF);
      // --- self attention ---
      size_t ln = layernorm(tok, bw.ln1, SP_LN_ATTN);
      // SP_QKV: rows [q | k | v | q_lo | k_lo | v_lo] (the GEMM's pair output), the hooks read the hi halves = fp16(q), as always
      const int s_qkv = spl(SP_QKV), lq = 3 * C * (1 + s_qkv);
      const size_t qkv = tmp(n * (size_t)lq * 2);
      { Epi e; e.out16 = ws(qkv); e.has_o16 = true; e.ldo16 = lq; e.o16_lo = s_qkv * 3 * C;
        gemm("attn1_qkv", ws(ln), C * p_lna, n, bw.qkv, 3 * C, C, 0, e, s_lna * C); }
      untmp(ln, nb_lna);
      hook_copy(want(bid + "-self-q", C, x.H, x.W), ws(qkv), lq, n, C);                 // attention_processor.py:3291-3294
      hook_copy(want(bid + "-self-k", C, x.H, x.W), ws(qkv + (size_t)C * 2), lq, n, C);
      hook_copy(want(bid + "-self-v", C, x.H, x.W), ws(qkv + (size_t)2 * C * 2), lq, n, C);
      size_t ao = tmp(nb_ao);
      const int ms = maps ? want_map(bid + "-self-map", heads, S, S) : (dry_map(bid + "-self-map"), -1);
      attention("attn1", ws(qkv), lq, ws(qkv + (size_t)C * 2), lq, ws(qkv + (size_t)2 * C * 2), lq, ws(ao), C * p_ao, heads,
                S, S, D, ms, -1, s_ao * C, s_qkv * 3 * C, s_qkv * 3 * C);
      untmp(qkv, n * (size_t)lq * 2);
      { Epi e; e.bias = wt(bw.o1.b); e.has_bias = true; residual_from(e, tok); out_to(e, tok, false);
        gemm("attn1_out", ws(ao), C * p_ao, n, bw.o1, C, C, 0, e, s_ao * C); }
      untmp(ao, nb_ao);
      if (stop) break;
      // --- cross attention ---
      ln = layernorm(tok, bw.ln2, SP_LN_ATTN);
      // a hooked `cross-q` / `ffn-inner` is a whole contiguous tensor with one producer: the GEMM writes it straight into
      // the caller's hook buffer and the consumer reads it from there (no workspace copy, no hook_store pass)
      // (SP_XQKV: the query is a pair [q | q_lo] in workspace and a hooked `cross-q` a copy of its hi half)
      const int s_xq = spl(SP_XQKV);
      const int hq = want(bid + "-cross-q", C, x.H, x.W);
      const bool q2_direct = hq >= 0 && !s_xq;
      const size_t q2 = q2_direct ? 0 : tmp(nb * (1 + s_xq));
      const Ref q2r = q2_direct ? Ref{BUF_HOOK0 + hq, 0} : ws(q2);
      { Epi e; e.out16 = q2r; e.has_o16 = true; e.ldo16 = C * (1 + s_xq); e.o16_lo = s_xq * C;
        gemm("attn2_q", ws(ln), C * p_lna, n, bw.q2, C, C, 0, e, s_lna * C); }
      untmp(ln, nb_lna);
      if (q2_direct) hook_done();
      else if (hq >= 0) hook_copy(hq, ws(q2), C * (1 + s_xq), n, C);
      // text K/V: precomputed for all blocks by the grouped GEMM at the head of the plan; with one prompt repeated over
      // the batch (reference diffusion_feature.py:272, opts.reserved[0]) there is a single K/V set per block
      const bool shared = opt.reserved[0] != 0;
      const size_t kv = dry ? 0 : kv_bufs[bw.kv_group].first + (size_t)bw.kv_index * kv_bufs[bw.kv_group].second;
      ao = tmp(nb_ao2);
      const int mc = maps ? want_map(bid + "-cross-map", heads, S, n_ctx) : (dry_map(bid + "-cross-map"), -1);
      // text K / V rows: [k | v] or, SP_XQKV, [k | v | k_lo | v_lo] (the grouped GEMM's pair output)
      attention("attn2", q2r, C * (1 + s_xq), ws(kv), 2 * C * (1 + s_xq), ws(kv + (size_t)C * 2), 2 * C * (1 + s_xq), ws(ao), C * p_ao2, heads, S, n_ctx, D, mc,
                shared ? 0 : n_ctx, s_ao2 * C, s_xq * C, s_xq * 2 * C);
      if (!q2_direct) untmp(q2, nb * (1 + s_xq));
      { Epi e; e.bias = wt(bw.o2.b); e.has_bias = true; residual_from(e, tok); out_to(e, tok, false);
        gemm("attn2_out", ws(ao), C * p_ao2, n, bw.o2, C, C, 0, e, s_ao2 * C); }
      untmp(ao, nb_ao2);
      if (stop) break;
      // --- feed forward (GEGLU) ---
      ln = layernorm(tok, bw.ln3, SP_LN_FF);
      const int hi = want(bid + "-ffn-inner", 4 * C, x.H, x.W);                             // attention.py:1255-1257
      // (split inner tensor: a pair in workspace; a hooked `ffn-inner` is then a copy of its hi half)
      const bool direct = hi >= 0 && !s_inn;
      const size_t inner_b = img_bytes(n, 4 * C, SP_FF_INNER);
      const size_t inner = direct ? 0 : tmp(inner_b);
      const Ref innr = direct ? Ref{BUF_HOOK0 + hi, 0} : ws(inner);
      { Epi e; e.bias = wt(bw.ff1.b); e.has_bias = true; e.geglu = geglu_group(8 * C); e.out16 = innr; e.has_o16 = true; e.ldo16 = 4 * C * p_inn;
        e.o16_lo = s_inn * 4 * C;
        gemm("ff_geglu", ws(ln), C * p_lnf, n, bw.ff1, 8 * C, C, 0, e, s_lnf * C); }
      untmp(ln, nb_lnf);
      if (direct) hook_done();
      else if (hi >= 0) hook_copy(hi, innr, 4 * C * p_inn, n, 4 * C);
      { // the fp16 image of the block output is only needed by the `blockN-out` hook and by proj_out (last block)
        const bool shadow = (bi + 1 == w.blocks.size()) || (!dry && P.requested.count(bid + "-out"));
        Epi e; e.bias = wt(bw.ff2.b); e.has_bias = true; residual_from(e, tok); out_to(e, tok, shadow);
        gemm("ff_out", innr, 4 * C * p_inn, n, bw.ff2, C, 4 * C, 0, e, s_inn * 4 * C); }
      if (!direct) untmp(inner, inner_b);
      gather(bid + "-out", tok);                                                           // attention.py:589-590
    }
    if (!stop) {
      Epi e; e.bias = wt(w.pout.b); e.has_bias = true; residual_from(e, x); out_to(e, y);
      gemm("proj_out", tok.h, tok.ld, n, w.pout, C, C, 0, e, tok.lo);
    }
    free_act(tok);
    gather(id + "-out", y);                                                                // transformer_2d.py:474-475
  }
  void dry_map(const std::string& id) { if (dry) P.dry_ids.push_back(id); }

  size_t layernorm(const Act& x, const NormW& w, int out_cls) {
    const size_t n = rows(x);
    const size_t y = tmp(img_bytes(n, x.C, out_cls));
    const Ref xh = x.h, xf = x.f; const bool hf = x.has_f; const int ld = x.ld, C = x.C;
    const Ref g = wt(w.g), bt = wt(w.b);
    const int ldy = C * pxc(out_cls), y_lo = spl(out_cls) ? C : 0;       // split output [hi | lo] when its operand class is split
    if (spl(out_cls) && !hf && !dry) { set_error("split LayerNorm output without an fp32 master of the input"); bad = true; }
    op("layernorm", 0, [=](const Bind& b, hipStream_t s) {
      return launch_layernorm(hf ? nullptr : (const half_t*)b.p(xh), hf ? (const float*)b.p(xf) : nullptr, hf ? C : ld,
                              (int)n, C, 1e-5f, (const float*)b.p(g), (const float*)b.p(bt), (half_t*)b.ws(y), s, ldy, y_lo);
    });
    return y;
  }

  // ---- whole UNet ---------------------------------------------------------------------------------
  void build(int H, int W) {
    const GdfArch& a = m.arch;
    const int L = a.n_levels, nl = a.layers_per_block, te = a.time_embed_dim;
    const int* boc = a.block_out_channels;
    const int Bq = Bn;

    // ---- time / additional embeddings (fp32 vectors) ----
    const size_t tsin = tmp((size_t)Bn * boc[0] * 4), t1 = tmp((size_t)Bn * te * 4), emb = tmp((size_t)Bn * te * 4);
    const size_t tall_b = (size_t)Bn * m.temb_total * 4;
    const size_t tall = tmp(tall_b);
    temb_all = ws(tall);
    {
      const int c0 = boc[0];
      const Ref w1 = wt(m.te1.w), b1 = wt(m.te1.b), w2 = wt(m.te2.w), b2 = wt(m.te2.b);
      op("time_embed", 0, [=](const Bind& b, hipStream_t s) {
        hipError_t e = launch_sinusoid((const float*)b.base[BUF_T], Bq, 1, c0, (float*)b.ws(tsin), c0, 0, 0, s);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(tsin), c0, Bq, c0, (const half_t*)b.p(w1), (const float*)b.p(b1), te, 0, 0,
                                (float*)b.ws(t1), te, s);
        if (e != hipSuccess) return e;
        return launch_small_linear((const float*)b.ws(t1), te, Bq, te, (const half_t*)b.p(w2), (const float*)b.p(b2), te, 1, 0,
                                   (float*)b.ws(emb), te, s);
      });
      if (a.addition_embed_text_time) {
        const int ain = a.add_in_dim, atd = a.addition_time_embed_dim, pooled = ain - 6 * atd;
        const size_t av = tmp((size_t)Bn * ain * 4), a1 = tmp((size_t)Bn * te * 4);
        const Ref aw1 = wt(m.ae1.w), ab1 = wt(m.ae1.b), aw2 = wt(m.ae2.w), ab2 = wt(m.ae2.b);
        op("add_embed", 0, [=](const Bind& b, hipStream_t s) {
          if (!b.base[BUF_TXT] || !b.base[BUF_TID]) return hipErrorInvalidValue;
          hipError_t e = launch_widen((const half_t*)b.base[BUF_TXT], Bq, pooled, (float*)b.ws(av), ain, 0, s);
          if (e != hipSuccess) return e;
          e = launch_sinusoid((const float*)b.base[BUF_TID], Bq, 6, atd, (float*)b.ws(av), ain, pooled, 0, s);
          if (e != hipSuccess) return e;
          e = launch_small_linear((const float*)b.ws(av), ain, Bq, ain, (const half_t*)b.p(aw1), (const float*)b.p(ab1), te, 0, 0,
                                  (float*)b.ws(a1), te, s);
          if (e != hipSuccess) return e;
          return launch_small_linear((const float*)b.ws(a1), te, Bq, te, (const half_t*)b.p(aw2), (const float*)b.p(ab2), te, 1,
                                     1, (float*)b.ws(emb), te, s);
        });
        untmp(av, (size_t)Bn * ain * 4); untmp(a1, (size_t)Bn * te * 4);
      }
      const Ref tw = wt(m.temb_all.w), tb = wt(m.temb_all.b);
      const int tt = m.temb_total;
      op("temb_proj_all", 0, [=](const Bind& b, hipStream_t s) {
        return launch_small_linear((const float*)b.ws(emb), te, Bq, te, (const half_t*)b.p(tw), (const float*)b.p(tb), tt, 1, 0,
                                   (float*)b.ws(tall), tt, s);
      });
    }

    // ---- text K/V of every transformer block: one grouped GEMM per channel width (blockIdx.y = block) ----
    kv_bufs.clear();
    {
      const bool shared = opt.reserved[0] != 0;
      const size_t nkv = (size_t)(shared ? 1 : Bn) * n_ctx;
      for (const KvGroup& g : m.kv_groups) {
        const int kvp = spl(SP_XQKV) ? 2 : 1;                     // SP_XQKV: rows [k | v | k_lo | v_lo]
        const size_t per = align_up(nkv * 2 * g.C * 2 * kvp, 256);
        const size_t off = tmp(per * g.count);
        kv_bufs.push_back({off, per});
        const Ref W = wt(g.base);
        const int N = 2 * g.C, K = a.cross_attention_dim, cnt = g.count;
        const long wst = (long)(g.stride / 2), ost = (long)(per / 2);
        op("attn2_kv", 2.0 * (double)nkv * N * K * cnt, [=](const Bind& b, hipStream_t s) {
          GemmParams gp{};
          gp.A = (const half_t*)b.base[BUF_CTX]; gp.lda = K; gp.a_bytes = (uint32_t)(nkv * K * 2);
          gp.M = (int)nkv; gp.N = N; gp.K = K; gp.mode = A_DENSE;
          gp.Wt = (const half_t*)b.p(W); gp.w_bytes = (uint32_t)((size_t)N * K * 2);
          gp.out16 = (half_t*)b.ws(off); gp.ldo16 = N * kvp; gp.o16_lo = kvp == 2 ? N : 0; gp.bn = 128; gp.rows_per_sample = 1;
          gp.batch = cnt; gp.w_bstride = wst; gp.o_bstride = ost;
          return launch_gemm(gp, s);
        });
      }
    }

    // ---- concat buffers of the up path: cat([h, skip]) laid out in place --------------------------
    // skip producers in order: conv_in, every down resnet(+vit) output, every downsampler output
    const int px = pxc(SP_STREAM); const bool sps = spl(SP_STREAM);
    struct Cat { size_t off, bytes; int ch, cs, H, W; };
    std::vector<Cat> cats;        // in up-path consumption order
    {
      int prev = boc[L - 1];
      int hh = H >> (L - 1), ww = W >> (L - 1);
      for (int i = 0; i < L; ++i) {
        const int lv = L - 1 - i, co = boc[lv], cin_skip = boc[std::max(lv - 1, 0)];
        for (int r = 0; r < nl + 1; ++r) {
          Cat c; c.ch = (r == 0) ? prev : co; c.cs = (r == nl) ? cin_skip : co; c.H = hh; c.W = ww;
          c.bytes = (size_t)Bn * hh * ww * (c.ch + c.cs) * 2 * px;          // split stream images: [h_hi | skip_hi | h_lo | skip_lo]
          c.off = tmp(c.bytes);
          cats.push_back(c);
        }
        prev = co; hh *= 2; ww *= 2;
      }
    }
    int n_skips = (int)cats.size();   // == number of skip tensors
    int skip_idx = 0;                 // k-th produced skip is consumed by cats[n_skips-1-k]
    auto skip_dst = [&](int C, int hh, int ww) -> Act {
      const Cat& c = cats[n_skips - 1 - skip_idx++];
      return view_act(ws(c.off + (size_t)c.ch * 2), (c.ch + c.cs) * px, C, hh, ww, true, sps ? c.ch + c.cs : 0);
    };

    // ---- conv_in ----
    const size_t lat8_b = (size_t)Bn * H * W * 16;
    const size_t lat8 = tmp(lat8_b);
    {
      const int slot = want("unet-in", a.in_channels, H, W);                                // unet_2d_condition.py:1169-1170
      const int cin = a.in_channels;
      op("pack_latents", 0, [=](const Bind& b, hipStream_t s) {
        return launch_pack_latents((const half_t*)b.base[BUF_LAT], Bq, cin, H, W, (half_t*)b.ws(lat8),
                                   slot >= 0 ? (half_t*)b.hook(slot) : nullptr, s);
      });
      if (slot >= 0) hook_done();
    }
    Act cur = skip_dst(boc[0], H, W);
    if (!stop) {
      Epi e; e.bias = wt(m.conv_in.b); e.has_bias = true; out_to(e, cur);
      Epi ee = e;
      const Ref Wr = wt(m.conv_in.w); const int N = boc[0];
      const size_t M = (size_t)Bn * H * W;
      GemmParams gk{}; gk.M = (int)M; gk.N = N; gk.K = 128; gk.mode = A_CONV_SMALLC; gk.bn = ee.bn; gk.o16_lo = ee.has_o16 ? ee.o16_lo : 0;
      op("conv_in", 2.0 * (double)M * N * 9 * a.in_channels, [=](const Bind& b, hipStream_t s) {
        GemmParams g{};
        g.A = (const half_t*)b.ws(lat8); g.lda = 8; g.a_bytes = (uint32_t)(M * 16);
        g.M = (int)M; g.N = N; g.K = 128; g.mode = A_CONV_SMALLC; g.H = H; g.W = W; g.OH = H; g.OW = W; g.stride = 1; g.Cin = 8;
        g.Wt = (const half_t*)b.p(Wr); g.w_bytes = (uint32_t)((size_t)N * 128 * 2);
        fill_epi(g, ee, b);
        return launch_gemm(g, s);
      }, gk.o16_lo > 0 ? gemm_kernel_name(gk) : nullptr);
    }
    untmp(lat8, lat8_b);
    gather("unet-after-conv-in", cur);                                                      // :1172-1173

    // ---- down path ----
    int hh = H, ww = W;
    for (int lv = 0; lv < L && !stop; ++lv) {
      const LevelW& lw = m.down[lv];
      for (int r = 0; r < nl && !stop; ++r) {
        const std::string id = "down-level" + std::to_string(lv) + "-repeat" + std::to_string(r);
        const bool attn = a.has_attn[lv];
        if (attn) {
          Act mid = new_act(boc[lv], hh, ww, true);
          resnet(id, lw.res[r], cur, mid);
          free_master(cur);
          Act nxt = skip_dst(boc[lv], hh, ww);
          vit(id + "-vit", lw.vit[r], mid, nxt);
          free_act(mid);
          cur = nxt;
        } else {
          Act nxt = skip_dst(boc[lv], hh, ww);
          resnet(id, lw.res[r], cur, nxt);
          free_master(cur);
          cur = nxt;
        }
      }
      if (lw.has_sampler && !stop) {
        Act nxt = skip_dst(boc[lv], hh / 2, ww / 2);
        Epi e; e.bias = wt(lw.sampler.b); e.has_bias = true; out_to(e, nxt);
        conv3("downsample", cur.h, cur.ld, cur.C, hh, ww, 2, false, lw.sampler, e, spl(SP_SAMPLER) ? cur.lo : 0);  // downsampling.py:132-152
        free_master(cur);
        cur = nxt; hh /= 2; ww /= 2;
        gather("down-level" + std::to_string(lv) + "-downsampler-out", cur);
      }
    }
    // ---- mid ----
    if (!stop) {
      Act a0 = new_act(boc[L - 1], hh, ww, true);
      resnet("mid-repeat0", m.mid_res0, cur, a0);
      free_master(cur);
      Act a1 = new_act(boc[L - 1], hh, ww, true);
      vit("mid-vit", m.mid_vit, a0, a1);
      free_act(a0);
      // mid output feeds cats[0] channel slice [0, ch)
      Act a2 = view_act(ws(cats[0].off), (cats[0].ch + cats[0].cs) * px, boc[L - 1], hh, ww, false, sps ? cats[0].ch + cats[0].cs : 0);
      resnet("mid-repeat1", m.mid_res1, a1, a2);
      free_act(a1);
      cur = a2;
    }
    // ---- up path ----
    int ci = 0;
    for (int i = 0; i < L && !stop; ++i) {
      const LevelW& lw = m.up[i];
      const int lv = L - 1 - i;
      for (int r = 0; r < nl + 1 && !stop; ++r, ++ci) {
        const std::string id = "up-level" + std::to_string(i) + "-repeat" + std::to_string(r);
        const Cat& c = cats[ci];
        Act cat = view_act(ws(c.off), (c.ch + c.cs) * px, c.ch + c.cs, c.H, c.W, false, sps ? c.ch + c.cs : 0);   // torch.cat([h, skip], 1)
        // destination: the h-slice of the next concat buffer, or a fresh tensor at the end of a level
        const bool last_in_level = (r == nl);
        const bool attn = a.has_attn[lv];
        auto make_dst = [&](bool master) -> Act {
          if (!last_in_level) {
            const Cat& nc = cats[ci + 1];
            return view_act(ws(nc.off), (nc.ch + nc.cs) * px, boc[lv], c.H, c.W, master, sps ? nc.ch + nc.cs : 0);
          }
          return new_act(boc[lv], c.H, c.W, master);
        };
        if (attn) {
          Act mid = new_act(boc[lv], c.H, c.W, true);
          resnet(id, lw.res[r], cat, mid);
          Act nxt = make_dst(false);
          vit(id + "-vit", lw.vit[r], mid, nxt);
          free_act(mid);
          cur = nxt;
        } else {
          Act nxt = make_dst(false);
          resnet(id, lw.res[r], cat, nxt);
          cur = nxt;
        }
        untmp(c.off, c.bytes);
      }
      if (lw.has_sampler && !stop) {
        const Cat& nc = cats[ci];
        Act nxt = view_act(ws(nc.off), (nc.ch + nc.cs) * px, boc[lv], cur.H * 2, cur.W * 2, false, sps ? nc.ch + nc.cs : 0);
        Epi e; e.bias = wt(lw.sampler.b); e.has_bias = true; out_to(e, nxt);
        conv3("upsample", cur.h, cur.ld, cur.C, cur.H, cur.W, 1, true, lw.sampler, e, spl(SP_UPSAMPLER) ? cur.lo : 0);   // upsampling.py:176-193
        free_act(cur);
        cur = nxt;
        gather("up-level" + std::to_string(i) + "-upsampler-out", cur);
      }
    }
    // ---- out ----
    if (!stop) {
      const size_t n = rows(cur);
      const size_t no = groupnorm(cur, m.norm_out, 1e-5f, true, SP_OUT);                     // :1304-1306
      Epi e; e.bias = wt(m.conv_out.b); e.has_bias = true; e.bn = 16;
      e.out16 = Ref{BUF_NOISE, 0}; e.has_o16 = true; e.ldo16 = a.out_channels;
      P.writes_noise = true;
      conv3("conv_out", ws(no), cur.C * pxc(SP_OUT), cur.C, cur.H, cur.W, 1, false, m.conv_out, e, spl(SP_OUT) ? cur.C : 0);
      untmp(no, img_bytes(n, cur.C, SP_OUT));
      const int slot = want("unet-out", a.out_channels, cur.H, cur.W);                       // :1309-1310
      hook_copy(slot, Ref{BUF_NOISE, 0}, a.out_channels, n, a.out_channels);
      free_act(cur);
    }
    (void)tsin; (void)t1; (void)emb;
  }
};

}  // namespace

int plan_build(const Model& m, Plan& P, int batch, int H, int W, int n_ctx, const char* const* ids, int n_ids,
               const PlanOpts& opts, bool dry) {
  const int L = m.arch.n_levels;
  if (batch < 1 || H < 1 || W < 1 || (H % (1 << (L - 1))) || (W % (1 << (L - 1)))) {
    set_error("latent size must be a positive multiple of 2^(levels-1)"); return GDF_ERR_ARG;
  }
  const bool precise = opts.reserved[1] != 0;
  if (precise && !opts.stream_fp32) { set_error("a precise plan needs the fp32 master stream (stream_fp32 = 1)"); return GDF_ERR_ARG; }
  // 32-bit buffer offsets: the widest row of a level must stay < 2 GiB.  Widest rows: the GEGLU inner tensor (4C, doubled when ITS class is
  // split), the fused q|k|v output (3C, doubled when SP_QKV is split) — both only where the level has attention — and the skip-concat buffers of the up path
  // (<= 3C, doubled when the STREAM class is split).  The limit follows the classes the mask actually splits (any non-zero mask
  // used to halve it, which refused SDXL 1024^2 B = 26..34 calls under the selective preset although their widest rows are not split).
  {
    const int split = opts.reserved[1] == 1 ? PlanBuilder::SP_ALL : ((opts.reserved[1] >> 8) & PlanBuilder::SP_ALL);
    for (int lv = 0; lv < L; ++lv) {
      const size_t r = (size_t)batch * (H >> lv) * (W >> lv), c = m.arch.block_out_channels[lv];
      size_t widest = 3 * ((split & PlanBuilder::SP_STREAM) ? 2 : 1);                                   // concat [hi | lo]
      if (m.arch.has_attn[lv] || lv == L - 1) widest = std::max<size_t>(widest, std::max<size_t>(3 * ((split & PlanBuilder::SP_QKV) ? 2 : 1), 4 * ((split & PlanBuilder::SP_FF_INNER) ? 2 : 1)));
      if (r * c * 2 * widest >= (1ull << 31)) {
        set_error("batch*H*W too large for 32-bit buffer offsets; split the batch"); return GDF_ERR_UNSUPPORTED;
      }
    }
  }
  P.batch = batch; P.H = H; P.W = W; P.n_ctx = n_ctx; P.opts = opts;
  B b(m, P, dry, opts);
  b.Bn = batch; b.n_ctx = n_ctx;
  // row N1's statistics half (round 5, ON by default): the resnet convs emit the GroupNorm partial sums of the fp16 image they store (gemm_body<..., GNS>:
  // one [slab of 64 / 128 rows][channel][sum, sum of squares] record per wave tile, now also on the 256x320 and 128x160 tiles the UNet's convs run on), and
  // the consuming norm2 / Transformer2DModel.norm runs gn_fold + gn_finalize instead of a statistics pass over the tensor (16 of SDXL's 26 statistics passes).
  // Same-box A/B, three alternations each (profiles/r05_ab_unet_gn_stats_from_conv_epilogue.txt): SDXL B = 16 +0.4-0.6 %, SD1.5 B = 32 +0.5-0.6 %.
  // GDF_UNET_GN_EPI=0 restores the separate passes.
  { static const bool on = [] { const char* e = getenv("GDF_UNET_GN_EPI"); return e ? atoi(e) != 0 : true; }(); b.gn_epi = on; }
  if (!dry) {
    std::unordered_set<std::string> known(m.hook_names.begin(), m.hook_names.end());
    for (int i = 0; i < n_ids; ++i)
      if (ids[i] && known.count(ids[i])) P.requested.insert(ids[i]);     // unknown ids silently ignored
    b.remaining = (int)P.requested.size();
    for (auto& s : P.requested) if (s.find("map") != std::string::npos) P.want_maps = true;   // diffusion_feature.py:72-77
    if (opts.early_exit && b.remaining == 0) b.stop = true;
  }
  b.build(H, W);
  if (b.bad) return GDF_ERR_UNSUPPORTED;
  P.ws_bytes = b.ar.peak + 256;
  return GDF_OK;
}

// kernel symbol of ops that do not go through gemm()/conv3() (those ask gemm_kernel_name() for the tile variant)
const char* kernel_label(const char* n) {
  auto is = [&](const char* x) { return strcmp(n, x) == 0; };
  if (is("conv_in")) return "gemm_kernel<2, 128, 128, 2, false>";
  if (is("attn2_kv")) return "gemm_kernel<0, 128, 128, 2, false>";
  if (is("attn1") || is("attn2") || is("joint_attn")) return "attn_kernel";
  if (is("adaln") || is("adaln_txt") || is("norm_out")) return "layernorm_mod_kernel";
  if (is("qk_norm_rope")) return "qk_norm_rope_kernel";
  if (is("layernorm")) return "layernorm_kernel";
  if (is("gn_stats")) return "gn_partial_kernel+gn_finalize_kernel";
  if (is("gn_apply") || is("gn_apply_silu")) return "gn_apply_kernel";
  if (is("gn_fused") || is("gn_fused_silu")) return "gn_fused_kernel";
  if (is("hook_store")) return "copy2d_kernel";
  return n;
}

Plan::~Plan() {
  CaptureExclusive guard;
  for (auto& v : ev) for (auto e : v) hipEventDestroy(e);
  for (auto& g : graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
}

static void timing_collect(Plan& P, int set) {
  if (!P.ev_used[set]) return;
  auto& v = P.ev[set];
  size_t k = 0;
  long nlab = 0;
  for (auto& op : P.ops) {
    if (op.label != P.timing_label) continue;
    if ((nlab++ % P.timing_stride) != 0) continue;
    float ms = 0.f;
    hipEventSynchronize(v[k + 1]);
    if (hipEventElapsedTime(&ms, v[k], v[k + 1]) == hipSuccess) { P.t_ms += ms; P.t_flops += op.flops; P.t_launches++; }
    k += 2;
  }
  P.ev_used[set] = false;
}

int plan_set_timing(Plan& P, const char* label) {
  for (int s = 0; s < Plan::EV_RING; ++s) timing_collect(P, s);
  P.timing_label = -1; P.t_ms = 0; P.t_flops = 0; P.t_launches = 0;
  if (!label) return GDF_OK;
  for (size_t i = 0; i < P.labels.size(); ++i) if (P.labels[i] == label) P.timing_label = (int)i;
  if (P.timing_label < 0) { set_error(std::string("no op with kernel label ") + label); return GDF_ERR_ARG; }
  size_t n = 0;
  for (auto& op : P.ops) n += op.label == P.timing_label;          // (an upper bound with a stride > 1)
  for (int s = 0; s < Plan::EV_RING; ++s)
    while (P.ev[s].size() < 2 * n) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate"); return GDF_ERR_HIP; } P.ev[s].push_back(e); }
  return GDF_OK;
}

int plan_read_timing(Plan& P, double* ms, long* launches, double* flops) {
  for (int s = 0; s < Plan::EV_RING; ++s) timing_collect(P, s);
  if (ms) *ms = P.t_ms;
  if (launches) *launches = P.t_launches;
  if (flops) *flops = P.t_flops;
  return GDF_OK;
}

int plan_forward(Plan& P, const Model& m, const void* lat, const float* t, const void* ctx, const void* txt,
                 const float* tid, void* const* hook_out, void* noise, void* ws, hipStream_t s, float* ms,
                 const char** names, double* flops, int cap) {
  if (m.kind != 0) { set_error("gdf_forward on a Flux model: use gdf_flux_forward"); return GDF_ERR_STATE; }
  if (m.n_set != (int)m.params.size()) { set_error("model weights incomplete"); return GDF_ERR_STATE; }
  Bind b;
  b.base[BUF_WS] = (char*)ws; b.base[BUF_WT] = (char*)m.weights; b.base[BUF_LAT] = (char*)lat; b.base[BUF_T] = (char*)t;
  b.base[BUF_CTX] = (char*)ctx; b.base[BUF_TXT] = (char*)txt; b.base[BUF_TID] = (char*)tid;
  b.hooks = hook_out;
  if (!lat || !t || !ctx || !ws) { set_error("null input pointer"); return GDF_ERR_ARG; }
  if (P.hooks.size() && !hook_out) { set_error("hook_out is null"); return GDF_ERR_ARG; }
  if (P.writes_noise && !noise) { set_error("noise_pred buffer required (the plan runs conv_out)"); return GDF_ERR_ARG; }
  b.base[BUF_NOISE] = (char*)noise;
  return plan_run(P, b, s, ms, names, flops, cap);
}

// An event-record NODE at the current point of the capture on `s` (fires at every replay).  Spelled with the graph API — current
// capture dependencies -> hipGraphAddEventRecordNode -> make the node the capture's dependency set — because
// hipEventRecordWithFlags(..., hipEventRecordExternal) returns "invalid argument" under the HIP runtime PyTorch 2.10 bundles
// (ROCm 7.0), while it works under /opt/rocm 7.2 (tools/micro/graph_events.hip).
static hipError_t record_in_capture(hipEvent_t ev, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, nullptr, &graph, &deps, &ndeps);
  if (e != hipSuccess) return e;
  if (st != hipStreamCaptureStatusActive || !graph) return hipErrorStreamCaptureInvalidated;
  hipGraphNode_t node = nullptr;
  e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev);
  if (e != hipSuccess) return e;
  return hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
}

// evset >= 0: record the timing events of set `evset` around every op of the timed label; `external` = inside a stream capture
// (event-record nodes that fire at every replay)
static int run_ops_eager(Plan& P, const Bind& b, hipStream_t s, int evset = -1, bool external = false) {
  size_t evk = 0;
  long nlab = 0;
  for (auto& op : P.ops) {
    const bool timed = evset >= 0 && op.label == P.timing_label && (nlab++ % P.timing_stride) == 0;
    if (timed) {
      const hipError_t ee = external ? record_in_capture(P.ev[evset][evk], s) : hipEventRecord(P.ev[evset][evk], s);
      if (ee != hipSuccess) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone; (void)hipStreamIsCapturing(s, &st);
        char buf[160]; snprintf(buf, sizeof buf, "hipEventRecord failed: %s (event %p, stream %p, capture status %d, evk %zu of %zu, external %d)",
                                hipGetErrorString(ee), (void*)P.ev[evset][evk], (void*)s, (int)st, evk, P.ev[evset].size(), (int)external);
        set_error(buf); return GDF_ERR_HIP;
      }
    }
    hipError_t e = op.fn(b, s);
    if (timed) {
      const hipError_t ee = external ? record_in_capture(P.ev[evset][evk + 1], s) : hipEventRecord(P.ev[evset][evk + 1], s);
      if (ee != hipSuccess && e == hipSuccess) { set_error("hipEventRecord failed"); return GDF_ERR_HIP; }
      evk += 2;
    }
    if (e != hipSuccess) { set_error(std::string("op '") + op.name + "' failed: " + hipGetErrorString(e)); return GDF_ERR_HIP; }
  }
  return GDF_OK;
}

// hipGraph path: the op program is captured once per distinct binding table on the caller's (non-default) stream and
// replayed with one hipGraphLaunch; ~2400 kernel launches per SDXL forward become one host call.
static int run_ops_graph(Plan& P, const Bind& b, hipStream_t s, int evset = -1) {
  const size_t nh = P.hooks.size();
  const int label = evset >= 0 ? P.timing_label : -1;
  for (auto& g : P.graphs) {
    bool same = g.evset == evset && g.label == label && memcmp(g.key.base, b.base, sizeof b.base) == 0 && memcmp(g.key.f, b.f, sizeof b.f) == 0 &&
                g.hook_ptrs.size() == nh;
    for (size_t i = 0; same && i < nh; ++i) same = g.hook_ptrs[i] == b.hooks[i];
    if (same) {
      g.stamp = ++P.graph_clock; ++P.graph_launches;
      if (hipGraphLaunch(g.exec, s) != hipSuccess) { set_error("hipGraphLaunch failed"); return GDF_ERR_HIP; }
      return GDF_OK;
    }
  }
  Plan::GraphEntry g;
  g.key = b; g.key.hooks = nullptr; g.evset = evset; g.label = label;
  for (size_t i = 0; i < nh; ++i) g.hook_ptrs.push_back(b.hooks[i]);
  // Relaxed mode: the op program only launches kernels (no allocation, no synchronisation), and in this mode HIP neither lists
  // the stream for its "unsafe call during capture" checks nor lets an unrelated call invalidate the capture — other host
  // threads (one extractor per thread: aggregation_network.py:67-95) keep allocating, synchronising and capturing freely.
  int rc;
  hipError_t e;
  {
    CaptureShared guard;                                 // no allocation / free of this library runs between Begin and End (model.h)
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
      (void)hipGetLastError();
      return run_ops_eager(P, b, s, evset);              // e.g. the legacy default stream cannot be captured
    }
    rc = run_ops_eager(P, b, s, evset, true);
    e = hipStreamEndCapture(s, &g.graph);
    if (rc != GDF_OK || e != hipSuccess) {
      // an invalidated capture: make sure the stream has really left capture mode before anything else is queued on it
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      for (int tries = 0; tries < 3 && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone; ++tries) {
        hipGraph_t junk = nullptr;
        (void)hipStreamEndCapture(s, &junk);
        if (junk) hipGraphDestroy(junk);
      }
      (void)hipGetLastError();
    }
  }
  static const bool dbg = getenv("GDF_DEBUG_GRAPH") != nullptr;
  if (dbg) fprintf(stderr, "[gdf] capture evset=%d rc=%d end=%s graph=%p (%s)\n", evset, rc, hipGetErrorString(e), (void*)g.graph, rc ? last_error() : "");
  auto failed = [&](const char* what, hipError_t err) {
    // reported once per plan: a persistently failing capture would otherwise silently turn every forward into ~1000 eager launches
    if (P.graph_capture_failures++ == 0)
      fprintf(stderr, "[gdf] hipGraph %s failed (%s%s%s): this forward runs eagerly; gdf_plan_graph_failures() counts such calls\n", what,
              hipGetErrorString(err), rc != GDF_OK ? "; " : "", rc != GDF_OK ? last_error() : "");
  };
  if (rc != GDF_OK || e != hipSuccess || !g.graph) {
    // a capture that failed or was invalidated has executed nothing: drop it and run this forward eagerly (the ops are pure
    // functions of their inputs); the next call tries to capture again
    if (g.graph) hipGraphDestroy(g.graph);
    (void)hipGetLastError();
    failed("capture", e);
    return run_ops_eager(P, b, s, evset);
  }
  hipError_t ie;
  { CaptureExclusive guard; ie = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0); }      // (allocates: not beside another thread's capture)
  if (ie != hipSuccess) {
    hipGraphDestroy(g.graph);
    (void)hipGetLastError();
    failed("instantiate", ie);
    return run_ops_eager(P, b, s, evset);
  }
  ++P.graph_captures;
  if (P.graphs.size() >= 12) {                           // evict the least recently used entry (timed replays: one graph per event set)
    size_t lru = 0;
    for (size_t i = 1; i < P.graphs.size(); ++i) if (P.graphs[i].stamp < P.graphs[lru].stamp) lru = i;
    { CaptureExclusive guard; hipGraphExecDestroy(P.graphs[lru].exec); hipGraphDestroy(P.graphs[lru].graph); }
    P.graphs.erase(P.graphs.begin() + lru);
  }
  g.stamp = ++P.graph_clock; ++P.graph_launches;
  P.graphs.push_back(g);
  if (hipGraphLaunch(P.graphs.back().exec, s) != hipSuccess) { set_error("hipGraphLaunch failed"); return GDF_ERR_HIP; }
  return GDF_OK;
}

int plan_run(Plan& P, const Bind& b, hipStream_t s, float* ms, const char** names, double* flops, int cap) {
  if (P.graph_mode && !ms && s != nullptr) {
    if (!P.warmed) { P.warmed = true; return run_ops_eager(P, b, s); }
    if (P.timing_label < 0) return run_ops_graph(P, b, s);
    // timed replay: this forward uses event set `evset`; its results are collected when the set comes round again (or on read)
    if (!P.timed_graph_broken) {
      const int evset = P.ev_next; P.ev_next = (P.ev_next + 1) % Plan::EV_RING;
      timing_collect(P, evset);
      const long fails = P.graph_capture_failures;
      const int rc = run_ops_graph(P, b, s, evset);
      if (P.graph_capture_failures != fails) P.timed_graph_broken = true;     // ran eagerly (with plain events); stay eager from now on
      if (rc == GDF_OK) P.ev_used[evset] = true;
      return rc;
    }
  }
  P.warmed = true;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms) { hipEventCreate(&e0); hipEventCreate(&e1); }
  int i = 0;
  int evset = -1; size_t evk = 0;
  if (P.timing_label >= 0 && !ms) {
    evset = P.ev_next; P.ev_next = (P.ev_next + 1) % Plan::EV_RING;
    timing_collect(P, evset);           // results of the forward that used this set EV_RING calls ago
    P.ev_used[evset] = true;
  }
  long nlab = 0;
  for (auto& op : P.ops) {
    if (ms && i < cap) hipEventRecord(e0, s);
    const bool timed = evset >= 0 && op.label == P.timing_label && (nlab++ % P.timing_stride) == 0;
    if (timed) hipEventRecord(P.ev[evset][evk], s);
    hipError_t e = op.fn(b, s);
    if (timed) { hipEventRecord(P.ev[evset][evk + 1], s); evk += 2; }
    if (e != hipSuccess) {
      set_error(std::string("op '") + op.name + "' failed: " + hipGetErrorString(e));
      return GDF_ERR_HIP;
    }
    if (ms && i < cap) {
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[i], e0, e1);
      if (names) names[i] = op.name;
      if (flops) flops[i] = op.flops;
    }
    ++i;
  }
  if (ms) { hipEventDestroy(e0); hipEventDestroy(e1); }
  return GDF_OK;
}

}  // namespace gdf
