// PixArt (alpha / sigma) DiT front end of libgdf.so (include/gdf_pixart.h).
//
// The op program restates (paths under /root/reference/feature):
//   Transformer2DModel.forward, patched inputs + ada_norm_single   diffusers/models/transformers/transformer_2d.py:404-475,
//       _operate_on_patched_inputs :496-516, _get_output_for_patched_inputs :540-575
//   BasicTransformerBlock.forward (ada_norm_single)                diffusers/models/attention.py:498-592
//   Attention + AttnProcessor2_0 (biases, additive text mask)      diffusers/models/attention_processor.py:3244-3331
//   FeedForward 'gelu-approximate'                                 diffusers/models/attention.py:1249-1258
// and the DiT hook ids of components/feature_extractor.py:250-286.  PatchEmbed, AdaLayerNormSingle and
// PixArtAlphaTextProjection are un-vendored diffusers==0.32.2 (published algorithm; oracle/pixart_ref.py).
//
// Layout: tokens-major fp32 residual stream [B*S][C] + an fp16 shadow (cross attention reads the UN-normalised stream,
// attention.py:541-543); every ada_norm_single modulation vector (scale_shift_table + timestep embedding, all blocks) is
// built by one broadcast-add launch; the 2x2 patch convolution is a K = 64 (16 real) GEMM over patchified rows with the
// sincos positional table added as a per-token row vector in the epilogue.
#include "builder.h"

namespace gdf {

namespace {

struct PixartModelBuilder : WeightBuilder {
  explicit PixartModelBuilder(Model& mm) : WeightBuilder(mm) {}

  LinW fused(const std::string& p, std::initializer_list<const char*> names, int C) {
    LinW w = lin_alloc((int)names.size() * C, C, true);
    int off = 0;
    for (const char* n : names) { lin_rows(p + n, w, C, off, false, true); off += C; }
    return w;
  }

  void build() {
    PixartW& x = m.pix;
    const gdf_pixart_desc& d = x.d;
    const int C = x.C, p = d.patch_size, kin = d.in_channels * p * p;
    x.kpad = (kin + 63) / 64 * 64;
    x.patch = lin_alloc(C, x.kpad, true);
    reg("pos_embed.proj.weight", {C, d.in_channels, p, p}, PK_ROWS_PADK, x.patch.w, C, kin, x.kpad);
    reg("pos_embed.proj.bias", {C}, PK_VEC, x.patch.b);
    x.t1 = lin("adaln_single.emb.timestep_embedder.linear_1", C, 256);
    x.t2 = lin("adaln_single.emb.timestep_embedder.linear_2", C, C);
    x.ada = lin("adaln_single.linear", 6 * C, C);
    x.cap1 = lin("caption_projection.linear_1", C, d.caption_channels);
    x.cap2 = lin("caption_projection.linear_2", C, C);
    x.tables = take((size_t)(d.num_layers * 6 + 2) * C * 4);
    for (int i = 0; i < d.num_layers; ++i) {
      const std::string b = "transformer_blocks." + std::to_string(i);
      PixartBlockW w;
      w.table = i * 6 * C;
      reg(b + ".scale_shift_table", {6, C}, PK_VEC_OFF, x.tables, 6 * C, w.table);
      w.qkv = fused(b + ".attn1.", {"to_q", "to_k", "to_v"}, C);
      w.o1 = lin(b + ".attn1.to_out.0", C, C);
      w.q2 = lin(b + ".attn2.to_q", C, C);
      w.kv2 = fused(b + ".attn2.", {"to_k", "to_v"}, C);
      w.o2 = lin(b + ".attn2.to_out.0", C, C);
      w.ff1 = lin(b + ".ff.net.0.proj", 4 * C, C);
      w.ff2 = lin(b + ".ff.net.2", C, 4 * C);
      x.blocks.push_back(w);
    }
    reg("scale_shift_table", {2, C}, PK_VEC_OFF, x.tables, 2 * C, d.num_layers * 6 * C);
    x.proj_out = lin("proj_out", p * p * d.out_channels, C);
    m.weight_bytes = cur;
  }
};

struct XB : PlanBuilder {   // PixArt op program
  const PixartW& x;
  int S = 0, gh = 0, gw = 0, T = 0;
  size_t mod = 0, xf = 0, xh = 0;
  int ldm = 0;

  XB(const Model& mm, Plan& pp, bool d, const PlanOpts& o) : PlanBuilder(mm, pp, d, o), x(mm.pix) {}

  Ref modv(int col) const { return ws(mod + (size_t)col * 4); }
  Epi plain(const LinW& w) { Epi e; e.dit = 1; e.bias = wt(w.b); e.has_bias = w.has_bias; return e; }
  // stream += [gate *] (A W^T + bias); keeps the fp16 shadow when asked
  Epi resid(const LinW& w, int gate_col, bool shadow) {
    Epi e = plain(w);
    if (gate_col >= 0) { e.rowvec = modv(gate_col); e.has_rv = true; e.rps = S; e.ldrv = ldm; e.rv_mul = 1; }
    e.res32 = ws(xf); e.has_r32 = true; e.ldres = x.C;
    e.out32 = ws(xf); e.has_o32 = true; e.ldo32 = x.C;
    if (shadow) { e.out16 = ws(xh); e.has_o16 = true; e.ldo16 = x.C; }
    return e;
  }
  void adaln(const char* name, int shift_col, int scale_col, Ref dst) {
    const size_t n = (size_t)Bn * S;
    const Ref src = ws(xf), sc = modv(scale_col), sh = modv(shift_col);
    const int C = x.C, ld = ldm, rps = S;
    op(name, 0, [=](const Bind& b, hipStream_t s) {
      return launch_layernorm_mod(nullptr, (const float*)b.p(src), C, (int)n, C, 1e-6f, (const float*)b.p(sc), (const float*)b.p(sh),
                                  ld, rps, 0, 0, (half_t*)b.p(dst), s);
    });
  }
  void hook16(const std::string& id, Ref src, int ld, int C) { hook_copy(want(id, C, gh, gw), src, ld, (size_t)Bn * S, C); }
  void attention(const char* name, Ref q, int ldq, Ref k, Ref v, int ldkv, Ref o, int Sk, bool masked, int map_slot = -1) {
    const int C = x.C, D = x.d.attention_head_dim, heads = x.d.num_attention_heads, Bq = Bn, Sq = S;
    op(name, 4.0 * (double)Bn * heads * Sq * (double)Sk * D, [=](const Bind& b, hipStream_t s) {
      AttnParams a{};
      a.q = (const half_t*)b.p(q); a.ldq = ldq; a.k = (const half_t*)b.p(k); a.ldk = ldkv; a.v = (const half_t*)b.p(v); a.ldv = ldkv;
      a.o = (half_t*)b.p(o); a.ldo = C; a.B = Bq; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D; a.kv_bstride = Sk;
      a.scale = 1.0f / sqrtf((float)D);
      a.kv_len = masked ? (const int*)b.base[BUF_TID] : nullptr;
      a.map = map_slot >= 0 ? (half_t*)b.hook(map_slot) : nullptr;          // AttnStoreProcessor `map` hook (B, heads, S, Sk)
      return launch_attention(a, s);
    });
    if (map_slot >= 0) hook_done();
  }

  void build(int H, int W) {
    const gdf_pixart_desc& d = x.d;
    const int C = x.C, p = d.patch_size, Bq = Bn, L = d.num_layers;
    gh = H / p; gw = W / p; S = gh * gw; T = n_ctx;
    const size_t n = (size_t)Bn * S, nt = (size_t)Bn * T;
    ldm = (L * 6 + 2) * C;
    // ---- timestep embedding, adaln_single, modulation tables of every block (:506-508; attention.py:498-503) ----
    const size_t vb = (size_t)Bn * C * 4;
    const size_t tsin = tmp((size_t)Bn * 256 * 4), t1 = tmp(vb), emb = tmp(vb), tvec = tmp(vb * 6);
    const size_t mod_b = (size_t)Bn * ldm * 4;
    mod = tmp(mod_b);
    {
      const Ref w1 = wt(x.t1.w), b1 = wt(x.t1.b), w2 = wt(x.t2.w), b2 = wt(x.t2.b), wa = wt(x.ada.w), ba = wt(x.ada.b), tb = wt(x.tables);
      const size_t mo = mod; const int ld = ldm;
      op("adaln_single", 0, [=](const Bind& b, hipStream_t s) {
        hipError_t e = launch_sinusoid((const float*)b.base[BUF_T], Bq, 1, 256, (float*)b.ws(tsin), 256, 0, 0, s);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(tsin), 256, Bq, 256, (const half_t*)b.p(w1), (const float*)b.p(b1), C, 0, 0, (float*)b.ws(t1), C, s);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(t1), C, Bq, C, (const half_t*)b.p(w2), (const float*)b.p(b2), C, 1, 0, (float*)b.ws(emb), C, s);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(emb), C, Bq, C, (const half_t*)b.p(wa), (const float*)b.p(ba), 6 * C, 1, 0, (float*)b.ws(tvec), 6 * C, s);
        if (e != hipSuccess) return e;
        // blocks: table_i + tvec (period 6C); final: scale_shift_table[2][C] + embedded_timestep (period C)
        e = launch_add_table((const float*)b.p(tb), (const float*)b.ws(tvec), 6 * C, 6 * C, Bq, (long)L * 6 * C, (float*)b.ws(mo), ld, s);
        if (e != hipSuccess) return e;
        return launch_add_table((const float*)b.p(tb) + (size_t)L * 6 * C, (const float*)b.ws(emb), C, C, Bq, 2L * C,
                                (float*)b.ws(mo) + (size_t)L * 6 * C, ld, s);
      });
    }
    untmp(tsin, (size_t)Bn * 256 * 4); untmp(t1, vb); untmp(emb, vb); untmp(tvec, vb * 6);
    // ---- caption projection (:510-512): Linear -> GELU(tanh) -> Linear ----
    const size_t enc = tmp(nt * C * 2);
    {
      const size_t c1 = tmp(nt * C * 2);
      { Epi e = plain(x.cap1); e.act = 1; e.out16 = ws(c1); e.has_o16 = true; e.ldo16 = C;
        gemm("caption_proj_1", Ref{BUF_CTX, 0}, d.caption_channels, nt, x.cap1, C, d.caption_channels, 0, e); }
      { Epi e = plain(x.cap2); e.out16 = ws(enc); e.has_o16 = true; e.ldo16 = C; gemm("caption_proj_2", ws(c1), C, nt, x.cap2, C, C, 0, e); }
      untmp(c1, nt * C * 2);
    }
    // ---- PatchEmbed: 2x2 conv as a GEMM over patch rows + bias + sincos table (per-token row vector) ----
    const size_t xf_b = n * C * 4, xh_b = n * C * 2;
    xf = tmp(xf_b); xh = tmp(xh_b);
    {
      const size_t pos_b = (size_t)S * C * 4, pos = tmp(pos_b), pr_b = n * x.kpad * 2, pr = tmp(pr_b);
      const int cin = d.in_channels, kp = x.kpad, ghh = gh, gww = gw, base = d.sample_size / p;
      const float isc = (float)d.interpolation_scale;
      op("patchify", 0, [=](const Bind& b, hipStream_t s) {
        hipError_t e = launch_sincos_pos_embed((float*)b.ws(pos), C, ghh, gww, base, isc, s);
        if (e != hipSuccess) return e;
        return launch_patchify((const half_t*)b.base[BUF_LAT], Bq, cin, H, W, p, kp, (half_t*)b.ws(pr), s);
      });
      Epi e = plain(x.patch); e.rowvec = ws(pos); e.has_rv = true; e.rps = S; e.ldrv = C; e.rv_tok = 1;
      e.out32 = ws(xf); e.has_o32 = true; e.ldo32 = C;
      gemm("patch_embed", ws(pr), x.kpad, n, x.patch, C, x.kpad, 0, e);
      untmp(pos, pos_b); untmp(pr, pr_b);
    }
    const size_t nb = n * C * 2;
    for (int i = 0; i < L && !stop; ++i) {
      const PixartBlockW& w = x.blocks[i];
      const std::string bid = "vit-block" + std::to_string(i);
      const int t0 = w.table;                       // chunks: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      // ---- self attention (attention.py:498-526) ----
      const size_t ln = tmp(nb);
      adaln("adaln", t0 + 0, t0 + C, ws(ln));
      const size_t qkv = tmp(n * 3 * C * 2);
      { Epi e = plain(w.qkv); e.out16 = ws(qkv); e.has_o16 = true; e.ldo16 = 3 * C; gemm("attn1_qkv", ws(ln), C, n, w.qkv, 3 * C, C, 0, e); }
      untmp(ln, nb);
      hook16(bid + "-self-q", ws(qkv), 3 * C, C);                                        // attention_processor.py:3291-3294
      hook16(bid + "-self-k", ws(qkv + (size_t)C * 2), 3 * C, C);
      hook16(bid + "-self-v", ws(qkv + (size_t)2 * C * 2), 3 * C, C);
      if (stop) { untmp(qkv, n * 3 * C * 2); break; }
      size_t ao = tmp(nb);
      attention("attn1", ws(qkv), 3 * C, ws(qkv + (size_t)C * 2), ws(qkv + (size_t)2 * C * 2), 3 * C, ws(ao), S, false,
                want_map(bid + "-self-map", x.d.num_attention_heads, S, S));        // components/attention.py:238-244
      untmp(qkv, n * 3 * C * 2);
      { Epi e = resid(w.o1, t0 + 2 * C, true); gemm("attn1_out", ws(ao), C, n, w.o1, C, C, 0, e); }   // gate_msa; fp16 shadow for attn2
      untmp(ao, nb);
      // ---- cross attention on the un-normalised stream (attention.py:541-558) ----
      const size_t q2 = tmp(nb), kv = tmp(nt * 2 * C * 2);
      { Epi e = plain(w.q2); e.out16 = ws(q2); e.has_o16 = true; e.ldo16 = C; gemm("attn2_q", ws(xh), C, n, w.q2, C, C, 0, e); }
      hook16(bid + "-cross-q", ws(q2), C, C);
      { Epi e = plain(w.kv2); e.out16 = ws(kv); e.has_o16 = true; e.ldo16 = 2 * C; gemm("attn2_kv", ws(enc), C, nt, w.kv2, 2 * C, C, 0, e); }
      ao = tmp(nb);
      attention("attn2", ws(q2), C, ws(kv), ws(kv + (size_t)C * 2), 2 * C, ws(ao), T, true,
                want_map(bid + "-cross-map", x.d.num_attention_heads, S, T));
      untmp(q2, nb); untmp(kv, nt * 2 * C * 2);
      { Epi e = resid(w.o2, -1, false); gemm("attn2_out", ws(ao), C, n, w.o2, C, C, 0, e); }
      untmp(ao, nb);
      if (stop) break;
      // ---- feed forward (attention.py:570-586) ----
      const size_t l2 = tmp(nb);
      adaln("adaln", t0 + 3 * C, t0 + 4 * C, ws(l2));
      const size_t inner = tmp(n * 4 * C * 2);
      { Epi e = plain(w.ff1); e.act = 1; e.out16 = ws(inner); e.has_o16 = true; e.ldo16 = 4 * C; gemm("ff_in", ws(l2), C, n, w.ff1, 4 * C, C, 0, e); }
      untmp(l2, nb);
      hook16(bid + "-ffn-inner", ws(inner), 4 * C, 4 * C);                               // attention.py:1255-1257
      { Epi e = resid(w.ff2, t0 + 5 * C, false); gemm("ff_out", ws(inner), 4 * C, n, w.ff2, C, 4 * C, 0, e); }
      untmp(inner, n * 4 * C * 2);
      {                                                                                  // attention.py:589-590 (fp32 stream -> fp16 hook)
        const int slot = want(bid + "-out", C, gh, gw);
        if (slot >= 0) {
          const Ref src = ws(xf);
          op("hook_store", 0, [=](const Bind& b, hipStream_t s) {
            return launch_copy2d(nullptr, (const float*)b.p(src), C, (half_t*)b.hook(slot), C, (int)n, C, s);
          });
          hook_done();
        }
      }
    }
    // ---- output: norm_out, (scale_shift_table + embedded_timestep) modulate, proj_out, unpatchify (:552-570) ----
    if (!stop) {
      const int po = p * p * d.out_channels;
      const size_t no = tmp(nb), tok = tmp(n * po * 2);
      adaln("norm_out", L * 6 * C + 0, L * 6 * C + C, ws(no));                          // chunk order: shift, scale
      { Epi e = plain(x.proj_out); e.out16 = ws(tok); e.has_o16 = true; e.ldo16 = po; gemm("final_proj_out", ws(no), C, n, x.proj_out, po, C, 0, e); }
      untmp(no, nb);
      const int oc = d.out_channels, ghh = gh, gww = gw;
      P.writes_noise = true;
      op("unpatchify", 0, [=](const Bind& b, hipStream_t s) {
        return launch_unpatchify((const half_t*)b.ws(tok), Bq, oc, ghh, gww, p, (half_t*)b.base[BUF_NOISE], s);
      });
      untmp(tok, n * po * 2);
    }
    untmp(xf, xf_b); untmp(xh, xh_b); untmp(enc, nt * C * 2); untmp(mod, mod_b);
  }
};

}  // namespace

Model* pixart_model_create(const gdf_pixart_desc& d) {
  const int D = d.attention_head_dim;
  if (!(D == 32 || D == 40 || D == 64 || D == 72 || D == 80 || D == 128 || D == 160)) { set_error("unsupported attention_head_dim"); return nullptr; }
  const int C = d.num_attention_heads * D;
  if (C % 64 || d.caption_channels % 64 || d.patch_size < 1 || d.num_layers < 1 || d.in_channels < 1 || d.out_channels < 1 ||
      (d.patch_size * d.patch_size * d.out_channels) % 8 || d.sample_size % d.patch_size) {
    set_error("bad pixart desc (inner dim and caption_channels must be multiples of 64)"); return nullptr;
  }
  Model* m = new Model();
  m->kind = 3;
  m->pix.d = d;
  m->pix.C = C;
  PixartModelBuilder b(*m);
  b.build();
  { CaptureExclusive guard; if (hipMalloc(&m->weights, m->weight_bytes) != hipSuccess) { set_error("hipMalloc(weights) failed"); delete m; return nullptr; } }
  (void)hipMemset(m->weights, 0, m->weight_bytes);
  PlanOpts o{}; o.stream_fp32 = 1;
  Plan dry;
  pixart_plan_build(*m, dry, 1, 4 * d.patch_size, 4 * d.patch_size, 8, nullptr, 0, o, /*dry=*/true);
  m->hook_names = dry.dry_ids;
  return m;
}

int pixart_plan_build(const Model& m, Plan& P, int batch, int lat_h, int lat_w, int n_txt, const char* const* ids, int n_ids,
                      const PlanOpts& opts, bool dry) {
  if (m.kind != 3) { set_error("not a PixArt model"); return GDF_ERR_ARG; }
  const int p = m.pix.d.patch_size;
  if (batch < 1 || lat_h < p || lat_w < p || (lat_h % p) || (lat_w % p) || n_txt < 1) {
    set_error("latent size must be a positive multiple of patch_size, n_txt positive"); return GDF_ERR_ARG;
  }
  const size_t rows = (size_t)batch * (lat_h / p) * (lat_w / p);
  if (rows * (size_t)m.pix.C * 4 * 2 >= (1ull << 31)) { set_error("batch * tokens too large for 32-bit buffer offsets; split the batch"); return GDF_ERR_UNSUPPORTED; }
  P.batch = batch; P.H = lat_h; P.W = lat_w; P.n_ctx = n_txt; P.opts = opts;
  XB b(m, P, dry, opts);
  b.Bn = batch; b.n_ctx = n_txt;
  if (!dry) {
    std::unordered_set<std::string> known(m.hook_names.begin(), m.hook_names.end());
    for (int i = 0; i < n_ids; ++i)
      if (ids[i] && known.count(ids[i])) P.requested.insert(ids[i]);
    b.remaining = (int)P.requested.size();
    if (opts.early_exit && b.remaining == 0) b.stop = true;
  }
  b.build(lat_h, lat_w);
  P.ws_bytes = b.ar.peak + 256;
  return GDF_OK;
}

int pixart_forward(Plan& P, const Model& m, const void* latents, const float* timestep, const void* enc, const int* text_lens,
                   void* const* hook_out, void* out, void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap) {
  if (m.kind != 3) { set_error("gdf_pixart_forward on a non-PixArt model"); return GDF_ERR_STATE; }
  if (m.n_set != (int)m.params.size()) { set_error("model weights incomplete"); return GDF_ERR_STATE; }
  if (!latents || !timestep || !enc || !ws) { set_error("null input pointer"); return GDF_ERR_ARG; }
  if (P.hooks.size() && !hook_out) { set_error("hook_out is null"); return GDF_ERR_ARG; }
  if (P.writes_noise && !out) { set_error("output buffer required (the plan runs proj_out)"); return GDF_ERR_ARG; }
  Bind b;
  b.base[BUF_WS] = (char*)ws; b.base[BUF_WT] = (char*)m.weights; b.base[BUF_LAT] = (char*)latents; b.base[BUF_T] = (char*)timestep;
  b.base[BUF_CTX] = (char*)enc; b.base[BUF_TID] = (char*)text_lens; b.base[BUF_NOISE] = (char*)out;
  b.hooks = hook_out;
  return plan_run(P, b, s, ms, names, flops, cap);
}

}  // namespace gdf
