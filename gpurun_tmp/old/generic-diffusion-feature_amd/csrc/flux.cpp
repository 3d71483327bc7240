// MMDiT (Flux) front end of libgdf.so: weight arena layout, static op program, forward entry.
//
// The op program restates, for one transformer forward, the orchestration of (paths under /root/reference/feature):
//   FluxTransformer2DModel.forward       diffusers/models/transformers/transformer_flux.py:414-603
//   FluxTransformerBlock.forward         transformer_flux.py:167-226
//   FluxSingleTransformerBlock.forward   transformer_flux.py:86-112
//   FluxAttnProcessor2_0.__call__        diffusers/models/attention_processor.py:2266-2362
//   FeedForward.forward                  diffusers/models/attention.py:1249-1258 (`gelu-approximate`)
// and the flux hook ids of components/feature_extractor.py:98-123.  AdaLayerNormZero/ZeroSingle/Continuous, RMSNorm,
// FluxPosEmbed, apply_rotary_emb and CombinedTimestepGuidanceTextProjEmbeddings are un-vendored diffusers==0.32.2
// (restated from the published algorithm; see oracle/flux_ref.py).
//
// Data layout in HBM:
//   * The residual stream is ONE fp32 tensor [B*T + B*S][C], region-major: all samples' text rows first, then all
//     samples' image rows.  The text / image halves of every double-block linear are then contiguous row ranges, the
//     single blocks see one matrix (`torch.cat([enc, hid], 1)`, transformer_flux.py:549, is free), and only the joint
//     attention kernel needs to know the per-sample token -> row map (AttnParams.seg_T).
//   * No fp16 shadow of the stream exists: its only readers are the adaLN LayerNorm (fp32 in, fp16 out) and the
//     gated residual epilogues (fp32 in/out).  Every MFMA operand is a normalised / activated fp16 tensor.
//   * All adaLN modulation vectors of the forward come from ONE stacked [mod_total][C] linear on silu(temb).
#include "builder.h"

namespace gdf {

namespace {

struct FluxModelBuilder : WeightBuilder {
  explicit FluxModelBuilder(Model& mm) : WeightBuilder(mm) {}

  size_t gain(const std::string& n, int d) { const size_t o = take((size_t)d * 4); reg(n + ".weight", {d}, PK_VEC, o); return o; }
  LinW fused3(const std::string& p, const char* a, const char* b, const char* c, int C) {
    LinW w = lin_alloc(3 * C, C, true);
    lin_rows(p + a, w, C, 0, false, true); lin_rows(p + b, w, C, C, false, true); lin_rows(p + c, w, C, 2 * C, false, true);
    return w;
  }

  void build() {
    FluxW& f = m.flux;
    const gdf_flux_desc& d = f.d;
    const int C = f.C, hid = f.hid, D = f.D;
    f.x_emb = lin("x_embedder", C, d.in_channels);
    f.ctx_emb = lin("context_embedder", C, d.joint_attention_dim);
    f.t1 = lin("time_text_embed.timestep_embedder.linear_1", C, 256);
    f.t2 = lin("time_text_embed.timestep_embedder.linear_2", C, C);
    if (d.guidance_embeds) {
      f.g1 = lin("time_text_embed.guidance_embedder.linear_1", C, 256);
      f.g2 = lin("time_text_embed.guidance_embedder.linear_2", C, C);
    }
    f.p1 = lin("time_text_embed.text_embedder.linear_1", C, d.pooled_projection_dim);
    f.p2 = lin("time_text_embed.text_embedder.linear_2", C, C);
    f.mod_total = (12 * d.num_layers + 3 * d.num_single_layers + 2) * C;
    f.mod_all = lin_alloc(f.mod_total, C, true);
    int mo = 0;
    for (int i = 0; i < d.num_layers; ++i) {
      const std::string p = "transformer_blocks." + std::to_string(i);
      FluxDoubleW w;
      w.mod = mo; lin_rows(p + ".norm1.linear", f.mod_all, 6 * C, mo, false, true); mo += 6 * C;
      w.cmod = mo; lin_rows(p + ".norm1_context.linear", f.mod_all, 6 * C, mo, false, true); mo += 6 * C;
      w.qkv = fused3(p + ".attn.", "to_q", "to_k", "to_v", C);
      w.cqkv = fused3(p + ".attn.", "add_q_proj", "add_k_proj", "add_v_proj", C);
      w.o = lin(p + ".attn.to_out.0", C, C);
      w.co = lin(p + ".attn.to_add_out", C, C);
      w.nq = gain(p + ".attn.norm_q", D); w.nk = gain(p + ".attn.norm_k", D);
      w.cnq = gain(p + ".attn.norm_added_q", D); w.cnk = gain(p + ".attn.norm_added_k", D);
      w.ff1 = lin(p + ".ff.net.0.proj", hid, C); w.ff2 = lin(p + ".ff.net.2", C, hid);
      w.cff1 = lin(p + ".ff_context.net.0.proj", hid, C); w.cff2 = lin(p + ".ff_context.net.2", C, hid);
      f.dbl.push_back(w);
    }
    for (int i = 0; i < d.num_single_layers; ++i) {
      const std::string p = "single_transformer_blocks." + std::to_string(i);
      FluxSingleW w;
      w.mod = mo; lin_rows(p + ".norm.linear", f.mod_all, 3 * C, mo, false, true); mo += 3 * C;
      w.qkv = fused3(p + ".attn.", "to_q", "to_k", "to_v", C);
      w.nq = gain(p + ".attn.norm_q", D); w.nk = gain(p + ".attn.norm_k", D);
      w.mlp = lin(p + ".proj_mlp", hid, C);
      w.out = lin(p + ".proj_out", C, C + hid);
      f.sgl.push_back(w);
    }
    f.mod_out = mo; lin_rows("norm_out.linear", f.mod_all, 2 * C, mo, false, true); mo += 2 * C;
    f.proj_out = lin("proj_out", d.in_channels, C);
    m.weight_bytes = cur;
  }
};

struct FB : PlanBuilder {   // Flux op program
  const FluxW& f;
  int T = 0, S = 0, gh = 0, gw = 0;
  size_t NT = 0, NS = 0, NR = 0;
  size_t mod = 0, cosb = 0, sinb = 0, xf = 0;

  FB(const Model& mm, Plan& pp, bool d, const PlanOpts& o) : PlanBuilder(mm, pp, d, o), f(mm.flux) {}

  Ref modv(int col) const { return ws(mod + (size_t)col * 4); }
  Ref stream_rows(size_t row) const { return ws(xf + row * (size_t)f.C * 4); }

  // y[rows r0..r0+n) = LN(stream rows) * (1 + scale) + shift  -> fp16 [n][C] at `dst`
  // 'bfloat16x2' plans (m.x2): every MFMA A operand is a bf16 hi + lo pair [hi C | lo C] in one row (px = 2), multiplied as
  // [hi | lo] x [W | W]; q / k / v and the attention internals are fp16
  int px() const { return m.x2 ? 2 : 1; }
  // q8 != null ('fp8-mx' plans): the same rows also as fp8 (e4m3) [n][C] + one scale per row, written in the same pass
  void adaln(const char* name, size_t r0, size_t n, int shift_col, int scale_col, int rps, size_t seg_rows, int rps2, Ref dst,
             const MxA* q8 = nullptr) {
    const Ref x = stream_rows(r0), sc = modv(scale_col), sh = modv(shift_col);
    const int C = f.C, ldm = f.mod_total, bf = m.bf16, ldy = C * px(), y_lo = m.x2 ? C : 0;
    const bool hq = q8 != nullptr;
    const MxA q = hq ? *q8 : MxA{};
    op(name, 0, [=](const Bind& b, hipStream_t s) {
      return launch_layernorm_mod(nullptr, (const float*)b.p(x), C, (int)n, C, 1e-6f, (const float*)b.p(sc), (const float*)b.p(sh),
                                  ldm, rps, (int)seg_rows, rps2, (half_t*)b.p(dst), s, bf, ldy, y_lo,
                                  hq ? (unsigned char*)b.p(q.a8) : nullptr, q.lda8, hq ? (float*)b.p(q.ascale) : nullptr);
    });
  }
  // fp8 copy of 16-bit rows [n][ld] (K columns) for an 'fp8-mx' GEMM: workspace for the bytes + the row scales
  MxA quant8(Ref src, int ld, size_t n, int K) {
    MxA q; q.lda8 = K;
    const size_t qb = tmp(n * (size_t)K), qs = tmp(n * 4);
    q.a8 = ws(qb); q.ascale = ws(qs);
    const int bf = m.bf16;
    const Ref d8 = q.a8, ds = q.ascale;
    op("quant_fp8", 0, [=](const Bind& b, hipStream_t s) {
      return launch_quant_rows_fp8((const half_t*)b.p(src), ld, (int)n, K, bf, (unsigned char*)b.p(d8), K, (float*)b.p(ds), s);
    });
    return q;
  }
  void free8(const MxA& q, size_t n) { untmp(q.a8.off, n * (size_t)q.lda8); untmp(q.ascale.off, n * 4); }
  MxA alloc8(size_t n, int K) { MxA q; q.lda8 = K; q.a8 = ws(tmp(n * (size_t)K)); q.ascale = ws(tmp(n * 4)); return q; }
  static MxA rows8(const MxA& q, size_t r0) { MxA o = q; o.a8.off += r0 * (size_t)q.lda8; o.ascale.off += r0 * 4; return o; }
  // image-token hook from a 16-bit matrix; s_lo > 0: a split pair; src_bf: element type (-1 = the model's)
  void hook_rows16(const std::string& id, Ref src, int ld, int C, int s_lo = 0, int src_bf = -1, float scale = 1.0f) {
    hook_copy(want(id, C, gh, gw), src, ld, NS, C, s_lo, src_bf, scale);
  }
  void hook_rows32(const std::string& id, Ref src, int ld, int C) {      // ... from the fp32 stream
    const int slot = want(id, C, gh, gw);
    if (slot < 0) return;
    const size_t n = NS;
    P.hooks[slot].copied = true;
    op("hook_store", 0, [=](const Bind& b, hipStream_t s) {
      return launch_copy2d(nullptr, (const float*)b.p(src), ld, (half_t*)b.hook(slot), C, (int)n, C, s, 0, /*sat=*/1);
    });
    hook_done();
  }
  // the fused form (GemmParams::qkn_*): possible when none of the block's pre-norm q / k / v hooks is requested
  bool qkn_fusable(const std::string& bid) const {
    if (f.D != 128) return false;
    if (dry) return false;
    return !P.requested.count(bid + "-q") && !P.requested.count(bid + "-k") && !P.requested.count(bid + "-v");
  }
  void qkn(Epi& e, size_t wq, size_t wk, int pos0, int rps, int seg_rows, int pos1, int rps2) {
    e.qkn_nq = f.C; e.qkn_wq = wt(wq); e.qkn_wk = wt(wk); e.rope_cos = ws(cosb); e.rope_sin = ws(sinb); e.qkn_eps = 1e-6f;
    e.qkn_pos0 = pos0; e.qkn_rps = rps; e.qkn_seg_rows = seg_rows; e.qkn_pos1 = pos1; e.qkn_rps2 = rps2;
  }
  // RMSNorm(q), RMSNorm(k) + RoPE in place on rows [r0, r0+n) of the qkv buffer (ld 3C)
  void qk_norm_rope(size_t qkv, size_t r0, size_t n, size_t wq, size_t wk, int pos0, int rps) {
    const int C = f.C, heads = C / f.D, D = f.D, bf = m.x2 ? 0 : m.bf16;          // ('bfloat16x2': the q / k / v buffer is fp16)
    const Ref x = ws(qkv + r0 * (size_t)(3 * C) * 2), q = wt(wq), k = wt(wk), cs = ws(cosb), sn = ws(sinb);
    op("qk_norm_rope", 0, [=](const Bind& b, hipStream_t s) {
      return launch_qk_norm_rope((half_t*)b.p(x), 3 * C, (int)n, heads, D, 0, C, (const float*)b.p(q), (const float*)b.p(k), 1e-6f,
                                 (const float*)b.p(cs), (const float*)b.p(sn), pos0, rps, s, bf);
    });
  }
  // cross_slot / self_slot: hook slots of `cross-map` (B, heads, S, T) / `self-map` (B, heads, S, S) or -1
  // (FluxAttnStoreProcessor, components/attention.py:493-502: image queries only, split by key)
  void joint_attention(size_t qkv, Ref o, int ldo, int cross_slot = -1, int self_slot = -1, int o_lo = 0, float o_scale = 0.f) {
    const int C = f.C, D = f.D, heads = C / D, Bq = Bn, Sj = T + S, Tq = T, bf = m.x2 ? 0 : m.bf16, pbf = m.x2;
    const Ref q = ws(qkv), k = ws(qkv + (size_t)C * 2), v = ws(qkv + (size_t)2 * C * 2);
    op("joint_attn", 4.0 * (double)Bn * heads * Sj * (double)Sj * D, [=](const Bind& b, hipStream_t s) {
      AttnParams a{};
      a.q = (const half_t*)b.p(q); a.ldq = 3 * C; a.k = (const half_t*)b.p(k); a.ldk = 3 * C;
      a.v = (const half_t*)b.p(v); a.ldv = 3 * C; a.o = (half_t*)b.p(o); a.ldo = ldo;
      a.B = Bq; a.heads = heads; a.Sq = Sj; a.Sk = Sj; a.D = D; a.scale = 1.0f / sqrtf((float)D);
      a.kv_bstride = Sj; a.seg_T = Tq; a.bf16 = bf; a.o_lo = o_lo; a.o_pair_bf16 = pbf; a.o_scale = o_scale;
      a.map = self_slot >= 0 ? (half_t*)b.hook(self_slot) : nullptr;
      a.map2 = cross_slot >= 0 ? (half_t*)b.hook(cross_slot) : nullptr;
      return launch_attention(a, s);
    });
    if (cross_slot >= 0) hook_done();
    if (self_slot >= 0) hook_done();
  }
  // ids are emitted in the reference's gather order: cross-map, then self-map
  void map_slots(const std::string& bid, int& cross_slot, int& self_slot) {
    const int heads = f.C / f.D;
    cross_slot = want_map(bid + "-cross-map", heads, S, T);
    self_slot = want_map(bid + "-self-map", heads, S, S);
  }
  Epi plain(const LinW& w) { Epi e; e.dit = 1; e.bias = wt(w.b); e.has_bias = w.has_bias; return e; }
  // stream[r0..] += gate * (A W^T + bias)
  Epi gated(const LinW& w, size_t r0, int gate_col, int rps, size_t seg_rows = 0, int rps2 = 0) {
    Epi e = plain(w);
    e.rowvec = modv(gate_col); e.has_rv = true; e.rps = rps; e.ldrv = f.mod_total; e.rv_mul = 1;
    e.rv_seg_rows = (int)seg_rows; e.rv_rps2 = rps2;
    e.res32 = stream_rows(r0); e.has_r32 = true; e.ldres = f.C;
    e.out32 = stream_rows(r0); e.has_o32 = true; e.ldo32 = f.C;
    return e;
  }

  void build() {
    const gdf_flux_desc& d = f.d;
    const int C = f.C, hid = f.hid, D = f.D, Bq = Bn, Tq = T, Sq = S;
    NT = (size_t)Bn * T; NS = (size_t)Bn * S; NR = NT + NS;
    const size_t nt = NT, ns = NS, nr = NR;

    // ---- rotary tables from txt_ids | img_ids (FluxPosEmbed over cat(txt_ids, img_ids), :498-499) ----
    const size_t rope_b = (size_t)(T + S) * D * 4;
    cosb = tmp(rope_b); sinb = tmp(rope_b);
    {
      const size_t cb = cosb, sb = sinb;
      const int a0 = d.axes_dims_rope[0], a1 = d.axes_dims_rope[1], a2 = d.axes_dims_rope[2];
      op("rope_table", 0, [=](const Bind& b, hipStream_t s) {
        const int ax[3] = {a0, a1, a2};
        if (!b.base[BUF_IDS_IMG] || !b.base[BUF_IDS_TXT]) return hipErrorInvalidValue;
        hipError_t e = launch_rope_table((const float*)b.base[BUF_IDS_TXT], Tq, 3, ax, 10000.0, (float*)b.ws(cb), (float*)b.ws(sb), 0, s);
        if (e != hipSuccess) return e;
        return launch_rope_table((const float*)b.base[BUF_IDS_IMG], Sq, 3, ax, 10000.0, (float*)b.ws(cb), (float*)b.ws(sb), Tq, s);
      });
    }
    // ---- temb = timestep_embedder(t*1000) [+ guidance_embedder(g*1000)] + text_embedder(pooled)  (:472-482) ----
    const size_t vb = (size_t)Bn * C * 4;
    const size_t tsin = tmp((size_t)Bn * 256 * 4), t1 = tmp(vb), temb = tmp(vb), stemb = tmp(vb);
    const size_t pv_b = (size_t)Bn * d.pooled_projection_dim * 4, pv = tmp(pv_b);
    {
      const Ref w1 = wt(f.t1.w), b1 = wt(f.t1.b), w2 = wt(f.t2.w), b2 = wt(f.t2.b);
      const Ref gw1 = wt(f.g1.w), gb1 = wt(f.g1.b), gw2 = wt(f.g2.w), gb2 = wt(f.g2.b);
      const Ref pw1 = wt(f.p1.w), pb1 = wt(f.p1.b), pw2 = wt(f.p2.w), pb2 = wt(f.p2.b);
      const bool guid = d.guidance_embeds != 0;
      const int pd = d.pooled_projection_dim, bf = m.bf16;
      op("time_text_embed", 0, [=](const Bind& b, hipStream_t s) {
        hipError_t e = launch_sinusoid((const float*)b.base[BUF_T], Bq, 1, 256, (float*)b.ws(tsin), 256, 0, 0, s, 1000.0f);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(tsin), 256, Bq, 256, (const half_t*)b.p(w1), (const float*)b.p(b1), C, 0, 0, (float*)b.ws(t1), C, s, bf);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(t1), C, Bq, C, (const half_t*)b.p(w2), (const float*)b.p(b2), C, 1, 0, (float*)b.ws(temb), C, s, bf);
        if (e != hipSuccess) return e;
        if (guid) {
          if (!b.base[BUF_TID]) return hipErrorInvalidValue;
          e = launch_sinusoid((const float*)b.base[BUF_TID], Bq, 1, 256, (float*)b.ws(tsin), 256, 0, 0, s, 1000.0f);
          if (e != hipSuccess) return e;
          e = launch_small_linear((const float*)b.ws(tsin), 256, Bq, 256, (const half_t*)b.p(gw1), (const float*)b.p(gb1), C, 0, 0, (float*)b.ws(t1), C, s, bf);
          if (e != hipSuccess) return e;
          e = launch_small_linear((const float*)b.ws(t1), C, Bq, C, (const half_t*)b.p(gw2), (const float*)b.p(gb2), C, 1, 1, (float*)b.ws(temb), C, s, bf);
          if (e != hipSuccess) return e;
        }
        e = launch_widen((const half_t*)b.base[BUF_TXT], Bq, pd, (float*)b.ws(pv), pd, 0, s, bf);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(pv), pd, Bq, pd, (const half_t*)b.p(pw1), (const float*)b.p(pb1), C, 0, 0, (float*)b.ws(t1), C, s, bf);
        if (e != hipSuccess) return e;
        e = launch_small_linear((const float*)b.ws(t1), C, Bq, C, (const half_t*)b.p(pw2), (const float*)b.p(pb2), C, 1, 1, (float*)b.ws(temb), C, s, bf);
        if (e != hipSuccess) return e;
        return launch_silu_vec((const float*)b.ws(temb), (float*)b.ws(stemb), (long)Bq * C, s);
      });
    }
    // ---- every adaLN modulation of the forward: one stacked linear on silu(temb) ----
    const size_t mod_b = (size_t)Bn * f.mod_total * 4;
    mod = tmp(mod_b);
    {
      const Ref mw = wt(f.mod_all.w), mb = wt(f.mod_all.b);
      const int mt = f.mod_total, bf = m.bf16;
      const size_t mo = mod;
      op("adaln_mod_all", 2.0 * (double)Bn * mt * C, [=](const Bind& b, hipStream_t s) {
        return launch_small_linear((const float*)b.ws(stemb), C, Bq, C, (const half_t*)b.p(mw), (const float*)b.p(mb), mt, 0, 0,
                                   (float*)b.ws(mo), mt, s, bf);
      });
    }
    untmp(tsin, (size_t)Bn * 256 * 4); untmp(t1, vb); untmp(temb, vb); untmp(pv, pv_b);

    // ---- embedders into the fp32 stream [text rows | image rows] (:470, :483) ----
    const size_t xf_b = nr * C * 4;
    xf = tmp(xf_b);
    { Epi e = plain(f.x_emb); e.out32 = stream_rows(nt); e.has_o32 = true; e.ldo32 = C;
      gemm("x_embedder", Ref{BUF_LAT, 0}, d.in_channels, ns, f.x_emb, C, d.in_channels, 0, e); }
    { Epi e = plain(f.ctx_emb); e.out32 = stream_rows(0); e.has_o32 = true; e.ldo32 = C;
      gemm("context_embedder", Ref{BUF_CTX, 0}, d.joint_attention_dim, nt, f.ctx_emb, C, d.joint_attention_dim, 0, e); }

    const int X = px(), x2 = m.x2;                                                  // 'bfloat16x2': operand rows hold [hi | lo]
    // 'float16s': fp16 operands everywhere; the one operand class without an a-priori bound — the MLP hidden tensors gelu(ff_in(.)) — is stored
    // scaled by hs = 2^-8 (fp16 mantissa, range +-1.7e7, absolute resolution 1.5e-5 below 0.016) and the consuming GEMM multiplies its
    // accumulators by 1 / hs.  The single blocks contract over [attn | mlp] rows in ONE GEMM, so their attention output carries the same scale.
    const float hs = m.hid_scale, ihs = hs != 0.f ? 1.0f / hs : 0.f;
    const float hk = hs != 0.f ? ihs : 1.0f;                                        // factor that turns a scaled buffer back into a hook
    const bool f8 = m.fp8 != 0;                                                     // 'fp8-mx': the large linears multiply e4m3 operands
    const size_t ln_b = nr * C * 2 * X, qkv_b = nr * 3 * C * 2;
    // ================= double (MMDiT) blocks =================
    for (int i = 0; i < d.num_layers && !stop; ++i) {
      const FluxDoubleW& w = f.dbl[i];
      const std::string bid = "vit-block" + std::to_string(i);
      // norm1 / norm1_context (AdaLayerNormZero): chunks shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      const size_t ln = tmp(ln_b);
      const MxA ln8 = f8 ? alloc8(nr, C) : MxA{};
      const MxA ln8t = rows8(ln8, 0), ln8i = rows8(ln8, nt);
      adaln("adaln_txt", 0, nt, w.cmod + 0, w.cmod + C, T, 0, 0, ws(ln), f8 ? &ln8t : nullptr);
      adaln("adaln", nt, ns, w.mod + 0, w.mod + C, S, 0, 0, ws(ln + nt * C * 2 * X), f8 ? &ln8i : nullptr);
      const size_t qkv = tmp(qkv_b);
      // un-hooked blocks: RMSNorm(q), RMSNorm(k) + RoPE ride in the QKV GEMM epilogue (on the fp32 accumulators); the `q/k/v`
      // hooks are the PRE-norm projections, so a block that has one of them requested keeps the separate pass
      const bool fuse = !f8 && qkn_fusable(bid) && gemm_qkn_ok((int)nt, 3 * C, C) && gemm_qkn_ok((int)ns, 3 * C, C);   // (the fp8 kernel has no fused RMSNorm + RoPE epilogue)
      { Epi e = plain(w.cqkv); e.out16 = ws(qkv); e.has_o16 = true; e.ldo16 = 3 * C; e.out_f16 = x2;
        if (fuse) qkn(e, w.cnq, w.cnk, 0, T, 0, 0, 1);
        gemm("add_qkv_proj", ws(ln), C * X, nt, w.cqkv, 3 * C, C, 0, e, x2 * C, f8 ? &ln8t : nullptr); }
      { Epi e = plain(w.qkv); e.out16 = ws(qkv + nt * 3 * C * 2); e.has_o16 = true; e.ldo16 = 3 * C; e.out_f16 = x2;
        if (fuse) qkn(e, w.nq, w.nk, T, S, 0, 0, 1);
        gemm("attn_qkv", ws(ln + nt * C * 2 * X), C * X, ns, w.qkv, 3 * C, C, 0, e, x2 * C, f8 ? &ln8i : nullptr); }
      untmp(ln, ln_b);
      if (f8) free8(ln8, nr);
      const size_t qi = qkv + nt * 3 * C * 2;                                        // image rows of the qkv buffer
      const int qbf = x2 ? 0 : -1;                                                   // ('bfloat16x2': the q / k / v buffer is fp16)
      hook_rows16(bid + "-q", ws(qi), 3 * C, C, 0, qbf);                             // attention_processor.py:2283-2286
      hook_rows16(bid + "-k", ws(qi + (size_t)C * 2), 3 * C, C, 0, qbf);
      hook_rows16(bid + "-v", ws(qi + (size_t)2 * C * 2), 3 * C, C, 0, qbf);
      if (stop) { untmp(qkv, qkv_b); break; }
      if (!fuse) {
        qk_norm_rope(qkv, 0, nt, w.cnq, w.cnk, 0, T);                                // norm_added_q/k, text positions
        qk_norm_rope(qkv, nt, ns, w.nq, w.nk, T, S);                                 // norm_q/k, image positions
      }
      const size_t ao = tmp(ln_b);
      int mc = -1, ms = -1;
      map_slots(bid, mc, ms);
      joint_attention(qkv, ws(ao), C * X, mc, ms, x2 * C);
      untmp(qkv, qkv_b);
      const MxA ao8 = f8 ? quant8(ws(ao), C, nr, C) : MxA{};
      const MxA ao8t = rows8(ao8, 0), ao8i = rows8(ao8, nt);
      { Epi e = gated(w.o, nt, w.mod + 2 * C, S);                                    // hidden += gate_msa * to_out(attn)
        e.aux_slot = want(bid + "-attn-out", C, gh, gw); e.ldaux = C;                // :2355-2356 (pre-gate projection)
        gemm("attn_out", ws(ao + nt * C * 2 * X), C * X, ns, w.o, C, C, 0, e, x2 * C, f8 ? &ao8i : nullptr);
        if (e.aux_slot >= 0) hook_done(); }
      { Epi e = gated(w.co, 0, w.cmod + 2 * C, T);                                   // enc += c_gate_msa * to_add_out(attn)
        gemm("attn_add_out", ws(ao), C * X, nt, w.co, C, C, 0, e, x2 * C, f8 ? &ao8t : nullptr); }
      if (f8) free8(ao8, nr);
      untmp(ao, ln_b);
      if (stop) break;
      // ---- image MLP: norm2 + modulate, hooks norm-out / ffn-inner / out (:194-207; `out` stores norm_hidden_states) ----
      const size_t nx = tmp(ns * C * 2 * X);
      const MxA nx8 = f8 ? alloc8(ns, C) : MxA{};
      adaln("adaln", nt, ns, w.mod + 3 * C, w.mod + 4 * C, S, 0, 0, ws(nx), f8 ? &nx8 : nullptr);
      hook_rows16(bid + "-norm-out", ws(nx), C * X, C, x2 * C);
      const size_t inner = tmp(ns * hid * 2 * X);
      { Epi e = plain(w.ff1); e.act = 1; e.out16 = ws(inner); e.has_o16 = true; e.ldo16 = hid * X; e.o16_lo = x2 * hid; e.out16_scale = hs;
        gemm("ff_in", ws(nx), C * X, ns, w.ff1, hid, C, 0, e, x2 * C, f8 ? &nx8 : nullptr); }
      if (f8) free8(nx8, ns);
      hook_rows16(bid + "-ffn-inner", ws(inner), hid * X, hid, x2 * hid, -1, hk);    // attention.py:1255-1257
      { const MxA in8 = f8 ? quant8(ws(inner), hid, ns, hid) : MxA{};
        Epi e = gated(w.ff2, nt, w.mod + 5 * C, S); e.acc_scale = ihs; gemm("ff_out", ws(inner), hid * X, ns, w.ff2, C, hid, 0, e, x2 * hid, f8 ? &in8 : nullptr);
        if (f8) free8(in8, ns); }
      untmp(inner, ns * hid * 2 * X);
      hook_rows16(bid + "-out", ws(nx), C * X, C, x2 * C);
      untmp(nx, ns * C * 2 * X);
      if (stop) break;
      // ---- text MLP (:211-218) ----
      const size_t ne = tmp(nt * C * 2 * X);
      const MxA ne8 = f8 ? alloc8(nt, C) : MxA{};
      adaln("adaln_txt", 0, nt, w.cmod + 3 * C, w.cmod + 4 * C, T, 0, 0, ws(ne), f8 ? &ne8 : nullptr);
      const size_t cin = tmp(nt * hid * 2 * X);
      { Epi e = plain(w.cff1); e.act = 1; e.out16 = ws(cin); e.has_o16 = true; e.ldo16 = hid * X; e.o16_lo = x2 * hid; e.out16_scale = hs;
        gemm("ff_context_in", ws(ne), C * X, nt, w.cff1, hid, C, 0, e, x2 * C, f8 ? &ne8 : nullptr); }
      if (f8) free8(ne8, nt);
      untmp(ne, nt * C * 2 * X);
      { const MxA ci8 = f8 ? quant8(ws(cin), hid, nt, hid) : MxA{};
        Epi e = gated(w.cff2, 0, w.cmod + 5 * C, T); e.acc_scale = ihs; gemm("ff_context_out", ws(cin), hid * X, nt, w.cff2, C, hid, 0, e, x2 * hid, f8 ? &ci8 : nullptr);
        if (f8) free8(ci8, nt); }
      untmp(cin, nt * hid * 2 * X);
    }
    // ================= single blocks over the joint stream =================
    const int CK = C + hid;
    for (int j = 0; j < d.num_single_layers && !stop; ++j) {
      const FluxSingleW& w = f.sgl[j];
      const std::string bid = "vit-block" + std::to_string(d.num_layers + j);
      const size_t ln = tmp(ln_b);
      const MxA ln8 = f8 ? alloc8(nr, C) : MxA{};
      adaln("adaln", 0, nr, w.mod + 0, w.mod + C, T, nt, S, ws(ln), f8 ? &ln8 : nullptr);   // AdaLayerNormZeroSingle: shift, scale, gate
      // 'bfloat16x2': the concatenated operand row is [attn_hi C | mlp_hi hid | attn_lo C | mlp_lo hid]
      const size_t cat_b = nr * CK * 2 * X;
      const size_t qkv = tmp(qkv_b), cat = tmp(cat_b);
      const bool fuse = !f8 && qkn_fusable(bid) && gemm_qkn_ok((int)nr, 3 * C, C);
      { Epi e = plain(w.qkv); e.out16 = ws(qkv); e.has_o16 = true; e.ldo16 = 3 * C; e.out_f16 = x2;
        if (fuse) qkn(e, w.nq, w.nk, 0, T, (int)nt, T, S);
        gemm("attn_qkv", ws(ln), C * X, nr, w.qkv, 3 * C, C, 0, e, x2 * C, f8 ? &ln8 : nullptr); }
      { Epi e = plain(w.mlp); e.act = 1; e.out16 = ws(cat + (size_t)C * 2); e.has_o16 = true; e.ldo16 = CK * X; e.o16_lo = x2 * CK; e.out16_scale = hs;   // :95
        gemm("proj_mlp", ws(ln), C * X, nr, w.mlp, hid, C, 0, e, x2 * C, f8 ? &ln8 : nullptr); }
      untmp(ln, ln_b);
      if (f8) free8(ln8, nr);
      const size_t qi = qkv + nt * 3 * C * 2;
      const int qbf = x2 ? 0 : -1;
      hook_rows16(bid + "-q", ws(qi), 3 * C, C, 0, qbf);                             // :2287-2291 image tokens only
      hook_rows16(bid + "-k", ws(qi + (size_t)C * 2), 3 * C, C, 0, qbf);
      hook_rows16(bid + "-v", ws(qi + (size_t)2 * C * 2), 3 * C, C, 0, qbf);
      if (stop) { untmp(qkv, qkv_b); untmp(cat, cat_b); break; }
      if (!fuse) {
        qk_norm_rope(qkv, 0, nt, w.nq, w.nk, 0, T);
        qk_norm_rope(qkv, nt, ns, w.nq, w.nk, T, S);
      }
      int mc = -1, ms = -1;
      map_slots(bid, mc, ms);
      joint_attention(qkv, ws(cat), CK * X, mc, ms, x2 * CK, hs);                    // cat([attn_output, mlp], 2) in place (:103)
      untmp(qkv, qkv_b);
      hook_rows16(bid + "-attn-out", ws(cat + nt * CK * 2 * X), CK * X, C, x2 * CK, -1, hk); // :2360-2361
      if (cat_b < (1ull << 31)) {
        const MxA c8 = f8 ? quant8(ws(cat), CK, nr, CK) : MxA{};                     // [attn | mlp] rows as e4m3, one scale per row
        Epi e = gated(w.out, 0, w.mod + 2 * C, T, nt, S); e.acc_scale = ihs; gemm("proj_out", ws(cat), CK * X, nr, w.out, C, CK, 0, e, x2 * CK, f8 ? &c8 : nullptr);   // :104-106
        if (f8) free8(c8, nr);
      } else {
        // the A operand is addressed through 32-bit buffer offsets (< 2 GiB): the pair form of [rows][C + hid] at batch 8 is 2.26 GB, so
        // the GEMM runs over row ranges cut at sample boundaries — text rows + the first image samples, then the remaining samples
        const size_t row_b = (size_t)CK * 2 * X;
        const int per = (int)(((1ull << 31) - 1) / row_b);                           // rows per launch
        int k0 = (int)std::min<size_t>((size_t)Bn, per > (int)nt ? (size_t)(per - (int)nt) / S : 0);
        if (per < (int)nt || k0 < 1) { set_error("bfloat16x2: tokens per sample too large for 32-bit buffer offsets"); bad = true; k0 = Bn; }
        { Epi e = gated(w.out, 0, w.mod + 2 * C, T, nt, S);
          gemm("proj_out", ws(cat), CK * X, nt + (size_t)k0 * S, w.out, C, CK, 0, e, x2 * CK); }
        for (int k = k0; k < Bn; ) {
          const int kn = std::min(Bn - k, std::max(1, per / S));
          Epi e = gated(w.out, nt + (size_t)k * S, w.mod + 2 * C + k * f.mod_total, S);      // gate rows of samples k.. (row vector table offset by k samples)
          gemm("proj_out", ws(cat + (nt + (size_t)k * S) * row_b), CK * X, (size_t)kn * S, w.out, C, CK, 0, e, x2 * CK);
          k += kn;
        }
      }
      untmp(cat, cat_b);
      hook_rows32(bid + "-out", stream_rows(nt), C, C);                              // :107-108
    }
    // ================= norm_out (AdaLayerNormContinuous: scale, shift) + proj_out (:591-594) =================
    if (!stop) {
      const size_t no = tmp(ns * C * 2 * X);
      adaln("norm_out", nt, ns, f.mod_out + C, f.mod_out + 0, S, 0, 0, ws(no));
      Epi e = plain(f.proj_out); e.out16 = Ref{BUF_NOISE, 0}; e.has_o16 = true; e.ldo16 = d.in_channels;
      P.writes_noise = true;
      gemm("final_proj_out", ws(no), C * X, ns, f.proj_out, d.in_channels, C, 0, e, x2 * C);
      untmp(no, ns * C * 2 * X);
    }
    untmp(xf, xf_b); untmp(mod, mod_b); untmp(stemb, vb); untmp(cosb, rope_b); untmp(sinb, rope_b);
  }
};

}  // namespace

Model* flux_model_create(const gdf_flux_desc& d) {
  if (d.attention_head_dim != 128 || d.axes_dims_rope[0] + d.axes_dims_rope[1] + d.axes_dims_rope[2] != 128) {
    set_error("attention_head_dim and sum(axes_dims_rope) must be 128"); return nullptr;
  }
  if (d.num_attention_heads < 1 || d.num_layers < 0 || d.num_single_layers < 0 || d.mlp_ratio < 1) { set_error("bad flux desc"); return nullptr; }
  if (d.in_channels % 64 || d.joint_attention_dim % 64 || d.pooled_projection_dim % 8) {
    set_error("in_channels / joint_attention_dim must be multiples of 64, pooled_projection_dim of 8"); return nullptr;
  }
  if (d.compute_dtype != GDF_F16 && d.compute_dtype != GDF_BF16 && d.compute_dtype != GDF_BF16X2 && d.compute_dtype != GDF_FP8MX && d.compute_dtype != GDF_F16S) {
    set_error("compute_dtype must be GDF_F16, GDF_F16S, GDF_BF16, GDF_BF16X2 or GDF_FP8MX"); return nullptr;
  }
  Model* m = new Model();
  m->kind = 1;
  m->bf16 = d.compute_dtype != GDF_F16 && d.compute_dtype != GDF_F16S;
  m->hid_scale = d.compute_dtype == GDF_F16S ? 1.0f / 256.0f : 0.f;
  m->x2 = d.compute_dtype == GDF_BF16X2;
  m->fp8 = d.compute_dtype == GDF_FP8MX;
  m->flux.d = d;
  m->flux.D = d.attention_head_dim;
  m->flux.C = d.num_attention_heads * d.attention_head_dim;
  m->flux.hid = m->flux.C * d.mlp_ratio;
  FluxModelBuilder b(*m);
  b.build();
  if (m->fp8) {                                  // fp8 copies + per-channel scales live in the same arena (one blob for the data-parallel broadcast)
    const size_t a = align_up(m->weight_bytes, 4096);
    m->f8_off = a; m->sc_off = a + a / 2;
    m->weight_bytes = a + a / 2 + a / 16 + 4096;
  }
  { CaptureExclusive guard; if (hipMalloc(&m->weights, m->weight_bytes) != hipSuccess) { set_error("hipMalloc(weights) failed"); delete m; return nullptr; } }
  (void)hipMemset(m->weights, 0, m->weight_bytes);
  PlanOpts o{}; o.stream_fp32 = 1;
  Plan dry;
  flux_plan_build(*m, dry, 1, 4, 4, 8, nullptr, 0, o, /*dry=*/true);
  m->hook_names = dry.dry_ids;
  return m;
}

int flux_plan_build(const Model& m, Plan& P, int batch, int img_h, int img_w, int n_txt, const char* const* ids, int n_ids,
                    const PlanOpts& opts, bool dry) {
  if (m.kind != 1) { set_error("not a Flux model"); return GDF_ERR_ARG; }
  if (batch < 1 || img_h < 1 || img_w < 1 || n_txt < 1) { set_error("batch, token grid and n_txt must be positive"); return GDF_ERR_ARG; }
  for (int i = 0; i < n_ids; ++i)
    if (ids[i] && strstr(ids[i], "-map") && (n_txt % 8)) { set_error("'-map' hooks need n_txt % 8 == 0"); return GDF_ERR_UNSUPPORTED; }
  const size_t rows = (size_t)batch * ((size_t)img_h * img_w + n_txt);
  // 32-bit buffer offsets of the GEMM A operands ('bfloat16x2': the widest one, the single blocks' [C + hid] pair rows, is cut into row
  // ranges by the builder; the next widest is the MLP hidden pair [2 hid])
  const size_t widest = m.x2 ? (size_t)m.flux.hid * 2 : (size_t)(m.flux.C + m.flux.hid);
  if (rows * widest * 2 >= (1ull << 31)) {
    set_error("batch * tokens too large for 32-bit buffer offsets; split the batch"); return GDF_ERR_UNSUPPORTED;
  }
  P.batch = batch; P.H = img_h; P.W = img_w; P.n_ctx = n_txt; P.opts = opts;
  FB b(m, P, dry, opts);
  b.Bn = batch; b.n_ctx = n_txt; b.T = n_txt; b.S = img_h * img_w; b.gh = img_h; b.gw = img_w;
  if (!dry) {
    std::unordered_set<std::string> known(m.hook_names.begin(), m.hook_names.end());
    for (int i = 0; i < n_ids; ++i)
      if (ids[i] && known.count(ids[i])) P.requested.insert(ids[i]);     // unknown ids silently ignored (feature_extractor.py:36)
    b.remaining = (int)P.requested.size();
    if (opts.early_exit && b.remaining == 0) b.stop = true;
  }
  b.build();
  if (b.bad) return GDF_ERR_UNSUPPORTED;
  P.ws_bytes = b.ar.peak + 256;
  return GDF_OK;
}

int flux_forward(Plan& P, const Model& m, const void* hidden, const void* enc, const void* pooled, const float* timestep,
                 const float* guidance, const float* img_ids, const float* txt_ids, void* const* hook_out, void* out,
                 void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap) {
  if (m.kind != 1) { set_error("gdf_flux_forward on a UNet model"); return GDF_ERR_STATE; }
  if (m.n_set != (int)m.params.size()) { set_error("model weights incomplete"); return GDF_ERR_STATE; }
  if (!hidden || !enc || !pooled || !timestep || !img_ids || !txt_ids || !ws) { set_error("null input pointer"); return GDF_ERR_ARG; }
  if (m.flux.d.guidance_embeds && !guidance) { set_error("guidance is required (guidance_embeds)"); return GDF_ERR_ARG; }
  if (P.hooks.size() && !hook_out) { set_error("hook_out is null"); return GDF_ERR_ARG; }
  if (P.writes_noise && !out) { set_error("output buffer required (the plan runs proj_out)"); return GDF_ERR_ARG; }
  Bind b;
  b.base[BUF_WS] = (char*)ws; b.base[BUF_WT] = (char*)m.weights; b.base[BUF_LAT] = (char*)hidden; b.base[BUF_T] = (char*)timestep;
  b.base[BUF_CTX] = (char*)enc; b.base[BUF_TXT] = (char*)pooled; b.base[BUF_TID] = (char*)guidance; b.base[BUF_NOISE] = (char*)out;
  b.base[BUF_IDS_IMG] = (char*)img_ids; b.base[BUF_IDS_TXT] = (char*)txt_ids;
  b.hooks = hook_out;
  return plan_run(P, b, s, ms, names, flops, cap);
}

}  // namespace gdf
