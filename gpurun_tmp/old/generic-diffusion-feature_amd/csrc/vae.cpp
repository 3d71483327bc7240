// VAE-encoder front end of libgdf.so (include/gdf_vae.h): weight layout, op program, encode entry.
//
// Restates the encoder half of diffusers==0.32.2 AutoencoderKL (un-vendored; wiring from the published algorithm, see
// oracle/vae_ref.py) out of blocks that ARE in the reference tree (paths under /root/reference/feature/diffusers/models):
//   ResnetBlock2D(temb=None, eps 1e-6)                         resnet.py:320-379      -> PlanBuilder::resnet
//   Downsample2D(padding=0): F.pad (0,1,0,1) + conv stride 2   downsampling.py:141-152 -> conv3 with pad0 (address generator)
//   Attention(1 head, dim_head = C, GroupNorm, residual)       attention_processor.py:3244-3331
// plus DiagonalGaussianDistribution.sample, scaling_factor, scheduler.add_noise and scale_model_input
// (call sites /root/reference/feature/diffusion_feature.py:371-380, :405-406) fused into one tail kernel.
//
// The mid-block attention has ONE head of dim 512 over (H/8 * W/8) tokens (16384 at 1024^2): too wide for the flash
// kernel's register tile, so it runs as MFMA GEMMs per image: S = Q K^T (fp16, 537 MB at 1024^2), in-place row softmax,
// O = P V with V^T produced directly by a GEMM whose A operand is W_v (V^T = W_v X^T); the V bias is added after the
// attention as a column bias (rows of P sum to 1, so P (V + 1 b^T) = P V + 1 b^T exactly).
#include "builder.h"

namespace gdf {

namespace {

struct VaeModelBuilder : WeightBuilder {
  explicit VaeModelBuilder(Model& mm) : WeightBuilder(mm) {}

  ResnetW resnet(const std::string& p, int ci, int co) {
    ResnetW r; r.cin = ci; r.cout = co; r.has_temb = false; r.eps = 1e-6f;
    r.n1 = norm(p + ".norm1", ci);
    r.c1 = conv3(p + ".conv1", co, ci);
    r.n2 = norm(p + ".norm2", co);
    r.c2 = conv3(p + ".conv2", co, co);
    r.has_sc = ci != co;
    if (r.has_sc) r.sc = lin(p + ".conv_shortcut", co, ci, true, true);
    return r;
  }

  void build() {
    VaeW& v = m.vae;
    const gdf_vae_desc& d = v.d;
    const int L = d.n_levels, nl = d.layers_per_block;
    const int* boc = d.block_out_channels;
    v.conv_in.cin = d.in_channels; v.conv_in.cout = boc[0];
    v.conv_in.w = take((size_t)boc[0] * 128 * 2); v.conv_in.b = take(boc[0] * 4);
    reg("encoder.conv_in.weight", {boc[0], d.in_channels, 3, 3}, PK_CONV_IN, v.conv_in.w, boc[0], d.in_channels);
    reg("encoder.conv_in.bias", {boc[0]}, PK_VEC, v.conv_in.b);
    int ci = boc[0];
    for (int lv = 0; lv < L; ++lv) {
      std::vector<ResnetW> rs;
      for (int r = 0; r < nl; ++r) {
        rs.push_back(resnet("encoder.down_blocks." + std::to_string(lv) + ".resnets." + std::to_string(r), ci, boc[lv]));
        ci = boc[lv];
      }
      v.down.push_back(rs);
      if (lv != L - 1) v.downsamplers.push_back(conv3("encoder.down_blocks." + std::to_string(lv) + ".downsamplers.0.conv", boc[lv], boc[lv]));
    }
    const int c = boc[L - 1];
    v.mid0 = resnet("encoder.mid_block.resnets.0", c, c);
    const std::string a = "encoder.mid_block.attentions.0";
    v.attn_gn = norm(a + ".group_norm", c);
    v.q = lin(a + ".to_q", c, c); v.k = lin(a + ".to_k", c, c); v.v = lin(a + ".to_v", c, c); v.o = lin(a + ".to_out.0", c, c);
    v.mid1 = resnet("encoder.mid_block.resnets.1", c, c);
    v.norm_out = norm("encoder.conv_norm_out", c);
    v.conv_out = conv3("encoder.conv_out", 2 * d.latent_channels, c);
    if (d.use_quant_conv) v.quant = lin("quant_conv", 2 * d.latent_channels, 2 * d.latent_channels, true, true);
    m.weight_bytes = cur;
  }

  // decoder half: `vae.state_dict()` names "post_quant_conv.*" and "decoder.*" (AutoencoderKL.decode, un-vendored diffusers==0.32.2:
  // post_quant_conv -> Decoder(conv_in, UNetMidBlock2D, UpDecoderBlock2D x L, conv_norm_out + SiLU, conv_out); oracle/vae_ref.py decode)
  void build_decoder() {
    VaeW& v = m.vae;
    const gdf_vae_desc& d = v.d;
    const int L = d.n_levels, nl = d.layers_per_block;
    const int* boc = d.block_out_channels;
    if (d.use_quant_conv) v.post_quant = lin("post_quant_conv", d.latent_channels, d.latent_channels, true, true);
    const int c = boc[L - 1];
    v.conv_in.cin = d.latent_channels; v.conv_in.cout = c;
    v.conv_in.w = take((size_t)c * 128 * 2); v.conv_in.b = take(c * 4);
    reg("decoder.conv_in.weight", {c, d.latent_channels, 3, 3}, PK_CONV_IN, v.conv_in.w, c, d.latent_channels);
    reg("decoder.conv_in.bias", {c}, PK_VEC, v.conv_in.b);
    v.mid0 = resnet("decoder.mid_block.resnets.0", c, c);
    const std::string a = "decoder.mid_block.attentions.0";
    v.attn_gn = norm(a + ".group_norm", c);
    v.q = lin(a + ".to_q", c, c); v.k = lin(a + ".to_k", c, c); v.v = lin(a + ".to_v", c, c); v.o = lin(a + ".to_out.0", c, c);
    v.mid1 = resnet("decoder.mid_block.resnets.1", c, c);
    int ci = c;
    for (int i = 0; i < L; ++i) {
      const int co = boc[L - 1 - i];
      std::vector<ResnetW> rs;
      for (int r = 0; r < nl + 1; ++r) {
        rs.push_back(resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(r), ci, co));
        ci = co;
      }
      v.up.push_back(rs);
      if (i != L - 1) v.upsamplers.push_back(conv3("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co));
    }
    v.norm_out = norm("decoder.conv_norm_out", boc[0]);
    v.conv_out = conv3("decoder.conv_out", d.in_channels, boc[0]);
    m.weight_bytes = cur;
  }
};

struct VB : PlanBuilder {
  const VaeW& v;
  VB(const Model& mm, Plan& pp, bool d, const PlanOpts& o) : PlanBuilder(mm, pp, d, o), v(mm.vae) {}

  // GEMM with both operands given as buffer references (activation x activation products of the mid attention)
  void gemm_raw(const char* name, Ref A, int lda, size_t M, Ref W, int N, int K, const Epi& e0) {
    const Epi e = e0;
    GemmParams gk{}; gk.M = (int)M; gk.N = N; gk.K = K; gk.mode = A_DENSE; gk.bn = e.bn;
    op(name, 2.0 * (double)M * N * K, [=](const Bind& b, hipStream_t s) {
      GemmParams g{};
      g.A = (const half_t*)b.p(A); g.lda = lda; g.a_bytes = (uint32_t)(((size_t)M - 1) * lda * 2 + (size_t)K * 2);
      g.M = (int)M; g.N = N; g.K = K; g.mode = A_DENSE;
      g.Wt = (const half_t*)b.p(W); g.w_bytes = (uint32_t)((size_t)N * K * 2);
      fill_epi(g, e, b);
      return launch_gemm(g, s);
    }, gemm_kernel_name(gk));
  }

  void mid_attention(const Act& x, Act& y) {
    const int C = x.C, S = x.H * x.W;
    const size_t n = rows(x), nb = n * C * 2;
    const size_t gn = groupnorm(x, v.attn_gn, 1e-6f, false);
    const size_t q = tmp(nb), k = tmp(nb), ao = tmp(nb);
    { Epi e; e.bias = wt(v.q.b); e.has_bias = true; e.out16 = ws(q); e.has_o16 = true; e.ldo16 = C; gemm("vae_attn_q", ws(gn), C, n, v.q, C, C, 0, e); }
    { Epi e; e.bias = wt(v.k.b); e.has_bias = true; e.out16 = ws(k); e.has_o16 = true; e.ldo16 = C; gemm("vae_attn_k", ws(gn), C, n, v.k, C, C, 0, e); }
    const size_t p_b = (size_t)S * S * 2, vt_b = (size_t)C * S * 2;
    const size_t pm = tmp(p_b), vt = tmp(vt_b);
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < Bn; ++b) {
      const size_t ro = (size_t)b * S * C * 2;                    // byte offset of image b's rows
      { Epi e; e.out16 = ws(pm); e.has_o16 = true; e.ldo16 = S;     // scores = scale * Q_b K_b^T (scaled BEFORE the fp16 store: range)
        e.acc_scale = scale;
        gemm_raw("vae_attn_qk", ws(q + ro), C, S, ws(k + ro), S, C, e); }
      op("vae_attn_softmax", 0, [=](const Bind& bd, hipStream_t s) {
        return launch_softmax_rows((half_t*)bd.ws(pm), S, S, S, 1.0f, s);
      });
      { Epi e; e.out16 = ws(vt); e.has_o16 = true; e.ldo16 = S;     // V_b^T = W_v X_b^T  (A operand = the weight matrix)
        gemm_raw("vae_attn_vt", wt(v.v.w), C, C, ws(gn + ro), S, C, e); }
      { Epi e; e.bias = wt(v.v.b); e.has_bias = true; e.out16 = ws(ao + ro); e.has_o16 = true; e.ldo16 = C;   // O_b = P V_b + b_v
        gemm_raw("vae_attn_pv", ws(pm), S, S, ws(vt), C, S, e); }
    }
    untmp(pm, p_b); untmp(vt, vt_b); untmp(q, nb); untmp(k, nb); untmp(gn, nb);
    { Epi e; e.bias = wt(v.o.b); e.has_bias = true; residual_from(e, x); out_to(e, y);
      gemm("vae_attn_out", ws(ao), C, n, v.o, C, C, 0, e); }
    untmp(ao, nb);
  }

  void build(int H, int W) {
    const gdf_vae_desc& d = v.d;
    const int L = d.n_levels, nl = d.layers_per_block, Bq = Bn;
    const int* boc = d.block_out_channels;
    // ---- conv_in on the NCHW image ----
    const size_t x8_b = (size_t)Bn * H * W * 16, x8 = tmp(x8_b);
    {
      const int cin = d.in_channels;
      op("pack_image", 0, [=](const Bind& b, hipStream_t s) {
        return launch_pack_latents((const half_t*)b.base[BUF_LAT], Bq, cin, H, W, (half_t*)b.ws(x8), nullptr, s);
      });
    }
    Act cur = new_act(boc[0], H, W, true);
    {
      Epi e; e.bias = wt(v.conv_in.b); e.has_bias = true; out_to(e, cur);
      const Ref Wr = wt(v.conv_in.w); const int N = boc[0];
      const size_t M = (size_t)Bn * H * W;
      size_t gp = NPOS;                                    // GroupNorm partial sums of conv_in's output (builder.h conv3, gn_epi)
      {
        GemmParams gk{}; gk.M = (int)M; gk.N = N; gk.K = 128; gk.mode = A_CONV_SMALLC;
        const int sr = (gn_epi && e.has_o16) ? gemm_gn_slab_rows(gk) : 0;
        if (sr > 0 && (H * W) % sr == 0) {
          cur.gp_rows = sr; cur.gp_bytes = M / sr * (size_t)N * 8;
          cur.gp_alloc = gp = dry ? 0 : ar.alloc(cur.gp_bytes);
        }
      }
      op("vae_conv_in", 2.0 * (double)M * N * 9 * d.in_channels, [=](const Bind& b, hipStream_t s) {
        GemmParams g{};
        if (gp != NPOS) g.gn_partial = (float*)b.ws(gp);
        g.A = (const half_t*)b.ws(x8); g.lda = 8; g.a_bytes = (uint32_t)(M * 16);
        g.M = (int)M; g.N = N; g.K = 128; g.mode = A_CONV_SMALLC; g.H = H; g.W = W; g.OH = H; g.OW = W; g.stride = 1; g.Cin = 8;
        g.Wt = (const half_t*)b.p(Wr); g.w_bytes = (uint32_t)((size_t)N * 128 * 2);
        fill_epi(g, e, b);
        return launch_gemm(g, s);
      });
    }
    untmp(x8, x8_b);
    int hh = H, ww = W;
    for (int lv = 0; lv < L; ++lv) {
      for (int r = 0; r < nl; ++r) {
        Act nxt = new_act(boc[lv], hh, ww, true);
        resnet("", v.down[lv][r], cur, nxt);
        free_act(cur);
        cur = nxt;
      }
      if (lv != L - 1) {
        Act nxt = new_act(boc[lv], hh / 2, ww / 2, true);
        Epi e; e.bias = wt(v.downsamplers[lv].b); e.has_bias = true; e.pad0 = 1; out_to(e, nxt);
        reads_image(e, cur);
        conv3("vae_downsample", cur.h, cur.ld, cur.C, hh, ww, 2, false, v.downsamplers[lv], e, 0, &nxt);     // downsampling.py:141-152
        free_act(cur);
        cur = nxt; hh /= 2; ww /= 2;
      }
    }
    const int c = boc[L - 1];
    { Act nxt = new_act(c, hh, ww, true); resnet("", v.mid0, cur, nxt); free_act(cur); cur = nxt; }
    { Act nxt = new_act(c, hh, ww, true); mid_attention(cur, nxt); free_act(cur); cur = nxt; }
    { Act nxt = new_act(c, hh, ww, true); resnet("", v.mid1, cur, nxt); free_act(cur); cur = nxt; }
    // ---- conv_norm_out + SiLU + conv_out (fp32 moments) + quant_conv / sample / scale / noise tail ----
    const size_t n = rows(cur);
    const int L2 = 2 * d.latent_channels;
    const size_t no = groupnorm(cur, v.norm_out, 1e-6f, true);
    const size_t mo_b = n * L2 * 4, mo = tmp(mo_b);
    { Epi e; e.bias = wt(v.conv_out.b); e.has_bias = true; e.bn = 16; e.out32 = ws(mo); e.has_o32 = true; e.ldo32 = L2;
      conv3("vae_conv_out", ws(no), c, c, hh, ww, 1, false, v.conv_out, e); }
    untmp(no, n * c * 2);
    free_act(cur);
    {
      const int HW = hh * ww, Lc = d.latent_channels;
      const bool uq = d.use_quant_conv != 0;
      const Ref qw = wt(v.quant.w), qb = wt(v.quant.b);
      op("vae_sample_noise", 0, [=](const Bind& b, hipStream_t s) {
        return launch_vae_finish((const float*)b.ws(mo), Bq, HW, Lc, uq ? (const half_t*)b.p(qw) : nullptr,
                                 uq ? (const float*)b.p(qb) : nullptr, (const half_t*)b.base[BUF_TXT], (const half_t*)b.base[BUF_CTX],
                                 b.f[0], b.f[1], b.f[2], b.f[3], (half_t*)b.base[BUF_NOISE], s);
      });
    }
    untmp(mo, mo_b);
  }

  // ---- decoder op program (`vae-out`): latents + noise_pred -> image ----
  // BUF_LAT = latents (B, L, h, w) fp16 NCHW, BUF_CTX = noise_pred (same shape) or NULL, BUF_NOISE = image out (B, H, W, 3) fp16
  // channels-last; Bind::f = {c_sample, c_eps, 1 / scaling_factor}
  void build_decoder(int h, int w) {
    const gdf_vae_desc& d = v.d;
    const int L = d.n_levels, nl = d.layers_per_block, Bq = Bn;
    const int* boc = d.block_out_channels;
    const int c = boc[L - 1];
    const size_t x8_b = (size_t)Bn * h * w * 16, x8 = tmp(x8_b);
    {
      const int Lc = d.latent_channels, HW = h * w;
      const bool pq = d.use_quant_conv != 0;
      const Ref qw = wt(v.post_quant.w), qb = wt(v.post_quant.b);
      op("vae_dec_prepare", 0, [=](const Bind& b, hipStream_t s) {
        return launch_vae_dec_prepare((const half_t*)b.base[BUF_LAT], (const half_t*)b.base[BUF_CTX], Bq, HW, Lc, b.f[0], b.f[1], b.f[2],
                                      pq ? (const half_t*)b.p(qw) : nullptr, pq ? (const float*)b.p(qb) : nullptr, (half_t*)b.ws(x8), s);
      });
    }
    Act cur = new_act(c, h, w, true);
    {
      Epi e; e.bias = wt(v.conv_in.b); e.has_bias = true; out_to(e, cur);
      const Ref Wr = wt(v.conv_in.w); const int N = c;
      const size_t M = (size_t)Bn * h * w;
      op("vae_dec_conv_in", 2.0 * (double)M * N * 9 * d.latent_channels, [=](const Bind& b, hipStream_t s) {
        GemmParams g{};
        g.A = (const half_t*)b.ws(x8); g.lda = 8; g.a_bytes = (uint32_t)(M * 16);
        g.M = (int)M; g.N = N; g.K = 128; g.mode = A_CONV_SMALLC; g.H = h; g.W = w; g.OH = h; g.OW = w; g.stride = 1; g.Cin = 8;
        g.Wt = (const half_t*)b.p(Wr); g.w_bytes = (uint32_t)((size_t)N * 128 * 2);
        fill_epi(g, e, b);
        return launch_gemm(g, s);
      });
    }
    untmp(x8, x8_b);
    { Act nxt = new_act(c, h, w, true); resnet("", v.mid0, cur, nxt); free_act(cur); cur = nxt; }
    { Act nxt = new_act(c, h, w, true); mid_attention(cur, nxt); free_act(cur); cur = nxt; }
    { Act nxt = new_act(c, h, w, true); resnet("", v.mid1, cur, nxt); free_act(cur); cur = nxt; }
    int hh = h, ww = w;
    for (int i = 0; i < L; ++i) {
      const int co = boc[L - 1 - i];
      for (int r = 0; r < nl + 1; ++r) {
        Act nxt = new_act(co, hh, ww, true);
        resnet("", v.up[i][r], cur, nxt);
        free_act(cur);
        cur = nxt;
      }
      if (i != L - 1) {                                                                      // Upsample2D: nearest x2 fused into the conv
        Act nxt = new_act(co, hh * 2, ww * 2, true);
        Epi e; e.bias = wt(v.upsamplers[i].b); e.has_bias = true; out_to(e, nxt);
        reads_image(e, cur);
        conv3("vae_upsample", cur.h, cur.ld, cur.C, hh, ww, 1, true, v.upsamplers[i], e, 0, &nxt);    // upsampling.py:176-193
        free_act(cur);
        cur = nxt; hh *= 2; ww *= 2;
      }
    }
    const size_t n = rows(cur);
    const size_t no = groupnorm(cur, v.norm_out, 1e-6f, true);
    { Epi e; e.bias = wt(v.conv_out.b); e.has_bias = true; e.bn = 16;
      e.out16 = Ref{BUF_NOISE, 0}; e.has_o16 = true; e.ldo16 = d.in_channels;
      conv3("vae_dec_conv_out", ws(no), cur.C, cur.C, hh, ww, 1, false, v.conv_out, e); }
    untmp(no, n * cur.C * 2);
    free_act(cur);
  }
};

}  // namespace

// GroupNorm statistics from the producing conv's epilogue (builder.h gn_epi; round 4).  At 1024^2 every VAE activation is a 1-4 GB tensor that
// no cache holds, so the separate statistics pass is a full HBM read; GDF_VAE_GN_EPI=0 restores it (same-process A/B, tools/bench_vae.py).
static bool gn_from_epilogue() {
  const char* e = getenv("GDF_VAE_GN_EPI");
  return !(e && e[0] == '0');
}

Model* vae_model_create(const gdf_vae_desc& d) {
  if (d.n_levels < 1 || d.n_levels > GDF_MAX_LEVELS || d.in_channels < 1 || d.in_channels > 8 || d.latent_channels < 1 ||
      d.latent_channels > 8 || d.layers_per_block < 1) { set_error("bad vae desc"); return nullptr; }
  for (int i = 0; i < d.n_levels; ++i)
    if (d.block_out_channels[i] % 64) { set_error("block_out_channels must be multiples of 64"); return nullptr; }
  Model* m = new Model();
  m->kind = 2;
  m->vae.d = d;
  VaeModelBuilder b(*m);
  b.build();
  { CaptureExclusive guard; if (hipMalloc(&m->weights, m->weight_bytes) != hipSuccess) { set_error("hipMalloc(weights) failed"); delete m; return nullptr; } }
  (void)hipMemset(m->weights, 0, m->weight_bytes);
  return m;
}

int vae_plan_build(const Model& m, Plan& P, int batch, int img_h, int img_w, bool dry) {
  if (m.kind != 2) { set_error("not a VAE model"); return GDF_ERR_ARG; }
  const int L = m.vae.d.n_levels, down = 1 << (L - 1);
  if (batch < 1 || img_h < 1 || img_w < 1 || (img_h % (8 * down)) || (img_w % (8 * down))) {   // latent grid multiple of 8: S % 64 == 0
    set_error("image size must be a positive multiple of 8 * 2^(levels-1)"); return GDF_ERR_ARG;
  }
  const size_t S = (size_t)(img_h / down) * (img_w / down);
  if (S > 16384) { set_error("mid-block attention supports up to 16384 latent tokens (1024^2 images)"); return GDF_ERR_UNSUPPORTED; }
  // sub-batch: largest divisor of `batch` whose widest fp16 activation stays below the 32-bit buffer-offset limit
  size_t widest = 0;
  for (int lv = 0; lv < L; ++lv)
    widest = std::max(widest, (size_t)(img_h >> lv) * (img_w >> lv) * m.vae.d.block_out_channels[lv] * 2);
  int chunk = 1;
  for (int c = 1; c <= batch; ++c)
    if (batch % c == 0 && (size_t)c * widest < (1ull << 31)) chunk = c;
  if (widest >= (1ull << 31)) { set_error("image too large for 32-bit buffer offsets"); return GDF_ERR_UNSUPPORTED; }
  P.batch = batch; P.chunk = chunk; P.H = img_h; P.W = img_w;
  PlanOpts o{}; o.stream_fp32 = 1;
  P.opts = o;
  VB b(m, P, dry, P.opts);
  b.Bn = chunk;
  // Range: the reference upcasts the SDXL VAE to fp32 because its residual stream exceeds 65504.  Here the stream lives in
  // fp32 masters; its fp16 images (GroupNorm input, shortcut / downsample conv operand) and the resnet-internal conv1 output
  // are stored scaled by 2^-6 (max magnitude 4.2e6), exactly undone by their consumers (builder.h act_scale).
  b.act_scale = 1.0f / 64.0f;
  b.gn_epi = gn_from_epilogue();
  b.build(img_h, img_w);
  P.ws_bytes = b.ar.peak + 256;
  return GDF_OK;
}

int vae_encode(Plan& P, const Model& m, const void* image, const void* eps, const void* noise, float scaling, float noise_a,
               float noise_b, float in_scale, void* out, void* ws, hipStream_t s, float* ms, const char** names, double* flops,
               int cap) {
  if (m.kind != 2) { set_error("gdf_vae_encode on a non-VAE model"); return GDF_ERR_STATE; }
  if (m.n_set != (int)m.params.size()) { set_error("model weights incomplete"); return GDF_ERR_STATE; }
  if (!image || !out || !ws) { set_error("null input pointer"); return GDF_ERR_ARG; }
  const gdf_vae_desc& d = m.vae.d;
  const size_t img_b = (size_t)d.in_channels * P.H * P.W * 2;                       // bytes per image
  const int f = 1 << (d.n_levels - 1);                                               // spatial reduction (8 for the SD VAEs)
  const size_t lat_b = (size_t)d.latent_channels * (P.H / f) * (P.W / f) * 2;        // bytes per latent
  for (int c0 = 0; c0 < P.batch; c0 += P.chunk) {
    Bind b;
    b.base[BUF_WS] = (char*)ws; b.base[BUF_WT] = (char*)m.weights;
    b.base[BUF_LAT] = (char*)image + (size_t)c0 * img_b;
    b.base[BUF_TXT] = eps ? (char*)eps + (size_t)c0 * lat_b : nullptr;
    b.base[BUF_CTX] = noise ? (char*)noise + (size_t)c0 * lat_b : nullptr;
    b.base[BUF_NOISE] = (char*)out + (size_t)c0 * lat_b;
    b.f[0] = scaling; b.f[1] = noise_a; b.f[2] = noise_b; b.f[3] = in_scale;
    const int rc = plan_run(P, b, s, ms, names, flops, cap);
    if (rc != GDF_OK) return rc;
    if (ms) break;                                                                  // profile: one sub-batch pass
  }
  return GDF_OK;
}

}  // namespace gdf

// =====================================================================================================================
// VAE decoder (`vae-out`): reference feature/diffusion_feature.py:60, :477-485
// =====================================================================================================================
namespace gdf {

Model* vae_decoder_create(const gdf_vae_desc& d) {
  if (d.n_levels < 1 || d.n_levels > GDF_MAX_LEVELS || d.in_channels < 1 || d.in_channels > 8 || d.latent_channels < 1 ||
      d.latent_channels > 8 || d.layers_per_block < 1) { set_error("bad vae desc"); return nullptr; }
  for (int i = 0; i < d.n_levels; ++i)
    if (d.block_out_channels[i] % 64) { set_error("block_out_channels must be multiples of 64"); return nullptr; }
  Model* m = new Model();
  m->kind = 4;
  m->vae.d = d;
  VaeModelBuilder b(*m);
  b.build_decoder();
  { CaptureExclusive guard; if (hipMalloc(&m->weights, m->weight_bytes) != hipSuccess) { set_error("hipMalloc(weights) failed"); delete m; return nullptr; } }
  (void)hipMemset(m->weights, 0, m->weight_bytes);
  return m;
}

int vae_dec_plan_build(const Model& m, Plan& P, int batch, int lat_h, int lat_w, bool dry) {
  if (m.kind != 4) { set_error("not a VAE decoder model"); return GDF_ERR_ARG; }
  const int L = m.vae.d.n_levels, upf = 1 << (L - 1);
  if (batch < 1 || lat_h < 8 || lat_w < 8 || (lat_h % 8) || (lat_w % 8)) {                     // latent grid multiple of 8: S % 64 == 0
    set_error("latent size must be a positive multiple of 8"); return GDF_ERR_ARG;
  }
  const size_t S = (size_t)lat_h * lat_w;
  if (S > 16384) { set_error("mid-block attention supports up to 16384 latent tokens (1024^2 images)"); return GDF_ERR_UNSUPPORTED; }
  // sub-batch: largest divisor of `batch` whose widest fp16 activation stays below the 32-bit buffer-offset limit.  Up block i works
  // at (lat << i) with boc[L-1-i] channels; its upsampler's OUTPUT has the same channels at twice the size
  size_t widest = 0;
  for (int i = 0; i < L; ++i) {
    const size_t hw = (size_t)(lat_h << i) * (lat_w << i), c = m.vae.d.block_out_channels[L - 1 - i];
    widest = std::max(widest, hw * c * 2 * (i != L - 1 ? 4 : 1));
  }
  if (widest >= (1ull << 31)) { set_error("image too large for 32-bit buffer offsets"); return GDF_ERR_UNSUPPORTED; }
  int chunk = 1;
  for (int c = 1; c <= batch; ++c)
    if (batch % c == 0 && (size_t)c * widest < (1ull << 31)) chunk = c;
  P.batch = batch; P.chunk = chunk; P.H = lat_h; P.W = lat_w;
  (void)upf;
  PlanOpts o{}; o.stream_fp32 = 1;
  P.opts = o;
  VB b(m, P, dry, P.opts);
  b.Bn = chunk;
  b.act_scale = 1.0f / 64.0f;          // fp16 range control of the stream images, as in the encoder (vae_plan_build)
  b.gn_epi = gn_from_epilogue();
  b.build_decoder(lat_h, lat_w);
  P.ws_bytes = b.ar.peak + 256;
  return GDF_OK;
}

int vae_decode(Plan& P, const Model& m, const void* latents, const void* noise_pred, float c_sample, float c_eps, float inv_scaling,
               void* image_out, void* ws, hipStream_t s, float* ms, const char** names, double* flops, int cap) {
  if (m.kind != 4) { set_error("gdf_vae_decode on a model that is not a VAE decoder"); return GDF_ERR_STATE; }
  if (m.n_set != (int)m.params.size()) { set_error("model weights incomplete"); return GDF_ERR_STATE; }
  if (!latents || !image_out || !ws) { set_error("null input pointer"); return GDF_ERR_ARG; }
  const gdf_vae_desc& d = m.vae.d;
  const int f = 1 << (d.n_levels - 1);
  const size_t lat_b = (size_t)d.latent_channels * P.H * P.W * 2;                    // bytes per latent
  const size_t img_b = (size_t)d.in_channels * (P.H * f) * (P.W * f) * 2;            // bytes per image
  for (int c0 = 0; c0 < P.batch; c0 += P.chunk) {
    Bind b;
    b.base[BUF_WS] = (char*)ws; b.base[BUF_WT] = (char*)m.weights;
    b.base[BUF_LAT] = (char*)latents + (size_t)c0 * lat_b;
    b.base[BUF_CTX] = noise_pred ? (char*)noise_pred + (size_t)c0 * lat_b : nullptr;
    b.base[BUF_NOISE] = (char*)image_out + (size_t)c0 * img_b;
    b.f[0] = c_sample; b.f[1] = c_eps; b.f[2] = inv_scaling;
    const int rc = plan_run(P, b, s, ms, names, flops, cap);
    if (rc != GDF_OK) return rc;
    if (ms) break;                                                                  // profile: one sub-batch pass
  }
  return GDF_OK;
}

}  // namespace gdf
