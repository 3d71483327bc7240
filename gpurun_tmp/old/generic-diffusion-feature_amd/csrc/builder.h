// Shared host-side builders of libgdf.so: weight-arena layout (WeightBuilder) and the static plan builder base
// (PlanBuilder: workspace arena with liveness, activation views, hook table, GEMM / hook-copy op emitters).
// The UNet (model.cpp) and the MMDiT / Flux (flux.cpp) op programs derive from these.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "model.h"

namespace gdf {

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- weight arena layout + diffusers parameter-name registration ---------------------------------
struct WeightBuilder {
  Model& m;
  size_t cur = 0;
  explicit WeightBuilder(Model& mm) : m(mm) {}

  size_t take(size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; }

  void reg(const std::string& name, std::initializer_list<int64_t> shape, int kind, size_t dst, int a0 = 0, int a1 = 0,
           int a2 = 0) {
    ParamRec p;
    p.name = name; p.ndim = (int)shape.size();
    int i = 0;
    for (auto s : shape) p.shape[i++] = s;
    p.kind = kind; p.dst = dst; p.a0 = a0; p.a1 = a1; p.a2 = a2;
    m.index[name] = (int)m.params.size();
    m.params.push_back(p);
  }

  // linear / 1x1 conv into rows [row_off, row_off+n) of a (possibly shared) [ntot][k] matrix
  void lin_rows(const std::string& n, LinW& w, int rows, int row_off, bool conv1x1, bool bias) {
    if (conv1x1) reg(n + ".weight", {rows, w.k, 1, 1}, PK_ROWS, w.w, rows, w.k, row_off);
    else reg(n + ".weight", {rows, w.k}, PK_ROWS, w.w, rows, w.k, row_off);
    if (bias) reg(n + ".bias", {rows}, PK_VEC_OFF, w.b, rows, row_off);
  }
  LinW lin_alloc(int ntot, int k, bool bias) {
    LinW w; w.n = ntot; w.k = k; w.w = take((size_t)ntot * k * 2); w.b = bias ? take(ntot * 4) : NPOS; w.has_bias = bias;
    return w;
  }
  NormW norm(const std::string& n, int c) {
    NormW w; w.c = c; w.g = take(c * 4); w.b = take(c * 4);
    reg(n + ".weight", {c}, PK_VEC, w.g); reg(n + ".bias", {c}, PK_VEC, w.b);
    return w;
  }
  ConvW conv3(const std::string& n, int co, int ci) {
    ConvW w; w.cin = ci; w.cout = co; w.w = take((size_t)co * 9 * ci * 2); w.b = take(co * 4);
    reg(n + ".weight", {co, ci, 3, 3}, PK_CONV3, w.w, co, ci); reg(n + ".bias", {co}, PK_VEC, w.b);
    return w;
  }

  LinW lin(const std::string& n, int co, int ci, bool bias = true, bool conv1x1 = false) {
    LinW w = lin_alloc(co, ci, bias);
    lin_rows(n, w, co, 0, conv1x1, bias);
    return w;
  }
};

struct Arena {   // plan-time first-fit allocator with coalescing free list
  struct Blk { size_t off, size; };
  std::vector<Blk> free_;
  size_t top = 0, peak = 0;
  size_t alloc(size_t bytes) {
    bytes = align_up(std::max<size_t>(bytes, 256), 256);
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].size >= bytes) {
        size_t o = free_[i].off;
        free_[i].off += bytes; free_[i].size -= bytes;
        if (!free_[i].size) free_.erase(free_.begin() + i);
        return o;
      }
    // extend: if the last free block touches the top, grow it
    if (!free_.empty() && free_.back().off + free_.back().size == top) {
      size_t o = free_.back().off;
      top = o + bytes; free_.pop_back();
      peak = std::max(peak, top);
      return o;
    }
    size_t o = top; top += bytes; peak = std::max(peak, top);
    return o;
  }
  void release(size_t off, size_t bytes) {
    bytes = align_up(std::max<size_t>(bytes, 256), 256);
    Blk b{off, bytes};
    auto it = std::lower_bound(free_.begin(), free_.end(), b, [](const Blk& x, const Blk& y) { return x.off < y.off; });
    it = free_.insert(it, b);
    if (it + 1 != free_.end() && it->off + it->size == (it + 1)->off) { it->size += (it + 1)->size; free_.erase(it + 1); }
    if (it != free_.begin() && (it - 1)->off + (it - 1)->size == it->off) { (it - 1)->size += it->size; free_.erase(it); }
  }
};

// fp16 activation view (+ optional fp32 master of the same logical tensor, contiguous ld = C)
struct Act {
  float hscale = 1.0f;        // the fp16 image holds value * hscale (a power of two; 1 everywhere but in the VAE encoder)
  Ref h{}; int ld = 0;        // fp16 [rows][C] with leading dimension ld
  int lo = 0;                 // > 0 ("precise" plans): the image is a split pair, hi = fp16(v) at h, lo = fp16(v - hi) `lo` elements further
  Ref f{}; bool has_f = false;
  int C = 0, H = 0, W = 0;
  size_t h_alloc = NPOS, h_bytes = 0;   // workspace block owned by h (NPOS: lives in a concat buffer / elsewhere)
  size_t f_alloc = NPOS, f_bytes = 0;
  // GroupNorm partial sums of the fp16 image, written by the epilogue of the conv that produced it (GemmParams::gn_partial; PlanBuilder::gn_epi):
  // [rows / gp_rows][C][2] floats.  groupnorm() then skips its statistics pass.
  size_t gp_alloc = NPOS, gp_bytes = 0; int gp_rows = 0;
};

struct PlanBuilder {
  const Model& m;
  Plan& P;
  Arena ar;
  bool dry;
  int Bn, n_ctx;
  bool stop = false;
  int remaining = 0;
  const PlanOpts& opt;
  // fp16 images of new activations are stored scaled by this power of two (range control, GemmParams::out16_scale): consumers
  // undo it exactly (GEMM: acc_scale; GroupNorm is scale invariant once eps is scaled by hscale^2).  Residual adds use the
  // fp32 master, which always holds the true values.
  float act_scale = 1.0f;
  // "precise" plans (gdf_plan_opts.reserved[1], UNet only): every 16-bit activation that feeds an MFMA contraction or a GroupNorm is
  // stored as a split pair (hi, lo) = 22 mantissa bits, and the contraction runs over [hi | lo] against the weights read twice
  // (GemmParams::k_w): the fp16-OPERAND rounding of DESIGN.md §4 disappears, at twice the MFMA work.  Attention internals (q, k,
  // v, P) stay fp16.  px = 2 in such a plan: width factor of every 16-bit image.
  //
  // Round 4: the split is PER OPERAND CLASS (gdf.h, gdf_plan_opts.reserved[1] = mask << 8; 1 = every class = the round-3 "precise" plan).
  // tools/operand_subsets.py measures each class's contribution to the hook error on the CPU oracle: the roundings on the MAIN PATH — the fp16
  // image of the residual stream as read by the 1x1 shortcuts, the down / upsampler convs, proj_out and the GroupNorms, and the GroupNorm output
  // in front of proj_in — carry most of it and are cheap to split (few, small GEMMs); the branch-internal operands (LayerNorm outputs, GEGLU
  // inner, resnet conv operands) are expensive and matter less.
  enum { SP_STREAM = 1,      // fp16 images of residual-stream tensors (incl. the skip-concat buffers): shortcut / proj_out operands, GroupNorm inputs
         SP_GNV = 2,         // Transformer2DModel.norm output  -> proj_in operand
         SP_LN_ATTN = 4,     // LayerNorm-1 / -2 outputs        -> to_q|k|v, cross to_q operands
         SP_ATTN_OUT = 8,    // self-attention outputs          -> attn1.to_out.0 operands
         SP_LN_FF = 16,      // LayerNorm-3 output              -> GEGLU projection operand
         SP_FF_INNER = 32,   // GEGLU inner tensor              -> ff.net.2 operand
         SP_RES = 64,        // resnet GroupNorm(+SiLU) outputs and the conv1 output -> conv1 / conv2 operands, norm2 input
         SP_OUT = 128,       // conv_norm_out output            -> conv_out operand
         SP_SAMPLER = 256,   // (with SP_STREAM) the DOWNsampler convs contract over hi + lo of the stream image (else over hi only)
         SP_ATTN2_OUT = 512, // cross-attention outputs         -> attn2.to_out.0 operands (SP_ATTN_OUT: the self-attention outputs)
         SP_UPSAMPLER = 1024,// (with SP_STREAM) the UPsampler convs contract over hi + lo (two of the largest convs of the step)
         SP_QKV = 2048,      // round 5: the SELF-attention q | k | v are stored as pairs and the flash kernel contracts over both halves
                             // (AttnParams::q_lo / kv_lo): the fp16 STORAGE rounding in front of the softmax; 2.5 x the attention MFMAs
         SP_XQKV = 4096,     // the same for the text CROSS-attention (its q and the grouped text K / V): 77 keys, < 1 % of the step — and by far the larger
                             // half of that rounding (benign weights 4.0e-4 of the worst hook against 0.7e-4 for the self-attention, heavy-tailed
                             // 8.0e-4 against 2.6e-4: DESIGN.md 3.9 h), so the selective and light presets carry it
         SP_ALL = 8191 };
  bool gn_epi = false;        // VAE op programs: 3x3 convs emit the GroupNorm partial sums of their output (Act::gp_*)
  int split = 0;              // mask of the classes above
  bool precise = false;       // split != 0
  bool spl(int cls) const { return (split & cls) != 0; }
  int pxc(int cls) const { return spl(cls) ? 2 : 1; }

  PlanBuilder(const Model& mm, Plan& pp, bool d, const PlanOpts& o) : m(mm), P(pp), dry(d), opt(o) {
    if (mm.kind == 0) split = o.reserved[1] == 1 ? SP_ALL : ((o.reserved[1] >> 8) & SP_ALL);      // the UNet op program only
    precise = split != 0;
  }
  size_t img_bytes(size_t nrows, int C, int cls) const { return nrows * (size_t)C * 2 * pxc(cls); }     // a contiguous 16-bit image (split: [hi | lo])

  Ref ws(size_t off) const { return Ref{BUF_WS, off}; }
  Ref wt(size_t off) const { return Ref{BUF_WT, off}; }

  void op(const char* name, double flops, std::function<hipError_t(const Bind&, hipStream_t)> fn,
          const char* kernel = nullptr) {
    if (dry || stop) return;
    Op o{name, flops, std::move(fn)};
    const std::string lab = kernel ? kernel : kernel_label(name);
    size_t li = 0;
    for (; li < P.labels.size(); ++li) if (P.labels[li] == lab) break;
    if (li == P.labels.size()) P.labels.push_back(lab);
    o.label = (int)li;
    P.ops.push_back(std::move(o));
  }

  size_t rows(const Act& a) const { return (size_t)Bn * a.H * a.W; }

  Act new_act(int C, int H, int W, bool master, int cls = SP_STREAM) {
    const int px = pxc(cls);
    Act a; a.C = C; a.H = H; a.W = W; a.ld = C * px; a.lo = spl(cls) ? C : 0; a.hscale = act_scale;
    a.h_bytes = (size_t)Bn * H * W * C * 2 * px;
    a.h_alloc = dry ? 0 : ar.alloc(a.h_bytes);
    a.h = ws(a.h_alloc);
    if (master && opt.stream_fp32) add_master(a);
    return a;
  }
  void add_master(Act& a) {
    a.f_bytes = (size_t)Bn * a.H * a.W * a.C * 4;
    a.f_alloc = dry ? 0 : ar.alloc(a.f_bytes);
    a.f = ws(a.f_alloc); a.has_f = true;
  }
  // activation whose fp16 image lives inside someone else's buffer (concat slice)
  Act view_act(Ref h, int ld, int C, int H, int W, bool master, int lo = 0) {
    Act a; a.C = C; a.H = H; a.W = W; a.ld = ld; a.h = h; a.lo = lo;
    if (master && opt.stream_fp32) add_master(a);
    return a;
  }
  void free_act(Act& a) {
    if (dry) return;
    if (a.h_alloc != NPOS) { ar.release(a.h_alloc, a.h_bytes); a.h_alloc = NPOS; }
    if (a.f_alloc != NPOS) { ar.release(a.f_alloc, a.f_bytes); a.f_alloc = NPOS; a.has_f = false; }
    if (a.gp_alloc != NPOS) { ar.release(a.gp_alloc, a.gp_bytes); a.gp_alloc = NPOS; a.gp_rows = 0; }
  }
  void free_master(Act& a) {
    if (dry) return;
    if (a.f_alloc != NPOS) { ar.release(a.f_alloc, a.f_bytes); a.f_alloc = NPOS; a.has_f = false; }
  }
  size_t tmp(size_t bytes) { return dry ? 0 : ar.alloc(bytes); }
  void untmp(size_t off, size_t bytes) { if (!dry) ar.release(off, bytes); }

  // ---- hooks ------------------------------------------------------------------------------------
  // returns hook slot (>= 0) if `id` is requested, else -1. Shape is logical (B, C, H, W) stored channels-last.
  int want(const std::string& id, int C, int H, int W) {
    if (dry) { P.dry_ids.push_back(id); return -1; }
    if (stop) return -1;
    if (!P.requested.count(id)) return -1;
    HookSlot hs; hs.id = id;
    hs.shape[0] = Bn; hs.shape[1] = C; hs.shape[2] = H; hs.shape[3] = W;
    hs.stride[0] = (int64_t)H * W * C; hs.stride[1] = 1; hs.stride[2] = (int64_t)W * C; hs.stride[3] = C;
    hs.bytes = (size_t)Bn * C * H * W * 2;
    P.hooks.push_back(hs);
    return (int)P.hooks.size() - 1;
  }
  int want_map(const std::string& id, int heads, int Sq, int Sk) {
    if (dry) { P.dry_ids.push_back(id); return -1; }
    if (stop || !P.requested.count(id)) return -1;
    HookSlot hs; hs.id = id;
    hs.shape[0] = Bn; hs.shape[1] = heads; hs.shape[2] = Sq; hs.shape[3] = Sk;
    hs.stride[0] = (int64_t)heads * Sq * Sk; hs.stride[1] = (int64_t)Sq * Sk; hs.stride[2] = Sk; hs.stride[3] = 1;
    hs.bytes = (size_t)Bn * heads * Sq * Sk * 2;
    P.hooks.push_back(hs);
    return (int)P.hooks.size() - 1;
  }
  void hook_done() {
    if (dry) return;
    if (--remaining == 0 && opt.early_exit) stop = true;
  }
  // coalesced hook store: fp16 copy of `rows x C` from (src, ld)
  // s_lo > 0: the source is a split pair (hook = fp16(hi + lo)); src_bf: element type of the source (-1 = the model's)
  // scale != 1: the source holds scale^-1-scaled values (range-scaled fp16 image): hook = fp16(scale * src)
  void hook_copy(int slot, Ref src, int ld, size_t nrows, int C, int s_lo = 0, int src_bf = -1, float scale = 1.0f) {
    if (slot < 0) return;
    P.hooks[slot].copied = true;
    const int bf = src_bf >= 0 ? src_bf : m.bf16, sat = (m.kind == 1);   // MMDiT hooks: bf16 or range-critical fp16 source -> saturating fp16
    op("hook_store", 0, [=](const Bind& b, hipStream_t s) {
      return launch_copy2d((const half_t*)b.p(src), nullptr, ld, (half_t*)b.hook(slot), C, (int)nrows, C, s, bf, sat, s_lo, scale);
    });
    hook_done();
  }
  void gather(const std::string& id, const Act& a) { hook_copy(want(id, a.C, a.H, a.W), a.h, a.ld, rows(a), a.C); }


  struct Epi {
    Ref bias{}; bool has_bias = false;
    Ref rowvec{}; bool has_rv = false; int rps = 1, ldrv = 0;
    Ref res32{}; bool has_r32 = false; Ref res16{}; bool has_r16 = false; int ldres = 0;
    Ref out16{}; bool has_o16 = false; int ldo16 = 0;
    Ref out32{}; bool has_o32 = false; int ldo32 = 0;
    int aux_slot = -1; int ldaux = 0;
    int o16_lo = 0;                                                       // out16 is a split (hi, lo) pair, lo at +o16_lo elements
    int geglu = 0; int bn = 128;
    int dit = 0, act = 0, rv_mul = 0, rv_seg_rows = 0, rv_rps2 = 0;     // MMDiT epilogue (kernels.h)
    int bf16 = 0;                                                         // bf16 operands / activations (set by gemm() from the model)
    int out_f16 = 0;                                                      // bf16 kernel storing out16 as saturating fp16 (GemmParams::out_f16)
    float acc_scale = 0.f, out16_scale = 0.f;                             // fp16 range control (kernels.h), 0 = 1
    int pad0 = 0;                                                         // conv3: 1 = pad right / bottom only
    int rv_tok = 0;                                                       // row vector indexed by token (row % rps)
    // fused RMSNorm(q), RMSNorm(k) + RoPE (GemmParams::qkn_*)
    int qkn_nq = 0; Ref qkn_wq{}, qkn_wk{}, rope_cos{}, rope_sin{}; float qkn_eps = 1e-6f;
    int qkn_pos0 = 0, qkn_rps = 1, qkn_seg_rows = 0, qkn_pos1 = 0, qkn_rps2 = 1;
  };
  void residual_from(Epi& e, const Act& x) {
    if (x.has_f) { e.res32 = x.f; e.has_r32 = true; e.ldres = x.C; }
    else {
      e.res16 = x.h; e.has_r16 = true; e.ldres = x.ld;             // (a scaled fp16 image always comes with an fp32 master)
      if (x.lo && !dry) { set_error("precise plan: residual source without an fp32 master"); bad = true; }
    }
  }
  bool bad = false;           // the op program could not be built (reported by plan_build)
  // A operand = the fp16 image of activation x: undo its storage scale on the accumulators
  static void reads_image(Epi& e, const Act& x) { if (x.hscale != 1.0f) e.acc_scale = 1.0f / x.hscale; }
  // need_shadow = false: the fp16 image of a stream tensor is not stored when its only consumers read the fp32
  // master (LayerNorm + the next residual add): saves one 2-byte/element write per residual GEMM
  void out_to(Epi& e, const Act& y, bool need_shadow = true) {
    if (need_shadow || !y.has_f) { e.out16 = y.h; e.has_o16 = true; e.ldo16 = y.ld; e.o16_lo = y.lo; if (y.hscale != 1.0f) e.out16_scale = y.hscale; }
    if (y.has_f) { e.out32 = y.f; e.has_o32 = true; e.ldo32 = y.C; }
  }
  static void fill_epi(GemmParams& g, const Epi& e, const Bind& b) {
    g.bias = e.has_bias ? (const float*)b.p(e.bias) : nullptr;
    g.rowvec = e.has_rv ? (const float*)b.p(e.rowvec) : nullptr; g.rows_per_sample = e.rps; g.ldrv = e.ldrv;
    g.res32 = e.has_r32 ? (const float*)b.p(e.res32) : nullptr;
    g.res16 = e.has_r16 ? (const half_t*)b.p(e.res16) : nullptr; g.ldres = e.ldres;
    g.out16 = e.has_o16 ? (half_t*)b.p(e.out16) : nullptr; g.ldo16 = e.ldo16;
    g.out32 = e.has_o32 ? (float*)b.p(e.out32) : nullptr; g.ldo32 = e.ldo32;
    g.aux16 = e.aux_slot >= 0 ? (half_t*)b.hook(e.aux_slot) : nullptr; g.ldaux = e.ldaux;
    g.out_f16 = e.out_f16;
    g.geglu = e.geglu; g.bn = e.bn; g.bf16 = e.bf16; g.acc_scale = e.acc_scale; g.out16_scale = e.out16_scale; g.o16_lo = e.o16_lo;
    g.dit = e.dit; g.act = e.act; g.rv_mul = e.rv_mul; g.rv_seg_rows = e.rv_seg_rows; g.rv_rps2 = e.rv_rps2; g.rv_tok = e.rv_tok;
    g.qkn_nq = e.qkn_nq;
    if (e.qkn_nq) {
      g.qkn_wq = (const float*)b.p(e.qkn_wq); g.qkn_wk = (const float*)b.p(e.qkn_wk); g.qkn_eps = e.qkn_eps;
      g.rope_cos = (const float*)b.p(e.rope_cos); g.rope_sin = (const float*)b.p(e.rope_sin);
      g.qkn_pos0 = e.qkn_pos0; g.qkn_rps = e.qkn_rps; g.qkn_seg_rows = e.qkn_seg_rows; g.qkn_pos1 = e.qkn_pos1; g.qkn_rps2 = e.qkn_rps2;
    }
  }

  // dense GEMM: A (fp16 [M][K], lda) x W[N][K].  a_lo > 0: A is a split pair (lo columns a_lo elements after the hi columns):
  // the contraction runs over [hi | lo] (2K) against W read twice (GemmParams::k_w)
  // mx != null ('fp8-mx' MMDiT plans): the A operand is the fp8 (e4m3) matrix mx->a8 [M][lda8 bytes] with per-row scales mx->ascale, the
  // weights their fp8 copy + per-output-channel scales inside the model arena (Model::f8_off / sc_off); A / lda are then unused
  struct MxA { Ref a8; int lda8 = 0; Ref ascale; };
  static bool mx_ok(size_t M, int N, int K) { return M >= 2048 && (N % 256) == 0 && (K % 128) == 0; }
  void gemm(const char* name, Ref A, int lda, size_t M, const LinW& w, int N, int Kw, size_t w_off_bytes, const Epi& e0, int a_lo = 0,
            const MxA* mxp = nullptr) {
    Epi e = e0;
    e.bf16 = (e.dit && m.bf16) ? 1 : 0;
    const Ref W = wt(w.w + w_off_bytes);
    const int K = a_lo > 0 ? 2 * Kw : Kw;
    const bool use_mx = mxp && m.fp8 && e.dit && a_lo == 0 && mx_ok(M, N, Kw);
    const MxA mxa = use_mx ? *mxp : MxA{};
    const Ref W8 = wt(m.f8_off + (w.w + w_off_bytes) / 2), WS = wt(m.sc_off + w.w / 16 + (w_off_bytes / ((size_t)Kw * 2)) * 4);
    GemmParams gk{}; gk.M = (int)M; gk.N = N; gk.K = K; gk.mode = A_DENSE; gk.geglu = e.geglu; gk.bn = e.bn; gk.dit = e.dit; gk.bf16 = e.bf16;
    gk.mx = use_mx ? 1 : 0;
    gk.k_w = a_lo > 0 ? Kw : 0; gk.o16_lo = e.has_o16 ? e.o16_lo : 0;      // split operands: their own kernel instantiations / tile set
    const int cus = opt.reserved[2]; gk.cus = cus;                           // CU partition of the launch stream (0 = whole chip)
    gk.res32 = e.has_r32 ? (const float*)1 : nullptr;      // tile selection looks at the epilogue form (never dereferenced)
    // few output tiles, long K (small batches: ff_out at 1024-2048 rows): deterministic split-K (see conv3)
    const int splitk = gemm_splitk_factor(gk);
    const size_t ws_b = splitk > 1 ? (size_t)splitk * M * N * 4 : 0;
    const size_t wsk = splitk > 1 ? tmp(ws_b) : 0;
    gk.splitk = splitk;
    op(name, 2.0 * (double)M * N * Kw, [=](const Bind& b, hipStream_t s) {      // algorithmic FLOPs (the split doubles the MFMA work, not these)
      GemmParams g{};
      g.A = (const half_t*)b.p(A); g.lda = lda; g.a_bytes = (uint32_t)(((size_t)M - 1) * lda * 2 + (size_t)(a_lo + Kw) * 2);
      g.M = (int)M; g.N = N; g.K = K; g.mode = A_DENSE; g.cus = cus;
      if (a_lo > 0) { g.k_w = Kw; g.a_lo_bytes = (uint32_t)a_lo * 2u; }
      g.Wt = (const half_t*)b.p(W); g.w_bytes = (uint32_t)((size_t)N * Kw * 2);
      fill_epi(g, e, b);
      if (use_mx) {                                        // fp8 rows in 2-byte units (kernels.h GemmParams::mx)
        g.A = (const half_t*)b.p(mxa.a8); g.lda = mxa.lda8 / 2; g.a_bytes = (uint32_t)(((size_t)M - 1) * mxa.lda8 + (size_t)Kw);
        g.K = Kw / 2; g.Wt = (const half_t*)b.p(W8); g.w_bytes = (uint32_t)((size_t)N * Kw);
        g.mx = 1; g.mx_rowscale = (const float*)b.p(mxa.ascale); g.mx_colscale = (const float*)b.p(WS);
      }
      return splitk > 1 ? launch_gemm_splitk(g, splitk, (float*)b.ws(wsk), s) : launch_gemm(g, s);
    }, gemm_kernel_name(gk));
    if (splitk > 1) untmp(wsk, ws_b);
  }

  Ref temb_all{};     // [B][temb_total] f32: every resnet's time_emb_proj(silu(emb)) (UNet only)

  // ---- primitive emitters -----------------------------------------------------------------------
  // GroupNorm (+SiLU) of x -> contiguous fp16 tensor (workspace offset returned)
  // out_cls: operand class of the OUTPUT (split pair when that class is split)
  size_t groupnorm(const Act& x, const NormW& w, float eps_true, bool silu, int out_cls = 0) {
    const float eps = eps_true * x.hscale * x.hscale;      // GN(s x, s^2 eps) == GN(x, eps): the scaled fp16 image normalises identically
    const size_t n = rows(x);
    const size_t y = tmp(img_bytes(n, x.C, out_cls));
    // a split input image: read the fp32 master where there is one, else the pair; the output is a split image when its class is
    const bool from_f = x.lo > 0 && x.has_f;
    const Ref xh = x.h, xf = x.f; const int ld = from_f ? x.C : x.ld, C = x.C, HW = x.H * x.W, Bq = Bn;
    const int x_lo = from_f ? 0 : x.lo, ldy = C * pxc(out_cls), y_lo = spl(out_cls) ? C : 0;
    const Ref g = wt(w.g), bt = wt(w.b);
    if (gn_fused_slab(Bn, x.H * x.W, x.C, 32)) {            // small feature map: statistics + apply in one launch
      op(silu ? "gn_fused_silu" : "gn_fused", 0, [=](const Bind& b, hipStream_t s) {
        return launch_gn_fused(from_f ? nullptr : (const half_t*)b.p(xh), from_f ? (const float*)b.p(xf) : nullptr, ld, Bq, HW, C, 32, eps,
                               (const float*)b.p(g), (const float*)b.p(bt), silu ? 1 : 0, (half_t*)b.ws(y), s, x_lo, ldy, y_lo);
      });
      return y;
    }
    if (x.gp_alloc != NPOS && x.gp_rows > 0 && !from_f && x_lo == 0) {     // the producing conv left the per-slab channel sums: finalize only
      const size_t ab_b = (size_t)Bn * x.C * 8, ab = tmp(ab_b);
      const size_t gp = x.gp_alloc; const int nslab = HW / x.gp_rows;
      // round 6: finalize + apply in ONE launch (after the fold pass where the producer left many short slabs) — built,
      // measured on the same box, REJECTED: SDXL B = 16 144.3 vs 145.0 img/s, VAE encode 143.4 vs 148.2, SD1.5 B = 32 equal
      // (profiles/r06_ab_gn_apply_after_fold.txt: every row block repeats the slab combine, and the apply pass loses its 64-row blocks' parallelism).
      // OFF by default; GDF_GN_FINALIZE_APPLY=1 selects it (diagnostics)
      static const bool fa_on = [] { const char* e = getenv("GDF_GN_FINALIZE_APPLY"); return e && atoi(e) != 0; }();
      const bool need_fold = gn_fold_floats(Bn, nslab, C) != 0;
      if (fa_on && gn_finalize_apply_slab(C, 32) && (!need_fold || gn_fold_ok(C))) {
        untmp(ab, ab_b);
        const size_t fb = need_fold ? gn_fold_floats(Bn, nslab, C) * 4 : 0, fo = fb ? tmp(fb) : 0;
        const int ns2 = need_fold ? gn_fold_out_slabs(nslab) : nslab;
        if (need_fold) op("gn_fold", 0, [=](const Bind& b, hipStream_t s) { return launch_gn_fold((const float*)b.ws(gp), nslab, Bq, C, (float*)b.ws(fo), s); });
        op(silu ? "gn_finalize_apply_silu" : "gn_finalize_apply", 0, [=](const Bind& b, hipStream_t s) {
          return launch_gn_finalize_apply(need_fold ? (const float*)b.ws(fo) : (const float*)b.ws(gp), ns2, (const half_t*)b.p(xh), ld, Bq, HW, C, 32, eps,
                                          (const float*)b.p(g), (const float*)b.p(bt), silu ? 1 : 0, (half_t*)b.ws(y), s, ldy, y_lo);
        });
        if (fb) untmp(fo, fb);
        return y;
      }
      const size_t fold_b = gn_fold_floats(Bn, nslab, C) * 4, fold = fold_b ? tmp(fold_b) : 0;
      op("gn_finalize", 0, [=](const Bind& b, hipStream_t s) {
        return launch_gn_finalize((const float*)b.ws(gp), nslab, Bq, HW, C, 32, eps, (const float*)b.p(g), (const float*)b.p(bt), (float*)b.ws(ab),
                                  fold_b ? (float*)b.ws(fold) : nullptr, s);
      });
      op(silu ? "gn_apply_silu" : "gn_apply", 0, [=](const Bind& b, hipStream_t s) {
        return launch_gn_apply((const half_t*)b.p(xh), nullptr, ld, Bq, HW, C, (const float*)b.ws(ab), silu ? 1 : 0, (half_t*)b.ws(y), s, 0, ldy, y_lo);
      });
      untmp(ab, ab_b);
      if (fold_b) untmp(fold, fold_b);
      return y;
    }
    const size_t part_b = gn_partial_floats(Bn, x.H * x.W, x.C) * 4, ab_b = (size_t)Bn * x.C * 8;
    const size_t part = tmp(part_b), ab = tmp(ab_b);
    op("gn_stats", 0, [=](const Bind& b, hipStream_t s) {
      return launch_gn_stats(from_f ? nullptr : (const half_t*)b.p(xh), from_f ? (const float*)b.p(xf) : nullptr, ld, Bq, HW, C, 32, eps,
                             (const float*)b.p(g), (const float*)b.p(bt), (float*)b.ws(part), (float*)b.ws(ab), s, x_lo);
    });
    op(silu ? "gn_apply_silu" : "gn_apply", 0, [=](const Bind& b, hipStream_t s) {
      return launch_gn_apply(from_f ? nullptr : (const half_t*)b.p(xh), from_f ? (const float*)b.p(xf) : nullptr, ld, Bq, HW, C,
                             (const float*)b.ws(ab), silu ? 1 : 0, (half_t*)b.ws(y), s, x_lo, ldy, y_lo);
    });
    untmp(part, part_b); untmp(ab, ab_b);
    return y;
  }

  // 3x3 conv as implicit GEMM over NHWC `src` (Bn, H, W, ld>=Cin)
  // a_lo > 0: the source pixels are split pairs (lo channels a_lo elements after the hi channels), see gemm()
  // stats != null (and gn_epi): the epilogue also writes the GroupNorm partial sums of the fp16 image it stores into a buffer attached to *stats
  // (the activation being produced), when the tile the launcher picks supports it (gemm_gn_slab_rows)
  void conv3(const char* name, Ref src, int ld, int Cin, int H, int W, int stride, bool ups, const ConvW& w, const Epi& e0, int a_lo = 0,
             Act* stats = nullptr) {
    Epi e = e0;
    const int kx = a_lo > 0 ? 2 : 1;
    const int IH = ups ? 2 * H : H, IW = ups ? 2 * W : W;
    const int OH = (IH + 2 - 3) / stride + 1, OW = (IW + 2 - 3) / stride + 1;
    const size_t M = (size_t)Bn * OH * OW;
    const Ref Wr = wt(w.w);
    const int N = w.cout, Bq = Bn;
    GemmParams gk{}; gk.M = (int)M; gk.N = N; gk.K = 9 * Cin * kx; gk.mode = A_CONV3; gk.bn = e.bn;
    gk.k_w = a_lo > 0 ? 9 * Cin : 0; gk.o16_lo = e.has_o16 ? e.o16_lo : 0;
    const int cus = opt.reserved[2]; gk.cus = cus;
    // few output tiles, long K (SD1.5's 8x8 level: 160 tiles of 128x128 walking 180-360 K-tiles each): deterministic split-K
    const int splitk = gemm_splitk_factor(gk);
    const size_t ws_b = splitk > 1 ? (size_t)splitk * M * N * 4 : 0;
    const size_t wsk = splitk > 1 ? tmp(ws_b) : 0;
    gk.splitk = splitk;
    size_t gp = NPOS;
    if (stats && gn_epi && !stop && e.has_o16 && a_lo == 0 && e.o16_lo == 0) {
      const int sr = gemm_gn_slab_rows(gk);
      if (sr > 0 && (OH * OW) % sr == 0) {
        stats->gp_rows = sr; stats->gp_bytes = M / sr * (size_t)N * 8;
        stats->gp_alloc = gp = dry ? 0 : ar.alloc(stats->gp_bytes);
        gk.gn_partial = (float*)1;                          // kernel label only (never dereferenced)
      }
    }
    op(name, 2.0 * (double)M * N * 9 * Cin, [=](const Bind& b, hipStream_t s) {
      GemmParams g{};
      if (gp != NPOS) g.gn_partial = (float*)b.ws(gp);
      g.A = (const half_t*)b.p(src); g.lda = ld;
      g.a_bytes = (uint32_t)(((size_t)Bq * H * W - 1) * ld * 2 + (size_t)(a_lo + Cin) * 2);
      g.M = (int)M; g.N = N; g.K = 9 * Cin * kx; g.mode = A_CONV3; g.H = H; g.W = W; g.OH = OH; g.OW = OW; g.cus = cus;
      if (a_lo > 0) { g.k_w = 9 * Cin; g.a_lo_bytes = (uint32_t)a_lo * 2u; }
      g.stride = stride; g.ups = ups ? 1 : 0; g.Cin = Cin; g.pad0 = e.pad0;
      g.Wt = (const half_t*)b.p(Wr); g.w_bytes = (uint32_t)((size_t)N * 9 * Cin * 2);
      fill_epi(g, e, b);
      return splitk > 1 ? launch_gemm_splitk(g, splitk, (float*)b.ws(wsk), s) : launch_gemm(g, s);
    }, gemm_kernel_name(gk));
    if (splitk > 1) untmp(wsk, ws_b);
  }

  // ---- ResnetBlock2D ------------------------------------------------------------------------------
  // x -> y (y.h destination prepared by the caller)
  void resnet(const std::string& id, const ResnetW& w, const Act& x, Act& y) {
    if (stop) return;
    const size_t n = rows(x);
    const int HW = x.H * x.W;
    const size_t n1 = groupnorm(x, w.n1, w.eps, true, SP_RES);
    const int slo = spl(SP_RES) ? 1 : 0, px = pxc(SP_RES);   // split GroupNorm outputs: [rows][2C], lo at +C
    Act h1 = new_act(w.cout, x.H, x.W, false, SP_RES);
    {
      Epi e; e.bias = wt(w.c1.b); e.has_bias = true;
      if (w.has_temb) {                                                                    // resnet.py:343-350
        e.rowvec = Ref{temb_all.buf, temb_all.off + (size_t)w.temb_off * 4}; e.has_rv = true; e.rps = HW; e.ldrv = m.temb_total;
      }
      out_to(e, h1);
      conv3("res_conv1", ws(n1), x.C * px, x.C, x.H, x.W, 1, false, w.c1, e, slo * x.C, &h1);
    }
    untmp(n1, img_bytes(n, x.C, SP_RES));
    const size_t n2 = groupnorm(h1, w.n2, w.eps, true, SP_RES);
    free_act(h1);
    // shortcut: 1x1 conv of x into an fp32 residual buffer
    size_t sc = NPOS; const size_t sc_b = n * w.cout * 4;
    if (w.has_sc) {
      sc = tmp(sc_b);
      Epi e; e.bias = wt(w.sc.b); e.has_bias = true; e.out32 = ws(sc); e.has_o32 = true; e.ldo32 = w.cout;
      reads_image(e, x);
      gemm("res_shortcut", x.h, x.ld, n, w.sc, w.cout, x.C, 0, e, x.lo);
    }
    {
      Epi e; e.bias = wt(w.c2.b); e.has_bias = true;
      e.aux_slot = want(id + "-res-increment", w.cout, x.H, x.W); e.ldaux = w.cout;      // resnet.py:371-372
      if (w.has_sc) { e.res32 = ws(sc); e.has_r32 = true; e.ldres = w.cout; }
      else residual_from(e, x);
      out_to(e, y);
      // (statistics only into an activation that OWNS its buffer and is therefore released through free_act: a concat-slice view never is)
      conv3("res_conv2", ws(n2), w.cout * px, w.cout, x.H, x.W, 1, false, w.c2, e, slo * w.cout, (m.kind != 0 || y.h_alloc != NPOS) ? &y : nullptr);
      if (e.aux_slot >= 0) hook_done();
    }
    untmp(n2, img_bytes(n, w.cout, SP_RES));
    if (w.has_sc) untmp(sc, sc_b);
    gather(id + "-res-out", y);                                                           // resnet.py:376-377
  }

};

}  // namespace gdf
