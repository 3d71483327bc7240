// extern "C" kernel-level entry points (include/gdf_ops.h): one launch each, used by tests and micro-benchmarks.
#include "../../include/gdf_ops.h"
#include "model.h"

using namespace gdf;

// the GEMM / conv kernels address both operands with 32-bit buffer offsets whose top bit marks "out of range"
static bool span_ok(size_t a_bytes, size_t w_bytes, const char* what) {
  if (a_bytes < (1ull << 31) && w_bytes < (1ull << 31)) return true;
  set_error(std::string(what) + ": operand larger than 2 GiB (32-bit buffer offsets); split the rows");
  return false;
}

static int fin(hipError_t e, const char* what) {
  if (e == hipSuccess) return GDF_OK;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return GDF_ERR_HIP;
}

extern "C" {

int gdf_op_gemm(const void* A, int lda, const void* W, const float* bias, const float* res32, const void* res16,
                int ldres, void* out16, int ldo16, float* out32, int ldo32, int M, int N, int K, int flags,
                void* stream) {
  GemmParams g{};
  if (!span_ok(((size_t)M - 1) * lda * 2 + (size_t)K * 2, (size_t)N * K * 2, "gemm")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)A; g.lda = lda; g.a_bytes = (uint32_t)(((size_t)M - 1) * lda * 2 + (size_t)K * 2);
  g.M = M; g.N = N; g.K = K; g.mode = A_DENSE;
  g.Wt = (const half_t*)W; g.w_bytes = (uint32_t)((size_t)N * K * 2);
  g.bias = bias; g.res32 = res32; g.res16 = (const half_t*)res16; g.ldres = ldres;
  g.out16 = (half_t*)out16; g.ldo16 = ldo16; g.out32 = out32; g.ldo32 = ldo32;
  g.geglu = (flags & 1) ? 16 : 0; g.bn = (flags & 2) ? 16 : 128; g.variant = (flags >> 8) & 0xfff; g.no_early_mma = (flags >> 20) & 1; g.no_superblock = (flags >> 21) & 1; g.rows_per_sample = 1;
  return fin(launch_gemm(g, (hipStream_t)stream), "gemm");
}

int gdf_op_conv3x3(const void* x, int ld, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                   const float* rowvec, int stride, int ups, const float* res32, void* aux16, void* out16,
                   float* out32, int narrow, void* stream) {
  const int IH = ups ? 2 * H : H, IW = ups ? 2 * W : W;
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  GemmParams g{};
  if (!span_ok(((size_t)B * H * W - 1) * ld * 2 + (size_t)Cin * 2, (size_t)Cout * 9 * Cin * 2, "conv3x3")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)x; g.lda = ld; g.a_bytes = (uint32_t)(((size_t)B * H * W - 1) * ld * 2 + (size_t)Cin * 2);
  g.M = B * OH * OW; g.N = Cout; g.K = 9 * Cin; g.mode = A_CONV3; g.H = H; g.W = W; g.OH = OH; g.OW = OW;
  g.stride = stride; g.ups = ups; g.Cin = Cin;
  g.Wt = (const half_t*)Wt; g.w_bytes = (uint32_t)((size_t)Cout * 9 * Cin * 2);
  g.bias = bias; g.rowvec = rowvec; g.rows_per_sample = OH * OW; g.ldrv = Cout;
  g.res32 = res32; g.ldres = Cout;
  g.aux16 = (half_t*)aux16; g.ldaux = Cout;
  g.out16 = (half_t*)out16; g.ldo16 = Cout; g.out32 = out32; g.ldo32 = Cout;
  g.bn = (narrow & 1) ? 16 : 128; g.variant = (narrow >> 8) & 0xfff; g.no_early_mma = (narrow >> 20) & 1;
  return fin(launch_gemm(g, (hipStream_t)stream), "conv3x3");
}

// ---- split-operand forms of the "precise" plans (kernels.h GemmParams::k_w / a_lo_bytes / o16_lo) ----
int gdf_op_gemm_split(const void* A, int lda, int a_lo, const void* W, const float* bias, const float* res32, int ldres, void* out16,
                      int ldo16, int o16_lo, float* out32, int ldo32, int M, int N, int Kw, int flags, void* stream) {
  GemmParams g{};
  const size_t a_bytes = ((size_t)M - 1) * lda * 2 + (size_t)(a_lo + Kw) * 2;
  if (!span_ok(a_bytes, (size_t)N * Kw * 2, "gemm_split")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)A; g.lda = lda; g.a_bytes = (uint32_t)a_bytes;
  g.M = M; g.N = N; g.K = a_lo > 0 ? 2 * Kw : Kw; g.mode = A_DENSE;
  if (a_lo > 0) { g.k_w = Kw; g.a_lo_bytes = (uint32_t)a_lo * 2u; }
  g.Wt = (const half_t*)W; g.w_bytes = (uint32_t)((size_t)N * Kw * 2);
  g.bias = bias; g.res32 = res32; g.ldres = ldres;
  g.out16 = (half_t*)out16; g.ldo16 = ldo16; g.o16_lo = o16_lo; g.out32 = out32; g.ldo32 = ldo32;
  g.geglu = (flags & 1) ? 16 : 0; g.bn = 128; g.rows_per_sample = 1;
  return fin(launch_gemm(g, (hipStream_t)stream), "gemm_split");
}

int gdf_op_conv3x3_split(const void* x, int ld, int a_lo, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                         int stride, int ups, const float* res32, void* out16, int ldo16, int o16_lo, float* out32, void* stream) {
  const int IH = ups ? 2 * H : H, IW = ups ? 2 * W : W;
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  GemmParams g{};
  const size_t a_bytes = ((size_t)B * H * W - 1) * ld * 2 + (size_t)(a_lo + Cin) * 2;
  if (!span_ok(a_bytes, (size_t)Cout * 9 * Cin * 2, "conv3x3_split")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)x; g.lda = ld; g.a_bytes = (uint32_t)a_bytes;
  g.M = B * OH * OW; g.N = Cout; g.K = 9 * Cin * (a_lo > 0 ? 2 : 1); g.mode = A_CONV3; g.H = H; g.W = W; g.OH = OH; g.OW = OW;
  if (a_lo > 0) { g.k_w = 9 * Cin; g.a_lo_bytes = (uint32_t)a_lo * 2u; }
  g.stride = stride; g.ups = ups; g.Cin = Cin;
  g.Wt = (const half_t*)Wt; g.w_bytes = (uint32_t)((size_t)Cout * 9 * Cin * 2);
  g.bias = bias; g.rows_per_sample = OH * OW; g.res32 = res32; g.ldres = Cout;
  g.out16 = (half_t*)out16; g.ldo16 = ldo16; g.o16_lo = o16_lo; g.out32 = out32; g.ldo32 = Cout; g.bn = 128;
  return fin(launch_gemm(g, (hipStream_t)stream), "conv3x3_split");
}

int gdf_op_layernorm_split(const float* x32, int ld, int R, int C, float eps, const float* gamma, const float* beta, void* y, int ldy,
                           int y_lo, void* stream) {
  return fin(launch_layernorm(nullptr, x32, ld, R, C, eps, gamma, beta, (half_t*)y, (hipStream_t)stream, ldy, y_lo), "layernorm_split");
}

int gdf_op_groupnorm_split(const void* x16, int x_lo, const float* x32, int ld, int B, int HW, int C, int G, float eps, const float* gamma,
                           const float* beta, int silu, void* y, int ldy, int y_lo, void* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (gn_fused_slab(B, HW, C, G))
    return fin(launch_gn_fused((const half_t*)x16, x32, ld, B, HW, C, G, eps, gamma, beta, silu, (half_t*)y, s, x_lo, ldy, y_lo), "gn_fused_split");
  float* partial = (float*)scratch;
  float* ab = partial + (gn_partial_floats(B, HW, C) + 63) / 64 * 64;
  hipError_t e = launch_gn_stats((const half_t*)x16, x32, ld, B, HW, C, G, eps, gamma, beta, partial, ab, s, x_lo);
  if (e != hipSuccess) return fin(e, "gn_stats_split");
  return fin(launch_gn_apply((const half_t*)x16, x32, ld, B, HW, C, ab, silu, (half_t*)y, s, x_lo, ldy, y_lo), "gn_apply_split");
}

int gdf_op_attention_split(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int o_lo, int B,
                           int heads, int Sq, int Sk, int D, void* map, void* stream) {
  AttnParams a{};
  a.q = (const half_t*)q; a.ldq = ldq; a.k = (const half_t*)k; a.ldk = ldk; a.v = (const half_t*)v; a.ldv = ldv;
  a.o = (half_t*)o; a.ldo = ldo; a.o_lo = o_lo; a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D; a.kv_bstride = Sk;
  a.scale = 1.0f / sqrtf((float)D); a.map = (half_t*)map;
  return fin(launch_attention(a, (hipStream_t)stream), "attention_split");
}

int gdf_op_attention_pair(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int qkv_lo, void* o, int ldo, int o_lo, int B,
                          int heads, int Sq, int Sk, int D, void* stream) {
  if (qkv_lo <= 0 || (qkv_lo & 7)) return fin(hipErrorInvalidValue, "attention_pair");
  AttnParams a{};
  a.q = (const half_t*)q; a.ldq = ldq; a.k = (const half_t*)k; a.ldk = ldk; a.v = (const half_t*)v; a.ldv = ldv; a.q_lo = a.kv_lo = qkv_lo;
  a.o = (half_t*)o; a.ldo = ldo; a.o_lo = o_lo; a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D; a.kv_bstride = Sk;
  a.scale = 1.0f / sqrtf((float)D);
  return fin(launch_attention(a, (hipStream_t)stream), "attention_pair");
}

int gdf_op_conv3x3_splitk(const void* x, int ld, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                          const float* rowvec, int stride, int ups, const float* res32, void* aux16, void* out16,
                          float* out32, int splitk, float* ws, void* stream) {
  const int IH = ups ? 2 * H : H, IW = ups ? 2 * W : W;
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  GemmParams g{};
  if (!span_ok(((size_t)B * H * W - 1) * ld * 2 + (size_t)Cin * 2, (size_t)Cout * 9 * Cin * 2, "conv3x3_splitk")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)x; g.lda = ld; g.a_bytes = (uint32_t)(((size_t)B * H * W - 1) * ld * 2 + (size_t)Cin * 2);
  g.M = B * OH * OW; g.N = Cout; g.K = 9 * Cin; g.mode = A_CONV3; g.H = H; g.W = W; g.OH = OH; g.OW = OW;
  g.stride = stride; g.ups = ups; g.Cin = Cin;
  g.Wt = (const half_t*)Wt; g.w_bytes = (uint32_t)((size_t)Cout * 9 * Cin * 2);
  g.bias = bias; g.rowvec = rowvec; g.rows_per_sample = OH * OW; g.ldrv = Cout;
  g.res32 = res32; g.ldres = Cout;
  g.aux16 = (half_t*)aux16; g.ldaux = Cout;
  g.out16 = (half_t*)out16; g.ldo16 = Cout; g.out32 = out32; g.ldo32 = Cout; g.bn = 128;
  if (splitk == 0) splitk = gemm_splitk_factor(g);          // 0: the plan builder's own choice (returned through *ws[0]? no: see gdf_op_splitk_factor)
  return fin(launch_gemm_splitk(g, splitk, ws, (hipStream_t)stream), "conv3x3_splitk");
}

int gdf_op_splitk_factor(int M, int N, int K, int conv) {
  GemmParams g{}; g.M = M; g.N = N; g.K = K; g.mode = conv ? A_CONV3 : A_DENSE; g.bn = 128;
  return gemm_splitk_factor(g);
}

int gdf_op_conv_in(const void* x_nchw, int B, int Cin, int H, int W, const void* w_oihw, const float* bias, int Cout,
                   void* out16, void* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  half_t* lat8 = (half_t*)scratch;
  half_t* w = (half_t*)((char*)scratch + (size_t)B * H * W * 16);
  hipError_t e = launch_pack_latents((const half_t*)x_nchw, B, Cin, H, W, lat8, nullptr, s);
  if (e != hipSuccess) return fin(e, "pack_latents");
  e = launch_relayout_conv(w_oihw, 0, w, Cout, Cin, 9, 8, 16, s);
  if (e != hipSuccess) return fin(e, "relayout");
  GemmParams g{};
  const size_t M = (size_t)B * H * W;
  g.A = lat8; g.lda = 8; g.a_bytes = (uint32_t)(M * 16);
  g.M = (int)M; g.N = Cout; g.K = 128; g.mode = A_CONV_SMALLC; g.H = H; g.W = W; g.OH = H; g.OW = W; g.stride = 1; g.Cin = 8;
  g.Wt = w; g.w_bytes = (uint32_t)((size_t)Cout * 256);
  g.bias = bias; g.out16 = (half_t*)out16; g.ldo16 = Cout; g.bn = 128; g.rows_per_sample = 1;
  return fin(launch_gemm(g, s), "conv_in");
}

int gdf_op_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                     int B, int heads, int Sq, int Sk, int D, void* map, void* stream) {
  AttnParams a{};
  a.q = (const half_t*)q; a.ldq = ldq; a.k = (const half_t*)k; a.ldk = ldk; a.v = (const half_t*)v; a.ldv = ldv;
  a.o = (half_t*)o; a.ldo = ldo; a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D; a.kv_bstride = Sk;
  a.scale = 1.0f / sqrtf((float)D); a.map = (half_t*)map;
  return fin(launch_attention(a, (hipStream_t)stream), "attention");
}

size_t gdf_op_groupnorm_scratch_bytes(int B, int HW, int C) { return gn_partial_floats(B, HW, C) * 4 + (size_t)B * C * 8 + 256; }

int gdf_op_groupnorm(const void* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps,
                     const float* gamma, const float* beta, int silu, void* y, void* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (gn_fused_slab(B, HW, C, G))
    return fin(launch_gn_fused((const half_t*)x16, x32, ld, B, HW, C, G, eps, gamma, beta, silu, (half_t*)y, s), "gn_fused");
  float* partial = (float*)scratch;
  float* ab = partial + (gn_partial_floats(B, HW, C) + 63) / 64 * 64;
  hipError_t e = launch_gn_stats((const half_t*)x16, x32, ld, B, HW, C, G, eps, gamma, beta, partial, ab, s);
  if (e != hipSuccess) return fin(e, "gn_stats");
  return fin(launch_gn_apply((const half_t*)x16, x32, ld, B, HW, C, ab, silu, (half_t*)y, s), "gn_apply");
}

int gdf_op_layernorm(const void* x16, const float* x32, int ld, int R, int C, float eps, const float* gamma,
                     const float* beta, void* y, void* stream) {
  return fin(launch_layernorm((const half_t*)x16, x32, ld, R, C, eps, gamma, beta, (half_t*)y, (hipStream_t)stream), "layernorm");
}

int gdf_op_copy2d(const void* s16, const float* s32, int lds, void* dst, int ldd, int R, int C, void* stream) {
  return fin(launch_copy2d((const half_t*)s16, s32, lds, (half_t*)dst, ldd, R, C, (hipStream_t)stream), "copy2d");
}

int gdf_op_relayout_conv3(const void* w, void* dst, int O, int I, void* stream) {
  return fin(launch_relayout_conv(w, 0, (half_t*)dst, O, I, 9, I, 9, (hipStream_t)stream, 64), "relayout_conv3");
}
int gdf_op_relayout_geglu(const void* w, const float* bias, void* w_dst, float* bias_dst, int R, int K, int group, void* stream) {
  hipError_t e = launch_relayout_rows(w, 0, (half_t*)w_dst, R, K, 0, group, (hipStream_t)stream);
  if (e != hipSuccess) return fin(e, "relayout_geglu");
  if (bias) e = launch_relayout_vec(bias, 1, bias_dst, R, 0, group, (hipStream_t)stream);
  return fin(e, "relayout_geglu_bias");
}

int gdf_op_small_linear(const float* x, int ldx, int M, int K, const void* W, const float* bias, int N, int silu_in,
                        int accumulate, float* out, int ldo, void* stream) {
  return fin(launch_small_linear(x, ldx, M, K, (const half_t*)W, bias, N, silu_in, accumulate, out, ldo, (hipStream_t)stream), "small_linear");
}

int gdf_op_softmax_rows(void* x, int ld, int R, int n, float scale, void* stream) {
  return fin(launch_softmax_rows((half_t*)x, ld, R, n, scale, (hipStream_t)stream), "softmax_rows");
}

int gdf_op_sincos_pos_embed(float* out, int C, int gh, int gw, int base_size, float interpolation_scale, void* stream) {
  return fin(launch_sincos_pos_embed(out, C, gh, gw, base_size, interpolation_scale, (hipStream_t)stream), "sincos_pos_embed");
}

int gdf_op_resize_concat(const void* src, int src_f32, long sb, long sc, long sy, long sx, int B, int C, int H, int W, void* out,
                         int Ctot, int coff, int S, void* stream) {
  return fin(launch_resize_concat(src_f32 ? nullptr : (const half_t*)src, src_f32 ? (const float*)src : nullptr, sb, sc, sy, sx, B, C, H, W,
                                  (half_t*)out, Ctot, coff, S, (hipStream_t)stream), "resize_concat");
}
int gdf_op_avg_pool(const void* src, long sb, long sy, long sx, int B, int C, int H, int W, int r, void* out, void* stream) {
  return fin(launch_avg_pool((const half_t*)src, sb, sy, sx, B, C, H, W, r, (half_t*)out, (hipStream_t)stream), "avg_pool");
}
int gdf_op_maps_mean(const void* const* maps, int n, int B, int heads, int Q, int K, float* out, void* stream) {
  return fin(launch_maps_mean((const half_t* const*)maps, n, B, heads, Q, K, out, (hipStream_t)stream), "maps_mean");
}

// element type of the 16-bit operands of the MMDiT entry points below, per calling thread (GDF_F16 default)
static thread_local int g_e16_bf = 0;
int gdf_op_set_e16(int dtype) {
  if (dtype != GDF_F16 && dtype != GDF_BF16) { set_error("gdf_op_set_e16: GDF_F16 or GDF_BF16"); return GDF_ERR_ARG; }
  g_e16_bf = dtype == GDF_BF16;
  return GDF_OK;
}

int gdf_op_gemm_dit(const void* A, int lda, const void* W, const float* bias, int act, const float* vec, int ldvec, int vec_mul,
                    int rps, int seg_rows, int rps2, const float* res32, int ldres, void* aux16, int ldaux, void* out16,
                    int ldo16, float* out32, int ldo32, int M, int N, int K, int variant, void* stream) {
  GemmParams g{};
  if (!span_ok(((size_t)M - 1) * lda * 2 + (size_t)K * 2, (size_t)N * K * 2, "gemm_dit")) return GDF_ERR_UNSUPPORTED;
  g.A = (const half_t*)A; g.lda = lda; g.a_bytes = (uint32_t)(((size_t)M - 1) * lda * 2 + (size_t)K * 2);
  g.M = M; g.N = N; g.K = K; g.mode = A_DENSE;
  g.Wt = (const half_t*)W; g.w_bytes = (uint32_t)((size_t)N * K * 2);
  g.bias = bias; g.dit = 1; g.act = act; g.rowvec = vec; g.ldrv = ldvec; g.rv_mul = vec_mul; g.rows_per_sample = rps > 0 ? rps : 1;
  g.rv_seg_rows = seg_rows; g.rv_rps2 = rps2 > 0 ? rps2 : 1;
  g.res32 = res32; g.ldres = ldres; g.aux16 = (half_t*)aux16; g.ldaux = ldaux;
  g.out16 = (half_t*)out16; g.ldo16 = ldo16; g.out32 = out32; g.ldo32 = ldo32; g.bn = 128; g.variant = variant; g.bf16 = g_e16_bf;
  return fin(launch_gemm(g, (hipStream_t)stream), "gemm_dit");
}

int gdf_op_quant_rows_fp8(const void* x16, int ld, int R, int K, int src_bf16, void* q8, int ldq, float* scale, void* stream) {
  return fin(launch_quant_rows_fp8((const half_t*)x16, ld, R, K, src_bf16, (unsigned char*)q8, ldq, scale, (hipStream_t)stream), "quant_rows_fp8");
}

int gdf_op_gemm_mx(const void* A8, int lda, const float* a_scale, const void* W8, const float* w_scale, const float* bias, int act,
                   const float* res32, int ldres, void* out16, int ldo16, float* out32, int ldo32, int M, int N, int K, void* stream) {
  if ((K % 128) || (lda % 2)) { gdf::set_error("gemm_mx: K must be a multiple of 128, lda even"); return GDF_ERR_ARG; }
  GemmParams g{};
  if (!span_ok(((size_t)M - 1) * lda + (size_t)K, (size_t)N * K, "gemm_mx")) return GDF_ERR_UNSUPPORTED;
  // fp8 rows in 2-byte units (kernels.h GemmParams::mx)
  g.A = (const half_t*)A8; g.lda = lda / 2; g.a_bytes = (uint32_t)(((size_t)M - 1) * lda + (size_t)K);
  g.M = M; g.N = N; g.K = K / 2; g.mode = A_DENSE;
  g.Wt = (const half_t*)W8; g.w_bytes = (uint32_t)((size_t)N * K);
  g.bias = bias; g.dit = 1; g.act = act; g.rows_per_sample = 1; g.rv_rps2 = 1;
  g.res32 = res32; g.ldres = ldres; g.out16 = (half_t*)out16; g.ldo16 = ldo16; g.out32 = out32; g.ldo32 = ldo32; g.bn = 128; g.bf16 = 1;
  g.mx = 1; g.mx_rowscale = a_scale; g.mx_colscale = w_scale;
  return fin(launch_gemm(g, (hipStream_t)stream), "gemm_mx");
}

int gdf_op_layernorm_mod(const float* x32, int ld, int R, int C, float eps, const float* scale, const float* shift, int ldm,
                         int rps, int seg_rows, int rps2, void* y, void* stream) {
  return fin(launch_layernorm_mod(nullptr, x32, ld, R, C, eps, scale, shift, ldm, rps, seg_rows, rps2, (half_t*)y, (hipStream_t)stream,
                                  g_e16_bf), "layernorm_mod");
}

int gdf_op_qk_norm_rope(void* x, int ld, int R, int heads, int q_col, int k_col, const float* wq, const float* wk, float eps,
                        const float* cos_t, const float* sin_t, int pos0, int rps, void* stream) {
  return fin(launch_qk_norm_rope((half_t*)x, ld, R, heads, 128, q_col, k_col, wq, wk, eps, cos_t, sin_t, pos0, rps, (hipStream_t)stream,
                                 g_e16_bf), "qk_norm_rope");
}

int gdf_op_rope_table(const float* ids, int S, int a0, int a1, int a2, float* cos_t, float* sin_t, int row0, void* stream) {
  const int ax[3] = {a0, a1, a2};
  return fin(launch_rope_table(ids, S, 3, ax, 10000.0, cos_t, sin_t, row0, (hipStream_t)stream), "rope_table");
}

int gdf_op_attention_joint(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B,
                           int heads, int T, int S, int D, void* stream) {
  AttnParams a{};
  a.q = (const half_t*)q; a.ldq = ldq; a.k = (const half_t*)k; a.ldk = ldk; a.v = (const half_t*)v; a.ldv = ldv;
  a.o = (half_t*)o; a.ldo = ldo; a.B = B; a.heads = heads; a.Sq = T + S; a.Sk = T + S; a.D = D; a.kv_bstride = T + S;
  a.scale = 1.0f / sqrtf((float)D); a.map = nullptr; a.seg_T = T; a.bf16 = g_e16_bf;
  return fin(launch_attention(a, (hipStream_t)stream), "attention_joint");
}

}  // extern "C"
