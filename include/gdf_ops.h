/* gdf_ops.h — kernel-level entry points of libgdf.so (diagnostics / unit tests / micro-benchmarks).
 * Same conventions as gdf.h: plain C, device pointers, asynchronous on `stream`, 0 = ok.
 * Each call is one launch of the kernel the plan executor uses for that op class.            */
#ifndef GDF_OPS_H
#define GDF_OPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* out[M,N] = A[M,K](lda) * W[N,K]^T + bias (+ residual).  fp16 operands, fp32 accumulate.
 * flags: bit0 GEGLU (W rows / bias already interleaved [16 h | 16 gate] by gdf_op_relayout_geglu(group 16); out is
 *        [M,N/2]); bit1 narrow-N tile (BN=16);
 *        bits 8..19 force a tile variant (0 = auto; 128 / 160 / 256 / 320 = LDS-ring kernels of that tile, 932 / 832 / 825 / 826 =
 *        8-phase main loops: 256x320 B-resident, 256x320 A-resident, 256x256 GEGLU, 256x256 conv).  Operands larger than 2 GiB
 *        are rejected (GDF_ERR_UNSUPPORTED: 32-bit buffer offsets).  Replaces nn.Linear / 1x1 conv
 *        (/root/reference/feature/diffusers/models/attention_processor.py:241-267, attention.py:1238-1258). */
int gdf_op_gemm(const void* A, int lda, const void* W, const float* bias, const float* res32, const void* res16,
                int ldres, void* out16, int ldo16, float* out32, int ldo32, int M, int N, int K, int flags,
                void* stream);

/* 3x3 convolution, padding 1, NHWC fp16 x[B,H,W,ld>=Cin], weights in the layout gdf_op_relayout_conv3 produces
 * ([Cout][Cin/64][tap][64] fp16: channel-block-major, the nine taps innermost); stride 1|2;
 * ups=1 fuses a nearest x2 upsample of x in front of the conv.  rowvec: optional [B][Cout] fp32 added per sample.
 * Replaces nn.Conv2d in resnet.py:269,285, downsampling.py:115-118, upsampling.py:131-134,176-193.       */
int gdf_op_conv3x3(const void* x, int ld, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                   const float* rowvec, int stride, int ups, const float* res32, void* aux16, void* out16,
                   float* out32, int narrow /* bit0: BN=16; bits 8..: tile variant */, void* stream);

/* SPLIT-OPERAND forms (the opt-in "precise" plans, gdf.h gdf_plan_opts.reserved[1]): an activation is a pair of fp16 numbers
 * hi = fp16(v), lo = fp16(v - hi) stored in ONE row, lo `a_lo` (input) / `o16_lo` / `y_lo` / `o_lo` (output) elements after hi, and a
 * contraction runs over [hi | lo] against the weight matrix read twice: out = (hi + lo) W^T to fp32 accuracy.  a_lo = 0 / *_lo = 0:
 * the plain form.  Kw / Cin are the WEIGHT matrix's contraction sizes.  They replace the same reference call sites as their
 * plain counterparts; the reference itself has no such mode (it rounds more: fp16 activations AND an fp16 residual stream). */
int gdf_op_gemm_split(const void* A, int lda, int a_lo, const void* W, const float* bias, const float* res32, int ldres, void* out16,
                      int ldo16, int o16_lo, float* out32, int ldo32, int M, int N, int Kw, int flags /* bit0 GEGLU */, void* stream);
int gdf_op_conv3x3_split(const void* x, int ld, int a_lo, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                         int stride, int ups, const float* res32, void* out16, int ldo16, int o16_lo, float* out32, void* stream);
int gdf_op_layernorm_split(const float* x32, int ld, int R, int C, float eps, const float* gamma, const float* beta, void* y, int ldy,
                           int y_lo, void* stream);
/* x: fp32 (x32, ld) or a split fp16 pair (x16, ld, x_lo) */
int gdf_op_groupnorm_split(const void* x16, int x_lo, const float* x32, int ld, int B, int HW, int C, int G, float eps, const float* gamma,
                           const float* beta, int silu, void* y, int ldy, int y_lo, void* scratch, void* stream);
int gdf_op_attention_split(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int o_lo, int B,
                           int heads, int Sq, int Sk, int D, void* map, void* stream);

/* q, k and v as split pairs (hi, lo = fp16(x - hi), lo at +qkv_lo elements in the same row: what a GEMM with o16_lo writes): the softmax sees
 * q k^T contracted over both halves, O accumulates P (v_hi + v_lo) — the full-split UNet plans (attention_processor.py:3311-3313 in fp32 terms).
 * Head dims 40 / 64 / 80; others read the hi halves only. */
int gdf_op_attention_pair(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int qkv_lo, void* o, int ldo, int o_lo, int B,
                          int heads, int Sq, int Sk, int D, void* stream);

/* Deterministic split-K form of gdf_op_conv3x3 (few output tiles, long K — the 8x8-level convs of SD1.5): the K range is cut
 * into `splitk` contiguous parts (0 = the plan builder's heuristic, gdf_op_splitk_factor), every part writes raw fp32 partial
 * sums to its own slab of `ws` (splitk * M * Cout floats, M = B * OH * OW), a second kernel adds the slabs in a fixed order and
 * applies the same epilogue as gdf_op_conv3x3.  Same result as the unsplit conv up to fp32 summation order. */
int gdf_op_conv3x3_splitk(const void* x, int ld, int B, int H, int W, int Cin, const void* Wt, int Cout, const float* bias,
                          const float* rowvec, int stride, int ups, const float* res32, void* aux16, void* out16,
                          float* out32, int splitk, float* ws, void* stream);
int gdf_op_splitk_factor(int M, int N, int K, int conv);

/* conv_in: x NCHW fp16 [B,Cin<=8,H,W] -> NHWC [B,H,W,Cout]; weights in diffusers OIHW fp16 layout.
 * scratch: B*H*W*16 + Cout*256 bytes.                                                                  */
int gdf_op_conv_in(const void* x_nchw, int B, int Cin, int H, int W, const void* w_oihw, const float* bias, int Cout,
                   void* out16, void* scratch, void* stream);

/* softmax(q k^T * D^-0.5) v per head; rows are tokens, head h at columns [h*D, h*D+D).
 * map != NULL additionally writes the probabilities (B, heads, Sq, Sk) fp16 ('-map' hooks).
 * Replaces F.scaled_dot_product_attention (attention_processor.py:3311-3313) /
 * get_attention_scores+bmm (attention_processor.py:640-685, components/attention.py:232-246).          */
int gdf_op_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                     int B, int heads, int Sq, int Sk, int D, void* map, void* stream);

/* GroupNorm(G groups, eps) [+SiLU] over NHWC x (fp16, ld) or x32 (fp32, ld); y contiguous fp16 [B*HW][C].
 * scratch: gdf_op_groupnorm_scratch_bytes(B,HW,C).                                                      */
size_t gdf_op_groupnorm_scratch_bytes(int B, int HW, int C);
int gdf_op_groupnorm(const void* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps,
                     const float* gamma, const float* beta, int silu, void* y, void* scratch, void* stream);

/* LayerNorm over the last dim. */
int gdf_op_layernorm(const void* x16, const float* x32, int ld, int R, int C, float eps, const float* gamma,
                     const float* beta, void* y, void* stream);

/* strided fp16/fp32 -> fp16 2-D copy (the hook-store kernel). */
int gdf_op_copy2d(const void* s16, const float* s32, int lds, void* dst, int ldd, int R, int C, void* stream);

/* weight re-layout helpers used by the tests: OIHW -> OHWI, GEGLU row interleave. */
int gdf_op_relayout_conv3(const void* w_oihw_f16, void* dst, int O, int I, void* stream);
int gdf_op_relayout_geglu(const void* w_f16, const float* bias, void* w_dst, float* bias_dst, int R, int K, int group /*16*/, void* stream);

/* out[m][n] = (accumulate ? out : 0) + bias[n] + sum_k act(x[m][k]) W[n][k]; x fp32 [M][ldx] (M small), W fp16 [N][K], out fp32.
 * The time / text / adaLN-modulation embedding linears (unet_2d_condition.py:1142-1162, resnet.py:343-346; MMDiT adaLN
 * `linear(silu(temb))`, stacked over all blocks: N = 1.06 M columns -> LDS-staged wide kernel).                          */
int gdf_op_small_linear(const float* x, int ldx, int M, int K, const void* W, const float* bias, int N, int silu_in,
                        int accumulate, float* out, int ldo, void* stream);

/* in-place row softmax of fp16 scores: x[r][0..n) = softmax(scale * x[r][0..n)) with fp32 math (VAE mid-block attention,
 * attention_processor.py:3311-3313 run as explicit GEMMs; n <= 16384, n % 8 == 0). */
int gdf_op_softmax_rows(void* x, int ld, int R, int n, float scale, void* stream);

/* PatchEmbed positional table (PixArt): out fp32 [gh*gw][C] = get_2d_sincos_pos_embed(C, (gh, gw), base_size, interpolation_scale). */
int gdf_op_sincos_pos_embed(float* out, int C, int gh, int gw, int base_size, float interpolation_scale, void* stream);

/* ---- output-stage post-processing (csrc/post.hip) ---- */

/* `--aggregate_output` (extract_feature.py:113-125): nearest-resize one layer — logical (B,C,H,W), element strides
 * sb/sc/sy/sx, fp16 (src_f32 = 0) or fp32 — to S x S (PyTorch `nearest`: floor(dst * in / out)) and store it as channels
 * [coff, coff + C) of out (B, Ctot, S, S) fp16 contiguous; one call per layer performs F.interpolate + torch.cat(dim=1). */
int gdf_op_resize_concat(const void* src, int src_f32, long sb, long sc, long sy, long sx, int B, int C, int H, int W, void* out,
                         int Ctot, int coff, int S, void* stream);
/* `feature_resize` (components/feature_extractor.py:51-53): r x r mean (adaptive_avg_pool2d to (H/r, W/r)) of a channels-last
 * hook (strides sb, 1, sy, sx; C % 8 == 0) -> (B, H/r, W/r, C) fp16, fp32 accumulation. */
int gdf_op_avg_pool(const void* src, long sb, long sy, long sx, int B, int C, int H, int W, int r, void* out, void* stream);
/* aggregated `attn` feature (components/attention.py:238-244, 141-161): mean over heads (rounded to fp16 like the
 * reference's `attention_probs.mean(1)`), then mean over the n <= 32 maps (B, heads, Q, K) fp16 of one (category, size)
 * group -> (B, Q, K) fp32; gdf_op_resize_concat then turns it into the (B, K, img/8, img/8) slice of the feature. */
int gdf_op_maps_mean(const void* const* maps, int n, int B, int heads, int Q, int K, float* out, void* stream);

/* ---- MMDiT (Flux) kernels (reference files cited in csrc/dit.hip, gdf_flux.h) ---- */

/* Element type of the 16-bit operands ("e16": A, W, out16, q/k/v/o, y) of the MMDiT entry points below, per calling thread:
 * GDF_F16 (default) or GDF_BF16 (what a gdf_flux_desc.compute_dtype = GDF_BF16 model runs).  aux16 (a hook) is always fp16. */
int gdf_op_set_e16(int dtype);

/* Dense GEMM with the MMDiT epilogue: v = A W^T + bias; act=1: tanh-GELU; vec != NULL: v = vec_mul ? v * vec[s] : v + vec[s]
 * (s = row / rps for row < seg_rows or seg_rows == 0, else (row - seg_rows) / rps2; vec fp32 rows of ldvec);
 * aux16 (optional) receives fp16(v) BEFORE the gate; then + res32, stores out16 / out32.
 * Replaces nn.Linear + gate/residual arithmetic of transformer_flux.py:95-106, 191-218. */
/* 'fp8-mx' MMDiT plans (gdf_flux.h, GDF_FP8MX; opt-in, LOWER precision than the reference's bf16): 16-bit rows -> OCP e4m3 bytes with one
 * power-of-two scale per row, q = fp8(v / scale[r]) (activations per token, weights per output channel), and the GEMM on such operands:
 * out = act((A8 W8^T) * a_scale[row] * w_scale[col] + bias) (+ res32), v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (the
 * per-row / per-channel scales are applied to the fp32 accumulators), bf16 out16.  K % 128 == 0; 256x256 tiles.
 * Replaces the same nn.Linear call sites as gdf_op_gemm_dit (transformer_flux.py:86-112, 167-226). */
int gdf_op_quant_rows_fp8(const void* x16, int ld, int R, int K, int src_bf16, void* q8, int ldq, float* scale, void* stream);
int gdf_op_gemm_mx(const void* A8, int lda, const float* a_scale, const void* W8, const float* w_scale, const float* bias, int act,
                   const float* res32, int ldres, void* out16, int ldo16, float* out32, int ldo32, int M, int N, int K, void* stream);

int gdf_op_gemm_dit(const void* A, int lda, const void* W, const float* bias, int act, const float* vec, int ldvec, int vec_mul,
                    int rps, int seg_rows, int rps2, const float* res32, int ldres, void* aux16, int ldaux, void* out16,
                    int ldo16, float* out32, int ldo32, int M, int N, int K, int variant, void* stream);

/* y = LayerNorm(x, eps, no affine) * (1 + scale[s]) + shift[s]  (AdaLayerNormZero & co.); x fp32 [R][ld], y fp16 [R][C]. */
int gdf_op_layernorm_mod(const float* x32, int ld, int R, int C, float eps, const float* scale, const float* shift, int ldm,
                         int rps, int seg_rows, int rps2, void* y, void* stream);

/* In place RMSNorm(q), RMSNorm(k) per head (D = 128) + rotary embedding on fp16 rows [R][ld]; position = pos0 + r % rps. */
int gdf_op_qk_norm_rope(void* x, int ld, int R, int heads, int q_col, int k_col, const float* wq, const float* wk, float eps,
                        const float* cos_t, const float* sin_t, int pos0, int rps, void* stream);

/* FluxPosEmbed: ids fp32 [S][3] -> cos/sin fp32 [S][axes0+axes1+axes2] written at row offset row0. */
int gdf_op_rope_table(const float* ids, int S, int a0, int a1, int a2, float* cos_t, float* sin_t, int row0, void* stream);

/* Joint attention over [T text | S image] tokens per sample, rows region-major [B*T text rows][B*S image rows]. */
int gdf_op_attention_joint(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B,
                           int heads, int T, int S, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif
