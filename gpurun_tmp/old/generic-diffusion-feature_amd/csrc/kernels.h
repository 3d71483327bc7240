// Internal launch interface of the gfx950 kernels (C++ linkage; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef _Float16 half_t;

namespace gdf {

// One-time raise of a kernel's dynamic-LDS limit, per (kernel, device): `mask` is a static of the calling launcher with one
// bit per device ordinal.  Thread-safe without a lock — several extractors may run in threads of one process, one per
// device (reference correspondence/correspondence/aggregation_network.py:67-95); two racing first launches both set the
// attribute, which is idempotent.
inline hipError_t ensure_dyn_smem(std::atomic<uint64_t>& mask, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  mask.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// ---- 16-bit element type of the MMDiT (Flux) path --------------------------------------------------------------------
// The reference runs Flux in torch.bfloat16 (components/models.py:158-169): real FLUX.1-dev activations leave the fp16 range.
// A model created with compute_dtype = GDF_BF16 keeps weights, activations and MFMA operands in bf16 (`half_t` is then only
// the 16-bit container); hooks stay fp16 like the reference's (feature_extractor.py:59-60), written with SATURATING casts.
#if defined(__HIPCC__)
__device__ __forceinline__ float e16_to_f32(half_t h, int bf) {
  return bf ? __uint_as_float((uint32_t)__builtin_bit_cast(unsigned short, h) << 16) : (float)h;
}
__device__ __forceinline__ half_t f32_to_e16(float v, int bf) {          // round to nearest even in both formats
  return bf ? __builtin_bit_cast(half_t, (__bf16)v) : (half_t)v;
}
__device__ __forceinline__ half_t f32_to_f16_sat(float v) {              // fp16 with saturation instead of +-inf (NaN stays NaN)
  return (half_t)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
}
#endif

// ------------------------------------------------------------------------------------------------
// MFMA GEMM / implicit-GEMM convolution:  D[M,N] = A[M,K] * Wt[N,K]^T  (+ fused epilogue)
// ------------------------------------------------------------------------------------------------
enum { A_DENSE = 0, A_CONV3 = 1, A_CONV_SMALLC = 2 };

struct GemmParams {
  // ---- A operand (fp16). DENSE: row-major [M][K] with leading dimension lda (elements).
  //      CONV3: NHWC activation [B][H][W][lda>=Cin]; rows of the GEMM are output pixels.
  const half_t* A;
  int lda;
  uint32_t a_bytes;      // extent of A in bytes (buffer bounds: loads past it return 0)
  int M, N, K;           // K = Cin (dense) or 9*Cin (conv3)
  int mode;
  int H, W;              // source spatial size (conv)
  int OH, OW;            // output spatial size (conv)
  int stride;            // 1 or 2 (conv)
  int ups;               // 1: source is nearest-upsampled x2 before the conv (Upsample2D fused)
  int pad0;              // 0: zero padding 1 on every side; 1: pad right / bottom only (F.pad (0,1,0,1), Downsample2D padding=0)
  int Cin;
  // ---- B operand: weights [N][K] fp16, K contiguous (conv3: K index = ((c / 64) * 9 + tap) * 64 + c % 64, launch_relayout_conv cblk = 64)
  const half_t* Wt;
  uint32_t w_bytes;
  // ---- epilogue:  v = acc + bias[col] + rowvec[row / rows_per_sample][col]
  //                 aux16 = fp16(v)                       (pre-residual copy: `res-increment` hook)
  //                 v += res32|res16[row][col]
  //                 out16 = fp16(v), out32 = v
  //      GEGLU: columns come in [32 h | 32 gate] groups; out col = h * gelu(gate), N_out = N/2
  const float* bias;
  const float* rowvec;
  int rows_per_sample;
  int ldrv;
  const float* res32;
  const half_t* res16;
  int ldres;
  half_t* out16;
  int ldo16;
  float* out32;
  int ldo32;
  half_t* aux16;
  int ldaux;
  int geglu;             // != 0: weight rows / bias are interleaved [16 h | 16 gate]; out = (h) * gelu(gate), N_out = N/2
  int bn;                // 128 (default) or 16 (very narrow N, e.g. conv_out)
  int splitk;            // > 1 (2-stage ring tiles, batch <= 1): blockIdx.y selects one of `splitk` contiguous K ranges; the launch then
                         // writes RAW fp32 partial sums to out32 + blockIdx.y * o32_sstride and launch_gemm_splitk() finishes the epilogue
  long o32_sstride;      // elements between consecutive partial-sum slabs
  int batch;             // > 1: blockIdx.y selects one of `batch` independent problems sharing A (grouped text-K/V projections)
  long w_bstride;        // elements between consecutive weight matrices
  long o_bstride;        // elements between consecutive out16 matrices
  int sb_gm, sb_gn;      // (set by launch_gemm) tiles per XCD super-block, 0 = linear XCD-chunked order
  int no_superblock;     // diagnostics: force the linear order
  int no_early_mma;      // diagnostics: disable the opposite-order heads of the two waves sharing a SIMD (256x128 variant)
  int variant;           // 0 = auto; 128 / 160 / 256 force the 128x128, 128x160 or 256x128 tile (diagnostics)
  // ---- MMDiT epilogue (dit != 0 selects gemm_dit_kernel; dense only) ----
  //      v = acc + bias;  act == 1: v = gelu_tanh(v);  rowvec: v = rv_mul ? v * rowvec[sample][col] : v + rowvec[sample][col];
  //      then residual / stores as above.  sample = row / rows_per_sample for row < rv_seg_rows (or rv_seg_rows == 0),
  //      else (row - rv_seg_rows) / rv_rps2  (joint stream laid out [all text rows][all image rows]).
  int dit;
  int act;
  int rv_mul;
  int rv_seg_rows, rv_rps2;
  int rv_tok;            // 1: the row vector is per TOKEN (row % rows_per_sample), e.g. the PatchEmbed positional table
  // MMDiT QKV projection (dit only, 256x256 tile: one 128-wide head per wave tile): RMSNorm over each 128-column head of the
  // q columns [0, qkn_nq) and the k columns [qkn_nq, 2 qkn_nq), then the rotary embedding of the row's position
  // (Attention.norm_q / norm_k + apply_rotary_emb, attention_processor.py:2300-2335), applied to the fp32 accumulators
  int qkn_nq;            // 0 = off
  const float* qkn_wq; const float* qkn_wk; float qkn_eps;
  const float* rope_cos; const float* rope_sin;      // [position][128]
  int qkn_pos0, qkn_rps;                             // position of row r: qkn_pos0 + r % qkn_rps           (r <  qkn_seg_rows or no segment)
  int qkn_seg_rows, qkn_pos1, qkn_rps2;              //                    qkn_pos1 + (r - seg_rows) % rps2  (r >= qkn_seg_rows > 0)
  int bf16;              // dit only: A, Wt and out16 are bf16 (mfma_f32_16x16x32_bf16); aux16 (a hook) stays fp16, saturating
  // fp16 RANGE control (VAE encoder: the stock SDXL VAE's residual stream leaves the fp16 range, the reference upcasts it to
  // fp32).  v = acc * acc_scale + bias ...; out16 = fp16(v * out16_scale).  A tensor whose fp16 image is stored with
  // out16_scale = 2^-k is read back by its consumer GEMM with acc_scale = 2^k (exact: powers of two).  0 means 1.
  float acc_scale, out16_scale;
  // SPLIT OPERANDS ("precise" plans, gdf_plan_opts.precise): an activation is stored as two fp16 numbers hi = fp16(v), lo = fp16(v - hi)
  // (22 mantissa bits) and the contraction runs over [hi | lo] against [W | W]:  A.W = hi.W + lo.W with K doubled, the weights read
  // twice from the SAME matrix.  k_w > 0: K of the weight matrix (K == 2 k_w); K-tile kt >= k_w / 64 of A starts a_lo_bytes after the
  // hi columns (conv3: per pixel) and reads weight K-tile kt - k_w / 64.  o16_lo > 0: out16 is written as such a pair, lo at
  // out16 + o16_lo elements (same row).  0 everywhere = plain fp16 operands (the default plans).
  int k_w;
  uint32_t a_lo_bytes;
  int o16_lo;
  // CU PARTITION (gdf_plan_opts.reserved[2]): > 0 = the launch stream is restricted to this many CUs (a CU-masked stream,
  // gdf_stream_create_cu_mask): tile selection, persistent grids and the XCD super-block order count workgroup slots on `cus`
  // CUs instead of the whole chip.  0 = the whole device.
  int cus;
  // MMDiT 'bfloat16x2' plans (gdf_flux_desc.compute_dtype = GDF_BF16X2): a bf16 kernel (dit, bf16) that stores out16 as SATURATING fp16 — the
  // q / k / v buffer of the attention kernel, whose internals run in fp16 in that mode (q, k are RMS-normalised, v is a linear of a
  // normalised tensor: inside the fp16 range by construction; 11 mantissa bits instead of 8)
  int out_f16;
  // MMDiT 'fp8-mx' plans: A and Wt hold fp8 (OCP e4m3) bytes — lda, K, a_bytes, w_bytes are given in 2-BYTE units exactly as for a 16-bit matrix
  // of half the width (a K-tile is 128 bytes either way) — multiplied with v_mfma_scale_f32_16x16x128_f8f6f4; the operands' power-of-two
  // scales are undone on the fp32 accumulators: acc * mx_rowscale[row] * mx_colscale[col] (either may be null = 1)
  int mx;
  const float* mx_rowscale;
  const float* mx_colscale;
  // GROUPNORM STATISTICS FROM THE PRODUCER (VAE op programs; 3x3 convs on the tiles gemm_gn_slab_rows() accepts): besides storing
  // out16, the epilogue sums x and x^2 of the stored values (v * out16_scale) per output channel over slabs of gemm_gn_slab_rows()
  // consecutive rows and writes them as gn_partial[(row / slab_rows) * N + c][2] — the layout launch_gn_finalize() reads, so the
  // GroupNorm that consumes the tensor needs no statistics pass of its own (a full HBM read of a 1-4 GB tensor at 1024^2).
  // Requires M % slab_rows == 0 and (rows per sample) % slab_rows == 0.  nullptr = off (every UNet / MMDiT plan).
  float* gn_partial;
  // DE-PHASING (round 6 experiment, compiled in by -DGDF_STAGGER only; set by launch_gemm from GDF_STAGGER_US / GDF_STAGGER_GROUPS; measured null: workgroup b of the FIRST
  // round (b < stagger_wgs) waits ((b >> 3) % groups) x ticks of the 100-MHz realtime counter before its first tile (stagger = ticks | groups << 24),
  // so that on a multi-round launch the CUs reach their epilogues at different times (tools/ab_stagger.sh,
  // profiles/r06_ab_deferred_epilogue.txt)
  int stagger, stagger_wgs;
};
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
// rows per statistics slab if launch_gemm can run `p` (shape, mode, epilogue form) with gn_partial set, else 0
int gemm_gn_slab_rows(const GemmParams& p);
// Deterministic split-K for problems with few output tiles and a long K (the 8x8-level 3x3 convs of SD1.5: 160 tiles of 128x128,
// K = 11520..23040): gemm_splitk_factor() > 1 says it pays; the caller provides `splitk * M * N` floats of workspace, the GEMM
// launch writes one raw partial-sum slab per K range and splitk_reduce_kernel sums them in a fixed order and applies the epilogue.
int gemm_splitk_factor(const GemmParams& p);
hipError_t launch_gemm_splitk(const GemmParams& p, int splitk, float* ws, hipStream_t s);
const char* gemm_kernel_name(const GemmParams& p);
bool gemm_qkn_ok(int M, int N, int K);   // kernel symbol launch_gemm would pick (only M,N,K,mode,geglu,bn,variant are read)

// ------------------------------------------------------------------------------------------------
// flash attention (self / cross), fp16 in, fp32 softmax, fp16 out
//   q: rows (b*Sq + i), head h at columns [h*D, h*D+D) with leading dim ldq; same for k, v, o.
// ------------------------------------------------------------------------------------------------
struct AttnParams {
  const half_t* q; int ldq;
  const half_t* k; int ldk;
  const half_t* v; int ldv;
  half_t* o; int ldo;
  int B, heads, Sq, Sk, D;
  int kv_bstride;        // rows between consecutive samples' K/V (Sk normally; 0 = all samples share one K/V set)
  float scale;
  half_t* map;           // optional: attention probabilities (B, heads, Sq, Sk) fp16 ('-map' hooks); seg_T > 0: `self-map`
  half_t* map2;          // seg_T > 0 only: `cross-map` (B, heads, Sq - seg_T, seg_T) (image queries x text keys)
  const int* kv_len;     // optional [B]: keys [kv_len[b], Sk) of sample b are masked out (prefix text mask)
  int seg_T;             // > 0: MMDiT joint sequence, region-major rows [B x seg_T text][B x (Sq - seg_T) image] (Sq == Sk)
  int bf16;              // D = 128 only: q, k, v, o are bf16 (mfma_f32_32x32x16_bf16, P rounded to bf16); maps stay fp16
  int o_lo;              // > 0 ("precise" plans): o is written as a split (hi, lo) pair, lo at o + o_lo elements in the same row
  int o_pair_bf16;       // with o_lo > 0 and fp16 q / k / v (bf16 == 0): the pair is written as bf16 hi + bf16 lo (MMDiT 'bfloat16x2' plans)
  int q_lo, kv_lo;       // both > 0 (full-split plans, fp16 UNet attention with 40 <= D <= 80): q, k and v are split (hi, lo) pairs, lo at +q_lo (q) /
                         // +kv_lo (k, v) elements in the same row; the kernel contracts over both halves (attn_kernel<..., QKP>).  Other head dims
                         // read the hi halves only.
  float o_scale;         // != 0: o is stored multiplied by this power of two (MMDiT 'float16s' plans: the [attn | mlp] operand rows of the single
                         // blocks share ONE fp16 range scale, undone on the consuming GEMM's accumulators)
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// normalisation / elementwise
// ------------------------------------------------------------------------------------------------
// GroupNorm over NHWC rows x [B][HW][ld] (C channels used, G groups).  Two launches:
//   gn_stats : per-(sample, channel) affine table ab[b][c] = (rstd*gamma, beta - mean*rstd*gamma)
//              (`partial` is scratch of gn_partial_floats(B, HW, C) floats)
//   gn_apply : y = act(x*a + b) as contiguous fp16 [B*HW][C]; silu=1 applies v*sigmoid(v)
size_t gn_partial_floats(int B, int HW, int C);
// split operands ("precise" plans): x_lo > 0 = the fp16 source is a (hi, lo) pair with lo x_lo elements after hi in the row;
// y_lo > 0 = y is written as such a pair (rows of ldy elements, lo at column offset y_lo); ldy = 0 means C
hipError_t launch_gn_stats(const half_t* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps,
                           const float* gamma, const float* beta, float* partial, float* ab, hipStream_t s, int x_lo = 0);
hipError_t launch_gn_apply(const half_t* x16, const float* x32, int ld, int B, int HW, int C,
                           const float* ab, int silu, half_t* y, hipStream_t s, int x_lo = 0, int ldy = 0, int y_lo = 0);
// the second stage of launch_gn_stats alone: `partial` = [B][nslab][C][2] per-slab channel sums (sum x, sum x^2) produced by a GEMM
// epilogue (GemmParams::gn_partial) -> ab
// `fold`: scratch of gn_fold_floats(B, nslab, C) floats (0 = none needed) for the pre-reduction of many short slabs
size_t gn_fold_floats(int B, int nslab, int C);
hipError_t launch_gn_finalize(const float* partial, int nslab, int B, int HW, int C, int G, float eps, const float* gamma,
                              const float* beta, float* ab, float* fold, hipStream_t s);
// the fold pass of launch_gn_finalize on its own (many short slabs -> gn_fold_out_slabs(nslab) <= 128 slabs)
int gn_fold_out_slabs(int nslab);
bool gn_fold_ok(int C);
hipError_t launch_gn_fold(const float* partial, int nslab, int B, int C, float* fold, hipStream_t s);
// statistics from the producer's epilogue -> finalize + apply (+SiLU) in ONE launch (round 6); gn_finalize_apply_slab(...) != 0 says whether it applies
int gn_finalize_apply_slab(int C, int G);
hipError_t launch_gn_finalize_apply(const float* partial, int nslab, const half_t* x16, int ld, int B, int HW, int C, int G, float eps,
                                    const float* gamma, const float* beta, int silu, half_t* y, hipStream_t s, int ldy = 0, int y_lo = 0);
// single-launch GroupNorm (+SiLU) for small feature maps; gn_fused_slab(...) != 0 says whether it applies
int gn_fused_slab(int B, int HW, int C, int G);
hipError_t launch_gn_fused(const half_t* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps, const float* gamma,
                           const float* beta, int silu, half_t* y, hipStream_t s, int x_lo = 0, int ldy = 0, int y_lo = 0);
// LayerNorm over the last dim (C), rows x [R][ld]; y contiguous fp16 [R][C]
hipError_t launch_layernorm(const half_t* x16, const float* x32, int ld, int R, int C, float eps,
                            const float* gamma, const float* beta, half_t* y, hipStream_t s, int ldy = 0, int y_lo = 0);
// LayerNorm without affine + adaLN modulation (AdaLayerNormZero / ZeroSingle / Continuous and the norm2 modulate of
// the MMDiT blocks): y = LN(x, eps) * (1 + scale[s][c]) + shift[s][c]; s = row / rps for row < seg_rows (or
// seg_rows == 0), else (row - seg_rows) / rps2.  x fp32 (or fp16) [R][ld], y fp16 [R][C], scale/shift fp32 rows of ldm.
// y_lo > 0: y is written as a split pair (rows of ldy elements, hi at column 0, lo = e16(v - hi) at column y_lo)
hipError_t launch_layernorm_mod(const half_t* x16, const float* x32, int ld, int R, int C, float eps, const float* scale,
                                const float* shift, int ldm, int rps, int seg_rows, int rps2, half_t* y, hipStream_t s,
                                int bf16 = 0, int ldy = 0, int y_lo = 0, unsigned char* q8 = nullptr, int ldq8 = 0, float* q8_scale = nullptr);
// 'fp8-mx' plans: 16-bit rows [R][ld] (K columns) -> fp8 e4m3 [R][ldq] with one power-of-two scale per row (q = fp8(v / scale[r])).
// launch_layernorm_mod's q8 / q8_scale write the same form of its own output in the same pass.
hipError_t launch_quant_rows_fp8(const half_t* x, int ld, int R, int K, int bf16, unsigned char* q, int ldq, float* scale, hipStream_t s);
// RMSNorm(q), RMSNorm(k) per head + rotary embedding, in place on rows [R][ld] fp16: q heads at columns
// q_col + h*D, k heads at k_col + h*D (D = 128); position of row r = pos0 + r % rps; cos/sin fp32 [pos][D].
hipError_t launch_qk_norm_rope(half_t* x, int ld, int R, int heads, int D, int q_col, int k_col, const float* wq,
                               const float* wk, float eps, const float* cos_t, const float* sin_t, int pos0, int rps,
                               hipStream_t s, int bf16 = 0);
// FluxPosEmbed: ids fp32 [S][n_axes] -> cos/sin fp32 [S][sum(axes_dim)] (float64 angles, repeat-interleaved pairs)
hipError_t launch_rope_table(const float* ids, int S, int n_axes, const int* axes_dim, double theta, float* cos_t,
                             float* sin_t, int row0, hipStream_t s);
// in-place row softmax of fp16 scores: x[r][0..n) = softmax(scale * x[r][0..n)) (fp32 math), rows of ld halves
hipError_t launch_softmax_rows(half_t* x, int ld, int R, int n, float scale, hipStream_t s);
// VAE tail: moments = quant_conv(h) (1x1, [2L][2L] fp16 weights, fp32 bias; wq == NULL: identity), mean / logvar split,
// logvar clamp(-30, 20), z = mean + exp(0.5 logvar) * eps (eps == NULL: mode), lat = scaling * z,
// out = in_scale * (noise_a * lat + noise_b * noise) (noise == NULL: no noise) -> NCHW fp16 (B, L, H, W).  h: fp32 [B*HW][2L].
hipError_t launch_vae_finish(const float* h, int B, int HW, int L, const half_t* wq, const float* bq, const half_t* eps,
                             const half_t* noise, float scaling, float noise_a, float noise_b, float in_scale, half_t* out,
                             hipStream_t s);
// VAE decoder head: z = (c_sample * latents + c_eps * noise_pred) * inv_scaling; y = post_quant_conv(z) (wq == NULL: identity)
// -> NHWC fp16 padded to 8 channels.  latents / noise_pred NCHW fp16 (B, L, H, W); noise_pred may be NULL (plain decode)
hipError_t launch_vae_dec_prepare(const half_t* lat, const half_t* eps, int B, int HW, int L, float ca, float cb, float inv_sf,
                                  const half_t* wq, const float* bq, half_t* nhwc8, hipStream_t s);
// PatchEmbed positional table (embeddings.get_2d_sincos_pos_embed as PatchEmbed calls it): out fp32 [gh*gw][C]
hipError_t launch_sincos_pos_embed(float* out, int C, int gh, int gw, int base_size, float interpolation_scale, hipStream_t s);
// out[b][i] = table[i] + vec[b][i % period]   (ada_norm_single: scale_shift_table + timestep embedding, all blocks at once)
hipError_t launch_add_table(const float* table, const float* vec, int ldvec, int period, int B, long n, float* out, long ldo,
                            hipStream_t s);
// latents NCHW fp16 (B,Cin,H,W) -> patch rows [B*(H/p)*(W/p)][kpad] fp16, column = (c*p + py)*p + px (Conv2d weight order), zero padded
hipError_t launch_patchify(const half_t* x, int B, int Cin, int H, int W, int p, int kpad, half_t* out, hipStream_t s);
// token rows [B*gh*gw][p*p*Cout] fp16 -> NCHW fp16 (B,Cout,gh*p,gw*p)  (transformer_2d.py:563-570 einsum nhwpqc->nchpwq)
hipError_t launch_unpatchify(const half_t* x, int B, int Cout, int gh, int gw, int p, half_t* out, hipStream_t s);
// dst[r][0..kdst) = src[r][0..ksrc) zero padded (weights whose K is not a multiple of 64: the 2x2 patch conv)
hipError_t launch_relayout_rows_padk(const void* src, int src_f32, half_t* dst, int R, int ksrc, int kdst, hipStream_t s);
// out[i] = silu(x[i])
hipError_t launch_silu_vec(const float* x, float* out, long n, hipStream_t s);
// strided 2-D copy with cast to fp16: dst[r][c] = src[r][c]   (hook stores)
// src_bf16: s16 holds bf16; sat: clamp to the fp16 range instead of producing +-inf (hook stores of the bf16 / MMDiT path)
// s_lo > 0: the 16-bit source is a split pair (lo s_lo elements after hi in the row): dst = fp16(hi + lo)
hipError_t launch_copy2d(const half_t* s16, const float* s32, int lds_, half_t* dst, int ldd, int R, int C,
                         hipStream_t s, int src_bf16 = 0, int sat = 0, int s_lo = 0, float scale = 1.0f);   // scale != 1: dst = fp16(scale * src)
// latents NCHW fp16 (B,Cin,H,W) -> NHWC padded to 8 channels (conv_in operand) and optional NHWC hook copy
hipError_t launch_pack_latents(const half_t* x, int B, int Cin, int H, int W, half_t* nhwc8, half_t* hook_nhwc,
                               hipStream_t s);
// sinusoidal embedding: out[b][off + j] (dim entries, [cos|sin] order = flip_sin_to_cos) of t[b*tstride + ti]
// tscale multiplies t first (Flux: `timestep * 1000`, transformer_flux.py:472)
hipError_t launch_sinusoid(const float* t, int B, int n_per_row, int dim, float* out, int ldo, int col_off,
                           int round_f16, hipStream_t s, float tscale = 1.0f);
// widen fp16 vector rows into fp32: out[b][col_off + j] = x[b][j]
hipError_t launch_widen(const half_t* x, int B, int n, float* out, int ldo, int col_off, hipStream_t s, int src_bf16 = 0);
// small-M linear in fp32 vectors: out[m][n] = (accum? out : 0) + bias[n] + sum_k act(x[m][k]) * W[n][k]
hipError_t launch_small_linear(const float* x, int ldx, int M, int K, const half_t* Wt, const float* bias, int N,
                               int silu_in, int accumulate, float* out, int ldo, hipStream_t s, int w_bf16 = 0);

// ------------------------------------------------------------------------------------------------
// output-stage post-processing (csrc/post.hip): `--aggregate_output`, `feature_resize`, aggregated attention feature
// ------------------------------------------------------------------------------------------------
// nearest-resize one layer (logical (B,C,H,W), element strides sb/sc/sy/sx, fp16 or fp32) to S x S and store it as channels
// [coff, coff + C) of out (B, Ctot, S, S) fp16 contiguous
hipError_t launch_resize_concat(const half_t* s16, const float* s32, long sb, long sc, long sy, long sx, int B, int C, int H, int W,
                                half_t* out, int Ctot, int coff, int S, hipStream_t s);
// r x r mean of a channels-last hook (B,C,H,W; strides sb, 1, sy, sx) -> (B, H/r, W/r, C) fp16
hipError_t launch_avg_pool(const half_t* src, long sb, long sy, long sx, int B, int C, int H, int W, int r, half_t* out, hipStream_t s);
// mean over heads and over n <= 32 maps (B, heads, Q, K) fp16 -> (B, Q, K) fp32
hipError_t launch_maps_mean(const half_t* const* maps, int n, int B, int heads, int Q, int K, float* out, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// weight re-layout (model load time)
// ------------------------------------------------------------------------------------------------
// `src_f32` is the source dtype code of gdf_model_set_param: 0 fp16, 1 fp32, 2 bf16
// generic: dst[o][t][i] = src[o][i][t]  (OIHW -> OHWI with T = kh*kw); i padded to ipad, t to tpad
// cblk > 0: dst[o][i / cblk][t][i % cblk] (the 3x3-conv K order of gemm_kernel<A_CONV3>: K index = ((c / 64) * 9 + tap) * 64 + c % 64)
hipError_t launch_relayout_conv(const void* src, int src_f32, half_t* dst, int O, int I, int T, int ipad, int tpad,
                                hipStream_t s, int cblk = 0);
// rows: dst[rowmap(r)][k] = src[r][k] for r in [0,R); rowmap: 0 identity+row_off, 1 GEGLU interleave (half = R/2)
hipError_t launch_relayout_rows(const void* src, int src_f32, half_t* dst, int R, int K, int row_off, int geglu /*0 or interleave group (16)*/,
                                hipStream_t s, int dst_bf16 = 0);
// vectors to fp32 with the same row mapping
hipError_t launch_relayout_vec(const void* src, int src_f32, float* dst, int R, int row_off, int geglu, hipStream_t s);

}  // namespace gdf
