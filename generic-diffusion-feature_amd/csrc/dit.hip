// HBM-bound kernels of the MMDiT (Flux) path for gfx950.
//
// Reference ops replaced (paths under /root/reference/feature/diffusers/models; the normalisation / embedding classes
// are un-vendored diffusers==0.32.2, imported at transformers/transformer_flux.py:34-38 and attention_processor.py:141,2331):
//   AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous `norm(x) * (1 + scale[:, None]) + shift[:, None]`
//   and the norm2 modulate of FluxTransformerBlock (transformer_flux.py:194-195, 214-215)      -> layernorm_mod_kernel
//   Attention.norm_q / norm_k / norm_added_q / norm_added_k (RMSNorm per head, attention_processor.py:2300-2303,
//   2321-2324) + apply_rotary_emb (attention_processor.py:2331-2335)                           -> qk_norm_rope_kernel
//   FluxPosEmbed (transformer_flux.py:498-499)                                                 -> rope_table_kernel
// All are streaming kernels: 16 bytes per lane, one pass over the data.
#include "kernels.h"

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per row, the row lives in registers (C <= 64 * 8 * MAXC), exact two-pass statistics
// ---- fp8 (OCP e4m3) row quantisation: q = fp8(v / s), s = the smallest power of two with |v| / s <= 448 (the e4m3 maximum) ----
__device__ __forceinline__ float fp8_row_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e; (void)frexpf(amax * (1.0f / 448.0f), &e);        // amax / 448 = m 2^e, m in [0.5, 1)  ->  2^e >= amax / 448
  return ldexpf(1.0f, e);
}
__device__ __forceinline__ uint2 pack_fp8x8(const float v[8], float inv) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
  return uint2{(unsigned)lo, (unsigned)hi};
}
// 16-bit rows [R][ld] (K columns used) -> fp8 [R][ldq] + scale[R]; one workgroup per row, two passes (the second read hits L2)
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const half_t* x, int ld, int K, int bf, unsigned char* q, int ldq, float* scale) {
  const size_t row = blockIdx.x;
  const half_t* xr = x + row * (size_t)ld;
  __shared__ float red[4];
  float amax = 0.f;
  for (int c = threadIdx.x * 8; c < K; c += 256 * 8) {
    const f16x8 a = *(const f16x8*)(xr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(e16_to_f32(a[e], bf)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = fp8_row_scale(amax), inv = 1.0f / sc;
  if (threadIdx.x == 0) scale[row] = sc;
  for (int c = threadIdx.x * 8; c < K; c += 256 * 8) {
    const f16x8 a = *(const f16x8*)(xr + c);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e16_to_f32(a[e], bf);
    *(uint2*)(q + row * (size_t)ldq + c) = pack_fp8x8(v, inv);
  }
}
hipError_t launch_quant_rows_fp8(const half_t* x, int ld, int R, int K, int bf16, unsigned char* q, int ldq, float* scale, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  if ((K & 7) || (ld & 7) || (ldq & 7)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)R), dim3(256), 0, s, x, ld, K, bf16, q, ldq, scale);
  return hipGetLastError();
}

template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const half_t* x16, const float* x32, int ld, int R, int C,
                                                            float eps, const float* scale, const float* shift, int ldm,
                                                            int rps, int seg_rows, int rps2, half_t* y, int bf, int ldy, int y_lo,
                                                            unsigned char* q8, int ldq8, float* q8_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int CH = C / 8;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < CH) {
      if (x32) {
        const f32x4 a = *(const f32x4*)(x32 + (size_t)row * ld + c * 8), b = *(const f32x4*)(x32 + (size_t)row * ld + c * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
      } else {
        const f16x8 a = *(const f16x8*)(x16 + (size_t)row * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (float)a[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < CH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)C + eps);
  const int smp = (seg_rows > 0 && row >= seg_rows) ? (row - seg_rows) / rps2 : row / rps;
  const float* sc = scale + (size_t)smp * ldm;
  const float* sh = shift + (size_t)smp * ldm;
  float amax = 0.f;                                     // fp8 output ('fp8-mx' plans): |max| of the modulated row
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < CH) {
      const f32x4 g0 = *(const f32x4*)(sc + c * 8), g1 = *(const f32x4*)(sc + c * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(sh + c * 8), b1 = *(const f32x4*)(sh + c * 8 + 4);
      f16x8 o;
      float t[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] = (v[i][e] - mean) * rstd * (1.0f + g0[e]) + b0[e];
        t[4 + e] = (v[i][4 + e] - mean) * rstd * (1.0f + g1[e]) + b1[e];
      }
      if (q8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = t[e]; amax = fmaxf(amax, fabsf(t[e])); }   // keep the modulated values for the fp8 pass
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f32_to_e16(t[e], bf);
      *(f16x8*)(y + (size_t)row * ldy + c * 8) = o;
      if (y_lo > 0) {                                    // split operand of the consumer GEMM ('bfloat16x2' plans): lo = e16(v - hi)
        f16x8 l;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = f32_to_e16(t[e] - e16_to_f32(o[e], bf), bf);
        *(f16x8*)(y + (size_t)row * ldy + c * 8 + y_lo) = l;
      }
    }
  }
  if (q8) {                                              // the same row as fp8 (e4m3) with one power-of-two scale (operand of an 'fp8-mx' GEMM)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    const float sc8 = fp8_row_scale(amax), inv = 1.0f / sc8;
    if (lane == 0) q8_scale[row] = sc8;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < CH) *(uint2*)(q8 + (size_t)row * ldq8 + c * 8) = pack_fp8x8(v[i], inv);
    }
  }
}

hipError_t launch_layernorm_mod(const half_t* x16, const float* x32, int ld, int R, int C, float eps, const float* scale,
                                const float* shift, int ldm, int rps, int seg_rows, int rps2, half_t* y, hipStream_t s,
                                int bf16, int ldy, int y_lo, unsigned char* q8, int ldq8, float* q8_scale) {
  if (C % 8 || C > 64 * 8 * 8 || (ldm & 3) || rps <= 0 || (y_lo & 7) || (ldy & 7) || (q8 && ((ldq8 & 7) || !q8_scale))) return hipErrorInvalidValue;
  if (ldy <= 0) ldy = C;
  if (R <= 0) return hipSuccess;
  const int CH = C / 8;
  dim3 grid((R + 3) / 4), blk(256);
#define GDF_LNM(N) hipLaunchKernelGGL(layernorm_mod_kernel<N>, grid, blk, 0, s, x16, x32, ld, R, C, eps, scale, shift, ldm, rps, seg_rows, rps2, y, bf16, ldy, y_lo, q8, ldq8, q8_scale)
  if (CH <= 64) GDF_LNM(1);
  else if (CH <= 128) GDF_LNM(2);
  else if (CH <= 256) GDF_LNM(4);
  else if (CH <= 384) GDF_LNM(6);
  else GDF_LNM(8);
#undef GDF_LNM
  return hipGetLastError();
}

// 16 lanes per (row, head): 8 halves (= 4 rotary pairs) per lane, D = 128.  q then k of the same (row, head).
// BF (bf16 container) is a COMPILE-TIME parameter of the body (round 6): with the runtime flag every input element went through
// `bf ? shift : cvt` — sixteen v_cndmask on a lane mask the wave held in VCC from its first instruction to its last — and with a second
// HSA queue active on the device (another host thread's stream) single elements of 16-lane groups came out as if the mask had been wrong
// for one instruction (tools/micro/op_race.py: this kernel beside a 128-row ring GEMM, 150-270 of 600 launches; 0 of 1000 without the
// per-element select; not a cvt -> select forwarding hazard: wait states between them change nothing).  The cause is below this library
// (wave state across queue switches is the suspect); the kernel no longer keeps anything in a mask register across its body.
template <bool BF>
__device__ __forceinline__ void qk_norm_rope_body(half_t* x, int ld, long R, int heads, int q_col, int k_col, const float* wq, const float* wk, float eps,
                                                  const float* cos_t, const float* sin_t, int pos0, int rps) {
  constexpr int D = 128;
  const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);     // (row, head) pair
  if (g >= R * heads) return;
  const int sub = threadIdx.x & 15;
  const long row = g / heads;
  const int head = (int)(g - row * heads);
  const int pos = pos0 + (int)(row % rps);
  const f32x4 c0 = *(const f32x4*)(cos_t + (size_t)pos * D + sub * 8), c1 = *(const f32x4*)(cos_t + (size_t)pos * D + sub * 8 + 4);
  const f32x4 s0 = *(const f32x4*)(sin_t + (size_t)pos * D + sub * 8), s1 = *(const f32x4*)(sin_t + (size_t)pos * D + sub * 8 + 4);
  float cs[8], sn[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { cs[e] = c0[e]; cs[4 + e] = c1[e]; sn[e] = s0[e]; sn[4 + e] = s1[e]; }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    half_t* px = x + (size_t)row * ld + (which ? k_col : q_col) + head * D + sub * 8;
    const float* w = (which ? wk : wq) + sub * 8;
    const f16x8 hv = *(const f16x8*)px;
    float v[8];
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = e16_to_f32(hv[e], BF ? 1 : 0); ss += v[e] * v[e]; }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);   // the 16 lanes of this (row, head)
    const float r = rsqrtf(ss / (float)D + eps);
    const f32x4 w0 = *(const f32x4*)w, w1 = *(const f32x4*)(w + 4);
#if defined(GDF_EXP_ROPE_FULLWAIT)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EXPERIMENT: every load of this iteration (x, cos, sin, gains) has landed before the math
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] *= r * w0[e]; v[4 + e] *= r * w1[e]; }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {       // x * cos + stack([-x_imag, x_real]) * sin
      o[e] = f32_to_e16(v[e] * cs[e] - v[e + 1] * sn[e], BF ? 1 : 0);
      o[e + 1] = f32_to_e16(v[e + 1] * cs[e + 1] + v[e] * sn[e + 1], BF ? 1 : 0);
    }
    *(f16x8*)px = o;
  }
}
template <bool BF>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(half_t* x, int ld, long R, int heads, int q_col, int k_col,
                                                           const float* wq, const float* wk, float eps,
                                                           const float* cos_t, const float* sin_t, int pos0, int rps) {
  qk_norm_rope_body<BF>(x, ld, R, heads, q_col, k_col, wq, wk, eps, cos_t, sin_t, pos0, rps);
}

hipError_t launch_qk_norm_rope(half_t* x, int ld, int R, int heads, int D, int q_col, int k_col, const float* wq,
                               const float* wk, float eps, const float* cos_t, const float* sin_t, int pos0, int rps,
                               hipStream_t s, int bf16) {
  if (D != 128 || (ld & 7) || (q_col & 7) || (k_col & 7) || rps <= 0) return hipErrorInvalidValue;
  if (R <= 0) return hipSuccess;
  const long groups = (long)R * heads;
  if (bf16)
    hipLaunchKernelGGL(qk_norm_rope_kernel<true>, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, s, x, ld, (long)R, heads, q_col,
                       k_col, wq, wk, eps, cos_t, sin_t, pos0, rps);
  else
    hipLaunchKernelGGL(qk_norm_rope_kernel<false>, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, s, x, ld, (long)R, heads, q_col,
                       k_col, wq, wk, eps, cos_t, sin_t, pos0, rps);
  return hipGetLastError();
}

struct RopeAxes { int n; int dim[4]; int off[4]; int D; };

__global__ void rope_table_kernel(const float* ids, int S, RopeAxes ax, double theta, float* cos_t, float* sin_t, int row0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = ax.D / 2;
  if (i >= S * half) return;
  const int r = i / half, pj = i - r * half;        // row, rotary pair index inside the row
  int a = 0;
  while (a + 1 < ax.n && 2 * pj >= ax.off[a + 1]) ++a;
  const int k = (2 * pj - ax.off[a]) / 2;           // pair index inside axis a
  const double freq = 1.0 / pow(theta, (double)(2 * k) / (double)ax.dim[a]);
  const double ang = (double)ids[(size_t)r * ax.n + a] * freq;
  const float c = (float)cos(ang), sn = (float)sin(ang);
  const size_t o = (size_t)(row0 + r) * ax.D + 2 * pj;
  cos_t[o] = c; cos_t[o + 1] = c;
  sin_t[o] = sn; sin_t[o + 1] = sn;
}

hipError_t launch_rope_table(const float* ids, int S, int n_axes, const int* axes_dim, double theta, float* cos_t,
                             float* sin_t, int row0, hipStream_t s) {
  if (n_axes < 1 || n_axes > 4) return hipErrorInvalidValue;
  RopeAxes ax{};
  ax.n = n_axes;
  int off = 0;
  for (int i = 0; i < n_axes; ++i) {
    if (axes_dim[i] % 2) return hipErrorInvalidValue;
    ax.dim[i] = axes_dim[i]; ax.off[i] = off; off += axes_dim[i];
  }
  ax.D = off;
  if (S <= 0) return hipSuccess;
  const int total = S * (off / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, s, ids, S, ax, theta, cos_t, sin_t, row0);
  return hipGetLastError();
}

// one workgroup per row; the row is held in registers as packed halves (n <= 256 * 8 * 8 = 16384 per pass, looped beyond)
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* x, int ld, int n, float scale) {
  __shared__ float red[8];
  half_t* row = x + (size_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float sl2 = scale * 1.44269504088896340736f;
  constexpr int MAXV = 8;                                // f16x8 vectors per thread kept in registers
  f16x8 v[MAXV];
  const int nvec = n / 8;                                // host guarantees n % 8 == 0 and n <= 256 * 8 * MAXV
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * 256;
    if (c < nvec) {
      v[i] = *(const f16x8*)(row + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[i][e]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;
  float sum = 0.f;
  float ev[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * 256;
    if (c < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ev[i][e] = __builtin_amdgcn_exp2f((float)v[i][e] * sl2 - mx); sum += ev[i][e]; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * 256;
    if (c < nvec) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(ev[i][e] * inv);
      *(f16x8*)(row + c * 8) = o;
    }
  }
}

hipError_t launch_softmax_rows(half_t* x, int ld, int R, int n, float scale, hipStream_t s) {
  if ((n & 7) || (ld & 7) || n > 256 * 8 * 8) return hipErrorInvalidValue;
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(R), dim3(256), 0, s, x, ld, n, scale);
  return hipGetLastError();
}

__global__ void vae_finish_kernel(const float* h, int HW, int L, const half_t* wq, const float* bq, const half_t* eps,
                                  const half_t* noise, float scaling, float na, float nb, float in_scale, half_t* out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (b, pixel)
  if (i >= total) return;
  const long b = i / HW;
  const int pix = (int)(i - b * HW);
  float m[16], hv[16];
  const int L2 = 2 * L;
  for (int c = 0; c < L2; ++c) hv[c] = h[(size_t)i * L2 + c];
  for (int o = 0; o < L2; ++o) {
    if (wq) {
      float a = bq[o];
      for (int c = 0; c < L2; ++c) a += (float)wq[o * L2 + c] * hv[c];
      m[o] = a;
    } else {
      m[o] = hv[o];
    }
  }
  for (int c = 0; c < L; ++c) {
    const size_t oi = ((size_t)b * L + c) * HW + pix;
    float z = m[c];
    if (eps) z += expf(0.5f * fminf(fmaxf(m[L + c], -30.0f), 20.0f)) * (float)eps[oi];
    float lat = scaling * z;
    if (noise) lat = na * lat + nb * (float)noise[oi];
    out[oi] = (_Float16)(in_scale * lat);
  }
}

hipError_t launch_vae_finish(const float* h, int B, int HW, int L, const half_t* wq, const float* bq, const half_t* eps,
                             const half_t* noise, float scaling, float noise_a, float noise_b, float in_scale, half_t* out,
                             hipStream_t s) {
  if (L < 1 || L > 8) return hipErrorInvalidValue;
  const long total = (long)B * HW;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(vae_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h, HW, L, wq, bq, eps, noise,
                     scaling, noise_a, noise_b, in_scale, out, total);
  return hipGetLastError();
}

__global__ void sincos_pos_embed_kernel(float* out, int C, int gh, int gw, float sh, float sw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = gh * gw;
  if (i >= S * C) return;
  const int tok = i / C, c = i - tok * C;
  // diffusers: grid = np.meshgrid(grid_w, grid_h) reshaped [2,1,gw,gh] and flattened; grid[0] (x coordinates) feeds the
  // first half of the channels.  Flat index m of the (gw, gh)-shaped view == m of the (gh, gw) meshgrid array.
  const int y = tok / gw, x = tok - y * gw;
  const int half = C / 2, quarter = C / 4;
  const int cc = c < half ? c : c - half;
  const double pos = c < half ? (double)((float)x * sw) : (double)((float)y * sh);
  const int k = cc < quarter ? cc : cc - quarter;
  const double omega = 1.0 / pow(10000.0, (double)k / (double)quarter);
  const double a = pos * omega;
  out[i] = (float)(cc < quarter ? sin(a) : cos(a));
}

hipError_t launch_sincos_pos_embed(float* out, int C, int gh, int gw, int base_size, float interpolation_scale, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  // grid_h = arange(gh) / (gh / base_size) / interpolation_scale (float32 arithmetic in numpy)
  const float sh = 1.0f / ((float)gh / (float)base_size) / interpolation_scale;
  const float sw = 1.0f / ((float)gw / (float)base_size) / interpolation_scale;
  const int total = gh * gw * C;
  hipLaunchKernelGGL(sincos_pos_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, out, C, gh, gw, sh, sw);
  return hipGetLastError();
}

__global__ void add_table_kernel(const float* table, const float* vec, int ldvec, int period, long n, float* out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = blockIdx.y;
  out[(size_t)b * ldo + i] = table[i] + vec[(size_t)b * ldvec + (int)(i % period)];
}

hipError_t launch_add_table(const float* table, const float* vec, int ldvec, int period, int B, long n, float* out, long ldo,
                            hipStream_t s) {
  if (n <= 0 || B <= 0) return hipSuccess;
  hipLaunchKernelGGL(add_table_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, table, vec, ldvec, period, n, out, ldo);
  return hipGetLastError();
}

__global__ void patchify_kernel(const half_t* x, int Cin, int H, int W, int p, int kpad, half_t* out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (row, col)
  if (i >= total) return;
  const long row = i / kpad;
  const int col = (int)(i - row * kpad);
  const int gh = H / p, gw = W / p;
  const long b = row / (gh * gw);
  const int t = (int)(row - b * gh * gw), ty = t / gw, tx = t - ty * gw;
  _Float16 v = (_Float16)0.f;
  if (col < Cin * p * p) {
    const int c = col / (p * p), r = col - c * p * p, py = r / p, px = r - py * p;
    v = x[(((size_t)b * Cin + c) * H + (ty * p + py)) * W + tx * p + px];
  }
  out[i] = v;
}

hipError_t launch_patchify(const half_t* x, int B, int Cin, int H, int W, int p, int kpad, half_t* out, hipStream_t s) {
  if (p < 1 || H % p || W % p || Cin * p * p > kpad) return hipErrorInvalidValue;
  const long total = (long)B * (H / p) * (W / p) * kpad;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, Cin, H, W, p, kpad, out, total);
  return hipGetLastError();
}

__global__ void unpatchify_kernel(const half_t* x, int Cout, int gh, int gw, int p, half_t* out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // output element (b, c, y, x)
  if (i >= total) return;
  const int W = gw * p, H = gh * p;
  const int xx = (int)(i % W);
  const int yy = (int)((i / W) % H);
  const int c = (int)((i / ((long)W * H)) % Cout);
  const long b = i / ((long)W * H * Cout);
  const int ty = yy / p, py = yy - ty * p, tx = xx / p, px = xx - tx * p;
  out[i] = x[((size_t)b * gh * gw + (size_t)ty * gw + tx) * (p * p * Cout) + (py * p + px) * Cout + c];
}

hipError_t launch_unpatchify(const half_t* x, int B, int Cout, int gh, int gw, int p, half_t* out, hipStream_t s) {
  const long total = (long)B * Cout * gh * p * gw * p;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, Cout, gh, gw, p, out, total);
  return hipGetLastError();
}

__global__ void relayout_rows_padk_kernel(const void* src, int f32, half_t* dst, int ksrc, int kdst, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long r = i / kdst;
  const int k = (int)(i - r * kdst);
  float v = 0.f;
  if (k < ksrc) v = f32 ? ((const float*)src)[r * ksrc + k] : (float)((const half_t*)src)[r * ksrc + k];
  dst[i] = (_Float16)v;
}

hipError_t launch_relayout_rows_padk(const void* src, int src_f32, half_t* dst, int R, int ksrc, int kdst, hipStream_t s) {
  const long total = (long)R * kdst;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(relayout_rows_padk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_f32, dst, ksrc,
                     kdst, total);
  return hipGetLastError();
}

__global__ void silu_vec_kernel(const float* x, float* out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i]; out[i] = v / (1.0f + expf(-v)); }
}

hipError_t launch_silu_vec(const float* x, float* out, long n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(silu_vec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n);
  return hipGetLastError();
}

}  // namespace gdf
