// HBM-bound normalisation / elementwise kernels for gfx950: GroupNorm (+SiLU), LayerNorm, strided
// copies (hook stores), input packing, timestep embeddings, small-M linears, weight re-layout.
//
// Reference ops replaced (paths under /root/reference/feature/diffusers/models):
//   GroupNorm(32, C, eps=1e-5)+SiLU in ResnetBlock2D (resnet.py:267,281,325-326,359-361),
//   GroupNorm(32, C, eps=1e-6) in Transformer2DModel (transformers/transformer_2d.py:175-177,484),
//   LayerNorm in BasicTransformerBlock (attention.py:494-495,549,566), conv_norm_out+SiLU
//   (unet/unet_2d_condition.py:1304-1306), Timesteps/TimestepEmbedding/add_embedding
//   (unet/unet_2d_condition.py:910-934,968-984,1142-1162), time_emb_proj (resnet.py:343-346),
//   FeatureStore.store's clone+fp16 cast (components/feature_extractor.py:56-60).
// All are pure streaming kernels: 16-byte-per-lane coalesced accesses over NHWC / token-major rows.
#include "kernels.h"

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pixel rows per stage-1 block: sized so that the stage-1 grid has >= ~2048 workgroups (32x32 levels used to run
// 64 workgroups at 1 TB/s)
// (round 4: the cap was 256 rows — the VAE's 1024^2 maps then produced 4096 slabs per sample, and the finalize pass, one wave per
// (sample, group) walking 16384 partial pairs in a dependent double-precision chain, took longer than the statistics read itself:
// gn_stats 3.2 ms vs gn_apply 2.6 ms per 4-image sub-batch although it moves a third of the bytes)
static int gn_slab(int B, int HW) {
  long s = (long)B * HW / 2048;
  int slab = 16;
  while (slab < s && slab < 4096) slab *= 2;
  return slab;
}

// x_lo > 0: the fp16 source is a split (hi, lo) pair, lo stored x_lo elements after hi in the same row ("precise" plans, kernels.h)
__device__ __forceinline__ void load8(const half_t* x16, const float* x32, size_t idx, float v[8], int x_lo = 0) {
  if (x32) {
    const f32x4 a = *(const f32x4*)(x32 + idx), b = *(const f32x4*)(x32 + idx + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
    const f16x8 a = *(const f16x8*)(x16 + idx);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    if (x_lo > 0) {
      const f16x8 l = *(const f16x8*)(x16 + idx + x_lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)l[e];
    }
  }
}
// fp16 store of 8 values, optionally as a split (hi, lo) pair (lo at y + y_lo)
__device__ __forceinline__ void store8(half_t* y, const float t[8], int y_lo) {
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)t[e];
  *(f16x8*)y = o;
  if (y_lo > 0) {
    f16x8 l;
#pragma unroll
    for (int e = 0; e < 8; ++e) l[e] = (_Float16)(t[e] - (float)o[e]);
    *(f16x8*)(y + y_lo) = l;
  }
}

size_t gn_partial_floats(int B, int HW, int C) {
  const int slab = gn_slab(B, HW);
  const int nslab = (HW + slab - 1) / slab;
  return (size_t)B * nslab * C * 2;
}

// stage 1: per (sample, slab) block: per-channel sum / sum of squares over the slab's pixel rows
__global__ __launch_bounds__(256) void gn_partial_kernel(const half_t* x16, const float* x32, int ld, int HW, int C,
                                                         float* partial, int slab_rows, int x_lo) {
  extern __shared__ float red[];                  // [rgroups][C][2]
  const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int CH = C / 8;
  const int cht = CH < 256 ? CH : 256;            // chunk columns handled in parallel
  const int rgroups = 256 / cht;                  // row groups working in parallel
  const int tc = threadIdx.x % cht, tr = threadIdx.x / cht;
  const int r0 = slab * slab_rows, r1 = min(HW, r0 + slab_rows);
  for (int c = tc; c < CH; c += cht) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (tr < rgroups) {
      int r = r0 + tr;
      for (; r + 3 * rgroups < r1; r += 4 * rgroups) {          // four rows in flight per thread (long slabs: the VAE's 1024^2 maps)
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8(x16, x32, ((size_t)b * HW + r + u * rgroups) * ld + c * 8, v[u], x_lo);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[e] += v[u][e]; q[e] += v[u][e] * v[u][e]; }
      }
      for (; r < r1; r += rgroups) {
        float v[8];
        load8(x16, x32, ((size_t)b * HW + r) * ld + c * 8, v, x_lo);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(tr * C + c * 8 + e) * 2 + 0] = s[e];
        red[(tr * C + c * 8 + e) * 2 + 1] = q[e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += 256) {
    float a = 0.f;
    for (int g = 0; g < rgroups; ++g) a += red[g * C * 2 + i];
    partial[((size_t)b * nslab + slab) * C * 2 + i] = a;
  }
}

// stage 2: one block per (sample, group): combine slabs (in double), emit the per-channel affine table
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* partial, int nslab, int HW, int C, int G, float eps,
                                                          const float* gamma, const float* beta, float* ab) {
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, tid = threadIdx.x;
  const int cpg = C / G;
  __shared__ double red[8];
  double s = 0.0, q = 0.0;
  // 256 threads, fixed assignment of partials to threads and a fixed combine order: deterministic
  for (int i = tid; i < nslab * cpg; i += 256) {
    const int sl = i / cpg, c = g * cpg + (i - sl * cpg);
    const float2 pp = *(const float2*)(partial + (((size_t)b * nslab + sl) * C + c) * 2);
    s += (double)pp.x; q += (double)pp.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
  if (lane == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
  __syncthreads();
  s = red[0] + red[2] + red[4] + red[6]; q = red[1] + red[3] + red[5] + red[7];
  if (tid >= 64) return;
  const double n = (double)HW * cpg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
    const float a = rstd * gamma[c];
    ab[((size_t)b * C + c) * 2 + 0] = a;
    ab[((size_t)b * C + c) * 2 + 1] = beta[c] - (float)mean * a;
  }
}

hipError_t launch_gn_stats(const half_t* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps,
                           const float* gamma, const float* beta, float* partial, float* ab, hipStream_t s, int x_lo) {
  if (C % 8 || C % G) return hipErrorInvalidValue;
  const int slab = gn_slab(B, HW);
  const int nslab = (HW + slab - 1) / slab;
  const int CH = C / 8, cht = CH < 256 ? CH : 256, rgroups = 256 / cht;
  const size_t smem = (size_t)rgroups * C * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nslab, B), dim3(256), smem, s, x16, x32, ld, HW, C, partial, slab, x_lo);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, s, partial, nslab, HW, C, G, eps, gamma, beta, ab);
  return hipGetLastError();
}

// Many short slabs (a GEMM epilogue emits one per 64 rows: 16384 per sample at 1024^2) are first folded into <= GN_FOLD slabs per sample
// with fully coalesced row reads; gn_finalize_kernel's per-group gather over all of them took 46-68 us per GroupNorm (poorly coalesced
// 32-byte pieces, 128 workgroups), about a third of the statistics pass it replaces.
static constexpr int GN_FOLD = 128;
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* partial, int nslab, int per, int C2, float* out) {
  __shared__ double red[256 * 4];
  const int b = blockIdx.y, j = blockIdx.x, nout = gridDim.x;
  const int s0 = j * per, s1 = min(nslab, s0 + per);
  const int tpr = min(256, C2 / 4), rg = 256 / tpr;      // threads per slab row (4 floats each), row groups working in parallel
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  for (int c0 = 0; c0 < C2; c0 += 1024) {                // (C2 <= 1024 for every GroupNorm of the models here: one trip)
    const int c = c0 + tc * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (tr < rg && c < C2) {
      int sl = s0 + tr;
      for (; sl + 7 * rg < s1; sl += 8 * rg) {            // eight independent row loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(partial + ((size_t)b * nslab + sl + u * rg) * C2 + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += (double)v[u][0]; a1 += (double)v[u][1]; a2 += (double)v[u][2]; a3 += (double)v[u][3]; }
      }
      for (; sl < s1; sl += rg) {
        const f32x4 v = *(const f32x4*)(partial + ((size_t)b * nslab + sl) * C2 + c);
        a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
      }
    }
    red[threadIdx.x * 4 + 0] = a0; red[threadIdx.x * 4 + 1] = a1; red[threadIdx.x * 4 + 2] = a2; red[threadIdx.x * 4 + 3] = a3;
    __syncthreads();
    if (tr == 0 && c < C2) {                              // fixed combine order: deterministic
      for (int g = 1; g < rg; ++g) {
        a0 += red[(g * tpr + tc) * 4 + 0]; a1 += red[(g * tpr + tc) * 4 + 1]; a2 += red[(g * tpr + tc) * 4 + 2]; a3 += red[(g * tpr + tc) * 4 + 3];
      }
      *(f32x4*)(out + ((size_t)b * nout + j) * C2 + c) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
    }
    __syncthreads();
  }
}

size_t gn_fold_floats(int B, int nslab, int C) { return nslab > 2 * GN_FOLD ? (size_t)B * GN_FOLD * C * 2 : 0; }

// the fold pass alone: nslab slabs -> gn_fold_out_slabs(nslab) slabs in `fold` (same [slab][C][2] layout); false = shape not supported
int gn_fold_out_slabs(int nslab) {
  const int per = (nslab + GN_FOLD - 1) / GN_FOLD;
  return (nslab + per - 1) / per;
}
bool gn_fold_ok(int C) { return (2 * C) % 4 == 0 && (2 * C >= 1024 ? (2 * C) % 1024 == 0 : 256 % (2 * C / 4) == 0); }
hipError_t launch_gn_fold(const float* partial, int nslab, int B, int C, float* fold, hipStream_t s) {
  if (!gn_fold_ok(C) || nslab < 1) return hipErrorInvalidValue;
  const int per = (nslab + GN_FOLD - 1) / GN_FOLD, nout = (nslab + per - 1) / per;
  hipLaunchKernelGGL(gn_fold_kernel, dim3(nout, B), dim3(256), 0, s, partial, nslab, per, 2 * C, fold);
  return hipGetLastError();
}

hipError_t launch_gn_finalize(const float* partial, int nslab, int B, int HW, int C, int G, float eps, const float* gamma,
                              const float* beta, float* ab, float* fold, hipStream_t s) {
  if (C % G || nslab < 1 || (C & 1)) return hipErrorInvalidValue;
  if (fold && gn_fold_floats(B, nslab, C) && (2 * C) % 4 == 0 && (2 * C >= 1024 ? (2 * C) % 1024 == 0 : 256 % (2 * C / 4) == 0)) {
    const int per = (nslab + GN_FOLD - 1) / GN_FOLD, nout = (nslab + per - 1) / per;
    hipLaunchKernelGGL(gn_fold_kernel, dim3(nout, B), dim3(256), 0, s, partial, nslab, per, 2 * C, fold);
    partial = fold; nslab = nout;
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, s, partial, nslab, HW, C, G, eps, gamma, beta, ab);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* x16, const float* x32, int ld, int HW, int C,
                                                       const float* ab, int silu, half_t* y, int rows_per_block, int x_lo, int ldy,
                                                       int y_lo) {
  const int b = blockIdx.y;
  const int CH = C / 8;
  const int cht = CH < 256 ? CH : 256, rgroups = 256 / cht;
  const int tc = threadIdx.x % cht, tr = threadIdx.x / cht;
  if (tr >= rgroups) return;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  for (int c = tc; c < CH; c += cht) {
    float a[8], bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = ab[((size_t)b * C + c * 8 + e) * 2];
      bb[e] = ab[((size_t)b * C + c * 8 + e) * 2 + 1];
    }
    for (int r = r0 + tr; r < r1; r += rgroups) {
      float v[8];
      load8(x16, x32, ((size_t)b * HW + r) * ld + c * 8, v, x_lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = v[e] * a[e] + bb[e];
        if (silu) t = t / (1.0f + __expf(-t));
        v[e] = t;
      }
      store8(y + ((size_t)b * HW + r) * ldy + c * 8, v, y_lo);
    }
  }
}

hipError_t launch_gn_apply(const half_t* x16, const float* x32, int ld, int B, int HW, int C, const float* ab, int silu,
                           half_t* y, hipStream_t s, int x_lo, int ldy, int y_lo) {
  if (C % 8) return hipErrorInvalidValue;
  const int rpb = 64;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((HW + rpb - 1) / rpb, B), dim3(256), 0, s, x16, x32, ld, HW, C, ab, silu, y,
                     rpb, x_lo, ldy > 0 ? ldy : C, y_lo);
  return hipGetLastError();
}

// Round 6 (the "apply half" of north_star's ResBlock fusion in the form that does not touch the conv's A path): the
// statistics left by the producing conv's epilogue are finalised AND applied in ONE launch.  Workgroup (channel slab of whole groups,
// row block, sample): pass 1 combines the per-slab channel sums of its SC channels (nslab x SC float2, coalesced, double accumulation,
// fixed order: deterministic), the slab's groups become the affine table in LDS, pass 2 normalises (+SiLU) the block's rows.  Every row
// block of a channel slab repeats pass 1 (nslab x SC x 8 bytes, L2 resident after the first) — the price of not needing a grid-wide
// barrier; rows_per_block is chosen so that it stays below a quarter of the block's own traffic.
__global__ __launch_bounds__(256) void gn_finalize_apply_kernel(const float* partial, int nslab, const half_t* x16, int ld, int HW, int C, int G,
                                                                float eps, const float* gamma, const float* beta, int silu, half_t* y, int SC,
                                                                int rows_per_block, int ldy, int y_lo) {
  extern __shared__ double fa_red[];              // [RG][SC][2] doubles, then float ab[SC][2] behind them
  const int b = blockIdx.z, c0 = blockIdx.x * SC;
  const int RG = 256 / SC;                        // slab-row groups working in parallel (SC <= 256)
  const int tc = threadIdx.x % SC, tg = threadIdx.x / SC;
  double s = 0.0, q = 0.0;
  if (tg < RG) {
    for (int sl = tg; sl < nslab; sl += RG) {
      const float2 pp = *(const float2*)(partial + (((size_t)b * nslab + sl) * C + c0 + tc) * 2);
      s += (double)pp.x; q += (double)pp.y;
    }
    fa_red[(tg * SC + tc) * 2] = s; fa_red[(tg * SC + tc) * 2 + 1] = q;
  }
  __syncthreads();
  float* ab = (float*)(fa_red + (size_t)RG * SC * 2);
  const int cpg = C / G, ngs = SC / cpg;
  if (threadIdx.x < ngs) {
    double sum = 0.0, sq = 0.0;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c)
      for (int g = 0; g < RG; ++g) { sum += fa_red[(g * SC + c) * 2]; sq += fa_red[(g * SC + c) * 2 + 1]; }
    const double n = (double)HW * cpg;
    const double mean = sum / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
      const float a = rstd * gamma[c0 + c];
      ab[c * 2] = a;
      ab[c * 2 + 1] = beta[c0 + c] - (float)mean * a;
    }
  }
  __syncthreads();
  const int LPR = SC / 8, RPP = 256 / LPR;
  const int lc = threadIdx.x % LPR, tr = threadIdx.x / LPR;
  if (tr >= RPP) return;
  float a[8], bb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = ab[(lc * 8 + e) * 2]; bb[e] = ab[(lc * 8 + e) * 2 + 1]; }
  const size_t rowbase = (size_t)b * HW;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  int r = r0 + tr;
  for (; r + 3 * RPP < r1; r += 4 * RPP) {                      // four rows in flight per thread
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) load8(x16, nullptr, (rowbase + r + u * RPP) * ld + c0 + lc * 8, v[u], 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = v[u][e] * a[e] + bb[e];
        if (silu) t = t / (1.0f + __expf(-t));
        v[u][e] = t;
      }
      store8(y + (rowbase + r + u * RPP) * ldy + c0 + lc * 8, v[u], y_lo);
    }
  }
  for (; r < r1; r += RPP) {
    float v[8];
    load8(x16, nullptr, (rowbase + r) * ld + c0 + lc * 8, v, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] * a[e] + bb[e];
      if (silu) t = t / (1.0f + __expf(-t));
      v[e] = t;
    }
    store8(y + (rowbase + r) * ldy + c0 + lc * 8, v, y_lo);
  }
}

// channels per workgroup of the finalize + apply kernel: whole groups, whole 16-byte chunks — the LARGEST such slab <= 256 channels that divides C
// (C = 320 / 640 / 1280 -> 160: 320-byte row pieces; VAE C = 128 -> 128, 256 / 512 -> 256); 0 = not applicable
int gn_finalize_apply_slab(int C, int G) {
  if (C % 8 || C % G) return 0;
  const int cpg = C / G;
  int L = cpg;
  while (L % 8) L += cpg;                                      // lcm(cpg, 8)
  int best = 0;
  for (int SC = L; SC <= 256; SC += L)
    if (C % SC == 0) best = SC;
  return best >= 32 ? best : 0;
}

hipError_t launch_gn_finalize_apply(const float* partial, int nslab, const half_t* x16, int ld, int B, int HW, int C, int G, float eps,
                                    const float* gamma, const float* beta, int silu, half_t* y, hipStream_t s, int ldy, int y_lo) {
  const int SC = gn_finalize_apply_slab(C, G);
  if (!SC || nslab < 1) return hipErrorInvalidValue;
  const int RG = 256 / SC;
  // rows per block: the statistics pass (nslab x SC x 8 B) at most ~1/4 of the block's own 4 B/element, and >= 512 workgroups when the tensor allows
  long rpb = ((long)nslab * 8 * 4 + 3) / 4;                    // rows such that rows * SC * 4 B = 4 x nslab * SC * 8 B
  if (rpb < 256) rpb = 256;
  while (rpb > 256 && (long)B * (C / SC) * ((HW + rpb - 1) / rpb) < 512) rpb /= 2;
  if (rpb > HW) rpb = HW;
  const size_t smem = (size_t)RG * SC * 2 * sizeof(double) + (size_t)SC * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_finalize_apply_kernel, dim3(C / SC, (unsigned)((HW + rpb - 1) / rpb), B), dim3(256), smem, s, partial, nslab, x16, ld, HW, C, G,
                     eps, gamma, beta, silu, y, SC, (int)rpb, ldy > 0 ? ldy : C, y_lo);
  return hipGetLastError();
}

// Small feature maps (<= 32 x 32 pixels): the three launches above are latency bound (1280 channels at 32^2, batch 16: 75 us for
// 126 MB of traffic).  One workgroup per (sample, slab of whole groups) instead: pass 1 accumulates per-channel sums over the
// slab's HW x SC block, the slab's group statistics are combined in double in LDS, pass 2 re-reads the block (L2 / MALL
// resident: it was just read), applies the affine (+SiLU) and writes fp16.  One launch, same 6 B/element.
__global__ __launch_bounds__(256) void gn_fused_kernel(const half_t* x16, const float* x32, int ld, int HW, int C, int G, float eps,
                                                       const float* gamma, const float* beta, int silu, half_t* y, int SC, int x_lo,
                                                       int ldy, int y_lo) {
  extern __shared__ float red[];                  // [RPP][SC][2] floats, then reused: double chan[SC][2], float ab[SC][2]
  const int b = blockIdx.y, c0 = blockIdx.x * SC;
  const int LPR = SC / 8, RPP = 256 / LPR;
  const int lc = threadIdx.x % LPR, tr = threadIdx.x / LPR;
  const bool act = tr < RPP;
  const size_t rowbase = (size_t)b * HW;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (act) {
    for (int r = tr; r < HW; r += RPP) {
      float v[8];
      load8(x16, x32, (rowbase + r) * ld + c0 + lc * 8, v, x_lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(tr * SC + lc * 8 + e) * 2 + 0] = s[e];
      red[(tr * SC + lc * 8 + e) * 2 + 1] = q[e];
    }
  }
  __syncthreads();
  double* chan = (double*)(red + (size_t)RPP * SC * 2);        // [SC][2]
  float* ab = (float*)(chan + SC * 2);                         // [SC][2]
  for (int i = threadIdx.x; i < SC * 2; i += 256) {
    double a = 0.0;
    for (int g = 0; g < RPP; ++g) a += (double)red[g * SC * 2 + i];
    chan[i] = a;
  }
  __syncthreads();
  const int cpg = C / G, ngs = SC / cpg;                       // groups in this slab
  if (threadIdx.x < ngs) {
    double sum = 0.0, sq = 0.0;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { sum += chan[c * 2]; sq += chan[c * 2 + 1]; }
    const double n = (double)HW * cpg;
    const double mean = sum / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
      const float a = rstd * gamma[c0 + c];
      ab[c * 2] = a;
      ab[c * 2 + 1] = beta[c0 + c] - (float)mean * a;
    }
  }
  __syncthreads();
  if (!act) return;
  float a[8], bb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = ab[(lc * 8 + e) * 2]; bb[e] = ab[(lc * 8 + e) * 2 + 1]; }
  for (int r = tr; r < HW; r += RPP) {
    float v[8];
    load8(x16, x32, (rowbase + r) * ld + c0 + lc * 8, v, x_lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] * a[e] + bb[e];
      if (silu) t = t / (1.0f + __expf(-t));
      v[e] = t;
    }
    store8(y + (rowbase + r) * ldy + c0 + lc * 8, v, y_lo);
  }
}

// channels per workgroup of the fused kernel: whole groups, whole 16-byte chunks, >= 64 channels; 0 = use the 3-launch path
int gn_fused_slab(int B, int HW, int C, int G) {
  if (C % 8 || C % G || HW > 1024) return 0;
  const int cpg = C / G;
  int L = cpg;
  while (L % 8) L += cpg;                                      // lcm(cpg, 8)
  int SC = L;
  while (SC < 64 && C % (SC * 2) == 0) SC *= 2;
  if (C % SC || SC > 256) return 0;
  // fewer than 64 workgroups cannot stream a LARGE tensor at full bandwidth; a small one (small batches: <= 8 MiB) is latency bound
  // and one launch beats the three of the statistics + apply path (batch 2: 44 GroupNorms, 7.5 % of the step in gn_stats alone)
  if ((long)B * (C / SC) < 64 && (long)B * HW * C * 2 > (8L << 20)) return 0;
  return SC;
}

hipError_t launch_gn_fused(const half_t* x16, const float* x32, int ld, int B, int HW, int C, int G, float eps, const float* gamma,
                           const float* beta, int silu, half_t* y, hipStream_t s, int x_lo, int ldy, int y_lo) {
  const int SC = gn_fused_slab(B, HW, C, G);
  if (!SC) return hipErrorInvalidValue;
  const int LPR = SC / 8, RPP = 256 / LPR;
  const size_t smem = (size_t)RPP * SC * 2 * 4 + (size_t)SC * 2 * 8 + (size_t)SC * 2 * 4;
  hipLaunchKernelGGL(gn_fused_kernel, dim3(C / SC, B), dim3(256), smem, s, x16, x32, ld, HW, C, G, eps, gamma, beta, silu, y, SC, x_lo,
                     ldy > 0 ? ldy : C, y_lo);
  return hipGetLastError();
}

// LayerNorm: one wave per row, row kept in registers (C <= 64*8*MAXC), exact two-pass statistics.
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* x16, const float* x32, int ld, int R, int C,
                                                        float eps, const float* gamma, const float* beta, half_t* y, int ldy, int y_lo) {
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
      load8(x16, x32, (size_t)row * ld + c * 8, v[i]);
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
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < CH) {
      const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + c * 8), b1 = *(const f32x4*)(beta + c * 8 + 4);
      float t[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
        t[4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
      }
      store8(y + (size_t)row * ldy + c * 8, t, y_lo);
    }
  }
}

hipError_t launch_layernorm(const half_t* x16, const float* x32, int ld, int R, int C, float eps, const float* gamma,
                            const float* beta, half_t* y, hipStream_t s, int ldy, int y_lo) {
  if (C % 8 || C > 64 * 8 * 4) return hipErrorInvalidValue;
  const int CH = C / 8;
  if (ldy <= 0) ldy = C;
  dim3 grid((R + 3) / 4), blk(256);
  if (CH <= 64) hipLaunchKernelGGL(layernorm_kernel<1>, grid, blk, 0, s, x16, x32, ld, R, C, eps, gamma, beta, y, ldy, y_lo);
  else if (CH <= 128) hipLaunchKernelGGL(layernorm_kernel<2>, grid, blk, 0, s, x16, x32, ld, R, C, eps, gamma, beta, y, ldy, y_lo);
  else hipLaunchKernelGGL(layernorm_kernel<4>, grid, blk, 0, s, x16, x32, ld, R, C, eps, gamma, beta, y, ldy, y_lo);
  return hipGetLastError();
}

// strided copy + cast (coalesced 16-B stores): the hook write when the producer cannot store directly
// CVT = false: fp16 / fp32 source, plain casts (every UNet hook).  CVT = true (MMDiT path): the 16-bit source may be bf16
// (`bf`) and the fp16 result saturates at +-65504 instead of overflowing to inf (`sat`).
template <bool CVT>
__global__ __launch_bounds__(256) void copy2d_kernel(const half_t* s16, const float* s32, int lds_, half_t* dst, int ldd,
                                                     long R, int C, int bf, int sat, int s_lo, float scale) {
  if ((C & 7) == 0 && (lds_ & 7) == 0 && (ldd & 7) == 0) {
    const int CH = C / 8;
    const long total = R * CH;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long r = i / CH;
      const int c = (int)(i - r * CH) * 8;
      float v[8];
      if (CVT && !s32 && (bf || s_lo > 0)) {
        const f16x8 raw = *(const f16x8*)(s16 + (size_t)r * lds_ + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e16_to_f32(raw[e], bf);
        if (s_lo > 0) {                                  // split pair source: hi + lo
          const f16x8 rl = *(const f16x8*)(s16 + (size_t)r * lds_ + c + s_lo);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += e16_to_f32(rl[e], bf);
        }
      } else {
        load8(s16, s32, (size_t)r * lds_ + c, v);
      }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (CVT && sat) ? f32_to_f16_sat(CVT ? v[e] * scale : v[e]) : (_Float16)(CVT ? v[e] * scale : v[e]);
      *(f16x8*)(dst + (size_t)r * ldd + c) = o;
    }
  } else {
    const long total = R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long r = i / C;
      const int c = (int)(i - r * C);
      float v = s32 ? s32[(size_t)r * lds_ + c] : e16_to_f32(s16[(size_t)r * lds_ + c], CVT && bf);
      if (CVT && s_lo > 0 && !s32) v += e16_to_f32(s16[(size_t)r * lds_ + c + s_lo], bf);
      if (CVT) v *= scale;
      dst[(size_t)r * ldd + c] = (CVT && sat) ? f32_to_f16_sat(v) : (_Float16)v;
    }
  }
}

hipError_t launch_copy2d(const half_t* s16, const float* s32, int lds_, half_t* dst, int ldd, int R, int C,
                         hipStream_t s, int src_bf16, int sat, int s_lo, float scale) {
  const long work = (long)R * ((C + 7) / 8);
  long blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (src_bf16 || sat || s_lo > 0 || scale != 1.0f)
    hipLaunchKernelGGL(copy2d_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, s16, s32, lds_, dst, ldd, (long)R, C, src_bf16, sat, s_lo, scale);
  else
    hipLaunchKernelGGL(copy2d_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, s16, s32, lds_, dst, ldd, (long)R, C, 0, 0, 0, 1.0f);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void pack_latents_kernel(const half_t* x, int Cin, int HW, long total, half_t* nhwc8,
                                                           half_t* hook) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW;
    const int pix = (int)(i - b * HW);
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < Cin && c < 8; ++c) o[c] = x[((size_t)b * Cin + c) * HW + pix];
    *(f16x8*)(nhwc8 + (size_t)i * 8) = o;
    if (hook)
      for (int c = 0; c < Cin; ++c) hook[(size_t)i * Cin + c] = o[c];
  }
}

hipError_t launch_pack_latents(const half_t* x, int B, int Cin, int H, int W, half_t* nhwc8, half_t* hook_nhwc,
                               hipStream_t s) {
  if (Cin > 8) return hipErrorInvalidValue;
  const long total = (long)B * H * W;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_latents_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, Cin, H * W, total, nhwc8,
                     hook_nhwc);
  return hipGetLastError();
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void sinusoid_kernel(const float* t, int n_per_row, int dim, float* out, int ldo, int col_off, int round_f16,
                                int total, float tscale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = dim / 2;
  const int j = i % dim;
  const int ti = (i / dim) % n_per_row;
  const int b = i / (dim * n_per_row);
  const int f = j < half ? j : j - half;
  const float freq = expf(-9.210340371976184f * (float)f / (float)half);   // ln(10000)
  const float arg = t[b * n_per_row + ti] * tscale * freq;
  float v = j < half ? cosf(arg) : sinf(arg);
  if (round_f16) v = (float)(_Float16)v;
  out[(size_t)b * ldo + col_off + ti * dim + j] = v;
}

hipError_t launch_sinusoid(const float* t, int B, int n_per_row, int dim, float* out, int ldo, int col_off,
                           int round_f16, hipStream_t s, float tscale) {
  const int total = B * n_per_row * dim;
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, n_per_row, dim, out, ldo, col_off,
                     round_f16, total, tscale);
  return hipGetLastError();
}

__global__ void widen_kernel(const half_t* x, int n, float* out, int ldo, int col_off, int total, int bf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / n, j = i - b * n;
  out[(size_t)b * ldo + col_off + j] = e16_to_f32(x[i], bf);
}

hipError_t launch_widen(const half_t* x, int B, int n, float* out, int ldo, int col_off, hipStream_t s, int src_bf16) {
  const int total = B * n;
  hipLaunchKernelGGL(widen_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, n, out, ldo, col_off, total, src_bf16);
  return hipGetLastError();
}

// small-M linear on fp32 vectors with fp16 weights: one wave per output column, 8 rows at a time
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, int ldx, int M, int K, const half_t* Wt,
                                                           const float* bias, int N, int silu_in, int accumulate,
                                                           float* out, int ldo, int wbf) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const half_t* w = Wt + (size_t)n * K;
  for (int m0 = 0; m0 < M; m0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      const f16x8 wv = *(const f16x8*)(w + k);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (m0 + i < M) {
          const float* xp = x + (size_t)(m0 + i) * ldx + k;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float xv = xp[e];
            if (silu_in) xv = xv / (1.0f + expf(-xv));
            acc[i] = __builtin_fmaf(xv, e16_to_f32(wv[e], wbf), acc[i]);     // (explicit: the same rounding for every row, see the wide kernel)
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m0 + i < M) {
          float v = acc[i] + (bias ? bias[n] : 0.f);
          float* op = out + (size_t)(m0 + i) * ldo + n;
          if (accumulate) v += *op;
          *op = v;
        }
    }
  }
}

// Wide version (N large: the stacked adaLN modulation linear of the MMDiT, N = 1.06 M columns, 6.5 GB of weights):
// HBM-streaming bound.  x (<= 8 rows x K fp32, activation already applied) is staged ONCE per workgroup in LDS instead
// of being re-read from L2 by every wave (the column-per-wave kernel above moved 104 GB through L2 for this op: 43 ms);
// every wave owns 4 output columns per step (4 independent 16-byte weight streams in flight per lane).
__global__ __launch_bounds__(256) void small_linear_wide_kernel(const float* x, int ldx, int M, int K, const half_t* Wt,
                                                                const float* bias, int N, int silu_in, int accumulate,
                                                                float* out, int ldo, int cols_per_block, int wbf) {
  extern __shared__ float xs[];                       // [8][K]
  for (int i = threadIdx.x; i < 8 * K; i += 256) {
    const int m = i / K, k = i - m * K;
    float v = (m < M) ? x[(size_t)m * ldx + k] : 0.f;
    if (silu_in) v = v / (1.0f + expf(-v));
    xs[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_beg = blockIdx.x * cols_per_block, n_end = min(N, n_beg + cols_per_block);
  for (int n0 = n_beg + wave * 4; n0 < n_end; n0 += 16) {
    float acc[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[c][m] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      f16x8 wv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = min(n0 + c, N - 1);
        wv[c] = *(const f16x8*)(Wt + (size_t)n * K + k);
      }
      float wf[4][8];                                   // weights widened once per step (fp16 or bf16 storage)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[c][e] = e16_to_f32(wv[c][e], wbf);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const f32x4 a = *(const f32x4*)(xs + m * K + k), b = *(const f32x4*)(xs + m * K + k + 4);
        // explicit fused multiply-adds in ONE fixed order (k ascending): `acc += a*w + b*w2` left the contraction to the compiler, which paired
        // rows into packed instructions and rounded rows {0, 3} differently from rows {1, 2} — identical samples of one batch then got
        // different time embeddings, the source of every batch-position difference of the UNet (tools/op_batch_position.py, round 5)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float t = acc[c][m];
#pragma unroll
          for (int e = 0; e < 4; ++e) t = __builtin_fmaf(a[e], wf[c][e], t);
#pragma unroll
          for (int e = 0; e < 4; ++e) t = __builtin_fmaf(b[e], wf[c][4 + e], t);
          acc[c][m] = t;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[c][m] += __shfl_xor(acc[c][m], off);
      }
    if (lane < 32) {                                  // lane -> (column c, row m)
      const int c = lane >> 3, m = lane & 7;
      const int n = n0 + c;
      if (n < n_end && m < M) {
        float v = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int mm = 0; mm < 8; ++mm) if (cc == c && mm == m) v = acc[cc][mm];
        v += bias ? bias[n] : 0.f;
        float* op = out + (size_t)m * ldo + n;
        if (accumulate) v += *op;
        *op = v;
      }
    }
  }
}

hipError_t launch_small_linear(const float* x, int ldx, int M, int K, const half_t* Wt, const float* bias, int N,
                               int silu_in, int accumulate, float* out, int ldo, hipStream_t s, int w_bf16) {
  if (K % 8) return hipErrorInvalidValue;
  // The LDS-staged kernel (activation applied once per workgroup, 4 weight streams per lane) also serves the mid-size stacked
  // linears (all time_emb_proj of a UNet in one matrix, N ~ 18 k: the column-per-wave kernel re-evaluated SiLU per column,
  // 0.54 ms) and more than 8 rows (one launch per 8 rows).
  if (N >= 1024 && (size_t)K * 32 <= 128 * 1024) {
    static std::atomic<uint64_t> attr_mask{0};
    {
      const hipError_t e = ensure_dyn_smem(attr_mask, (const void*)small_linear_wide_kernel, 128 * 1024);
      if (e != hipSuccess) return e;
    }
    // 512 columns per workgroup when N is huge (x staging amortised); fewer for mid-size N so that >= ~512 workgroups exist
    int cpb = 512;
    while (cpb > 16 && (N + cpb - 1) / cpb < 512) cpb >>= 1;
    for (int m0 = 0; m0 < M; m0 += 8)
      hipLaunchKernelGGL(small_linear_wide_kernel, dim3((N + cpb - 1) / cpb), dim3(256), (size_t)K * 32, s, x + (size_t)m0 * ldx, ldx,
                         (M - m0 < 8 ? M - m0 : 8), K, Wt, bias, N, silu_in, accumulate, out + (size_t)m0 * ldo, ldo, cpb, w_bf16);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(small_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, ldx, M, K, Wt, bias, N, silu_in,
                     accumulate, out, ldo, w_bf16);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// weight re-layout
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ldsrc(const void* src, int dt, size_t i) {      // dt: 0 fp16, 1 fp32, 2 bf16
  return dt == 1 ? ((const float*)src)[i] : e16_to_f32(((const half_t*)src)[i], dt == 2);
}
__device__ __forceinline__ int geglu_row(int r, int half, int g) {
  // source row r of the [2*half][K] GEGLU projection -> GEMM row so that every 2g-column group is [g h | g gate]
  const int is_gate = r >= half;
  const int rr = is_gate ? r - half : r;
  return (rr / g) * 2 * g + (is_gate ? g : 0) + (rr % g);
}

// VAE decoder head (`vae-out`, reference diffusion_feature.py:477-485): z = (c_sample * latents + c_eps * noise_pred) * inv_scaling
// (scheduler.step on the un-scaled latents, then `/ vae.config.scaling_factor`), y = post_quant_conv(z) (1x1, [L][L] fp16 weights,
// fp32 bias; wq == NULL: identity) -> NHWC fp16 padded to 8 channels (the conv_in operand).  latents / noise_pred: NCHW fp16.
__global__ __launch_bounds__(256) void vae_dec_prepare_kernel(const half_t* lat, const half_t* eps, int HW, int L, long total, float ca,
                                                              float cb, float inv_sf, const half_t* wq, const float* bq, half_t* nhwc8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW;
    const int pix = (int)(i - b * HW);
    float z[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      z[c] = 0.f;
      if (c < L) {
        const size_t idx = ((size_t)b * L + c) * HW + pix;
        z[c] = (ca * (float)lat[idx] + (eps ? cb * (float)eps[idx] : 0.f)) * inv_sf;
      }
    }
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int oc = 0; oc < 8; ++oc) {
      if (oc < L) {
        float y = z[oc];
        if (wq) {
          y = bq ? bq[oc] : 0.f;
          for (int c = 0; c < L; ++c) y += (float)wq[oc * L + c] * z[c];
        }
        o[oc] = (_Float16)y;
      }
    }
    *(f16x8*)(nhwc8 + (size_t)i * 8) = o;
  }
}
hipError_t launch_vae_dec_prepare(const half_t* lat, const half_t* eps, int B, int HW, int L, float ca, float cb, float inv_sf,
                                  const half_t* wq, const float* bq, half_t* nhwc8, hipStream_t s) {
  if (L < 1 || L > 8) return hipErrorInvalidValue;
  const long total = (long)B * HW;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(vae_dec_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lat, eps, HW, L, total, ca, cb, inv_sf, wq, bq, nhwc8);
  return hipGetLastError();
}

// cblk == 0: dst[o][t][i];  cblk > 0 (3x3 convs, cblk = 64 = one K-tile): dst[o][i / cblk][t][i % cblk] — channel-block-major with
// the filter taps INNERMOST, so that the implicit GEMM walks the nine shifted windows of one 64-channel slab of the input in nine
// CONSECUTIVE K-tiles (they overlap in all but one image row / column: the re-reads hit the XCD's L2 instead of the fabric)
__global__ void relayout_conv_kernel(const void* src, int f32, half_t* dst, int O, int I, int T, int ipad, int tpad, int cblk,
                                     long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ci, t, o;
    if (cblk > 0) {
      const int cl = (int)(i % cblk);
      t = (int)((i / cblk) % tpad);
      const int cb = (int)((i / ((long)cblk * tpad)) % (ipad / cblk));
      o = (int)(i / ((long)ipad * tpad));
      ci = cb * cblk + cl;
    } else {
      ci = (int)(i % ipad);
      t = (int)((i / ipad) % tpad);
      o = (int)(i / ((long)ipad * tpad));
    }
    float v = 0.f;
    if (ci < I && t < T) v = ldsrc(src, f32, ((size_t)o * I + ci) * T + t);
    dst[i] = (_Float16)v;
  }
}
hipError_t launch_relayout_conv(const void* src, int src_f32, half_t* dst, int O, int I, int T, int ipad, int tpad,
                                hipStream_t s, int cblk) {
#if defined(GDF_CONV_TAP_MAJOR)
  cblk = 0;
#endif
  if (cblk > 0 && (ipad % cblk) != 0) return hipErrorInvalidValue;
  const long total = (long)O * ipad * tpad;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relayout_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, src_f32, dst, O, I, T, ipad,
                     tpad, cblk, total);
  return hipGetLastError();
}

__global__ void relayout_rows_kernel(const void* src, int f32, half_t* dst, int R, int K, int row_off, int geglu,
                                     long total, int dbf) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K);
    const int k = (int)(i - (long)r * K);
    const int dr = geglu ? geglu_row(r, R / 2, geglu) : r + row_off;
    dst[(size_t)dr * K + k] = f32_to_e16(ldsrc(src, f32, i), dbf);
  }
}
hipError_t launch_relayout_rows(const void* src, int src_f32, half_t* dst, int R, int K, int row_off, int geglu,
                                hipStream_t s, int dst_bf16) {
  const long total = (long)R * K;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relayout_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, src_f32, dst, R, K, row_off,
                     geglu, total, dst_bf16);
  return hipGetLastError();
}

__global__ void relayout_vec_kernel(const void* src, int f32, float* dst, int R, int row_off, int geglu) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int dr = geglu ? geglu_row(r, R / 2, geglu) : r + row_off;
  dst[dr] = ldsrc(src, f32, r);
}
hipError_t launch_relayout_vec(const void* src, int src_f32, float* dst, int R, int row_off, int geglu, hipStream_t s) {
  hipLaunchKernelGGL(relayout_vec_kernel, dim3((R + 255) / 256), dim3(256), 0, s, src, src_f32, dst, R, row_off, geglu);
  return hipGetLastError();
}

}  // namespace gdf
