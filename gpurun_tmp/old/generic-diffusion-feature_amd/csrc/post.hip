// Post-processing kernels of the output stages that sit right after the hot path, gfx950.
// All three are HBM-bound byte movers over hook tensors: 16-byte lanes where the layout allows, LDS only to turn the
// channels-last hook layout into the NCHW rows the consumers store.
//
// Reference ops replaced (paths under /root/reference):
//   extract_feature.py:113-125   `--aggregate_output`: F.interpolate(v, max_hw) (nearest) of every layer + torch.cat(dim=1)
//                                                                                                   -> resize_concat_kernel
//   feature/components/feature_extractor.py:51-53   `feature_resize`: F.adaptive_avg_pool2d(feat, (H/r, W/r))
//                                                                                                   -> avg_pool_kernel
//   feature/components/attention.py:238-244, 141-161 + feature/diffusion_feature.py:492-500   aggregated `attn` feature:
//   head mean, mean over the layers of one (category, size) group ('b (h w) c -> b c h w'), nearest resize to img/8, concat
//                                                                            -> maps_mean_kernel + resize_concat_kernel
#include "kernels.h"

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PyTorch `nearest`: src = min(floor(dst * (in / out)), in - 1) with the scale held in fp32
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

// One workgroup = one (sample b, output row y, 64-channel chunk of one source layer).  The source row is read once,
// channel-fastest (128 B per pixel when sc == 1: the hook layout), transposed through LDS, and written x-fastest.
//   src: logical (B, C, H, W) with element strides (sb, sc, sy, sx), fp16 or fp32;  out: (B, Ctot, S, S) fp16 contiguous,
//   this layer occupying channels [coff, coff + C)
__global__ __launch_bounds__(256) void resize_concat_kernel(const half_t* s16, const float* s32, long sb, long sc, long sy, long sx,
                                                            int C, int H, int W, half_t* out, int Ctot, int coff, int S) {
  __shared__ _Float16 tile[128 * 66];                 // [xs][64 ch + 2 pad]: conflict-free x-fastest reads
  const int y = blockIdx.x % S, b = blockIdx.x / S;
  const int c0 = blockIdx.y * 64;
  const int nc = min(64, C - c0);
  const float fy = (float)H / (float)S, fx = (float)W / (float)S;
  const int ys = nearest_src(y, fy, H);
  for (int x0 = 0; x0 < W; x0 += 128) {               // source rows wider than 128 pixels: in slabs
    const int nw = min(128, W - x0);
    for (int i = threadIdx.x; i < nw * 64; i += 256) {
      const int xs = i >> 6, c = i & 63;
      float v = 0.f;
      if (c < nc) {
        const long o = (long)b * sb + (long)(c0 + c) * sc + (long)ys * sy + (long)(x0 + xs) * sx;
        v = s32 ? s32[o] : (float)s16[o];
      }
      tile[xs * 66 + c] = (_Float16)v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc * S; i += 256) {
      const int c = i / S, x = i - c * S;
      const int xs = nearest_src(x, fx, W) - x0;
      if (xs >= 0 && xs < nw) out[(((size_t)b * Ctot + coff + c0 + c) * S + y) * S + x] = tile[xs * 66 + c];
    }
    __syncthreads();
  }
}

hipError_t launch_resize_concat(const half_t* s16, const float* s32, long sb, long sc, long sy, long sx, int B, int C, int H, int W,
                                half_t* out, int Ctot, int coff, int S, hipStream_t s) {
  if (B <= 0 || C <= 0 || S <= 0) return hipSuccess;
  if (H <= 0 || W <= 0 || coff < 0 || coff + C > Ctot) return hipErrorInvalidValue;
  hipLaunchKernelGGL(resize_concat_kernel, dim3((unsigned)(B * S), (unsigned)((C + 63) / 64)), dim3(256), 0, s, s16, s32, sb, sc, sy, sx, C,
                     H, W, out, Ctot, coff, S);
  return hipGetLastError();
}

// r x r mean of a channels-last hook: src (B, C, H, W) fp16 with strides (sb, 1, sy, sx) -> out (B, H/r, W/r, C) fp16
// (returned to the caller as the (B, C, H/r, W/r) channels-last view, like every hook); fp32 accumulation, 16 B per lane
__global__ __launch_bounds__(256) void avg_pool_kernel(const half_t* src, long sb, long sy, long sx, int C, int OH, int OW, int r,
                                                       half_t* out, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (b, oy, ox, c8)
  if (i >= total) return;
  const int C8 = C / 8;
  const int c = (int)(i % C8) * 8;
  const int ox = (int)((i / C8) % OW), oy = (int)((i / ((long)C8 * OW)) % OH);
  const long b = i / ((long)C8 * OW * OH);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int dy = 0; dy < r; ++dy)
    for (int dx = 0; dx < r; ++dx) {
      const f16x8 v = *(const f16x8*)(src + b * sb + (long)(oy * r + dy) * sy + (long)(ox * r + dx) * sx + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
  const float inv = 1.0f / (float)(r * r);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[e] * inv);
  *(f16x8*)(out + (((size_t)b * OH + oy) * OW + ox) * C + c) = o;
}

hipError_t launch_avg_pool(const half_t* src, long sb, long sy, long sx, int B, int C, int H, int W, int r, half_t* out, hipStream_t s) {
  if (r < 1 || (C & 7) || (sb & 7) || (sy & 7) || (sx & 7) || H < r || W < r) return hipErrorInvalidValue;
  const int OH = H / r, OW = W / r;
  const long total = (long)B * OH * OW * (C / 8);
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(avg_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, sb, sy, sx, C, OH, OW, r, out, total);
  return hipGetLastError();
}

// mean over heads and over `n` <= 32 maps of one (category, size) group: maps[l] (B, heads, Q, K) fp16 contiguous ->
// out (B, Q, K) fp32 (= the channels-last image of the (B, K, sqrt Q, sqrt Q) tensor resize_concat_kernel then consumes)
struct MapPtrs { const half_t* p[32]; };
__global__ __launch_bounds__(256) void maps_mean_kernel(MapPtrs m, int n, int heads, long QK, float* out, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (b, q*K + k)
  if (i >= total) return;
  const long b = i / QK, r = i - b * QK;
  float acc = 0.f;
  for (int l = 0; l < n; ++l) {
    const half_t* p = m.p[l] + (size_t)b * heads * QK + r;
    float a = 0.f;
    for (int h = 0; h < heads; ++h) a += (float)p[(size_t)h * QK];
    // the reference rounds the head mean to fp16 (`attention_probs.mean(1)` on an fp16 tensor) before averaging the layers
    acc += (float)(_Float16)(a / (float)heads);
  }
  out[i] = acc / (float)n;
}

hipError_t launch_maps_mean(const half_t* const* maps, int n, int B, int heads, int Q, int K, float* out, hipStream_t s) {
  if (n < 1 || n > 32 || heads < 1) return hipErrorInvalidValue;
  MapPtrs m{};
  for (int l = 0; l < n; ++l) m.p[l] = maps[l];
  const long QK = (long)Q * K, total = (long)B * QK;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(maps_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, m, n, heads, QK, out, total);
  return hipGetLastError();
}

}  // namespace gdf
