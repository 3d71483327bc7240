// MFMA GEMM + implicit-GEMM 3x3 convolution for gfx950 (MI355X), fp16 operands, fp32 accumulate.
//
//   D[M,N] = A[M,K] * Wt[N,K]^T  with a fused epilogue (bias, per-sample row vector, residual add,
//   GEGLU gate, fp16/fp32 dual store, pre-residual aux store).
//
// Reference ops this one kernel family replaces (paths under /root/reference/feature/diffusers/models):
//   nn.Linear in Attention.to_q/to_k/to_v/to_out (attention_processor.py:241-267), FeedForward.net
//   (attention.py:1238-1258, GEGLU), Transformer2DModel.proj_in/proj_out (transformers/transformer_2d.py:178-209),
//   nn.Conv2d 3x3 in ResnetBlock2D.conv1/conv2 (resnet.py:269,285), conv_shortcut 1x1 (resnet.py:311-318),
//   Downsample2D.conv stride 2 (downsampling.py:115-118), Upsample2D nearest x2 + conv (upsampling.py:176-193),
//   UNet conv_in / conv_out (unet/unet_2d_condition.py:260-262,480-482).
//
// Structure: 128 x BN x 64 block tile, 4 waves (64-lane), mfma_f32_16x16x32_f16.
//   * Both operands are streamed HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip).
//     The LDS image is lane-linear, so the bank-conflict XOR swizzle is applied on the SOURCE address
//     (chunk ^= row&7) and mirrored on the ds_read_b128 side.
//   * Convolution zero padding, M/N tails and the nearest-x2 upsample are all done in the address
//     generator: out-of-image taps get an out-of-range buffer offset, which the hardware returns as 0.
//   * Double-buffered LDS, one barrier per K-tile; the next tile's DMA is issued before the MFMAs
//     of the current one.
//   * Epilogue is staged per wave through LDS so every global store / residual load is a full
//     16-byte-per-lane, 128-byte-per-row access.
#include "kernels.h"

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

static constexpr int BM = 128;
static constexpr int BK = 64;             // halves per K-tile -> 128-byte LDS rows
static constexpr uint32_t OOB = 0x80000000u;   // any offset >= num_records reads as zero

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// XCD-aware bijective remap: consecutive tiles (which share the same A row-block) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

template <int MODE, int BN, bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  constexpr int WGN = (BN == 128) ? 2 : 1;       // waves along N
  constexpr int WGM = 4 / WGN;                   // waves along M
  constexpr int WTM = BM / WGM;                  // 64 or 32
  constexpr int WTN = BN / WGN;                  // 64 or 16
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_TILE = BM * 128;               // bytes
  constexpr int B_TILE = BN * 128;
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int B_INSTR = BN / 8;                // 1-KiB wave-instructions per B tile
  constexpr int B_PER_WAVE = (B_INSTR + 3) / 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nblk = gridDim.x;
  const int t = xcd_remap(blockIdx.x, nblk);
  const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, p.w_bytes, 0x00020000);

  // ---- per-lane load geometry: every wave issues 4 A instructions (8 rows x 128 B each) ----
  const int lrow = lane >> 3;                         // row inside an 8-row instruction
  const int chunk = (lane & 7) ^ lrow;                // source 16-B chunk (swizzle on the source side)
  uint32_t a_off[4];                                  // DENSE: byte offset of (row, chunk); CONV: pixel row base
  int a_oy[4], a_ox[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + (wave * 4 + j) * 8 + lrow;
    if (MODE == A_DENSE) {
      a_off[j] = (m < p.M) ? (uint32_t)m * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u : OOB;
    } else {
      const int hw = p.OH * p.OW;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[j] = (uint32_t)(n * p.H * p.W);           // pixel index base of sample n
      a_oy[j] = (m < p.M) ? oy * p.stride - 1 : -(1 << 20);
      a_ox[j] = ox * p.stride - 1;
    }
  }
  uint32_t b_off[B_PER_WAVE];
  bool b_act[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int q = wave * B_PER_WAVE + j;              // instruction index inside the B tile
    b_act[j] = q < B_INSTR;
    const int n = n0 + q * 8 + lrow;
    b_off[j] = (n < p.N) ? (uint32_t)n * (uint32_t)p.K * 2u + (uint32_t)chunk * 16u : OOB;
  }

  const int nk = (MODE == A_CONV_SMALLC) ? 2 : p.K / BK;
  const int cpb = (MODE == A_CONV3) ? p.Cin / BK : 1;  // K-tiles per filter tap
  const int IH = p.ups ? 2 * p.H : p.H, IW = p.ups ? 2 * p.W : p.W;

  auto issue = [&](int kt, int buf, int tap, int cb) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t off;
      if (MODE == A_DENSE) {
        off = a_off[j] + (uint32_t)kt * 128u;          // OOB stays >= 2^31
      } else if (MODE == A_CONV3) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
        const bool ok = (iy >= 0) & (iy < IH) & (ix >= 0) & (ix < IW);
        const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
        off = ok ? (a_off[j] + (uint32_t)(sy * p.W + sx)) * (uint32_t)p.lda * 2u + (uint32_t)(cb * BK + chunk * 8) * 2u
                 : OOB;
      } else {  // SMALLC: 8 channels per pixel = one 16-B chunk per tap; chunk index == tap - 8*kt
        const int tp = kt * 8 + chunk;
        const int ky = tp / 3, kx = tp - ky * 3;
        const int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
        const bool ok = (tp < 9) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        off = ok ? (a_off[j] + (uint32_t)(iy * p.W + ix)) * 16u : OOB;
      }
      glds16(rsA, sA + (wave * 4 + j) * 1024, off);
    }
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      if (b_act[j]) {
        const uint32_t off = b_off[j] + (uint32_t)kt * 128u;
        glds16(rsB, sB + (wave * B_PER_WAVE + j) * 1024, off);
      }
    }
  };

  // ---- accumulators ----
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int frow = lane & 15, fk = lane >> 4;

  int tap = 0, cb = 0;
  issue(0, 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (all outstanding DMA of this wave) and every wave is done reading buf[(kt+1)&1]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) {
      if (MODE == A_CONV3) { if (++cb == cpb) { cb = 0; ++tap; } }
      issue(kt + 1, (kt + 1) & 1, tap, cb);
    }
    const char* sA = smem + (kt & 1) * STAGE;
    const char* sB = sA + A_TILE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 af[FM], bf[FN];
      const int kc = kk * 4 + fk;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm * WTM + i * 16 + frow;
        af[i] = *(const f16x8*)(sA + r * 128 + ((kc ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int r = wn * WTN + j * 16 + frow;
        bf[j] = *(const f16x8*)(sB + r * 128 + ((kc ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all waves finished reading the last tile: LDS is free for epilogue staging

  // ---- epilogue: per-wave staging of 32-row slabs through LDS ----
  constexpr int SLD = WTN + 4;                         // padded row length (floats)
  constexpr int PASSES = WTM / 32;                     // 2 (64-row wave tile) or 1
  constexpr int FPP = FM / PASSES;                     // 16-row fragments per pass (2)
  float* st = (float*)(smem) + wave * (32 * SLD);
  constexpr int OUTW = GEGLU ? WTN / 2 : WTN;          // output columns produced by this wave tile
  constexpr int LPR = OUTW / 8;                        // lanes per row (8 output columns per lane)
  constexpr int RPI = 64 / LPR;                        // rows per iteration
  const int Nout = GEGLU ? p.N / 2 : p.N;
  const int ocol0 = GEGLU ? (n0 + wn * WTN) / 2 : (n0 + wn * WTN);

#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
    for (int i2 = 0; i2 < FPP; ++i2)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st[(i2 * 16 + fk * 4 + r) * SLD + j * 16 + frow] = acc[ps * FPP + i2][j][r];
    // same-wave LDS RAW across lanes: DS ops of one wave execute in order
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int lr = it * RPI + lane / LPR;
      const int lc = (lane % LPR) * 8;
      const int row = m0 + wm * WTM + ps * 32 + lr;
      const int col = ocol0 + lc;
      float v[8];
      if (GEGLU) {
        const f32x4 h0 = *(const f32x4*)(st + lr * SLD + lc), h1 = *(const f32x4*)(st + lr * SLD + lc + 4);
        const f32x4 g0 = *(const f32x4*)(st + lr * SLD + 32 + lc), g1 = *(const f32x4*)(st + lr * SLD + 32 + lc + 4);
        const int bcol = n0 + wn * WTN + lc;           // bias is stored in the interleaved GEMM column order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float hb0 = h0[e], hb1 = h1[e], gb0 = g0[e], gb1 = g1[e];
          if (p.bias) {
            if (bcol + e < p.N) { hb0 += p.bias[bcol + e]; gb0 += p.bias[bcol + 32 + e]; }
            if (bcol + 4 + e < p.N) { hb1 += p.bias[bcol + 4 + e]; gb1 += p.bias[bcol + 32 + 4 + e]; }
          }
          v[e] = hb0 * gelu_erf(gb0);
          v[4 + e] = hb1 * gelu_erf(gb1);
        }
      } else {
        const f32x4 x0 = *(const f32x4*)(st + lr * SLD + lc), x1 = *(const f32x4*)(st + lr * SLD + lc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
      }
      if (row < p.M && col < Nout) {
        const int nv = (Nout - col >= 8) ? 8 : (Nout - col);
        if (!GEGLU && p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nv) v[e] += p.bias[col + e];
        }
        if (p.rowvec) {
          const float* rv = p.rowvec + (size_t)(row / p.rows_per_sample) * p.ldrv + col;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nv) v[e] += rv[e];
        }
        if (nv == 8) {
          if (p.aux16) {
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[e];
            *(f16x8*)(p.aux16 + (size_t)row * p.ldaux + col) = hv;
          }
          if (p.res32) {
            const f32x4* rp = (const f32x4*)(p.res32 + (size_t)row * p.ldres + col);
            const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          } else if (p.res16) {
            const f16x8 r = *(const f16x8*)(p.res16 + (size_t)row * p.ldres + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
          }
          if (p.out16) {
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[e];
            *(f16x8*)(p.out16 + (size_t)row * p.ldo16 + col) = hv;
          }
          if (p.out32) {
            f32x4* op = (f32x4*)(p.out32 + (size_t)row * p.ldo32 + col);
            op[0] = f32x4{v[0], v[1], v[2], v[3]};
            op[1] = f32x4{v[4], v[5], v[6], v[7]};
          }
        } else {
          for (int e = 0; e < nv; ++e) {
            float x = v[e];
            if (p.aux16) p.aux16[(size_t)row * p.ldaux + col + e] = (_Float16)x;
            if (p.res32) x += p.res32[(size_t)row * p.ldres + col + e];
            else if (p.res16) x += (float)p.res16[(size_t)row * p.ldres + col + e];
            if (p.out16) p.out16[(size_t)row * p.ldo16 + col + e] = (_Float16)x;
            if (p.out32) p.out32[(size_t)row * p.ldo32 + col + e] = x;
          }
        }
      }
    }
  }
}

template <int MODE, int BN, bool GEGLU>
static hipError_t launch_t(const GemmParams& p, hipStream_t s) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int smem = 2 * (BM * 128 + BN * 128);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<MODE, BN, GEGLU>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<MODE, BN, GEGLU>), dim3(tiles_m * tiles_n), dim3(256), smem, s, p);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.mode != A_CONV_SMALLC && (p.K % BK) != 0) return hipErrorInvalidValue;
  if (p.mode == A_CONV3 && (p.Cin % BK) != 0) return hipErrorInvalidValue;
  if (p.geglu) {
    if (p.mode != A_DENSE || p.bn == 16 || (p.N % 64) != 0) return hipErrorInvalidValue;
    return launch_t<A_DENSE, 128, true>(p, s);
  }
  if (p.bn == 16) {
    if (p.mode == A_CONV3) return launch_t<A_CONV3, 16, false>(p, s);
    if (p.mode == A_DENSE) return launch_t<A_DENSE, 16, false>(p, s);
    return hipErrorInvalidValue;
  }
  switch (p.mode) {
    case A_DENSE: return launch_t<A_DENSE, 128, false>(p, s);
    case A_CONV3: return launch_t<A_CONV3, 128, false>(p, s);
    case A_CONV_SMALLC: return launch_t<A_CONV_SMALLC, 128, false>(p, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gdf
