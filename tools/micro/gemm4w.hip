// Micro-benchmark (experiment, not a product path): a 4-wave / ONE-wave-per-SIMD GEMM main loop with 128 x (BN/2) wave tiles.
//
//   D[M,N] = A[M,K] * Wt[N,K]^T, fp16 operands, fp32 accumulate, fp16 out.   BM = 256, BN = 256 | 320, BK = 64.
//
// Question it answers: the shipped 8-wave kernels (csrc/gemm.hip, two waves per SIMD, 128x80 / 64x128
// wave tiles, 160 / 128 accumulators) read 1/128 + 1/80 bytes of LDS per FLOP.  With one wave per SIMD the register file is
// 512 per lane, so a wave can own 128 x 128 (256 accumulators) or 128 x 160 (320): fewer fragment re-reads, no role
// alternation between two waves, ONE barrier per K-tile.  mfma_f32_32x32x16_f16 (32-cycle issue gaps) instead of 16x16x32.
//
// Structure per wave and K-tile (4 k-steps of 16): fragments of k-step s+1 are read from LDS while the MFMAs of k-step s run;
// the single workgroup barrier sits before k-step 3 (every fragment of the tile has been read by then), after it the DMA of
// tile kt+2 is issued into the buffer just drained, spread over the following k-steps.  LDS: 2 x (A 32 KiB + B BN x 128 B).
// Swizzle: 16-byte chunk c of row r lives at slot c ^ ((r >> 1) & 7) (conflict-free for the 32-row fragments of 32x32x16).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm4w.hip -o gpurun_out/gemm4w && gpurun_out/gemm4w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_AS __attribute__((address_space(3)))
static constexpr uint32_t OOB = 0x80000000u;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <int BN, int VARIANT>
__global__ __launch_bounds__(256, 1) void gemm4w(const _Float16* __restrict__ A, const _Float16* __restrict__ Wt,
                                                 _Float16* __restrict__ D, int M, int N, int K) {
  constexpr int BM = 256;
  constexpr int FN = BN / 64;                     // 32-column blocks per wave (4 or 5)
  constexpr int A_TILE = BM * 128, B_TILE = BN * 128, STAGE = A_TILE + B_TILE;
  constexpr int A_PW = 8, B_PW = BN / 32;         // DMA instructions (8 rows x 128 B each) per wave per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / BN;
  // XCD-aware order: consecutive tiles of one XCD share the A panel
  const int nblk = gridDim.x;
  const int q = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int t = ((xcd < r8) ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (uint32_t)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (uint32_t)((size_t)N * K * 2), 0x00020000);

  // ---- DMA geometry: instruction j of this wave covers rows (wave * PW + j) * 8 + lrow, source chunk = slot ^ ((row >> 1) & 7)
  const int lrow = lane >> 3, slot = lane & 7;
  uint32_t a_off[A_PW], b_off[B_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (wave * A_PW + j) * 8 + lrow;
    const int chunk = slot ^ ((row >> 1) & 7);
    a_off[j] = (m0 + row < M) ? (uint32_t)(m0 + row) * (uint32_t)K * 2u + (uint32_t)chunk * 16u : OOB;
  }
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int row = (wave * B_PW + j) * 8 + lrow;
    const int chunk = slot ^ ((row >> 1) & 7);
    b_off[j] = (n0 + row < N) ? (uint32_t)(n0 + row) * (uint32_t)K * 2u + (uint32_t)chunk * 16u : OOB;
  }
  auto dma_a = [&](int kt, int buf, int j) { glds16(rsA, smem + buf * STAGE + (wave * A_PW + j) * 1024, a_off[j] + (uint32_t)kt * 128u); };
  auto dma_b = [&](int kt, int buf, int j) { glds16(rsB, smem + buf * STAGE + A_TILE + (wave * B_PW + j) * 1024, b_off[j] + (uint32_t)kt * 128u); };

  // ---- fragment geometry (mfma_f32_32x32x16_f16): lane -> row lane & 31, k-chunk (lane >> 5) of the 16-deep k-step ----
  const int fr = lane & 31, fh = lane >> 5;
  // (row >> 1) & 7 == (fr >> 1) & 7 for every fragment row of this lane (block offsets are multiples of 32 rows), so a lane
  // needs 4 swizzled chunk offsets (one per k-step) and two row bases; everything else is an immediate ds_read offset
  uint32_t sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = (uint32_t)(((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4);
  const uint32_t a_base = (uint32_t)((wm * 128 + fr) * 128), b_base = (uint32_t)(A_TILE + (wn * (BN / 2) + fr) * 128);
  auto rd_a = [&](int buf, int ks, int i) -> f16x8 {
    return *(const f16x8*)(smem + buf * STAGE + (a_base + sw[ks]) + i * 4096);
  };
  auto rd_b = [&](int buf, int ks, int j) -> f16x8 {
    return *(const f16x8*)(smem + buf * STAGE + (b_base + sw[ks]) + j * 4096);
  };

  f32x16 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / 64;
  // prologue: tiles 0 and 1 in flight, fragments of (0, k-step 0) in registers
#pragma unroll
  for (int j = 0; j < A_PW; ++j) dma_a(0, 0, j);
#pragma unroll
  for (int j = 0; j < B_PW; ++j) dma_b(0, 0, j);
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < A_PW; ++j) dma_a(1, 1, j);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) dma_b(1, 1, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_PW + B_PW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  f16x8 fa[2][4], fb[2][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[0][i] = rd_a(0, 0, i);
#pragma unroll
  for (int j = 0; j < FN; ++j) fb[0][j] = rd_b(0, 0, j);

  // One k-step (16 deep): FN x 4 MFMAs from fragment set KS & 1; between the MFMAs, in program order: the 4 + FN fragment
  // reads of the NEXT k-step into the other set, then up to one LDS-DMA instruction per MFMA (VARIANT 1) so that the issue
  // cost of a DMA (~60 cycles) sits inside a 32-cycle-per-MFMA stream instead of in front of it.
  //   nbuf / nks: where the next k-step's fragments live;  dma_kt >= 0: tile whose DMA instructions [d0, d1) are issued here
  auto step = [&](auto ksc, int nbuf, auto nksc, bool prefetch, int dma_kt, int dma_buf, auto d0c, auto d1c) {
    constexpr int KS = decltype(ksc)::value, NKS = decltype(nksc)::value;
    constexpr int set = KS & 1, nset = set ^ 1;
    constexpr int D0 = decltype(d0c)::value, D1 = decltype(d1c)::value;
#pragma unroll
    for (int idx = 0; idx < 4 * FN; ++idx) {
      const int j = idx / 4, i = idx % 4;
      if (prefetch) {
        if (idx < 4) fa[nset][idx] = rd_a(nbuf, NKS, idx);
        else if (idx < 4 + FN) fb[nset][idx - 4] = rd_b(nbuf, NKS, idx - 4);
      }
      const int d = D0 + idx - (4 + FN);                 // compile-time DMA instruction index of this slot
      if (VARIANT == 1 && idx >= 4 + FN && d < D1 && dma_kt >= 0) {
        if (d < A_PW) dma_a(dma_kt, dma_buf, d); else dma_b(dma_kt, dma_buf, d - A_PW);
      }
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    }
  };
  constexpr std::integral_constant<int, 0> K0{};
  constexpr std::integral_constant<int, 1> K1{};
  constexpr std::integral_constant<int, 2> K2{};
  constexpr std::integral_constant<int, 3> K3{};
  constexpr int NDMA = A_PW + B_PW;
  constexpr int SLOTS = 4 * FN - 4 - FN;            // DMA slots per k-step (VARIANT 1): 8 (BN 256) / 11 (BN 320)
  static_assert(2 * SLOTS >= NDMA, "two k-steps must carry one tile's DMA");
  constexpr std::integral_constant<int, 0> Z{};
  constexpr std::integral_constant<int, SLOTS> SL{};
  constexpr std::integral_constant<int, NDMA> ND{};

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // k-step 0 also carries the second part of the DMA of tile kt+1 (its buffer was released by the barrier of tile kt-1)
    step(K0, cur, K1, true, (kt >= 1 && kt + 1 < nk) ? kt + 1 : -1, cur ^ 1, SL, ND);
    step(K1, cur, K2, true, -1, 0, Z, Z);
    step(K2, cur, K3, true, -1, 0, Z, Z);
    // every fragment of tile kt has been requested: once landed the buffer is free; tile kt+1 must have landed for everybody
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (VARIANT == 0 && kt + 2 < nk) {
#pragma unroll
      for (int j = 0; j < A_PW; ++j) dma_a(kt + 2, cur, j);
#pragma unroll
      for (int j = 0; j < B_PW; ++j) dma_b(kt + 2, cur, j);
    }
    // k-step 3: prefetch (kt+1, k-step 0) from the other buffer; first part of the DMA of tile kt+2 into the drained buffer
    step(K3, cur ^ 1, K0, kt + 1 < nk, (kt + 2 < nk) ? kt + 2 : -1, cur, Z, SL);
  }

  // ---- epilogue (not tuned: direct stores; the experiment is about the main loop) ----
  // acc[i][j][r]: row = i*32 + 8*(r/4) + 4*(lane>>5) + r%4, col = j*32 + (lane & 31)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
        const int col = n0 + wn * (BN / 2) + j * 32 + fr;
        if (row < M && col < N) D[(size_t)row * N + col] = (_Float16)acc[i][j][r];
      }
}

__global__ void ref_gemm(const _Float16* A, const _Float16* Wt, float* D, int M, int N, int K, int rows) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y * (M / rows);   // a sample of `rows` rows
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)Wt[(size_t)n * K + k];
  D[(size_t)blockIdx.y * N + n] = s;
}

template <int BN, int VARIANT>
static void run(const char* name, int M, int N, int K, const _Float16* dA, const _Float16* dW, _Float16* dD, float* dRef) {
  if (N % BN || M % 256 || K % 64) { printf("%-26s %6d %6d %5d  (shape not tiled by 256x%d)\n", name, M, N, K, BN); return; }
  const int smem = 2 * (256 * 128 + BN * 128);
  hipFuncSetAttribute((const void*)gemm4w<BN, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const dim3 grid((M / 256) * (N / BN));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm4w<BN, VARIANT>), grid, dim3(256), smem, 0, dA, dW, dD, M, N, K);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm4w<BN, VARIANT>), grid, dim3(256), smem, 0, dA, dW, dD, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  // check 16 sampled rows against an fp32 reference
  const int rows = 16;
  hipLaunchKernelGGL(ref_gemm, dim3((N + 255) / 256, rows), dim3(256), 0, 0, dA, dW, dRef, M, N, K, rows);
  std::vector<float> ref((size_t)rows * N); std::vector<_Float16> got((size_t)rows * N);
  hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int r = 0; r < rows; ++r) {
    hipMemcpy(got.data() + (size_t)r * N, dD + (size_t)(r * (M / rows)) * N, (size_t)N * 2, hipMemcpyDeviceToHost);
    for (int n = 0; n < N; ++n) { const double d = (double)(float)got[(size_t)r * N + n] - ref[(size_t)r * N + n]; num += d * d; den += (double)ref[(size_t)r * N + n] * ref[(size_t)r * N + n]; }
  }
  hipError_t err = hipGetLastError();
  printf("%-26s %6d %6d %5d  %8.4f ms  %7.1f TFLOP/s  rel err %.2e %s\n", name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, std::sqrt(num / den),
         err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
  struct S { const char* n; int M, N, K; };
  const S shapes[] = {{"square8k", 8192, 8192, 8192}, {"square4k", 4096, 4096, 4096}, {"sdxl qkv", 16384, 3840, 1280},
                      {"sdxl ff_out", 16384, 1280, 5120}, {"sdxl attn2_q / out", 16384, 1280, 1280}, {"sdxl geglu (plain)", 16384, 10240, 1280},
                      {"sdxl qkv640", 65536, 1920, 640}, {"flux qkv", 36864, 9216, 3072}, {"flux proj_out", 36864, 3072, 15360}};
  size_t maxA = 0, maxW = 0, maxD = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxD = std::max(maxD, (size_t)s.M * s.N); }
  _Float16 *dA, *dW, *dD; float* dRef;
  hipMalloc(&dA, maxA * 2); hipMalloc(&dW, maxW * 2); hipMalloc(&dD, maxD * 2); hipMalloc(&dRef, 16 * 16384 * 4);
  // random data (never zeros: DVFS), uniform [-1, 1) scaled
  std::vector<_Float16> h(std::max(maxA, maxW));
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.5f);
  hipMemcpy(dA, h.data(), maxA * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f);
  hipMemcpy(dW, h.data(), maxW * 2, hipMemcpyHostToDevice);
  for (auto& s : shapes) {
    run<256, 0>("4w 256x256 v0", s.M, s.N, s.K, dA, dW, dD, dRef);
    run<256, 1>("4w 256x256 v1", s.M, s.N, s.K, dA, dW, dD, dRef);
    run<320, 0>("4w 256x320 v0", s.M, s.N, s.K, dA, dW, dD, dRef);
    run<320, 1>("4w 256x320 v1", s.M, s.N, s.K, dA, dW, dD, dRef);
  }
  return 0;
}
