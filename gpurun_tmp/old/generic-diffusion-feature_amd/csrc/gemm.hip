// MFMA GEMM + implicit-GEMM 3x3 convolution for gfx950 (MI355X), fp16 operands, fp32 accumulate: kernels, tile selection, launchers, split-K.
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
// Structure: BM x BN x 64 block tile, mfma_f32_16x16x32_f16, template <MODE, BM, BN, STAGES, GEGLU, DIT>:
//   * STAGES = 9 / 8: the 8-phase main loops of the large tiles (256x320 with 2x4 waves of 128x80 and B resident in registers;
//     256x256 with 4x2 waves of 64x128 and A resident): the two waves of a SIMD run one workgroup barrier apart, so one multiplies
//     while the other reads fragments and issues DMA; half-tile / quarter-tile DMA runs 1.5 K-tiles ahead with ONE counted
//     `s_waitcnt vmcnt(N)` per K-tile.  These carry 65 % of an SDXL step and 70 % of a Flux step (see the blocks below).
//   * STAGES = 3: 256x128, 8 waves, 3-stage LDS ring (144 KiB), counted waits, raw `s_barrier`, one per K-tile.
//   * STAGES = 2: 128x128 / 128x160 / 128x16 (4 waves, 2 workgroups per CU: narrow N, few tiles, epilogue-heavy GEMMs) and
//     the 2-stage ring form of the 256-row tiles (kept as the bit-exact reference of the 8-phase loops, tools/stress_gemm8.py).
//   * Both operands are streamed HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip).
//     The LDS image is lane-linear, so the bank-conflict XOR swizzle is applied on the SOURCE address
//     (chunk ^= row&7) and mirrored on the ds_read_b128 side.
//   * Convolution zero padding, M/N tails and the nearest-x2 upsample are all done in the address
//     generator: out-of-image taps get an out-of-range buffer offset, which the hardware returns as 0.
//   * Epilogue is staged per wave through LDS so every global store / residual load is a full
//     16-byte-per-lane, 128-byte-per-row access.
//
// Source layout (round 6): gemm_common.h (types, LDS-DMA, MFMA wrappers, GELU, tile order), gemm_tile.h (GemmTile: tile geometry, tile origin,
// operand address generators), gemm_mainloop_ring.h (2- / 3-stage LDS rings), gemm_mainloop_8phase.h (the two-group 256x256 / 256x320 loops),
// gemm_epilogue.h (the staged epilogue), this file (gemm_body = locate -> main loop -> epilogue, kernel instantiations, pick_variant, launch_gemm).
#include "gemm_mainloop_ring.h"
#include "gemm_mainloop_8phase.h"
#include "gemm_epilogue.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>

namespace gdf {

template <int MODE, int BM, int BN, int STAGES, bool GEGLU, bool DIT, bool BF = false, bool QKN = false, bool SPLIT = false, bool MX = false,
          bool GNS = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p) {
  using T = GemmTile<MODE, BM, BN, STAGES, GEGLU, DIT, BF, QKN, SPLIT, MX, GNS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T t(p, smem);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nblk = ((p.M + BM - 1) / BM) * tiles_n;
  // Persistent form: the launcher may start fewer workgroups than tiles (one per CU for the two-group 256x256 kernels); workgroup b
  // then walks the tiles b, b + gridDim.x, ... — with gridDim.x a multiple of 8 these are the tiles the hardware would have given the
  // same XCD round after round, so the super-block order below is unchanged.  Saves the workgroup relaunch between rounds
  // (tools/trace_gemm.py: 2.6 us from a tile's last instruction to the first of the next tile on that CU, of ~37 us per tile at
  // K = 1280) and the kernel-argument / descriptor setup.  A plain launch has gridDim.x == nblk: one trip.
  // Compiled as a loop only where the register budget has room for the loop-carried lane constants (256x320: 9-11 VGPRs spilled).
  constexpr bool PERSIST = (STAGES == 8);
  int vb = blockIdx.x;
#if defined(GDF_STAGGER)                                        // diagnostics build (tools/build_variant.sh stagger -DGDF_STAGGER): the de-phasing experiment
  if (p.stagger > 0 && (int)blockIdx.x < p.stagger_wgs) {      // (kernels.h GemmParams::stagger); uniform per workgroup
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long d = (unsigned long long)(p.stagger & 0xffffff) * (unsigned)((blockIdx.x >> 3) % (unsigned)(p.stagger >> 24));
    while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(16);
  }
#endif
  do {
  GDF_TR(0); GDF_TR_ID();
  t.locate(vb, tiles_n, nblk);
  f32x4 acc[T::FM][T::FN];
#pragma unroll
  for (int i = 0; i < T::FM; ++i)
#pragma unroll
    for (int j = 0; j < T::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (STAGES == 8) gemm_mainloop_8phase_256(t, acc);
  else if constexpr (STAGES == 9) gemm_mainloop_8phase_320(t, acc);
  else gemm_mainloop_ring(t, acc);
#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 2
  wait_vmcnt<0>();
#endif
  __syncthreads();   // all waves finished reading the last tile: LDS is free for epilogue staging
  GDF_TR(3);
#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 1
  // diagnostics build (tools/ab_epilogue_bound.sh): NO epilogue — the accumulators are kept alive and dropped.  Results are garbage; the
  // time per launch is what a PERFECTLY overlapped epilogue would leave (the bound on any deferred-epilogue scheme).
#pragma unroll
  for (int i = 0; i < T::FM; ++i)
#pragma unroll
    for (int j = 0; j < T::FN; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
  gemm_epilogue(t, acc);
#endif   // GDF_ABLATE_EPI == 1
  GDF_TR(4);
  if constexpr (!PERSIST) break;
  vb += gridDim.x;
  if (vb >= nblk) break;
  // every wave is done with the staging area before the next tile's DMA lands.  Raw barrier + lgkmcnt only: __syncthreads() would
  // also wait (vmcnt) for this tile's global stores, which may drain under the next tile's prologue
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  } while (true);
}

template <int MODE, int BM, int BN, int STAGES, bool GEGLU>
__global__ __launch_bounds__(BM * 2, 2) void gemm_kernel(const GemmParams p) {
  gemm_body<MODE, BM, BN, STAGES, GEGLU, false>(p);
}
// 3x3 conv whose epilogue also writes GroupNorm partial sums (GemmParams::gn_partial; the VAE and UNet op programs: PlanBuilder::gn_epi)
template <int MODE, int BM, int BN, int STAGES>
__global__ __launch_bounds__(BM * 2, 2) void gemm_gn_kernel(const GemmParams p) {
  gemm_body<MODE, BM, BN, STAGES, false, false, false, false, false, false, true>(p);
}
// the same tiles with split fp16 hi + lo operands ("precise" plans)
template <int MODE, int BM, int BN, int STAGES, bool GEGLU>
__global__ __launch_bounds__(BM * 2, 2) void gemm_split_kernel(const GemmParams p) {
  gemm_body<MODE, BM, BN, STAGES, GEGLU, false, false, false, true>(p);
}
// dense GEMM with the MMDiT epilogue (tanh-GELU / per-sample gate / two-region sample map); BF: bf16 operands and activations
template <int BM, int BN, int STAGES, bool BF, bool QKN>
__global__ __launch_bounds__(BM * 2, 2) void gemm_dit_kernel(const GemmParams p) {
  gemm_body<A_DENSE, BM, BN, STAGES, false, true, BF, QKN>(p);
}

// the MMDiT kernels with split bf16 hi + lo A operands / outputs ('bfloat16x2' plans, gdf_flux.h)
template <int BM, int BN, int STAGES, bool BF, bool QKN>
__global__ __launch_bounds__(BM * 2, 2) void gemm_dit_split_kernel(const GemmParams p) {
  gemm_body<A_DENSE, BM, BN, STAGES, false, true, BF, QKN, true>(p);
}

// MMDiT GEMM on fp8 (e4m3) operands, bf16 output ('fp8-mx' plans, gdf_flux.h): MX-scaled MFMA, K = 128 per instruction
template <int BM, int BN, int STAGES>
__global__ __launch_bounds__(BM * 2, 2) void gemm_mx_kernel(const GemmParams p) {
  gemm_body<A_DENSE, BM, BN, STAGES, false, true, true, false, false, true>(p);
}

// workgroups of a persistent launch of a 1-workgroup-per-CU kernel: the CU count of the current device (a multiple of 8 XCDs);
// GDF_PERSIST=0 (diagnostics) launches one workgroup per tile instead
static int persist_wgs() {
  static const int off = [] { const char* e = getenv("GDF_PERSIST"); return e && atoi(e) == 0; }();
  if (off) return 1 << 30;
  static std::atomic<int> cus[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 1 << 30;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8 || (n & 7)) n = 1 << 30;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <int MODE, int BM, int BN, int STAGES, bool GEGLU, bool DIT = false, bool BF = false, bool QKN = false, bool SPLIT = false, bool MX = false,
          bool GNS = false>
static hipError_t launch_t(const GemmParams& p, hipStream_t s) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int smem = (STAGES >= 8 ? 2 : STAGES) * (BM * 128 + BN * 128);
  static std::atomic<uint64_t> attr_mask{0};             // per template instantiation, one bit per device
  {
    const void* fn;
    if constexpr (MX) fn = (const void*)gemm_mx_kernel<BM, BN, STAGES>;
    else if constexpr (GNS) fn = (const void*)gemm_gn_kernel<MODE, BM, BN, STAGES>;
    else if constexpr (DIT && SPLIT) fn = (const void*)gemm_dit_split_kernel<BM, BN, STAGES, BF, QKN>;
    else if constexpr (DIT) fn = (const void*)gemm_dit_kernel<BM, BN, STAGES, BF, QKN>;
    else if constexpr (SPLIT) fn = (const void*)gemm_split_kernel<MODE, BM, BN, STAGES, GEGLU>;
    else fn = (const void*)gemm_kernel<MODE, BM, BN, STAGES, GEGLU>;
    const hipError_t e = ensure_dyn_smem(attr_mask, fn, smem);
    if (e != hipSuccess) return e;
  }
  GemmParams q = p;
  q.sb_gm = q.sb_gn = 0;
  if (!p.no_superblock) {
    const int conc = (p.cus > 0 ? p.cus / 8 : 32) * ((BM == 256) ? 1 : 2);   // workgroups one XCD keeps resident (32 CUs x 1 or 2; a CU partition: cus / 8)
    static const int gn_max = [] { const char* e = getenv("GDF_SB_GN_MAX"); return e ? atoi(e) : 4; }();   // diagnostics: widest super-block
    for (int gn = gn_max; gn >= 2; gn >>= 1) {
      const int gm = conc / gn;
      if (tiles_n % gn == 0 && tiles_m % gm == 0 && (tiles_m / gm) * (tiles_n / gn) >= 8 && tiles_n > gn) {
        q.sb_gm = gm; q.sb_gn = gn;
        break;
      }
    }
  }
  int gx = tiles_m * tiles_n;
  const int pw = (p.cus > 0 && p.cus < persist_wgs()) ? p.cus : persist_wgs();
  {
    // de-phasing experiment (kernels.h GemmParams::stagger): GDF_STAGGER_US = delay in microseconds, applied to launches of >= GDF_STAGGER_MIN_ROUNDS
    // (default 2) rounds of one-workgroup-per-CU tiles
    static const float us = [] { const char* e = getenv("GDF_STAGGER_US"); return e ? (float)atof(e) : 0.f; }();
    static const int min_rounds = [] { const char* e = getenv("GDF_STAGGER_MIN_ROUNDS"); return e ? atoi(e) : 2; }();
    q.stagger = 0; q.stagger_wgs = 0;
    static const int groups = [] { const char* e = getenv("GDF_STAGGER_GROUPS"); return e ? atoi(e) : 2; }();   // delay of workgroup b: ((b >> 3) % groups) x us
    if (us > 0.f && groups > 1 && BM == 256 && pw < (1 << 30) && gx >= min_rounds * pw) { q.stagger = (int)(us * 100.f) | (groups << 24); q.stagger_wgs = pw; }
  }
  if (STAGES == 8 && gx > pw && !(p.batch > 1)) gx = pw;   // persistent: one workgroup per CU walks the tiles
  const dim3 grid(gx, (STAGES == 2 && p.splitk > 1) ? p.splitk : p.batch > 1 ? p.batch : 1);
  if constexpr (MX) hipLaunchKernelGGL((gemm_mx_kernel<BM, BN, STAGES>), grid, dim3(BM * 2), smem, s, q);
  else if constexpr (GNS) hipLaunchKernelGGL((gemm_gn_kernel<MODE, BM, BN, STAGES>), grid, dim3(BM * 2), smem, s, q);
  else if constexpr (DIT && SPLIT) hipLaunchKernelGGL((gemm_dit_split_kernel<BM, BN, STAGES, BF, QKN>), grid, dim3(BM * 2), smem, s, q);
  else if constexpr (DIT) hipLaunchKernelGGL((gemm_dit_kernel<BM, BN, STAGES, BF, QKN>), grid, dim3(BM * 2), smem, s, q);
  else if constexpr (SPLIT) hipLaunchKernelGGL((gemm_split_kernel<MODE, BM, BN, STAGES, GEGLU>), grid, dim3(BM * 2), smem, s, q);
  else hipLaunchKernelGGL((gemm_kernel<MODE, BM, BN, STAGES, GEGLU>), grid, dim3(BM * 2), smem, s, q);
  return hipGetLastError();
}

// Tile selection.  Every channel count of the SD / SDXL UNets is a multiple of 160 (320 k), so the 128x160 tile
// (2 workgroups per CU, 72 KiB LDS each) covers N without a ragged last column tile and makes M/128 * N/160 a
// multiple of the 512 workgroup slots for the SDXL batch-16 shapes (no tail round).  128x128 serves other N;
// 256x128 (8 waves, 3-stage ring) wins for very large problems.
// fraction of the workgroup slots that do useful work when `tiles` equal tiles run on `slots` concurrent slots (whole rounds)
static double round_fill(long tiles, int slots) {
  if (tiles <= 0) return 0.0;
  const long rounds = (tiles + slots - 1) / slots;
  return (double)tiles / (double)(rounds * slots);
}

static int pick_variant_any(const GemmParams& p);
// split-operand launches ("precise" plans) are instantiated for a reduced set of tiles: dense / conv 256x320 two-group, 128x160,
// 128x128; GEGLU 256x256 two-group and 128x128; the narrow-N tile
static bool is_split(const GemmParams& p) { return !p.dit && (p.k_w > 0 || p.o16_lo > 0); }
static bool is_dit_split(const GemmParams& p) { return p.dit && (p.k_w > 0 || p.o16_lo > 0); }
static int pick_variant(const GemmParams& p) {
  const int v = pick_variant_any(p);
  if (is_dit_split(p)) return (v == 8256 || v == 1256) ? 8256 : 128;      // 'bfloat16x2' MMDiT plans: 256x256 two-group or 128x128
  if (!is_split(p) || v == 16) return v;
  if (p.geglu) return v == 825 ? 825 : 128;
  if (p.mode == A_CONV_SMALLC) return v == 160 ? 160 : 128;
  return (v == 932 || v == 160) ? v : 128;
}
static int pick_variant_any(const GemmParams& p) {
  const int S1 = p.cus > 0 ? p.cus : 256, S2 = 2 * S1;   // workgroup slots at 1 / 2 workgroups per CU (whole chip or a CU partition)
  if (p.dit) {   // MMDiT widths are multiples of 256 (3072 = 24 x 128): 256x256 tiles (128 KiB ring, 1 workgroup / CU)
    const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (p.variant == 128 || p.variant == 1256 || p.variant == 2128 || p.variant == 8256) return p.variant;
    // 8-phase schedule: 1177-1362 vs 1002-1188 TFLOP/s (2-stage ring) at the Flux shapes.  A ragged last column tile is fine up to
    // 1/8 of padding (PixArt C = 1152 = 4.5 x 256: 1017-1187 vs 873-1020 on the 256x128 ring)
    if (t256 >= 128 && (long)((p.N + 255) / 256) * 256 <= (long)p.N + p.N / 8) return 8256;
    return (p.N % 128 == 0 && (long)((p.M + 255) / 256) * (p.N / 128) >= 256) ? 2128 : 128;   // PixArt: C = 1152 = 9 x 128
  }
  if (p.bn == 16) return 16;
  if (p.splitk > 1) {                                                            // split-K lives in the 2-stage ring tiles
    static const int force = [] { const char* e = getenv("GDF_SPLITK_TILE"); return e ? atoi(e) : 0; }();   // diagnostics: 128 | 160
    if (force == 160 && p.N % 160 == 0) return 160;
    if (force == 128 && p.N % 128 == 0) return 128;
    // round 5: the 128x160 tile whenever it divides N (SD1.5's 8x8 level, N = 1280: 750 / 891 vs 715 / 824 TFLOP/s at K = 11520 / 23040,
    // tools/bench_conv_small_m.py); gemm_splitk_factor counts its tiles the same way
    return (p.N % 160 == 0) ? 160 : 128;
  }
  if (p.variant) return p.variant;
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  const long tiles320 = (long)((p.M + 255) / 256) * ((p.N + 319) / 320);
  if (p.geglu) {
    // 8-phase 256x256: 1130 vs 1073 TFLOP/s (256x320 ring) at 16384 x 10240 x 1280, 899 vs 869 at 65536 x 5120 x 640
    // The batch-16 shapes fill whole rounds of every tile; other batch sizes may leave a mostly idle last round, so the
    // candidates are ranked by (measured rate at full rounds) x (fill of the rounds they need)
    const long tm256 = (p.M + 255) / 256, tm128 = (p.M + 127) / 128;
    double best = 0.0; int bv = 128;
    auto cand = [&](int v, double rate, long tiles, int slots) { const double sc = rate * round_fill(tiles, slots); if (sc > best) { best = sc; bv = v; } };
    if (p.N % 256 == 0) cand(825, 1.00, tm256 * (p.N / 256), S1);
    if (p.N % 320 == 0) cand(320, 0.95, tm256 * (p.N / 320), S1);
    cand(256, 0.80, tm256 * ((p.N + 127) / 128), S1);
    cand(128, 0.70, tm128 * ((p.N + 127) / 128), S2);
    return bv;
  }
  if (p.mode != A_DENSE) {                                                 // convs (K = 9 Cin is long)
    if (p.N % 320 == 0) {                                                  // 8-phase 256x320: 1150-1350 TFLOP/s (ring 1090-1310, 128x160 950-1140)
      const double s932 = 1.00 * round_fill(tiles320, S1), s160 = 0.85 * round_fill((long)((p.M + 127) / 128) * (p.N / 160), S2);
      const double s128 = 0.70 * round_fill((long)((p.M + 127) / 128) * ((p.N + 127) / 128), S2);
      return (s932 >= s160 && s932 >= s128) ? 932 : (s160 >= s128 ? 160 : 128);
    }
    if (p.N % 160 == 0) return 160;
    if (p.N % 256 == 0 && (long)((p.M + 255) / 256) * (p.N / 256) >= 256) return 826;   // VAE widths 256 / 512: 989-1146 vs 830-965 (256x128 ring)
    return (p.N <= 128 && p.M >= (1 << 20)) ? 256 : 128;                   // VAE level-0 convs (N = 128, 4 M pixels): 797 vs 697
  }
  // short-K GEMMs with the fp32 residual epilogue (attention out-projections: 10 B/element of epilogue traffic against
  // 20 K-tiles of MFMA work) fill the chip in ONE round of 256x320 tiles, so main loop and epilogue traffic never overlap;
  // 128x160 tiles run 2 workgroups per CU and 2+ rounds (80 vs 89 us at 16384 x 1280 x 1280)
  // (round 2: with the two-phase main loop the 256x320 tile wins again where its tiles fill whole rounds — 16384 x 1280 x 1280
  // 64.7 vs 77.7 us, 32768 x 640 x 640 47.0 vs 49.8, 65536 x 640 x 640 equal; it still loses at half-filled rounds, 8192 x 1280 x 1280
  // 47.3 vs 37.3, and at N = 320, tools/bench_res32.py)
  if (p.res32 && p.K <= 1536 && p.N % 160 == 0 && tiles320 <= S2 &&
      !(p.N % 320 == 0 && p.N >= 640 && p.K >= 640 && tiles320 % S1 == 0)) return 160;
  if (p.N % 320 == 0) {                                                    // 8-phase: qkv 1113, ff_out 1088, attn2_q 1045, shortcut 1086 (ring: 1051 / 983 / 980 / 1002)
    const double s932 = 1.00 * round_fill(tiles320, S1), s160 = 0.87 * round_fill((long)((p.M + 127) / 128) * (p.N / 160), S2);
    const double s128 = 0.72 * round_fill((long)((p.M + 127) / 128) * ((p.N + 127) / 128), S2);
    return (s932 >= s160 && s932 >= s128) ? 932 : (s160 >= s128 ? 160 : 128);
  }
  if (p.N % 160 == 0 && p.K >= 1024) return 160;
  if ((long)p.M * p.N >= (1L << 26) && tiles256 >= S2) return 256;        // short-K, large MxN (qkv @ C=640): 712 vs 642
  return 128;
}

// true when an MMDiT GEMM of this shape runs on the 256x256 tile, i.e. may carry the fused RMSNorm + RoPE epilogue (qkn_*)
bool gemm_qkn_ok(int M, int N, int K) {
  GemmParams g{}; g.M = M; g.N = N; g.K = K; g.dit = 1; g.mode = A_DENSE;
  const int v = pick_variant(g);
  return v == 8256 || v == 1256;
}

// kernel symbol (as rocprofv3 prints it) that launch_gemm would pick for these parameters
const char* gemm_kernel_name(const GemmParams& p) {
  const int v = pick_variant(p);
  int bm = 128, bn = 128, st = 2;
  if (v == 16) bn = 16; else if (v == 160) bn = 160; else if (v == 256) { bm = 256; st = 3; } else if (v == 320) { bm = 256; bn = 320; }
  else if (v == 825) { bm = 256; bn = 256; st = 8; } else if (v == 932) { bm = 256; bn = 320; st = 9; } else if (v == 826) { bm = 256; bn = 256; st = 8; }
  if (p.mode == A_CONV_SMALLC && v != 160) { bm = 128; bn = 128; st = 2; }
  char tmp[64];
  if (p.gn_partial && !p.dit) snprintf(tmp, sizeof tmp, "gemm_gn_kernel<%d, %d, %d, %d>", p.mode, bm, bn, st);
  else if (p.dit && p.mx) snprintf(tmp, sizeof tmp, "gemm_mx_kernel<256, 256, 8>");
  else if (p.dit) snprintf(tmp, sizeof tmp, "%s<%d, %d, %d, %s, %s>", is_dit_split(p) ? "gemm_dit_split_kernel" : "gemm_dit_kernel", v == 128 ? 128 : 256, (v == 1256 || v == 8256) ? 256 : 128, v == 8256 ? 8 : v == 2128 ? 3 : 2, p.bf16 ? "true" : "false", p.qkn_nq ? "true" : "false");
  else snprintf(tmp, sizeof tmp, "%s<%d, %d, %d, %d, %s>", is_split(p) ? "gemm_split_kernel" : "gemm_kernel", p.mode, bm, bn, st, p.geglu ? "true" : "false");
  // interned: the returned pointer stays valid for the life of the library (plan build time only, mutex-protected)
  static std::mutex mu;
  static std::deque<std::string> names;
  std::lock_guard<std::mutex> lk(mu);
  for (const std::string& n : names) if (n == tmp) return n.c_str();
  names.emplace_back(tmp);
  return names.back().c_str();
}

// GroupNorm partial sums from the epilogue: plain 3x3 convs on the tiles whose wave tile is 64 rows (128x128, 256x128 ring, 256x256 two-group)
int gemm_gn_slab_rows(const GemmParams& p) {
  if ((p.mode != A_CONV3 && p.mode != A_CONV_SMALLC) || p.dit || p.geglu || p.splitk > 1 || p.batch > 1 || is_split(p) || p.bn == 16) return 0;
  if ((p.M % 64) != 0 || (p.N % 8) != 0) return 0;
  const int v = pick_variant(p);
  if (p.mode == A_CONV_SMALLC) return v != 160 ? 64 : 0;                 // conv_in: the 128x128 tile
  if (v == 932) return (p.M % 128) == 0 ? 128 : 0;                       // round 5: 256x320 two-group (the UNet's N = 320 k convs): 128-row wave tiles
  return (v == 128 || v == 160 || v == 256 || v == 826) ? 64 : 0;
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.mode != A_CONV_SMALLC && (p.K % BK) != 0) return hipErrorInvalidValue;
  if (p.mode == A_CONV3 && (p.Cin % BK) != 0) return hipErrorInvalidValue;
  if (p.k_w > 0 && (p.K != 2 * p.k_w || (p.k_w % BK) != 0 || p.mode == A_CONV_SMALLC || (p.dit && !p.bf16) || (p.mode == A_CONV3 && (p.k_w % (9 * BK)) != 0)))
    return hipErrorInvalidValue;                                                         // split operands: K = [hi | lo] over one weight matrix
  if (p.o16_lo > 0 && ((p.dit && !p.bf16) || p.bn == 16 || (p.o16_lo % 8) != 0)) return hipErrorInvalidValue;   // (MMDiT: the bf16 pair form only)
  if (p.out_f16 && !(p.dit && p.bf16)) return hipErrorInvalidValue;
  const int v = pick_variant(p);
  if (v != 16 && ((p.geglu ? p.N / 2 : p.N) % 8) != 0) return hipErrorInvalidValue;   // ragged N only in the BN = 16 variant
  if (p.bf16 && !p.dit) return hipErrorInvalidValue;                                      // bf16 exists on the MMDiT path only
  if (p.gn_partial) {                                                                     // GroupNorm partial sums from the epilogue (VAE convs)
    if (gemm_gn_slab_rows(p) == 0 || !p.out16) return hipErrorInvalidValue;
    if (p.mode == A_CONV_SMALLC) return launch_t<A_CONV_SMALLC, 128, 128, 2, false, false, false, false, false, false, true>(p, s);
    if (v == 932) return launch_t<A_CONV3, 256, 320, 9, false, false, false, false, false, false, true>(p, s);
    if (v == 160) return launch_t<A_CONV3, 128, 160, 2, false, false, false, false, false, false, true>(p, s);
    if (v == 826) return launch_t<A_CONV3, 256, 256, 8, false, false, false, false, false, false, true>(p, s);
    if (v == 256) return launch_t<A_CONV3, 256, 128, 3, false, false, false, false, false, false, true>(p, s);
    return launch_t<A_CONV3, 128, 128, 2, false, false, false, false, false, false, true>(p, s);
  }
  if (p.dit) {
    if (p.mode != A_DENSE || p.geglu || p.batch > 1) return hipErrorInvalidValue;
    if (p.qkn_nq && ((v != 8256 && v != 1256) || (p.qkn_nq % 128) != 0)) return hipErrorInvalidValue;   // one head per 128-column wave tile
    if (p.qkn_nq && (p.res32 || p.res16 || p.rowvec || p.aux16 || p.out32)) return hipErrorInvalidValue;  // the QKN instantiation: bias -> norm + RoPE -> out16 only
    if (p.mx) {                                                                           // fp8 (e4m3) operands ('fp8-mx' plans): 256x256 two-group tile only
      if (!p.bf16 || p.qkn_nq || is_dit_split(p) || (p.N % 8)) return hipErrorInvalidValue;
      return launch_t<A_DENSE, 256, 256, 8, false, true, true, false, false, true>(p, s);
    }
    if (is_dit_split(p)) {                                                                // bf16 hi + lo operands ('bfloat16x2' plans)
      if (v == 8256) return p.qkn_nq ? launch_t<A_DENSE, 256, 256, 8, false, true, true, true, true>(p, s) : launch_t<A_DENSE, 256, 256, 8, false, true, true, false, true>(p, s);
      return p.qkn_nq ? hipErrorInvalidValue : launch_t<A_DENSE, 128, 128, 2, false, true, true, false, true>(p, s);
    }
    if (p.bf16) {
      if (v == 8256) return p.qkn_nq ? launch_t<A_DENSE, 256, 256, 8, false, true, true, true>(p, s) : launch_t<A_DENSE, 256, 256, 8, false, true, true>(p, s);
      if (v == 1256) return p.qkn_nq ? launch_t<A_DENSE, 256, 256, 2, false, true, true, true>(p, s) : launch_t<A_DENSE, 256, 256, 2, false, true, true>(p, s);
      if (v == 2128) return launch_t<A_DENSE, 256, 128, 3, false, true, true>(p, s);
      return launch_t<A_DENSE, 128, 128, 2, false, true, true>(p, s);
    }
    if (v == 8256) return p.qkn_nq ? launch_t<A_DENSE, 256, 256, 8, false, true, false, true>(p, s) : launch_t<A_DENSE, 256, 256, 8, false, true>(p, s);
    if (v == 1256) return p.qkn_nq ? launch_t<A_DENSE, 256, 256, 2, false, true, false, true>(p, s) : launch_t<A_DENSE, 256, 256, 2, false, true>(p, s);
    if (v == 2128) return launch_t<A_DENSE, 256, 128, 3, false, true>(p, s);
    return launch_t<A_DENSE, 128, 128, 2, false, true>(p, s);
  }
  if (is_split(p)) {                                                                      // "precise" plans: the reduced tile set of pick_variant
    if (p.geglu) {
      if (p.mode != A_DENSE || (p.N % 32) != 0) return hipErrorInvalidValue;
      return v == 825 ? launch_t<A_DENSE, 256, 256, 8, true, false, false, false, true>(p, s)
                      : launch_t<A_DENSE, 128, 128, 2, true, false, false, false, true>(p, s);
    }
    if (v == 16) {
      if (p.mode == A_CONV3) return launch_t<A_CONV3, 128, 16, 2, false, false, false, false, true>(p, s);   // conv_out
      return hipErrorInvalidValue;
    }
    switch (p.mode) {
      case A_DENSE:
        if (v == 160) return launch_t<A_DENSE, 128, 160, 2, false, false, false, false, true>(p, s);
        if (v == 932) return launch_t<A_DENSE, 256, 320, 9, false, false, false, false, true>(p, s);
        return launch_t<A_DENSE, 128, 128, 2, false, false, false, false, true>(p, s);
      case A_CONV3:
        if (v == 160) return launch_t<A_CONV3, 128, 160, 2, false, false, false, false, true>(p, s);
        if (v == 932) return launch_t<A_CONV3, 256, 320, 9, false, false, false, false, true>(p, s);
        return launch_t<A_CONV3, 128, 128, 2, false, false, false, false, true>(p, s);
      case A_CONV_SMALLC:
        return v == 160 ? launch_t<A_CONV_SMALLC, 128, 160, 2, false, false, false, false, true>(p, s)
                        : launch_t<A_CONV_SMALLC, 128, 128, 2, false, false, false, false, true>(p, s);
    }
    return hipErrorInvalidValue;
  }
  if (p.geglu) {
    // weight rows / bias interleaved [16 h | 16 gate] (launch_relayout_rows geglu = 16)
    if (p.mode != A_DENSE || (p.N % 32) != 0) return hipErrorInvalidValue;
    if (v == 825) return launch_t<A_DENSE, 256, 256, 8, true>(p, s);
    if (v == 320) return launch_t<A_DENSE, 256, 320, 2, true>(p, s);
    return v == 256 ? launch_t<A_DENSE, 256, 128, 3, true>(p, s) : launch_t<A_DENSE, 128, 128, 2, true>(p, s);
  }
  if (v == 16) {
    if (p.mode == A_CONV3) return launch_t<A_CONV3, 128, 16, 2, false>(p, s);
    if (p.mode == A_DENSE) return launch_t<A_DENSE, 128, 16, 2, false>(p, s);
    return hipErrorInvalidValue;
  }
  switch (p.mode) {
    case A_DENSE:
      if (v == 160) return launch_t<A_DENSE, 128, 160, 2, false>(p, s);
      if (v == 932) return launch_t<A_DENSE, 256, 320, 9, false>(p, s);
      if (v == 320) return launch_t<A_DENSE, 256, 320, 2, false>(p, s);
      if (v == 256) return launch_t<A_DENSE, 256, 128, 3, false>(p, s);
      return launch_t<A_DENSE, 128, 128, 2, false>(p, s);
    case A_CONV3:
      if (v == 160) return launch_t<A_CONV3, 128, 160, 2, false>(p, s);
      if (v == 932) return launch_t<A_CONV3, 256, 320, 9, false>(p, s);
      if (v == 826) return launch_t<A_CONV3, 256, 256, 8, false>(p, s);
      if (v == 320) return launch_t<A_CONV3, 256, 320, 2, false>(p, s);
      if (v == 256) return launch_t<A_CONV3, 256, 128, 3, false>(p, s);
      return launch_t<A_CONV3, 128, 128, 2, false>(p, s);
    case A_CONV_SMALLC:
      return v == 160 ? launch_t<A_CONV_SMALLC, 128, 160, 2, false>(p, s) : launch_t<A_CONV_SMALLC, 128, 128, 2, false>(p, s);
  }
  return hipErrorInvalidValue;
}

// ---- split-K: sum the partial slabs in a fixed order (deterministic) and apply the GEMM epilogue of kernels.h ----
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const float* ws, int splitk) {
  const int CH = p.N / 8;
  const long total = (long)p.M * CH;
  const float a_sc = p.acc_scale != 0.f ? p.acc_scale : 1.0f, o_sc = p.out16_scale != 0.f ? p.out16_scale : 1.0f;
  const size_t slab = (size_t)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / CH), col = (int)(i - (long)row * CH) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int sp = 0; sp < splitk; ++sp) {
      const f32x4* q = (const f32x4*)(ws + sp * slab + (size_t)row * p.N + col);
      const f32x4 a = q[0], b = q[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] *= a_sc;
      if (p.bias) v[e] += p.bias[col + e];
      if (p.rowvec) v[e] += p.rowvec[(size_t)(row / p.rows_per_sample) * p.ldrv + col + e];
    }
    if (p.aux16) {
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
      *(f16x8*)(p.aux16 + (size_t)row * p.ldaux + col) = h;
    }
    if (p.res32) {
      const f32x4* q = (const f32x4*)(p.res32 + (size_t)row * p.ldres + col);
      const f32x4 a = q[0], b = q[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    } else if (p.res16) {
      const f16x8 r = *(const f16x8*)(p.res16 + (size_t)row * p.ldres + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    if (p.out16) {
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (_Float16)(v[e] * o_sc);
      *(f16x8*)(p.out16 + (size_t)row * p.ldo16 + col) = h;
      if (p.o16_lo > 0) {
        f16x8 l;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = (_Float16)(v[e] * o_sc - (float)h[e]);
        *(f16x8*)(p.out16 + (size_t)row * p.ldo16 + col + p.o16_lo) = l;
      }
    }
    if (p.out32) {
      f32x4* q = (f32x4*)(p.out32 + (size_t)row * p.ldo32 + col);
      q[0] = f32x4{v[0], v[1], v[2], v[3]};
      q[1] = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

// > 1 when splitting K pays: a plain (UNet) GEMM / conv whose 128-row tiles fill less than half of the chip's 512 workgroup slots
// while every tile walks a long K.  The factor keeps >= 16 K-tiles per range and aims at ~2 workgroups per CU.
int gemm_splitk_factor(const GemmParams& p) {
  static const int off = [] { const char* e = getenv("GDF_SPLITK"); return e && atoi(e) == 0; }();     // diagnostics: GDF_SPLITK=0
  if (off || p.dit || p.geglu || p.bn == 16 || p.batch > 1 || p.mode == A_CONV_SMALLC || p.variant || (p.N % 128) || (p.K % BK)) return 1;
  const long tiles = (long)((p.M + 127) / 128) * ((p.N % 160 == 0) ? p.N / 160 : p.N / 128);       // (the split tile: pick_variant_any)
  const int nk = p.K / BK;
  const int S1 = p.cus > 0 ? p.cus : 256;
  if (tiles >= S1 || nk < 64) return 1;
  int s = (int)(2 * S1 / tiles);
  if (s > nk / 16) s = nk / 16;
  if (s > 8) s = 8;
  return s < 2 ? 1 : s;
}

hipError_t launch_gemm_splitk(const GemmParams& p, int splitk, float* ws, hipStream_t s) {
  if (splitk <= 1) return launch_gemm(p, s);
  if (p.dit || p.geglu || p.batch > 1 || (p.N % 8) || p.mode == A_CONV_SMALLC || (p.K % BK)) return hipErrorInvalidValue;
  if (splitk > p.K / BK) splitk = p.K / BK;
  if (splitk <= 1) return launch_gemm(p, s);
  GemmParams g = p;                                       // pass 1: raw partial sums, one slab per K range
  g.bias = nullptr; g.rowvec = nullptr; g.res32 = nullptr; g.res16 = nullptr; g.aux16 = nullptr; g.out16 = nullptr;
  g.acc_scale = 0.f; g.out16_scale = 0.f;
  g.out32 = ws; g.ldo32 = p.N; g.splitk = splitk; g.o32_sstride = (long)p.M * p.N; g.variant = 0;
  hipError_t e = launch_gemm(g, s);
  if (e != hipSuccess) return e;
  long blocks = ((long)p.M * (p.N / 8) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, (const float*)ws, splitk);
  return hipGetLastError();
}

}  // namespace gdf

#if defined(GDF_TRACE)
extern "C" int gdf_debug_trace(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gdf::gdf_trace), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
