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
#include "kernels.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

static constexpr int BK = 64;                  // halves per K-tile -> 128-byte LDS rows
static constexpr uint32_t OOB = 0x80000000u;   // any offset >= num_records reads as zero

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, 0, 0, 0);
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF: the 16-byte fragments hold bf16 (MMDiT path of a bf16 model); same MFMA rate, same register layout
// one K = 128 step on fp8 (e4m3) operands: a = [a0 | a1], b = [b0 | b1] (16 bytes each), unit e8m0 block scales (127 = 2^0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_mx8(const f16x8 a0, const f16x8 a1, const f16x8 b0, const f16x8 b1, const f32x4 c) {
  const i32x4 x0 = __builtin_bit_cast(i32x4, a0), x1 = __builtin_bit_cast(i32x4, a1);
  const i32x4 y0 = __builtin_bit_cast(i32x4, b0), y1 = __builtin_bit_cast(i32x4, b1);
  const i32x8 a = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  const i32x8 b = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, /*A fp8 e4m3*/ 0, /*B fp8 e4m3*/ 0, 0, 127, 0, 127);
}
template <bool BF>
__device__ __forceinline__ f32x4 mfma16(const f16x8 a, const f16x8 b, const f32x4 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// 16-bit store conversions of the epilogue.  UNet kernels (DIT = false): plain fp16 casts, unchanged code.  MMDiT kernels:
// activations (out16) are bf16 or SATURATING fp16, hook copies (aux16) always saturating fp16 (the reference's hooks are fp16,
// feature_extractor.py:59-60, and real FLUX.1-dev activations leave the fp16 range).
template <bool DIT, bool BF>
__device__ __forceinline__ _Float16 act16(float v) {
  if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)v);
  else if constexpr (DIT) return f32_to_f16_sat(v);
  else return (_Float16)v;
}
template <bool DIT>
__device__ __forceinline__ _Float16 hook16(float v) {
  if constexpr (DIT) return f32_to_f16_sat(v);
  else return (_Float16)v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// exact (erf) GELU, two values at a time (the GEGLU epilogue evaluates 64 per lane; at 2 waves per SIMD the VALU time of the
// round-1 form — Abramowitz-Stegun 7.1.26 with one v_rcp_f32 + one v_exp_f32 and ~15 scalar-float ops — was 5.4 us of a 37-us
// tile, tools/trace_gemm.py).  With u = |x| and q(u) = 1 - Phi(u) = erfc(u / sqrt 2) / 2:
//     gelu(x) = x Phi(x) = max(x, 0) - u q(u),        q(u) = 2^P(u),  P = degree-7 fit of log2 q on [0, 5.5], P(0) = -1
// ONE transcendental, and the Horner chain + the final ops run as packed fp32 (v_pk_fma_f32: two lanes' worth per issue).
// |gelu - exact| <= 6.5e-7 absolute and <= 5.4e-6 relative on x > -4.5 (fp16 output rounding: 4.9e-4); u is clamped at 5.5,
// beyond which q < 2e-8 (coefficients: Lawson-weighted least squares on 4000 Chebyshev nodes, checked on 400k points in fp32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 u = {fminf(fabsf(x[0]), 5.5f), fminf(fabsf(x[1]), 5.5f)};
  f32x2 P = u * -1.735116371e-06f + 5.974406668e-05f;
  P = P * u + -9.168680408e-04f;
  P = P * u + 8.457269520e-03f;
  P = P * u + -5.386104062e-02f;
  P = P * u + -4.585619271e-01f;
  P = P * u + -1.151209950e+00f;
  P = P * u + -1.0f;
  const f32x2 q = {__builtin_amdgcn_exp2f(P[0]), __builtin_amdgcn_exp2f(P[1])};
  const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return r - u * q;
}

// XCD-aware bijective remap: consecutive tiles (which share the same A row-block) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// tanh-approximate GELU (activations.GELU(approximate="tanh"), Flux FeedForward / proj_mlp):
// 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp_f32 + one v_rcp_f32
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);               // 2u
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u2));
}

// DIT = true adds the MMDiT epilogue forms (Flux): optional tanh-GELU on (acc + bias), the per-sample
// row vector applied as a GATE (multiply) instead of an addend, and a two-region row -> sample map (text rows first,
// image rows second).  It is a compile-time switch so that the UNet kernels keep their code and register budget.
// QKN: compile the fused RMSNorm(q) / RMSNorm(k) + RoPE epilogue (GemmParams::qkn_*; 256x256 MMDiT QKV projections only).  It is
// its own instantiation because its live state (cos / sin rows, norm gains) on top of the gated-residual operands pushed the
// one-size-fits-all MMDiT epilogue over 256 VGPRs (9 spilled, 40 B of scratch per lane in EVERY 256x256 MMDiT GEMM).
// diagnostics build (tools/trace_gemm.sh, -DGDF_TRACE): workgroup time stamps (100 MHz s_memrealtime) at kernel entry, after the
// prologue's DMA issue, when K-tile 0 has landed, after the main loop and after the epilogue, + the CU the workgroup ran on
#if defined(GDF_TRACE)
__device__ unsigned long long gdf_trace[16384 * 8];
#define GDF_TR(i) do { if (threadIdx.x == 0) gdf_trace[(vb & 16383) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GDF_TR_ID() do { if (threadIdx.x == 0) gdf_trace[(vb & 16383) * 8 + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } while (0)
#else
#define GDF_TR(i)
#define GDF_TR_ID()
#endif

// SPLIT: split fp16 hi + lo operands of the opt-in "precise" plans (GemmParams::k_w / a_lo_bytes / o16_lo, kernels.h).  A compile-time
// switch with its own instantiations (gemm_split_kernel): compiled into the default kernels, its few extra live values pushed
// the 256x320 dense kernel from 253 VGPRs to 139 spilled (140 -> 100 img/s on the SDXL step).
// MX: fp8 (OCP e4m3) operands multiplied with v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: the per-row / per-column power-of-two
// scales of the operands are applied to the fp32 accumulators in the epilogue, GemmParams::mx_rowscale / mx_colscale).  A K-tile is 128
// fp8 values = the same 128 bytes per row as 64 halves, so staging, swizzle and the 8-phase schedule are unchanged: the host passes lda /
// K in 2-byte units; a lane's 32-byte fragment of the K = 128 MFMA is the two adjacent 16-byte chunks 2 fk, 2 fk + 1 of its row.
// GNS: the epilogue also emits per-channel GroupNorm partial sums of the stored fp16 image (GemmParams::gn_partial; 3x3 convs of the VAE AND, since round 5, of the UNet op programs on the tiles gemm_gn_slab_rows() accepts)
// Order of the MFMAs of a register tile: "snake" — the column index runs backwards on every other row, so that exactly ONE operand register
// changes between consecutive MFMAs (row-major changes both at every row change).  At the power cap the rate follows the energy:
// tools/micro/energy.hip mfma-order: 1930 (snake) vs 1913 (row-major) vs 1849 TFLOP/s (both operands change every time).  -DGDF_MMA_ROWMAJOR: A/B.
#if defined(GDF_MMA_ROWMAJOR)
#define GDF_SNAKE(row, j, n) (j)
#else
#define GDF_SNAKE(row, j, n) ((((row) & 1) != 0) ? (n) - 1 - (j) : (j))
#endif
template <int MODE, int BM, int BN, int STAGES, bool GEGLU, bool DIT, bool BF = false, bool QKN = false, bool SPLIT = false, bool MX = false,
          bool GNS = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p) {
  static_assert(!BF || DIT, "bf16 operands exist on the MMDiT path only");
  static_assert(!GNS || (!DIT && !GEGLU && !SPLIT && !MX && BN >= 128), "GroupNorm partial sums: plain epilogues, one statistics slab per wave tile (WTM rows)");
  static_assert(!MX || (DIT && STAGES == 8 && !SPLIT && !QKN && !GEGLU), "fp8 operands: the 256x256 two-group MMDiT kernel only");
  constexpr int NW = BM / 32;                    // waves per workgroup (4 or 8)
  // waves along N (the GEGLU form of the 256x320 tile uses 4x2 waves of 64x160: an EVEN number of 16-column fragments,
  // so that every h fragment has its gate fragment in the same lane and register index)
  constexpr int WGN = (BN == 320 && !GEGLU && STAGES != 8) ? 4 : (BN >= 128) ? 2 : 1;
  constexpr int WGM = NW / WGN;                  // waves along M
  constexpr int WTM = BM / WGM;                  // 64 or 32
  constexpr int WTN = BN / WGN;                  // 80, 64 or 16
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_TILE = BM * 128;               // bytes
  constexpr int B_TILE = BN * 128;
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int A_PER_WAVE = BM / 8 / NW;        // 1-KiB wave-instructions of the A tile per wave (4)
  constexpr int B_INSTR = BN / 8;                // 1-KiB wave-instructions per B tile
  constexpr int B_PER_WAVE = (B_INSTR + NW - 1) / NW;
  constexpr int LPT = A_PER_WAVE + B_PER_WAVE;   // DMA instructions per wave per K-tile (uniform when BN == 128)
  static_assert(STAGES == 2 || STAGES == 8 || STAGES == 9 || (STAGES == 3 && B_INSTR % NW == 0), "3-stage ring needs a uniform per-wave load count");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  int tile_m, tile_n;
  if (p.sb_gn > 0) {
    // 2-D super-block order: the workgroups one XCD runs concurrently cover sb_gm x sb_gn tiles, so its private L2
    // fetches sb_gm A panels + sb_gn B panels per round instead of one A panel + a whole row of B panels
    // (N = 10240 GEGLU: 27 MB -> 8.5 MB of L2 fills per XCD and round; the "fixed cost" of that GEMM was this traffic).
    // The super-blocks are dealt to the XCDs in groups of 8; when their number is not a multiple of 8 the last few
    // are walked in plain order (one super-block after the other, round-robin over the XCDs: only that tail loses locality).
    const int conc = p.sb_gm * p.sb_gn;
    const int sbn = tiles_n / p.sb_gn;
    const int nsb = (nblk / conc);
    const int grouped = (nsb >> 3) * 8 * conc;            // workgroups covered by whole groups of 8 super-blocks
    int sb, li;
    if (vb < grouped) {
      const int xcd = vb & 7, j = vb >> 3;
      sb = (j / conc) * 8 + xcd; li = j - (j / conc) * conc;
    } else {
      const int t = vb - grouped;
      sb = (nsb >> 3) * 8 + t / conc; li = t - (t / conc) * conc;
    }
    const int sbr = sb / sbn, sbc = sb - sbr * sbn;
    tile_m = sbr * p.sb_gm + li / p.sb_gn;
    tile_n = sbc * p.sb_gn + li % p.sb_gn;
  } else {
    const int t = xcd_remap(vb, nblk);
    tile_m = t / tiles_n; tile_n = t - tile_m * tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + (size_t)blockIdx.y * p.w_bstride), 0, p.w_bytes, 0x00020000);
  _Float16* const out16 = p.out16 ? p.out16 + (size_t)blockIdx.y * p.o_bstride : nullptr;

  // ---- per-lane load geometry: one wave-instruction moves 8 rows x 128 B ----
  const int lrow = lane >> 3;                         // row inside an 8-row instruction
  const int chunk = (lane & 7) ^ lrow;                // source 16-B chunk (swizzle on the source side)
  uint32_t a_off[A_PER_WAVE];                         // DENSE: byte offset of (row, chunk); CONV3: byte offset of filter tap (0, 0)
  uint32_t a_msk[A_PER_WAVE];                         // CONV3: validity mask of the 9 taps (conv_row below)
  int a_oy[A_PER_WAVE], a_ox[A_PER_WAVE];             // SMALLC: pixel base of the sample in a_off, top-left input pixel here
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int m = m0 + (wave * A_PER_WAVE + j) * 8 + lrow;
    a_msk[j] = 0; a_oy[j] = a_ox[j] = 0;
    if (MODE == A_DENSE) {
      a_off[j] = (m < p.M) ? (uint32_t)m * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u : OOB;
    } else if (MODE == A_CONV3) {
      a_off[j] = 0;                                   // filled by conv_row once its scalars are known
    } else {
      const int hw = p.OH * p.OW;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[j] = (uint32_t)(n * p.H * p.W);           // pixel index base of sample n
      a_oy[j] = (m < p.M) ? oy * p.stride - 1 + p.pad0 : -(1 << 20);   // pad0 = 1: no top / left padding (VAE downsample)
      a_ox[j] = ox * p.stride - 1 + p.pad0;
    }
  }
  const uint32_t ldb = (uint32_t)((SPLIT && p.k_w > 0) ? p.k_w : p.K) * 2u;   // bytes per weight row (split operands: the matrix holds k_w columns, read twice)
  uint32_t b_off[B_PER_WAVE];
  bool b_act[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int q = wave * B_PER_WAVE + j;              // instruction index inside the B tile
    b_act[j] = q < B_INSTR;
    const int n = n0 + q * 8 + lrow;
    b_off[j] = (n < p.N) ? (uint32_t)n * ldb + (uint32_t)chunk * 16u : OOB;
  }

#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 2
  // diagnostics build (tools/ab_epilogue_bound.sh): NO main loop — the prologue / epilogue skeleton with every global load and store of the
  // epilogue, on zero accumulators.  Results are garbage; the time per launch is the epilogue's (+ launch, prologue) alone.
  const int nk = 0;
#else
  const int nk = (MODE == A_CONV_SMALLC) ? 2 : p.K / BK;
#endif
  // split-K (2-stage ring tiles only): this workgroup accumulates the K-tiles [kt0, kt1) and stores raw partial sums
  int kt0 = 0, kt1 = nk;
  if (STAGES == 2 && p.splitk > 1) {
    kt0 = (int)((long)nk * blockIdx.y / p.splitk);
    kt1 = (int)((long)nk * (blockIdx.y + 1) / p.splitk);
  }
  // 3x3 conv: K-tiles are channel-block-major with the nine filter taps innermost (K-tile kt = tap kt % 9 of channel block kt / 9;
  // weights laid out [Cout][Cin / 64][tap][64] by launch_relayout_conv).  Round 2 walked them tap-major: the nine shifted reads of
  // one 64-channel slab were Cin / 64 K-tiles apart, times all resident workgroups of the XCD >> its 4 MB L2, and rocprofv3 counted
  // 1.38 GB of fabric fetches per launch for 134 MB of input + weights (profiles/r02_final_pmc_traffic.json).
  const int IH = p.ups ? 2 * p.H : p.H, IW = p.ups ? 2 * p.W : p.W;

  // byte offset of K-tile kt inside a dense row.  The two-group schedules also stage the tiles nk, nk + 1 (so that the counted
  // waits are the same in every iteration): those get an out-of-range offset — zero fill, no L2 / HBM traffic (scalar select).
  // Split operands (GemmParams::k_w): the lo half of A starts a_lo_bytes after the hi half, the weight K-tiles repeat.
  const int nkw = (SPLIT && p.k_w > 0) ? p.k_w / BK : nk;      // K-tiles of the weight matrix (== nk without a split)
  const uint32_t a_lo = SPLIT ? p.a_lo_bytes : 0u;
  auto koffA = [&](int kt) -> uint32_t {
    if constexpr (!SPLIT) return kt < nk ? (uint32_t)kt * 128u : OOB;
    else return kt < nkw ? (uint32_t)kt * 128u : (kt < nk ? a_lo + (uint32_t)(kt - nkw) * 128u : OOB);
  };
  auto koffB = [&](int kt) -> uint32_t {
    if constexpr (!SPLIT) return kt < nk ? (uint32_t)kt * 128u : OOB;
    else return kt < nk ? (uint32_t)(kt < nkw ? kt : kt - nkw) * 128u : OOB;
  };
  const int cbw = nkw / 9;                                      // conv3: 64-channel blocks of the weight matrix
  auto chanb = [&](int cbk) -> uint32_t {
    if constexpr (!SPLIT) return (uint32_t)cbk * 128u;
    else return cbk < cbw ? (uint32_t)cbk * 128u : a_lo + (uint32_t)(cbk - cbw) * 128u;
  };
  // 3x3-conv rows of the two-group schedules (round 2): per output row ONE byte offset — that of filter tap (0, 0), which may lie
  // outside the image — and a 9-bit validity mask, so that a K-tile's source offset is `base + scalar tap offset` and one bit test
  // (4 VALU instructions per row and K-tile instead of ~12: bounds compares, pixel arithmetic and two multiplies; VALU issue time
  // is not hidden by MFMAs on this hardware).  Nearest-x2 upsampling fused into the conv: source row of tap ky is
  // (uy + ky) >> 1 = (uy >> 1) + {0, parity, 1}[ky]; the two parities travel in mask bits 9 / 10.
  auto conv_row = [&](int m, uint32_t& base, uint32_t& mask) {
    const int hw = p.OH * p.OW;
    const int n = m / hw;
    const int rem = m - n * hw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    const int uy = oy * p.stride - 1 + p.pad0, ux = ox * p.stride - 1 + p.pad0;      // pad0 = 1: no top / left padding (VAE downsample)
    const int by = p.ups ? (uy >> 1) : uy, bx = p.ups ? (ux >> 1) : ux;
    base = (uint32_t)((n * p.H + by) * p.W + bx) * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u;
    uint32_t mk = 0;
    if (m < p.M) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int iy = uy + tp / 3, ix = ux + tp % 3;
        if ((iy >= 0) & (iy < IH) & (ix >= 0) & (ix < IW)) mk |= 1u << tp;
      }
    }
    if (p.ups) mk |= ((uint32_t)(uy & 1) << 9) | ((uint32_t)(ux & 1) << 10);
    mask = mk;
  };
  auto conv_tap_off = [&](int tp, int cbk, uint32_t base, uint32_t mask, bool live = true) -> uint32_t {     // filter tap tp, channel block cbk (scalars)
    const int ky = tp / 3, kx = tp - ky * 3;
    const uint32_t rowb = (uint32_t)p.W * (uint32_t)p.lda * 2u, pixb = (uint32_t)p.lda * 2u;
    uint32_t off;
    if (!p.ups) {
      off = base + ((uint32_t)ky * rowb + (uint32_t)kx * pixb + chanb(cbk));
    } else {
      off = base + ((ky == 2 ? rowb : 0u) + (kx == 2 ? pixb : 0u) + chanb(cbk));
      if (ky == 1 && (mask & 512u)) off += rowb;
      if (kx == 1 && (mask & 1024u)) off += pixb;
    }
    const uint32_t bit = live ? (1u << tp) : 0u;                 // tiles >= nk (over-staged): zero fill
    return (mask & bit) ? off : OOB;
  };
  auto conv_off = [&](int kt, uint32_t base, uint32_t mask) -> uint32_t {
#if defined(GDF_CONV_TAP_MAJOR)                                   // diagnostics build (tools/build_variant.sh): the round-2 K order, for same-box A/Bs
    const int cpb = p.Cin / BK, tp = kt / cpb;
    return conv_tap_off(tp < 9 ? tp : 0, kt - tp * cpb, base, mask, tp < 9);
#else
    const int cbk = kt / 9;                                     // scalar: channel block, then filter tap
    return conv_tap_off(kt - cbk * 9, cbk, base, mask, kt < nk);
#endif
  };
  if (MODE == A_CONV3 && STAGES < 8) {
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) conv_row(m0 + (wave * A_PER_WAVE + j) * 8 + lrow, a_off[j], a_msk[j]);
  }
#if defined(GDF_CONV_TAP_MAJOR)
  const int cpb_ = (MODE == A_CONV3) ? p.Cin / BK : 1;
  int tap = kt0 / cpb_, cb = kt0 - (kt0 / cpb_) * cpb_;
#else
  int cb = kt0 / 9, tap = kt0 - (kt0 / 9) * 9;         // channel block / filter tap of the NEXT tile to issue
#endif
  auto issue = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_TILE;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      uint32_t off;
      if (MODE == A_DENSE) {
        off = a_off[j] + (SPLIT ? koffA(kt) : (uint32_t)kt * 128u);   // OOB stays >= 2^31
      } else if (MODE == A_CONV3) {
        off = conv_tap_off(tap, cb, a_off[j], a_msk[j]);
      } else {  // SMALLC: 8 channels per pixel = one 16-B chunk per tap; chunk index == tap - 8*kt
        const int tp = kt * 8 + chunk;
        const int kyy = tp / 3, kxx = tp - kyy * 3;
        const int iy = a_oy[j] + kyy, ix = a_ox[j] + kxx;
        const bool okk = (tp < 9) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        off = okk ? (a_off[j] + (uint32_t)(iy * p.W + ix)) * 16u : OOB;
      }
      glds16(rsA, sA + (wave * A_PER_WAVE + j) * 1024, off);
    }
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      if (b_act[j]) {
        const uint32_t off = b_off[j] + ((MODE == A_CONV_SMALLC || !SPLIT) ? (uint32_t)kt * 128u : koffB(kt));
        glds16(rsB, sB + (wave * B_PER_WAVE + j) * 1024, off);
      }
    }
#if defined(GDF_CONV_TAP_MAJOR)
    if (MODE == A_CONV3) { if (++cb == cpb_) { cb = 0; ++tap; } }
#else
    if (MODE == A_CONV3) { if (++tap == 9) { tap = 0; ++cb; } }
#endif
  };

  // ---- accumulators ----
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int frow = lane & 15, fk = lane >> 4;

  // One K-tile = two 32-deep MFMA steps (kk = 0, 1): 2 x (FM + FN) ds_read_b128 and 2 x FM x FN MFMAs per wave.
  // A wave issues in order, so its own DMA issue (a `buffer_load ... lds` costs ~60-180 issue cycles) cannot overlap
  // its own MFMAs; the overlap comes from the partner wave on the same SIMD.  With 8 waves the two waves of a SIMD
  // (w, w+4) therefore run the head of a K-tile in opposite orders (EARLY_MMA): one issues the next tile's DMA
  // while the other already multiplies.  (Measured and rejected: rotating the loop by half a tile so that MFMAs from
  // registers follow the barrier, 929 -> 684 TFLOP/s on the 256x320 GEGLU GEMM; a two-group ping-pong with 2 barriers
  // per K-tile, 1002 -> 903 at 8192^3.)
  f16x8 af[FM], bf[FN];
  // lane part of a fragment address per k-step (all wave-tile origins are multiples of 16 rows, so the swizzle term depends on
  // frow only); opaque to the optimiser so that buffer + fragment offsets stay `one add + immediate` instead of an add per fragment
  uint32_t rfa[2], rfb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    rfa[kk] = (uint32_t)((wm * WTM + frow) * 128 + (((kk * 4 + fk) ^ (frow & 7)) << 4));
    rfb[kk] = (uint32_t)(A_TILE + (wn * WTN + frow) * 128 + (((kk * 4 + fk) ^ (frow & 7)) << 4));
    if (STAGES < 8) asm volatile("" : "+v"(rfa[kk]), "+v"(rfb[kk]));
  }
  auto read_kk = [&](int buf, int kk) {
    const char* pa = smem + buf * STAGE + rfa[kk];
    const char* pb = smem + buf * STAGE + rfb[kk];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(pa + i * 2048);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(pb + j * 2048);
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int jj = 0; jj < FN; ++jj) { const int j = GDF_SNAKE(i, jj, FN); acc[i][j] = mfma16<BF>(af[i], bf[j], acc[i][j]); }
  };
  // compile-time off for the 256x320 variant: its 160 accumulator VGPRs leave no room for the second code path
  constexpr bool EARLY_OK = (NW == 8) && (FM * FN <= 16);
  const bool early_mma = EARLY_OK && !p.no_early_mma && (wave >= 4);

  if constexpr (STAGES == 8) {
    // ---- 8-phase schedule (256x256 dense tile, 2 K-tile buffers of 64 KiB) ----
    // The two waves of a SIMD (w, w + 4) belong to two groups that run ONE BARRIER apart: while one group multiplies a
    // quadrant of its 64x128 wave tile (16 MFMAs) the other reads its next fragments from LDS and issues
    // its share of the next half-tile DMA, then they swap (2 barriers per phase, 4 phases per K-tile).  The MFMA pipe of
    // every SIMD therefore always has a wave that is multiplying.  DMA runs 1.5 K-tiles ahead in 16-KiB half-tiles
    // (A rows 0-127 / 128-255, B rows likewise; every wave issues 2 of a half-tile's 16 instructions), the one counted wait
    // per K-tile leaves three half-tiles in flight:
    //   K-tile T (buffer T & 1)   phase 1: read A (all 64 rows) + B cols 0-63     stage B-hi of T+1      MFMA (A0,B0)
    //                             phase 2:                                        stage A-lo of T+2      MFMA (A1,B0)
    //                             phase 3: read B cols 64-127, retire the reads   stage A-hi of T+2      MFMA (A1,B1)
    //                             phase 4: wait vmcnt(6) = tile T+1 has landed    stage B-lo of T+2      MFMA (A0,B1)
    // Slot lifetimes (why each staging is safe): A slots are last read in phase 1 (A-lo by group 0 only, A-hi by group 1
    // only), B slots in phase 3 with the reads retired (lgkmcnt) BEFORE the reader's next barrier; a slot is restaged by
    // a wave that has passed a barrier the last reader arrived at after retiring its reads.  Tiles >= nk are staged too
    // (garbage or zeros, never read) so that the wait count is the same in every iteration.
    static_assert((MODE == A_DENSE || MODE == A_CONV3) && BM == 256 && (BN == 256 || BN == 320) && FM == 4 && FN == BN / 32, "8-phase schedule: 4x2 waves of 64 x BN/2");
    // A half-tile = 128 rows = 16 DMA instructions, 2 per wave.  B half-tile = BN/2 rows: 16 instructions (2 per wave) at
    // BN = 256; 20 at BN = 320: the group whose turn it is (group 0 for B-lo, group 1 for B-hi) issues 3 per wave, the other 2,
    // so every wave issues 5 per B tile and the counted wait is 6 or 7 depending on the group.
    constexpr int FNH = FN / 2;                         // 16-column fragments per B half
    constexpr int BHALF = BN / 2;                       // rows per B half-tile
    constexpr bool B3 = (BN == 320);
    const bool g1 = wave >= 4;
    uint32_t ha[2][2], hb[2][3];                        // ha: DENSE byte offset of (row, chunk); CONV byte offset of filter tap (0, 0)
    uint32_t hm[2][2];                                  // CONV: validity mask of the 9 taps (conv_row)
    int hbq[2];                                         // first instruction index of this wave in B half-tile h
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool big = B3 && (g1 == (h == 1));
      hbq[h] = !B3 ? wave * 2 : (big ? (wave & 3) * 3 : 12 + (wave & 3) * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + h * 128 + (wave * 2 + j) * 8 + lrow;
        if (MODE == A_DENSE) {
          ha[h][j] = (m < p.M) ? (uint32_t)m * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u : OOB;
          hm[h][j] = 0;
        } else {
          conv_row(m, ha[h][j], hm[h][j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r = h * BHALF + (hbq[h] + j) * 8 + lrow;
        hb[h][j] = (n0 + r < p.N) ? (uint32_t)(n0 + r) * ldb + (uint32_t)chunk * 16u : OOB;
      }
    }
    auto stage = [&](int kt, int buf, auto which) {            // which: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
      constexpr int W = decltype(which)::value;
      if constexpr (W < 2) {
        char* base = smem + buf * A_TILE + W * 16384 + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          glds16(rsA, base + j * 1024, MODE == A_DENSE ? ha[W][j] + koffA(kt) : conv_off(kt, ha[W][j], hm[W][j]));
        }
      } else {
        constexpr int H = W - 2;
        char* base = smem + 2 * A_TILE + buf * B_TILE + H * (BHALF * 128) + hbq[H] * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(rsB, base + j * 1024, hb[H][j] + koffB(kt));
        if (B3 && (g1 == (H == 1))) glds16(rsB, base + 2 * 1024, hb[H][2] + koffB(kt));
      }
    };
    // DMA instructions of this wave in the three youngest stagings at the phase-4 wait (A-lo, A-hi, B-lo of tile T+2)
    auto wait_tile = [&]() {
      if (B3 && !g1) wait_vmcnt<7>(); else wait_vmcnt<6>();
    };
    constexpr std::integral_constant<int, 0> ALO{};
    constexpr std::integral_constant<int, 1> AHI{};
    constexpr std::integral_constant<int, 2> BLO{};
    constexpr std::integral_constant<int, 3> BHI{};
    f16x8 a8[4][2], b8[FNH][2];
    // lane part of a fragment address for k-step kk: (first row of the wave tile + frow) * 128 + swizzled 16-byte chunk; the row of
    // fragment i and the ring buffer are compile-time constants -> the ds_read's immediate offset (opaque to the optimiser, or it
    // re-associates them back into per-fragment VGPRs: 26-41 spilled)
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = MX ? 2 * fk + kk : kk * 4 + fk;        // fp8: the two halves of the lane's 32-byte K = 128 fragment
      fa[kk] = (uint32_t)((wm * WTM + frow) * 128 + ((ch ^ (frow & 7)) << 4));
      fb[kk] = (uint32_t)(2 * A_TILE + (wn * WTN + frow) * 128 + ((ch ^ (frow & 7)) << 4));
      asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));
    }
    auto rd_a = [&](const int cur) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a8[i][kk] = *(const f16x8*)(smem + fa[kk] + (cur * A_TILE + i * 2048));
    };
    auto rd_b = [&](const int cur, int half) {
#pragma unroll
      for (int j = 0; j < FNH; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b8[j][kk] = *(const f16x8*)(smem + fb[kk] + (cur * B_TILE + (half * FNH + j) * 2048));
    };
    auto mma_q = [&](auto ah, auto bh) {
      constexpr int AH = decltype(ah)::value, BH = decltype(bh)::value;
      if constexpr (MX) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < FNH; ++jj) {
            const int j = GDF_SNAKE(i, jj, FNH);
            acc[AH * 2 + i][BH * FNH + j] = mfma_mx8(a8[AH * 2 + i][0], a8[AH * 2 + i][1], b8[j][0], b8[j][1], acc[AH * 2 + i][BH * FNH + j]);
          }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < FNH; ++jj) {
              const int j = GDF_SNAKE(i + kk, jj, FNH);
              acc[AH * 2 + i][BH * FNH + j] = mfma16<BF>(a8[AH * 2 + i][kk], b8[j][kk], acc[AH * 2 + i][BH * FNH + j]);
            }
      }
    };
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    constexpr std::integral_constant<int, 0> Q0{};
    constexpr std::integral_constant<int, 1> Q1{};

    stage(0, 0, ALO); stage(0, 0, AHI); stage(0, 0, BLO); stage(0, 0, BHI);
    stage(1, 1, ALO); stage(1, 1, AHI); stage(1, 1, BLO);
    GDF_TR(1);
    wait_tile();                     // this wave's share of K-tile 0
    bar();                           // ... everyone's
    GDF_TR(2);
    if (g1) bar();                   // group 1 runs one barrier behind group 0
#if !defined(GDF_PHASES4)
    // TWO phases of 32 MFMAs per K-tile (round 2; the round-1 schedule below ran four phases of 16).  Per barrier interval one
    // group multiplies while the other reads fragments / issues DMA; the hand-over itself costs ~115 cycles per interval
    // (MFMA + barrier skeleton without reads and DMA: 70 % of the MFMA peak with 16-MFMA clusters, tools/ablate_gemm.py), so
    // twice as long clusters halve that overhead with the SAME registers (A stays resident, the B halves take turns in b8), the
    // same accumulation order (bit-identical results) and the same prefetch depth (the three youngest half-tiles stay in flight
    // at the one counted wait).  Measured, 4 -> 2 phases: Flux QKV 1285 -> 1357, proj_out 1395 -> 1483, 8192^3 1412 -> 1501
    // (hipBLASLt: 1491), SDXL GEGLU shape 1128 -> 1170..1186 TFLOP/s.  Splitting the DMA issue between the read slot and the
    // middle of the MFMA cluster gives the gain back (1285 -> 1293): LDS-DMA issue belongs in the read role.
    //   K-tile T (buffer T & 1)   phase 1: read A (all 64 rows), B cols 0-63      stage B-hi of T+1                  32 MFMAs (A, B-lo)
    //                             phase 2: read B cols 64-127                     stage A-lo, A-hi, B-lo of T+2,     32 MFMAs (A, B-hi)
    //                                                                            wait vmcnt(6) = tile T+1 has landed
    // Slot lifetimes: A-lo is read by group 0 only and A-hi by group 1 only, both in phase 1 — group 1 one barrier after group 0 —
    // and B-lo by both; every read is retired (lgkmcnt) before the reader's next barrier, so all three are free from phase 2's
    // read slot of either group on; B-hi (read in phase 2) is free from the next tile's phase 1 on.
    // Ring layout [A buf 0][A buf 1][B buf 0][B buf 1] and the K loop unrolled by two: the buffer index is a compile-time constant
    // in each copy of the body, so it lands in the 16-bit immediate offset of the ds_read (A_TILE, B_TILE <= 40 KiB) instead of ~20
    // v_add per K-tile that rebuild every fragment address from `cur * STAGE` (VALU issue time adds to MFMA time on this hardware).
    auto ktile = [&](int kt, const int cur) {
      rd_a(cur); rd_b(cur, 0); stage(kt + 1, cur ^ 1, BHI);
      bar(); lgkm0(); mma_q(Q0, Q0); mma_q(Q1, Q0); bar();
      rd_b(cur, 1); stage(kt + 2, cur, ALO); stage(kt + 2, cur, AHI); stage(kt + 2, cur, BLO); wait_tile(); lgkm0();
      bar(); mma_q(Q1, Q1); mma_q(Q0, Q1); bar();
    };
    {
      int kt = 0;
      if constexpr (MODE == A_DENSE) {
        for (; kt + 1 < nk; kt += 2) { ktile(kt, 0); ktile(kt + 1, 1); }
        if (kt < nk) ktile(kt, 0);
      } else {                                                 // conv: the unrolled form spills 6-7 VGPRs; 4 v_add per K-tile instead
        for (; kt < nk; ++kt) ktile(kt, kt & 1);
      }
    }
#else
    auto ktile4 = [&](int kt, const int cur) {
      // phase 1
      rd_a(cur); rd_b(cur, 0); stage(kt + 1, cur ^ 1, BHI);
      bar(); lgkm0(); mma_q(Q0, Q0); bar();
      // phase 2
      stage(kt + 2, cur, ALO);
      bar(); mma_q(Q1, Q0); bar();
      // phase 3
      rd_b(cur, 1); stage(kt + 2, cur, AHI); lgkm0();
      bar(); mma_q(Q1, Q1); bar();
      // phase 4
      stage(kt + 2, cur, BLO); wait_tile();
      bar(); mma_q(Q0, Q1); bar();
    };
    {
      int kt = 0;
      if constexpr (MODE == A_DENSE) {
        for (; kt + 1 < nk; kt += 2) { ktile4(kt, 0); ktile4(kt + 1, 1); }
        if (kt < nk) ktile4(kt, 0);
      } else {
        for (; kt < nk; ++kt) ktile4(kt, kt & 1);
      }
    }
#endif
    if (!g1) bar();
    wait_vmcnt<0>();                 // the over-staged tiles must not land in the epilogue's staging area
  } else if constexpr (STAGES == 9) {
    // ---- 8-phase schedule on the 256x320 tile (2x4 waves of 128x80, dense or 3x3-conv A operand) ----
    // Same two-group ping-pong as STAGES == 8, with the roles of A and B swapped so that the 160 accumulators leave room:
    // a wave keeps ALL of its B fragments (80 columns x 64 K = 10 registers of 8 halves) after phase 1 and reads one quarter
    // of its A rows (32 rows) per phase: 20 MFMAs per phase, 26 ds_read_b128 per K-tile.  DMA units: A_q = the q-th 32-row
    // quarter of BOTH 128-row halves (64 rows, one instruction per wave), B_1 = B rows 0-191 (3 per wave), B_2 = rows
    // 192-319 (2 per wave).  Unit lifetimes in K-tile T: B is read in phase 1 only, A_q in phase q+1 only; every read is
    // retired (lgkmcnt) before the reader's next barrier, so a unit may be restaged from the phase after its read:
    //   phase 1: read B, A_0   stage A_3 of T+1             phase 3: read A_2   stage B_2, A_0 of T+2
    //   phase 2: read A_1      stage B_1 of T+2             phase 4: read A_3   stage A_1, A_2 of T+2,  wait vmcnt(8)
    // (8 = the DMA instructions of phases 2-4: everything staged up to phase 1, i.e. all of tile T+1, has landed).
    static_assert((MODE == A_DENSE || MODE == A_CONV3) && BM == 256 && BN == 320 && FM == 8 && FN == 5 && WGN == 4,
                  "8-phase schedule, 2x4 waves of 128x80");
    // A unit q: this wave's instruction covers rows (wave>>2)*128 + q*32 + (wave&3)*8 + lrow
    uint32_t ua[4];                                     // DENSE: byte offset of (row, chunk); CONV: byte offset of filter tap (0, 0)
    uint32_t um[4];                                     // CONV: validity mask of the 9 taps (conv_row)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + (wave >> 2) * 128 + q * 32 + (wave & 3) * 8 + lrow;
      if (MODE == A_DENSE) {
        ua[q] = (m < p.M) ? (uint32_t)m * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u : OOB;
        um[q] = 0;
      } else {
        conv_row(m, ua[q], um[q]);
      }
    }
    uint32_t ub[5];                                     // B_1: instructions wave*3 + {0,1,2}; B_2: 24 + wave*2 + {0,1}
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int qi = j < 3 ? wave * 3 + j : 24 + wave * 2 + (j - 3);
      const int n = n0 + qi * 8 + lrow;
      ub[j] = (n < p.N) ? (uint32_t)n * ldb + (uint32_t)chunk * 16u : OOB;
    }
    auto stage_a = [&](int kt, int buf, int q) {
      char* dst = smem + buf * A_TILE + ((wave >> 2) * 128 + q * 32 + (wave & 3) * 8) * 128;
      glds16(rsA, dst, MODE == A_DENSE ? ua[q] + koffA(kt) : conv_off(kt, ua[q], um[q]));
    };
    auto stage_b = [&](int kt, int buf, auto part) {
      constexpr int PT = decltype(part)::value;           // 0: B_1 (3 instructions), 1: B_2 (2)
      char* base = smem + 2 * A_TILE + buf * B_TILE;
#pragma unroll
      for (int j = (PT ? 3 : 0); j < (PT ? 5 : 3); ++j) {
        const int qi = j < 3 ? wave * 3 + j : 24 + wave * 2 + (j - 3);
        glds16(rsB, base + qi * 1024, ub[j] + koffB(kt));
      }
    };
    constexpr std::integral_constant<int, 0> B1{};
    constexpr std::integral_constant<int, 1> B2{};
    f16x8 a4[2][2], b10[5][2];
    uint32_t fa[2], fb[2];                              // lane part of a fragment address per k-step (see STAGES == 8)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      fa[kk] = (uint32_t)((wm * WTM + frow) * 128 + (((kk * 4 + fk) ^ (frow & 7)) << 4));
      fb[kk] = (uint32_t)(2 * A_TILE + (wn * WTN + frow) * 128 + (((kk * 4 + fk) ^ (frow & 7)) << 4));
      asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));
    }
    auto rd_aq = [&](const int cur, int q) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a4[i][kk] = *(const f16x8*)(smem + fa[kk] + (cur * A_TILE + (q * 2 + i) * 2048));
    };
    auto rd_ball = [&](const int cur) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b10[j][kk] = *(const f16x8*)(smem + fb[kk] + (cur * B_TILE + j * 2048));
    };
    auto mma_q = [&](auto qq) {
      constexpr int Q = decltype(qq)::value;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 5; ++jj) {
            const int j = GDF_SNAKE(i + kk, jj, 5);
            acc[Q * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[i][kk], b10[j][kk], acc[Q * 2 + i][j], 0, 0, 0);
          }
    };
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::integral_constant<int, 2> P2{};
    constexpr std::integral_constant<int, 3> P3{};
    const bool g1 = wave >= 4;

    // Dense A operand: TWO phases of 40 MFMAs per K-tile, A read by 64-row halves (a8: 16 more VGPRs than the quarter form,
    // 253 in all; the conv form would need 271 and keeps the four-phase schedule).  Same reasoning and same bit-identical
    // results as the 256x256 tile above; measured 4 -> 2 phases: 8192x7680x8192 1390..1426 -> 1471..1478, SDXL qkv 1104..1122 ->
    // 1158..1162, ff_out 1341..1362 -> 1404, attn2_q 1057..1072 -> 1096 TFLOP/s.
    //   phase 1: read B, A rows 0-63 of the wave tile     stage B_2, A_2, A_3 of T+1                        40 MFMAs
    //   phase 2: read A rows 64-127                       stage B_1, A_0, A_1 of T+2, wait vmcnt(5)         40 MFMAs
    // (vmcnt(5): the five instructions just issued may be in flight, everything older — all of tile T+1 — has landed)
#if defined(GDF_PHASES4) || defined(GDF_ABLATE)
    constexpr bool TWO_PHASE = false;
#else
    constexpr bool TWO_PHASE = (MODE == A_DENSE) && !SPLIT;      // (the split-operand form of the two-phase loop spills 139 VGPRs)
#endif
    stage_b(0, 0, B1); stage_b(0, 0, B2); stage_a(0, 0, 0); stage_a(0, 0, 1); stage_a(0, 0, 2); stage_a(0, 0, 3);
    GDF_TR(1);
    if constexpr (TWO_PHASE) {
      stage_b(1, 1, B1); stage_a(1, 1, 0); stage_a(1, 1, 1);
      wait_vmcnt<5>();
    } else {
      stage_b(1, 1, B1); stage_b(1, 1, B2); stage_a(1, 1, 0); stage_a(1, 1, 1); stage_a(1, 1, 2);
      wait_vmcnt<8>();               // this wave's share of K-tile 0
    }
    bar();                           // ... everyone's
    GDF_TR(2);
    if (g1) bar();                   // group 1 runs one barrier behind group 0
    if constexpr (TWO_PHASE) {
      f16x8 a8[4][2];
      auto rd_ah = [&](const int cur, int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) a8[i][kk] = *(const f16x8*)(smem + fa[kk] + (cur * A_TILE + (h * 4 + i) * 2048));
      };
      auto mma_h = [&](auto hh) {
        constexpr int H = decltype(hh)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
              const int j = GDF_SNAKE(i + kk, jj, 5);
              acc[H * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[i][kk], b10[j][kk], acc[H * 4 + i][j], 0, 0, 0);
            }
      };
      auto ktile = [&](int kt, const int cur) {                 // ring [A0][A1][B0][B1], loop unrolled by two: see STAGES == 8
        rd_ball(cur); rd_ah(cur, 0); stage_b(kt + 1, cur ^ 1, B2); stage_a(kt + 1, cur ^ 1, 2); stage_a(kt + 1, cur ^ 1, 3); lgkm0();
        bar(); mma_h(P0); bar();
        rd_ah(cur, 1); stage_b(kt + 2, cur, B1); stage_a(kt + 2, cur, 0); stage_a(kt + 2, cur, 1); wait_vmcnt<5>(); lgkm0();
        bar(); mma_h(P1); bar();
      };
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) { ktile(kt, 0); ktile(kt + 1, 1); }
      if (kt < nk) ktile(kt, 0);
    } else {
#if !defined(GDF_ABLATE)
    auto ktile4 = [&](int kt, const int cur) {                  // ring [A0][A1][B0][B1]: see STAGES == 8
      rd_ball(cur); rd_aq(cur, 0); stage_a(kt + 1, cur ^ 1, 3); lgkm0();
      bar(); mma_q(P0); bar();
      rd_aq(cur, 1); stage_b(kt + 2, cur, B1); lgkm0();
      bar(); mma_q(P1); bar();
      rd_aq(cur, 2); stage_b(kt + 2, cur, B2); stage_a(kt + 2, cur, 0); lgkm0();
      bar(); mma_q(P2); bar();
      rd_aq(cur, 3); stage_a(kt + 2, cur, 1); stage_a(kt + 2, cur, 2); wait_vmcnt<8>(); lgkm0();
      bar(); mma_q(P3); bar();
    };
    {
      int kt = 0;
      if constexpr (MODE == A_DENSE) {
        for (; kt + 1 < nk; kt += 2) { ktile4(kt, 0); ktile4(kt + 1, 1); }
        if (kt < nk) ktile4(kt, 0);
      } else {                                                 // conv: the unrolled form spills 6-7 VGPRs; 4 v_add per K-tile instead
        for (; kt < nk; ++kt) ktile4(kt, kt & 1);
      }
    }
#else
    // ---- diagnostics build (tools/ablate_gemm.sh): the same loop with parts compiled out; results are garbage, timing is the point ----
    //   bit 0: no fragment reads   bit 1: no LDS-DMA   bit 2: no workgroup barriers   bit 3: no MFMAs
    constexpr int ABL = GDF_ABLATE;
    rd_ball(0); rd_aq(0, 0);
    auto keep = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j) { asm volatile("" : "+v"(b10[j][0])); asm volatile("" : "+v"(b10[j][1])); }
#pragma unroll
      for (int i = 0; i < 2; ++i) { asm volatile("" : "+v"(a4[i][0])); asm volatile("" : "+v"(a4[i][1])); }
    };
    auto xbar = [&]() { if constexpr (!(ABL & 4)) bar(); };
    auto xmma = [&](auto q) { if constexpr (!(ABL & 8)) mma_q(q); else keep(); };
    auto ktile_abl = [&](int kt, const int cur) {
      if constexpr (!(ABL & 1)) { rd_ball(cur); rd_aq(cur, 0); } else keep();
      if constexpr (!(ABL & 2)) stage_a(kt + 1, cur ^ 1, 3);
      lgkm0(); xbar(); xmma(P0); xbar();
      if constexpr (!(ABL & 1)) rd_aq(cur, 1); else keep();
      if constexpr (!(ABL & 2)) stage_b(kt + 2, cur, B1);
      lgkm0(); xbar(); xmma(P1); xbar();
      if constexpr (!(ABL & 1)) rd_aq(cur, 2); else keep();
      if constexpr (!(ABL & 2)) { stage_b(kt + 2, cur, B2); stage_a(kt + 2, cur, 0); }
      lgkm0(); xbar(); xmma(P2); xbar();
      if constexpr (!(ABL & 1)) rd_aq(cur, 3); else keep();
      if constexpr (!(ABL & 2)) { stage_a(kt + 2, cur, 1); stage_a(kt + 2, cur, 2); wait_vmcnt<8>(); }
      lgkm0(); xbar(); xmma(P3); xbar();
    };
    {
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) { ktile_abl(kt, 0); ktile_abl(kt + 1, 1); }
      if (kt < nk) ktile_abl(kt, 0);
    }
#endif
    }
    if (!g1) bar();
    wait_vmcnt<0>();                 // the over-staged tiles must not land in the epilogue's staging area
  } else if (STAGES == 2) {
    if (kt0 < kt1) issue(kt0, kt0 & 1);                    // (an empty split-K range stores zeros)
    for (int kt = kt0; kt < kt1; ++kt) {
      // tile kt has landed (all outstanding DMA of this wave) and every wave is done reading buf[(kt+1)&1]
      wait_vmcnt<0>();
      __syncthreads();
      if (early_mma) {
        read_kk(kt & 1, 0); mma();
        if (kt + 1 < kt1) issue(kt + 1, (kt + 1) & 1);
      } else {
        if (kt + 1 < kt1) issue(kt + 1, (kt + 1) & 1);
        read_kk(kt & 1, 0); mma();
      }
      read_kk(kt & 1, 1); mma();
    }
  } else {
    // 3-stage ring: DMA of tiles kt+1 and kt+2 overlaps the MFMAs of tile kt; counted waits (never 0 in steady state)
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();   // this wave's share of tile kt has landed
      __builtin_amdgcn_s_barrier();                                // ... everyone's has; tile kt-1 fully consumed
      int nxt2 = cur + 2; if (nxt2 >= 3) nxt2 -= 3;
      if (early_mma) {
        read_kk(cur, 0); mma();
        if (kt + 2 < nk) issue(kt + 2, nxt2);
      } else {
        if (kt + 2 < nk) issue(kt + 2, nxt2);
        read_kk(cur, 0); mma();
      }
      read_kk(cur, 1); mma();
      if (++cur == 3) cur = 0;
    }
  }
#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 2
  wait_vmcnt<0>();
#endif
  __syncthreads();   // all waves finished reading the last tile: LDS is free for epilogue staging
  GDF_TR(3);
#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 1
  // diagnostics build (tools/ab_epilogue_bound.sh): NO epilogue — the accumulators are kept alive and dropped.  Results are garbage; the
  // time per launch is what a PERFECTLY overlapped epilogue would leave (the bound on any deferred-epilogue scheme).
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
#else

  // ---- epilogue: per-wave staging of 32-row slabs through LDS ----
  // epilogue operands; the QKN instantiation (QKV projection: bias -> RMSNorm + RoPE -> 16-bit store) has none of the
  // residual / row-vector / aux forms, and compiling them out is what keeps it inside the register budget
  const float* const e_res32 = QKN ? nullptr : p.res32;
  const float* const e_rowvec = QKN ? nullptr : p.rowvec;
  const _Float16* const e_res16 = QKN ? nullptr : p.res16;
  _Float16* const e_aux16 = QKN ? nullptr : p.aux16;
  float* const e_out32 = QKN ? nullptr : (STAGES == 2 && p.splitk > 1) ? p.out32 + (size_t)blockIdx.y * p.o32_sstride : p.out32;
  // Every epilogue operand (bias, temb row vector, residual) is fetched BEFORE the staging pass that needs
  // it, so the pass itself is LDS + stores only (a dependent global load per iteration made the first
  // version of this epilogue latency bound: ~17k cycles per tile).
  // GEGLU is evaluated IN REGISTERS before staging: weight rows are interleaved [16 h | 16 gate], fragment 2q holds h and
  // fragment 2q+1 the gate of the same 16 output columns in the same lane / register index -> all 64 lanes busy, half
  // the staging traffic (the first version staged h and gate and ran the GELU on 40 of 64 lanes: 17 us per tile).
  static_assert(!GEGLU || (FN % 2 == 0), "GEGLU needs an even number of column fragments per wave");
  constexpr int FNV = GEGLU ? FN / 2 : FN;             // staged 16-column fragments
  constexpr int WTNV = FNV * 16;                       // staged (= output) columns of this wave tile
  constexpr int SLD = WTNV + 4;                        // padded row length (floats)
  // rows per staging pass: 16 for the 160-accumulator tiles (VGPR budget) and for 256x256 (8 x 32 x 132 floats would not fit the ring)
  constexpr int PR = (FM * FN >= 32) ? 16 : 32;
  constexpr int PASSES = WTM / PR;
  constexpr int FPP = FM / PASSES;                     // 16-row fragments per pass
  float* st = (float*)(smem) + wave * (PR * SLD);
  constexpr int OUTW = WTNV;                           // output columns produced by this wave tile
  constexpr int LPR = OUTW / 8;                        // lanes per row (8 output columns per lane)
  constexpr int RPI = 64 / LPR;                        // rows per iteration (lanes >= RPI*LPR idle when LPR = 5 or 10)
  constexpr int NIT = (PR + RPI - 1) / RPI;            // iterations per pass
  const bool lane_ok = lane < RPI * LPR;
  const int Nout = GEGLU ? p.N / 2 : p.N;
  const int ocol0 = GEGLU ? (n0 + wn * WTN) / 2 : (n0 + wn * WTN);
  const int lc = (lane % LPR) * 8;
  const int col = ocol0 + lc;                          // this lane's 8 output columns (fixed for the whole tile)
  const int nv = (col < Nout) ? ((Nout - col >= 8) ? 8 : (Nout - col)) : 0;
  const bool full = nv == 8;

  const float a_sc = p.acc_scale != 0.f ? p.acc_scale : 1.0f;     // range control of the fp16 images (kernels.h)
  const float o_sc = p.out16_scale != 0.f ? p.out16_scale : 1.0f;
  float mxc[8];                                        // fp8 operands: power-of-two scale of this lane's 8 output columns (weight rows)
#pragma unroll
  for (int e = 0; e < 8; ++e) mxc[e] = 1.f;
  if constexpr (MX) {
    if (full && p.mx_colscale) {
      const f32x4 c0 = *(const f32x4*)(p.mx_colscale + col), c1 = *(const f32x4*)(p.mx_colscale + col + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { mxc[e] = c0[e]; mxc[4 + e] = c1[e]; }
    }
  }
  float bv[8];                                         // bias of this lane's 8 output columns (plain epilogue)
  float bh[FNV], bgt[FNV];                             // GEGLU: bias of this lane's h / gate accumulator column per fragment pair
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
#pragma unroll
  for (int j = 0; j < FNV; ++j) { bh[j] = 0.f; bgt[j] = 0.f; }
  if (p.bias) {
    if (GEGLU) {
#pragma unroll
      for (int j = 0; j < FNV; ++j) {
        const int bcol = n0 + wn * WTN + j * 32 + frow;  // bias is stored in the interleaved GEMM column order
        if (bcol + 16 < p.N) { bh[j] = p.bias[bcol]; bgt[j] = p.bias[bcol + 16]; }
      }
    } else if (full) {
      const f32x4 a0 = *(const f32x4*)(p.bias + col), a1 = *(const f32x4*)(p.bias + col + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bv[e] = a0[e]; bv[4 + e] = a1[e]; }
    } else if (BN == 16) {
      for (int e = 0; e < nv; ++e) bv[e] = p.bias[col + e];
    }
  }

  // Uniform epilogue flags are tested OUTSIDE the per-iteration loops (one scalar branch per flag and pass; the
  // first version branched inside every unrolled iteration: ~800 basic blocks, no overlap between iterations).
  constexpr bool RAGGED = (BN == 16);                    // only the narrow-N variant handles N % 8 != 0 (host-checked)
  const bool rv_in_opnd = e_rowvec && !e_res32;
  auto sample_of = [&](int row) -> int {                 // row of the per-sample vector table that applies to `row`
    if (DIT && p.rv_seg_rows > 0 && row >= p.rv_seg_rows) return (row - p.rv_seg_rows) / p.rv_rps2;
    if (DIT && p.rv_tok) return row % p.rows_per_sample;
    return row / p.rows_per_sample;
  };
  float gsum[GNS ? 8 : 1], gsq[GNS ? 8 : 1];           // GNS: sum x / sum x^2 of this lane's 8 columns over the rows it stores
#pragma unroll
  for (int e = 0; e < (GNS ? 8 : 1); ++e) gsum[e] = gsq[e] = 0.f;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    // ---- prefetch this pass's residual / row-vector operands (overlaps the LDS staging below) ----
    // `opnd` holds the fp32 residual, or the temb row vector when there is no fp32 residual (the plan never
    // combines the two: conv1 = bias + temb, conv2 / out-projections = bias + residual).
    f32x4 opnd[NIT][2];                                  // an fp16 residual travels as raw bits in opnd[it][0]
    int rowi[NIT];
    bool okr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int lrp = it * RPI + lane / LPR;
      rowi[it] = m0 + wm * WTM + ps * PR + lrp;
      okr[it] = (RAGGED ? nv > 0 : full) && lane_ok && lrp < PR && rowi[it] < p.M;
      opnd[it][0] = opnd[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!RAGGED) {
      if (e_res32) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            const f32x4* rp = (const f32x4*)(e_res32 + (size_t)rowi[it] * p.ldres + col);
            opnd[it][0] = rp[0]; opnd[it][1] = rp[1];
          }
      } else if (e_rowvec) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            const f32x4* rv = (const f32x4*)(e_rowvec + (size_t)sample_of(rowi[it]) * p.ldrv + col);
            opnd[it][0] = rv[0]; opnd[it][1] = rv[1];
          }
      } else if (e_res16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) opnd[it][0] = *(const f32x4*)(e_res16 + (size_t)rowi[it] * p.ldres + col);
      }
    }
#pragma unroll
    for (int i2 = 0; i2 < FPP; ++i2)
#pragma unroll
      for (int j = 0; j < FNV; ++j)
        if constexpr (GEGLU) {
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 hh = {acc[ps * FPP + i2][2 * j][r], acc[ps * FPP + i2][2 * j][r + 1]};
            const f32x2 gg = {acc[ps * FPP + i2][2 * j + 1][r], acc[ps * FPP + i2][2 * j + 1][r + 1]};
            const f32x2 x = (hh + bh[j]) * gelu_erf2(gg + bgt[j]);
            st[(i2 * 16 + fk * 4 + r) * SLD + j * 16 + frow] = x[0];
            st[(i2 * 16 + fk * 4 + r + 1) * SLD + j * 16 + frow] = x[1];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) st[(i2 * 16 + fk * 4 + r) * SLD + j * 16 + frow] = acc[ps * FPP + i2][j][r] * a_sc;
        }
    // same-wave LDS RAW across lanes: DS ops of one wave execute in order
    __builtin_amdgcn_wave_barrier();
    float v[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int lr = it * RPI + lane / LPR;
      if (!(lane_ok && lr < PR)) lr = 0;
      const f32x4 x0 = *(const f32x4*)(st + lr * SLD + lc), x1 = *(const f32x4*)(st + lr * SLD + lc + 4);
      if constexpr (MX) {                                // undo the operand scales: row (activation) x column (weight row)
        const float rs = (okr[it] && p.mx_rowscale) ? p.mx_rowscale[rowi[it]] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[it][e] = x0[e] * (rs * mxc[e]) + bv[e]; v[it][4 + e] = x1[e] * (rs * mxc[4 + e]) + bv[4 + e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[it][e] = x0[e] + bv[e]; v[it][4 + e] = x1[e] + bv[4 + e]; }
      }
    }
    if (DIT && p.act == 1) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = gelu_tanh(v[it][e]);
    }
    if constexpr (DIT && WTN == 128 && QKN) {
      // RMSNorm per head + rotary embedding on the q / k columns: a wave tile is exactly one 128-column head, whose row lives in
      // the 16 lanes of one staged row (8 consecutive columns = 4 rotary pairs per lane)
      if (p.qkn_nq > 0 && ocol0 < 2 * p.qkn_nq) {
        const float* nw = (ocol0 < p.qkn_nq ? p.qkn_wq : p.qkn_wk) + lc;
        const f32x4 w0 = *(const f32x4*)nw, w1 = *(const f32x4*)(nw + 4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += v[it][e] * v[it][e];
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);       // the 16 lanes of this row
          const float r = rsqrtf(ss * (1.0f / 128.0f) + p.qkn_eps);
          const int row = rowi[it];
          const int pos = (p.qkn_seg_rows > 0 && row >= p.qkn_seg_rows) ? p.qkn_pos1 + (row - p.qkn_seg_rows) % p.qkn_rps2
                                                                        : p.qkn_pos0 + row % p.qkn_rps;
          f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
          if (okr[it]) {
            const float* cp = p.rope_cos + (size_t)pos * 128 + lc;
            const float* sp = p.rope_sin + (size_t)pos * 128 + lc;
            c0 = *(const f32x4*)cp; c1 = *(const f32x4*)(cp + 4); s0 = *(const f32x4*)sp; s1 = *(const f32x4*)(sp + 4);
          }
          float t[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { t[e] = v[it][e] * r * w0[e]; t[4 + e] = v[it][4 + e] * r * w1[e]; }
          // x * cos + stack([-x_imag, x_real]) * sin
          v[it][0] = t[0] * c0[0] - t[1] * s0[0]; v[it][1] = t[1] * c0[1] + t[0] * s0[1];
          v[it][2] = t[2] * c0[2] - t[3] * s0[2]; v[it][3] = t[3] * c0[3] + t[2] * s0[3];
          v[it][4] = t[4] * c1[0] - t[5] * s1[0]; v[it][5] = t[5] * c1[1] + t[4] * s1[1];
          v[it][6] = t[6] * c1[2] - t[7] * s1[2]; v[it][7] = t[7] * c1[3] + t[6] * s1[3];
        }
      }
    }
    const bool aux_early = DIT && p.rv_mul && e_aux16;     // MMDiT `attn-out` hook: the projection BEFORE the gate
    if (!RAGGED && aux_early) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (okr[it]) {
          f16x8 hv;
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] = hook16<DIT>(v[it][e]);
          *(f16x8*)(e_aux16 + (size_t)rowi[it] * p.ldaux + col) = hv;
        }
    }
    if (!RAGGED) {
      if (rv_in_opnd) {
        if (DIT && p.rv_mul) {
#pragma unroll
          for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] *= opnd[it][0][e]; v[it][4 + e] *= opnd[it][1][e]; }
        } else {
#pragma unroll
          for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] += opnd[it][0][e]; v[it][4 + e] += opnd[it][1][e]; }
        }
      } else if (e_rowvec) {                             // row vector AND fp32 residual (MMDiT gate + residual): late load
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            const f32x4* rv = (const f32x4*)(e_rowvec + (size_t)sample_of(rowi[it]) * p.ldrv + col);
            if (DIT && p.rv_mul) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[it][e] *= rv[0][e]; v[it][4 + e] *= rv[1][e]; }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[it][e] += rv[0][e]; v[it][4 + e] += rv[1][e]; }
            }
          }
      }
      if (e_aux16 && !aux_early) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = hook16<DIT>(v[it][e]);
            *(f16x8*)(e_aux16 + (size_t)rowi[it] * p.ldaux + col) = hv;
          }
      }
      if (e_res32) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[it][e] += opnd[it][0][e]; v[it][4 + e] += opnd[it][1][e]; }
      } else if (e_res16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          f16x8 rh = __builtin_bit_cast(f16x8, opnd[it][0]);
          if (e_rowvec && okr[it]) rh = *(const f16x8*)(e_res16 + (size_t)rowi[it] * p.ldres + col);   // (not produced by the plan)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] += (float)rh[e];
        }
      }
      if (out16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            f16x8 hv;
            if (DIT && BF && p.out_f16) {                // 'bfloat16x2' plans: the attention operands q / k / v as saturating fp16
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = f32_to_f16_sat(v[it][e] * o_sc);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = act16<DIT, BF>(v[it][e] * o_sc);
            }
            *(f16x8*)(out16 + (size_t)rowi[it] * p.ldo16 + col) = hv;
            if constexpr (GNS) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { const float x = v[it][e] * o_sc; gsum[e] += x; gsq[e] += x * x; }
            }
          }
        if (SPLIT && p.o16_lo > 0) {                     // split operand for the consumer GEMM: lo = e16(v - hi)
#pragma unroll
          for (int it = 0; it < NIT; ++it)
            if (okr[it]) {
              f16x8 lv;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float x = v[it][e] * o_sc;
                if constexpr (BF) lv[e] = __builtin_bit_cast(_Float16, (__bf16)(x - (float)(__bf16)x));
                else lv[e] = (_Float16)(x - (float)(_Float16)x);
              }
              *(f16x8*)(out16 + (size_t)rowi[it] * p.ldo16 + col + p.o16_lo) = lv;
            }
        }
      }
      if (e_out32) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (okr[it]) {
            f32x4* op = (f32x4*)(e_out32 + (size_t)rowi[it] * p.ldo32 + col);
            op[0] = f32x4{v[it][0], v[it][1], v[it][2], v[it][3]};
            op[1] = f32x4{v[it][4], v[it][5], v[it][6], v[it][7]};
          }
      }
    } else {                                             // narrow / ragged N (conv_out, N = 4): scalar path
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (okr[it]) {
          const int row = rowi[it];
          for (int e = 0; e < nv; ++e) {
            float x = v[it][e];
            if (e_rowvec) x += e_rowvec[(size_t)(row / p.rows_per_sample) * p.ldrv + col + e];
            if (e_aux16) e_aux16[(size_t)row * p.ldaux + col + e] = (_Float16)x;
            if (e_res32) x += e_res32[(size_t)row * p.ldres + col + e];
            else if (e_res16) x += (float)e_res16[(size_t)row * p.ldres + col + e];
            if (out16) out16[(size_t)row * p.ldo16 + col + e] = (_Float16)x;
            if (e_out32) e_out32[(size_t)row * p.ldo32 + col + e] = x;
          }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);                   // keep the next pass's prefetch from being hoisted (VGPR pressure)
  }
  if constexpr (GNS) {
    // the wave tile is one 64-row statistics slab: combine the RPI lanes that hold the same 8 columns through the (idle) staging
    // rows of this wave, then one 64-byte store per column chunk: gn_partial[slab][col .. col+7][sum, sum of squares]
    static_assert((WTM == 64 || WTM == 128) && PR * SLD >= 64 * 16, "one slab per wave tile");
#pragma unroll
    for (int e = 0; e < 8; ++e) { st[lane * 16 + e] = gsum[e]; st[lane * 16 + 8 + e] = gsq[e]; }
    __builtin_amdgcn_wave_barrier();
    // (M % WTM == 0 is required, so a wave tile lies entirely inside or entirely outside the matrix: in the last M tile of a 128- / 256-row
    //  workgroup the waves whose 64 rows start at or beyond M own NO slab and must not store — M/64 slabs are allocated)
    if (lane < LPR && full && p.gn_partial && m0 + wm * WTM < p.M) {
      float a[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
      for (int r = 0; r < RPI; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] += st[(r * LPR + lane) * 16 + e];
      const int slab = (m0 + wm * WTM) / WTM;            // slabs of WTM rows: 64 (128x128, 128x160, 256x128, 256x256 tiles) or 128 (256x320: 2 x 4 waves of 128 x 80)
      f32x4* gp = (f32x4*)(p.gn_partial + ((size_t)slab * p.N + col) * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) gp[q] = f32x4{a[2 * q], a[8 + 2 * q], a[2 * q + 1], a[8 + 2 * q + 1]};
    }
    __builtin_amdgcn_wave_barrier();
  }
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
