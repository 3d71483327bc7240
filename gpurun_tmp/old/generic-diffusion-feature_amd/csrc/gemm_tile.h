// Tile geometry and operand address generation of the GEMM family (csrc/gemm.hip).
//
// GemmTile<...> carries (a) the compile-time geometry of one tile variant — waves per workgroup, wave tile, fragments, LDS image sizes, DMA
// instructions per wave — and (b) the run-time state of one workgroup's visit to one output tile: lane / wave coordinates, tile origin,
// buffer descriptors, the K range, and the byte-offset generators of the A / B operands (dense rows, 3x3-conv taps with zero padding,
// stride 2 and the fused nearest-x2 upsample, split hi | lo operands).  The main loops (gemm_mainloop_*.h) and the epilogue (gemm_epilogue.h)
// are free function templates over this type; everything is force-inlined, so the struct dissolves into registers.
// Reference ops: see the header of gemm.hip.
#pragma once
#include "gemm_common.h"

namespace gdf {

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
template <int MODE_, int BM_, int BN_, int STAGES_, bool GEGLU_, bool DIT_, bool BF_, bool QKN_, bool SPLIT_, bool MX_, bool GNS_>
struct GemmTile {
  static constexpr int MODE = MODE_, BM = BM_, BN = BN_, STAGES = STAGES_;
  static constexpr bool GEGLU = GEGLU_, DIT = DIT_, BF = BF_, QKN = QKN_, SPLIT = SPLIT_, MX = MX_, GNS = GNS_;
  static_assert(!BF || DIT, "bf16 operands exist on the MMDiT path only");
  static_assert(!GNS || (!DIT && !GEGLU && !SPLIT && !MX && BN >= 128), "GroupNorm partial sums: plain epilogues, one statistics slab per wave tile (WTM rows)");
  static_assert(!MX || (DIT && STAGES == 8 && !SPLIT && !QKN && !GEGLU), "fp8 operands: the 256x256 two-group MMDiT kernel only");
  static constexpr int NW = BM / 32;                    // waves per workgroup (4 or 8)
  // waves along N (the GEGLU form of the 256x320 tile uses 4x2 waves of 64x160: an EVEN number of 16-column fragments,
  // so that every h fragment has its gate fragment in the same lane and register index)
  static constexpr int WGN = (BN == 320 && !GEGLU && STAGES != 8) ? 4 : (BN >= 128) ? 2 : 1;
  static constexpr int WGM = NW / WGN;                  // waves along M
  static constexpr int WTM = BM / WGM;                  // 64 or 32
  static constexpr int WTN = BN / WGN;                  // 80, 64 or 16
  static constexpr int FM = WTM / 16, FN = WTN / 16;
  static constexpr int A_TILE = BM * 128;               // bytes
  static constexpr int B_TILE = BN * 128;
  static constexpr int STAGE = A_TILE + B_TILE;
  static constexpr int A_PER_WAVE = BM / 8 / NW;        // 1-KiB wave-instructions of the A tile per wave (4)
  static constexpr int B_INSTR = BN / 8;                // 1-KiB wave-instructions per B tile
  static constexpr int B_PER_WAVE = (B_INSTR + NW - 1) / NW;
  static constexpr int LPT = A_PER_WAVE + B_PER_WAVE;   // DMA instructions per wave per K-tile (uniform when BN == 128)
  static_assert(STAGES == 2 || STAGES == 8 || STAGES == 9 || (STAGES == 3 && B_INSTR % NW == 0), "3-stage ring needs a uniform per-wave load count");

  // ---- run-time state of one tile visit ----
  const GemmParams& p;
  char* const smem;
  const int tid, lane, wave;
  // (the lane constants below are recomputed by locate() for every tile instead of living in registers across a persistent walk: the
  //  two-group kernels have no room for loop-carried values)
  int lrow;                                           // row inside an 8-row DMA instruction
  int chunk;                                          // source 16-B chunk (swizzle on the source side)
  int wm, wn;                                         // wave coordinates inside the workgroup tile
  int frow, fk;                                       // lane coordinates inside a 16 x 16 x 32 MFMA fragment
  int vb;                                             // virtual workgroup index (persistent walks advance it)
  int m0, n0;                                         // tile origin
  __amdgpu_buffer_rsrc_t rsA, rsB;
  _Float16* out16;
  uint32_t ldb;                                       // bytes per weight row (split operands: the matrix holds k_w columns, read twice)
  int nk, kt0, kt1;                                   // K-tiles of the launch; split-K: this workgroup's range [kt0, kt1)
  int IH, IW;                                         // conv source size after the fused nearest-x2 upsample
  int nkw;                                            // K-tiles of the weight matrix (== nk without a split)
  uint32_t a_lo;                                      // split operands: byte offset of the lo half of A
  int cbw;                                            // conv3: 64-channel blocks of the weight matrix

  __device__ __forceinline__ GemmTile(const GemmParams& p_, char* smem_)
      : p(p_), smem(smem_), tid(threadIdx.x), lane(threadIdx.x & 63), wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {}

  // origin of virtual workgroup `vb_`'s tile (2-D super-block order / XCD-chunked linear order), buffer descriptors, K range
  __device__ __forceinline__ void locate(int vb_, int tiles_n, int nblk) {
    vb = vb_;
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
    m0 = tile_m * BM; n0 = tile_n * BN;
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + (size_t)blockIdx.y * p.w_bstride), 0, p.w_bytes, 0x00020000);
    out16 = p.out16 ? p.out16 + (size_t)blockIdx.y * p.o_bstride : nullptr;
    lrow = lane >> 3;
    chunk = (lane & 7) ^ lrow;
    ldb = (uint32_t)((SPLIT && p.k_w > 0) ? p.k_w : p.K) * 2u;
#if defined(GDF_ABLATE_EPI) && GDF_ABLATE_EPI == 2
    // diagnostics build (tools/ab_epilogue_bound.sh): NO main loop — the prologue / epilogue skeleton with every global load and store of the
    // epilogue, on zero accumulators.  Results are garbage; the time per launch is the epilogue's (+ launch, prologue) alone.
    nk = 0;
#else
    nk = (MODE == A_CONV_SMALLC) ? 2 : p.K / BK;
#endif
    // split-K (2-stage ring tiles only): this workgroup accumulates the K-tiles [kt0, kt1) and stores raw partial sums
    kt0 = 0; kt1 = nk;
    if (STAGES == 2 && p.splitk > 1) {
      kt0 = (int)((long)nk * blockIdx.y / p.splitk);
      kt1 = (int)((long)nk * (blockIdx.y + 1) / p.splitk);
    }
    // 3x3 conv: K-tiles are channel-block-major with the nine filter taps innermost (K-tile kt = tap kt % 9 of channel block kt / 9;
    // weights laid out [Cout][Cin / 64][tap][64] by launch_relayout_conv).  Round 2 walked them tap-major: the nine shifted reads of
    // one 64-channel slab were Cin / 64 K-tiles apart, times all resident workgroups of the XCD >> its 4 MB L2, and rocprofv3 counted
    // 1.38 GB of fabric fetches per launch for 134 MB of input + weights (profiles/r02_final_pmc_traffic.json).
    IH = p.ups ? 2 * p.H : p.H; IW = p.ups ? 2 * p.W : p.W;
    nkw = (SPLIT && p.k_w > 0) ? p.k_w / BK : nk;
    a_lo = SPLIT ? p.a_lo_bytes : 0u;
    cbw = nkw / 9;
    wm = wave / WGN; wn = wave - wm * WGN;
    frow = lane & 15; fk = lane >> 4;
  }


  // byte offset of K-tile kt inside a dense row.  The two-group schedules also stage the tiles nk, nk + 1 (so that the counted
  // waits are the same in every iteration): those get an out-of-range offset — zero fill, no L2 / HBM traffic (scalar select).
  // Split operands (GemmParams::k_w): the lo half of A starts a_lo_bytes after the hi half, the weight K-tiles repeat.
  __device__ __forceinline__ uint32_t koffA(int kt) const {
    if constexpr (!SPLIT) return kt < nk ? (uint32_t)kt * 128u : OOB;
    else return kt < nkw ? (uint32_t)kt * 128u : (kt < nk ? a_lo + (uint32_t)(kt - nkw) * 128u : OOB);
  }
  __device__ __forceinline__ uint32_t koffB(int kt) const {
    if constexpr (!SPLIT) return kt < nk ? (uint32_t)kt * 128u : OOB;
    else return kt < nk ? (uint32_t)(kt < nkw ? kt : kt - nkw) * 128u : OOB;
  }
  __device__ __forceinline__ uint32_t chanb(int cbk) const {
    if constexpr (!SPLIT) return (uint32_t)cbk * 128u;
    else return cbk < cbw ? (uint32_t)cbk * 128u : a_lo + (uint32_t)(cbk - cbw) * 128u;
  }
  // 3x3-conv rows of the two-group schedules (round 2): per output row ONE byte offset — that of filter tap (0, 0), which may lie
  // outside the image — and a 9-bit validity mask, so that a K-tile's source offset is `base + scalar tap offset` and one bit test
  // (4 VALU instructions per row and K-tile instead of ~12: bounds compares, pixel arithmetic and two multiplies; VALU issue time
  // is not hidden by MFMAs on this hardware).  Nearest-x2 upsampling fused into the conv: source row of tap ky is
  // (uy + ky) >> 1 = (uy >> 1) + {0, parity, 1}[ky]; the two parities travel in mask bits 9 / 10.
  __device__ __forceinline__ void conv_row(int m, uint32_t& base, uint32_t& mask) const {
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
  }
  __device__ __forceinline__ uint32_t conv_tap_off(int tp, int cbk, uint32_t base, uint32_t mask, bool live = true) const {     // filter tap tp, channel block cbk (scalars)
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
  }
  __device__ __forceinline__ uint32_t conv_off(int kt, uint32_t base, uint32_t mask) const {
#if defined(GDF_CONV_TAP_MAJOR)                                   // diagnostics build (tools/build_variant.sh): the round-2 K order, for same-box A/Bs
    const int cpb = p.Cin / BK, tp = kt / cpb;
    return conv_tap_off(tp < 9 ? tp : 0, kt - tp * cpb, base, mask, tp < 9);
#else
    const int cbk = kt / 9;                                     // scalar: channel block, then filter tap
    return conv_tap_off(kt - cbk * 9, cbk, base, mask, kt < nk);
#endif
  };
};

// The pieces of the kernel (main loops, epilogue) are written against local names; these three lines give a function template over a GemmTile
// type T / object t the tile's compile-time geometry, its per-visit state and its operand address generators under those names.
#define GDF_TILE_GEOMETRY(T)                                                                                                                    \
  [[maybe_unused]] constexpr int MODE = T::MODE, BM = T::BM, BN = T::BN, STAGES = T::STAGES;                                                   \
  [[maybe_unused]] constexpr bool GEGLU = T::GEGLU, DIT = T::DIT, BF = T::BF, QKN = T::QKN, SPLIT = T::SPLIT, MX = T::MX, GNS = T::GNS;        \
  [[maybe_unused]] constexpr int NW = T::NW, WGN = T::WGN, WGM = T::WGM, WTM = T::WTM, WTN = T::WTN, FM = T::FM, FN = T::FN;                  \
  [[maybe_unused]] constexpr int A_TILE = T::A_TILE, B_TILE = T::B_TILE, STAGE = T::STAGE, A_PER_WAVE = T::A_PER_WAVE, B_INSTR = T::B_INSTR,   \
                                 B_PER_WAVE = T::B_PER_WAVE, LPT = T::LPT
#define GDF_TILE_STATE(t)                                                                                                                       \
  [[maybe_unused]] const GemmParams& p = t.p;                                                                                                   \
  [[maybe_unused]] char* const smem = t.smem;                                                                                                   \
  [[maybe_unused]] const int lane = t.lane, wave = t.wave, lrow = t.lrow, chunk = t.chunk, wm = t.wm, wn = t.wn, frow = t.frow, fk = t.fk;     \
  [[maybe_unused]] const int vb = t.vb, m0 = t.m0, n0 = t.n0, nk = t.nk, kt0 = t.kt0, kt1 = t.kt1;                                             \
  [[maybe_unused]] const uint32_t ldb = t.ldb;                                                                                                  \
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsA = t.rsA, rsB = t.rsB
#define GDF_TILE_OPERANDS(t)                                                                                                                    \
  [[maybe_unused]] auto koffA = [&](int kt) -> uint32_t { return t.koffA(kt); };                                                                \
  [[maybe_unused]] auto koffB = [&](int kt) -> uint32_t { return t.koffB(kt); };                                                                \
  [[maybe_unused]] auto conv_row = [&](int m, uint32_t& base, uint32_t& mask) { t.conv_row(m, base, mask); };                                  \
  [[maybe_unused]] auto conv_tap_off = [&](int tp, int cbk, uint32_t base, uint32_t mask) -> uint32_t { return t.conv_tap_off(tp, cbk, base, mask); }; \
  [[maybe_unused]] auto conv_off = [&](int kt, uint32_t base, uint32_t mask) -> uint32_t { return t.conv_off(kt, base, mask); }

}  // namespace gdf
