// The two-group ("8-phase") main loops of the GEMM family (csrc/gemm.hip): 8 waves, the two waves of a SIMD in two groups that run one workgroup
// barrier apart — one multiplies while the other reads fragments from LDS and issues LDS-DMA — with half- / quarter-tile DMA 1.5 K-tiles ahead and
// ONE counted `s_waitcnt vmcnt(N)` per K-tile.  gemm_mainloop_8phase_256: the 256x256 tile (4x2 waves of 64x128, A resident; dense, 3x3 conv,
// GEGLU, MMDiT incl. fp8); gemm_mainloop_8phase_320: the 256x320 tile (2x4 waves of 128x80, B resident; dense and 3x3 conv).
#pragma once
#include "gemm_tile.h"

namespace gdf {

template <class T>
__device__ __forceinline__ void gemm_mainloop_8phase_256(T& t, f32x4 (&acc)[T::FM][T::FN]) {
  GDF_TILE_GEOMETRY(T);
  GDF_TILE_STATE(t);
  GDF_TILE_OPERANDS(t);
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
}

template <class T>
__device__ __forceinline__ void gemm_mainloop_8phase_320(T& t, f32x4 (&acc)[T::FM][T::FN]) {
  GDF_TILE_GEOMETRY(T);
  GDF_TILE_STATE(t);
  GDF_TILE_OPERANDS(t);
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
}

}  // namespace gdf
