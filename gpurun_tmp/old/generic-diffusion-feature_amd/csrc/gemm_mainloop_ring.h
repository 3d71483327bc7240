// Main loops of the LDS-ring tiles of the GEMM family (csrc/gemm.hip): 2-stage ring (128x128 / 128x160 / 128x16 with 4 waves and two workgroups per
// CU, split-K; and the 2-stage form of the 256-row tiles, the bit-exact reference of the 8-phase loops) and the 3-stage ring of the 256x128 tile
// (8 waves, counted `s_waitcnt vmcnt(N)`, one raw `s_barrier` per K-tile).  Both operands stream HBM -> LDS with `buffer_load ... lds`.
#pragma once
#include "gemm_tile.h"

namespace gdf {

template <class T>
__device__ __forceinline__ void gemm_mainloop_ring(T& t, f32x4 (&acc)[T::FM][T::FN]) {
  GDF_TILE_GEOMETRY(T);
  GDF_TILE_STATE(t);
  GDF_TILE_OPERANDS(t);
  // ---- per-lane load geometry: one wave-instruction moves 8 rows x 128 B ----
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
  uint32_t b_off[B_PER_WAVE];
  bool b_act[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int q = wave * B_PER_WAVE + j;              // instruction index inside the B tile
    b_act[j] = q < B_INSTR;
    const int n = n0 + q * 8 + lrow;
    b_off[j] = (n < p.N) ? (uint32_t)n * ldb + (uint32_t)chunk * 16u : OOB;
  }
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

  if (STAGES == 2) {
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
}

}  // namespace gdf
