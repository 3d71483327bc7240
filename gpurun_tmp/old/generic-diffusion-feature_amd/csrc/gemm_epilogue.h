// Epilogue of the GEMM family (csrc/gemm.hip): the accumulators of a wave tile are staged through LDS in 16- / 32-row passes so that every
// global store / residual load is a full 16-byte-per-lane, 128-byte-per-row access.  Forms (GemmParams): bias, per-sample row vector (addend or
// MMDiT gate), fp32 / fp16 residual, GEGLU gate in registers, fp16 / bf16 / saturating stores, split hi | lo output pairs, pre-residual hook copy
// (aux16), fp32 master store, fused RMSNorm + RoPE (QKN), fp8 operand scales (MX), GroupNorm partial sums of the stored image (GNS).
#pragma once
#include "gemm_tile.h"

namespace gdf {

template <class T>
__device__ __forceinline__ void gemm_epilogue(T& t, f32x4 (&acc)[T::FM][T::FN]) {
  GDF_TILE_GEOMETRY(T);
  GDF_TILE_STATE(t);
  _Float16* const out16 = t.out16;
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
}

}  // namespace gdf
