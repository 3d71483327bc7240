// Flash attention forward for gfx950 (MI355X): fp16 Q/K/V, fp32 online softmax, fp16 output.
//
// Replaces, for the UNet hot path, `F.scaled_dot_product_attention(q, k, v)` in
// AttnProcessor2_0.__call__ (/root/reference/feature/diffusers/models/attention_processor.py:3311-3313,
// scale = dim_head**-0.5 at :166) for self attention (Sk = H*W) and cross attention (Sk = 77 text tokens).
//
// Layout: 4 waves x 32 query rows per workgroup, 64-key K/V tiles staged in LDS.
//   * scores are computed TRANSPOSED, S^T = K Q^T with mfma_f32_32x32x16_f16, so each lane owns one
//     query column: row max / row sum / rescale are lane-local (one cross-half exchange for the max).
//   * P never leaves registers: the S^T accumulator layout is re-used as the B operand of
//     O^T = V^T P^T by relabelling the key index inside a 16-key step (the contraction is
//     permutation invariant), and V^T fragments are produced by the gfx950 LDS transpose read
//     `ds_read_b64_tr_b16` from a row-major V tile.
//   * head dims 40/64/80/160 (SD1.5: 40,80,160; SDXL: 64) are zero-padded to MFMA granularity in LDS.
#include "kernels.h"
#include <type_traits>

namespace gdf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define LDS_AS __attribute__((address_space(3)))

static constexpr int KT = 64;     // keys per tile
#if !defined(GDF_ATTN_PV16_DEFAULT)
#define GDF_ATTN_PV16_DEFAULT true    // P V on mfma_f32_16x16x32_f16 for D = 40 / 72 / 80 (attn_kernel<..., PV16>): same-box A/B of round 5: D = 40 564 -> 604 (B = 32: 589 -> 640), D = 72 674 -> 715, D = 80 659 -> 703 TFLOP/s (profiles/r05_ab_attn_pv16.txt); GDF_ATTN_PV16=0 restores the 32x32x16 form
#endif

// ---- element type of q / k / v / o: fp16, or bf16 for a bf16 MMDiT model (BF; the 16-byte fragments are only containers) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ f32x16 mfma32(const f16x8 a, const f16x8 b, const f32x16 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ f16x2_t cvt_pair(float e0, float e1) {       // v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (BF) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(f16x2_t, __builtin_convertvector(f32x2{e0, e1}, bf16x2));
  } else {
    return __builtin_convertvector(f32x2{e0, e1}, f16x2_t);
  }
}
template <bool BF>
__device__ __forceinline__ _Float16 out16(float v) {
  if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)v);
  else return (_Float16)v;
}
// split output pair (AttnParams::o_lo): hi / lo halves of v as fp16, or as bf16 when `pbf` (o_pair_bf16: an fp16-internal attention
// feeding a bf16 split-operand GEMM, the MMDiT 'bfloat16x2' plans)
__device__ __forceinline__ _Float16 pair_hi(float v, bool pbf) {
  return pbf ? __builtin_bit_cast(_Float16, (__bf16)v) : (_Float16)v;
}
__device__ __forceinline__ _Float16 pair_lo(float v, bool pbf) {
  if (pbf) { const __bf16 h = (__bf16)v; return __builtin_bit_cast(_Float16, (__bf16)(v - (float)h)); }
  const _Float16 h = (_Float16)v;
  return (_Float16)(v - (float)h);
}

// combine a value with the one held by lane ^ 32 (the other half-wave owns the other keys of the same query):
// v_permlane32_swap instead of an LDS round trip (ds_bpermute)
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// token j of sample b -> row of the activation matrix.  seg_T == 0: sample-major rows (b * per_b + j).
// seg_T > 0 (MMDiT joint attention, Flux): the joint sequence [seg_T text tokens | S_tot - seg_T image tokens] lives
// region-major in HBM, [all samples' text rows][all samples' image rows], so that the text / image halves of every
// linear are contiguous row ranges (the reference concatenates per sample: attention_processor.py:2327-2329).
__device__ __forceinline__ size_t seg_row(int b, int j, int per_b, int seg_T, int B, int S_tot) {
  if (seg_T == 0) return (size_t)b * per_b + j;
  return j < seg_T ? (size_t)b * seg_T + j : (size_t)B * seg_T + (size_t)b * (S_tot - seg_T) + (j - seg_T);
}

// diagnostics build (tools/ablate_attn.sh, -DGDF_ATTN_TRACE): shader-clock time per loop phase, summed over the tiles of wave 0 of
// every workgroup: [QK^T, softmax, PV, stage + barrier + next loads, whole kernel]
#if defined(GDF_ATTN_TRACE)
__device__ unsigned long long gdf_attn_trace[8192 * 8];
#define GDF_AT_ENTRY const unsigned long long at_entry = __builtin_readcyclecounter();
#define GDF_AT_DECL unsigned long long at_acc[4] = {0, 0, 0, 0}; unsigned long long at_t = __builtin_readcyclecounter(); const unsigned long long at_t0 = at_t; \
    if (threadIdx.x == 0 && blockIdx.x < 8192) gdf_attn_trace[blockIdx.x * 8 + 5] = at_t0 - at_entry;
#define GDF_AT_EXIT do { if (threadIdx.x == 0 && blockIdx.x < 8192) gdf_attn_trace[blockIdx.x * 8 + 6] = __builtin_readcyclecounter() - at_entry; } while (0)
#define GDF_AT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); at_acc[i] += n_ - at_t; at_t = n_; } while (0)
#define GDF_AT_END do { if (threadIdx.x == 0 && blockIdx.x < 8192) { for (int i_ = 0; i_ < 4; ++i_) gdf_attn_trace[blockIdx.x * 8 + i_] = at_acc[i_]; \
    gdf_attn_trace[blockIdx.x * 8 + 4] = __builtin_readcyclecounter() - at_t0; } } while (0)
#else
#define GDF_AT_ENTRY
#define GDF_AT_DECL
#define GDF_AT_EXIT
#define GDF_AT(i)
#define GDF_AT_END
#endif
#if !defined(GDF_ATTN_PRIO)
#define GDF_ATTN_PRIO 1
#endif
#if GDF_ATTN_PRIO == 1          // MFMA phases at priority 1 (shipped)
#define GDF_ATTN_PRIO_MFMA(x) __builtin_amdgcn_s_setprio(x)
#elif GDF_ATTN_PRIO == 2        // experiment: the softmax (VALU) phase at priority 1 instead
#define GDF_ATTN_PRIO_MFMA(x) __builtin_amdgcn_s_setprio(1 - (x))
#else                           // experiment: no priorities
#define GDF_ATTN_PRIO_MFMA(x)
#endif
// (Round 4, measured and rejected: v_pk_fma_f32 / v_pk_add_f32 (inline asm: the compiler scalarises a <2 x float> fma whose lanes are extracted) for the
// softmax's scale-and-shift and row sum, halving those 96 VALU instructions per tile: 731-740 vs 755-760 TFLOP/s at D = 64, 910 vs 957-969 on the Flux joint
// shape — the opaque asm blocks cost the scheduler more than the issue slots save.)
// NW = waves per workgroup (4, or 8: twice the query rows share every staged K / V tile)
// (Measured and rejected: an explicit ping-pong — 8 waves, the two waves of a SIMD one workgroup barrier apart, iteration =
// softmax phase | barrier | PV(t) + QK^T(t+1) phase | barrier — which is what lifts the GEMM main loops.  Here the two phases
// are data dependent and unequal, and the lock step costs more than the free-running overlap of two independent workgroups
// per CU gives: D = 128 928 -> 821 TFLOP/s, D = 64 726 -> 573 (32 rows per wave) / 366 (64 rows per wave: 76 VGPRs spilled).)
// (Round 2, measured and rejected: a software-pipelined D = 64 kernel that issues the PV / QK^T MFMAs of one 32-query block between
// the softmax instructions of the wave's other block, K / V in rings of three: 770-795 vs 812-829 TFLOP/s.  tools/micro/overlap.hip
// shows why no such schedule can pay on gfx950: ordinary VALU instructions do not overlap with MFMAs on a SIMD at all — 16 MFMAs +
// 128 v_fma take 533 + 321 cycles whether they come from one wave, interleaved, or from two waves — only transcendentals do
// (16 MFMAs + 64 v_exp: 727 cycles against 533 and 644 alone).  The loop's bound is therefore MFMA + plain-VALU + LDS-read issue
// time, and the lever is the instruction count, not the placement.)
// OCC = workgroups per CU the register budget is sized for (2; 1 only in the GDF_ATTN_QW4 experiment below)
// PV16 (round 5, head dims whose 16-row padding is smaller than their 32-row padding: 40 -> 48 instead of 64, 72 / 80 -> 80 instead of 96): O^T += V^T P^T on
// mfma_f32_16x16x32_f16.  The S^T accumulator of the 32x32x16 score MFMAs keeps queries 0-15 in lane rows 0 / 2 and queries 16-31 in rows 1 / 3 (a row = 16
// lanes); one v_permlane16_swap per pair of packed P registers (X = keys {0-3, 8-11} + 4 lh, Y = keys {16-19, 24-27} + 4 lh of a 32-key block) turns them into
// X' = [X.r0, Y.r0, X.r2, Y.r2] = the four 8-key groups of queries 0-15 and Y' = the same for queries 16-31: exactly the B-operand layout of the 16x16x32
// instruction.  The key order inside the contraction is free, so the V^T fragment is simply read in the order the groups hold (row bases 0, 16, 4, 20 of the block).
// QKP (round 5, AttnParams::q_lo / kv_lo > 0, the full-split plans): q, k and v arrive as split fp16 pairs (hi, lo = fp16(x - hi), lo at +q_lo / +kv_lo elements
// in the same row) and the kernel contracts over them: S^T = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T, O^T += (V_hi^T + V_lo^T) P^T.  The fp16 STORAGE of q / k / v in
// front of the softmax is the one rounding no GEMM operand class removes; with peaked softmaxes (heavy-tailed weight statistics) it is the whole floor of the
// full split (DESIGN.md 3.9 h: 8.0e-4 on the worst hook by itself).  2.5 x the MFMAs (3 x Q K^T, 2 x P V); K_lo / V_lo tiles are staged like K / V.
template <int D, int QW, int NW = 4, bool BF = false, int OCC = 2, bool PV16 = false, bool QKP = false>
__global__ __launch_bounds__(NW * 64, OCC) void attn_kernel(const AttnParams p) {
  GDF_AT_ENTRY
  static_assert(!PV16 || !BF, "the 16-row P V form is fp16 only");
  static_assert(!QKP || (!PV16 && !BF), "split q / k / v pairs: fp16, 32x32x16 form");
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int ND16 = (D + 15) / 16;            // 16-row blocks of O^T (PV16)
  constexpr int NT = NW * 64;                    // threads per workgroup
  constexpr int DQK = (D + 15) / 16 * 16;        // contraction length of QK^T, padded
  constexpr int DV = (D + 31) / 32 * 32;         // output rows of O^T, padded
  constexpr int DP = DV;                         // data halves per LDS row (DV >= DQK)
  constexpr int LDR = DP + 8;                    // K row stride in halves (+16 B: conflict-free b128 fragment reads)
  // V row stride: the transposed fragment read (ds_read_b64_tr_b16) touches 4 rows x 64 B per half-wave, so the row
  // stride must be an odd multiple of 64 B (PMC: with the K stride half of all LDS cycles were bank conflicts)
  constexpr int LDV = (DP % 64 == 32) ? DP : DP + 32;
  constexpr int CPR = DP / 8;                    // 16-B chunks per row
  constexpr int NCH = (KT * CPR + NT - 1) / NT;  // chunks per thread per tile
  constexpr int NS = DQK / 16;                   // k-steps of QK^T
  constexpr int NDB = DV / 32;                   // 32-row blocks of O^T
  constexpr int PD = 4;                          // fragment reads in flight ahead of the MFMAs that consume them
  constexpr int QBW = 32 * QW;                   // query rows per wave
  constexpr int QBLK = NW * QBW;                 // query rows per workgroup
  constexpr bool PADDED = (DP != D);             // head dims 40 / 80: zero-filled pad chunks

  // double-buffered K / V tiles: one barrier per tile
  __shared__ __attribute__((aligned(16))) _Float16 sK[2][KT * LDR];
  __shared__ __attribute__((aligned(16))) _Float16 sV[2][KT * LDV];
  __shared__ __attribute__((aligned(16))) _Float16 sKl[2][QKP ? KT * LDR : 8];     // QKP: the lo halves of the K / V tiles
  __shared__ __attribute__((aligned(16))) _Float16 sVl[2][QKP ? KT * LDV : 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.Sq + QBLK - 1) / QBLK;
  int bid = blockIdx.x;
  const int qb = bid % nqb; bid /= nqb;
  const int head = bid % p.heads;
  const int b = bid / p.heads;

  const int lq = lane & 31, lh = lane >> 5;
  int q_row[QW];
  bool q_ok[QW];
  // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16 s + 8 lh .. +8] ----
  f16x8 qf[QW][NS];
  f16x8 qfl[QKP ? QW : 1][QKP ? NS : 1];                   // QKP: the lo halves of the Q fragments
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    q_row[w] = qb * QBLK + wave * QBW + w * 32 + lq;       // query index inside the sequence
    q_ok[w] = q_row[w] < p.Sq;
  }
  // (Measured and rejected: fetching Q with whole rows per group of lanes through the same staging slab that transposes O at
  // the end — the extra LDS round trip and workgroup barrier before the first K / V tile cost more than the scattered 16-byte
  // fragment loads: cross-attention 31 -> 36 us, self-attention at 1024 tokens 122 -> 129 us.)
  const bool pbf = p.o_lo > 0 && p.o_pair_bf16;                // output pair as bf16 hi + bf16 lo (fp16 internals): 'bfloat16x2' MMDiT plans
  constexpr int RSH = D + 8;                                    // O staging row stride in halves (16-byte aligned rows)
  constexpr bool STG = (NW / 2) * QBW * RSH <= 2 * KT * LDR && (NW / 2) * QBW * RSH <= 2 * KT * LDV;   // not at D = 32 (tiny rings)
  constexpr int LPRO = D / 8;                                   // lanes per staged row
  constexpr int RPIO = 64 / LPRO;                               // rows per store instruction
  _Float16* const stg = (wave < NW / 2) ? &sK[0][0] + wave * (QBW * RSH) : &sV[0][0] + (wave - NW / 2) * (QBW * RSH);
  const int orow = lane / LPRO, oc = (lane - orow * LPRO) * 8;
  {

#pragma unroll
    for (int w = 0; w < QW; ++w) {
      const _Float16* qp = p.q + seg_row(b, q_ok[w] ? q_row[w] : 0, p.Sq, p.seg_T, p.B, p.Sq) * p.ldq + head * D;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int d0 = 16 * s + 8 * lh;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok[w] && d0 < D) v = *(const f16x8*)(qp + d0);
        qf[w][s] = v;
        if constexpr (QKP) {
          f16x8 vl = {0, 0, 0, 0, 0, 0, 0, 0};
          if (q_ok[w] && d0 < D) vl = *(const f16x8*)(qp + p.q_lo + d0);
          qfl[w][s] = vl;
        }
      }
    }
  }

  f32x16 o[QW][PV16 ? 1 : NDB];
  f32x4_t o16[QW][PV16 ? ND16 : 1][2];            // PV16: [d block of 16][query half]: lane (q = 16 qh + lane % 16, d = 16 db + 4 (lane / 16) + j)
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    m_run[w] = -INFINITY; l_run[w] = 0.f;
    if constexpr (PV16) {
#pragma unroll
      for (int i = 0; i < ND16; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) o16[w][i][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[w][i][r] = 0.f;
    }
  }
  const float sl2 = p.scale * 1.44269504088896340736f;    // softmax(x*scale) via exp2

  // ---- K / V tile loads: global -> registers -> LDS.  Key kv of sample b lives in row kv + (kv < T ? c0 : c1)
  // (sample-major: c0 = c1 = b * kv_bstride; joint MMDiT layout: text rows first, see seg_row).  All address
  // arithmetic is 32-bit and scalar-based (uniform base pointer + per-lane unsigned element offset: the first
  // version spent ~150 of its 330 VALU instructions per tile on 64-bit address math and was VALU bound).
  const _Float16* kbase = p.k + head * D;
  const _Float16* vbase = p.v + head * D;
  // per-sample key count (PixArt cross attention: the text mask keeps a prefix of the Sk caption tokens; the reference adds
  // -10000 to the masked scores, transformer_2d.py:397-399, whose softmax weight is exactly 0 in fp32)
  const int Sk = p.kv_len ? max(1, min(p.kv_len[b], p.Sk)) : p.Sk;
  const int ntiles = (Sk + KT - 1) / KT;
  const int segT = p.seg_T > 0 ? p.seg_T : 0x7fffffff;
  const uint32_t c0 = p.seg_T > 0 ? (uint32_t)b * (uint32_t)p.seg_T : (uint32_t)b * (uint32_t)p.kv_bstride;
  const uint32_t c1 = p.seg_T > 0 ? (uint32_t)p.B * (uint32_t)p.seg_T + (uint32_t)b * (uint32_t)(p.Sk - p.seg_T) - (uint32_t)p.seg_T : c0;
  const uint32_t ldk = (uint32_t)p.ldk, ldv = (uint32_t)p.ldv;

  f16x8 kreg[NCH], vreg[NCH];
  f16x8 klreg[QKP ? NCH : 1], vlreg[QKP ? NCH : 1];
  auto gload = [&](int t) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * NT;
      const int row = idx / CPR, ch = idx - row * CPR;
      int kv = t * KT + row;
      kv = min(kv, Sk - 1);                    // tail rows re-read the last key: their scores are masked to -inf below
      const uint32_t r = (uint32_t)kv + (kv < segT ? c0 : c1);
      f16x8 kk = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kk;
      if ((KT * CPR) % NT == 0 || idx < KT * CPR) {
        if (!PADDED || ch * 8 < D) {
          kk = *(const f16x8*)(kbase + (r * ldk + (uint32_t)(ch * 8)));
          vv = *(const f16x8*)(vbase + (r * ldv + (uint32_t)(ch * 8)));
          if constexpr (QKP) {
            klreg[c] = *(const f16x8*)(kbase + p.kv_lo + (r * ldk + (uint32_t)(ch * 8)));
            vlreg[c] = *(const f16x8*)(vbase + p.kv_lo + (r * ldv + (uint32_t)(ch * 8)));
          }
        } else if constexpr (QKP) {
          klreg[c] = kk; vlreg[c] = kk;
        }
      } else if constexpr (QKP) {
        klreg[c] = kk; vlreg[c] = kk;
      }
      kreg[c] = kk; vreg[c] = vv;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * NT;
      const int row = idx / CPR, ch = idx - row * CPR;
      if ((KT * CPR) % NT == 0 || idx < KT * CPR) {
        *(f16x8*)(&sK[buf][row * LDR + ch * 8]) = kreg[c];
        *(f16x8*)(&sV[buf][row * LDV + ch * 8]) = vreg[c];
        if constexpr (QKP) {
          *(f16x8*)(&sKl[buf][row * LDR + ch * 8]) = klreg[c];
          *(f16x8*)(&sVl[buf][row * LDV + ch * 8]) = vlreg[c];
        }
      }
    }
  };

  // Fast tile load (tiles that lie inside one region and hold KT valid keys): buffer loads with a
  // per-lane byte offset computed ONCE and the tile's row offset in the scalar operand — no per-tile address arithmetic (the
  // generic form above spends ~25 VALU instructions per tile on clamping, region select and 64-bit address math, 10 % of the
  // loop's VALU work; tools/ablate_attn.py: the K / V global loads cost 12-15 % of the kernel, their LDS stores nothing).
  // Padded head dims (40 / 80) take the same path: a pad chunk's lane re-reads one of the row's own VALID chunks instead of being
  // zero-filled — K's pad columns meet the zero pad columns of the Q fragments, V's pad columns only reach O^T rows d >= D that
  // are never stored — so the loads stay unconditional, in bounds, and finite whenever the row itself is.
  // (same-box A/B, round 4: D = 72 635 -> 653 TFLOP/s, D = 80 598-612 -> 623, D = 40 unchanged; SD1.5 B = 32 856 -> 860 img/s)
  constexpr bool FASTLD = (KT * CPR) % NT == 0;
  constexpr int CPV = D / 8;                     // chunks per row that hold data
  static_assert(CPR - CPV <= CPV, "pad chunks alias valid ones");
  uint32_t kvo[NCH], vvo[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = tid + c * NT;
    const int row = idx / CPR, ch = idx - row * CPR;
    const int chv = (PADDED && ch >= CPV) ? ch - CPV : ch;
    kvo[c] = ((uint32_t)row * ldk + (uint32_t)(chv * 8)) * 2u;
    vvo[c] = ((uint32_t)row * ldv + (uint32_t)(chv * 8)) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0xffffffffu, 0x00020000);
  const bool seg_aligned = p.seg_T <= 0 || (p.seg_T % KT) == 0;
  const bool small_off = ((size_t)p.B * (size_t)(p.kv_bstride > p.Sk ? p.kv_bstride : p.Sk) + KT) * (size_t)(ldk > ldv ? ldk : ldv) * 2 < (1ull << 32);   // 32-bit byte offsets
  auto gload_fast = [&](int t) {
    const uint32_t r0 = (uint32_t)(t * KT) + ((t * KT) < segT ? c0 : c1);       // first row of the tile (uniform)
    const uint32_t sk = r0 * ldk * 2u, sv = r0 * ldv * 2u;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      kreg[c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[c], sk, 0));
      vreg[c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[c], sv, 0));
      if constexpr (QKP) {
        klreg[c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[c] + (uint32_t)p.kv_lo * 2u, sk, 0));
        vlreg[c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[c] + (uint32_t)p.kv_lo * 2u, sv, 0));
      }
    }
  };
  const bool fast_ok = FASTLD && seg_aligned && small_off;
  auto load_tile = [&](int t) {
    if (fast_ok && (t + 1) * KT <= Sk) gload_fast(t); else gload(t);
  };
  load_tile(0);
  lstore(0);
  __syncthreads();
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 1)
  lstore(1);                                   // diagnostics (tools/ablate_attn.sh): no K / V traffic inside the loop
  __syncthreads();
#else
  if (ntiles > 1) load_tile(1);
#endif
  GDF_AT_DECL
  // Fragment addresses = one lane-dependent LDS pointer per operand (opaque to the optimiser, which otherwise rebuilds every
  // fragment address with its own VALU adds: ~30 per tile; VALU issue time adds to MFMA time on this hardware, tools/micro/overlap.hip)
  // + the ring buffer's offset (one add per tile) + compile-time constants (key block, k-step) in the instruction's immediate.
  // (Unrolling the tile loop by two to make the buffer a constant as well spills 123 VGPRs at 64 rows per wave.)
  const int i16 = lane & 15;
  LDS_AS const char* kl = (LDS_AS const char*)&sK[0][0] + (lq * LDR + 8 * lh) * 2;
  LDS_AS const char* vl = (LDS_AS const char*)&sV[0][0] + ((4 * lh + (i16 >> 2)) * LDV + 16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
  if constexpr (PV16) {
    const int g4 = lane >> 4;                      // 8-key group of the 16x16x32 A operand: row base 16 (g & 1) + 4 (g >> 1) of the 32-key block
    vl = (LDS_AS const char*)&sV[0][0] + ((16 * (g4 & 1) + 4 * (g4 >> 1) + (i16 >> 2)) * LDV + (i16 & 3) * 4) * 2;
  }
  asm volatile("" : "+v"(kl), "+v"(vl));
  // QKP: the same lane-dependent pointers into the K_lo / V_lo tiles
  [[maybe_unused]] LDS_AS const char* kll = kl;
  [[maybe_unused]] LDS_AS const char* vll = vl;
  if constexpr (QKP) {
    kll = (LDS_AS const char*)&sKl[0][0] + (lq * LDR + 8 * lh) * 2;
    vll = (LDS_AS const char*)&sVl[0][0] + ((4 * lh + (i16 >> 2)) * LDV + 16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
    asm volatile("" : "+v"(kll), "+v"(vll));
  }
  for (int t = 0; t < ntiles; ++t) {
    const int BUF = t & 1;
    LDS_AS const char* const klt = kl + BUF * (KT * LDR * 2);
    LDS_AS const char* const vlt = vl + BUF * (KT * LDV * 2);
    [[maybe_unused]] LDS_AS const char* const kllt = kll + BUF * (KT * LDR * 2);
    [[maybe_unused]] LDS_AS const char* const vllt = vll + BUF * (KT * LDV * 2);
    GDF_AT(3);

    // ---- S^T = K Q^T : two 32-key blocks; every K fragment feeds QW query blocks ----
    // Fragment reads run PD steps ahead of the MFMAs that consume them (round 2: the compiler's order — read, wait, multiply —
    // exposed the LDS latency at every step; tools/trace_attn.py: QK^T 766 and PV 1622 cycles per tile for 512 cycles of MFMA each)
    f32x16 s[QW][2];
    constexpr int NQK = NS * 2 * (QKP ? 2 : 1);  // K fragments per tile, step i -> (st = i / 2, kb = i % 2): the key blocks alternate (QKP: then K_lo's)
    auto rdk = [&](int i) -> f16x8 {
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 64)
      return qf[0][i % NS];                              // diagnostics: no K fragment reads
#endif
      if constexpr (QKP) if (i >= NS * 2) {              // steps NQK0 .. 2 NQK0 - 1: the same fragments of the K_lo tile
        const int j = i - NS * 2;
        return *(LDS_AS const f16x8*)(kllt + ((j & 1) * 32 * LDR + 16 * (j >> 1)) * 2);
      }
      return *(LDS_AS const f16x8*)(klt + ((i & 1) * 32 * LDR + 16 * (i >> 1)) * 2);
    };
    f16x8 kq[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) kq[i] = rdk(i);
    // the first V^T fragments of this tile are fetched here as well: they land during the softmax
    auto rdv = [&](int i) -> f16x8 {             // step i -> (s4 = i / NDB, db = i % NDB)
      if constexpr (PV16) {                      // step i -> (kb = i / ND16, db = i % ND16): V[32 kb + base(g) + {0..3, 8..11}][16 db + lane % 16]
        const int kb = i / ND16, db = i - kb * ND16;
        LDS_AS const char* vp = vlt + (32 * kb * LDV + db * 16) * 2;
        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)vp);
        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(vp + 8 * LDV * 2));
        union { fp16x4_t q[2]; f16x8 h; } vf;
        vf.q[0] = lo; vf.q[1] = hi;
        return vf.h;
      }
      const bool vlo = QKP && i >= 4 * NDB;      // steps 4 NDB .. 8 NDB - 1: the same fragments of the V_lo tile
      const int i0 = vlo ? i - 4 * NDB : i;
      const int s4 = i0 / NDB, db = i0 - s4 * NDB;
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 32)
      return qf[0][(s4 + db) % NS];                      // diagnostics: no V^T fragment reads
#endif
      // V^T fragment: lane (d = db*32 + lq, lh) needs V[16 s4 + 4 lh + {0..3}][d] and V[16 s4 + 8 + 4 lh + {0..3}][d]
      // (row 4 lh + (i16 >> 2), column 16 ((lane >> 4) & 1) + 4 (i16 & 3) are in `vl`)
      LDS_AS const char* vp = ((QKP && vlo) ? vllt : vlt) + (16 * s4 * LDV + db * 32) * 2;
      const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)vp);
      const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(vp + 8 * LDV * 2));
      union { fp16x4_t q[2]; f16x8 h; } vf;              // pure register re-interpretation, no conversion
      vf.q[0] = lo; vf.q[1] = hi;
      return vf.h;
    };
    GDF_ATTN_PRIO_MFMA(1);
#pragma unroll
    for (int i = 0; i < NQK; ++i) {
      const bool klo = QKP && i >= NS * 2;       // a K_lo fragment: contracts with Q_hi only (the lo x lo term is 2^-22 relative)
      const int st = (klo ? i - NS * 2 : i) >> 1, kb = i & 1;
      const f16x8 kf = kq[i % PD];
#pragma unroll
      for (int w = 0; w < QW; ++w) {
        if constexpr (QKP) {
          if (klo) { s[w][kb] = mfma32<BF>(kf, qf[w][st], s[w][kb]); continue; }
          if (st == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            s[w][kb] = mfma32<BF>(kf, qfl[w][st], z);            // the small term first
          } else {
            s[w][kb] = mfma32<BF>(kf, qfl[w][st], s[w][kb]);
          }
          s[w][kb] = mfma32<BF>(kf, qf[w][st], s[w][kb]);
          continue;
        }
        if (st == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          s[w][kb] = mfma32<BF>(kf, qf[w][st], z);
        } else {
          s[w][kb] = mfma32<BF>(kf, qf[w][st], s[w][kb]);
        }
      }
      if (i + PD < NQK) kq[i % PD] = rdk(i + PD);
    }
    constexpr int NPV = PV16 ? 2 * ND16 : 4 * NDB * (QKP ? 2 : 1);   // V^T fragments per tile (QKP: then V_lo's)
    f16x8 vq[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vq[i] = rdv(i);
    GDF_ATTN_PRIO_MFMA(0);
    GDF_AT(0);
    // ---- mask the tail tile, online softmax (per-lane query column) ----
    if ((t + 1) * KT > Sk) {
#pragma unroll
      for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (kv >= Sk) s[w][kb][r] = -INFINITY;
          }
    }
    f16x8 pf[QW][4];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      float mx = s[w][0][0];
#if !(defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 2))
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[w][kb][r]);
      mx = half_max(mx) * sl2;
#endif
      // lazy rescale: the running max only moves when the new one exceeds it by more than 2^8 (probabilities then stay
      // <= 256 in fp16 and the fp32 accumulators never need the per-tile alpha multiply after the first tiles)
      const float m_new = (mx > m_run[w] + 8.0f) ? mx : m_run[w];
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 2)
          const float e0 = s[w][kb][r], e1 = s[w][kb][r + 1];      // diagnostics: no exp / fma / sum
#else
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 128)
          const float e0 = s[w][kb][r] * sl2 - m_new;                  // diagnostics: everything but the v_exp_f32
          const float e1 = s[w][kb][r + 1] * sl2 - m_new;
#else
          const float e0 = __builtin_amdgcn_exp2f(s[w][kb][r] * sl2 - m_new);
          const float e1 = __builtin_amdgcn_exp2f(s[w][kb][r + 1] * sl2 - m_new);
#endif
          psum += e0 + e1;
#endif
          const f16x2_t h2 = cvt_pair<BF>(e0, e1);
          pf[w][kb * 2 + (r >> 3)][r & 7] = h2[0];
          pf[w][kb * 2 + (r >> 3)][(r & 7) + 1] = h2[1];
        }
      if constexpr (PV16) {
        // the accumulators of a lane belong to the queries lane % 16 (+ 16): their factors live in the even / odd lane row of the same column
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[w]) != 0) {
          const float alpha = (m_new != m_run[w]) ? __builtin_amdgcn_exp2f(m_run[w] - m_new) : 1.0f;
          l_run[w] *= alpha;
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
          const float a0 = __uint_as_float(sw[0]), a1 = __uint_as_float(sw[1]);
#pragma unroll
          for (int i = 0; i < ND16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { o16[w][i][0][r] *= a0; o16[w][i][1][r] *= a1; }
        }
      } else if (m_new != m_run[w]) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[w] - m_new);   // raw v_exp_f32; first tile: exp2(-inf) = 0
        l_run[w] *= alpha;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[w][i][r] *= alpha;
      }
      l_run[w] += psum;
      m_run[w] = m_new;
    }

    // ---- O^T += V^T P^T : 4 steps of 16 (relabelled) keys; every V^T fragment feeds QW query blocks ----
    GDF_AT(1);
    GDF_ATTN_PRIO_MFMA(1);
    if constexpr (PV16) {
      // P: S^T layout -> 16x16x32 B operands (one permlane16_swap per packed register pair, see the kernel's header comment)
      f16x8 pa[QW][2], pb[QW][2];                // [key block of 32]: queries 0-15 / 16-31 of the wave's 32-query block
#pragma unroll
      for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 x = __builtin_bit_cast(u32x4, pf[w][2 * kb]), y = __builtin_bit_cast(u32x4, pf[w][2 * kb + 1]);
          u32x4 xa, ya;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(x[r], y[r], false, false);
            xa[r] = sw[0]; ya[r] = sw[1];
          }
          pa[w][kb] = __builtin_bit_cast(f16x8, xa); pb[w][kb] = __builtin_bit_cast(f16x8, ya);
        }
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        const int kb = i / ND16, db = i - kb * ND16;
        const f16x8 vf = vq[i % PD];
#pragma unroll
        for (int w = 0; w < QW; ++w) {
          o16[w][db][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pa[w][kb], o16[w][db][0], 0, 0, 0);
          o16[w][db][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[w][kb], o16[w][db][1], 0, 0, 0);
        }
        if (i + PD < NPV) vq[i % PD] = rdv(i + PD);
      }
    } else {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int i0 = (QKP && i >= 4 * NDB) ? i - 4 * NDB : i;
      const int s4 = i0 / NDB, db = i0 - s4 * NDB;
      const f16x8 vf = vq[i % PD];
#pragma unroll
      for (int w = 0; w < QW; ++w)
        o[w][db] = mfma32<BF>(vf, pf[w][s4], o[w][db]);
      if (i + PD < NPV) vq[i % PD] = rdv(i + PD);
    }
    }

    GDF_ATTN_PRIO_MFMA(0);
    GDF_AT(2);
    // ---- stage tile t+1 into the other buffer (last read during tile t-1, fenced by the previous barrier) ----
#if defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 1)
#if !(GDF_ATTN_ABLATE & 4)
    __syncthreads();
#endif
#elif defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 8)
    if (t + 1 < ntiles) lstore(BUF ^ 1);         // diagnostics: LDS stores of stale registers, no global loads
    __syncthreads();
#elif defined(GDF_ATTN_ABLATE) && (GDF_ATTN_ABLATE & 16)
    __syncthreads();                                 // diagnostics: global loads, no LDS stores
    if (t + 2 < ntiles) load_tile(t + 2);
    asm volatile("" :: "v"(kreg[0]), "v"(vreg[0]), "v"(kreg[NCH - 1]), "v"(vreg[NCH - 1]));
#else
    if (t + 1 < ntiles) lstore(BUF ^ 1);
    __syncthreads();
    if (t + 2 < ntiles) load_tile(t + 2);           // HBM latency hides under the next tile's MFMAs
#endif
  }

  GDF_AT(3);
  GDF_AT_END;
  // ---- finalize: O[q][d] = O^T[d][q] / l ----
  // The accumulators hold O^T (lane = query column): stored directly, every lane would write 8 bytes into a different row.
  // Each wave transposes its QBW x D block through the (now idle) K / V staging memory instead, so that the global stores
  // are 16 bytes per lane and whole D-wide rows per group of D/8 lanes.
  // PV16: a lane's accumulators belong to the queries lane % 16 (+ 16); their 1 / l lives in the even / odd lane row of the same column
  float inv16[QW][2];
  if constexpr (PV16) {
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      const float invq = (p.o_scale != 0.f ? p.o_scale : 1.0f) / half_sum(l_run[w]);
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(invq), __float_as_uint(invq), false, false);
      inv16[w][0] = __uint_as_float(sw[0]); inv16[w][1] = __uint_as_float(sw[1]);
    }
  }
  if constexpr (STG) if ((p.ldo & 7) == 0) {
    if constexpr (PV16) {
#pragma unroll
      for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int db = 0; db < ND16; ++db)
#pragma unroll
          for (int qh = 0; qh < 2; ++qh) {
            const int d0 = db * 16 + 4 * (lane >> 4);
            if (d0 < D) {
              f16x4 hv;
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = pbf ? pair_hi(o16[w][db][qh][e] * inv16[w][qh], true) : out16<BF>(o16[w][db][qh][e] * inv16[w][qh]);
              *(f16x4*)(stg + (w * 32 + 16 * qh + i16) * RSH + d0) = hv;
            }
          }
    } else {
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      const float inv = (p.o_scale != 0.f ? p.o_scale : 1.0f) / half_sum(l_run[w]);
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = db * 32 + 8 * rq + 4 * lh;
          if (d0 < D) {
            f16x4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (!BF && pbf) ? pair_hi(o[w][db][rq * 4 + e] * inv, true) : out16<BF>(o[w][db][rq * 4 + e] * inv);
            *(f16x4*)(stg + (w * 32 + lq) * RSH + d0) = hv;
          }
        }
    }
    }
    __builtin_amdgcn_wave_barrier();                            // same-wave LDS RAW across lanes: DS ops of one wave execute in order
#pragma unroll
    for (int it = 0; it < (QBW + RPIO - 1) / RPIO; ++it) {
      const int r = it * RPIO + orow;
      const int q = qb * QBLK + wave * QBW + r;
      if (lane < RPIO * LPRO && r < QBW && q < p.Sq)
        *(f16x8*)(p.o + seg_row(b, q, p.Sq, p.seg_T, p.B, p.Sq) * p.ldo + head * D + oc) = *(const f16x8*)(stg + r * RSH + oc);
    }
    if (!BF && p.o_lo > 0) {
      // split operand for the out-projection of a "precise" plan: lo = fp16(O - fp16(O)), staged and stored like the hi half
      __builtin_amdgcn_wave_barrier();
      if constexpr (PV16) {
#pragma unroll
        for (int w = 0; w < QW; ++w)
#pragma unroll
          for (int db = 0; db < ND16; ++db)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
              const int d0 = db * 16 + 4 * (lane >> 4);
              if (d0 < D) {
                f16x4 lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) lv[e] = pair_lo(o16[w][db][qh][e] * inv16[w][qh], pbf);
                *(f16x4*)(stg + (w * 32 + 16 * qh + i16) * RSH + d0) = lv;
              }
            }
      } else {
#pragma unroll
      for (int w = 0; w < QW; ++w) {
        const float inv = (p.o_scale != 0.f ? p.o_scale : 1.0f) / half_sum(l_run[w]);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int d0 = db * 32 + 8 * rq + 4 * lh;
            if (d0 < D) {
              f16x4 lv;
#pragma unroll
              for (int e = 0; e < 4; ++e) lv[e] = pair_lo(o[w][db][rq * 4 + e] * inv, pbf);
              *(f16x4*)(stg + (w * 32 + lq) * RSH + d0) = lv;
            }
          }
      }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < (QBW + RPIO - 1) / RPIO; ++it) {
        const int r = it * RPIO + orow;
        const int q = qb * QBLK + wave * QBW + r;
        if (lane < RPIO * LPRO && r < QBW && q < p.Sq)
          *(f16x8*)(p.o + seg_row(b, q, p.Sq, p.seg_T, p.B, p.Sq) * p.ldo + head * D + oc + p.o_lo) = *(const f16x8*)(stg + r * RSH + oc);
      }
    }
    GDF_AT_EXIT;
    return;
  }
  if constexpr (PV16) {
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const int qr = qb * QBLK + wave * QBW + w * 32 + 16 * qh + i16;       // query row of this lane's column
        if (qr < p.Sq) {
          _Float16* op = p.o + seg_row(b, qr, p.Sq, p.seg_T, p.B, p.Sq) * p.ldo + head * D;
#pragma unroll
          for (int db = 0; db < ND16; ++db) {
            const int d0 = db * 16 + 4 * (lane >> 4);
            if (d0 < D) {
              f16x4 hv;
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = pbf ? pair_hi(o16[w][db][qh][e] * inv16[w][qh], true) : out16<BF>(o16[w][db][qh][e] * inv16[w][qh]);
              *(f16x4*)(op + d0) = hv;
              if (p.o_lo > 0) {
                f16x4 lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) lv[e] = pair_lo(o16[w][db][qh][e] * inv16[w][qh], pbf);
                *(f16x4*)(op + d0 + p.o_lo) = lv;
              }
            }
          }
        }
      }
    return;
  }
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    const float l_tot = half_sum(l_run[w]);
    const float inv = (p.o_scale != 0.f ? p.o_scale : 1.0f) / l_tot;
    if (q_ok[w]) {
      _Float16* op = p.o + seg_row(b, q_row[w], p.Sq, p.seg_T, p.B, p.Sq) * p.ldo + head * D;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = db * 32 + 8 * rq + 4 * lh;
          if (d0 < D) {
            f16x4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (!BF && pbf) ? pair_hi(o[w][db][rq * 4 + e] * inv, true) : out16<BF>(o[w][db][rq * 4 + e] * inv);
            *(f16x4*)(op + d0) = hv;
            if (!BF && p.o_lo > 0) {
              f16x4 lv;
#pragma unroll
              for (int e = 0; e < 4; ++e) lv[e] = pair_lo(o[w][db][rq * 4 + e] * inv, pbf);
              *(f16x4*)(op + d0 + p.o_lo) = lv;
            }
          }
        }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// attention with materialised probabilities ('-map' hooks):
// AttnStoreProcessor.__call__ (/root/reference/feature/components/attention.py:176-263) +
// Attention.get_attention_scores (attention_processor.py:640-685): probs = softmax(scale * q k^T),
// stored as (B, heads, Sq, Sk) fp16, then out = probs v.
//
// HBM-write bound (SD1.5 level 0: 268 MB per image and layer).  Same MFMA skeleton as attn_kernel (4 waves x 32
// queries, S^T = K Q^T), two passes over the keys: pass A only tracks the row max / row sum, pass B recomputes
// the scores, normalises, feeds O^T += V^T P^T and writes the probability tile: each wave transposes its
// 32 x 64 tile through LDS so that the global stores are 16 B per lane, 128 contiguous bytes per query row.
// -------------------------------------------------------------------------------------------------
// a wave-uniform pointer the compiler could not prove uniform, moved into an SGPR pair
__device__ __forceinline__ const _Float16* uniform_ptr(const _Float16* ptr) {
  const uint64_t a = (uint64_t)ptr;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (const _Float16*)(((uint64_t)hi << 32) | lo);
}
// Workgroup barrier that only orders LDS traffic.  __syncthreads() carries a release fence, i.e. `s_waitcnt vmcnt(0)`:
// inside the map kernel that made every key tile wait for the round trip of the probability stores just issued
// (PMC: 54 % of wave cycles parked in s_waitcnt).  The K/V staging only needs the ds_writes to have landed.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// FULL = Sq % 128 == 0 and Sk % 64 == 0, sample-major, no key mask (every SD / SDXL level): no bounds predicates in the
// loops and unconditional tile loads (with the predicated form every loop iteration started with `s_waitcnt vmcnt(0)`,
// i.e. waited for the store round trip of the previous tile).
// OCC  = workgroups per CU the LDS / VGPR budget is cut for (3 for D <= 48: K rows trimmed to DQK columns).
// LW   = a fifth wave does nothing but move K/V tiles HBM -> LDS (see the loader block below); FULL, D <= 48, Sk % 128 == 0.
// SD1.5 level 0 (8 heads x 40, 4096^2 map, B = 8) on MI355X: 1.225 ms -> 0.92..0.97 ms (unconditional loads + K/V staged
// before the stores 1.10, 3 workgroups/CU 1.03, staggered key-tile order 0.99, loader wave 0.95); with the stores
// compiled out the kernel takes 0.63 ms, a pure 128-byte-strip store of the same tensor 0.39 ms
// (tools/micro/strip_store.hip, 5.1-5.5 TB/s): the remaining gap is issue latency of the two-pass softmax (PMC: VALU busy
// 33 %, MFMA 16 % of SIMD time), not HBM.
template <int D, bool FULL, int OCC, bool LW, bool BF = false>
__global__ __launch_bounds__(LW ? 320 : 256, OCC) void attn_map_kernel(const AttnParams p) {
  static_assert(!LW || FULL, "the loader-wave variant has no bounds predicates");
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int DV = (D + 31) / 32 * 32;
  constexpr int DP = DV;
  constexpr int LDR = DP + 8;
  constexpr int LDK = OCC > 2 ? DQK + 8 : LDR;    // K rows only hold the DQK columns the score MFMAs read
  constexpr int CPR = DP / 8;
  constexpr int NCH = (KT * CPR + 255) / 256;
  constexpr int NS = DQK / 16;
  constexpr int NDB = DV / 32;
  constexpr int PLD = KT + 8;                     // staging row stride (halves) of the probability tile

  __shared__ __attribute__((aligned(16))) _Float16 sK[2][KT * LDK];
  __shared__ __attribute__((aligned(16))) _Float16 sV[2][KT * LDR];
  __shared__ __attribute__((aligned(16))) _Float16 sP[4][32 * PLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.Sq + 127) / 128;
  int bid = blockIdx.x;
  const int qb = bid % nqb; bid /= nqb;
  const int head = bid % p.heads;
  const int b = bid / p.heads;
  const int lq = lane & 31, lh = lane >> 5;
  const int q0 = qb * 128 + wave * 32;             // first query row of this wave
  const int q_row = q0 + lq;
  const bool q_ok = q_row < p.Sq;

  f16x8 qf[NS];
  if (!(LW && wave == 4)) {        // (the loader wave must not have a compiler-tracked load in flight next to its asm loads)
    const _Float16* qp = p.q + seg_row(b, q_ok ? q_row : 0, p.Sq, p.seg_T, p.B, p.Sq) * p.ldq + head * D;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int d0 = 16 * s + 8 * lh;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q_ok && d0 < D) v = *(const f16x8*)(qp + d0);
      qf[s] = v;
    }
  }
  const float sl2 = p.scale * 1.44269504088896340736f;
  const _Float16* kbase = p.k + head * D;
  const _Float16* vbase = p.v + head * D;
  const int ntiles = (p.Sk + KT - 1) / KT;
  // keys [Skv, Sk) are masked (PixArt caption mask, AttnParams.kv_len): probability 0, still written to the map
  const int Skv = p.kv_len ? max(1, min(p.kv_len[b], p.Sk)) : p.Sk;

  f16x8 kreg[NCH], vreg[NCH];
  constexpr int CHV = (D + 7) / 8;                 // 16-byte chunks of a K / V row that hold data
  if (FULL) {
    // FULL: the zero padding of the staged rows (columns [D, DP)) is written once; the tile loads are unconditional
    // (pad lanes re-read the last data chunk and drop it), so no load sits under a lane predicate or behind a zero-fill
    // of its destination registers — both made the compiler start every key tile with `s_waitcnt vmcnt(0)`
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 2 * KT * LDK / 8; i += 256) *(f16x8*)(&sK[0][i * 8]) = z8;
    for (int i = tid; i < 2 * KT * LDR / 8; i += 256) *(f16x8*)(&sV[0][i * 8]) = z8;
    __syncthreads();
  }
  const _Float16* kfull = kbase + (size_t)b * p.kv_bstride * p.ldk;
  const _Float16* vfull = vbase + (size_t)b * p.kv_bstride * p.ldv;
  auto gload = [&](int t, bool with_v) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * 256;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int kv = t * KT + row;
      if (FULL) {
        const int che = ch < CHV ? ch : CHV - 1;
        kreg[c] = *(const f16x8*)(kfull + (size_t)kv * p.ldk + che * 8);
        if (with_v) vreg[c] = *(const f16x8*)(vfull + (size_t)kv * p.ldv + che * 8);
      } else {
        f16x8 kk = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kk;
        if (((KT * CPR) % 256 == 0 || idx < KT * CPR) && kv < p.Sk && ch * 8 < D) {
          const size_t r = seg_row(b, kv, p.kv_bstride, p.seg_T, p.B, p.Sk);
          kk = *(const f16x8*)(kbase + r * p.ldk + ch * 8);
          if (with_v) vv = *(const f16x8*)(vbase + r * p.ldv + ch * 8);
        }
        kreg[c] = kk; vreg[c] = vv;
      }
    }
  };
  auto lstore = [&](int buf, bool with_v) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * 256;
      const int row = idx / CPR, ch = idx - row * CPR;
      if (FULL) {
        if (ch < CHV) {
          *(f16x8*)(&sK[buf][row * LDK + ch * 8]) = kreg[c];
          if (with_v) *(f16x8*)(&sV[buf][row * LDR + ch * 8]) = vreg[c];
        }
      } else if (idx < KT * CPR) {
        if (ch * 8 < DQK) *(f16x8*)(&sK[buf][row * LDK + ch * 8]) = kreg[c];
        if (with_v) *(f16x8*)(&sV[buf][row * LDR + ch * 8]) = vreg[c];
      }
    }
  };
  auto scores = [&](const _Float16* cK, int t, f32x16 (&s)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const f16x8 kf = *(const f16x8*)(cK + (kb * 32 + lq) * LDK + 16 * st + 8 * lh);
        if (st == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          s[kb] = mfma32<BF>(kf, qf[st], z);
        } else {
          s[kb] = mfma32<BF>(kf, qf[st], s[kb]);
        }
      }
    if (!FULL && (t + 1) * KT > Skv) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (kv >= Skv) s[kb][r] = -INFINITY;
        }
    }
  };

  // every workgroup walks the key tiles of pass B from its own starting tile (wrapping): at any instant the resident
  // workgroups then write different 128-byte columns of their rows instead of the same one (HBM channel spread)
  const int t0 = (int)((blockIdx.x * 37u) % (unsigned)ntiles);
  auto tile_of = [&](int t) { int x = t + t0; return x >= ntiles ? x - ntiles : x; };

  // ---------------- LW: wave 4 only moves K / V tiles HBM -> LDS ----------------
  // `vmcnt` is one counter for loads and stores, and the compiler has to assume stores retire out of order with loads: in a
  // wave that does both, every wait for a K/V tile is also a wait for the probability stores issued before it (store round
  // trip per key tile).  With the loads in a wave of their own the four compute waves only ever have stores in flight and
  // never wait on the counter inside the loops.  Step i of 2 * ntiles (pass A then pass B) lives in buffer i & 1; the
  // loader runs two steps ahead in registers and one step ahead in LDS; every wave meets at one barrier per step.
  if constexpr (LW) if (wave == 4) {
    constexpr int NLC = KT * CHV / 64;
    static_assert((KT * CHV) % 64 == 0, "whole wave loads");
    const int nsteps = 2 * ntiles;
    f16x8 ak[NLC], av[NLC], bk[NLC], bv[NLC];
    // The loads are inline asm with hand-counted waits: for loads carried over a loop back edge the compiler's own
    // waitcnt insertion falls back to `vmcnt(0)`, which would also wait for the tile just requested.  Loads retire in order,
    // so "all but the N youngest have landed" is exact; the "+v" ties keep every use of a register after its wait.
    auto gld = [&](const _Float16* base, uint32_t off) {      // uniform 64-bit tile base (SGPR pair) + per-lane byte offset
      f16x8 r;
      // s_nop: the base may have just been written by v_readfirstlane (VALU-writes-SGPR -> VMEM-reads-it needs 5 wait states,
      // and the hazard recogniser does not look inside inline asm)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
      return r;
    };
    uint32_t offk[NLC], offv[NLC];                            // byte offsets inside a tile: the same for every step
#pragma unroll
    for (int c = 0; c < NLC; ++c) {
      const int idx = lane + c * 64;
      const int row = idx / CHV, ch = idx - row * CHV;
      offk[c] = (uint32_t)(row * p.ldk + ch * 8) * 2u;
      offv[c] = (uint32_t)(row * p.ldv + ch * 8) * 2u;
    }
    auto ld = [&](int step, auto with_v, f16x8 (&rk)[NLC], f16x8 (&rv)[NLC]) {
      const int tile = decltype(with_v)::value ? tile_of(step - ntiles) : step;    // V travels in pass B only
      const _Float16* kt = uniform_ptr(kfull + (size_t)tile * KT * p.ldk);
      const _Float16* vt = uniform_ptr(vfull + (size_t)tile * KT * p.ldv);
#pragma unroll
      for (int c = 0; c < NLC; ++c) {
        rk[c] = gld(kt, offk[c]);
        if constexpr (decltype(with_v)::value) rv[c] = gld(vt, offv[c]);
      }
    };
    // wait until at most `younger` loads are in flight, then stage the tile
    auto st = [&](int step, auto with_v, auto younger, f16x8 (&rk)[NLC], f16x8 (&rv)[NLC]) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(younger)::value) : "memory");
#pragma unroll
      for (int c = 0; c < NLC; ++c) {
        asm volatile("" : "+v"(rk[c]));
        if constexpr (decltype(with_v)::value) asm volatile("" : "+v"(rv[c]));
      }
#pragma unroll
      for (int c = 0; c < NLC; ++c) {
        const int idx = lane + c * 64;
        const int row = idx / CHV, ch = idx - row * CHV;
        *(f16x8*)(&sK[step & 1][row * LDK + ch * 8]) = rk[c];
        if constexpr (decltype(with_v)::value) *(f16x8*)(&sV[step & 1][row * LDR + ch * 8]) = rv[c];
      }
    };
    constexpr std::false_type K_{};
    constexpr std::true_type KV_{};
    constexpr std::integral_constant<int, 0> N0{};
    constexpr std::integral_constant<int, NLC> NK{};          // loads of one K tile
    constexpr std::integral_constant<int, 2 * NLC> NKV{};     // loads of one K + V tile
    // straight-line pairs of steps; ntiles is even (host-checked)
    ld(0, K_, ak, av); st(0, K_, N0, ak, av);
    lds_barrier();
    ld(1, K_, bk, bv);
    int i = 0;
    for (; i + 2 < ntiles; i += 2) {                          // pass A
      ld(i + 2, K_, ak, av); st(i + 1, K_, NK, bk, bv);
      lds_barrier();                                          // end of step i
      ld(i + 3, K_, bk, bv); st(i + 2, K_, NK, ak, av);
      lds_barrier();                                          // end of step i + 1
    }
    ld(i + 2, KV_, ak, av); st(i + 1, K_, NKV, bk, bv);       // pass A -> pass B
    lds_barrier();
    ld(i + 3, KV_, bk, bv); st(i + 2, KV_, NKV, ak, av);
    lds_barrier();
    for (i += 2; i + 2 < nsteps; i += 2) {                    // pass B
      ld(i + 2, KV_, ak, av); st(i + 1, KV_, NKV, bk, bv);
      lds_barrier();
      ld(i + 3, KV_, bk, bv); st(i + 2, KV_, NKV, ak, av);
      lds_barrier();
    }
    st(i + 1, KV_, N0, bk, bv);
    lds_barrier();
    lds_barrier();
    return;
  }

  // ---------------- pass A: row max and row sum ----------------
  float m_run = -INFINITY, l_run = 0.f;
  if (!LW) { gload(0, false); lstore(0, false); }
  lds_barrier();
  if (!LW && ntiles > 1) gload(1, false);
  for (int t = 0; t < ntiles; ++t) {
    f32x16 s[2];
    scores(sK[t & 1], t, s);
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = half_max(mx);
    const float m_new = fmaxf(m_run, mx * sl2);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(s[kb][r] * sl2 - m_new);
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + psum;
    m_run = m_new;
    if (!LW && t + 1 < ntiles) lstore((t + 1) & 1, false);
    lds_barrier();
    if (!LW && t + 2 < ntiles) gload(t + 2, false);
  }
  const float m_fin = m_run + __builtin_amdgcn_logf(half_sum(l_run));   // p = 2^(s - m - log2 l)   (v_log_f32 is log2)

  // ---------------- pass B: probabilities -> HBM, O^T += V^T P^T ----------------
  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  _Float16* sPw = sP[wave];
  // sample-major: map = (B, heads, Sq, Sk).  MMDiT joint layout (seg_T = T > 0, FluxAttnStoreProcessor,
  // components/attention.py:493-502): only the image queries are kept, split by key into
  //   map  = `self-map`  (B, heads, Sq - T, Sk - T)      map2 = `cross-map` (B, heads, Sq - T, T)      (either may be NULL)
  const int T = FULL ? 0 : p.seg_T, Si = p.Sq - T;
  _Float16* mbase = p.map ? p.map + (((size_t)b * p.heads + head) * (T ? Si : p.Sq)) * (T ? Si : p.Sk) : nullptr;
  _Float16* m2base = (T && p.map2) ? p.map2 + (((size_t)b * p.heads + head) * Si) * T : nullptr;
  const bool vec_ok = T ? (((T | Si) & 7) == 0) : ((p.Sk & 7) == 0);            // 16-byte aligned probability rows
  const int bofs = LW ? (ntiles & 1) : 0;          // LW: pass B continues the step numbering of pass A
  if (!LW) {
    gload(tile_of(0), true); lstore(0, true);
    lds_barrier();
  }
  for (int t = 0; t < ntiles; ++t) {
    const int tt = tile_of(t);
    const int cb = (t + bofs) & 1;
    // K/V of tile t+1 are requested at the top and staged at the bottom of the SAME iteration: no load is pending across
    // the loop edge, so the only wait on the probability stores is the one the hardware needs (with loads carried over the
    // back edge the compiler started every iteration with `s_waitcnt vmcnt(0)`, i.e. after the previous tile's stores)
    if (!LW && t + 1 < ntiles) gload(tile_of(t + 1), true);
    const _Float16* cV = sV[cb];
    f32x16 s[2];
    scores(sK[cb], tt, s);
    f16x8 pf[4];                                     // fp16 probabilities: what the map stores (and the P V operand unless BF)
    f16x8 pb[BF ? 4 : 1];                            // BF: the same probabilities rounded to bf16 for the P V MFMA
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(s[kb][r] * sl2 - m_fin);      // 1 / l folded into the exponent
        const float e1 = __builtin_amdgcn_exp2f(s[kb][r + 1] * sl2 - m_fin);
        const f16x2_t h2 = cvt_pair<false>(e0, e1);
        pf[kb * 2 + (r >> 3)][r & 7] = h2[0];
        pf[kb * 2 + (r >> 3)][(r & 7) + 1] = h2[1];
        if constexpr (BF) {
          const f16x2_t b2 = cvt_pair<true>(e0, e1);
          pb[kb * 2 + (r >> 3)][r & 7] = b2[0];
          pb[kb * 2 + (r >> 3)][(r & 7) + 1] = b2[1];
        }
      }
    // stage K/V of tile t+1 BEFORE this tile's probability stores are issued: the wait for the loads then sits behind the
    // stores of tile t-1 only (a whole iteration old), not behind the ones about to be issued
    if (!LW && t + 1 < ntiles) lstore((t + 1) & 1, true);
    // transpose the 32 x 64 probability tile through this wave's LDS slab: lane (q = lq, lh) owns keys
    // kb*32 + 8*g + 4*lh + {0..3}  (g = r >> 2)  ->  row q, 4 consecutive halves
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = pf[kb * 2 + (g >> 1)][(g & 1) * 4 + e];
        *(f16x4*)(sPw + lq * PLD + kb * 32 + 8 * g + 4 * lh) = w4;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const int q = q0 + row, kv = tt * KT + ch * 8;
      if (T == 0) {
        if (FULL) {
          *(f16x8*)(mbase + (size_t)q * p.Sk + kv) = *(const f16x8*)(sPw + row * PLD + ch * 8);
        } else if (q < p.Sq && kv < p.Sk) {
          const f16x8 v8 = *(const f16x8*)(sPw + row * PLD + ch * 8);
          _Float16* dst = mbase + (size_t)q * p.Sk + kv;
          if (vec_ok) {
            *(f16x8*)dst = v8;
          } else {
            for (int e = 0; e < 8; ++e) if (kv + e < p.Sk) dst[e] = v8[e];
          }
        }
      } else if (q >= T && q < p.Sq && kv < p.Sk) {          // T % 8 == 0 (host-checked): an 8-key chunk never straddles T
        const f16x8 v8 = *(const f16x8*)(sPw + row * PLD + ch * 8);
        _Float16* dst = (kv < T) ? (m2base ? m2base + (size_t)(q - T) * T + kv : nullptr)
                                 : (mbase ? mbase + (size_t)(q - T) * Si + (kv - T) : nullptr);
        if (dst) {
          const int lim = (kv < T) ? T : p.Sk;
          if (vec_ok && kv + 8 <= lim) *(f16x8*)dst = v8;
          else for (int e = 0; e < 8; ++e) if (kv + e < lim) dst[e] = v8[e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const int i16 = lane & 15;
        const int c0 = db * 32 + 16 * ((lane >> 4) & 1) + (i16 & 3) * 4;
        const int r0 = 16 * s4 + 4 * lh + (i16 >> 2);
        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(cV + r0 * LDR + c0));
        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(cV + (r0 + 8) * LDR + c0));
        union { fp16x4_t q[2]; f16x8 h; } vf;
        vf.q[0] = lo; vf.q[1] = hi;
        if constexpr (BF) o[db] = mfma32<true>(vf.h, pb[s4], o[db]);
        else o[db] = mfma32<false>(vf.h, pf[s4], o[db]);
      }
    lds_barrier();
  }
  if (q_ok) {
    _Float16* op = p.o + seg_row(b, q_row, p.Sq, p.seg_T, p.B, p.Sq) * p.ldo + head * D;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d0 = db * 32 + 8 * rq + 4 * lh;
        if (d0 < D) {
          f16x4 hv;
          const bool pbf = p.o_lo > 0 && p.o_pair_bf16;
          const float osc = p.o_scale != 0.f ? p.o_scale : 1.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (!BF && pbf) ? pair_hi(o[db][rq * 4 + e] * osc, true) : out16<BF>(o[db][rq * 4 + e] * osc);
          *(f16x4*)(op + d0) = hv;
          if (!BF && p.o_lo > 0) {
            f16x4 lv;
#pragma unroll
            for (int e = 0; e < 4; ++e) lv[e] = pair_lo(o[db][rq * 4 + e] * osc, pbf);
            *(f16x4*)(op + d0 + p.o_lo) = lv;
          }
        }
      }
  }
}

template <int D, bool BF = false>
static hipError_t launch_d(const AttnParams& p, hipStream_t s) {
  const bool maps = p.map || p.map2;
  if (maps && p.seg_T && (p.seg_T & 7)) return hipErrorInvalidValue;    // an 8-key chunk must not straddle the text / image boundary
  if (maps) {
    const int nqb = (p.Sq + 127) / 128;
    constexpr bool can3 = D <= 48;   // K trimmed to DQK columns: 3 workgroups per CU fit the 160 KiB of LDS
    constexpr bool canlw = D <= 48;  // loader-wave variant (two K+V tiles of staging registers must fit 128 VGPRs)
    const dim3 grid(p.B * p.heads * nqb);
    if (!BF && p.Sq % 128 == 0 && p.Sk % KT == 0 && !p.seg_T && !p.kv_len) {
      if (canlw && p.Sk % (2 * KT) == 0) hipLaunchKernelGGL((attn_map_kernel<D, true, can3 ? 3 : 2, canlw>), grid, dim3(320), 0, s, p);
      else hipLaunchKernelGGL((attn_map_kernel<D, true, can3 ? 3 : 2, false>), grid, dim3(256), 0, s, p);
    } else hipLaunchKernelGGL((attn_map_kernel<D, false, 2, false, BF>), grid, dim3(256), 0, s, p);
  } else {
    // 64 query rows per wave (every K / V fragment feeds two MFMAs) when the sequence is long and the
    // accumulators fit (D <= 64); 32 rows per wave otherwise
    // (D = 128 with 64 rows per wave needs all 512 registers, one wave per SIMD: measured 822 vs 886 TFLOP/s on the Flux joint shape)
    constexpr bool can2 = D <= 64;
    // 64 rows per wave wins even at 2.5 rounds of the slots (S = 1024, batch 16: 0.125 vs 0.130 ms) but halves the number of
    // workgroups: pick by rate x fill of the 512 workgroup slots
    const long nb2 = (long)p.B * p.heads * ((p.Sq + 255) / 256), nb1 = (long)p.B * p.heads * ((p.Sq + 127) / 128);
    auto fill = [](long n, long slots) { const long r = (n + slots - 1) / slots; return (double)n / (double)(r * slots); };
#if defined(GDF_ATTN_QW4)
    // experiment (tools/build_variant.sh qw4 -DGDF_ATTN_QW4=4 | =3): ONE wave per SIMD holding 4 (3) query blocks of 32 rows — every K / V fragment
    // read from LDS feeds 4 (3) MFMAs, all 512 registers to one wave (result in DESIGN.md §3.11)
#if GDF_ATTN_QW4 == 28       // 8 waves x 2 query blocks, one workgroup per CU: every staged K / V tile serves 512 query rows (half the L2 -> LDS traffic)
    if (!BF && D == 64 && p.Sq >= 1024) {
      const int nqb = (p.Sq + 511) / 512;
      hipLaunchKernelGGL((attn_kernel<D, (D == 64) ? 2 : 1, 8, false, 1>), dim3(p.B * p.heads * nqb), dim3(512), 0, s, p);
      return hipGetLastError();
    }
#endif
    if (!BF && D == 64 && p.Sq >= 1024) {
      constexpr int Q4 = (D == 64 && GDF_ATTN_QW4 <= 4) ? GDF_ATTN_QW4 : 1;
      const int nqb = (p.Sq + 128 * Q4 - 1) / (128 * Q4);
      hipLaunchKernelGGL((attn_kernel<D, Q4, 4, false, 1>), dim3(p.B * p.heads * nqb), dim3(256), 0, s, p);
      return hipGetLastError();
    }
#endif
    // split q / k / v pairs (AttnParams::q_lo / kv_lo, the full-split UNet plans): 8 waves x 32 query rows share the four staged tiles (K, K_lo, V, V_lo), one
    // workgroup per CU; head dims whose tiles do not fit (160) or do not split evenly over 512 threads use 4 waves / fall through to the hi halves
    if constexpr (!BF && D <= 80 && D >= 40) {
      if (p.q_lo > 0 && p.kv_lo > 0) {
        constexpr int W8 = (KT * ((D + 31) / 32 * 32 / 8)) % 512 == 0 ? 8 : 4;
        const int nqb = (p.Sq + 32 * W8 - 1) / (32 * W8);
        hipLaunchKernelGGL((attn_kernel<D, 1, W8, false, 1, false, true>), dim3(p.B * p.heads * nqb), dim3(64 * W8), 0, s, p);
        return hipGetLastError();
      }
    }
    // round 5: P V on mfma_f32_16x16x32_f16 where the 16-row padding of D is smaller than the 32-row one (40 / 72 / 80); GDF_ATTN_PV16=0 / 1: A/B switch
    constexpr bool pv16c = !BF && ((D + 15) / 16 * 16 < (D + 31) / 32 * 32);
    static const bool pv16 = [] { const char* e = getenv("GDF_ATTN_PV16"); return e ? atoi(e) != 0 : GDF_ATTN_PV16_DEFAULT; }();
    if (pv16c && pv16) {
      if (can2 && p.Sq >= 512 && 1.00 * fill(nb2, 512) >= 0.80 * fill(nb1, 512)) {
        const int nqb = (p.Sq + 255) / 256;
        hipLaunchKernelGGL((attn_kernel<D, can2 ? 2 : 1, 4, false, 2, pv16c>), dim3(p.B * p.heads * nqb), dim3(256), 0, s, p);
      } else {
        const int nqb = (p.Sq + 127) / 128;
        hipLaunchKernelGGL((attn_kernel<D, 1, 4, false, 2, pv16c>), dim3(p.B * p.heads * nqb), dim3(256), 0, s, p);
      }
      return hipGetLastError();
    }
    if (!BF && can2 && p.Sq >= 512 && 1.00 * fill(nb2, 512) >= 0.80 * fill(nb1, 512)) {
      const int nqb = (p.Sq + 255) / 256;
      hipLaunchKernelGGL((attn_kernel<D, can2 ? 2 : 1>), dim3(p.B * p.heads * nqb), dim3(256), 0, s, p);
    } else if (D == 128 && p.Sq >= 1024) {
      // 8 waves share every staged K / V tile (half the L2 -> LDS traffic per query row): 877 -> 917 TFLOP/s on the Flux joint
      // shape; at D = 72 the 768-chunk tile does not split evenly over 512 threads (690 -> 591), so only D = 128 takes it
      const int nqb = (p.Sq + 255) / 256;
      hipLaunchKernelGGL((attn_kernel<D, 1, D == 128 ? 8 : 4, BF>), dim3(p.B * p.heads * nqb), dim3(D == 128 ? 512 : 256), 0, s, p);
    } else {
      const int nqb = (p.Sq + 127) / 128;
      hipLaunchKernelGGL((attn_kernel<D, 1, 4, BF>), dim3(p.B * p.heads * nqb), dim3(256), 0, s, p);
    }
  }
  return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
  if ((p.ldq | p.ldk | p.ldv) & 7) return hipErrorInvalidValue;     // 16-byte aligned rows
  if (p.ldo & 3) return hipErrorInvalidValue;
  if (p.bf16) return p.D == 128 ? launch_d<128, true>(p, s) : hipErrorInvalidValue;      // bf16: the MMDiT head dim only
  switch (p.D) {
    case 32: return launch_d<32>(p, s);
    case 40: return launch_d<40>(p, s);
    case 64: return launch_d<64>(p, s);
    case 72: return launch_d<72>(p, s);
    case 80: return launch_d<80>(p, s);
    case 128: return launch_d<128>(p, s);
    case 160: return launch_d<160>(p, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gdf

#if defined(GDF_ATTN_TRACE)
extern "C" int gdf_debug_attn_trace(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gdf::gdf_attn_trace), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
