// Can VALU work hide under MFMA work on a gfx950 SIMD?  One workgroup per CU of 4 or 8 waves; every wave runs ITER rounds of
//   mode 0: 16 x v_mfma_f32_32x32x16_f16 (4 independent accumulators)          mode 1: NV x v_fma_f32 (8 independent chains)
//   mode 2: both, one MFMA followed by NV/16 fmas (same wave)                   mode 3: waves 0-3 MFMA only, waves 4-7 VALU only
//   mode 4: mode 1 with v_exp_f32 instead of v_fma_f32                          mode 5: mode 2 with v_exp_f32
//   mode 6 / 7: modes 1 / 2 with v_pk_fma_f32 (NV counts fp32 elements: NV/2 packed instructions)
// hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip && ./overlap
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int NV>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
  const bool do_m = MODE == 0 || MODE == 2 || MODE == 5 || MODE == 7 || (MODE == 3 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || (MODE == 3 && wave >= 4);
  constexpr bool PK = MODE == 6 || MODE == 7;
  f32x2 pv[8];
  for (int j = 0; j < 8; ++j) pv[j] = f32x2{threadIdx.x * 0.01f + j, j * 0.5f};
  constexpr bool EXP = MODE == 4 || MODE == 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (do_m) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
      if (do_v) {
        if (PK) {
#pragma unroll
          for (int q = 0; q < NV / 32; ++q) { f32x2& x = pv[(i * (NV / 32) + q) & 7]; x = x * 1.0001f + 0.5f; }
        } else {
#pragma unroll
          for (int q = 0; q < NV / 16; ++q) {
            float& x = v[(i * (NV / 16) + q) & 7];
            if (EXP) x = __builtin_amdgcn_exp2f(x); else x = x * 1.0001f + 0.5f;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 8; ++j) s += v[j] + pv[j][0] + pv[j][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV>
float run(int waves, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV><<<256, waves * 64>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NV><<<256, waves * 64>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int warm = 0; warm < 3; ++warm) run<0, 128>(8, out, iters);
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / iters; };   // cycles per round at 2.4 GHz nominal
  printf("per round of 16 MFMA (512 pipe cycles) / NV VALU; ms and nominal cycles per round\n");
  float t;
  t = run<0, 128>(4, out, iters); printf("4 waves (1/SIMD) MFMA only            %8.3f ms %7.0f\n", t, cyc(t));
  t = run<0, 128>(8, out, iters); printf("8 waves (2/SIMD) MFMA only            %8.3f ms %7.0f\n", t, cyc(t));
  t = run<1, 128>(4, out, iters); printf("4 waves 128 fma only                  %8.3f ms %7.0f\n", t, cyc(t));
  t = run<1, 128>(8, out, iters); printf("8 waves 128 fma only                  %8.3f ms %7.0f\n", t, cyc(t));
  t = run<2, 128>(4, out, iters); printf("4 waves MFMA + 128 fma interleaved    %8.3f ms %7.0f\n", t, cyc(t));
  t = run<2, 128>(8, out, iters); printf("8 waves MFMA + 128 fma interleaved    %8.3f ms %7.0f\n", t, cyc(t));
  t = run<2, 64>(4, out, iters);  printf("4 waves MFMA + 64 fma interleaved     %8.3f ms %7.0f\n", t, cyc(t));
  t = run<2, 64>(8, out, iters);  printf("8 waves MFMA + 64 fma interleaved     %8.3f ms %7.0f\n", t, cyc(t));
  t = run<3, 128>(8, out, iters); printf("8 waves: 4 MFMA-only + 4 fma-only(128)%8.3f ms %7.0f\n", t, cyc(t));
  t = run<4, 64>(4, out, iters);  printf("4 waves 64 exp only                   %8.3f ms %7.0f\n", t, cyc(t));
  t = run<4, 64>(8, out, iters);  printf("8 waves 64 exp only                   %8.3f ms %7.0f\n", t, cyc(t));
  t = run<5, 64>(4, out, iters);  printf("4 waves MFMA + 64 exp interleaved     %8.3f ms %7.0f\n", t, cyc(t));
  t = run<5, 64>(8, out, iters);  printf("8 waves MFMA + 64 exp interleaved     %8.3f ms %7.0f\n", t, cyc(t));
  t = run<6, 128>(4, out, iters); printf("4 waves 64 pk_fma (128 elements) only %8.3f ms %7.0f\n", t, cyc(t));
  t = run<6, 128>(8, out, iters); printf("8 waves 64 pk_fma only                %8.3f ms %7.0f\n", t, cyc(t));
  t = run<7, 128>(4, out, iters); printf("4 waves MFMA + 64 pk_fma interleaved  %8.3f ms %7.0f\n", t, cyc(t));
  t = run<7, 128>(8, out, iters); printf("8 waves MFMA + 64 pk_fma interleaved  %8.3f ms %7.0f\n", t, cyc(t));
  return 0;
}
