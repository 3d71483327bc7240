// Which MFMA shape should the tile kernels multiply with?  gfx950, fp16 operands, fp32 accumulate, RANDOM register contents (the chip is
// power-limited under MFMA load: MI355X_MICROARCH.md "DVFS give-back" — zero-filled operands flatter every number).
//   S16: v_mfma_f32_16x16x32_f16  (16,384 FLOP; what csrc/gemm.hip uses)        S32: v_mfma_f32_32x32x16_f16  (32,768 FLOP; attn.hip)
// Register pattern of a GEMM wave tile: NA A fragments x NB B fragments (all pairs), accumulators resident; 1 or 2 waves per SIMD.
// Reports TFLOP/s at the chip's own clock and cycles per instruction per SIMD at the 2.4 GHz nominal clock.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape tools/micro/mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NA, int NB>
__global__ __launch_bounds__(512, 2) void k(const f16x8* __restrict__ in, float* out, int iters) {
  f16x8 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095];
  for (int j = 0; j < NB; ++j) b[j] = in[(threadIdx.x * 13 + j * 257 + blockIdx.x * 3 + 1024) & 4095];
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[NA][NB];
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      // keep the operands "fresh" without memory traffic: rotate fragments (no VALU: register renaming by the unrolled loop is enough)
      asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  } else {
    f32x16 acc[NA][NB];
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NA, int NB>
void run(const char* name, int waves, const f16x8* in, float* out) {
  const int iters = 40000;                                        // 15-25 ms per launch: the clock has settled
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<SHAPE, NA, NB><<<256, waves * 64>>>(in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, NA, NB><<<256, waves * 64>>>(in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_inst = (double)iters * NA * NB;                       // per wave
  const double flop = (SHAPE == 16 ? 16384.0 : 32768.0) * n_inst * waves * 256;
  const double cyc = ms * 1e-3 * 2.4e9 / (n_inst * (waves / 4.0));     // nominal cycles per instruction per SIMD
  printf("%-44s %d waves/CU  %8.3f ms  %8.1f TFLOP/s  %6.2f nominal cycles / instruction / SIMD\n", name, waves, ms, flop / ms / 1e9, cyc);
}

int main() {
  f16x8* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 512 * 4);
  _Float16* h = (_Float16*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f));      // uniform random [-1, 1)
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  for (int waves : {4, 8}) {
    if (waves == 8) {
      run<16, 4, 5>("16x16x32, 4 x 5 fragments (80 acc, 128x80 tile/2)", waves, in, out);
      run<16, 4, 8>("16x16x32, 4 x 8 fragments (128 acc, 64x128 tile)", waves, in, out);
      run<32, 2, 4>("32x32x16, 2 x 4 blocks    (128 acc, 64x128 tile)", waves, in, out);
      run<32, 2, 2>("32x32x16, 2 x 2 blocks    (64 acc)", waves, in, out);
    } else {
      run<16, 4, 8>("16x16x32, 4 x 8 fragments (128 acc)", waves, in, out);
      run<32, 2, 4>("32x32x16, 2 x 4 blocks    (128 acc)", waves, in, out);
    }
  }
  // zero-filled operands for comparison (clock give-back)
  hipMemset(in, 0, 4096 * 16);
  run<16, 4, 8>("16x16x32, 4 x 8 fragments, ZERO operands", 8, in, out);
  run<32, 2, 4>("32x32x16, 2 x 4 blocks,    ZERO operands", 8, in, out);
  return 0;
}
