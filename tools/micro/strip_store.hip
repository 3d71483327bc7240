// Micro-benchmark: HBM write bandwidth of the attention-map store pattern on MI355X.
// A (B*heads, S, S) fp16 tensor is written by workgroups that each own 128 consecutive rows (4 waves x 32 rows) and walk
// the S columns in tiles of W bytes per row ("strips").  Variants: strip width 128..1024 B, lock-step vs staggered start.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/strip_store.hip -o gpurun_out/strip_store && gpurun_out/strip_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int W>   // strip bytes per row per step
__global__ __launch_bounds__(256, 3) void strips(_Float16* out, int S, int stagger) {
  constexpr int LPR = W / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per store instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row0 = (size_t)blockIdx.x * 128 + wave * 32;
  const int nsteps = S * 2 / W;
  const int t0 = stagger ? (blockIdx.x * 37u) % nsteps : 0;
  f16x8 v; for (int e = 0; e < 8; ++e) v[e] = (_Float16)(lane + e);
  for (int t = 0; t < nsteps; ++t) {
    int tt = t + t0; if (tt >= nsteps) tt -= nsteps;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const size_t r = row0 + it * RPI + lane / LPR;
      *(f16x8*)(out + r * S + (size_t)tt * (W / 2) + (lane % LPR) * 8) = v;
    }
  }
}

int main() {
  const int S = 4096, BH = 64;
  const size_t n = (size_t)BH * S * S;
  _Float16* d; hipMalloc(&d, n * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int stagger) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(BH * S / 128), dim3(256), 0, 0, d, S, stagger);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(BH * S / 128), dim3(256), 0, 0, d, S, stagger);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-10s stagger=%d  %.3f ms  %.2f TB/s\n", name, stagger, ms, n * 2 / ms / 1e9);
  };
  for (int st = 0; st < 2; ++st) {
    run(strips<128>, "strip128", st); run(strips<256>, "strip256", st); run(strips<512>, "strip512", st);
    run(strips<1024>, "strip1024", st);
  }
  hipMemsetAsync(d, 0, n * 2, 0); hipEventRecord(e0); hipMemsetAsync(d, 0, n * 2, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("memset     %.3f ms  %.2f TB/s\n", ms, n * 2 / ms / 1e9);
  return 0;
}
