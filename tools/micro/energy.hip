// What does a byte cost in WATTS?  The SDXL step runs at the 1400-W package cap with the shader clock throttled to ~1.9 GHz (tools/power_probe.py), so
// throughput is set by energy per image, not by cycles.  This program loops single-purpose kernels for ~2 s each while a host thread samples the
// card's hwmon power / clock, and prints watts, MHz and the rate:
//   read streams whose footprint lives in L2 (2 MB per XCD), in the Infinity Cache (128 MB) or in HBM (8 GB); LDS fragment reads; pure MFMA
//   (16x16x32 f16, random / zero operands); an idle spin.  pJ per byte / per flop = (watts - spin watts) / rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/energy tools/micro/energy.hip -lpthread && tools/micro/build/energy
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#include <cctype>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every workgroup re-reads its own slice `reps` times, 16 B per lane, 8 loads in flight
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t slice16, int reps, unsigned* out) {
  const u32x4* base = p + (size_t)blockIdx.x * slice16;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x; i < slice16; i += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + ((i + (size_t)u * 256) < slice16 ? i + (size_t)u * 256 : i));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  if (acc.x == 0x12345678u && acc.y == 1) out[0] = acc.z ^ acc.w;
}
// plain (cacheable) variant: L2 / MALL residency
__global__ __launch_bounds__(256) void read_kernel_c(const u32x4* __restrict__ p, size_t slice16, int reps, unsigned* out) {
  const u32x4* base = p + (size_t)blockIdx.x * slice16;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x; i < slice16; i += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(i + (size_t)u * 256) < slice16 ? i + (size_t)u * 256 : i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
      asm volatile("" : "+v"(acc));
    }
  }
  if (acc.x == 0x12345678u && acc.y == 1) out[0] = acc.z ^ acc.w;
}
__global__ __launch_bounds__(256) void write_kernel(u32x4* p, size_t slice16, int reps) {
  u32x4* base = p + (size_t)blockIdx.x * slice16;
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < slice16; i += 256) base[i] = u32x4{(unsigned)i, (unsigned)r, 3u, 4u};
}
__global__ __launch_bounds__(512, 2) void lds_read_kernel(const f16x8* in, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16x8* s = (f16x8*)smem;
  for (int i = threadIdx.x; i < 4096; i += 512) s[i] = in[i];
  __syncthreads();
  f16x8 a = s[threadIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f16x8 v = s[((lane ^ (u & 7)) + 64 * ((w + u) & 7) + 512 * (u >> 3)) & 4095];      // conflict-free 16-B reads, 1 KiB per wave-instruction
      a += v;
    }
  }
  float r = 0; for (int e = 0; e < 8; ++e) r += (float)a[e];
  if (r == 12345.f) out[0] = r;
}
template <int NA, int NB>
__global__ __launch_bounds__(512, 2) void mfma_kernel(const f16x8* __restrict__ in, float* out, int iters) {
  f16x8 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095];
  for (int j = 0; j < NB; ++j) b[j] = in[(threadIdx.x * 13 + j * 257 + blockIdx.x * 3 + 1024) & 4095];
  f32x4 acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  float s = 0;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  if (s == 12345.f) out[0] = s;
}
// Does the ORDER of the MFMAs of a register tile matter for power?  4 x 8 fragments, 32 accumulators, the same 32 products per trip:
//   MODE 0 row-major (a[i] fixed for 8 MFMAs, b changes every time, both change at a row change)     <- the GEMM kernels' order
//   MODE 1 snake (exactly one operand changes between consecutive MFMAs)
//   MODE 2 diagonal (BOTH operands change at every MFMA)
//   MODE 3 one fixed pair a[0], b[0] for all 32 accumulators (no operand change at all; random values)
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_order_kernel(const f16x8* __restrict__ in, float* out, int iters) {
  constexpr int NA = 4, NB = 8;
  f16x8 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095];
  for (int j = 0; j < NB; ++j) b[j] = in[(threadIdx.x * 13 + j * 257 + blockIdx.x * 3 + 1024) & 4095];
  f32x4 acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NA * NB; ++k) {
      int i, j;
      if (MODE == 0) { i = k / NB; j = k % NB; }
      else if (MODE == 1) { i = k / NB; j = (i & 1) ? NB - 1 - k % NB : k % NB; }
      else if (MODE == 2) { i = k % NA; j = (k % NA + k / NA) % NB; }
      else { i = k / NB; j = k % NB; }
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(MODE == 3 ? a[0] : a[i], MODE == 3 ? b[0] : b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  float s = 0;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  if (s == 12345.f) out[0] = s;
}
// Does the SHAPE of the MFMA matter for power?  The same 128 accumulator registers per lane as the 4 x 8 tile of 16x16x32 above, held as 2 x 4 tiles of
// 32x32x16 (each operand register then feeds 32 output columns instead of 16), walked in snake order.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int NA, int NB>
__global__ __launch_bounds__(512, 2) void mfma32_kernel(const f16x8* __restrict__ in, float* out, int iters) {
  f16x8 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095];
  for (int j = 0; j < NB; ++j) b[j] = in[(threadIdx.x * 13 + j * 257 + blockIdx.x * 3 + 1024) & 4095];
  f32x16_t acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NA * NB; ++k) {
      const int i = k / NB, j = (i & 1) ? NB - 1 - k % NB : k % NB;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  float s = 0;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(512, 2) void spin_kernel(int iters) {
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(512, 2) void valu_kernel(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  float e = a + 1, f = a + 2, g = a + 3, h = a + 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { a = a * b + c; e = e * b + d; f = f * b + c; g = g * b + d; h = h * b + a; }
  }
  if (a + e + f + g + h == 12345.f) out[0] = a;
}

struct Sampler {
  std::vector<std::string> pw, fq;
  std::atomic<bool> stop{false};
  std::vector<std::vector<long>> p, f;
  std::thread th;
  static long rd(const std::string& s) { FILE* fp = fopen(s.c_str(), "r"); if (!fp) return -1; long v = -1; if (fscanf(fp, "%ld", &v) != 1) v = -1; fclose(fp); return v; }
  Sampler() {
    // our card = the one whose PCI address is HIP device 0's (other cards of the node run other tenants' jobs)
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) bus[0] = 0;
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return;
    while (dirent* e = readdir(d)) {
      if (strncmp(e->d_name, "card", 4) || strchr(e->d_name, '-')) continue;
      char link[512] = {0};
      const std::string devp = std::string("/sys/class/drm/") + e->d_name + "/device";
      const ssize_t ln = readlink(devp.c_str(), link, sizeof link - 1);
      if (bus[0] && ln > 0 && !strstr(link, bus)) continue;
      std::string hm = devp + "/hwmon";
      DIR* h = opendir(hm.c_str());
      if (!h) continue;
      while (dirent* e2 = readdir(h)) {
        if (strncmp(e2->d_name, "hwmon", 5)) continue;
        std::string base = hm + "/" + e2->d_name;
        if (rd(base + "/power1_input") >= 0) { pw.push_back(base + "/power1_input"); fq.push_back(base + "/freq1_input"); }
      }
      closedir(h);
    }
    closedir(d);
    p.resize(pw.size()); f.resize(pw.size());
  }
  void start() { stop = false; for (auto& v : p) v.clear(); for (auto& v : f) v.clear();
    th = std::thread([this] { while (!stop) { for (size_t i = 0; i < pw.size(); ++i) { p[i].push_back(rd(pw[i])); f[i].push_back(rd(fq[i])); }
                                                std::this_thread::sleep_for(std::chrono::milliseconds(20)); } }); }
  // mean watts / MHz of the busiest card over the last `frac` of the samples
  void finish(double& watts, double& mhz) {
    stop = true; th.join(); watts = mhz = 0;
    for (size_t i = 0; i < pw.size(); ++i) {
      const size_t n = p[i].size(), s0 = n / 3;
      double a = 0, b = 0; size_t c = 0;
      for (size_t k = s0; k < n; ++k) { a += p[i][k]; b += f[i][k]; ++c; }
      if (c && a / c / 1e6 > watts) { watts = a / c / 1e6; mhz = b / c / 1e6; }
    }
  }
};

template <class F>
static void run(const char* name, Sampler& S, F launch, double units_per_launch, const char* unit, double secs = 2.0) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  int n = (int)(secs * 1e3 / (ms > 0.01f ? ms : 0.01f)); if (n < 3) n = 3;
  S.start();
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  double w, mhz; S.finish(w, mhz);
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double rate = units_per_launch * n / (ms * 1e-3);
  printf("%-44s %8.1f W %6.0f MHz  %10.3f %s\n", name, w, mhz, rate / 1e12, unit); fflush(stdout);
}

int main(int argc, char** argv) {
  Sampler S;
  printf("hwmon cards matched to HIP device 0: %zu (%s)\n", S.pw.size(), S.pw.empty() ? "-" : S.pw[0].c_str());
  const size_t big = 8ull << 30;
  char* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
  unsigned* out; CK(hipMalloc(&out, 64));
  f16x8* rnd; CK(hipMalloc(&rnd, 4096 * 16));
  { std::vector<_Float16> h(4096 * 8); for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f); CK(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
  f16x8* zer; CK(hipMalloc(&zer, 4096 * 16)); CK(hipMemset(zer, 0, 4096 * 16));
  CK(hipFuncSetAttribute((const void*)lds_read_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

  if (argc > 1 && !strcmp(argv[1], "mfma-order")) {
    const double fl = 512.0 * 8 * 32 * 16384 * 4000;
    run("idle spin (s_sleep), 512 WGs", S, [&] { hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(512), 0, 0, 20000); }, 0, "-");
    for (int rep = 0; rep < 2; ++rep) {
      run("MFMA 4x8 tile, row-major order", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<0>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, fl, "TFLOP/s");
      run("MFMA 4x8 tile, snake order", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<1>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, fl, "TFLOP/s");
      run("MFMA 4x8 tile, diagonal (both change)", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<2>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, fl, "TFLOP/s");
      run("MFMA one fixed random operand pair", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<3>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, fl, "TFLOP/s");
      run("MFMA 4x8 tile, row-major, zero operands", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<0>), dim3(512), dim3(512), 0, 0, zer, (float*)out, 4000); }, fl, "TFLOP/s");
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "mfma-shape")) {
    const double fl16 = 512.0 * 8 * 32 * 16384 * 4000, fl32 = 512.0 * 8 * 8 * 32768 * 8000;
    run("idle spin (s_sleep), 512 WGs", S, [&] { hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(512), 0, 0, 20000); }, 0, "-");
    for (int rep = 0; rep < 3; ++rep) {
      run("MFMA 16x16x32 f16, 4x8 tile, snake, random", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<1>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, fl16, "TFLOP/s");
      run("MFMA 32x32x16 f16, 2x4 tile, snake, random", S, [&] { hipLaunchKernelGGL((mfma32_kernel<2, 4>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 8000); }, fl32, "TFLOP/s");
    }
    run("MFMA 16x16x32 f16, 4x8 tile, snake, zeros", S, [&] { hipLaunchKernelGGL((mfma_order_kernel<1>), dim3(512), dim3(512), 0, 0, zer, (float*)out, 4000); }, fl16, "TFLOP/s");
    run("MFMA 32x32x16 f16, 2x4 tile, snake, zeros", S, [&] { hipLaunchKernelGGL((mfma32_kernel<2, 4>), dim3(512), dim3(512), 0, 0, zer, (float*)out, 8000); }, fl32, "TFLOP/s");
    return 0;
  }
  run("idle spin (s_sleep), 512 WGs", S, [&] { hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(512), 0, 0, 20000); }, 0, "-");
  const int WG = 2048;
  { const size_t slice = (1ull << 20) / 16 / 8;      // 8 KB per WG x 2048 = 16 MB total; each XCD's 256 WGs touch 2 MB: L2-resident
    run("read, L2-resident (8 KB / WG, 2 MB / XCD)", S, [&] { hipLaunchKernelGGL(read_kernel_c, dim3(WG), dim3(256), 0, 0, (const u32x4*)buf, slice * 16 / 16, 4000, out); }, (double)WG * slice * 16 * 4000, "TB/s"); }
  { const size_t slice = (128ull << 20) / WG / 16;   // 128 MB total: Infinity-Cache-resident, misses L2
    run("read, MALL-resident (128 MB)", S, [&] { hipLaunchKernelGGL(read_kernel_c, dim3(WG), dim3(256), 0, 0, (const u32x4*)buf, slice, 40, out); }, (double)WG * slice * 16 * 40, "TB/s"); }
  { const size_t slice = big / WG / 16;              // 8 GB: HBM
    run("read, HBM (8 GB)", S, [&] { hipLaunchKernelGGL(read_kernel_c, dim3(WG), dim3(256), 0, 0, (const u32x4*)buf, slice, 1, out); }, (double)WG * slice * 16, "TB/s"); }
  { const size_t slice = big / WG / 16;
    run("write, HBM (8 GB)", S, [&] { hipLaunchKernelGGL(write_kernel, dim3(WG), dim3(256), 0, 0, (u32x4*)buf, slice, 1); }, (double)WG * slice * 16, "TB/s"); }
  run("LDS ds_read_b128 only (8 waves / CU)", S, [&] { hipLaunchKernelGGL(lds_read_kernel, dim3(512), dim3(512), 65536, 0, rnd, (float*)out, 20000); }, 512.0 * 512 * 16 * 16 * 20000, "TB/s");
  run("VALU v_fma_f32 only", S, [&] { hipLaunchKernelGGL(valu_kernel, dim3(512), dim3(512), 0, 0, (float*)out, 20000); }, 512.0 * 512 * 80 * 2 * 20000, "TFLOP/s");
  run("MFMA 16x16x32 f16 only, random operands", S, [&] { hipLaunchKernelGGL((mfma_kernel<4, 8>), dim3(512), dim3(512), 0, 0, rnd, (float*)out, 4000); }, 512.0 * 8 * 32 * 16384 * 4000, "TFLOP/s");
  run("MFMA 16x16x32 f16 only, zero operands", S, [&] { hipLaunchKernelGGL((mfma_kernel<4, 8>), dim3(512), dim3(512), 0, 0, zer, (float*)out, 4000); }, 512.0 * 8 * 32 * 16384 * 4000, "TFLOP/s");
  return 0;
}
