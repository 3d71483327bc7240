// Does `s_waitcnt vmcnt(N)` still mean "everything but the N youngest loads has landed" when the youngest loads are OUT-OF-RANGE buffer loads?
// The two-group GEMM main loops (csrc/gemm_mainloop_8phase.h) staged tiles past the end of K as out-of-range `buffer_load ... lds` (zeros, no memory
// access) to keep one counted wait per K-tile.  This probe: every wave first fills its LDS slot with a sentinel, issues ONE real LDS-DMA load from a
// cold address (a different 128-B line per wave and round, far apart), then NYOUNG out-of-range LDS-DMA loads to other slots, waits vmcnt(NYOUNG) and
// immediately reads the first slot.  With in-order retirement the slot holds the loaded data; every sentinel read counts as a violation.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmcnt_order tools/micro/vmcnt_order.hip && /tmp/vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define LDS_AS __attribute__((address_space(3)))
static constexpr uint32_t OOB = 0x80000000u;
static constexpr int NYOUNG = 6;

__global__ __launch_bounds__(256) void probe(const uint32_t* src, size_t src_bytes, size_t stride_bytes, int rounds, unsigned long long* violations,
                                             unsigned long long* checks, int use_oob) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(src_bytes < 0x7fffffffu ? src_bytes : 0x7fffffffu), 0x00020000);
  char* slot = smem + wave * 8192;                      // 8 slots of 1 KiB per wave: slot 0 = the real load, 1..6 = the young ones
  unsigned long long bad = 0, n = 0;
  for (int r = 0; r < rounds; ++r) {
    // sentinel into slot 0 (16 B per lane), retired before the DMA is issued
    *(uint4*)(slot + lane * 16) = uint4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const size_t line = ((size_t)(blockIdx.x * 4 + wave) * 977u + (size_t)r * 7919u) % (src_bytes / stride_bytes);
    const uint32_t off = (uint32_t)(line * stride_bytes) + lane * 16u;          // one 1-KiB row of a cold region
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)slot, 16, off, 0, 0, 0);
#pragma unroll
    for (int j = 1; j <= NYOUNG; ++j) {
      const uint32_t yo = use_oob ? OOB : (uint32_t)(((line + j * 13) % (src_bytes / stride_bytes)) * stride_bytes) + lane * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(slot + j * 1024), 16, yo, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NYOUNG) : "memory");
    const uint4 got = *(const uint4*)(slot + lane * 16);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the source holds its own byte offset / 4 in every word
    const uint32_t want = off / 4;
    if (got.x != want) ++bad;
    ++n;
  }
  atomicAdd(violations, bad);
  atomicAdd(checks, n);
}

// Second question: when vmcnt says an OUT-OF-RANGE LDS-DMA load is complete, has its LDS write (zeros) been performed?  Slot 1 is filled with a
// sentinel, one out-of-range load targets it, the wave waits vmcnt(0) and reads the slot at once and again after a delay.
__global__ __launch_bounds__(256) void probe_oob_write(const uint32_t* src, size_t src_bytes, int rounds, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(src_bytes < 0x7fffffffu ? src_bytes : 0x7fffffffu), 0x00020000);
  char* slot = smem + wave * 8192 + 1024;
  unsigned long long untouched = 0, zero_now = 0, zero_later = 0, other = 0;
  for (int r = 0; r < rounds; ++r) {
    *(uint4*)(slot + lane * 16) = uint4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)slot, 16, OOB, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 a = *(const uint4*)(slot + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < 64; ++k) __builtin_amdgcn_s_sleep(8);
    const uint4 b = *(const uint4*)(slot + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a.x == 0) ++zero_now;
    else if (a.x == 0xDEADBEEFu && b.x == 0) ++zero_later;
    else if (a.x == 0xDEADBEEFu && b.x == 0xDEADBEEFu) ++untouched;
    else ++other;
  }
  atomicAdd(out + 0, zero_now); atomicAdd(out + 1, zero_later); atomicAdd(out + 2, untouched); atomicAdd(out + 3, other);
}

__global__ void hammer(float* p, size_t n, int iters) {      // a second stream's traffic: keeps the memory path busy
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it)
    for (size_t k = i; k < n; k += (size_t)gridDim.x * blockDim.x) acc += p[k];
  if (acc == 123.456f) p[0] = acc;
}

int main() {
  const size_t bytes = 1ull << 30, stride = 1 << 16;          // 1 GiB, rows 64 KiB apart: every real load misses every cache
  uint32_t* src; hipMalloc(&src, bytes);
  std::vector<uint32_t> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
  hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  float* ham; const size_t hn = 1ull << 28; hipMalloc(&ham, hn * 4); hipMemset(ham, 0, hn * 4);
  unsigned long long *d; hipMalloc(&d, 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int mode = 0; mode < 4; ++mode) {
    const int use_oob = mode & 1, loaded = mode >> 1;
    hipMemset(d, 0, 16);
    if (loaded) hipLaunchKernelGGL(hammer, dim3(2048), dim3(256), 0, s2, ham, hn, 6);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 65536, s1, src, bytes, stride, 400, d, d + 1, use_oob);
    hipDeviceSynchronize();
    unsigned long long r[2]; hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("young loads %-12s %-28s: %llu of %llu waits returned with the OLD load still in flight\n", use_oob ? "OUT OF RANGE" : "in range",
           loaded ? "beside a bandwidth-bound kernel" : "alone", r[0], r[1]);
  }
  unsigned long long* o; hipMalloc(&o, 32); hipMemset(o, 0, 32);
  hipLaunchKernelGGL(probe_oob_write, dim3(1024), dim3(256), 65536, s1, src, bytes, 200, o);
  hipDeviceSynchronize();
  unsigned long long q[4]; hipMemcpy(q, o, 32, hipMemcpyDeviceToHost);
  printf("out-of-range LDS-DMA load, slot read right after vmcnt(0): zeros %llu, still the sentinel but zeros ~2 us later %llu, never written %llu, other %llu\n",
         q[0], q[1], q[2], q[3]);
  return 0;
}
