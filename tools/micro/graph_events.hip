// Does a captured hipGraph replay time its kernels through event-record nodes (hipEventRecordWithFlags + hipEventRecordExternal)?
// Variations: stream kind, eager work + cross-stream wait before the capture, many event pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
  float* d; CK(hipMalloc(&d, 4096));
  for (int var = 0; var < 4; ++var) {
    hipStream_t s, other;
    if (var & 1) { CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0)); } else { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    const int NP = (var & 2) ? 400 : 1;
    std::vector<hipEvent_t> ev(2 * NP);
    for (auto& e : ev) CK(hipEventCreate(&e));
    // eager work and a cross-stream dependency before the capture (what torch's wait_stream does)
    hipEvent_t dep; CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    spin<<<1, 64, 0, other>>>(d, 10); CK(hipEventRecord(dep, other)); CK(hipStreamWaitEvent(s, dep, 0));
    spin<<<1, 64, 0, s>>>(d, 10);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    spin<<<1, 64, 0, s>>>(d, 1000);
    for (int i = 0; i < NP; ++i) {
      CK(hipEventRecordWithFlags(ev[2 * i], s, hipEventRecordExternal));
      spin<<<1, 64, 0, s>>>(d, 20000);
      CK(hipEventRecordWithFlags(ev[2 * i + 1], s, hipEventRecordExternal));
    }
    CK(hipStreamEndCapture(s, &g));
    size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n)); printf("variation %d (priority stream %d, %d event pairs): graph has %zu nodes\n", var, var & 1, NP, n);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      float ms = -1, tot = 0; hipError_t e = hipSuccess;
      for (int i = 0; i < NP; ++i) { e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); tot += ms; }
      printf("  replay %d: elapsed %s, mean %.4f ms per pair\n", rep, hipGetErrorString(e), tot / NP);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
