// Kernel launches of libgdf.so go through gdf::launch_kernel (the hipLaunchKernelGGL spelling is kept at the 56 launch sites and redirected
// here).  Normally that is hipLaunchKernel.  While the calling THREAD is recording a plan (model.cpp run_ops_graph) the same call appends a
// kernel node to that plan's hipGraph instead — the graph is built with the explicit graph API, not by stream capture (round 6).
// Why: a stream capture is a device-wide state on this runtime.  While one thread's plan stream was capturing, another thread's
// hipFree / hipMalloc / hipGraphExecDestroy (a model being garbage-collected), a second capture, or a plain torch.cuda.synchronize()
// (hipDeviceSynchronize is illegal beside ANY capture, relaxed mode included — as with CUDA) invalidated the capture; the invalidated stream
// stayed unusable and a second hipStreamEndCapture on it crashed inside the runtime (tools/micro/thread_race.py, tools/soak.py).  With
// explicit construction nothing is ever "being captured": other host threads allocate, free, synchronise and record freely (one extractor
// per thread is a supported mode: reference correspondence/correspondence/aggregation_network.py:67-95).  Nodes form one chain in launch
// order — what a single-stream capture produced.
#pragma once
#include <hip/hip_runtime.h>
#include <tuple>
#include <type_traits>
#include <utility>

namespace gdf {

struct GraphRecorder {
  hipGraph_t graph = nullptr;
  hipGraphNode_t last = nullptr;     // tail of the chain (null: the next node has no dependency)
  hipError_t err = hipSuccess;       // first error of the recording
  long nodes = 0;
};
// the recorder of the calling thread (nullptr = launch for real)
GraphRecorder*& thread_recorder();
hipError_t record_kernel_node(GraphRecorder& r, const void* fn, dim3 grid, dim3 block, void** args, unsigned smem);
hipError_t record_event_node(GraphRecorder& r, hipEvent_t ev);

template <typename Tuple, size_t... I>
inline void arg_pointers(Tuple& t, void** out, std::index_sequence<I...>) {
  ((out[I] = (void*)&std::get<I>(t)), ...);
}

template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned smem, hipStream_t stream, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  std::tuple<std::decay_t<KArgs>...> vals(std::forward<Args>(args)...);      // converted to the kernel's own parameter types
  void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {};
  arg_pointers(vals, ptrs, std::index_sequence_for<KArgs...>{});
  GraphRecorder* r = thread_recorder();
  if (r) {
    const hipError_t e = record_kernel_node(*r, (const void*)kernel, grid, block, ptrs, smem);
    if (e != hipSuccess && r->err == hipSuccess) r->err = e;
  } else {
    (void)hipLaunchKernel((const void*)kernel, grid, block, ptrs, smem, stream);
  }
}

}  // namespace gdf

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) ::gdf::launch_kernel(kernel, dim3(grid), dim3(block), (unsigned)(smem), stream, ##__VA_ARGS__)
