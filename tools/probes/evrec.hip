// What does a fork cost the recording stream?  A chain of N dependent ~5 us kernels on stream A with, after every kernel,
//   (0) nothing, (1) hipEventRecord (default event) + hipStreamWaitEvent on stream B + a small kernel on B,
//   (2) the same with a hipEventDisableSystemFence event, (3) the event passed as stopEvent of hipExtLaunchKernelGGL
//   (no separate record), (4) = (3) with hipEventDisableSystemFence.
// Build: hipcc --offload-arch=gfx950 -O2 -o evrec evrec.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
int main() {
  const int N = 200;
  float *a, *b;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  for (int mode = 0; mode < 8; ++mode) {
    std::vector<hipEvent_t> ev(N);
    // modes 5-7: JOIN cost - stream A waits, before every kernel, for an event of stream B that completed long ago
    // (5: default event, 6: hipEventDisableSystemFence event, 7: stop event of B's kernel, device scope)
    const unsigned fl = hipEventDisableTiming | ((mode == 2 || mode == 4 || mode >= 6) ? hipEventDisableSystemFence : 0);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, fl));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto h0 = std::chrono::steady_clock::now();
      if (mode >= 5) {
        for (int i = 0; i < N; ++i) {
          if (mode == 7) hipExtLaunchKernelGGL(spin, dim3(4), dim3(256), 0, sb, nullptr, ev[i], 0, b, 200);
          else { hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, sb, b, 200); CK(hipEventRecord(ev[i], sb)); }
        }
        CK(hipStreamSynchronize(sb));
      }
      CK(hipEventRecord(t0, sa));
      for (int i = 0; i < N; ++i) {
        if (mode >= 5) {
          CK(hipStreamWaitEvent(sa, ev[i], 0));
          hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, a, 300);
          continue;
        }
        if (mode >= 3) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, nullptr, ev[i], 0, a, 300);
        else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, a, 300);
        if (mode == 1 || mode == 2) CK(hipEventRecord(ev[i], sa));
        if (mode >= 1) {
          CK(hipStreamWaitEvent(sb, ev[i], 0));
          hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, sb, b, 200);
        }
      }
      CK(hipEventRecord(t1, sa));
      auto h1 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t1));
      printf("mode %d rep %d: stream A %.2f us per kernel (host enqueue %.2f us per iteration)\n", mode, rep, ms * 1000 / N,
             std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
    }
    for (auto& e : ev) CK(hipEventDestroy(e));
  }
  return 0;
}
