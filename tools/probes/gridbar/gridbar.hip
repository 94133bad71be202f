// Probe: cost of a hand-rolled grid barrier on MI355X (256 workgroups x 256 threads, one per CU).
// Build: hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar ; run: ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// sense-reversing counter barrier; bounded spin (a lost arrival sets *err and lets everybody leave)
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned* err, unsigned nblocks, unsigned& phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned target = (phase + 1) * nblocks;
    atomicAdd(counter, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { atomicExch(err, 1u); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  ++phase;
  __syncthreads();
}

__global__ __launch_bounds__(256) void k(unsigned* counter, unsigned* err, float* buf, int nbar, int n) {
  unsigned phase = 0;
  const int tid = blockIdx.x * 256 + threadIdx.x, total = gridDim.x * 256;
  for (int b = 0; b < nbar; ++b) {
    // a little dependent work across the barrier: every thread reads its neighbour block's value of the previous phase
    const int src = (tid + 256) % total;
    const float v = buf[(size_t)(b & 1) * n + src];
    buf[(size_t)((b + 1) & 1) * n + tid] = v + 1.0f;
    grid_barrier(counter, err, gridDim.x, phase);
  }
}

int main() {
  const int nb = 256, n = nb * 256;
  unsigned *counter, *err;
  float* buf;
  hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&buf, sizeof(float) * 2 * n);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int nbar : {1, 8, 16, 32, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(counter, 0, 4); hipMemset(err, 0, 4); hipMemset(buf, 0, sizeof(float) * 2 * n);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, counter, err, buf, nbar, n);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    std::vector<float> h(2 * n); hipMemcpy(h.data(), buf, sizeof(float) * 2 * n, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && h[(size_t)(nbar & 1) * n + i] == (float)nbar;
    printf("%3d barriers: %.1f us total -> %.2f us per barrier (err %u, values %s)\n", nbar, best * 1e3, best * 1e3 / nbar, e, ok ? "ok" : "WRONG");
  }
  return 0;
}
