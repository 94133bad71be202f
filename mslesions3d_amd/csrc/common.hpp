// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the 3D-SSD hot path.
// Wavefront = 64 lanes everywhere.  All reductions here have a FIXED combination order, so every
// kernel that uses them is run-to-run bit-reproducible (no atomics on floating point anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSL_WAVE 64
#define MSL_OK 0
#define MSL_ERR_ARG (-1)
#define MSL_ERR_UNSUPPORTED (-2)

#define MSL_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

namespace msl {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // lane 0 holds the total
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum over the whole block (blockDim.x multiple of 64, <= 1024).  Result valid in thread 0.
// `scratch` must hold >= blockDim.x/64 doubles; the caller must __syncthreads() before reusing it.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < nw; ++i) t += scratch[i];
  }
  return t;
}

// relu(x * s + t) when affine is requested, else x.  fmaf is what every consumer uses, so all
// kernels that re-create an activation from (raw conv output, scale, shift) agree bit for bit.
// NaN must survive the ReLU (torch's relu propagates it; fmaxf would swallow it and hide a diverged run from
// the NaN guards of ssd3d.py:95-98,:258-261).
__device__ __forceinline__ float act(float x, float s, float t) {
  const float v = fmaf(x, s, t);
  return v < 0.0f ? 0.0f : v;
}

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }


// out[i] = sum_k slabs[k * stride + i], k = 0..nslabs-1.  32 outputs x 8 slab groups per workgroup; every thread
// walks its slabs with independent loads (deep memory-level parallelism), then the 8 group sums are folded
// in a fixed order -> bit-reproducible.  Launch with cdiv(count, 32) workgroups of 256 threads.
__device__ __forceinline__ float reduce_slabs_256(const float* __restrict__ slabs, size_t stride, int count,
                                                  int nslabs, float* lds /* [8][33] */) {
  const int li = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  float s = 0.f;
  if (i < count) {
    const float* p = slabs + i;
    int k = g;
#pragma unroll 1
    for (; k + 56 < nslabs; k += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 8 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nslabs; k += 8) s += p[(size_t)k * stride];
  }
  lds[g * 33 + li] = s;
  __syncthreads();
  float t = 0.f;
  if (g == 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t += lds[u * 33 + li];
  }
  return t;  // valid for g == 0 && i < count
}

}  // namespace msl
