// Probe: how fast can 134 MB (4 x 32 planes of 1 MB) be WRITTEN with different store shapes?  (tools/probes, not shipped)
//   mode 0: wave instruction = 2 x 128 B (lanes 0-31 one plane, lanes 32-63 another), dword per lane  [MFMA D layout]
//   mode 1: wave instruction = 256 B contiguous of one plane, dword per lane
//   mode 2: wave instruction = 1 KB contiguous of one plane, 16 B per lane
//   mode 3: wave instruction = 4 planes x 256 B, 16 B per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void wk(float* y, int mode, int OS, int tiles_per_wave) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wv;  // global wave
  const int n = blockIdx.y;
  float* yn = y + (size_t)n * 32 * OS;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int pos0 = (gw * tiles_per_wave + t) * 64;  // 64 positions x 32 planes per tile
    if (pos0 >= OS) return;
    if (mode == 0) {
      for (int r = 0; r < 32; ++r)  // 32 instrs x 2 tiles of 32 columns: emulate [rows pairs]
        for (int half = 0; half < 1; ++half) {
          const int row = (r & 15) * 2 + (lane >> 5), col = (r >> 4) * 32 + (lane & 31);
          yn[(size_t)row * OS + pos0 + col] = (float)r;
        }
    } else if (mode == 1) {
      for (int r = 0; r < 32; ++r) yn[(size_t)r * OS + pos0 + lane] = (float)r;
    } else if (mode == 2) {
      // a wave takes 256 positions x 8 planes per "tile" to keep bytes equal: 1 KB contiguous per instruction
      const int p0 = (gw * tiles_per_wave + t) / 4 * 256, g = (gw * tiles_per_wave + t) % 4;
      if (p0 >= OS) return;
      for (int r = 0; r < 8; ++r)
        *reinterpret_cast<float4*>(yn + (size_t)(g * 8 + r) * OS + p0 + lane * 4) = make_float4(r, r, r, r);
    } else {
      for (int r = 0; r < 8; ++r)
        *reinterpret_cast<float4*>(yn + (size_t)(r * 4 + (lane >> 4)) * OS + pos0 + (lane & 15) * 4) = make_float4(r, r, r, r);
    }
  }
}
int main() {
  const int OS = 64 * 64 * 64, N = 4, NBUF = 6;
  std::vector<float*> bufs(NBUF);
  for (auto& b : bufs) hipMalloc(&b, (size_t)N * 32 * OS * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode)
    for (int tpw : {1, 4}) {
      const int waves = OS / 64 / tpw;
      dim3 grid(waves / 4, N);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wk, grid, dim3(256), 0, 0, bufs[i % NBUF], mode, OS, tpw);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 30;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wk, grid, dim3(256), 0, 0, bufs[i % NBUF], mode, OS, tpw);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, mb = (double)N * 32 * OS * 4 / 1e6;
      printf("mode %d tiles/wave %d: %.1f us  %.2f TB/s\n", mode, tpw, us, mb / us / 1e6 * 1e6 / 1e6);
    }
  return 0;
}
