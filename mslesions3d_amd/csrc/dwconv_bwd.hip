// Depthwise Conv3d backward (autograd of Block.conv1, lesions3d/mobilenet.py:38,44).
//   bwd-data  : g_in[n,c,i] = sum_k w[c,k] * dy[n,c,(i + 1 - k)/s]   (only taps with integral, in-range o)
//   bwd-weight: dw[c,k]     = sum_{n,o} dy[n,c,o] * a_in[n,c,o*s - 1 + k],  a_in = relu(bn(x)) rebuilt on load
// Both are HBM-bound like the forward.  bwd-data writes 16 B per lane along W (the 8x larger tensor for
// stride 2) and reads dy through the cache; the per-parity tap sets of a stride-2 transpose conv are
// enumerated explicitly instead of testing 27 taps.  bwd-weight keeps 27 accumulators per thread and ends
// in a fixed-order workgroup reduction to fp64 partials (no atomics -> bit-reproducible).
#include "common.hpp"
#include <algorithm>

namespace {

// contributing (o, k) pairs of one axis for input index i (pad 1, kernel 3)
struct Taps {
  int o[3], k[3], n;
};
template <int STRIDE>
__device__ __forceinline__ Taps axis_taps(int i, int O) {
  Taps t;
  t.n = 0;
  if (STRIDE == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int o = i + 1 - k;
      if (o >= 0 && o < O) {
        t.o[t.n] = o;
        t.k[t.n] = k;
        ++t.n;
      }
    }
  } else {
    if ((i & 1) == 0) {
      const int o = i >> 1;
      if (o < O) {
        t.o[0] = o;
        t.k[0] = 1;
        t.n = 1;
      }
    } else {
      const int o0 = (i + 1) >> 1, o2 = (i - 1) >> 1;
      if (o0 < O) {
        t.o[t.n] = o0;
        t.k[t.n] = 0;
        ++t.n;
      }
      if (o2 >= 0 && o2 < O) {
        t.o[t.n] = o2;
        t.k[t.n] = 2;
        ++t.n;
      }
    }
  }
  return t;
}

// grid (ceil(D*H*W/4 / 256), N*C); 4 consecutive iw per thread (W % 4 == 0)
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ g_in, int C, int D, int H, int W,
                                                          int OD, int OH, int OW, int accumulate) {
  const int nc = blockIdx.y, c = nc % C;
  const int W4 = W >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= D * H * W4) return;
  const int iw0 = (q % W4) * 4, ih = (q / W4) % H, id = q / (W4 * H);
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[c * 27 + k];
  const float* dyc = dy + (size_t)nc * OD * OH * OW;
  const Taps td = axis_taps<STRIDE>(id, OD), th = axis_taps<STRIDE>(ih, OH);
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int MAXT = STRIDE == 1 ? 3 : 2;
#pragma unroll
  for (int a = 0; a < MAXT; ++a) {
#pragma unroll
    for (int b = 0; b < MAXT; ++b) {
      if (a >= td.n || b >= th.n) continue;
      const float* row = dyc + ((size_t)td.o[a] * OH + th.o[b]) * OW;
      const int kb = td.k[a] * 9 + th.k[b] * 3;
      // select the 3 weights of this (kd, kh) without dynamic register indexing
      float w0 = 0.f, w1 = 0.f, w2 = 0.f;
#pragma unroll
      for (int s = 0; s < 9; ++s)
        if (kb == s * 3) {
          w0 = wk[s * 3];
          w1 = wk[s * 3 + 1];
          w2 = wk[s * 3 + 2];
        }
      if (STRIDE == 2) {
        const int a0 = iw0 >> 1;
        const float d0 = row[a0];
        const float d1 = a0 + 1 < OW ? row[a0 + 1] : 0.f;
        const float d2 = a0 + 2 < OW ? row[a0 + 2] : 0.f;
        g[0] = fmaf(w1, d0, g[0]);
        g[1] = fmaf(w0, d1, fmaf(w2, d0, g[1]));
        g[2] = fmaf(w1, d1, g[2]);
        g[3] = fmaf(w0, d2, fmaf(w2, d1, g[3]));
      } else {
        float d[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const int ow = iw0 - 1 + t;
          d[t] = (ow >= 0 && ow < OW) ? row[ow] : 0.f;
        }
        // g[v] = sum_kw w[kw] * dy[iw0 + v + 1 - kw]  ->  d index (v + 2 - kw)
#pragma unroll
        for (int v = 0; v < 4; ++v) g[v] = fmaf(w0, d[v + 2], fmaf(w1, d[v + 1], fmaf(w2, d[v], g[v])));
      }
    }
  }
  float* dst = g_in + (size_t)nc * D * H * W + ((size_t)id * H + ih) * W + iw0;
  if (accumulate) {
    const float4 old = *reinterpret_cast<const float4*>(dst);
    g[0] += old.x; g[1] += old.y; g[2] += old.z; g[3] += old.w;
  }
  *reinterpret_cast<float4*>(dst) = make_float4(g[0], g[1], g[2], g[3]);
}

// generic fallback: one input voxel per thread, any shape
__global__ __launch_bounds__(256) void dw_bwd_data_naive_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                float* __restrict__ g_in, int C, int D, int H, int W,
                                                                int OD, int OH, int OW, int stride, int accumulate) {
  const int nc = blockIdx.y, c = nc % C;
  const int S = D * H * W;
  const float* dyc = dy + (size_t)nc * OD * OH * OW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
    const int iw = i % W, ih = (i / W) % H, id = i / (W * H);
    float g = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int td = id + 1 - kd;
      if (td < 0 || td % stride != 0 || td / stride >= OD) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int t2 = ih + 1 - kh;
        if (t2 < 0 || t2 % stride != 0 || t2 / stride >= OH) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int t3 = iw + 1 - kw;
          if (t3 < 0 || t3 % stride != 0 || t3 / stride >= OW) continue;
          g = fmaf(w[c * 27 + kd * 9 + kh * 3 + kw], dyc[((size_t)(td / stride) * OH + t2 / stride) * OW + t3 / stride], g);
        }
      }
    }
    float* dst = g_in + (size_t)nc * S + i;
    *dst = accumulate ? *dst + g : g;
  }
}

// bwd-weight: grid (chunks, C, N); each block reduces `BW_CHUNK` outputs of one (n, c) into 27 fp64 partials
constexpr int BW_CHUNK = 4096;
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            double* __restrict__ partials, int C, int D, int H, int W,
                                                            int OD, int OH, int OW, int stride, int chunks) {
  __shared__ double scratch[8];
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int OS = OD * OH * OW;
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
  const float* xc = x + ((size_t)n * C + c) * D * H * W;
  const float* dyc = dy + ((size_t)n * C + c) * OS;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const int lo = chunk * BW_CHUNK, hi = min(OS, lo + BW_CHUNK);
  for (int o = lo + threadIdx.x; o < hi; o += 256) {
    const int ow = o % OW, oh = (o / OW) % OH, od = o / (OW * OH);
    const float d = dyc[o];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int id = od * stride - 1 + kd;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * stride - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow * stride - 1 + kw;
          const bool ok = id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
          float v = ok ? xc[((size_t)id * H + ih) * W + iw] : 0.f;
          if (affine) v = ok ? msl::act(v, sc, sh) : 0.f;
          acc[kd * 9 + kh * 3 + kw] = fmaf(d, v, acc[kd * 9 + kh * 3 + kw]);
        }
      }
    }
  }
  const int NP = gridDim.z * chunks, p = n * chunks + chunk;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const double t = msl::block_sum((double)acc[k], scratch);
    if (threadIdx.x == 0) partials[((size_t)c * 27 + k) * NP + p] = t;
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void dw_bwd_weight_finalize_kernel(const double* __restrict__ partials, int NP,
                                                                    float* __restrict__ dw, int count) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int p = lane; p < NP; p += 64) s += partials[(size_t)i * NP + p];
  s = msl::wave_sum(s);
  if (lane == 0) dw[i] = (float)s;
}

}  // namespace

extern "C" {

// dy (N,C,OD,OH,OW) -> g_in (N,C,D,H,W); accumulate != 0 adds into g_in
int msl_dwconv_bwd_data(const float* dy, const float* w, float* g_in, int N, int C, int D, int H, int W,
                        int stride, int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (W % 4 == 0) {
    dim3 grid(msl::cdiv(D * H * (W / 4), 256), N * C);
    if (stride == 2) hipLaunchKernelGGL(dw_bwd_data_kernel<2>, grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW, accumulate);
    else hipLaunchKernelGGL(dw_bwd_data_kernel<1>, grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW, accumulate);
  } else {
    dim3 grid(std::min(msl::cdiv(D * H * W, 256), 256), N * C);
    hipLaunchKernelGGL(dw_bwd_data_naive_kernel, grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW, stride, accumulate);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_dwconv_bwd_weight_num_partials(int N, int C, int D, int H, int W, int stride) {
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return N * msl::cdiv(OD * OH * OW, BW_CHUNK);
}

// dw (C,27) from dy and the raw input x (+ its affine); partials: fp64 [C*27][NP]
int msl_dwconv_bwd_weight(const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                          double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int chunks = msl::cdiv(OD * OH * OW, BW_CHUNK);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(chunks, C, N), dim3(256), 0, st, dy, x, in_scale, in_shift, partials,
                     C, D, H, W, OD, OH, OW, stride, chunks);
  MSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_bwd_weight_finalize_kernel, dim3(C * 27), dim3(64), 0, st, partials, N * chunks, dw, C * 27);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
