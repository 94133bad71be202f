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

// Stride-2 specialisation: one thread owns the 2 x 2 x 4 input patch (id in {2a, 2a+1}) x (ih in {2b, 2b+1}) x
// (iw0 .. iw0+3).  The parity of every coordinate is then static: the 16 outputs read the same 12 dy values and
// every weight index is a compile-time constant (54 FMAs, no tap selection), and the patch is written as four
// 16-B stores.  This is the kernel that writes the 134 MB stem-gradient tensor.
// REDUCE: the tensor written here is dL/d relu(bn(y_prev)); its BatchNorm backward needs sum(gm) and sum(gm * xhat)
// per channel.  The kernel already holds every g value in registers, so it reads y_prev at the same 16 positions
// and emits the fp64 partials itself: the separate reduce pass (a second read of the 134 MB gradient at block 1)
// disappears.
template <bool REDUCE, typename T = float>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_patch_kernel(const T* __restrict__ dy,
                                                                   const float* __restrict__ w,
                                                                   T* __restrict__ g_in, int C, int D, int H, int W,
                                                                   int OD, int OH, int OW, int accumulate,
                                                                   const T* __restrict__ y_prev,
                                                                   const float* __restrict__ bn_scale,
                                                                   const float* __restrict__ bn_shift,
                                                                   const float* __restrict__ bn_mean,
                                                                   const float* __restrict__ bn_invstd,
                                                                   double* __restrict__ partials) {
  __shared__ double scratch[8];
  const int nc = blockIdx.y, c = nc % C;
  const int W4 = W >> 2, H2 = (H + 1) >> 1, D2 = (D + 1) >> 1;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const bool live = q0 < D2 * H2 * W4;
  if (!REDUCE && !live) return;
  const int q = live ? q0 : 0;
  const int cw = q % W4, b = (q / W4) % H2, a = q / (W4 * H2);
  const int iw0 = cw * 4, c2 = cw * 2;
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[c * 27 + k];
  const T* dyc = dy + (size_t)nc * OD * OH * OW;
  // every global read of the thread is issued here, unconditionally on clamped addresses (masked afterwards), so
  // they share one memory round trip: the 12 gradients, and the 4 destination rows' previous gradient (accumulate)
  // and forward value (REDUCE)
  float dv[2][2][3];
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int ww = 0; ww < 3; ++ww) {
        const bool ok = a + dd < OD && b + hh < OH && c2 + ww < OW;
        const float v = msl::ld1(dyc + (ok ? ((size_t)(a + dd) * OH + b + hh) * OW + c2 + ww : 0));
        dv[dd][hh][ww] = ok ? v : 0.f;
      }
  bool okp[2][2];
  size_t offp[2][2];
  float4 yv[2][2], old[2][2];
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int id = 2 * a + pd, ih = 2 * b + ph;
      okp[pd][ph] = live && id < D && ih < H;
      offp[pd][ph] = (size_t)nc * D * H * W + (okp[pd][ph] ? ((size_t)id * H + ih) * W + iw0 : 0);
      if (REDUCE) yv[pd][ph] = msl::ld4(y_prev + offp[pd][ph]);
    }
  if (accumulate) {
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) old[pd][ph] = msl::ld4(g_in + offp[pd][ph]);
  } else {
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) old[pd][ph] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (REDUCE) msl::pin(yv[pd][ph]);
      msl::pin(old[pd][ph]);
    }
  // per axis: even index -> (offset 0, k = 1); odd index -> (offset 1, k = 0) and (offset 0, k = 2)
  float g[2][2][4];
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
      for (int td = 0; td < (pd ? 2 : 1); ++td) {
        const int dd = pd ? (td == 0 ? 1 : 0) : 0, kd = pd ? (td == 0 ? 0 : 2) : 1;
#pragma unroll
        for (int th = 0; th < (ph ? 2 : 1); ++th) {
          const int hh = ph ? (th == 0 ? 1 : 0) : 0, kh = ph ? (th == 0 ? 0 : 2) : 1;
          const float w0 = wk[kd * 9 + kh * 3 + 0], w1 = wk[kd * 9 + kh * 3 + 1], w2 = wk[kd * 9 + kh * 3 + 2];
          const float d0 = dv[dd][hh][0], d1 = dv[dd][hh][1], d2 = dv[dd][hh][2];
          o0 = fmaf(w1, d0, o0);
          o1 = fmaf(w0, d1, fmaf(w2, d0, o1));
          o2 = fmaf(w1, d1, o2);
          o3 = fmaf(w0, d2, fmaf(w2, d1, o3));
        }
      }
      g[pd][ph][0] = o0; g[pd][ph][1] = o1; g[pd][ph][2] = o2; g[pd][ph][3] = o3;
    }
  float s1 = 0.f, s2 = 0.f;
  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
  if (REDUCE) {
    sc = bn_scale[c]; sh = bn_shift[c]; mu = bn_mean[c]; is = bn_invstd[c];
  }
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (okp[pd][ph]) {
        const float4 o = old[pd][ph];
        float4 v = make_float4(g[pd][ph][0] + o.x, g[pd][ph][1] + o.y, g[pd][ph][2] + o.z, g[pd][ph][3] + o.w);
        msl::st4(g_in + offp[pd][ph], v);
        if (REDUCE) v = msl::as_stored(g_in, v);  // the sums must see what the apply pass will read back
        if (REDUCE) {
          const float4 y4 = yv[pd][ph];
          const float ga[4] = {v.x, v.y, v.z, v.w}, ya[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gm = fmaf(ya[k], sc, sh) > 0.f ? ga[k] : 0.f;
            s1 += gm;
            s2 += gm * ((ya[k] - mu) * is);
          }
        }
      }
    }
  if (REDUCE) {
    const int NP = (gridDim.y / C) * gridDim.x, p = (nc / C) * gridDim.x + blockIdx.x;
    const double t1 = msl::block_sum((double)s1, scratch);
    __syncthreads();
    const double t2 = msl::block_sum((double)s2, scratch);
    if (threadIdx.x == 0) {
      partials[(size_t)c * NP + p] = t1;
      partials[((size_t)C + c) * NP + p] = t2;
    }
  }
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

// bwd-weight: one WAVE per work item (n, c, chunk of BW_CHUNK outputs): 27 accumulators per lane, a shuffle
// reduction at the end, no workgroup barrier at all (the tail layers have only 64 outputs per (n, c)).
constexpr int BW_CHUNK = 1024;
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            double* __restrict__ partials, int C, int D, int H, int W,
                                                            int OD, int OH, int OW, int stride, int chunks, int NP,
                                                            int total_items) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= total_items) return;
  const int chunk = item % chunks, nc = item / chunks;
  const int c = nc % C, n = nc / C;
  const int OS = OD * OH * OW;
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
  const float* xc = x + (size_t)nc * D * H * W;
  const float* dyc = dy + (size_t)nc * OS;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const int lo = chunk * BW_CHUNK, hi = min(OS, lo + BW_CHUNK);
  for (int o = lo + lane; o < hi; o += 64) {
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
  const int p = n * chunks + chunk;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const double t = msl::wave_sum((double)acc[k]);
    if (lane == 0) partials[((size_t)c * 27 + k) * NP + p] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Stride-2 depthwise backward WITHOUT materialising the input gradient (the 128^3 stem: 134 MB written, then read
// again by the stem's weight gradient).  One pass over (dL/dz, y_prev) produces everything that needs a full-batch
// reduction:
//   * the BatchNorm-backward sums of the producer layer: g = dwconv^T(dL/dz) is formed in registers only to be
//     folded into  sum gm  and  sum gm * xhat   (gm = g where relu(bn(y_prev)) > 0);
//   * this layer's weight gradient  dW[c][k] = sum dL/dz[o] * relu(bn(y_prev))[2o-1+k]  - the same (position, tap,
//     output) triples as g, so it costs one more FMA per triple.
// The consumer (msl_stem_conv_bwd_weight_fused) rebuilds g from dL/dz on the fly.  Thread = PPT patches of
// 2 x 2 x 4 input voxels of one (n, c); fixed-order reductions, fp64 partials.
template <int PPT, typename T = float>
__global__ __launch_bounds__(256) void dw_s2_bwd_reduce_bww_kernel(
    const T* __restrict__ dy, const float* __restrict__ w, int C, int D, int H, int W, int OD, int OH, int OW,
    const T* __restrict__ y_prev, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd, double* __restrict__ bn_partials,
    double* __restrict__ w_partials, float* __restrict__ w_taps_t, T* __restrict__ g_in, int accumulate) {
  __shared__ float red[4][32];
  const int nc = blockIdx.y, c = nc % C;
  // tap-major copy of the weights, (27, C): the consumer kernel then fetches 8 channels of one tap with one scalar load
  if (w_taps_t && blockIdx.x == 0 && nc < C && threadIdx.x < 27) w_taps_t[threadIdx.x * C + c] = w[c * 27 + threadIdx.x];
  const int W4 = W >> 2, H2 = (H + 1) >> 1, D2 = (D + 1) >> 1;
  const int total = D2 * H2 * W4;
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[c * 27 + k];
  const float sc = bn_scale[c], sh = bn_shift[c], mu = bn_mean[c], is = bn_invstd[c];
  const T* dyc = dy + (size_t)nc * OD * OH * OW;
  const T* yc = y_prev + (size_t)nc * D * H * W;
  float aw[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) aw[k] = 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
  for (int it = 0; it < PPT; ++it) {
    const int q0 = (blockIdx.x * PPT + it) * 256 + threadIdx.x;
    const bool live = q0 < total;
    const int q = live ? q0 : 0;
    const int cw = q % W4, b = (q / W4) % H2, a = q / (W4 * H2);
    const int iw0 = cw * 4, c2 = cw * 2;
    float dv[2][2][3];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ww = 0; ww < 3; ++ww) {
          const bool ok = live && a + dd < OD && b + hh < OH && c2 + ww < OW;
          const float v = msl::ld1(dyc + (ok ? ((size_t)(a + dd) * OH + b + hh) * OW + c2 + ww : 0));
          dv[dd][hh][ww] = ok ? v : 0.f;
        }
    bool okp[2][2];
    float4 yv[2][2];
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int id = 2 * a + pd, ih = 2 * b + ph;
        okp[pd][ph] = live && id < D && ih < H;
        yv[pd][ph] = msl::ld4(yc + (okp[pd][ph] ? ((size_t)id * H + ih) * W + iw0 : 0));
      }
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) msl::pin(yv[pd][ph]);
    // per axis: even index -> (offset 0, k = 1); odd index -> (offset 1, k = 0) and (offset 0, k = 2)
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const float ya[4] = {yv[pd][ph].x, yv[pd][ph].y, yv[pd][ph].z, yv[pd][ph].w};
        float av[4];  // relu(bn(y_prev)) of the 4 voxels; 0 outside the volume
#pragma unroll
        for (int k = 0; k < 4; ++k) av[k] = okp[pd][ph] ? msl::act(ya[k], sc, sh) : 0.f;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int td = 0; td < (pd ? 2 : 1); ++td) {
          const int dd = pd ? (td == 0 ? 1 : 0) : 0, kd = pd ? (td == 0 ? 0 : 2) : 1;
#pragma unroll
          for (int th = 0; th < (ph ? 2 : 1); ++th) {
            const int hh = ph ? (th == 0 ? 1 : 0) : 0, kh = ph ? (th == 0 ? 0 : 2) : 1;
            const int kb = kd * 9 + kh * 3;
            const float d0 = dv[dd][hh][0], d1 = dv[dd][hh][1], d2 = dv[dd][hh][2];
            o0 = fmaf(wk[kb + 1], d0, o0);
            o1 = fmaf(wk[kb], d1, fmaf(wk[kb + 2], d0, o1));
            o2 = fmaf(wk[kb + 1], d1, o2);
            o3 = fmaf(wk[kb], d2, fmaf(wk[kb + 2], d1, o3));
            aw[kb + 1] = fmaf(av[0], d0, fmaf(av[2], d1, aw[kb + 1]));
            aw[kb] = fmaf(av[1], d1, fmaf(av[3], d2, aw[kb]));
            aw[kb + 2] = fmaf(av[1], d0, fmaf(av[3], d1, aw[kb + 2]));
          }
        }
        float ga[4] = {o0, o1, o2, o3};
        if (g_in) {  // (block 2+: the input gradient is a tensor of its own; with `accumulate` the heads' share is added first)
          T* gp = g_in + (size_t)nc * D * H * W + (okp[pd][ph] ? ((size_t)(2 * a + pd) * H + 2 * b + ph) * W + iw0 : 0);
          if (accumulate) {
            const float4 old = msl::ld4(gp);
            ga[0] += old.x; ga[1] += old.y; ga[2] += old.z; ga[3] += old.w;
          }
          const float4 gs = msl::as_stored(gp, make_float4(ga[0], ga[1], ga[2], ga[3]));
          if (okp[pd][ph]) msl::st4(gp, gs);
          ga[0] = gs.x; ga[1] = gs.y; ga[2] = gs.z; ga[3] = gs.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gm = av[k] > 0.f ? ga[k] : 0.f;  // av > 0  <=>  inside the volume and relu(bn(y)) > 0
          s1 += gm;
          s2 += gm * ((ya[k] - mu) * is);
        }
      }
  }
  // block reduction of the 27 tap sums and the two BatchNorm sums: DPP wave sums, then the 4 waves in order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float t = msl::wave_sum(aw[k]);
    if (lane == 0) red[wv][k] = t;
  }
  {
    const float t1 = msl::wave_sum(s1), t2 = msl::wave_sum(s2);
    if (lane == 0) {
      red[wv][27] = t1;
      red[wv][28] = t2;
    }
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    const int k = threadIdx.x;
    const double t = ((double)red[0][k] + (double)red[1][k]) + ((double)red[2][k] + (double)red[3][k]);
    const int NP = (gridDim.y / C) * gridDim.x, p = (nc / C) * gridDim.x + blockIdx.x;
    if (k < 27) w_partials[((size_t)c * 27 + k) * NP + p] = t;
    else bn_partials[((size_t)(k - 27) * C + c) * NP + p] = t;
  }
}

__global__ __launch_bounds__(64) void dw_bwd_weight_finalize_kernel(const double* __restrict__ partials, int NP,
                                                                    float* __restrict__ dw, int count) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const double* ps = partials + (size_t)i * NP;
  double s = 0.0;
  int p = lane;
  for (; p + 7 * 64 < NP; p += 8 * 64) {  // 8 independent loads in flight; additions in index order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ps[p + u * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < NP; p += 64) s += ps[p];
  s = msl::wave_sum(s);
  if (lane == 0) dw[i] = (float)s;
}

}  // namespace

extern "C" int msl_dwconv_bwd_weight_tiled(const float* dy, const float* x, const float* in_scale, const float* in_shift,
                                           double* partials, int N, int C, int D, int H, int W, int stride, void* stream);
extern "C" int msl_dwconv_bwd_weight_tiled_num_partials(int N, int C, int D, int H, int W, int stride);
extern "C" int msl_dwconv_s1_bwd_data_resident(const float* dy, const float* w, float* g_in, int N, int C, int D, int H,
                                               int W, int accumulate, void* stream);

extern "C" {

// dy (N,C,OD,OH,OW) -> g_in (N,C,D,H,W); accumulate != 0 adds into g_in
int msl_dwconv_bwd_data(const float* dy, const float* w, float* g_in, int N, int C, int D, int H, int W,
                        int stride, int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    const int rc = msl_dwconv_s1_bwd_data_resident(dy, w, g_in, N, C, D, H, W, accumulate, stream);
    if (rc != MSL_ERR_UNSUPPORTED) return rc;
  }
  if (W % 4 == 0) {
    if (stride == 2) {
      dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256), N * C);
      MSL_LAUNCH((dw_bwd_data_s2_patch_kernel<false, float>), grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW,
                         accumulate, (const float*)nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    } else {
      dim3 grid(msl::cdiv(D * H * (W / 4), 256), N * C);
      MSL_LAUNCH(dw_bwd_data_kernel<1>, grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW, accumulate);
    }
  } else {
    dim3 grid(std::min(msl::cdiv(D * H * W, 256), 256), N * C);
    MSL_LAUNCH(dw_bwd_data_naive_kernel, grid, dim3(256), 0, st, dy, w, g_in, C, D, H, W, OD, OH, OW, stride, accumulate);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// Stride-2 bwd-data that also emits the BatchNorm-backward partials (sum gm, sum gm*xhat; fp64 [2][C][NP]) of the
// layer whose activation gradient it writes (y_prev = that layer's raw conv output, vec = its folded BatchNorm).
// -2 unless stride == 2 and W % 4 == 0.
int msl_dwconv_bwd_data_bnreduce_num_partials(int N, int C, int D, int H, int W) {
  if (W % 4 != 0) return -1;
  return N * msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256);
}

int msl_dwconv_bwd_data_bnreduce(const float* dy, const float* w, float* g_in, const float* y_prev,
                                 const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                 const float* bn_invstd, double* partials, int N, int C, int D, int H, int W, int stride,
                                 int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MSL_ERR_ARG;
  if (stride != 2 || W % 4 != 0) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256), N * C);
  MSL_LAUNCH((dw_bwd_data_s2_patch_kernel<true, float>), grid, dim3(256), 0, (hipStream_t)stream, dy, w, g_in, C, D, H, W,
                     OD, OH, OW, accumulate, y_prev, bn_scale, bn_shift, bn_mean, bn_invstd, partials);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ---- bf16 storage (the bf16 activation path): the stride-2 patch kernel on bf16 gradients ---------------------------------
// y_prev != NULL: also emit the BatchNorm-backward partials [2][C][NP] (NP = msl_dwconv_bwd_data_bnreduce_num_partials) of the
// layer whose activation gradient is written, computed from the ROUNDED gradient.  W % 4 == 0, else MSL_ERR_UNSUPPORTED.
int msl_dwconv_bwd_data_s2_patch_bf16(const void* dy, const float* w, void* g_in, const void* y_prev, const float* bn_vec,
                                      double* partials, int N, int C, int D, int H, int W, int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MSL_ERR_ARG;
  if (W % 4 != 0) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256), N * C);
  typedef msl::su16 u16;
  if (y_prev) {
    if (!bn_vec || !partials) return MSL_ERR_ARG;
    MSL_LAUNCH((dw_bwd_data_s2_patch_kernel<true, u16>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)dy, w,
                       (u16*)g_in, C, D, H, W, OD, OH, OW, accumulate, (const u16*)y_prev, bn_vec, bn_vec + C, bn_vec + 2 * C,
                       bn_vec + 3 * C, partials);
  } else {
    MSL_LAUNCH((dw_bwd_data_s2_patch_kernel<false, u16>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)dy, w,
                       (u16*)g_in, C, D, H, W, OD, OH, OW, accumulate, (const u16*)nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_dwconv_bwd_weight_num_partials(int N, int C, int D, int H, int W, int stride) {
  const int tiled = msl_dwconv_bwd_weight_tiled_num_partials(N, C, D, H, W, stride);
  if (tiled > 0) return tiled;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return N * msl::cdiv(OD * OH * OW, BW_CHUNK);
}

// dw (C,27) from dy and the raw input x (+ its affine); partials: fp64 [C*27][NP], NP = msl_dwconv_bwd_weight_num_partials.
// dw == NULL: leave the partials (deferred reduction, msl_grad_reduce_batch kind 1).
int msl_dwconv_bwd_weight(const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                          double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = msl_dwconv_bwd_weight_tiled(dy, x, in_scale, in_shift, partials, N, C, D, H, W, stride, stream);
    if (rc == MSL_OK) {
      if (!dw) return MSL_OK;
      const int NPt = msl_dwconv_bwd_weight_tiled_num_partials(N, C, D, H, W, stride);
      MSL_LAUNCH(dw_bwd_weight_finalize_kernel, dim3(C * 27), dim3(64), 0, st, partials, NPt, dw, C * 27);
      MSL_LAUNCH_CHECK();
      return MSL_OK;
    }
    if (rc != MSL_ERR_UNSUPPORTED) return rc;
  }
  const int chunks = msl::cdiv(OD * OH * OW, BW_CHUNK);
  const int total_items = N * C * chunks;
  MSL_LAUNCH(dw_bwd_weight_kernel, dim3(msl::cdiv(total_items, 4)), dim3(256), 0, st, dy, x, in_scale, in_shift,
                     partials, C, D, H, W, OD, OH, OW, stride, chunks, N * chunks, total_items);
  MSL_LAUNCH_CHECK();
  if (!dw) return MSL_OK;
  MSL_LAUNCH(dw_bwd_weight_finalize_kernel, dim3(C * 27), dim3(64), 0, st, partials, N * chunks, dw, C * 27);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

constexpr int S2F_PPT = 4;

// Number of fp64 partials per channel (BatchNorm sums) and per (channel, tap) (weight gradient) written by
// msl_dwconv_s2_bwd_bnreduce_bww; -1 if the shape is not supported.
int msl_dwconv_s2_bwd_bnreduce_bww_num_partials(int N, int C, int D, int H, int W) {
  if (W % 4 != 0) return -1;
  return N * msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256 * S2F_PPT);
}

// dy = dL/dz (N,C,OD,OH,OW) of a stride-2 depthwise layer, y_prev (N,C,D,H,W) = raw output of the producer layer with
// its BatchNorm vectors.  Writes bn_partials [2][C][NP] (sum gm, sum gm*xhat of the producer's BatchNorm backward)
// and w_partials [C*27][NP] (this layer's weight gradient); the input gradient itself is NOT written.
// w_taps_t (may be NULL): receives the (27, C) transpose of w for msl_stem_conv_bwd_weight_fused.
int msl_dwconv_s2_bwd_bnreduce_bww(const float* dy, const float* w, const float* y_prev, const float* bn_scale,
                                   const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                                   double* bn_partials, double* w_partials, float* w_taps_t, int N, int C, int D, int H,
                                   int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MSL_ERR_ARG;
  if (W % 4 != 0) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256 * S2F_PPT), N * C);
  MSL_LAUNCH((dw_s2_bwd_reduce_bww_kernel<S2F_PPT, float>), grid, dim3(256), 0, (hipStream_t)stream, dy, w, C, D, H, W, OD,
                     OH, OW, y_prev, bn_scale, bn_shift, bn_mean, bn_invstd, bn_partials, w_partials, w_taps_t, (float*)nullptr, 0);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// The same pass for a stride-2 depthwise layer whose input gradient IS a tensor (any block but the one behind a fused stem):
// additionally writes g_in (N,C,D,H,W) = dL/d relu(bn(y_prev)) (accumulate != 0: adds the heads' share already in g_in, and the
// BatchNorm sums are those of the total).  One launch instead of msl_dwconv_bwd_data_bnreduce + msl_dwconv_bwd_weight: the
// weight gradient pairs each activation with the very dL/dz values the transposed convolution multiplies with the taps.
int msl_dwconv_s2_bwd_data_bnreduce_bww(const float* dy, const float* w, float* g_in, const float* y_prev,
                                        const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                        const float* bn_invstd, double* bn_partials, double* w_partials, int N, int C, int D,
                                        int H, int W, int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || !g_in) return MSL_ERR_ARG;
  if (W % 4 != 0) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256 * S2F_PPT), N * C);
  MSL_LAUNCH((dw_s2_bwd_reduce_bww_kernel<S2F_PPT, float>), grid, dim3(256), 0, (hipStream_t)stream, dy, w, C, D, H, W, OD,
                     OH, OW, y_prev, bn_scale, bn_shift, bn_mean, bn_invstd, bn_partials, w_partials, (float*)nullptr, g_in,
                     accumulate);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// bf16 storage: dy (dL/dz) and y_prev are bf16 tensors, bn_vec = (>= 4, C) rows [scale, shift, mean, invstd]
int msl_dwconv_s2_bwd_bnreduce_bww_bf16(const void* dy, const float* w, const void* y_prev, const float* bn_vec,
                                        double* bn_partials, double* w_partials, float* w_taps_t, int N, int C, int D, int H,
                                        int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || !bn_vec) return MSL_ERR_ARG;
  if (W % 4 != 0) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  dim3 grid(msl::cdiv(((D + 1) / 2) * ((H + 1) / 2) * (W / 4), 256 * S2F_PPT), N * C);
  typedef msl::su16 u16;
  MSL_LAUNCH((dw_s2_bwd_reduce_bww_kernel<S2F_PPT, u16>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)dy, w, C,
                     D, H, W, OD, OH, OW, (const u16*)y_prev, bn_vec, bn_vec + C, bn_vec + 2 * C, bn_vec + 3 * C, bn_partials,
                     w_partials, w_taps_t, (u16*)nullptr, 0);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// dw[c][k] = sum_p w_partials[c*27+k][p]  (fixed order)
int msl_dwconv_bwd_weight_finalize(const double* w_partials, int num_partials, float* dw, int C, void* stream) {
  if (C <= 0 || num_partials <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(dw_bwd_weight_finalize_kernel, dim3(C * 27), dim3(64), 0, (hipStream_t)stream, w_partials,
                     num_partials, dw, C * 27);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
