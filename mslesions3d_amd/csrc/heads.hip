// SSD detection heads: per scale, Conv3d(C -> 12, k3, p1, bias) (box offsets, 2 anchors x 6) and
// Conv3d(C -> 2*n_classes, k3, p1, bias) (class scores), followed in the reference by
// permute(0,2,3,4,1).contiguous().view(N,-1,6|n_classes) and a concat over scales.
// Reference: PredictionConvolutions (lesions3d/ssd3d.py:113-169).
//
// Here both convolutions of a scale are ONE implicit GEMM (M = 12 + 2*n_classes output channels padded to
// 16, K = C*27, N = positions) on v_mfma_f32_16x16x4_f32, sharing a single read of the feature map, and
// the epilogue writes straight into the final (N, P, 6) / (N, P, n_classes) rows at the scale's prior
// offset: the permute, the .contiguous() copy and the concat never happen.
// The feature map is read from a zero-haloed copy (N, C, D+2, H+2, W+2), so the 27 taps are plain
// address offsets with no bounds tests.  Weights are pre-packed into MFMA fragment order (one coalesced
// 256-B load per fragment).  K is split over the 4 waves of a workgroup (and over workgroups for the
// small, channel-heavy scales) with a fixed-order reduction.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));  // bf16 activation path
typedef unsigned short hu16;
typedef hu16 hu16x8 __attribute__((ext_vector_type(8)));

// ---- weight packing ---------------------------------------------------------------------------
// forward fragments: Wf[((cg*27 + tap)*MT + mt)*64 + lane] = Wall[co = mt*16 + (lane&15)][ci = 4*cg + (lane>>4)][tap]
// bwd-data fragments: Wb[(((ct*(4*MT) + cog)*27) + tap)*64 + lane] = Wall[co = 4*cog + (lane>>4)][ci = 16*ct + (lane&15)][tap]
__global__ void head_pack_weights_kernel(const float* __restrict__ loc_w, const float* __restrict__ cl_w,
                                         float* __restrict__ Wf, float* __restrict__ Wb, int C, int co_total,
                                         int MT) {
  const int total = C / 4 * 27 * MT * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    {
      const int lane = i & 63;
      int r = i >> 6;
      const int mt = r % MT;
      r /= MT;
      const int tap = r % 27, cg = r / 27;
      const int co = mt * 16 + (lane & 15), ci = 4 * cg + (lane >> 4);
      float v = 0.f;
      if (co < 12) v = loc_w[((size_t)co * C + ci) * 27 + tap];
      else if (co < co_total) v = cl_w[((size_t)(co - 12) * C + ci) * 27 + tap];
      Wf[i] = v;
    }
    {
      const int lane = i & 63;
      int r = i >> 6;
      const int tap = r % 27;
      r /= 27;
      const int cog = r % (4 * MT), ct = r / (4 * MT);
      const int co = 4 * cog + (lane >> 4), ci = 16 * ct + (lane & 15);
      float v = 0.f;
      if (co < 12) v = loc_w[((size_t)co * C + ci) * 27 + tap];
      else if (co < co_total) v = cl_w[((size_t)(co - 12) * C + ci) * 27 + tap];
      Wb[i] = v;
    }
  }
}

// all scales of the model in one launch (blockIdx.y = scale): the packed copies are refreshed once per step
struct HeadPackBatch {
  const float* loc_w[4];
  const float* cl_w[4];
  float* Wf[4];
  float* Wb[4];
  int C[4];
};
__global__ void head_pack_weights_batch_kernel(HeadPackBatch b, int co_total, int MT) {
  const int k = blockIdx.y;
  const float *loc_w = b.loc_w[k], *cl_w = b.cl_w[k];
  float *Wf = b.Wf[k], *Wb = b.Wb[k];
  const int C = b.C[k];
  const int total = C / 4 * 27 * MT * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63;
    {
      int r = i >> 6;
      const int mt = r % MT;
      r /= MT;
      const int tap = r % 27, cg = r / 27;
      const int co = mt * 16 + (lane & 15), ci = 4 * cg + (lane >> 4);
      float v = 0.f;
      if (co < 12) v = loc_w[((size_t)co * C + ci) * 27 + tap];
      else if (co < co_total) v = cl_w[((size_t)(co - 12) * C + ci) * 27 + tap];
      Wf[i] = v;
    }
    {
      int r = i >> 6;
      const int tap = r % 27;
      r /= 27;
      const int cog = r % (4 * MT), ct = r / (4 * MT);
      const int co = 4 * cog + (lane >> 4), ci = 16 * ct + (lane & 15);
      float v = 0.f;
      if (co < 12) v = loc_w[((size_t)co * C + ci) * 27 + tap];
      else if (co < co_total) v = cl_w[((size_t)(co - 12) * C + ci) * 27 + tap];
      Wb[i] = v;
    }
  }
}

__device__ __forceinline__ void write_head_outputs(const f32x4 v, int mt, int q, int n, int P, int S,
                                                   const float* loc_b, const float* cl_b, float* locs,
                                                   float* scores, int Ptot, int prior_off, int ncls,
                                                   int co_total) {
  if (P >= S) return;
  const size_t lbase = ((size_t)n * Ptot + prior_off) * 6 + (size_t)P * 12;
  const size_t sbase = ((size_t)n * Ptot + prior_off) * ncls + (size_t)P * 2 * ncls;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = mt * 16 + 4 * q + r;
    if (co < 12) locs[lbase + co] = v[r] + loc_b[co];
    else if (co < co_total) scores[sbase + co - 12] = v[r] + cl_b[co - 12];
  }
}

// ---- forward ------------------------------------------------------------------------------------
// grid (ceil(S/32), N, KSG); block 256 = 4 waves, each wave a quarter of this block's channel range.
template <int MT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ a_pad,
                                                       const float* __restrict__ Wf,
                                                       const float* __restrict__ loc_b,
                                                       const float* __restrict__ cl_b, float* __restrict__ locs,
                                                       float* __restrict__ scores, float* __restrict__ slabs,
                                                       int C, int D, int H, int W, int Ptot, int prior_off,
                                                       int ncls, int co_total, int KSG) {
  __shared__ f32x4 red[4][2][MT][64];
  const int n = blockIdx.y, ksg = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int S = D * H * W, Hp = H + 2, Wp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  const int p0 = blockIdx.x * 32;
  const int cpw = C / (4 * KSG);          // channels per wave
  const int cbase = (ksg * 4 + wv) * cpw;  // first channel of this wave

  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int P = p0 + t * 16 + j;
    if (P >= S) P = S - 1;
    const int w = P % W, h = (P / W) % H, d = P / (W * H);
    boff[t] = (d * Hp + h) * Wp + w;
  }
  const float* ap = a_pad + ((size_t)n * C + cbase + q) * volp;
  const float* wp = Wf + (size_t)(cbase / 4) * 27 * MT * 64 + lane;

  f32x4 acc[2][MT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int cg = 0; cg < cpw / 4; ++cg) {
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = kd * 9 + kh * 3 + kw;
          const int toff = (kd * Hp + kh) * Wp + kw;
          const float b0 = ap[boff[0] + toff];
          const float b1 = ap[boff[1] + toff];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float a = wp[(size_t)(tap * MT + m) * 64];
            acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0][m], 0, 0, 0);
            acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[1][m], 0, 0, 0);
          }
        }
    ap += 4 * volp;
    wp += (size_t)27 * MT * 64;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) red[wv][t][m][lane] = acc[t][m];
  __syncthreads();
  // (tile, mt) pairs are finalised by waves 0 .. 2*MT-1; fixed summation order over the 4 k-split waves
  for (int pair = wv; pair < 2 * MT; pair += 4) {
    const int t = pair / MT, m = pair % MT;
    f32x4 s = red[0][t][m][lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) s += red[w2][t][m][lane];
    const int P = p0 + t * 16 + j;
    if (KSG == 1) {
      write_head_outputs(s, m, q, n, P, S, loc_b, cl_b, locs, scores, Ptot, prior_off, ncls, co_total);
    } else if (P < S) {
      // slabs[ksg][n][P][16*MT]
      float* dst = slabs + (((size_t)ksg * gridDim.y + n) * S + P) * (16 * MT) + m * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(dst) = s;
    }
  }
}

__global__ __launch_bounds__(256) void head_fwd_finalize_kernel(const float* __restrict__ slabs,
                                                                const float* __restrict__ loc_b,
                                                                const float* __restrict__ cl_b,
                                                                float* __restrict__ locs, float* __restrict__ scores,
                                                                int N, int S, int MT, int KSG, int Ptot,
                                                                int prior_off, int ncls, int co_total) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // over N*S*(4*MT) quads
  const int quads = 4 * MT;
  if (i >= N * S * quads) return;
  const int qd = i % quads, P = (i / quads) % S, n = i / (quads * S);
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* p = slabs + ((size_t)n * S + P) * (16 * MT) + qd * 4;
  const size_t kstride = (size_t)N * S * (16 * MT);
  int k = 0;
  for (; k + 8 <= KSG; k += 8) {  // 8 independent loads in flight; the additions keep the order k = 0, 1, 2, ...
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + u) * kstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < KSG; ++k) s += *reinterpret_cast<const f32x4*>(p + (size_t)k * kstride);
  write_head_outputs(s, qd / 4, qd % 4, n, P, S, loc_b, cl_b, locs, scores, Ptot, prior_off, ncls, co_total);
}

// ---- backward -----------------------------------------------------------------------------------
// dO_pad (N, 16*MT, D+2, H+2, W+2): channel-major, zero halo, zero for padded channels.
__global__ __launch_bounds__(256) void head_grad_pack_kernel(const float* __restrict__ dlocs,
                                                             const float* __restrict__ dscores,
                                                             float* __restrict__ dO_pad, int N, int D, int H, int W,
                                                             int CO, int Ptot, int prior_off, int ncls,
                                                             int co_total) {
  const int S = D * H * W, Hp = H + 2, Wp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  const int total = N * CO * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int P = i % S, co = (i / S) % CO, n = i / (S * CO);
    float v = 0.f;
    if (co < 12) v = dlocs[((size_t)n * Ptot + prior_off) * 6 + (size_t)P * 12 + co];
    else if (co < co_total) v = dscores[((size_t)n * Ptot + prior_off) * ncls + (size_t)P * 2 * ncls + co - 12];
    const int w = P % W, h = (P / W) % H, d = P / (W * H);
    dO_pad[((size_t)n * CO + co) * volp + ((size_t)(d + 1) * Hp + h + 1) * Wp + w + 1] = v;
  }
}

// the same for all (<= 4) scales in one launch: blockIdx.y = scale (three ~6 us launches, two of them on a side stream,
// become one: the rows of every scale are final as soon as the loss gradient is)
struct HeadGradPackBatch {
  float* dO[4];
  int D[4], H[4], W[4], prior_off[4];
};
__global__ __launch_bounds__(256) void head_grad_pack_batch_kernel(const float* __restrict__ dlocs,
                                                                   const float* __restrict__ dscores, HeadGradPackBatch b,
                                                                   int N, int CO, int Ptot, int ncls, int co_total) {
  const int k = blockIdx.y;
  const int D = b.D[k], H = b.H[k], W = b.W[k], prior_off = b.prior_off[k];
  float* __restrict__ dO_pad = b.dO[k];
  const int S = D * H * W, Hp = H + 2, Wp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  const int total = N * CO * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int P = i % S, co = (i / S) % CO, n = i / (S * CO);
    float v = 0.f;
    if (co < 12) v = dlocs[((size_t)n * Ptot + prior_off) * 6 + (size_t)P * 12 + co];
    else if (co < co_total) v = dscores[((size_t)n * Ptot + prior_off) * ncls + (size_t)P * 2 * ncls + co - 12];
    const int w = P % W, h = (P / W) % H, d = P / (W * H);
    dO_pad[((size_t)n * CO + co) * volp + ((size_t)(d + 1) * Hp + h + 1) * Wp + w + 1] = v;
  }
}

// g_a (N, C, S) = conv_transpose(dO, W).  grid (ceil(S/32), C/64, N); wave -> 16 input channels x 32 positions
template <int MT, bool BF16OUT = false>
__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ dO_pad,
                                                            const float* __restrict__ Wb, void* __restrict__ g_a_,
                                                            int C, int D, int H, int W) {
  const int n = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int S = D * H * W, Hp = H + 2, Wp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  const int p0 = blockIdx.x * 32;
  const int ct = blockIdx.y * 4 + wv;  // 16-channel tile
  if (ct * 16 >= C) return;
  constexpr int COG = 4 * MT;

  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int P = p0 + t * 16 + j;
    if (P >= S) P = S - 1;
    const int w = P % W, h = (P / W) % H, d = P / (W * H);
    boff[t] = (d * Hp + h) * Wp + w;
  }
  const float* dp = dO_pad + ((size_t)n * 16 * MT + q) * volp;
  const float* wp = Wb + (size_t)ct * COG * 27 * 64 + lane;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int cog = 0; cog < COG; ++cog) {
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = kd * 9 + kh * 3 + kw;
          const int toff = ((2 - kd) * Hp + (2 - kh)) * Wp + (2 - kw);
          const float a = wp[(size_t)tap * 64];
          const float b0 = dp[boff[0] + toff];
          const float b1 = dp[boff[1] + toff];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
        }
    dp += 4 * volp;
    wp += (size_t)27 * 64;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = ct * 16 + 4 * q + r;
    const size_t rb = ((size_t)n * C + ci) * S;
    const int P0 = p0 + j, P1 = p0 + 16 + j;
    if (BF16OUT) {
      unsigned short* dst = reinterpret_cast<unsigned short*>(g_a_) + rb;
      if (P0 < S) dst[P0] = msl::f2bf(acc0[r]);
      if (P1 < S) dst[P1] = msl::f2bf(acc1[r]);
    } else {
      float* dst = reinterpret_cast<float*>(g_a_) + rb;
      if (P0 < S) dst[P0] = acc0[r];
      if (P1 < S) dst[P1] = acc1[r];
    }
  }
}

// dW slabs.  grid (position blocks, C/16, 3); block = 4 waves = 4 consecutive position chunks of `steps` MFMA k-steps
// (4 positions each).  A workgroup owns the 9 taps of ONE kd plane of its 16-channel tile (16 co x 16 ci x 9 taps): with
// the taps split three ways the same number of workgroups needs a third of the position blocks, i.e. a third of the
// partial-sum slabs (every position block costs one slab of the whole weight tensor), and a wave carries 9 instead of
// 27 accumulator tiles.  The bias gradient sum_p dO[co][p] rides along as one more MFMA per step against a B operand
// of ones (workgroups with blockIdx.y == 0 && kd == 0 only), so dO is not read a second time by a kernel of its own.
// BF16CL: the feature map is the bf16 channels-last copy (N, D+2, H+2, W+2, C) of the bf16 activation path.
template <int MT, bool BF16CL = false>
__global__ __launch_bounds__(256) void head_bwd_weight_kernel(const float* __restrict__ dO_pad,
                                                              const void* __restrict__ a_pad_,
                                                              float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                              int N, int C, int D, int H, int W, int steps) {
  extern __shared__ __align__(16) float red[];  // [(9 + 1)*MT][256]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int S = D * H * W, Hp = H + 2, Wp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  const int total = N * S;
  const int ct = blockIdx.y, kd = blockIdx.z;
  const bool do_bias = ct == 0 && kd == 0;
  const int chunk = blockIdx.x * 4 + wv;
  const int pbeg = chunk * steps * 4;

  f32x4 acc[10 * MT];  // [0, 9*MT): taps (kh, kw) of plane kd; [9*MT, 10*MT): bias
#pragma unroll
  for (int i = 0; i < 10 * MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < steps; ++s) {
    int gp = pbeg + s * 4 + q;  // global position index over (n, d, h, w)
    const bool ok = gp < total;
    if (!ok) gp = total - 1;
    const int n = gp / S, P = gp % S;
    const int w = P % W, h = (P / W) % H, d = P / (W * H);
    const size_t off = ((size_t)d * Hp + h) * Wp + w;  // tap (0,0,0) corner in padded coordinates
    float a[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float v = dO_pad[((size_t)n * 16 * MT + m * 16 + j) * volp + off + (size_t)Hp * Wp + Wp + 1];
      a[m] = ok ? v : 0.f;
    }
    float b[9];
    if (BF16CL) {
      const unsigned short* bp = reinterpret_cast<const unsigned short*>(a_pad_) +
                                 ((size_t)n * volp + off + (size_t)kd * Hp * Wp) * C + ct * 16 + j;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) b[kh * 3 + kw] = msl::bf2f(bp[(size_t)(kh * Wp + kw) * C]);
    } else {
      const float* bp = reinterpret_cast<const float*>(a_pad_) + ((size_t)n * C + ct * 16 + j) * volp + off + (size_t)kd * Hp * Wp;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) b[kh * 3 + kw] = bp[kh * Wp + kw];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[t * MT + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[t * MT + m], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[9 * MT + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], 1.0f, acc[9 * MT + m], 0, 0, 0);
    }
  }
  // fixed-order reduction over the 4 waves
  for (int w2 = 3; w2 >= 1; --w2) {
    if (wv == w2) {
#pragma unroll
      for (int i = 0; i < 10 * MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = red + ((size_t)i * 4 + r) * 64 + lane;
          if (w2 == 3) *p = acc[i][r];
          else *p += acc[i][r];
        }
    }
    __syncthreads();
  }
  if (wv == 0) {
    // slabs[blockIdx.x][ct][tap][mt][co_local 16][ci_local 16]
    float* out = slabs + (((size_t)blockIdx.x * gridDim.y + ct) * 27 + kd * 9) * (MT * 256);
#pragma unroll
    for (int i = 0; i < 9 * MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)i * 256 + (4 * q + r) * 16 + j] = acc[i][r] + red[((size_t)i * 4 + r) * 64 + lane];
    if (do_bias && j == 0) {  // every column of the ones-product holds the row sums: take column 0
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bias_slabs[(size_t)blockIdx.x * 16 * MT + m * 16 + 4 * q + r] =
              acc[9 * MT + m][r] + red[((size_t)(9 * MT + m) * 4 + r) * 64 + lane];
    }
  }
}

// dW[co][ci][tap] = sum over slabs (fixed order); also splits into the loc / cls weight gradients
__global__ __launch_bounds__(256) void head_bwd_weight_reduce_kernel(const float* __restrict__ slabs,
                                                                     float* __restrict__ dloc_w,
                                                                     float* __restrict__ dcl_w, int C, int MT,
                                                                     int nslabs, int co_total) {
  const int total = co_total * C * 27;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int tap = i % 27, ci = (i / 27) % C, co = i / (27 * C);
  const int ct = ci / 16, cil = ci % 16, mt = co / 16, col = co % 16;
  const int CT = C / 16;
  float s = 0.f;
  const float* p = slabs + (size_t)ct * (27 * MT * 256) + (size_t)(tap * MT + mt) * 256 + col * 16 + cil;
  const size_t kstride = (size_t)CT * (27 * MT * 256);
  int k = 0;
  for (; k + 8 <= nslabs; k += 8) {  // 8 independent loads in flight; additions in slab order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * kstride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nslabs; ++k) s += p[(size_t)k * kstride];
  if (co < 12) dloc_w[((size_t)co * C + ci) * 27 + tap] = s;
  else dcl_w[((size_t)(co - 12) * C + ci) * 27 + tap] = s;
}

// dbias[co] = sum over the position blocks' partial row sums (fixed order); one wave
__global__ __launch_bounds__(64) void head_bias_reduce_kernel(const float* __restrict__ bias_slabs,
                                                              float* __restrict__ dloc_b, float* __restrict__ dcl_b,
                                                              int CO, int nslabs, int co_total) {
  const int co = threadIdx.x;
  if (co >= co_total) return;
  float s = 0.f;
  for (int k = 0; k < nslabs; ++k) s += bias_slabs[(size_t)k * CO + co];
  if (co < 12) dloc_b[co] = s;
  else dcl_b[co - 12] = s;
}

// =====================================================================================================================
// LDS-staged forms of the three head kernels for the cubic power-of-two feature maps of the training configurations
// (W = H in {4, 8, 16}, S % 64 == 0).  The register-fed kernels above issue one 4-byte gather per MFMA operand (1.5
// vector-memory instructions per 16x16x4 MFMA): with four waves per CU that saturates the CU's L1 path (64 B/clk) at a
// quarter of the MFMA rate.  Here a workgroup owns a block of 64 consecutive positions (four 16-position tiles, one per
// wave); the zero-haloed input slab of the block and the packed weight fragments are staged ONCE per workgroup through
// LDS with coalesced loads (prefetched into registers one chunk ahead), and every MFMA operand is a conflict-free
// ds_read_b32 (128 B/clk/CU, no tag lookup).  Blocks: W = 16 -> 4 rows of a plane, W = 8 -> a plane, W = 4 -> 4 planes.
// staging helpers: arrays by reference into force-inlined functions stay in registers (a lambda capturing them did not)
template <int NA, int NT = 256>
__device__ __forceinline__ void head_ld_f4(f32x4 (&av)[NA], const f32x4* __restrict__ wp, int tid, int total) {
#pragma unroll
  for (int i = 0; i < NA; ++i) av[i] = wp[min(tid + NT * i, total - 1)];
}
template <int NA, int NT = 256>
__device__ __forceinline__ void head_st_f4(const f32x4 (&av)[NA], float* __restrict__ dst, int tid, int total) {
#pragma unroll
  for (int i = 0; i < NA; ++i)
    if (tid + NT * i < total) reinterpret_cast<f32x4*>(dst)[tid + NT * i] = av[i];
}
template <int NS>
__device__ __forceinline__ void head_ld_f1(float (&sv)[NS], const float* __restrict__ ap, const int (&goff)[NS]) {
#pragma unroll
  for (int i = 0; i < NS; ++i) sv[i] = ap[goff[i] < 0 ? 0 : goff[i]];
}
template <int NS>
__device__ __forceinline__ void head_st_f1(const float (&sv)[NS], float* __restrict__ dst, const int (&goff)[NS],
                                           const int (&loff)[NS]) {
#pragma unroll
  for (int i = 0; i < NS; ++i)
    if (goff[i] >= 0) dst[loff[i]] = sv[i];
}

template <int W> struct HeadGeo;
template <> struct HeadGeo<16> { static constexpr int PD = 3, PH = 6, PW = 18, RB = 4, CS = 336, CSK = 130; };
template <> struct HeadGeo<8>  { static constexpr int PD = 3, PH = 10, PW = 10, RB = 8, CS = 304, CSK = 130; };
template <> struct HeadGeo<4>  { static constexpr int PD = 6, PH = 6, PW = 6, RB = 4, CS = 240, CSK = 162; };
// CS: channel stride of a slab in LDS = PD*PH*PW rounded up to 16 (mod 32): lanes (q, j) of a 32-lane group then hit 32
// different banks; CSK: the same for the per-kd slab of the weight-gradient kernel where the lane's j is the channel
// (stride 2 (mod 32): banks 2j + q).

template <int W>
__device__ __forceinline__ void head_block_origin(int b, int& d0, int& h0) {
  if (W == 16) { d0 = b >> 2; h0 = (b & 3) * 4; }
  else if (W == 8) { d0 = b; h0 = 0; }
  else { d0 = 4 * b; h0 = 0; }
}
// The forward kernel tiles ANY map whose extents are multiples of the block's (W = 16: 1 x 4 x 16, W = 8: 1 x 8 x 8,
// W = 4: 4 x 4 x 4 positions; e.g. the 24^3 / 12^3 maps of a 192^3 volume take the 8- and 4-wide blocks): block b ->
// origin (d0, h0, w0), blocks ordered w-fastest.  For the cubic maps above this is head_block_origin.
template <int W>
__device__ __forceinline__ void head_block_origin_any(int b, int Hr, int Wr, int& d0, int& h0, int& w0) {
  typedef HeadGeo<W> G;
  const int nbw = Wr / W, nbh = Hr / G::RB;
  const int bw = b % nbw, bh = (b / nbw) % nbh, bd = b / (nbw * nbh);
  d0 = bd * (G::PD - 2);
  h0 = bh * G::RB;
  w0 = bw * W;
}
// position pb (0..63) of a block -> offset of its tap (0,0,0) corner inside the slab
template <int W>
__device__ __forceinline__ int head_block_off(int pb) {
  typedef HeadGeo<W> G;
  const int ww = pb % W, hh = (pb / W) % G::RB, dd = pb / (W * G::RB);
  return (dd * G::PH + hh) * G::PW + ww;
}

// ---- forward.  grid (S/64, N, KSG); the block's channel range is walked in chunks of HEAD_FWD_CH channels.  The next
// chunk's loads are issued one chunk ahead: a chunk must hold about as much MFMA time (27 per k-group of 4 channels,
// 32 cycles each) as an L2 round trip takes under load (~1.5 us), or every chunk waits for its loads - 16 channels.
constexpr int HEAD_FWD_CH = 16;
// Eight waves per workgroup: two per 16-position tile, each contracting half of a chunk's channel groups, so that every
// SIMD holds two waves and one wave's LDS-operand waits are covered by the other's MFMAs.  (Ablation of the four-wave
// form at 16^3 x 4, C = 128: 30.7 us = 11.5 MFMA + 10 B-operand reads + 6.5 A-operand reads + staging, back to back: with
// one wave per SIMD nothing hides a wait.  More workgroups per CU did not help: they duplicate the staging.)
constexpr int HEAD_FWD_NT = 512;
template <int W, int MT, int CH = HEAD_FWD_CH>
__global__ __launch_bounds__(HEAD_FWD_NT) void head_fwd_lds_kernel(const float* __restrict__ a_pad, const float* __restrict__ Wf,
                                                                   const float* __restrict__ loc_b, const float* __restrict__ cl_b,
                                                                   float* __restrict__ locs, float* __restrict__ scores,
                                                                   float* __restrict__ slabs, int C, int D, int Hr, int Wr,
                                                                   int Ptot, int prior_off, int ncls, int co_total, int KSG) {
  typedef HeadGeo<W> G;  // W: the block's width; the map is D x Hr x Wr
  constexpr int NT = HEAD_FWD_NT;
  constexpr int SLAB = G::PD * G::PH * G::PW, NCG = CH / 4;
  constexpr int NS = (CH * SLAB + NT - 1) / NT;        // slab floats per thread and chunk
  constexpr int AF4 = NCG * 27 * MT * 64 / 4;          // weight fragments of a chunk, in float4
  constexpr int NA = (AF4 + NT - 1) / NT;
  static_assert(NCG % 2 == 0, "two waves share a tile's channel groups");
  extern __shared__ __align__(16) float head_fwd_dyn[];  // [CH * CS] input slab | [AF4 * 4] weight fragments
  float* slab = head_fwd_dyn;
  float* afr = head_fwd_dyn + CH * G::CS;
  const int n = blockIdx.y, ksg = blockIdx.z, b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, q = lane >> 4;
  const int tile = wv & 3, khalf = wv >> 2;
  const int S = D * Hr * Wr, Hp = Hr + 2, Wp = Wr + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Wp;
  int d0, h0, w0;
  head_block_origin_any<W>(b, Hr, Wr, d0, h0, w0);
  const int cpb = C / KSG, c_begin = ksg * cpb, nchunks = cpb / CH;
  // chunk-invariant staging offsets of this thread
  int goff[NS], loff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + NT * i;
    const int ch = e / SLAB, r = e % SLAB;
    const int pd = r / (G::PH * G::PW), ph = (r / G::PW) % G::PH, pw = r % G::PW;
    goff[i] = e < CH * SLAB ? (int)(ch * volp) + ((d0 + pd) * Hp + h0 + ph) * Wp + w0 + pw : -1;
    loff[i] = ch * G::CS + r;
  }
  const float* abase = a_pad + ((size_t)n * C + c_begin) * volp;
  const f32x4* wbase = reinterpret_cast<const f32x4*>(Wf + (size_t)(c_begin / 4) * 27 * MT * 64);
  float sv[NS];
  f32x4 av[NA];
  const int ob = head_block_off<W>(tile * 16 + j);
  // two accumulator chains per output tile (even / odd taps): a 16x16x4 fp32 MFMA issues every 32 cycles but a dependent
  // one only every 40
  f32x4 acc[MT], acc2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = acc2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  head_ld_f1(sv, abase, goff);
  head_ld_f4<NA, NT>(av, wbase, tid, AF4);
  for (int ck = 0; ck < nchunks; ++ck) {
    __syncthreads();  // the previous chunk has been consumed
    head_st_f1(sv, slab, goff, loff);
    head_st_f4<NA, NT>(av, afr, tid, AF4);
    __syncthreads();
    if (ck + 1 < nchunks) {  // in flight during this chunk's MFMAs
      head_ld_f1(sv, abase + (size_t)(ck + 1) * CH * volp, goff);
      head_ld_f4<NA, NT>(av, wbase + (size_t)(ck + 1) * AF4, tid, AF4);
    }
    // The operand reads of the wave's SECOND channel group are issued between the MFMAs of its first one (order pinned):
    // reads and matrix work of one wave overlap - the two waves of a SIMD run in lock-step between the chunk barriers, so
    // nothing else overlaps them
    static_assert(NCG / 2 == 2, "two channel groups per wave and chunk");
    {
      const int cga = khalf * 2, cgb = cga + 1;
      const float* spa = slab + (cga * 4 + q) * G::CS + ob;
      const float* spb = slab + (cgb * 4 + q) * G::CS + ob;
      const float* fpa = afr + (size_t)cga * 27 * MT * 64 + lane;
      const float* fpb = afr + (size_t)cgb * 27 * MT * 64 + lane;
      float bqa[27], aqa[27 * MT], bqb[27], aqb[27 * MT];
#pragma unroll
      for (int t = 0; t < 27; ++t) bqa[t] = spa[((t / 9) * G::PH + (t / 3) % 3) * G::PW + t % 3];
#pragma unroll
      for (int t = 0; t < 27 * MT; ++t) aqa[t] = fpa[t * 64];
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (tap & 1) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aqa[tap * MT + m], bqa[tap], acc2[m], 0, 0, 0);
          else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aqa[tap * MT + m], bqa[tap], acc[m], 0, 0, 0);
        }
        bqb[tap] = spb[((tap / 9) * G::PH + (tap / 3) % 3) * G::PW + tap % 3];
#pragma unroll
        for (int m = 0; m < MT; ++m) aqb[tap * MT + m] = fpb[(tap * MT + m) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (tap & 1) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aqb[tap * MT + m], bqb[tap], acc2[m], 0, 0, 0);
          else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aqb[tap * MT + m], bqb[tap], acc[m], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] += acc2[m];
  // the two waves of a tile meet in LDS (the slab region is free now): fixed order, first half + second half
  __syncthreads();
  f32x4* xch = reinterpret_cast<f32x4*>(slab);
  if (khalf == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) xch[(tile * MT + m) * 64 + lane] = acc[m];
  }
  __syncthreads();
  if (khalf == 1) return;
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] += xch[(tile * MT + m) * 64 + lane];
  const int pb = tile * 16 + j;
  const int P = ((d0 + pb / (W * G::RB)) * Hr + h0 + (pb / W) % G::RB) * Wr + w0 + pb % W;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (KSG == 1) {
      write_head_outputs(acc[m], m, q, n, P, S, loc_b, cl_b, locs, scores, Ptot, prior_off, ncls, co_total);
    } else {  // slabs[ksg][n][P][16*MT], folded by head_fwd_finalize_kernel
      float* dst = slabs + (((size_t)ksg * gridDim.y + n) * S + P) * (16 * MT) + m * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(dst) = acc[m];
    }
  }
}

// ---- bwd-data.  grid (S/64, ceil(C/16 / cts), N): the dO slab of the block is staged once, the weight fragments per
// 16-channel tile.
template <int W, int MT, bool BF16OUT = false>
__global__ __launch_bounds__(256) void head_bwd_data_lds_kernel(const float* __restrict__ dO_pad,
                                                                const float* __restrict__ Wb, void* __restrict__ g_a_,
                                                                int C, int D, int cts) {
  typedef HeadGeo<W> G;
  constexpr int SLAB = G::PD * G::PH * G::PW, CO = 16 * MT, COG = 4 * MT;
  constexpr int NS = (CO * SLAB + 255) / 256;
  constexpr int AF4 = COG * 27 * 64 / 4;
  constexpr int NA = (AF4 + 255) / 256;
  __shared__ __align__(16) float slab[CO * G::CS];
  __shared__ __align__(16) float afr[AF4 * 4];
  const int n = blockIdx.z, b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, q = lane >> 4;
  const int S = D * W * W, Hp = W + 2;
  const size_t volp = (size_t)(D + 2) * Hp * Hp;
  int d0, h0;
  head_block_origin<W>(b, d0, h0);
  const int ct_lo = blockIdx.y * cts, ct_hi = min(C / 16, ct_lo + cts);
  {
    const float* dp = dO_pad + (size_t)n * CO * volp;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int e = tid + 256 * i;
      if (e < CO * SLAB) {
        const int ch = e / SLAB, r = e % SLAB;
        const int pd = r / (G::PH * G::PW), ph = (r / G::PW) % G::PH, pw = r % G::PW;
        slab[ch * G::CS + r] = dp[(size_t)ch * volp + ((d0 + pd) * Hp + h0 + ph) * Hp + pw];
      }
    }
  }
  f32x4 av[NA];
  const f32x4* wb4 = reinterpret_cast<const f32x4*>(Wb);
  const int ob = head_block_off<W>(wv * 16 + j);
  const int P = b * 64 + wv * 16 + j;
  head_ld_f4(av, wb4 + (size_t)ct_lo * AF4, tid, AF4);
  for (int ct = ct_lo; ct < ct_hi; ++ct) {
    __syncthreads();  // the previous tile's fragments have been consumed (first pass: nothing)
    head_st_f4(av, afr, tid, AF4);
    __syncthreads();  // also publishes the dO slab on the first pass
    if (ct + 1 < ct_hi) head_ld_f4(av, wb4 + (size_t)(ct + 1) * AF4, tid, AF4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};  // two chains: see head_fwd_lds_kernel
    // software pipeline over the output-channel groups: the operand reads of group g + 1 ride between the MFMAs of group g
    // (order pinned; same accumulation order as the plain loop - see head_fwd_lds_kernel)
    static_assert(COG % 2 == 0, "groups are walked in pairs");
    auto rd = [&](int cog, int tap, float& b_, float& a_) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      b_ = slab[(cog * 4 + q) * G::CS + ob + ((2 - kd) * G::PH + (2 - kh)) * G::PW + (2 - kw)];
      a_ = afr[(size_t)cog * 27 * 64 + lane + tap * 64];
    };
    float bqa[27], aqa[27], bqb[27], aqb[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) rd(0, t, bqa[t], aqa[t]);
#pragma unroll 1
    for (int cog = 0; cog < COG; cog += 2) {
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        if (tap & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aqa[tap], bqa[tap], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aqa[tap], bqa[tap], acc, 0, 0, 0);
        rd(cog + 1, tap, bqb[tap], aqb[tap]);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool more = cog + 2 < COG;  // uniform
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        if (tap & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aqb[tap], bqb[tap], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aqb[tap], bqb[tap], acc, 0, 0, 0);
        if (more) rd(cog + 2, tap, bqa[tap], aqa[tap]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    acc += acc2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)n * C + ct * 16 + 4 * q + r) * S + P;
      if (BF16OUT) reinterpret_cast<unsigned short*>(g_a_)[o] = msl::f2bf(acc[r]);
      else reinterpret_cast<float*>(g_a_)[o] = acc[r];
    }
  }
}

// ---- bwd-weight.  grid (position splits, C/16, 3 tap planes): per 64-position block the 16-channel input slab of the
// workgroup's kd plane and the dO values of the block are staged; a wave contracts its 16 positions (4 MFMA k-steps)
// against the 9 taps of the plane.  HEAD_BWW_SB blocks form one stage (36 MFMAs per wave and block are 0.5 us, an L2
// round trip under load ~1.5 us: with one block per stage every stage waited for its loads).  Same slab / bias-slab
// layout as head_bwd_weight_kernel.
// BF16CL: the feature map is the bf16 channels-last copy (N, D+2, H+2, W+2, C) of the bf16 activation path; a thread then
// stages 8 channels of a voxel per 16-byte load.
constexpr int HEAD_BWW_SB = 4;
template <int W, int MT, bool BF16CL = false>
__global__ __launch_bounds__(256) void head_bww_lds_kernel(const float* __restrict__ dO_pad, const void* __restrict__ a_pad_,
                                                           float* __restrict__ slabs, float* __restrict__ bias_slabs, int N,
                                                           int C, int D, int blocks_per_wg) {
  const float* a_pad = reinterpret_cast<const float*>(a_pad_);
  const hu16* a_cl = reinterpret_cast<const hu16*>(a_pad_);
  typedef HeadGeo<W> G;
  constexpr int PDK = G::PD - 2, SLABK = PDK * G::PH * G::PW, CO = 16 * MT, DLD = 66, SB = HEAD_BWW_SB;
  constexpr int NS = (16 * SLABK + 255) / 256;
  constexpr int ND = (CO * 64 + 255) / 256;
  constexpr int FSZ = 16 * G::CSK, DSZ = CO * DLD;
  constexpr int STAGE = SB * (FSZ + DSZ), REDN = 10 * MT * 256;
  __shared__ __align__(16) float lds[STAGE > REDN ? STAGE : REDN];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, q = lane >> 4;
  const int S = D * W * W, Hp = W + 2, bpi = S / 64;
  const size_t volp = (size_t)(D + 2) * Hp * Hp;
  const int ct = blockIdx.y, kd = blockIdx.z;
  const bool do_bias = ct == 0 && kd == 0;
  const int total = N * bpi;
  const int blk_lo = blockIdx.x * blocks_per_wg, blk_hi = min(total, blk_lo + blocks_per_wg);
  // block-invariant parts of the staging offsets
  int gsp[NS], lsp[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / SLABK, r = e % SLABK;
    const int pd = r / (G::PH * G::PW), ph = (r / G::PW) % G::PH, pw = r % G::PW;
    gsp[i] = e < 16 * SLABK ? (int)(ch * volp) + ((kd + pd) * Hp + ph) * Hp + pw : -1;
    lsp[i] = ch * G::CSK + r;
  }
  int gdo[ND], ldo[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int e = tid + 256 * i;
    const int co = e >> 6, pb = e & 63;
    const int ww = pb % W, hh = (pb / W) % G::RB, dd = pb / (W * G::RB);
    gdo[i] = e < CO * 64 ? (int)(co * volp) + ((dd + 1) * Hp + hh + 1) * Hp + ww + 1 : -1;
    ldo[i] = co * DLD + pb;
  }
  // channels-last staging roles: element e -> (slab position r, channel half): one 16-byte load, 8 LDS stores
  constexpr int NSB = (2 * SLABK + 255) / 256;
  int gcl[NSB], lcl[NSB];
#pragma unroll
  for (int i = 0; i < NSB; ++i) {
    const int e = tid + 256 * i;
    const int r = e >> 1, hf = e & 1;
    const int pd = r / (G::PH * G::PW), ph = (r / G::PW) % G::PH, pw = r % G::PW;
    gcl[i] = e < 2 * SLABK ? (((kd + pd) * Hp + ph) * Hp + pw) * C + ct * 16 + hf * 8 : -1;
    lcl[i] = hf * 8 * G::CSK + r;
  }
  float sv[BF16CL ? 1 : SB][BF16CL ? 1 : NS], dv[SB][ND];
  hu16x8 cv[BF16CL ? SB : 1][BF16CL ? NSB : 1];
  // loads of the stage that starts at block b0 (blocks past the range are clamped: their products are skipped)
  auto load_stage = [&](int b0) {
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int blk = min(b0 + u, blk_hi - 1);
      const int n = blk / bpi, b = blk - n * bpi;
      int d0, h0;
      head_block_origin<W>(b, d0, h0);
      const int org = (d0 * Hp + h0) * Hp;
      if constexpr (BF16CL) {
        const hu16* ap = a_cl + ((size_t)n * volp + org) * C;
#pragma unroll
        for (int i = 0; i < NSB; ++i) cv[u][i] = *reinterpret_cast<const hu16x8*>(ap + (gcl[i] < 0 ? 0 : gcl[i]));
      } else {
        head_ld_f1(sv[u], a_pad + ((size_t)n * C + ct * 16) * volp + org, gsp);
      }
      head_ld_f1(dv[u], dO_pad + (size_t)n * CO * volp + org, gdo);
    }
  };
  f32x4 acc[10 * MT];  // [0, 9*MT): taps (kh, kw) of plane kd; [9*MT, 10*MT): bias
#pragma unroll
  for (int i = 0; i < 10 * MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (blk_lo < blk_hi) load_stage(blk_lo);
  for (int b0 = blk_lo; b0 < blk_hi; b0 += SB) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if constexpr (BF16CL) {
#pragma unroll
        for (int i = 0; i < NSB; ++i)
          if (gcl[i] >= 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) lds[u * (FSZ + DSZ) + lcl[i] + k * G::CSK] = msl::bf2f(cv[u][i][k]);
          }
      } else {
        head_st_f1(sv[u], lds + u * (FSZ + DSZ), gsp, lsp);
      }
      head_st_f1(dv[u], lds + u * (FSZ + DSZ) + FSZ, gdo, ldo);
    }
    __syncthreads();
    if (b0 + SB < blk_hi) load_stage(b0 + SB);
#pragma unroll 1
    for (int u = 0; u < SB; ++u) {
      if (b0 + u >= blk_hi) break;  // uniform
      const float* fslab = lds + u * (FSZ + DSZ);
      const float* dtile = fslab + FSZ;
      // every operand read of the block first, then its 36-40 MFMAs (see head_fwd_lds_kernel: the compiler otherwise waits
      // for each pair of operands right in front of their MFMAs)
      float a[4][MT], bq[4][9];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pb = wv * 16 + 4 * s + q;
        const float* sp = fslab + j * G::CSK + head_block_off<W>(pb);
#pragma unroll
        for (int m = 0; m < MT; ++m) a[s][m] = dtile[(m * 16 + j) * DLD + pb];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) bq[s][kh * 3 + kw] = sp[kh * G::PW + kw];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[t * MT + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m], bq[s][t], acc[t * MT + m], 0, 0, 0);
        if (do_bias) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[9 * MT + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m], 1.0f, acc[9 * MT + m], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // fixed-order reduction over the 4 waves
  float* red = lds;
  __syncthreads();
  for (int w2 = 3; w2 >= 1; --w2) {
    if (wv == w2) {
#pragma unroll
      for (int i = 0; i < 10 * MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = red + ((size_t)i * 4 + r) * 64 + lane;
          if (w2 == 3) *p = acc[i][r];
          else *p += acc[i][r];
        }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float* out = slabs + (((size_t)blockIdx.x * gridDim.y + ct) * 27 + kd * 9) * (MT * 256);
#pragma unroll
    for (int i = 0; i < 9 * MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)i * 256 + (4 * q + r) * 16 + j] = acc[i][r] + red[((size_t)i * 4 + r) * 64 + lane];
    if (do_bias && j == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bias_slabs[(size_t)blockIdx.x * 16 * MT + m * 16 + 4 * q + r] =
              acc[9 * MT + m][r] + red[((size_t)(9 * MT + m) * 4 + r) * 64 + lane];
    }
  }
}

// the LDS-staged kernels take cubic W = H in {4, 8, 16} feature maps (W = 4: whole groups of 4 planes) and 16 output
// channels (12 + 2*2: two classes)
inline int head_lds_w(int C, int D, int H, int W, int MT = 1) {
  constexpr int on = 1;
  if (!on || MT != 1 || H != W || C % 16 != 0 || (D * H * W) % 64 != 0) return 0;
  if (W == 16 || W == 8) return W;
  if (W == 4 && D % 4 == 0) return 4;
  return 0;
}

// the forward kernel's block width for N maps of D x H x W (0: the register-fed kernel).  Measured (tools/probes/
// head_bw_sweep.sh, us, register-fed / 4- / 8- / 16-wide blocks): 16^3 x 4, C 128: 34.9 / 31.2 / 30.2 / 28.3; 8^3 x 4, C 256:
// 14.0 / 13.3 / 12.8 / -; 12^3 x 2, C 512: 39.5 / 31.5 / - / -; 24^3 x 2, C 256: 101.4 / 98.0 / 104.6 / -; 32^3 x 1, C 128:
// 54.4 / 56.7 / 60.9 / 57.2; 48^3 x 2, C 128: 313 / 345 / 377 / 358 - staging through LDS pays while the launch has too few
// workgroups to hide the register-fed kernel's load latency by occupancy (under ~512 blocks of 64 positions).
inline int head_fwd_lds_bw(int N, int C, int D, int H, int W, int MT = 1) {
  if (MT != 1 || C % HEAD_FWD_CH != 0) return 0;
  if (const int cubic = head_lds_w(C, D, H, W, MT)) return cubic;  // the training maps (16^3 / 8^3 / 4^3)
  if ((long long)N * D * H * W / 64 >= 512) return 0;
  if (W % 4 == 0 && H % 4 == 0 && D % 4 == 0) return 4;
  if (W % 16 == 0 && H % 4 == 0) return 16;
  if (W % 8 == 0 && H % 8 == 0) return 8;
  return 0;
}

// =====================================================================================================================
// bf16 activation path (csrc/bf16.hip): the head convolutions on v_mfma_f32_16x16x32_bf16.  The feature map is the bf16
// CHANNELS-LAST zero-haloed copy (N, D+2, H+2, W+2, C) written by msl_bn_relu_materialize_bf16, so the B operand of a
// k-step - 8 consecutive channels of one voxel per lane - is ONE 16-byte load; k-step = (tap, 32-channel group).  The
// weights are packed per step as [tap][C/32][lane][8] bf16 (co = lane & 15, ci = 32*cg + 8*(lane >> 4) + j; rows >= 12 +
// 2*ncls are zero).  Outputs are the same fp32 (N, P, 6) / (N, P, ncls) rows as the fp32 kernels write.

__global__ void head_pack_weights_bf16_kernel(const float* __restrict__ loc_w, const float* __restrict__ cl_w,
                                              hu16* __restrict__ Wp, int C, int co_total) {
  const int total = 27 * (C / 32) * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63;
    const int cg = (i >> 6) % (C / 32), tap = (i >> 6) / (C / 32);
    const int co = lane & 15;
    hu16x8 o;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int ci = cg * 32 + 8 * (lane >> 4) + jj;
      float v = 0.f;
      if (co < 12) v = loc_w[((size_t)co * C + ci) * 27 + tap];
      else if (co < co_total) v = cl_w[((size_t)(co - 12) * C + ci) * 27 + tap];
      o[jj] = msl::f2bf(v);
    }
    *reinterpret_cast<hu16x8*>(Wp + (size_t)i * 8) = o;
  }
}

// grid (ceil(S / 64), N); a wave owns one 16-position tile and walks the 27 * C/32 k-steps four at a time: the 8
// operand loads of a group (16 bytes per lane each) leave together, so a wave has ~8 KB in flight instead of one
// dependent round trip per MFMA (the first version: 90 us per scale, latency-bound).
__global__ __launch_bounds__(256) void head_fwd_bf16_kernel(const hu16* __restrict__ a_cl, const hu16* __restrict__ Wp,
                                                            const float* __restrict__ loc_b, const float* __restrict__ cl_b,
                                                            float* __restrict__ locs, float* __restrict__ scores, int C,
                                                            int D, int H, int W, int Ptot, int prior_off, int ncls,
                                                            int co_total) {
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int S = D * H * W, Hp = H + 2, Wp2 = W + 2, CG = C / 32;
  const int p0 = (blockIdx.x * 4 + wv) * 16;
  int P = p0 + j;
  if (P >= S) P = S - 1;
  const int w = P % W, h = (P / W) % H, d = P / (W * H);
  const hu16* ab = a_cl + ((((size_t)n * (D + 2) + d) * Hp + h) * Wp2 + w) * C + 8 * g;  // tap (0,0,0), this lane's channels
  const hu16* wp = Wp + (size_t)lane * 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains (even / odd k-steps)
  const int nsteps = 27 * CG;  // CG is a multiple of 4 for C in {128, 256, 512}; other C: the tail loop
  int step = 0;
  int tap = 0, cg = 0;
  size_t toff = 0;
  auto advance = [&]() {  // (tap, cg) -> next k-step; toff = element offset of the tap
    if (++cg == CG) {
      cg = 0;
      ++tap;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      toff = (((size_t)kd * Hp + kh) * Wp2 + kw) * C;
    }
  };
  for (; step + 4 <= nsteps; step += 4) {
    hbf16x8 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const hbf16x8*>(wp + ((size_t)tap * CG + cg) * 512);
      b[u] = *reinterpret_cast<const hbf16x8*>(ab + toff + cg * 32);
      advance();
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], b[3], acc1, 0, 0, 0);
  }
  for (; step < nsteps; ++step) {
    const hbf16x8 a = *reinterpret_cast<const hbf16x8*>(wp + ((size_t)tap * CG + cg) * 512);
    const hbf16x8 b = *reinterpret_cast<const hbf16x8*>(ab + toff + cg * 32);
    advance();
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
  }
  acc0 += acc1;
  write_head_outputs(acc0, 0, g, n, p0 + j, S, loc_b, cl_b, locs, scores, Ptot, prior_off, ncls, co_total);
}

inline int head_mt(int ncls) { return (12 + 2 * ncls + 15) / 16; }

inline int head_ksg(int N, int C, int S) {
  int blocks = msl::cdiv(S, 32) * N;
  int ksg = 1;
  while (blocks * ksg < 256 && C / (4 * ksg * 2) >= 4 && (C % (4 * ksg * 2 * 4)) == 0) ksg *= 2;
  return ksg;
}

struct HwPlan {
  int lds_w;            // 0: register-fed kernel (steps / nblocks); else the LDS-staged kernel (bpw blocks per workgroup)
  int steps, nblocks, bpw;
};
// Position splits of the weight-gradient kernels: tiles = (C/16) x 3 tap planes; about 256 workgroups in total.  Every
// position split costs one partial slab of the whole weight tensor.
inline HwPlan head_bw_plan(int N, int C, int D, int H, int W, int MT) {
  const int S = D * H * W, tiles = (C / 16) * 3;
  HwPlan p;
  p.lds_w = head_lds_w(C, D, H, W, MT);
  constexpr int target = 256;
  const int want_blocks = std::max(1, target / tiles);
  if (p.lds_w) {
    const int total = N * (S / 64);
    p.bpw = msl::cdiv(total, std::min(want_blocks, total));
    p.nblocks = msl::cdiv(total, p.bpw);
    p.steps = 0;
    return p;
  }
  const int total_steps = msl::cdiv(N * S, 4);
  p.steps = std::max(8, msl::cdiv(total_steps, want_blocks * 4));
  p.nblocks = msl::cdiv(total_steps, p.steps * 4);
  p.bpw = 0;
  return p;
}
// K split of the forward kernel over workgroups (partial slabs + head_fwd_finalize_kernel)
inline int head_fwd_ksg(int N, int C, int D, int H, int W, int MT) {
  const int S = D * H * W;
  if (head_fwd_lds_bw(N, C, D, H, W, MT)) {
    // about one workgroup per CU.  More (512, 768 for the forward / bwd-data / weight-gradient kernels: a second wave per SIMD to cover the
    // loop's LDS-read waits) was measured 0.5-1.5 % SLOWER on the step (tools/probes/r02_headwgs.sh): the extra staging and
    // partial slabs cost what the latency hiding gains
    constexpr int target = 256;
    const int blocks = N * (S / 64);
    int ksg = 1;
    while (blocks * ksg < target && C / (ksg * 2) >= HEAD_FWD_CH && (C / (ksg * 2)) % HEAD_FWD_CH == 0) ksg *= 2;
    return ksg;
  }
  return head_ksg(N, C, S);
}

}  // namespace

extern "C" {

size_t msl_head_packed_weight_elems(int C, int ncls) { return (size_t)C / 4 * 27 * head_mt(ncls) * 64; }

int msl_head_pack_weights(const float* loc_w, const float* cl_w, float* Wf, float* Wb, int C, int ncls,
                          void* stream) {
  if (C % 16 != 0 || ncls < 1 || head_mt(ncls) > 2) return MSL_ERR_ARG;
  const int MT = head_mt(ncls);
  const int total = C / 4 * 27 * MT * 64;
  MSL_LAUNCH(head_pack_weights_kernel, dim3(std::min(msl::cdiv(total, 256), 1024)), dim3(256), 0,
                     (hipStream_t)stream, loc_w, cl_w, Wf, Wb, C, 12 + 2 * ncls, MT);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// the same for up to 4 scales in ONE launch; the arrays (n entries each) live on the HOST
int msl_head_pack_weights_batch(const float* const* loc_w, const float* const* cl_w, float* const* Wf, float* const* Wb,
                                const int* C, int n, int ncls, void* stream) {
  if (n < 1 || n > 4 || ncls < 1 || head_mt(ncls) > 2) return MSL_ERR_ARG;
  HeadPackBatch b;
  int cmax = 0;
  for (int k = 0; k < n; ++k) {
    if (C[k] % 16 != 0) return MSL_ERR_ARG;
    b.loc_w[k] = loc_w[k]; b.cl_w[k] = cl_w[k]; b.Wf[k] = Wf[k]; b.Wb[k] = Wb[k]; b.C[k] = C[k];
    cmax = std::max(cmax, C[k]);
  }
  const int MT = head_mt(ncls);
  const int total = cmax / 4 * 27 * MT * 64;
  MSL_LAUNCH(head_pack_weights_batch_kernel, dim3(std::min(msl::cdiv(total, 256), 512), n), dim3(256), 0,
                     (hipStream_t)stream, b, 12 + 2 * ncls, MT);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ---- bf16 activation path -------------------------------------------------------------------------------------------
size_t msl_head_packed_weight_bf16_elems(int C) { return (size_t)27 * (C / 32) * 64 * 8; }

int msl_head_pack_weights_bf16(const float* loc_w, const float* cl_w, void* Wp, int C, int ncls, void* stream) {
  if (C % 32 != 0 || ncls < 1 || 12 + 2 * ncls > 16) return MSL_ERR_UNSUPPORTED;
  const int total = 27 * (C / 32) * 64;
  MSL_LAUNCH(head_pack_weights_bf16_kernel, dim3(std::min(msl::cdiv(total, 256), 512)), dim3(256), 0,
                     (hipStream_t)stream, loc_w, cl_w, (hu16*)Wp, C, 12 + 2 * ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// a_cl (N,D+2,H+2,W+2,C) bf16 channels-last -> rows [prior_off, prior_off + 2*D*H*W) of locs / scores (fp32)
int msl_head_conv_fwd_bf16(const void* a_cl, const void* Wp, const float* loc_b, const float* cl_b, float* locs,
                           float* scores, int N, int C, int D, int H, int W, int Ptot, int prior_off, int ncls,
                           void* stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || ncls < 1) return MSL_ERR_ARG;
  if (C % 32 != 0 || 12 + 2 * ncls > 16) return MSL_ERR_UNSUPPORTED;
  const int S = D * H * W;
  MSL_LAUNCH(head_fwd_bf16_kernel, dim3(msl::cdiv(S, 64), N), dim3(256), 0, (hipStream_t)stream, (const hu16*)a_cl,
                     (const hu16*)Wp, loc_b, cl_b, locs, scores, C, D, H, W, Ptot, prior_off, ncls, 12 + 2 * ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

size_t msl_head_fwd_workspace_bytes(int N, int C, int D, int H, int W, int ncls) {
  const int S = D * H * W, ksg = head_fwd_ksg(N, C, D, H, W, head_mt(ncls));
  return ksg > 1 ? (size_t)ksg * N * S * 16 * head_mt(ncls) * sizeof(float) : 0;
}

// a_pad (N,C,D+2,H+2,W+2) -> rows [prior_off, prior_off + 2*D*H*W) of locs (N,Ptot,6) / scores (N,Ptot,ncls)
int msl_head_conv_fwd(const float* a_pad, const float* Wf, const float* loc_b, const float* cl_b, float* locs,
                      float* scores, float* workspace, int N, int C, int D, int H, int W, int Ptot,
                      int prior_off, int ncls, void* stream) {
  if (N <= 0 || C % 16 != 0 || D <= 0 || H <= 0 || W <= 0 || ncls < 1 || head_mt(ncls) > 2) return MSL_ERR_ARG;
  const int S = D * H * W, MT = head_mt(ncls), co_total = 12 + 2 * ncls;
  const int ksg = head_fwd_ksg(N, C, D, H, W, MT);
  hipStream_t st = (hipStream_t)stream;
  const int lw = head_fwd_lds_bw(N, C, D, H, W, MT);
  if (lw) {
    dim3 grid(S / 64, N, ksg);
#define MSL_HF(W_, MT_)                                                                                              \
  do {                                                                                                               \
    const size_t smem = ((size_t)HEAD_FWD_CH * HeadGeo<W_>::CS + (size_t)(HEAD_FWD_CH / 4) * 27 * MT_ * 64) * sizeof(float); \
    MSL_LAUNCH((head_fwd_lds_kernel<W_, MT_>), grid, dim3(HEAD_FWD_NT), smem, st, a_pad, Wf, loc_b, cl_b, locs,  \
                       scores, workspace, C, D, H, W, Ptot, prior_off, ncls, co_total, ksg);                          \
  } while (0)
    // (chunks of 32 channels - half the barriers, twice the MFMA time per prefetch - measured equal: 27.3 vs 28.0 us at 16^3)
    if (lw == 16) MSL_HF(16, 1); else if (lw == 8) MSL_HF(8, 1); else MSL_HF(4, 1);
#undef MSL_HF
  } else {
    dim3 grid(msl::cdiv(S, 32), N, ksg);
    if (MT == 1)
      MSL_LAUNCH(head_fwd_kernel<1>, grid, dim3(256), 0, st, a_pad, Wf, loc_b, cl_b, locs, scores, workspace, C, D, H, W, Ptot, prior_off, ncls, co_total, ksg);
    else
      MSL_LAUNCH(head_fwd_kernel<2>, grid, dim3(256), 0, st, a_pad, Wf, loc_b, cl_b, locs, scores, workspace, C, D, H, W, Ptot, prior_off, ncls, co_total, ksg);
  }
  MSL_LAUNCH_CHECK();
  if (ksg > 1) {
    const int total = N * S * 4 * MT;
    MSL_LAUNCH(head_fwd_finalize_kernel, dim3(msl::cdiv(total, 256)), dim3(256), 0, st, workspace, loc_b,
                       cl_b, locs, scores, N, S, MT, ksg, Ptot, prior_off, ncls, co_total);
    MSL_LAUNCH_CHECK();
  }
  return MSL_OK;
}

// (dlocs, dscores) rows of this scale -> dO_pad (N, 16*MT, D+2, H+2, W+2); halo must already be zero
int msl_head_grad_pack(const float* dlocs, const float* dscores, float* dO_pad, int N, int D, int H, int W,
                       int Ptot, int prior_off, int ncls, void* stream) {
  const int MT = head_mt(ncls), CO = 16 * MT;
  const int total = N * CO * D * H * W;
  MSL_LAUNCH(head_grad_pack_kernel, dim3(std::min(msl::cdiv(total, 256), 2048)), dim3(256), 0,
                     (hipStream_t)stream, dlocs, dscores, dO_pad, N, D, H, W, CO, Ptot, prior_off, ncls,
                     12 + 2 * ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// msl_head_grad_pack for all (n <= 4) scales in one launch; dO_pad / D / H / W / prior_off: host arrays of n entries
int msl_head_grad_pack_batch(const float* dlocs, const float* dscores, float* const* dO_pad, const int* D, const int* H,
                             const int* W, const int* prior_off, int n, int N, int Ptot, int ncls, void* stream) {
  if (n < 1 || n > 4 || N <= 0 || head_mt(ncls) > 2) return MSL_ERR_ARG;
  const int MT = head_mt(ncls), CO = 16 * MT;
  HeadGradPackBatch b;
  int smax = 0;
  for (int k = 0; k < 4; ++k) {
    const int j = k < n ? k : 0;
    if (!dO_pad[j] || D[j] <= 0 || H[j] <= 0 || W[j] <= 0) return MSL_ERR_ARG;
    b.dO[k] = dO_pad[j]; b.D[k] = D[j]; b.H[k] = H[j]; b.W[k] = W[j]; b.prior_off[k] = prior_off[j];
    smax = std::max(smax, D[j] * H[j] * W[j]);
  }
  MSL_LAUNCH(head_grad_pack_batch_kernel, dim3(std::min(msl::cdiv(N * CO * smax, 256), 1024), n), dim3(256), 0,
                     (hipStream_t)stream, dlocs, dscores, b, N, CO, Ptot, ncls, 12 + 2 * ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

static int head_bwd_data_impl(const float* dO_pad, const float* Wb, void* g_a, int N, int C, int D, int H, int W, int ncls,
                              bool bf16_out, void* stream) {
  if (C % 16 != 0) return MSL_ERR_ARG;
  const int S = D * H * W, MT = head_mt(ncls);
  hipStream_t st = (hipStream_t)stream;
  const int lw = head_lds_w(C, D, H, W, MT);
  if (lw) {
    constexpr int target = 256;
    const int blocks = N * (S / 64), ctiles = C / 16;
    const int gy = std::max(1, std::min(ctiles, target / blocks));
    const int cts = msl::cdiv(ctiles, gy);
    dim3 grid(S / 64, msl::cdiv(ctiles, cts), N);
#define MSL_HB(W_, B_) \
  MSL_LAUNCH((head_bwd_data_lds_kernel<W_, 1, B_>), grid, dim3(256), 0, st, dO_pad, Wb, g_a, C, D, cts)
    if (bf16_out) {
      if (lw == 16) MSL_HB(16, true); else if (lw == 8) MSL_HB(8, true); else MSL_HB(4, true);
    } else {
      if (lw == 16) MSL_HB(16, false); else if (lw == 8) MSL_HB(8, false); else MSL_HB(4, false);
    }
#undef MSL_HB
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(msl::cdiv(S, 32), msl::cdiv(C, 64), N);
  if (bf16_out) {
    if (MT == 1) MSL_LAUNCH((head_bwd_data_kernel<1, true>), grid, dim3(256), 0, st, dO_pad, Wb, g_a, C, D, H, W);
    else MSL_LAUNCH((head_bwd_data_kernel<2, true>), grid, dim3(256), 0, st, dO_pad, Wb, g_a, C, D, H, W);
  } else {
    if (MT == 1) MSL_LAUNCH((head_bwd_data_kernel<1, false>), grid, dim3(256), 0, st, dO_pad, Wb, g_a, C, D, H, W);
    else MSL_LAUNCH((head_bwd_data_kernel<2, false>), grid, dim3(256), 0, st, dO_pad, Wb, g_a, C, D, H, W);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_head_conv_bwd_data(const float* dO_pad, const float* Wb, float* g_a, int N, int C, int D, int H, int W,
                           int ncls, void* stream) {
  return head_bwd_data_impl(dO_pad, Wb, g_a, N, C, D, H, W, ncls, false, stream);
}

// bf16 activation path: the same with a bf16 gradient tensor g_a (N,C,D,H,W)
int msl_head_conv_bwd_data_bf16(const float* dO_pad, const float* Wb, void* g_a_bf16, int N, int C, int D, int H, int W,
                                int ncls, void* stream) {
  return head_bwd_data_impl(dO_pad, Wb, g_a_bf16, N, C, D, H, W, ncls, true, stream);
}

// workspace = [nblocks][C/16][27*MT][16][16] weight slabs followed by [nblocks][16*MT] bias slabs
size_t msl_head_bwd_weight_workspace_bytes(int N, int C, int D, int H, int W, int ncls) {
  HwPlan p = head_bw_plan(N, C, D, H, W, head_mt(ncls));
  return ((size_t)p.nblocks * (C / 16) * 27 * head_mt(ncls) * 256 + (size_t)p.nblocks * 16 * head_mt(ncls)) * sizeof(float);
}

// number of slabs msl_head_conv_bwd_weight leaves in its workspace
int msl_head_conv_bwd_weight_nslabs(int N, int C, int D, int H, int W, int ncls) {
  if (N <= 0 || C % 16 != 0 || D <= 0 || H <= 0 || W <= 0 || ncls < 1) return MSL_ERR_ARG;
  return head_bw_plan(N, C, D, H, W, head_mt(ncls)).nblocks;
}

// dloc_w == NULL (then dcl_w / dloc_b / dcl_b are ignored): leave the partial slabs in `workspace` (deferred reduction:
// msl_grad_reduce_batch kind 3 for the weights, kind 0 on the bias slabs at float offset nslabs*(C/16)*27*MT*256 with
// stride 16*MT: rows 0-11 -> dloc_b, rows 12.. -> dcl_b).
static int head_bww_impl(const float* dO_pad, const void* a_pad, bool bf16_cl, float* dloc_w, float* dcl_w, float* dloc_b,
                         float* dcl_b, float* workspace, int N, int C, int D, int H, int W, int ncls, void* stream);

int msl_head_conv_bwd_weight(const float* dO_pad, const float* a_pad, float* dloc_w, float* dcl_w,
                             float* dloc_b, float* dcl_b, float* workspace, int N, int C, int D, int H, int W,
                             int ncls, void* stream) {
  return head_bww_impl(dO_pad, a_pad, false, dloc_w, dcl_w, dloc_b, dcl_b, workspace, N, C, D, H, W, ncls, stream);
}

// bf16 activation path: the feature map is the bf16 channels-last copy (N,D+2,H+2,W+2,C) of msl_bn_relu_materialize_bf16.
int msl_head_conv_bwd_weight_bf16(const float* dO_pad, const void* a_cl, float* dloc_w, float* dcl_w, float* dloc_b,
                                  float* dcl_b, float* workspace, int N, int C, int D, int H, int W, int ncls,
                                  void* stream) {
  return head_bww_impl(dO_pad, a_cl, true, dloc_w, dcl_w, dloc_b, dcl_b, workspace, N, C, D, H, W, ncls, stream);
}

static int head_bww_impl(const float* dO_pad, const void* a_pad, bool bf16_cl, float* dloc_w, float* dcl_w, float* dloc_b,
                         float* dcl_b, float* workspace, int N, int C, int D, int H, int W, int ncls, void* stream) {
  if (C % 16 != 0) return MSL_ERR_ARG;
  const int MT = head_mt(ncls), co_total = 12 + 2 * ncls;
  HwPlan p = head_bw_plan(N, C, D, H, W, MT);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(p.nblocks, C / 16, 3);
  float* bias_slabs = workspace + (size_t)p.nblocks * (C / 16) * 27 * MT * 256;
  if (p.lds_w) {
#define MSL_HW(W_, B_)                                                                                               \
  MSL_LAUNCH((head_bww_lds_kernel<W_, 1, B_>), grid, dim3(256), 0, st, dO_pad, a_pad, workspace, bias_slabs, N, \
                     C, D, p.bpw)
    if (bf16_cl) {
      if (p.lds_w == 16) MSL_HW(16, true); else if (p.lds_w == 8) MSL_HW(8, true); else MSL_HW(4, true);
    } else {
      if (p.lds_w == 16) MSL_HW(16, false); else if (p.lds_w == 8) MSL_HW(8, false); else MSL_HW(4, false);
    }
#undef MSL_HW
  } else {
    const size_t lds = (size_t)10 * MT * 256 * sizeof(float);
#define MSL_HG(M_, B_)                                                                                                  \
  MSL_LAUNCH((head_bwd_weight_kernel<M_, B_>), grid, dim3(256), lds, st, dO_pad, a_pad, workspace, bias_slabs, N, C, \
                     D, H, W, p.steps)
    if (bf16_cl) {
      if (MT == 1) MSL_HG(1, true); else MSL_HG(2, true);
    } else {
      if (MT == 1) MSL_HG(1, false); else MSL_HG(2, false);
    }
#undef MSL_HG
  }
  MSL_LAUNCH_CHECK();
  if (!dloc_w) return MSL_OK;
  const int total = co_total * C * 27;
  MSL_LAUNCH(head_bwd_weight_reduce_kernel, dim3(msl::cdiv(total, 256)), dim3(256), 0, st, workspace,
                     dloc_w, dcl_w, C, MT, p.nblocks, co_total);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(head_bias_reduce_kernel, dim3(1), dim3(64), 0, st, bias_slabs, dloc_b, dcl_b, 16 * MT, p.nblocks, co_total);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
