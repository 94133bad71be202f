// Stem: dense Conv3d(C_in -> 32, k3, stride (2,2,2) for cubes / (1,2,2) otherwise, pad 1, no bias)
// Reference: conv_bn (lesions3d/mobilenet.py:26-31) used as features[0] (lesions3d/ssd3d.py:60-61).
//
// HBM-bound layer (C_in = 1: 33.5 MB in, 134 MB out at 128^3 x 4).  One thread owns ONE output voxel
// and produces all 32 output channels from a single 27*C_in-value register patch, so the input is read
// once; every per-channel store is a 256-B row segment per wave (lanes run along W, the fastest axis).
// Weights are wave-uniform -> scalar loads feeding v_fma with an SGPR operand (no LDS, no VGPRs).
// BN statistics: per-workgroup (sum, sumsq) per channel via an LDS transpose, emitted as fp64 partials.
#include "common.hpp"
#include <algorithm>

namespace {

constexpr int STEM_COUT = 32;

template <int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w, float* __restrict__ y,
                                                       double* __restrict__ partials, int D, int H, int W,
                                                       int OD, int OH, int OW, int sd, int sh, int sw) {
  constexpr int K = CIN * 27;
  __shared__ float red[8][256];
  const int n = blockIdx.y;
  const int OS = OD * OH * OW;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const bool valid = o < OS;
  const int oo = valid ? o : 0;
  const int ow = oo % OW, oh = (oo / OW) % OH, od = oo / (OW * OH);

  float in[K];
  {
    const int id0 = od * sd - 1, ih0 = oh * sh - 1, iw0 = ow * sw - 1;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float* xc = x + ((size_t)n * CIN + ci) * D * H * W;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int id = id0 + kd;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = ih0 + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = iw0 + kw;
            const bool ok = valid && id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
            in[ci * 27 + kd * 9 + kh * 3 + kw] = ok ? xc[((size_t)id * H + ih) * W + iw] : 0.f;
          }
        }
      }
    }
  }

  const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;  // 8 groups of 32 threads
  double chs[4], chq[4];
  float* yo = y + (size_t)n * STEM_COUT * OS + oo;
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(w[(gp * 8 + c) * K + k], in[k], a);
      acc[c] = valid ? a : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (valid) yo[(size_t)(gp * 8 + c) * OS] = acc[c];
      red[c][threadIdx.x] = acc[c];
    }
    __syncthreads();
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = red[grp][i * 32 + lane32];
      s += v;
      q = fmaf(v, v, q);
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) {
      ds += __shfl_xor(ds, m, 64);
      dq += __shfl_xor(dq, m, 64);
    }
    chs[gp] = ds;
    chq[gp] = dq;
    __syncthreads();
  }
  if (partials && lane32 == 0) {
    const int NP = gridDim.x * gridDim.y;
    const int p = n * gridDim.x + blockIdx.x;
#pragma unroll
    for (int gp = 0; gp < 4; ++gp) {
      const int c = gp * 8 + grp;
      partials[(size_t)c * NP + p] = chs[gp];
      partials[((size_t)STEM_COUT + c) * NP + p] = chq[gp];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Stem bwd-weight: dW[co][k] = sum_{n,o} dy[n,co,o] * x[n,ci,o*s-1+tap], k = ci*27 + tap.
// A GEMM with M = 32 (co), N = 27*Cin (k, padded to 32*NT), K = all output positions (1M at 128^3 x 4), on
// v_mfma_f32_32x32x2_f32.  Each wave walks chunks of <= 64 consecutive outputs of one output row: the dy tile
// (32 x 64, coalesced along W) and the 9*Cin input rows it needs are staged in the wave's own LDS region;
// the im2col operand is gathered from those rows (row pitch == 3 mod 32 -> the 27 taps hit distinct banks).
// The read of dy (134 MB) dominates; accumulators stay in registers over all chunks of a wave.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SB_DY_LD = 65;     // dy tile pitch
constexpr int SB_ROW_LD = 131;   // input row pitch (2*64 + 1 = 129 needed; 131 % 32 == 3)

// APPLY: `dy` holds dL/d relu(bn(y)) (not yet through the BatchNorm backward); the kernel reads y as well and
// applies  dL/dy = scale * (gm - c1 - xhat * c2)  while staging the tile, so the 402 MB apply pass over the stem
// gradient never runs.  bnv = [scale, shift, mean, invstd, c1, c2] x 32 channels.
template <int CIN, bool APPLY>
__global__ __launch_bounds__(256) void stem_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ slabs, int N, int D, int H, int W,
                                                              int OD, int OH, int OW, int sd, int sh, int sw,
                                                              int chunks_per_row, int total_chunks, int iters,
                                                              const float* __restrict__ yraw,
                                                              const float* __restrict__ bnv) {
  constexpr int NT = (CIN * 27 + 31) / 32;
  constexpr int WAVE_LDS = 32 * SB_DY_LD + CIN * 9 * SB_ROW_LD;
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* dyt = lds + wv * WAVE_LDS;
  float* rows = dyt + 32 * SB_DY_LD;
  const int OS = OD * OH * OW;

  // this lane's im2col column(s): tap k = nt*32 + (lane & 31)
  int tapoff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int k = nt * 32 + (lane & 31);
    if (k < CIN * 27) {
      const int ci = k / 27, t = k % 27;
      tapoff[nt] = (ci * 9 + t / 3) * SB_ROW_LD + (t % 3);
    } else {
      tapoff[nt] = -1;
    }
  }
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x16){0};

  for (int it = 0; it < iters; ++it) {
    const int chunk = (blockIdx.x * iters + it) * 4 + wv;
    const bool live = chunk < total_chunks;
    const int cc = live ? chunk : 0;
    const int seg = cc % chunks_per_row;
    int r = cc / chunks_per_row;
    const int oh = r % OH;
    r /= OH;
    const int od = r % OD, n = r / OD;
    const int ow0 = seg * 64;
    const int npos = live ? min(64, OW - ow0) : 0;
    __syncthreads();  // previous chunk's LDS reads are done
    // Issue every global load of the chunk first (32 dy + 9*CIN*3 input values per lane in flight), then
    // store to LDS: a load->store->load chain would expose the full memory latency 59 times per chunk.
    // Every load is unconditional on a clamped (always valid) address and masked afterwards: a predicated load
    // becomes a branch per load, and the compiler then waits for each one before issuing the next.  The
    // sched_barrier keeps the machine scheduler from re-interleaving the loads with their uses to save VGPRs.
    float dreg[32], yreg[APPLY ? 32 : 1];
    const bool in = lane < npos;
    {
      const size_t off = (size_t)n * 32 * OS + ((size_t)od * OH + oh) * OW + ow0 + (in ? lane : 0);
      const float* src = dy + off;
#pragma unroll
      for (int co = 0; co < 32; ++co) dreg[co] = src[(size_t)co * OS];
      if (APPLY) {
        const float* ysrc = yraw + off;
#pragma unroll
        for (int co = 0; co < 32; ++co) yreg[co] = ysrc[(size_t)co * OS];
      }
    }
    float xreg[CIN * 9][3];
    bool xok[CIN * 9][3];
    {
      const int iw0 = ow0 * sw - 1;
      const int span = 64 * sw + 1;  // j in [0, span]
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          const int id = od * sd - 1 + rr / 3, ih = oh * sh - 1 + rr % 3;
          const bool rok = live && id >= 0 && id < D && ih >= 0 && ih < H;
          const float* src = x + (((size_t)n * CIN + ci) * D + (rok ? id : 0)) * H * W + (size_t)(rok ? ih : 0) * W;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int j = lane + 64 * t, iw = iw0 + j;
            xok[ci * 9 + rr][t] = rok && j <= span && iw >= 0 && iw < W;
            xreg[ci * 9 + rr][t] = src[xok[ci * 9 + rr][t] ? iw : 0];
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // everything above is in flight before anything below waits
    if (APPLY) {
#pragma unroll
      for (int co = 0; co < 32; ++co) {
        const float yv = yreg[co];
        const float sc = bnv[co], sf = bnv[32 + co], mu = bnv[64 + co], is = bnv[96 + co];
        const float gm = fmaf(yv, sc, sf) > 0.f ? dreg[co] : 0.f;
        dreg[co] = sc * (gm - bnv[128 + co] - ((yv - mu) * is) * bnv[160 + co]);
      }
    }
#pragma unroll
    for (int co = 0; co < 32; ++co) dreg[co] = in ? dreg[co] : 0.f;
#pragma unroll
    for (int r2 = 0; r2 < CIN * 9; ++r2)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        xreg[r2][t] = xok[r2][t] ? xreg[r2][t] : 0.f;
        if (t == 2) asm volatile("" : "+v"(xreg[r2][t]));  // pin it here: else the load is sunk into the conditional store
      }
#pragma unroll
    for (int co = 0; co < 32; ++co) dyt[co * SB_DY_LD + lane] = dreg[co];
#pragma unroll
    for (int r2 = 0; r2 < CIN * 9; ++r2)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int j = lane + 64 * t;
        if (j < SB_ROW_LD) rows[r2 * SB_ROW_LD + j] = xreg[r2][t];
      }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int pos = 2 * s + (lane >> 5);
      const float a = dyt[(lane & 31) * SB_DY_LD + pos];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float b = tapoff[nt] >= 0 ? rows[tapoff[nt] + pos * sw] : 0.f;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
      }
    }
  }
  // fixed-order reduction over the 4 waves through LDS, then one slab per block: slab[32][32*NT]
  __syncthreads();
  float* red = lds;
  for (int w2 = 3; w2 >= 1; --w2) {
    if (wv == w2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
          float* p = red + row * (32 * NT) + nt * 32 + (lane & 31);
          if (w2 == 3) *p = acc[nt][rg];
          else *p += acc[nt][rg];
        }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float* out = slabs + (size_t)blockIdx.x * 32 * 32 * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        const int idx = row * (32 * NT) + nt * 32 + (lane & 31);
        out[idx] = acc[nt][rg] + red[idx];
      }
  }
}

__global__ __launch_bounds__(256) void stem_bwd_weight_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                     int K, int NT, int nslabs) {
  __shared__ float lds[8 * 33];
  // reduce the padded [32][32*NT] slab image, then drop the padding columns
  const int count = 32 * 32 * NT;
  const float t = msl::reduce_slabs_256(slabs, (size_t)count, count, nslabs, lds);
  const int i = blockIdx.x * 32 + (threadIdx.x & 31);
  if ((threadIdx.x >> 5) == 0 && i < count) {
    const int co = i / (32 * NT), k = i % (32 * NT);
    if (k < K) dw[co * K + k] = t;
  }
}

constexpr int STEM_BW_BLOCKS = 512;

}  // namespace

extern "C" {

int msl_stem_conv_fwd_num_partials(int N, int OD, int OH, int OW) { return N * msl::cdiv(OD * OH * OW, 256); }

// x (N,Cin,D,H,W) -> y (N,32,OD,OH,OW) raw conv output + fp64 stat partials [2][32][NP].
int msl_stem_conv_fwd(const float* x, const float* w, float* y, double* partials, int N, int Cin, int D,
                      int H, int W, int sd, int sh, int sw, void* stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || sd < 1 || sd > 2 || sh < 1 || sh > 2 || sw < 1 || sw > 2)
    return MSL_ERR_ARG;
  const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
  dim3 grid(msl::cdiv(OD * OH * OW, 256), N);
  hipStream_t st = (hipStream_t)stream;
  switch (Cin) {
    case 1: hipLaunchKernelGGL(stem_fwd_kernel<1>, grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, OW, sd, sh, sw); break;
    case 2: hipLaunchKernelGGL(stem_fwd_kernel<2>, grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, OW, sd, sh, sw); break;
    case 3: hipLaunchKernelGGL(stem_fwd_kernel<3>, grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, OW, sd, sh, sw); break;
    case 4: hipLaunchKernelGGL(stem_fwd_kernel<4>, grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, OW, sd, sh, sw); break;
    default: return MSL_ERR_UNSUPPORTED;
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

size_t msl_stem_conv_bwd_weight_workspace_bytes(int Cin) {
  const int NT = (Cin * 27 + 31) / 32;
  return (size_t)STEM_BW_BLOCKS * 32 * 32 * NT * sizeof(float);
}

// dw (32,Cin,3,3,3) = correlation of dy (N,32,OD,OH,OW) with x (N,Cin,D,H,W)
static int stem_bww_impl(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                         int sd, int sh, int sw, const float* yraw, const float* bnv, void* stream);

int msl_stem_conv_bwd_weight(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D,
                             int H, int W, int sd, int sh, int sw, void* stream) {
  return stem_bww_impl(dy, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, nullptr, nullptr, stream);
}

// g = dL/d relu(bn(y)): BatchNorm backward applied on load.  bn_vec = (6, 32) fp32 rows [scale, shift, mean, invstd,
// c1 = dbeta/n, c2 = dgamma/n] as produced by msl_bn_finalize + msl_bn_bwd_finalize.
int msl_stem_conv_bwd_weight_bnapply(const float* g, const float* yraw, const float* bn_vec, const float* x, float* dw,
                                     float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                     void* stream) {
  if (!yraw || !bn_vec) return MSL_ERR_ARG;
  return stem_bww_impl(g, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, yraw, bn_vec, stream);
}

static int stem_bww_impl(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                         int sd, int sh, int sw, const float* yraw, const float* bnv, void* stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || sd < 1 || sd > 2 || sh < 1 || sh > 2 || sw < 1 || sw > 2)
    return MSL_ERR_ARG;
  const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
  const int chunks_per_row = msl::cdiv(OW, 64);
  const int total_chunks = N * OD * OH * chunks_per_row;
  const int nblocks = std::min(STEM_BW_BLOCKS, msl::cdiv(total_chunks, 4));
  const int iters = msl::cdiv(total_chunks, nblocks * 4);
  hipStream_t st = (hipStream_t)stream;
  const int NT = (Cin * 27 + 31) / 32;
  const size_t lds = (size_t)4 * (32 * SB_DY_LD + Cin * 9 * SB_ROW_LD) * sizeof(float);
#define MSL_STEM_BW1(CI, AP)                                                                                         \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_weight_kernel<CI, AP>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e_ != hipSuccess) return (int)e_;                                                                          \
    }                                                                                                                \
    hipLaunchKernelGGL((stem_bwd_weight_kernel<CI, AP>), dim3(nblocks), dim3(256), lds, st, dy, x, workspace, N, D,  \
                       H, W, OD, OH, OW, sd, sh, sw, chunks_per_row, total_chunks, iters, yraw, bnv);                \
  } while (0)
#define MSL_STEM_BW(CI)            \
  do {                             \
    if (yraw) MSL_STEM_BW1(CI, true); \
    else MSL_STEM_BW1(CI, false);  \
  } while (0)
  switch (Cin) {
    case 1: MSL_STEM_BW(1); break;
    case 2: MSL_STEM_BW(2); break;
    case 3: MSL_STEM_BW(3); break;
    case 4: MSL_STEM_BW(4); break;
    default: return MSL_ERR_UNSUPPORTED;
  }
#undef MSL_STEM_BW
#undef MSL_STEM_BW1
  MSL_LAUNCH_CHECK();
  const int K = Cin * 27;
  hipLaunchKernelGGL(stem_bwd_weight_reduce_kernel, dim3(msl::cdiv(32 * 32 * NT, 32)), dim3(256), 0, st, workspace, dw, K,
                     NT, nblocks);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
