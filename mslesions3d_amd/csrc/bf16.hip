// bf16 activation path (BASELINE configs[2] / [3]; a build-side extension: the reference is fp32 everywhere).
// Activations live in HBM as bf16 (NCDHW, raw conv outputs as in the fp32 path: the consumer applies
// relu(fma(x, scale, shift)) while loading); weights, BatchNorm vectors / statistics and every accumulator stay fp32.
//   depthwise 3x3x3 (mobilenet.py:38)  : LDS-tiled, any shape, stride 1 / 2                  - HBM-bound byte mover
//   pointwise 1x1x1 (mobilenet.py:40)  : GEMM on v_mfma_f32_32x32x16_bf16, fp32 accumulate   - the MFMA user
//   BatchNorm+ReLU materialisation of the head feature maps: bf16 CHANNELS-LAST zero-haloed copy, so that the bf16
//   head kernel (heads.hip) reads 8 consecutive channels of a voxel with one 16-byte load.
// The statistics of a layer are taken from the fp32 accumulators BEFORE the rounding to bf16.
#include "common.hpp"
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));

// ---- depthwise ---------------------------------------------------------------------------------------------------
// Workgroup = (n, c) x an output tile of TD x TH x TW voxels.  The activated input tile (with halo, zero outside the
// volume - the padding value AFTER the activation) is staged in LDS as fp32; a thread produces TD outputs of one
// (h, w) column.  256 threads = TH * TW.
constexpr int DW_TD = 2, DW_TH = 8, DW_TW = 32;

template <int STRIDE>
__global__ __launch_bounds__(256) void dw_fwd_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, const float* __restrict__ w,
                                                          u16* __restrict__ y, double* __restrict__ partials, int C, int D,
                                                          int H, int W, int OD, int OH, int OW, int tiles_h, int tiles_w,
                                                          int NP) {
  constexpr int ID = (DW_TD - 1) * STRIDE + 3, IH = (DW_TH - 1) * STRIDE + 3, IW = (DW_TW - 1) * STRIDE + 3;
  __shared__ float tile[ID * IH * IW];
  __shared__ double scratch[8];
  const int nc = blockIdx.y, c = nc % C, n = nc / C;
  const int t = blockIdx.x;
  const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, td = t / (tiles_w * tiles_h);
  const int od0 = td * DW_TD, oh0 = th * DW_TH, ow0 = tw * DW_TW;
  const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
  const u16* xc = x + (size_t)nc * D * H * W;
  if ((W & 7) == 0) {
    // 16-byte loads: iw0 = (a multiple of 8) - 1, so the tile's columns iw0 .. iw0 + IW - 1 lie inside the NG aligned
    // 8-column groups that start at cbase = iw0 - 7; a row of the volume is a whole number of groups
    constexpr int NG = (IW + 7 + 7) / 8;
    const int cbase = iw0 - 7;
    for (int e = threadIdx.x; e < ID * IH * NG; e += 256) {
      const int gi = e % NG, lh = (e / NG) % IH, ld = e / (NG * IH);
      const int id = id0 + ld, ih = ih0 + lh, col = cbase + gi * 8;
      const bool ok = id >= 0 && id < D && ih >= 0 && ih < H && col >= 0 && col < W;
      const u16x8 v = *reinterpret_cast<const u16x8*>(xc + (ok ? ((size_t)id * H + ih) * W + col : 0));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int lw = col + i - iw0;
        if (lw >= 0 && lw < IW) {
          float f = msl::bf2f(v[i]);
          if (affine) f = msl::act(f, sc, sh);
          tile[(ld * IH + lh) * IW + lw] = ok ? f : 0.f;
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < ID * IH * IW; e += 256) {
      const int lw = e % IW, lh = (e / IW) % IH, ld = e / (IW * IH);
      const int id = id0 + ld, ih = ih0 + lh, iw = iw0 + lw;
      const bool ok = id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
      float v = msl::bf2f(xc[ok ? ((size_t)id * H + ih) * W + iw : 0]);
      if (affine) v = msl::act(v, sc, sh);
      tile[e] = ok ? v : 0.f;
    }
  }
  float wt[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wt[k] = w[c * 27 + k];
  __syncthreads();
  const int lw = threadIdx.x % DW_TW, lh = threadIdx.x / DW_TW;
  const int ow = ow0 + lw, oh = oh0 + lh;
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int d = 0; d < DW_TD; ++d) {
    const int od = od0 + d;
    float acc = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc = fmaf(wt[kd * 9 + kh * 3 + kw], tile[((d * STRIDE + kd) * IH + lh * STRIDE + kh) * IW + lw * STRIDE + kw], acc);
    if (od < OD && oh < OH && ow < OW) {
      y[((size_t)nc * OD + od) * OH * OW + (size_t)oh * OW + ow] = msl::f2bf(acc);
      s += (double)acc;
      q += (double)acc * (double)acc;
    }
  }
  if (partials) {
    const double ts = msl::block_sum(s, scratch);
    __syncthreads();
    const double tq = msl::block_sum(q, scratch);
    if (threadIdx.x == 0) {
      const int p = n * gridDim.x + blockIdx.x;
      partials[(size_t)c * NP + p] = ts;
      partials[((size_t)C + c) * NP + p] = tq;
    }
  }
}

// ---- pointwise ---------------------------------------------------------------------------------------------------
// Y_n[M x S] = W[M x K] . act(Z_n)[K x S].  Workgroup tile 64 (rows) x 64 (columns); K in chunks of 32; both operands go
// through LDS as bf16 with k contiguous (the MFMA wants 8 consecutive k per lane; NCDHW has k strided by S, so the
// activation chunk is transposed on its way into LDS).  4 waves = 2 x 2 tiles of 32 x 32.
constexpr int PB_BM = 64, PB_BN = 64, PB_BK = 32, PB_LD = PB_BK + 8;  // 80-byte rows: 16-byte aligned fragments

template <bool STATS>
__global__ __launch_bounds__(256) void pw_fwd_bf16_kernel(const u16* __restrict__ Z, const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, const float* __restrict__ Wt,
                                                          u16* __restrict__ Y, double* __restrict__ partials, int M, int K,
                                                          int S) {
  __shared__ __align__(16) u16 Ws[PB_BM * PB_LD];
  __shared__ __align__(16) u16 Xs[PB_BN * PB_LD];
  const int n = blockIdx.z, m0 = blockIdx.y * PB_BM, s0 = blockIdx.x * PB_BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
  const u16* Zn = Z + (size_t)n * K * S;
  const bool affine = in_scale != nullptr;
  const bool vec_ok = (S & 7) == 0;
  // staging roles
  const int wrow = tid >> 2, wk8 = (tid & 3) * 8;   // weight chunk: row, 8 consecutive k
  const int xk = tid >> 3, xc8 = (tid & 7) * 8;     // activation chunk: k row, 8 consecutive columns
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += PB_BK) {
    float wv8[8];
    {
      const int row = m0 + wrow < M ? m0 + wrow : 0;
      const float4 a = *reinterpret_cast<const float4*>(Wt + (size_t)row * K + k0 + wk8);
      const float4 b = *reinterpret_cast<const float4*>(Wt + (size_t)row * K + k0 + wk8 + 4);
      wv8[0] = a.x; wv8[1] = a.y; wv8[2] = a.z; wv8[3] = a.w; wv8[4] = b.x; wv8[5] = b.y; wv8[6] = b.z; wv8[7] = b.w;
    }
    float xv8[8];
    {
      const u16* zp = Zn + (size_t)(k0 + xk) * S;
      const int col = s0 + xc8;
      if (vec_ok) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(zp + (col < S ? col : 0));
#pragma unroll
        for (int i = 0; i < 8; ++i) xv8[i] = col < S ? msl::bf2f(v[i]) : 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv8[i] = col + i < S ? msl::bf2f(zp[col + i]) : 0.f;
      }
      if (affine) {
        const float sc = in_scale[k0 + xk], sh = in_shift[k0 + xk];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv8[i] = (s0 + xc8 + i < S) ? msl::act(xv8[i], sc, sh) : 0.f;
      }
    }
    __syncthreads();  // the previous chunk has been consumed
    {
      u16x8 p;
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = msl::f2bf(wv8[i]);
      *reinterpret_cast<u16x8*>(&Ws[wrow * PB_LD + wk8]) = p;
#pragma unroll
      for (int i = 0; i < 8; ++i) Xs[(xc8 + i) * PB_LD + xk] = msl::f2bf(xv8[i]);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < PB_BK / 16; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ws[(wm * 32 + r) * PB_LD + ks * 16 + 8 * h]);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Xs[(wn * 32 + r) * PB_LD + ks * 16 + 8 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  // D[row][col]: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  u16* Yn = Y + (size_t)n * M * S;
  const int col = s0 + wn * 32 + r;
  const int NP = gridDim.z * gridDim.x * 2, p = (n * gridDim.x + blockIdx.x) * 2 + wn;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    const float v = acc[i];
    if (row < M && col < S) Yn[(size_t)row * S + col] = msl::f2bf(v);
    if (STATS) {  // columns past S hold exact zeros
      const float sm = msl::half32_sum(v), q = msl::half32_sum(v * v);
      if (r == msl::HALF32_SUM_LANE && row < M && partials) {
        partials[(size_t)row * NP + p] = (double)sm;
        partials[((size_t)M + row) * NP + p] = (double)q;
      }
    }
  }
}

// ---- materialise -------------------------------------------------------------------------------------------------
// relu(bn(y)) of a head feature map: bf16 NCDHW raw -> bf16 channels-last zero-haloed copy (N, D+2, H+2, W+2, C) (the
// halo is zeroed once at allocation and never written) and, optionally, the plain fp32 NCDHW activation.
__global__ __launch_bounds__(256) void materialize_bf16_kernel(const u16* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ plain,
                                                               u16* __restrict__ pad_cl, int C, int D, int H, int W) {
  const int S = D * H * W;
  const int p = blockIdx.x * 256 + threadIdx.x, c8 = blockIdx.y * 8, n = blockIdx.z;
  if (p >= S) return;
  const int w = p % W, hh = (p / W) % H, d = p / (W * H);
  u16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = c8 + i;
    const float v = msl::act(msl::bf2f(y[((size_t)n * C + c) * S + p]), scale[c], shift[c]);
    if (plain) plain[((size_t)n * C + c) * S + p] = v;
    o[i] = msl::f2bf(v);
  }
  const size_t pos = (((size_t)n * (D + 2) + d + 1) * (H + 2) + hh + 1) * (W + 2) + w + 1;
  *reinterpret_cast<u16x8*>(pad_cl + pos * C + c8) = o;
}

}  // namespace

extern "C" {

int msl_dwconv_fwd_bf16_num_partials(int N, int C, int D, int H, int W, int stride) {
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return N * msl::cdiv(OD, DW_TD) * msl::cdiv(OH, DW_TH) * msl::cdiv(OW, DW_TW);
}

// x (N,C,D,H,W) bf16 raw + optional input affine -> y (N,C,OD,OH,OW) bf16 raw (+ fp64 stat partials [2][C][NP] or NULL)
int msl_dwconv_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int tiles_d = msl::cdiv(OD, DW_TD), tiles_h = msl::cdiv(OH, DW_TH), tiles_w = msl::cdiv(OW, DW_TW);
  const int NP = N * tiles_d * tiles_h * tiles_w;
  dim3 grid(tiles_d * tiles_h * tiles_w, N * C);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1)
    hipLaunchKernelGGL(dw_fwd_bf16_kernel<1>, grid, dim3(256), 0, st, (const u16*)x, in_scale, in_shift, w, (u16*)y, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP);
  else
    hipLaunchKernelGGL(dw_fwd_bf16_kernel<2>, grid, dim3(256), 0, st, (const u16*)x, in_scale, in_shift, w, (u16*)y, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_pwconv_fwd_bf16_num_partials(int N, int S) { return N * msl::cdiv(S, PB_BN) * 2; }

// z (N,Cin,S) bf16 raw + optional input affine -> y (N,Cout,S) bf16 raw (+ stat partials [2][Cout][NP] or NULL)
int msl_pwconv_fwd_bf16(const void* z, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int Cin, int Cout, int S, void* stream) {
  if (N <= 0 || S <= 0 || Cin % PB_BK != 0 || Cout <= 0) return MSL_ERR_ARG;
  dim3 grid(msl::cdiv(S, PB_BN), msl::cdiv(Cout, PB_BM), N);
  hipStream_t st = (hipStream_t)stream;
  if (partials)
    hipLaunchKernelGGL(pw_fwd_bf16_kernel<true>, grid, dim3(256), 0, st, (const u16*)z, in_scale, in_shift, w, (u16*)y,
                       partials, Cout, Cin, S);
  else
    hipLaunchKernelGGL(pw_fwd_bf16_kernel<false>, grid, dim3(256), 0, st, (const u16*)z, in_scale, in_shift, w, (u16*)y,
                       partials, Cout, Cin, S);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// y (N,C,D,H,W) bf16 raw + affine -> pad_cl (N,D+2,H+2,W+2,C) bf16 (halo pre-zeroed by the caller) [+ plain fp32 NCDHW]
int msl_bn_relu_materialize_bf16(const void* y, const float* scale, const float* shift, float* plain, void* pad_cl, int N,
                                 int C, int D, int H, int W, void* stream) {
  if (N <= 0 || C % 8 != 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift) return MSL_ERR_ARG;
  dim3 grid(msl::cdiv(D * H * W, 256), C / 8, N);
  hipLaunchKernelGGL(materialize_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)y, scale, shift, plain,
                     (u16*)pad_cl, C, D, H, W);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
