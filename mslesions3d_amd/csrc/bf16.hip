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

// dwconv.hip: the register-marching wave kernels on bf16 storage (square power-of-two planes)
extern "C" int msl_dwconv_wave_num_partials(int N, int C, int D, int H, int W, int stride);
extern "C" int msl_dwconv_fwd_eval_rows_ok(int N, int C, int D, int H, int W, int stride);
extern "C" int msl_dwconv_fwd_wave_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                                        double* partials, int N, int C, int D, int H, int W, int stride, int flip,
                                        int accumulate, void* stream);
extern "C" int msl_dwconv_fwd_small_eval_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                                              int N, int C, int D, int H, int W, int stride, void* stream);
extern "C" int msl_dwconv_bwd_data_s2_patch_bf16(const void* dy, const float* w, void* g_in, const void* y_prev, const float* bn_vec,
                                                 double* partials, int N, int C, int D, int H, int W, int accumulate,
                                                 void* stream);
extern "C" int msl_dwconv_bwd_weight_wave_bf16(const void* dz, const void* x, const float* in_scale, const float* in_shift,
                                               double* partials, int N, int C, int D, int H, int W, int stride, void* stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- depthwise ---------------------------------------------------------------------------------------------------
// Workgroup = (n, c) x an output tile of TD x TH x TW voxels.  The activated input tile (with halo, zero outside the
// volume - the padding value AFTER the activation) is staged in LDS as fp32; a thread produces TD outputs of one
// (h, w) column.  256 threads = TH * TW.
constexpr int DW_TD = 2, DW_TH = 8, DW_TW = 32;

// flip != 0: taps reversed (w[26 - k]) - the stride-1 bwd-data pass; accumulate != 0: y += result (the feature-map
// gradient already holds the heads' share).
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_fwd_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, const float* __restrict__ w,
                                                          u16* __restrict__ y, double* __restrict__ partials, int C, int D,
                                                          int H, int W, int OD, int OH, int OW, int tiles_h, int tiles_w,
                                                          int NP, int flip, int accumulate) {
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
  for (int k = 0; k < 27; ++k) wt[k] = w[c * 27 + (flip ? 26 - k : k)];
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
      u16* yo = y + ((size_t)nc * OD + od) * OH * OW + (size_t)oh * OW + ow;
      if (accumulate) acc += msl::bf2f(*yo);
      *yo = msl::f2bf(acc);
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
constexpr int PB_BM = 64, PB_BN = 64, PB_BK = 32, PB_LD = PB_BK + 8, PB_MAXK = 1024;  // 80-byte rows: 16-byte aligned fragments

// TRANS_W: weight element (m, k) = Wt[k * M + m] (bwd-data: M = Cin, K = Cout, the same weights read transposed).
template <bool STATS, bool TRANS_W = false>
__global__ __launch_bounds__(256) void pw_fwd_bf16_kernel(const u16* __restrict__ Z, const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, const float* __restrict__ Wt,
                                                          u16* __restrict__ Y, double* __restrict__ partials, int M, int K,
                                                          int S, msl::BnFold fold) {
  __shared__ __align__(16) u16 Ws[PB_BM * PB_LD];
  __shared__ __align__(16) u16 Xs[PB_BN * PB_LD];
  __shared__ float fsc[PB_MAXK], fsh[PB_MAXK];  // input BatchNorm folded from its producer's partials (fold.partials)
  if (fold.partials) {  // K <= PB_MAXK, fold.NP <= 64 (host-checked): one thread per channel, the finalize kernel's order
    for (int c = threadIdx.x; c < K; c += 256) {
      float a, b, m_, i_;
      double v_;
      msl::bn_fold_serial(fold, c, a, b, m_, i_, v_);
      fsc[c] = a;
      fsh[c] = b;
    }
    __syncthreads();
  }
  const int n = blockIdx.z, m0 = blockIdx.y * PB_BM, s0 = blockIdx.x * PB_BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
  const u16* Zn = Z + (size_t)n * K * S;
  const bool folded = fold.partials != nullptr;
  const bool affine = in_scale != nullptr || folded;
  const bool vec_ok = (S & 7) == 0;
  // staging roles
  const int wrow = tid >> 2, wk8 = (tid & 3) * 8;   // weight chunk: row, 8 consecutive k
  const int xk = tid >> 3, xc8 = (tid & 7) * 8;     // activation chunk: k row, 8 consecutive columns
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += PB_BK) {
    float wv8[8];
    if (!TRANS_W) {
      const int row = m0 + wrow < M ? m0 + wrow : 0;
      const float4 a = *reinterpret_cast<const float4*>(Wt + (size_t)row * K + k0 + wk8);
      const float4 b = *reinterpret_cast<const float4*>(Wt + (size_t)row * K + k0 + wk8 + 4);
      wv8[0] = a.x; wv8[1] = a.y; wv8[2] = a.z; wv8[3] = a.w; wv8[4] = b.x; wv8[5] = b.y; wv8[6] = b.z; wv8[7] = b.w;
    } else {  // thread -> k row xk, 8 consecutive m (contiguous in memory); M % 8 == 0
      const int mm = m0 + xc8 < M ? m0 + xc8 : 0;
      const float4 a = *reinterpret_cast<const float4*>(Wt + (size_t)(k0 + xk) * M + mm);
      const float4 b = *reinterpret_cast<const float4*>(Wt + (size_t)(k0 + xk) * M + mm + 4);
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
        const float sc = folded ? fsc[k0 + xk] : in_scale[k0 + xk], sh = folded ? fsh[k0 + xk] : in_shift[k0 + xk];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv8[i] = (s0 + xc8 + i < S) ? msl::act(xv8[i], sc, sh) : 0.f;
      }
    }
    __syncthreads();  // the previous chunk has been consumed
    {
      if (!TRANS_W) {
        u16x8 p;
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = msl::f2bf(wv8[i]);
        *reinterpret_cast<u16x8*>(&Ws[wrow * PB_LD + wk8]) = p;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) Ws[(xc8 + i) * PB_LD + xk] = msl::f2bf(wv8[i]);
      }
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

// ---- pointwise, pipelined form (S % 8 == 0, K % BK == 0) ------------------------------------------------------------------
// Same GEMM and tile (64 rows x 64 positions, 4 waves = 2 x 2 tiles of 32 x 32) as pw_fwd_bf16_kernel, restructured around
// what made that kernel slow (MFMA-busy 0.5 %): (1) the activation chunk keeps its memory orientation in LDS - rows = k,
// 64 positions + pad per row, written with ONE 16-byte store per thread and group - and the MFMA operand (8 consecutive k
// of one position) is gathered by the hardware transpose read ds_read_b64_tr_b16 (two per operand; row stride 192 B makes
// the 32-lane halves conflict-free); the same for the weights of the bwd-data form, which arrive k-major; (2) the global
// loads of iteration it+1 are in flight during the MFMAs of iteration it (registers), iterations = (position tile, K chunk)
// flattened, so a workgroup that owns several position tiles never waits for a cold load between tiles; (3) BK up to
// 128: the deep layers (K = 256 / 512 over 32-128 workgroups) take 2-4 memory round trips instead of 8-16.
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PT_BM = 64, PT_BN = 64, PT_RS = PT_BN + 32;  // LDS row stride of a k-major image: 192 bytes

__device__ __forceinline__ bf16x8 pt_tr_operand(const u16* img, int krow0, int col0, int lane) {
  // 32-column operand tile of a k-major image: lane (c = lane & 31, h = lane >> 5) gets k = krow0 + 8h .. + 7 of column col0 + c
  const int g = lane >> 4, l = lane & 15, q = l >> 2, pp = l & 3, h = g >> 1;
  const u16* a = img + (krow0 + 8 * h + q) * PT_RS + col0 + 16 * (g & 1) + 4 * pp;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * PT_RS));
  union { s16x4 v[2]; bf16x8 b; } u;
  u.v[0] = lo;
  u.v[1] = hi;
  return u.b;
}

template <int BK, bool STATS, bool TRANS_W>
__global__ __launch_bounds__(256) void pw_bf16_tr_kernel(const u16* __restrict__ Z, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, const float* __restrict__ Wt,
                                                         u16* __restrict__ Y, double* __restrict__ partials, int M, int K,
                                                         int S, int tiles_per_wg, msl::BnFold fold) {
  constexpr int NG = BK / 32;                    // 8-element groups per thread and chunk (X and W alike)
  constexpr int WLD = BK + 8;                    // row-major weight image: k contiguous, 16-byte aligned rows
  __shared__ __align__(16) u16 Xs[BK * PT_RS];
  __shared__ __align__(16) u16 Ws[TRANS_W ? BK * PT_RS : PT_BM * WLD];
  __shared__ float fsc[PB_MAXK], fsh[PB_MAXK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.z, m0 = blockIdx.y * PT_BM;
  const int tiles_img = (S + PT_BN - 1) / PT_BN, tile0 = blockIdx.x * tiles_per_wg;  // the last tile may be partial (S % 8 == 0)
  const int ntiles = min(tiles_per_wg, tiles_img - tile0), nk = K / BK, total = ntiles * nk;
  const u16* Zn = Z + (size_t)n * K * S;
  u16* Yn = Y + (size_t)n * M * S;
  const bool folded = fold.partials != nullptr, affine = folded || in_scale != nullptr;
  if (affine) {  // every channel's (scale, shift) once per workgroup (K <= PB_MAXK, host-checked)
    for (int c = tid; c < K; c += 256) {
      float a, b;
      if (folded) {
        float m_, i_;
        double v_;
        msl::bn_fold_serial(fold, c, a, b, m_, i_, v_);
      } else {
        a = in_scale[c];
        b = in_shift[c];
      }
      fsc[c] = a;
      fsh[c] = b;
    }
  }
  u16x8 xr[NG];
  f32x4 wr[NG][2];
  auto issue = [&](int it) {
    const int t = it / nk, kc = it - t * nk, k0 = kc * BK, s0 = (tile0 + t) * PT_BN;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int idx = tid + 256 * j, krow = idx >> 3, c8 = (idx & 7) * 8;
      const bool in = s0 + c8 < S;  // whole 8-column groups are inside or outside
      xr[j] = *reinterpret_cast<const u16x8*>(Zn + (size_t)(k0 + krow) * S + (in ? s0 + c8 : 0));
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int idx = tid + 256 * j;
      const float* src;
      if (TRANS_W) {  // k-major weights: thread -> k row, 8 consecutive m
        const int krow = idx >> 3, c8 = (idx & 7) * 8;
        src = Wt + (size_t)(k0 + krow) * M + (m0 + c8 < M ? m0 + c8 : 0);
      } else {        // m-major weights: thread -> row, 8 consecutive k
        const int row = idx / (BK / 8), k8 = (idx % (BK / 8)) * 8;
        src = Wt + (size_t)(m0 + row < M ? m0 + row : 0) * K + k0 + k8;
      }
      wr[j][0] = *reinterpret_cast<const f32x4*>(src);
      wr[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  };
  f32x16 acc = {0};
  if (total > 0) issue(0);
  for (int it = 0; it < total; ++it) {
    const int t = it / nk, kc = it - t * nk, k0 = kc * BK;
    __syncthreads();  // the previous iteration's operands have been consumed (first pass: publishes fsc / fsh)
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int idx = tid + 256 * j, krow = idx >> 3, c8 = (idx & 7) * 8;
      u16x8 o = xr[j];
      if (affine) {
        const float sc = fsc[k0 + krow], sh = fsh[k0 + krow];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = msl::f2bf(msl::act(msl::bf2f(o[e]), sc, sh));
      }
      if ((tile0 + t) * PT_BN + c8 >= S) o = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};  // columns past the map: exact zeros
      *reinterpret_cast<u16x8*>(&Xs[krow * PT_RS + c8]) = o;
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int idx = tid + 256 * j;
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = msl::f2bf(wr[j][0][e]);
        o[4 + e] = msl::f2bf(wr[j][1][e]);
      }
      if (TRANS_W) {
        const int krow = idx >> 3, c8 = (idx & 7) * 8;
        *reinterpret_cast<u16x8*>(&Ws[krow * PT_RS + c8]) = o;
      } else {
        const int row = idx / (BK / 8), k8 = (idx % (BK / 8)) * 8;
        *reinterpret_cast<u16x8*>(&Ws[row * WLD + k8]) = o;
      }
    }
    __syncthreads();
    if (it + 1 < total) issue(it + 1);  // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a;
      if (TRANS_W) a = pt_tr_operand(Ws, ks * 16, wm * 32, lane);
      else a = *reinterpret_cast<const bf16x8*>(&Ws[(wm * 32 + r) * WLD + ks * 16 + 8 * h]);
      const bf16x8 b = pt_tr_operand(Xs, ks * 16, wn * 32, lane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (kc == nk - 1) {  // D[row][col]: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
      const int tile = tile0 + t, col = tile * PT_BN + wn * 32 + r;
      const int NP = gridDim.z * tiles_img * 2, pidx = (n * tiles_img + tile) * 2 + wn;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const float v = acc[i];
        if (row < M && col < S) Yn[(size_t)row * S + col] = msl::f2bf(v);
        if (STATS) {
          const float sm = msl::half32_sum(v), q = msl::half32_sum(v * v);
          if (r == msl::HALF32_SUM_LANE && row < M && partials) {
            partials[(size_t)row * NP + pidx] = (double)sm;
            partials[((size_t)M + row) * NP + pidx] = (double)q;
          }
        }
      }
      acc = (f32x16){0};
    }
  }
}

// the pipelined transposed-read kernel takes maps of whole 8-position groups and K in whole chunks (always, when the shape allows)
static bool pw_tr_ok(int M, int K, int S) {
  constexpr int on = 1;
  return on && S % 8 == 0 && K % 32 == 0 && K <= PB_MAXK && M % 8 == 0;
}
template <bool STATS, bool TRANS_W>
static void pw_tr_launch(const u16* z, const float* in_scale, const float* in_shift, const float* w, u16* y, double* partials,
                         int N, int M, int K, int S, const msl::BnFold& fold, hipStream_t st) {
  const int tiles_img = msl::cdiv(S, PT_BN), mtiles = msl::cdiv(M, PT_BM);
  // several position tiles per workgroup once the launch has more than ~2 workgroups per CU (block 1: 2048 tiles)
  const int T = std::max(1, std::min(std::min(4, tiles_img), (tiles_img * mtiles * N) / 512));
  dim3 grid(msl::cdiv(tiles_img, T), mtiles, N);
#define MSL_PT(BK_) MSL_LAUNCH((pw_bf16_tr_kernel<BK_, STATS, TRANS_W>), grid, dim3(256), 0, st, z, in_scale, in_shift, w, y, \
                                       partials, M, K, S, T, fold)
  if (K % 128 == 0) MSL_PT(128);
  else if (K % 64 == 0) MSL_PT(64);
  else MSL_PT(32);
#undef MSL_PT
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

// The training step runs its head convolutions on the fp32 kernels (heads.hip: LDS-staged, faster at these sizes than the
// bf16 head kernel, and the head operands then carry no second rounding): relu(bn(y)) -> fp32 zero-haloed NCDHW copy
// (N, C, D+2, H+2, W+2), halo zeroed once at allocation.
// `fold`: the BatchNorm affine of the workgroup's channel rebuilt from the producer's statistics partials (no finalize launch
// in front; the batched finalize at the end of the pass writes the vectors the backward pass needs)
__global__ __launch_bounds__(256) void materialize_bf16_pad32_kernel(const u16* __restrict__ y, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, float* __restrict__ pad,
                                                                     int C, int D, int H, int W, msl::BnFold fold) {
  __shared__ float s_aff[2];
  const int S = D * H * W;
  const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, n = blockIdx.z;
  float sc, sh;
  if (fold.partials) {
    msl::bn_fold_block(fold, c, 1, &s_aff[0], &s_aff[1]);  // (all threads: ends with a barrier)
    sc = s_aff[0];
    sh = s_aff[1];
  } else {
    sc = scale[c];
    sh = shift[c];
  }
  if (p >= S) return;
  const int w = p % W, hh = (p / W) % H, d = p / (W * H);
  const float v = msl::act(msl::bf2f(y[((size_t)n * C + c) * S + p]), sc, sh);
  pad[(((size_t)(n * C + c) * (D + 2) + d + 1) * (H + 2) + hh + 1) * (W + 2) + w + 1] = v;
}

// ---- pointwise weight gradient -------------------------------------------------------------------------------------
// dW[co][ci] = sum over (n, p) of dY[n][co][p] * act(Z[n][ci][p]): positions are the MFMA K dimension and the contiguous
// axis of both operands, so lane (r, h) of v_mfma_f32_32x32x16_bf16 takes its 8 consecutive positions with one 16-byte
// load per operand row - no transposition, no LDS in the loop.  Same decomposition and slab layout as the fp32 kernel
// (pwconv.hip pw_bww_wave_kernel): wave = (32*MT) x 32 tile, 64-position chunks, 4 waves interleave chunks and meet in
// LDS once; [nslabs][Cout][Cin] fp32 slabs folded by msl_grad_reduce_batch.
template <int MT, bool AFFINE>
__global__ __launch_bounds__(256) void pw_bww_bf16_kernel(const u16* __restrict__ dY, const u16* __restrict__ Z,
                                                          const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                          float* __restrict__ out, int Cout, int Cin, int S, int chunks_per_img,
                                                          int total_chunks, int chunks_per_block) {
  __shared__ __align__(16) float red[4 * MT * 1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
  const int ks = blockIdx.x;
  const int tiles_n = Cin / 32;
  const int tm = blockIdx.y / tiles_n, tn = blockIdx.y % tiles_n;
  const int m0 = tm * 32 * MT, n0 = tn * 32;
  const int ch_lo = ks * chunks_per_block, ch_hi = min(total_chunks, ch_lo + chunks_per_block);
  const int mine = max(0, (ch_hi - ch_lo - wv + 3) / 4);
  float sc = 1.f, sh = 0.f;
  if (AFFINE) {
    sc = in_scale[n0 + c];
    sh = in_shift[n0 + c];
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x16){0};
  for (int i = 0; i < mine; ++i) {
    const int ch = ch_lo + wv + 4 * i;
    const int n = ch / chunks_per_img, s0 = (ch - n * chunks_per_img) * 64 + 8 * h;
    u16x8 a[MT][4], b[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const u16* p = dY + ((size_t)n * Cout + m0 + mt * 32 + c) * S + s0;
#pragma unroll
      for (int j = 0; j < 4; ++j) a[mt][j] = *reinterpret_cast<const u16x8*>(p + 16 * j);
    }
    const u16* q = Z + ((size_t)n * Cin + n0 + c) * S + s0;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const u16x8*>(q + 16 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u16x8 bv = b[j];
      if (AFFINE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = msl::f2bf(msl::act(msl::bf2f(bv[e]), sc, sh));
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mt][j]), __builtin_bit_cast(bf16x8, bv),
                                                          acc[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wv * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[mt][r];
  __syncthreads();
  float* dst = out + (size_t)ks * Cout * Cin;
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int q4 = threadIdx.x + e * 256;
    const int row = q4 >> 3, col = (q4 & 7) * 4;
    float4 v = *reinterpret_cast<const float4*>(&red[row * 32 + col]);
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const float4 t = *reinterpret_cast<const float4*>(&red[(w2 * MT * 32 + row) * 32 + col]);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    *reinterpret_cast<float4*>(dst + (size_t)(m0 + row) * Cin + n0 + col) = v;
  }
}

// Any-shape fallback of the pointwise weight gradient (position counts that are no multiple of 64, e.g. the 2^3 maps of a
// 64^3 input): workgroup = 16 x 16 tile of dW, positions staged through LDS 64 at a time; one slab.
template <bool AFFINE>
__global__ __launch_bounds__(256) void pw_bww_bf16_any_kernel(const u16* __restrict__ dY, const u16* __restrict__ Z,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift, float* __restrict__ out,
                                                              int N, int Cout, int Cin, int S) {
  __shared__ float sy[16][65], sz[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int co0 = blockIdx.y * 16, ci0 = blockIdx.x * 16;
  float acc = 0.f;
  for (int n = 0; n < N; ++n)
    for (int p0 = 0; p0 < S; p0 += 64) {
      __syncthreads();
      for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int r = e >> 6, c = e & 63, p = p0 + c;
        float a = 0.f, b = 0.f;
        if (p < S) {
          if (co0 + r < Cout) a = msl::bf2f(dY[((size_t)n * Cout + co0 + r) * S + p]);
          if (ci0 + r < Cin) {
            b = msl::bf2f(Z[((size_t)n * Cin + ci0 + r) * S + p]);
            if (AFFINE) b = msl::bf2f(msl::f2bf(fmaxf(fmaf(b, in_scale[ci0 + r], in_shift[ci0 + r]), 0.f)));
          }
        }
        sy[r][c] = a;
        sz[r][c] = b;
      }
      __syncthreads();
#pragma unroll 16
      for (int c = 0; c < 64; ++c) acc = fmaf(sy[ty][c], sz[tx][c], acc);
    }
  if (co0 + ty < Cout && ci0 + tx < Cin) out[(size_t)(co0 + ty) * Cin + ci0 + tx] = acc;
}

// ---- depthwise, stride-2 bwd-data ------------------------------------------------------------------------------------
// g_in[i] = sum over taps k with (i + 1 - k) even of w[k] * dy[(i + 1 - k) / 2].  Workgroup = (n, c) x an input tile of
// 4 x 8 x 64 voxels; the dy tile it touches (3 x 5 x 33) is staged in LDS as fp32.
constexpr int DB_TD = 4, DB_TH = 8, DB_TW = 64;
__global__ __launch_bounds__(256) void dw_bwd_data_s2_bf16_kernel(const u16* __restrict__ dy, const float* __restrict__ w,
                                                                  u16* __restrict__ g_in, int C, int D, int H, int W, int OD,
                                                                  int OH, int OW, int tiles_h, int tiles_w, int accumulate) {
  constexpr int LD = DB_TD / 2 + 1, LH = DB_TH / 2 + 1, LW = DB_TW / 2 + 1;
  __shared__ float tile[LD * LH * LW];
  const int nc = blockIdx.y, c = nc % C;
  const int t = blockIdx.x;
  const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, td = t / (tiles_w * tiles_h);
  const int id0 = td * DB_TD, ih0 = th * DB_TH, iw0 = tw * DB_TW;
  // outputs o with 2o - 1 + k = i for i in the tile: o from id0 / 2 (i = id0, k = 1; id0 even) to (id0 + TD) / 2
  const int od0 = id0 >> 1, oh0 = ih0 >> 1, ow0 = iw0 >> 1;
  const u16* dc = dy + (size_t)nc * OD * OH * OW;
  for (int e = threadIdx.x; e < LD * LH * LW; e += 256) {
    const int lw = e % LW, lh = (e / LW) % LH, ld = e / (LW * LH);
    const int od = od0 + ld, oh = oh0 + lh, ow = ow0 + lw;
    const bool ok = od < OD && oh < OH && ow < OW;
    tile[e] = ok ? msl::bf2f(dc[((size_t)od * OH + oh) * OW + ow]) : 0.f;
  }
  float wt[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wt[k] = w[c * 27 + k];
  __syncthreads();
  // thread -> (h, w pair): 8 x 32 pairs; 2 voxels along w, all TD planes
  const int lwp = threadIdx.x % 32, lh = threadIdx.x / 32;
  u16* gc = g_in + (size_t)nc * D * H * W;
#pragma unroll
  for (int d = 0; d < DB_TD; ++d) {
    const int id = id0 + d, ih = ih0 + lh;
#pragma unroll
    for (int ww = 0; ww < 2; ++ww) {
      const int iw = iw0 + 2 * lwp + ww;
      float acc = 0.f;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int tdd = d + 1 - kd;           // (id + 1 - kd) - id0, id0 even
        if (tdd < 0 || (tdd & 1)) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int thh = lh + 1 - kh;
          if (thh < 0 || (thh & 1)) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tww = 2 * lwp + ww + 1 - kw;
            if (tww < 0 || (tww & 1)) continue;
            acc = fmaf(wt[kd * 9 + kh * 3 + kw], tile[((tdd >> 1) * LH + (thh >> 1)) * LW + (tww >> 1)], acc);
          }
        }
      }
      if (id < D && ih < H && iw < W) {
        u16* go = gc + ((size_t)id * H + ih) * W + iw;
        if (accumulate) acc += msl::bf2f(*go);
        *go = msl::f2bf(acc);
      }
    }
  }
}

// ---- depthwise weight gradient ---------------------------------------------------------------------------------------
// dW[c][k] = sum over (n, o) of dz[n][c][o] * act(x[n][c][o*s - 1 + k]); the activated input tile is staged exactly as in the
// forward kernel, a thread multiplies its TD outputs' dz against the 27 taps, and the workgroup's 27 sums go out as fp64
// partials [C*27][NP] (folded by msl_grad_reduce_batch kind 1).
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_bww_bf16_kernel(const u16* __restrict__ dz, const u16* __restrict__ x,
                                                          const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                          double* __restrict__ partials, int C, int D, int H, int W, int OD,
                                                          int OH, int OW, int tiles_h, int tiles_w, int NP) {
  constexpr int ID = (DW_TD - 1) * STRIDE + 3, IH = (DW_TH - 1) * STRIDE + 3, IW = (DW_TW - 1) * STRIDE + 3;
  __shared__ float tile[ID * IH * IW];
  __shared__ float wsum[4][27];
  const int nc = blockIdx.y, c = nc % C, n = nc / C;
  const int t = blockIdx.x;
  const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, td = t / (tiles_w * tiles_h);
  const int od0 = td * DW_TD, oh0 = th * DW_TH, ow0 = tw * DW_TW;
  const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
  const u16* xc = x + (size_t)nc * D * H * W;
  for (int e = threadIdx.x; e < ID * IH * IW; e += 256) {
    const int lw = e % IW, lh = (e / IW) % IH, ld = e / (IW * IH);
    const int id = id0 + ld, ih = ih0 + lh, iw = iw0 + lw;
    const bool ok = id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
    float v = msl::bf2f(xc[ok ? ((size_t)id * H + ih) * W + iw : 0]);
    if (affine) v = msl::act(v, sc, sh);
    tile[e] = ok ? v : 0.f;
  }
  __syncthreads();
  const int lw = threadIdx.x % DW_TW, lh = threadIdx.x / DW_TW;
  const int ow = ow0 + lw, oh = oh0 + lh;
  float s[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) s[k] = 0.f;
#pragma unroll
  for (int d = 0; d < DW_TD; ++d) {
    const int od = od0 + d;
    const bool ok = od < OD && oh < OH && ow < OW;
    const float g = ok ? msl::bf2f(dz[((size_t)nc * OD + (ok ? od : 0)) * OH * OW + (size_t)(ok ? oh : 0) * OW + (ok ? ow : 0)]) : 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          s[kd * 9 + kh * 3 + kw] = fmaf(g, tile[((d * STRIDE + kd) * IH + lh * STRIDE + kh) * IW + lw * STRIDE + kw], s[kd * 9 + kh * 3 + kw]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float v = msl::wave_sum(s[k]);
    if (lane == 0) wsum[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int p = n * gridDim.x + blockIdx.x;
    const double tot = ((double)wsum[0][threadIdx.x] + (double)wsum[1][threadIdx.x]) + ((double)wsum[2][threadIdx.x] + (double)wsum[3][threadIdx.x]);
    partials[((size_t)c * 27 + threadIdx.x) * NP + p] = tot;
  }
}

// ---- BatchNorm + ReLU backward ---------------------------------------------------------------------------------------
// gm = g * [fma(y, scale, shift) > 0];  dbeta = sum gm;  dgamma = sum gm * xhat;  dy = scale * (gm - c1 - xhat * c2),
// c1 = dbeta / count, c2 = dgamma / count (same algebra and vectors as the fp32 kernels of bn.hip).
constexpr int BB_CHUNK = 4096;
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_bf16_kernel(const u16* __restrict__ g, const u16* __restrict__ y,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      double* __restrict__ partials, int C, int S, int chunks) {
  __shared__ double scratch[8];
  const int ck = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const size_t base = ((size_t)n * C + c) * S;
  const int lo = ck * BB_CHUNK, hi = min(S, lo + BB_CHUNK);
  float s1 = 0.f, s2 = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float yv = msl::bf2f(y[base + i]);
    const float gm = fmaf(yv, sc, sh) > 0.f ? msl::bf2f(g[base + i]) : 0.f;
    s1 += gm;
    s2 += gm * ((yv - mu) * is);
  }
  const double t1 = msl::block_sum((double)s1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum((double)s2, scratch);
  if (threadIdx.x == 0) {
    const int NP = gridDim.z * chunks, p = n * chunks + ck;
    partials[(size_t)c * NP + p] = t1;
    partials[((size_t)C + c) * NP + p] = t2;
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_bf16_kernel(const u16* __restrict__ g, const u16* __restrict__ y,
                                                                     const float* __restrict__ vec, u16* __restrict__ dy, int C,
                                                                     int S) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float sc = vec[c], sh = vec[C + c], mu = vec[2 * C + c], is = vec[3 * C + c], k1 = vec[4 * C + c], k2 = vec[5 * C + c];
  const size_t base = ((size_t)n * C + c) * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
    const float yv = msl::bf2f(y[base + i]);
    const float gm = fmaf(yv, sc, sh) > 0.f ? msl::bf2f(g[base + i]) : 0.f;
    dy[base + i] = msl::f2bf(sc * (gm - k1 - ((yv - mu) * is) * k2));
  }
}

// one workgroup per channel: reduce, then apply (the channel's data is re-read from cache): small layers in one launch
__global__ __launch_bounds__(256) void bn_relu_bwd_fused_bf16_kernel(const u16* __restrict__ g, const u16* __restrict__ y,
                                                                     const float* __restrict__ vec, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, u16* __restrict__ dy, int N,
                                                                     int C, int S, double count) {
  __shared__ double scratch[8];
  __shared__ float coef[2];
  const int c = blockIdx.x;
  const float sc = vec[c], sh = vec[C + c], mu = vec[2 * C + c], is = vec[3 * C + c];
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * S;
    for (int i = threadIdx.x; i < S; i += 256) {
      const float yv = msl::bf2f(y[base + i]);
      const float gm = fmaf(yv, sc, sh) > 0.f ? msl::bf2f(g[base + i]) : 0.f;
      s1 += gm;
      s2 += gm * ((yv - mu) * is);
    }
  }
  const double t1 = msl::block_sum((double)s1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum((double)s2, scratch);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)t1;
    dgamma[c] = (float)t2;
    coef[0] = (float)(t1 / count);
    coef[1] = (float)(t2 / count);
  }
  __syncthreads();
  const float k1 = coef[0], k2 = coef[1];
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * S;
    for (int i = threadIdx.x; i < S; i += 256) {
      const float yv = msl::bf2f(y[base + i]);
      const float gm = fmaf(yv, sc, sh) > 0.f ? msl::bf2f(g[base + i]) : 0.f;
      dy[base + i] = msl::f2bf(sc * (gm - k1 - ((yv - mu) * is) * k2));
    }
  }
}

// Register-resident form for N*S <= NT*IPT*8 elements per channel (S % 8 == 0): the channel's g and y are read ONCE (16-byte
// loads, all of a thread's loads in flight together), kept in registers across the block-wide reduction, and dL/dy is
// written from them (bn.hip: bn_relu_bwd_fused_reg_kernel, here with 8 bf16 per load).
template <int NT, int IPT>
__global__ __launch_bounds__(NT) void bn_relu_bwd_fused_reg_bf16_kernel(const u16* __restrict__ g, const u16* __restrict__ y,
                                                                        const float* __restrict__ vec, float* __restrict__ dgamma,
                                                                        float* __restrict__ dbeta, u16* __restrict__ dy, int N,
                                                                        int C, int S, double count) {
  __shared__ double scratch[16];
  __shared__ float coef[2];
  const int c = blockIdx.x;
  const float sc = vec[c], sh = vec[C + c], mu = vec[2 * C + c], is = vec[3 * C + c];
  const int S8 = S >> 3, total8 = N * S8;
  u16x8 gv[IPT], yv[IPT];
  size_t off[IPT];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    const int t = threadIdx.x + u * NT;
    const int tt = t < total8 ? t : 0;
    const int n = tt / S8, i8 = tt - n * S8;
    off[u] = ((size_t)n * C + c) * S + (size_t)i8 * 8;
    gv[u] = *reinterpret_cast<const u16x8*>(g + off[u]);
    yv[u] = *reinterpret_cast<const u16x8*>(y + off[u]);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    const bool in = threadIdx.x + u * NT < total8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float yy = msl::bf2f(yv[u][k]);
      const bool on = in && fmaf(yy, sc, sh) > 0.f;
      if (!on) gv[u][k] = 0;  // gm (bf16 zero)
      const float gm = msl::bf2f(gv[u][k]);
      s1 += gm;
      s2 += gm * ((yy - mu) * is);
    }
  }
  const double t1 = msl::block_sum((double)s1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum((double)s2, scratch);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)t1;
    dgamma[c] = (float)t2;
    coef[0] = (float)(t1 / count);
    coef[1] = (float)(t2 / count);
  }
  __syncthreads();
  const float k1 = coef[0], k2 = coef[1];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    if (threadIdx.x + u * NT < total8) {
      u16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        o[k] = msl::f2bf(sc * (msl::bf2f(gv[u][k]) - k1 - ((msl::bf2f(yv[u][k]) - mu) * is) * k2));
      *reinterpret_cast<u16x8*>(dy + off[u]) = o;
    }
  }
}

}  // namespace

extern "C" {

static bool dw_bf16_use_wave() {  // (the LDS-tiled any-shape kernels take what the wave kernels do not)
  constexpr int on = 1;
  return on != 0;
}

int msl_dwconv_fwd_bf16_num_partials(int N, int C, int D, int H, int W, int stride) {
  if (dw_bf16_use_wave()) {
    const int np = msl_dwconv_wave_num_partials(N, C, D, H, W, stride);
    if (np > 0) return np;
  }
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return N * msl::cdiv(OD, DW_TD) * msl::cdiv(OH, DW_TH) * msl::cdiv(OW, DW_TW);
}

// x (N,C,D,H,W) bf16 raw + optional input affine -> y (N,C,OD,OH,OW) bf16 raw (+ fp64 stat partials [2][C][NP] or NULL)
int msl_dwconv_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  if (dw_bf16_use_wave() && (msl_dwconv_wave_num_partials(N, C, D, H, W, stride) > 0 ||
                             (!partials && msl_dwconv_fwd_eval_rows_ok(N, C, D, H, W, stride) == 1)))
    return msl_dwconv_fwd_wave_bf16(x, in_scale, in_shift, w, y, partials, N, C, D, H, W, stride, 0, 0, stream);
  if (!partials && D * H * W <= 512 && (D + 2) * (H + 2) * (W + 2) <= 1024)  // eval mode, a map no wave / rows kernel takes
    return msl_dwconv_fwd_small_eval_bf16(x, in_scale, in_shift, w, y, N, C, D, H, W, stride, stream);
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int tiles_d = msl::cdiv(OD, DW_TD), tiles_h = msl::cdiv(OH, DW_TH), tiles_w = msl::cdiv(OW, DW_TW);
  const int NP = N * tiles_d * tiles_h * tiles_w;
  dim3 grid(tiles_d * tiles_h * tiles_w, N * C);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1)
    MSL_LAUNCH(dw_fwd_bf16_kernel<1>, grid, dim3(256), 0, st, (const u16*)x, in_scale, in_shift, w, (u16*)y, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP, 0, 0);
  else
    MSL_LAUNCH(dw_fwd_bf16_kernel<2>, grid, dim3(256), 0, st, (const u16*)x, in_scale, in_shift, w, (u16*)y, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP, 0, 0);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_pwconv_fwd_bf16_num_partials(int N, int S) { return N * msl::cdiv(S, PB_BN) * 2; }

// z (N,Cin,S) bf16 raw + optional input affine -> y (N,Cout,S) bf16 raw (+ stat partials [2][Cout][NP] or NULL)
static const msl::BnFold pb_nofold{nullptr, 0, 0, 1.0, nullptr, nullptr, 0.f};

static int pw_fwd_bf16_impl(const void* z, const float* in_scale, const float* in_shift, const msl::BnFold& fold, const float* w,
                            void* y, double* partials, int N, int Cin, int Cout, int S, void* stream) {
  if (N <= 0 || S <= 0 || Cin % PB_BK != 0 || Cout <= 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (pw_tr_ok(Cout, Cin, S)) {
    if (partials) pw_tr_launch<true, false>((const u16*)z, in_scale, in_shift, w, (u16*)y, partials, N, Cout, Cin, S, fold, st);
    else pw_tr_launch<false, false>((const u16*)z, in_scale, in_shift, w, (u16*)y, partials, N, Cout, Cin, S, fold, st);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(msl::cdiv(S, PB_BN), msl::cdiv(Cout, PB_BM), N);
  if (partials)
    MSL_LAUNCH(pw_fwd_bf16_kernel<true>, grid, dim3(256), 0, st, (const u16*)z, in_scale, in_shift, w, (u16*)y,
                       partials, Cout, Cin, S, fold);
  else
    MSL_LAUNCH(pw_fwd_bf16_kernel<false>, grid, dim3(256), 0, st, (const u16*)z, in_scale, in_shift, w, (u16*)y,
                       partials, Cout, Cin, S, fold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_pwconv_fwd_bf16(const void* z, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int Cin, int Cout, int S, void* stream) {
  return pw_fwd_bf16_impl(z, in_scale, in_shift, pb_nofold, w, y, partials, N, Cin, Cout, S, stream);
}

// ... with the input BatchNorm folded from its producer's statistics partials [2][Cin][in_np] (in_np <= 64, Cin <= 1024: every
// workgroup rebuilds the Cin (scale, shift) pairs in its prologue, in the finalize kernel's summation order -> the same bits)
int msl_pwconv_fwd_bf16_fold(const void* z, const double* in_partials, int in_np, double in_count, const float* gamma,
                             const float* beta, float eps, const float* w, void* y, double* partials, int N, int Cin, int Cout,
                             int S, void* stream) {
  if (!in_partials || in_np <= 0) return MSL_ERR_ARG;
  if (in_np > 64 || Cin > PB_MAXK) return MSL_ERR_UNSUPPORTED;
  const msl::BnFold fold{in_partials, in_np, Cin, in_count, gamma, beta, eps};
  return pw_fwd_bf16_impl(z, nullptr, nullptr, fold, w, y, partials, N, Cin, Cout, S, stream);
}

// y (N,C,D,H,W) bf16 raw + affine -> pad_cl (N,D+2,H+2,W+2,C) bf16 (halo pre-zeroed by the caller) [+ plain fp32 NCDHW]
int msl_bn_relu_materialize_bf16(const void* y, const float* scale, const float* shift, float* plain, void* pad_cl, int N,
                                 int C, int D, int H, int W, void* stream) {
  if (N <= 0 || C % 8 != 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift) return MSL_ERR_ARG;
  dim3 grid(msl::cdiv(D * H * W, 256), C / 8, N);
  MSL_LAUNCH(materialize_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)y, scale, shift, plain,
                     (u16*)pad_cl, C, D, H, W);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// y (N,C,D,H,W) bf16 raw + affine -> pad (N,C,D+2,H+2,W+2) fp32 (halo pre-zeroed by the caller): what the fp32 head kernels read
int msl_bn_relu_materialize_bf16_pad32(const void* y, const float* scale, const float* shift, float* pad, int N, int C, int D,
                                       int H, int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !pad) return MSL_ERR_ARG;
  dim3 grid(msl::cdiv(D * H * W, 256), C, N);
  MSL_LAUNCH(materialize_bf16_pad32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)y, scale, shift, pad, C,
                     D, H, W, pb_nofold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// the same with the affine folded from the producer's statistics partials [2][C][in_np] (training mode)
int msl_bn_relu_materialize_bf16_pad32_fold(const void* y, const double* in_partials, int in_np, double in_count,
                                            const float* gamma, const float* beta, float eps, float* pad, int N, int C, int D,
                                            int H, int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || !in_partials || in_np <= 0 || !pad) return MSL_ERR_ARG;
  const msl::BnFold fold{in_partials, in_np, C, in_count, gamma, beta, eps};
  dim3 grid(msl::cdiv(D * H * W, 256), C, N);
  MSL_LAUNCH(materialize_bf16_pad32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)y, (const float*)nullptr,
             (const float*)nullptr, pad, C, D, H, W, fold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ---- bf16 training step: backward kernels ----------------------------------------------------------------------------
// dy (N,Cout,S) bf16 -> g_in (N,Cin,S) bf16 = W^T . dy   (bf16 MFMA, fp32 accumulate)
int msl_pwconv_bwd_data_bf16(const void* dy, const float* w, void* g_in, int N, int Cin, int Cout, int S, void* stream) {
  if (N <= 0 || S <= 0 || Cout % PB_BK != 0 || Cin % 8 != 0) return MSL_ERR_ARG;
  if (pw_tr_ok(Cin, Cout, S)) {
    pw_tr_launch<false, true>((const u16*)dy, nullptr, nullptr, w, (u16*)g_in, nullptr, N, Cin, Cout, S, pb_nofold, (hipStream_t)stream);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(msl::cdiv(S, PB_BN), msl::cdiv(Cin, PB_BM), N);
  MSL_LAUNCH((pw_fwd_bf16_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)dy, nullptr, nullptr,
                     w, (u16*)g_in, nullptr, Cin, Cout, S, pb_nofold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// position split of msl_pwconv_bwd_weight_slabs_bf16 (number of [Cout][Cin] fp32 slabs; 1: the result itself)
static inline bool bww_bf16_plan(int N, int Cin, int Cout, int S, int& mt, int& tiles, int& cpi, int& total, int& ks, int& cpb) {
  if (S % 64 != 0 || Cin % 32 != 0 || Cout % 32 != 0) return false;
  mt = (Cout % 64 == 0) ? 2 : 1;
  tiles = (Cout / (32 * mt)) * (Cin / 32);
  cpi = S / 64;
  total = N * cpi;
  ks = std::max(1, std::min(256 / std::max(1, tiles), total / 4));
  cpb = msl::cdiv(total, ks);
  ks = msl::cdiv(total, cpb);
  return true;
}
int msl_pwconv_bwd_weight_bf16_nslabs(int N, int Cin, int Cout, int S) {
  int mt, tiles, cpi, total, ks, cpb;
  if (N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0) return MSL_ERR_ARG;
  return bww_bf16_plan(N, Cin, Cout, S, mt, tiles, cpi, total, ks, cpb) ? ks : 1;
}
// dy (N,Cout,S), z (N,Cin,S) bf16 -> out = [nslabs][Cout][Cin] fp32 partial weight gradients (MFMA kernel when S % 64 == 0
// and the channel counts are multiples of 32; a plain one-slab kernel otherwise)
int msl_pwconv_bwd_weight_slabs_bf16(const void* dy, const void* z, const float* in_scale, const float* in_shift, float* out,
                                     int N, int Cin, int Cout, int S, void* stream) {
  int mt, tiles, cpi, total, ks, cpb;
  if (N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!bww_bf16_plan(N, Cin, Cout, S, mt, tiles, cpi, total, ks, cpb)) {
    dim3 g(msl::cdiv(Cin, 16), msl::cdiv(Cout, 16));
    if (in_scale)
      MSL_LAUNCH(pw_bww_bf16_any_kernel<true>, g, dim3(256), 0, st, (const u16*)dy, (const u16*)z, in_scale, in_shift, out,
                         N, Cout, Cin, S);
    else
      MSL_LAUNCH(pw_bww_bf16_any_kernel<false>, g, dim3(256), 0, st, (const u16*)dy, (const u16*)z, in_scale, in_shift, out,
                         N, Cout, Cin, S);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(ks, tiles);
#define MSL_BW(MT_, A_)                                                                                              \
  MSL_LAUNCH((pw_bww_bf16_kernel<MT_, A_>), grid, dim3(256), 0, st, (const u16*)dy, (const u16*)z, in_scale, \
                     in_shift, out, Cout, Cin, S, cpi, total, cpb)
  if (mt == 2) {
    if (in_scale) MSL_BW(2, true); else MSL_BW(2, false);
  } else {
    if (in_scale) MSL_BW(1, true); else MSL_BW(1, false);
  }
#undef MSL_BW
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// dy (N,C,OD,OH,OW) bf16 -> g_in (N,C,D,H,W) bf16; accumulate != 0 adds into g_in
int msl_dwconv_bwd_data_bf16(const void* dy, const float* w, void* g_in, int N, int C, int D, int H, int W, int stride,
                             int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && dw_bf16_use_wave() && msl_dwconv_wave_num_partials(N, C, D, H, W, 1) > 0)
    return msl_dwconv_fwd_wave_bf16(dy, nullptr, nullptr, w, g_in, nullptr, N, C, D, H, W, 1, 1, accumulate, stream);
  if (stride == 2 && dw_bf16_use_wave() && W % 4 == 0)
    return msl_dwconv_bwd_data_s2_patch_bf16(dy, w, g_in, nullptr, nullptr, nullptr, N, C, D, H, W, accumulate, stream);
  if (stride == 1) {  // the forward kernel with reversed taps
    const int tiles_d = msl::cdiv(D, DW_TD), tiles_h = msl::cdiv(H, DW_TH), tiles_w = msl::cdiv(W, DW_TW);
    dim3 grid(tiles_d * tiles_h * tiles_w, N * C);
    MSL_LAUNCH(dw_fwd_bf16_kernel<1>, grid, dim3(256), 0, st, (const u16*)dy, nullptr, nullptr, w, (u16*)g_in, nullptr, C,
                       D, H, W, D, H, W, tiles_h, tiles_w, 0, 1, accumulate);
  } else {
    const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int tiles_d = msl::cdiv(D, DB_TD), tiles_h = msl::cdiv(H, DB_TH), tiles_w = msl::cdiv(W, DB_TW);
    dim3 grid(tiles_d * tiles_h * tiles_w, N * C);
    MSL_LAUNCH(dw_bwd_data_s2_bf16_kernel, grid, dim3(256), 0, st, (const u16*)dy, w, (u16*)g_in, C, D, H, W, OD, OH, OW,
                       tiles_h, tiles_w, accumulate);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// dz (N,C,OD,OH,OW), x (N,C,D,H,W) bf16 (+ input affine) -> fp64 partials [C*27][NP], NP = msl_dwconv_fwd_bf16_num_partials
int msl_dwconv_bwd_weight_bf16(const void* dz, const void* x, const float* in_scale, const float* in_shift, double* partials,
                               int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !partials) return MSL_ERR_ARG;
  if (dw_bf16_use_wave() && msl_dwconv_wave_num_partials(N, C, D, H, W, stride) > 0)
    return msl_dwconv_bwd_weight_wave_bf16(dz, x, in_scale, in_shift, partials, N, C, D, H, W, stride, stream);
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int tiles_d = msl::cdiv(OD, DW_TD), tiles_h = msl::cdiv(OH, DW_TH), tiles_w = msl::cdiv(OW, DW_TW);
  const int NP = N * tiles_d * tiles_h * tiles_w;
  dim3 grid(tiles_d * tiles_h * tiles_w, N * C);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1)
    MSL_LAUNCH(dw_bww_bf16_kernel<1>, grid, dim3(256), 0, st, (const u16*)dz, (const u16*)x, in_scale, in_shift, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP);
  else
    MSL_LAUNCH(dw_bww_bf16_kernel<2>, grid, dim3(256), 0, st, (const u16*)dz, (const u16*)x, in_scale, in_shift, partials,
                       C, D, H, W, OD, OH, OW, tiles_h, tiles_w, NP);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_relu_bwd_bf16_num_partials(int N, int S) { return N * msl::cdiv(S, BB_CHUNK); }

// partials [2][C][NP] of (sum gm, sum gm * xhat); fold them with msl_bn_bwd_finalize (bn.hip), then _apply_bf16
int msl_bn_relu_bwd_reduce_bf16(const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, double* partials, int N, int C, int S, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  const int chunks = msl::cdiv(S, BB_CHUNK);
  MSL_LAUNCH(bn_relu_bwd_reduce_bf16_kernel, dim3(chunks, C, N), dim3(256), 0, (hipStream_t)stream, (const u16*)g,
                     (const u16*)y, scale, shift, mean, invstd, partials, C, S, chunks);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// vec = (6, C) rows [scale, shift, mean, invstd, c1, c2]; dy may alias g
int msl_bn_relu_bwd_apply_bf16(const void* g, const void* y, const float* vec, void* dy, int N, int C, int S, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_relu_bwd_apply_bf16_kernel, dim3(std::min(msl::cdiv(S, 1024), 64), C, N), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)g, (const u16*)y, vec, (u16*)dy, C, S);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// whole BatchNorm + ReLU backward of small layers in one launch (one workgroup per channel); vec rows 0-3 are read
int msl_bn_relu_bwd_fused_bf16(const void* g, const void* y, const float* vec, float* dgamma, float* dbeta, void* dy, int N,
                               int C, int S, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total8 = (long long)N * (S >> 3);
#define MSL_BN_REG(NT_, IPT_)                                                                                              \
  MSL_LAUNCH((bn_relu_bwd_fused_reg_bf16_kernel<NT_, IPT_>), dim3(C), dim3(NT_), 0, st, (const u16*)g, (const u16*)y, \
                     vec, dgamma, dbeta, (u16*)dy, N, C, S, (double)N * (double)S)
  if ((S & 7) == 0 && total8 <= 4096) {
    if (total8 <= 64) MSL_BN_REG(64, 1);
    else if (total8 <= 256) MSL_BN_REG(256, 1);
    else if (total8 <= 512) MSL_BN_REG(256, 2);
    else if (total8 <= 1024) MSL_BN_REG(512, 2);
    else if (total8 <= 2048) MSL_BN_REG(1024, 2);
    else MSL_BN_REG(1024, 4);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
#undef MSL_BN_REG
  MSL_LAUNCH(bn_relu_bwd_fused_bf16_kernel, dim3(C), dim3(256), 0, st, (const u16*)g, (const u16*)y, vec,
                     dgamma, dbeta, (u16*)dy, N, C, S, (double)N * (double)S);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
