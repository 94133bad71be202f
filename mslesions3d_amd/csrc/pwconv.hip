// Pointwise Conv3d(Cin -> Cout, k1, no bias) = per-image GEMM  Y_n[Cout x S] = W[Cout x Cin] . A_n[Cin x S]
// Reference: Block.conv2 (lesions3d/mobilenet.py:40,45).  NCDHW fp32: the spatial axis S = D*H*W is the
// contiguous one, so it is the GEMM's N (lane) dimension: every global access is a coalesced row segment.
//
// The only GEMM-shaped work on the path -> the only MFMA user (v_mfma_f32_32x32x2_f32: exact fp32, bit
// equal to a k-ordered fmaf chain).  A_n = relu(bn(z)) is re-created from the raw depthwise output while
// the tile is staged into LDS; per-channel (sum, sumsq) of the raw output are reduced from the accumulator
// registers and emitted as fp64 partials for the following BatchNorm.
//   forward : M = Cout, K = Cin,  W element (m,k) = W[m*K + k]
//   bwd-data: M = Cin,  K = Cout, W element (m,k) = W[k*M + m]   (transposed read of the same weights)
//   bwd-wgt : dW[Cout x Cin] = sum_{n,s} dY[co,s] * A[ci,s]      (split over s, fixed-order slab reduction)
#include "common.hpp"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 128, BK = 32;
constexpr int FOLD_MAXK = 512;  // input channels whose (scale, shift) are kept in LDS
constexpr int WS_LD = BK + 1;

template <bool AFFINE, bool STATS, bool TRANS_W>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const float* __restrict__ X,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ Wt, float* __restrict__ Y,
                                                      double* __restrict__ partials, int M, int K, int S,
                                                      msl::BnFold fold) {
  __shared__ __align__(16) float Xs[BK][BN];
  __shared__ float Ws[BM][WS_LD];
  __shared__ float red[2][2][BM];  // [sum|sumsq][wave column][row]
  __shared__ float f_sc[AFFINE ? FOLD_MAXK : 1], f_sh[AFFINE ? FOLD_MAXK : 1];
  const int n = blockIdx.z, m0 = blockIdx.y * BM, s0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;  // 2 x 2 waves: 32 rows x 64 columns each
  const float* Xn = X + (size_t)n * K * S;
  const bool vec_ok = (S & 3) == 0;

  // chunk k0: 32 activation rows x 128 columns + 64 x 32 weights, through registers (the next chunk's loads are in
  // flight during this chunk's MFMAs)
  float4 xr[4], wr[2];
  auto load_chunk = [&](int k0) {
    if (vec_ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xr[i] = msl::load4_zfill<true>(Xn + (size_t)(k0 + (tid >> 5) + i * 8) * S, s0 + (tid & 31) * 4, S);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xr[i] = msl::load4_zfill<false>(Xn + (size_t)(k0 + (tid >> 5) + i * 8) * S, s0 + (tid & 31) * 4, S);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!TRANS_W) {
        const int m = (tid >> 3) + i * 32, k4 = (tid & 7) * 4;
        // rows >= M read row 0: their products land in output rows that are never stored or counted
        wr[i] = *reinterpret_cast<const float4*>(Wt + (size_t)(m0 + m < M ? m0 + m : 0) * K + k0 + k4);
      } else {
        const int k = (tid >> 4) + i * 16, m4 = (tid & 15) * 4;
        // M % 4 == 0; columns >= M read column 0 (output rows that are never stored)
        wr[i] = *reinterpret_cast<const float4*>(Wt + (size_t)(k0 + k) * M + (m0 + m4 < M ? m0 + m4 : 0));
      }
    }
  };
  auto store_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 5) + i * 8, c4 = (tid & 31) * 4, col = s0 + c4;
      float4 v = xr[i];
      if (AFFINE) {
        const float sc = f_sc[k0 + r], sh = f_sh[k0 + r];
        v.x = col < S ? msl::act(v.x, sc, sh) : 0.f;
        v.y = col + 1 < S ? msl::act(v.y, sc, sh) : 0.f;
        v.z = col + 2 < S ? msl::act(v.z, sc, sh) : 0.f;
        v.w = col + 3 < S ? msl::act(v.w, sc, sh) : 0.f;
      }
      *reinterpret_cast<float4*>(&Xs[r][c4]) = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = wr[i];
      if (!TRANS_W) {
        const int m = (tid >> 3) + i * 32, k4 = (tid & 7) * 4;
        Ws[m][k4] = v.x; Ws[m][k4 + 1] = v.y; Ws[m][k4 + 2] = v.z; Ws[m][k4 + 3] = v.w;
      } else {
        const int k = (tid >> 4) + i * 16, m4 = (tid & 15) * 4;
        Ws[m4][k] = v.x; Ws[m4 + 1][k] = v.y; Ws[m4 + 2][k] = v.z; Ws[m4 + 3][k] = v.w;
      }
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  load_chunk(0);
  // the input affine goes to LDS behind the first chunk's loads (one memory round trip for both); the chunk loop's
  // first barrier publishes it
  if (AFFINE) {
    if (fold.partials) {
      msl::bn_fold_block(fold, 0, K, f_sc, f_sh);
    } else {
      for (int k = threadIdx.x; k < K && k < FOLD_MAXK; k += blockDim.x) {
        f_sc[k] = in_scale[k];
        f_sh[k] = in_shift[k];
      }
    }
  }
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // the previous chunk has been consumed
    store_chunk(k0);
    __syncthreads();
    if (k0 + BK < K) load_chunk(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int kr = 2 * kk + (lane >> 5);
      const float a = Ws[wm * 32 + (lane & 31)][kr];
      const float b0 = Xs[kr][wn * 64 + (lane & 31)];
      const float b1 = Xs[kr][wn * 64 + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
  }

  // ---- epilogue: D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* Yn = Y + (size_t)n * M * S;
  const int colbase = s0 + wn * 64 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && colbase < S) Yn[(size_t)row * S + colbase] = acc0[r];
    if (row < M && colbase + 32 < S) Yn[(size_t)row * S + colbase + 32] = acc1[r];
  }
  if (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = acc0[r] + acc1[r];
      float q = fmaf(acc0[r], acc0[r], acc1[r] * acc1[r]);
#pragma unroll
      for (int m = 16; m > 0; m >>= 1) {
        s += __shfl_xor(s, m, 64);
        q += __shfl_xor(q, m, 64);
      }
      if ((lane & 31) == 0) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[0][wn][row] = s;
        red[1][wn][row] = q;
      }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < M && partials) {
      const int NP = gridDim.z * gridDim.x, p = n * gridDim.x + blockIdx.x;
      partials[(size_t)(m0 + tid) * NP + p] = (double)red[0][0][tid] + (double)red[0][1][tid];
      partials[((size_t)M + m0 + tid) * NP + p] = (double)red[1][0][tid] + (double)red[1][1][tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K-split variant for K >= 128 (blocks 3-7 forward, blocks 2-7 bwd-data): the GEMMs of the tail layers are tiny
// (down to 512 x 64 x 512 per image), so a 64 x 128 tile with a serial K loop leaves the chip idle behind load
// latency.  Here a workgroup owns a 32 x 64 tile and its 4 waves each take a quarter of K (private LDS
// staging, next chunk prefetched into registers during the MFMAs), followed by a fixed-order reduction.
constexpr int KS_BM = 32, KS_BN = 64;
constexpr int KS_WAVE_LDS = BK * KS_BN + KS_BM * WS_LD;  // floats per wave

template <bool AFFINE, bool STATS, bool TRANS_W, int NW>
__global__ __launch_bounds__(NW * 64) void pw_gemm_ksplit_kernel(const float* __restrict__ X,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const float* __restrict__ Wt, float* __restrict__ Y,
                                                             double* __restrict__ partials, int M, int K, int S,
                                                             msl::BnFold fold) {
  extern __shared__ __align__(16) float lds[];  // NW * KS_WAVE_LDS floats
  __shared__ float f_sc[AFFINE ? FOLD_MAXK : 1], f_sh[AFFINE ? FOLD_MAXK : 1];
  const int n = blockIdx.z, m0 = blockIdx.y * KS_BM, s0 = blockIdx.x * KS_BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* Xs = lds + wv * KS_WAVE_LDS;   // [BK][KS_BN]
  float* Ws = Xs + BK * KS_BN;          // [KS_BM][WS_LD]
  const float* Xn = X + (size_t)n * K * S;
  const bool vec_ok = (S & 3) == 0;
  const int kq = K / NW;                // K range of this wave
  const int kbeg = wv * kq;
  const int nchunks = kq / BK;

  float4 xr[8], wr[4];
  auto load_chunk = [&](int k0) {
    if (vec_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xr[i] = msl::load4_zfill<true>(Xn + (size_t)(k0 + (lane >> 4) + i * 4) * S, s0 + (lane & 15) * 4, S);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xr[i] = msl::load4_zfill<false>(Xn + (size_t)(k0 + (lane >> 4) + i * 4) * S, s0 + (lane & 15) * 4, S);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!TRANS_W) {
        const int m = (lane >> 3) + i * 8, k4 = (lane & 7) * 4;
        // rows >= M read row 0: their products land in output rows that are never stored or counted
        wr[i] = *reinterpret_cast<const float4*>(Wt + (size_t)(m0 + m < M ? m0 + m : 0) * K + k0 + k4);
      } else {
        const int k = (lane >> 3) + i * 8, m4 = (lane & 7) * 4;
        // M % 4 == 0; columns >= M read column 0 (output rows that are never stored)
        wr[i] = *reinterpret_cast<const float4*>(Wt + (size_t)(k0 + k) * M + (m0 + m4 < M ? m0 + m4 : 0));
      }
    }
  };
  auto store_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (lane >> 4) + i * 4, c4 = (lane & 15) * 4, col = s0 + c4;
      float4 v = xr[i];
      if (AFFINE) {
        const float sc = f_sc[k0 + r], sh = f_sh[k0 + r];
        v.x = col < S ? msl::act(v.x, sc, sh) : 0.f;
        v.y = col + 1 < S ? msl::act(v.y, sc, sh) : 0.f;
        v.z = col + 2 < S ? msl::act(v.z, sc, sh) : 0.f;
        v.w = col + 3 < S ? msl::act(v.w, sc, sh) : 0.f;
      }
      *reinterpret_cast<float4*>(Xs + r * KS_BN + c4) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!TRANS_W) {
        const int m = (lane >> 3) + i * 8, k4 = (lane & 7) * 4;
        float* d = Ws + m * WS_LD + k4;
        d[0] = wr[i].x; d[1] = wr[i].y; d[2] = wr[i].z; d[3] = wr[i].w;
      } else {
        const int k = (lane >> 3) + i * 8, m4 = (lane & 7) * 4;
        Ws[(m4 + 0) * WS_LD + k] = wr[i].x; Ws[(m4 + 1) * WS_LD + k] = wr[i].y;
        Ws[(m4 + 2) * WS_LD + k] = wr[i].z; Ws[(m4 + 3) * WS_LD + k] = wr[i].w;
      }
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  load_chunk(kbeg);
  // the input affine goes to LDS behind the first chunk's loads (one memory round trip for both); the chunk loop's
  // first barrier publishes it
  if (AFFINE) {
    if (fold.partials) {
      msl::bn_fold_block(fold, 0, K, f_sc, f_sh);
    } else {
      for (int k = threadIdx.x; k < K && k < FOLD_MAXK; k += blockDim.x) {
        f_sc[k] = in_scale[k];
        f_sh[k] = in_shift[k];
      }
    }
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();  // the previous chunk has been consumed
    store_chunk(kbeg + ch * BK);
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk(kbeg + (ch + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int kr = 2 * kk + (lane >> 5);
      const float a = Ws[(lane & 31) * WS_LD + kr];
      const float b0 = Xs[kr * KS_BN + (lane & 31)];
      const float b1 = Xs[kr * KS_BN + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
  }
  // fixed-order tree reduction over the NW K-slices: the upper half of the live waves parks its tile in LDS, the
  // lower half adds it, until wave 0 holds the sum
  __syncthreads();
  constexpr int RLD = KS_BN + 1;
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wv >= half && wv < 2 * half) {
      float* slot = lds + (wv - half) * (KS_BM * RLD);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slot[row * RLD + (lane & 31)] = acc0[r];
        slot[row * RLD + 32 + (lane & 31)] = acc1[r];
      }
    }
    __syncthreads();
    if (wv < half) {
      const float* slot = lds + wv * (KS_BM * RLD);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        acc0[r] += slot[row * RLD + (lane & 31)];
        acc1[r] += slot[row * RLD + 32 + (lane & 31)];
      }
    }
    __syncthreads();
  }
  if (wv != 0) return;
  float* Yn = Y + (size_t)n * M * S;
  const int colbase = s0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int row = m0 + rl;
    const float v0 = acc0[r];
    const float v1 = acc1[r];
    if (row < M && colbase < S) Yn[(size_t)row * S + colbase] = v0;
    if (row < M && colbase + 32 < S) Yn[(size_t)row * S + colbase + 32] = v1;
    if (STATS) {
      float s = v0 + v1;
      float q = fmaf(v0, v0, v1 * v1);
#pragma unroll
      for (int m = 16; m > 0; m >>= 1) {
        s += __shfl_xor(s, m, 64);
        q += __shfl_xor(q, m, 64);
      }
      if ((lane & 31) == 0 && row < M && partials) {
        const int NP = gridDim.z * gridDim.x, p = n * gridDim.x + blockIdx.x;
        partials[(size_t)row * NP + p] = (double)s;
        partials[((size_t)M + row) * NP + p] = (double)q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wave-autonomous variant (the default for channel counts that are multiples of 32).  A wave owns a 32-row x
// 32-column output tile over KW input channels and feeds the MFMA straight from registers:
//   B operand  lane (h, c) holds X[kb + h*KW/2 + kk][col0 + c], kk = 0..KW/2-1  - one coalesced dword load per kk
//   A operand  lane (h, c) holds W[m0 + c][kb + h*KW/2 + kk]                    - contiguous in k: 16-byte loads
//              (bwd-data reads W^T: element (m, k) = W[k*M + m], one coalesced dword load per kk)
// so MFMA step kk multiplies k = kb + kk (lanes 0-31) and k = kb + KW/2 + kk (lanes 32-63).  No LDS staging and no
// barrier before the MFMAs: every load of the wave is in flight at once (one memory round trip), and waves drift
// apart, so loads, MFMAs and stores of different waves overlap instead of marching in phases.
//   NWK == 1 (K = KW = 32 | 64): a workgroup is 4 waves = 4 neighbouring column tiles, MT row tiles each.
//   NWK  > 1 (K = NWK * 64):     a workgroup is one tile; wave j takes channels [64j, 64j+64); the NWK partial tiles
//                                meet in LDS and every wave finishes 32/NWK rows (sum order j = 0..NWK-1, fixed).
constexpr int WV_LD = 33;

template <int KW, int MT, bool AFFINE, bool STATS, bool TRANS_W, int NWK, int NT = 1>
__global__ __launch_bounds__(NWK == 1 ? 256 : NWK * 64) void pw_wave_kernel(
    const float* __restrict__ X, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ Wt, float* __restrict__ Y, double* __restrict__ partials, int M, int K, int S,
    msl::BnFold fold) {
  constexpr int KH = KW / 2;
  static_assert(NWK == 1 || (MT == 1 && NT == 1), "the K-split form owns one tile");
  __shared__ float red[NWK == 1 ? 2 * 4 * MT * 32 : NWK * 32 * WV_LD];
  __shared__ float f_sc[AFFINE ? FOLD_MAXK : 1], f_sh[AFFINE ? FOLD_MAXK : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
  const int n = blockIdx.z;
  const int m0 = blockIdx.y * 32 * MT;
  const int ctile = NWK == 1 ? (blockIdx.x * 4 + wv) * NT : blockIdx.x;  // NT neighbouring column tiles per wave
  const int kb = (NWK == 1 ? 0 : wv * KW) + h * KH;  // this lane's first input channel
  int col[NT];
  bool cin[NT];
  float xr[NT][KH], wr[MT][KH];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    col[nt] = (ctile + nt) * 32 + c;
    cin[nt] = col[nt] < S;
    const float* xp = X + ((size_t)n * K + kb) * S + (cin[nt] ? col[nt] : S - 1);
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) xr[nt][kk] = xp[(size_t)kk * S];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!TRANS_W) {
      const float4* wp = reinterpret_cast<const float4*>(Wt + (size_t)(m0 + mt * 32 + c) * K + kb);
#pragma unroll
      for (int j = 0; j < KH / 4; ++j) {
        const float4 t = wp[j];
        wr[mt][4 * j] = t.x; wr[mt][4 * j + 1] = t.y; wr[mt][4 * j + 2] = t.z; wr[mt][4 * j + 3] = t.w;
      }
    } else {
      const float* wp = Wt + (size_t)kb * M + m0 + mt * 32 + c;
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) wr[mt][kk] = wp[(size_t)kk * M];
    }
  }
  // all of the above leaves before anything below waits (the scheduler would otherwise trade the memory-level
  // parallelism for registers: load two, wait, MFMA, ...)
  __builtin_amdgcn_sched_barrier(0);
  if (AFFINE) {
    if (fold.partials) {  // optional in-kernel BatchNorm fold (Engine.fold_bn): same (scale, shift) bits as the vectors
      msl::bn_fold_block(fold, 0, K, f_sc, f_sh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) xr[nt][kk] = cin[nt] ? msl::act(xr[nt][kk], f_sc[kb + kk], f_sh[kb + kk]) : 0.f;
    } else {
      float sc[KH], sh[KH];
      const float4* sp = reinterpret_cast<const float4*>(in_scale + kb);
      const float4* hp = reinterpret_cast<const float4*>(in_shift + kb);
#pragma unroll
      for (int j = 0; j < KH / 4; ++j) {
        const float4 a = sp[j], b = hp[j];
        sc[4 * j] = a.x; sc[4 * j + 1] = a.y; sc[4 * j + 2] = a.z; sc[4 * j + 3] = a.w;
        sh[4 * j] = b.x; sh[4 * j + 1] = b.y; sh[4 * j + 2] = b.z; sh[4 * j + 3] = b.w;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) xr[nt][kk] = cin[nt] ? msl::act(xr[nt][kk], sc[kk], sh[kk]) : 0.f;
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) xr[nt][kk] = cin[nt] ? xr[nt][kk] : 0.f;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x16){0};
#pragma unroll
  for (int kk = 0; kk < KH; ++kk)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[mt][kk], xr[nt][kk], acc[mt][nt], 0, 0, 0);

  float* Yn = Y + (size_t)n * M * S;
  const int NP = gridDim.z * gridDim.x, p = n * gridDim.x + blockIdx.x;
  if (NWK == 1) {
    // D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = 0.f, v2 = 0.f;  // over this wave's NT column tiles
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float a = acc[mt][nt][r];
          if (cin[nt]) Yn[(size_t)(m0 + row) * S + col[nt]] = a;
          v += a;
          v2 = fmaf(a, a, v2);
        }
        if (STATS) {
          const float sm = msl::half32_sum(v), q = msl::half32_sum(v2);  // columns past S hold exact zeros
          if (c == msl::HALF32_SUM_LANE) {
            red[(wv * MT * 32 + row) * 2] = sm;
            red[(wv * MT * 32 + row) * 2 + 1] = q;
          }
        }
      }
    if (STATS) {
      __syncthreads();
      const int t = threadIdx.x;
      if (t < MT * 32 && partials) {
        double sm = 0.0, q = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          sm += (double)red[(w2 * MT * 32 + t) * 2];
          q += (double)red[(w2 * MT * 32 + t) * 2 + 1];
        }
        partials[(size_t)(m0 + t) * NP + p] = sm;
        partials[((size_t)M + m0 + t) * NP + p] = q;
      }
    }
  } else {
    float* mine = red + wv * 32 * WV_LD;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * WV_LD + c] = acc[0][0][r];
    __syncthreads();
    constexpr int RPW = 32 / NWK;  // rows finished by each wave (NWK = 2, 4, 8)
#pragma unroll
    for (int e = 0; e < RPW / 2; ++e) {
      const int row = wv * RPW + 2 * e + h;
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < NWK; ++j) v += red[(j * 32 + row) * WV_LD + c];
      if (cin[0]) Yn[(size_t)(m0 + row) * S + col[0]] = v;
      if (STATS) {
        const float sm = msl::half32_sum(v), q = msl::half32_sum(v * v);
        if (c == msl::HALF32_SUM_LANE && partials) {
          partials[(size_t)(m0 + row) * NP + p] = (double)sm;
          partials[((size_t)M + m0 + row) * NP + p] = (double)q;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Column-strip variant for the big layers (blocks 1-3: K <= 128, >= 128 strips).  A workgroup owns COLS consecutive
// positions of one image and ALL M output rows: the activated input strip [K][COLS] is read from HBM exactly once
// (coalesced 16-byte loads, affine + ReLU applied on the way into LDS), the weight rows of a wave stay in registers, and
// each of the 4 waves finishes its own 32-row x (32*TPW)-column part - one barrier in the whole kernel, no cross-wave
// reduction, statistics straight from the accumulators.  The wave-autonomous form above re-reads the input strip once
// per 32-row tile (M/32 times through L2) and needs K/64 waves to meet in LDS for every tile.
template <int K, int COLS, int MT, bool AFFINE, bool STATS, bool TRANS_W>
__global__ __launch_bounds__(256) void pw_strip_kernel(const float* __restrict__ X, const float* __restrict__ in_scale,
                                                       const float* __restrict__ in_shift, const float* __restrict__ Wt,
                                                       float* __restrict__ Y, double* __restrict__ partials, int S,
                                                       msl::BnFold fold) {
  constexpr int M = 32 * MT, CT = COLS / 32, TPW = MT * CT / 4, KH = K / 2;
  static_assert(TPW >= 1 && (TPW <= CT) && CT % TPW == 0, "a wave owns TPW column tiles of one row tile");
  constexpr int LPT = K * COLS / 4 / 256;  // float4 loads per thread for the strip
  constexpr int C4 = COLS / 4;             // float4 per strip row
  __shared__ __align__(16) float Xs[K * COLS];
  __shared__ float f_sc[AFFINE ? K : 1], f_sh[AFFINE ? K : 1];
  __shared__ float rs[(STATS && TPW != CT) ? 4 * 32 * 2 : 1];
  const int n = blockIdx.y, s0 = blockIdx.x * COLS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5, c = lane & 31;
  const int t0 = wv * TPW, mt = t0 / CT, ct0 = t0 % CT;
  const int m0 = mt * 32;
  const float* Xn = X + (size_t)n * K * S + s0;
  // the strip: thread -> rows r0 + i * (256 / C4), one float4 each
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v xv[LPT];
  const int r0 = tid / C4, c4 = (tid % C4) * 4;
#pragma unroll
  for (int i = 0; i < LPT; ++i) xv[i] = *reinterpret_cast<const f32x4v*>(Xn + (size_t)(r0 + i * (256 / C4)) * S + c4);
  // this wave's weight rows: lane (h, c) holds W[m0 + c][h*KH + kk]
  float wr[KH];
  if (!TRANS_W) {
    const f32x4v* wp = reinterpret_cast<const f32x4v*>(Wt + (size_t)(m0 + c) * K + h * KH);
#pragma unroll
    for (int j = 0; j < KH / 4; ++j) {
      const f32x4v t = wp[j];
      wr[4 * j] = t.x; wr[4 * j + 1] = t.y; wr[4 * j + 2] = t.z; wr[4 * j + 3] = t.w;
    }
  } else {
    const float* wp = Wt + (size_t)(h * KH) * M + m0 + c;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) wr[kk] = wp[(size_t)kk * M];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (AFFINE) {
    if (fold.partials) {
      msl::bn_fold_block(fold, 0, K, f_sc, f_sh);  // ends with a barrier
    } else {
      if (tid < K) {
        f_sc[tid] = in_scale[tid];
        f_sh[tid] = in_shift[tid];
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int r = r0 + i * (256 / C4);
    f32x4v v = xv[i];
    if (AFFINE) {
      const float sc = f_sc[r], sh = f_sh[r];
      v.x = msl::act(v.x, sc, sh); v.y = msl::act(v.y, sc, sh); v.z = msl::act(v.z, sc, sh); v.w = msl::act(v.w, sc, sh);
    }
    *reinterpret_cast<f32x4v*>(&Xs[r * COLS + c4]) = v;
  }
  __syncthreads();
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = (f32x16){0};
  const float* xb = Xs + (size_t)(h * KH) * COLS + ct0 * 32 + c;
#pragma unroll
  for (int kk = 0; kk < KH; ++kk)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[kk], xb[kk * COLS + t * 32], acc[t], 0, 0, 0);
  // D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* Yn = Y + (size_t)n * M * S + s0 + ct0 * 32 + c;
  const int NP = gridDim.y * gridDim.x, p = n * gridDim.x + blockIdx.x;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    float v = 0.f, v2 = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const float a = acc[t][r];
      Yn[(size_t)row * S + t * 32] = a;
      v += a;
      v2 = fmaf(a, a, v2);
    }
    if (STATS) {
      const float sm = msl::half32_sum(v), q = msl::half32_sum(v2);
      if (TPW == CT) {  // the wave covers the whole strip of its rows
        if (c == msl::HALF32_SUM_LANE && partials) {
          partials[(size_t)row * NP + p] = (double)sm;
          partials[((size_t)M + row) * NP + p] = (double)q;
        }
      } else if (c == msl::HALF32_SUM_LANE) {  // CT / TPW waves share a row tile: their row sums meet in LDS below
        rs[(wv * 32 + row - m0) * 2] = sm;
        rs[(wv * 32 + row - m0) * 2 + 1] = q;
      }
    }
  }
  if (STATS && TPW != CT) {
    constexpr int PW = CT / TPW;  // consecutive waves share a row tile
    __syncthreads();
    if (tid < M && partials) {
      const int w0 = (tid / 32) * PW, rl = tid % 32;
      double sm = 0.0, q = 0.0;
#pragma unroll
      for (int u = 0; u < PW; ++u) {
        sm += (double)rs[((w0 + u) * 32 + rl) * 2];
        q += (double)rs[((w0 + u) * 32 + rl) * 2 + 1];
      }
      partials[(size_t)tid * NP + p] = sm;
      partials[((size_t)M + tid) * NP + p] = q;
    }
  }
}

// which (K, M, S) take the column-strip kernel, and with which strip width (0: none): the forward and bwd-data GEMMs of
// blocks 1-3 (K = input channels of the GEMM, M = output rows)
static inline int strip_cols(int N, int K, int M, int S) {
  constexpr int on = 1;
  if (!on) return 0;
  int cols = 0;
  if (K == 32 && M == 64) cols = 256;
  else if (K == 64 && M == 32) cols = 128;
  else if ((K == 64 && M == 128) || (K == 128 && (M == 128 || M == 64))) cols = 64;
  // (32-column strips - two workgroups per CU - measured equal or slower: 8.2 -> 9.4 us and 11.1 -> 12.4 us on blocks 2 / 3)
  if (!cols || S % cols != 0 || (long long)N * (S / cols) < 128) return 0;
  return cols;
}

template <bool AFFINE, bool STATS, bool TRANS_W>
static int launch_strip(const float* X, const float* in_scale, const float* in_shift, const float* Wt, float* Y,
                        double* partials, int N, int M, int K, int S, int cols, const msl::BnFold& fold, hipStream_t st) {
  dim3 grid(S / cols, N);
#define MSL_ST(K_, C_, MT_) \
  MSL_LAUNCH((pw_strip_kernel<K_, C_, MT_, AFFINE, STATS, TRANS_W>), grid, dim3(256), 0, st, X, in_scale, in_shift, Wt, Y, partials, S, fold)
  if (K == 32 && M == 64) MSL_ST(32, 256, 2);
  else if (K == 64 && M == 32) MSL_ST(64, 128, 1);
  else if (K == 64 && M == 128) MSL_ST(64, 64, 4);
  else if (K == 128 && M == 128) MSL_ST(128, 64, 4);
  else if (K == 128 && M == 64) MSL_ST(128, 64, 2);
  else return MSL_ERR_UNSUPPORTED;
#undef MSL_ST
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// column tiles per wave of the unsplit form (K <= 64): two when that still leaves >= 2 waves per SIMD - the weight rows, the
// input affine and the statistics reduction are then paid once per 64 columns
static inline int wave_nt(int N, int K, int M, int S) {
  constexpr int force = 0;
  if (K > 64) return 1;
  if (force) return force;
  const int mt = (M % 64 == 0) ? 2 : 1;
  const long long waves = (long long)N * msl::cdiv(S, 32) * (M / (32 * mt));
  return waves >= 4096 ? 2 : 1;
}

// which form runs a (K, M) GEMM: 0 = the LDS-staged kernels above, else KW (NWK = K / KW)
static inline int wave_form(int K, int M) {
  if (M % 32 != 0) return 0;
  if (K == 32) return 32;
  if (K == 64 || K == 128 || K == 256 || K == 512) return 64;
  return 0;
}

template <bool AFFINE, bool STATS, bool TRANS_W>
static int launch_wave(const float* X, const float* in_scale, const float* in_shift, const float* Wt, float* Y,
                       double* partials, int N, int M, int K, int S, const msl::BnFold& fold, hipStream_t st) {
#define MSL_WV(KW_, MT_, NWK_)                                                                                        \
  MSL_LAUNCH((pw_wave_kernel<KW_, MT_, AFFINE, STATS, TRANS_W, NWK_>),                                        \
                     dim3(msl::cdiv(S, NWK_ == 1 ? 128 : 32), M / (32 * MT_), N), dim3(NWK_ == 1 ? 256 : NWK_ * 64), \
                     0, st, X, in_scale, in_shift, Wt, Y, partials, M, K, S, fold)
#define MSL_WV1(KW_, MT_)                                                                                              \
  do {                                                                                                                 \
    if (nt == 2)                                                                                                       \
      MSL_LAUNCH((pw_wave_kernel<KW_, MT_, AFFINE, STATS, TRANS_W, 1, 2>), dim3(msl::cdiv(S, 256), M / (32 * MT_), N), \
                         dim3(256), 0, st, X, in_scale, in_shift, Wt, Y, partials, M, K, S, fold);                     \
    else                                                                                                               \
      MSL_WV(KW_, MT_, 1);                                                                                             \
  } while (0)
  const bool two = M % 64 == 0;
  const int nt = STATS ? wave_nt(N, K, M, S) : 1;  // pays where the statistics epilogue is amortised (measured: tools/bench_pw.py)
  if (K == 32) {
    if (two) MSL_WV1(32, 2); else MSL_WV1(32, 1);
  } else if (K == 64) {
    if (two) MSL_WV1(64, 2); else MSL_WV1(64, 1);
  } else {
    switch (K / 64) {
      case 2: MSL_WV(64, 1, 2); break;
      case 4: MSL_WV(64, 1, 4); break;
      case 8: MSL_WV(64, 1, 8); break;
      default: return MSL_ERR_UNSUPPORTED;
    }
  }
#undef MSL_WV1
#undef MSL_WV
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------
// bwd-weight.  Block = (k-split, output tile 64 x BNN).  The 4 waves split every 64-position chunk four
// ways and each keeps the whole tile's accumulators; fixed-order in-block reduction, then one slab per block.
constexpr int PC = 64, PC_LD = PC + 1;

template <int BNN, bool AFFINE>
__global__ __launch_bounds__(256) void pw_bwd_weight_kernel(const float* __restrict__ dY,
                                                            const float* __restrict__ Z,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            float* __restrict__ slabs, int Cout, int Cin,
                                                            int S, int N, int chunks_per_img,
                                                            int chunks_per_block) {
  constexpr int NSUB_N = BNN / 32;
  constexpr int NSUB = 2 * NSUB_N;
  __shared__ __align__(16) float lds[(64 + BNN) * PC_LD > 64 * BNN ? (64 + BNN) * PC_LD : 64 * BNN];
  float* dys = lds;                 // [64][PC_LD]
  float* as_ = lds + 64 * PC_LD;    // [BNN][PC_LD]
  const int ks = blockIdx.x;
  const int tiles_n = Cin / BNN;
  const int tm = blockIdx.y / tiles_n, tn = blockIdx.y % tiles_n;
  const int m0 = tm * 64, n0 = tn * BNN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int total_chunks = N * chunks_per_img;
  const int ch_lo = ks * chunks_per_block, ch_hi = min(total_chunks, ch_lo + chunks_per_block);
  const bool vec_ok = (S & 3) == 0;

  f32x16 acc[NSUB];
#pragma unroll
  for (int i = 0; i < NSUB; ++i) acc[i] = (f32x16){0};

  // the BatchNorm affine of this thread's activation rows: loaded once, not once per chunk (inside the chunk loop the
  // compiler serialised them behind the tile loads: four extra memory round trips per 64-position chunk)
  float scv[BNN / 16], shv[BNN / 16];
#pragma unroll
  for (int j = 0; j < BNN / 16; ++j) {
    scv[j] = AFFINE ? in_scale[n0 + (tid >> 4) + j * 16] : 1.f;
    shv[j] = AFFINE ? in_shift[n0 + (tid >> 4) + j * 16] : 0.f;
  }

  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    const int n = ch / chunks_per_img, s0 = (ch % chunks_per_img) * PC;
    const float* dyn = dY + ((size_t)n * Cout + m0) * S;
    const float* zn = Z + ((size_t)n * Cin + n0) * S;
    float4 st[(64 + BNN) / 16];
    if (vec_ok) {
#pragma unroll
      for (int i = 0; i < (64 + BNN) / 16; ++i) {
        const int r = (tid >> 4) + i * 16;
        st[i] = msl::load4_zfill<true>(r < 64 ? dyn + (size_t)r * S : zn + (size_t)(r - 64) * S, s0 + (tid & 15) * 4, S);
      }
    } else {
#pragma unroll
      for (int i = 0; i < (64 + BNN) / 16; ++i) {
        const int r = (tid >> 4) + i * 16;
        st[i] = msl::load4_zfill<false>(r < 64 ? dyn + (size_t)r * S : zn + (size_t)(r - 64) * S, s0 + (tid & 15) * 4, S);
      }
    }
#pragma unroll
    for (int i = 0; i < (64 + BNN) / 16; ++i) {
      const int r = (tid >> 4) + i * 16, c4 = (tid & 15) * 4, col = s0 + c4;
      float4 v = st[i];
      if (AFFINE && r >= 64) {  // uniform per i: 64 % 16 == 0
        const float sc = scv[i >= 4 ? i - 4 : 0], sh = shv[i >= 4 ? i - 4 : 0];
        v.x = col < S ? msl::act(v.x, sc, sh) : 0.f;
        v.y = col + 1 < S ? msl::act(v.y, sc, sh) : 0.f;
        v.z = col + 2 < S ? msl::act(v.z, sc, sh) : 0.f;
        v.w = col + 3 < S ? msl::act(v.w, sc, sh) : 0.f;
      }
      float* dst = lds + r * PC_LD + c4;
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int pos = wv * 16 + 2 * kk + (lane >> 5);
      float a[2], b[NSUB_N];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = dys[(i * 32 + (lane & 31)) * PC_LD + pos];
#pragma unroll
      for (int j = 0; j < NSUB_N; ++j) b[j] = as_[(j * 32 + (lane & 31)) * PC_LD + pos];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSUB_N; ++j)
          acc[i * NSUB_N + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i * NSUB_N + j], 0, 0, 0);
    }
    __syncthreads();
  }

  // in-block reduction over the 4 waves, order 3,2,1,0 (fixed)
  float* redt = lds;  // [64][BNN]
  for (int w = 3; w >= 1; --w) {
    if (wv == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSUB_N; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = j * 32 + (lane & 31);
            const float v = acc[i * NSUB_N + j][r];
            if (w == 3) redt[row * BNN + col] = v;
            else redt[row * BNN + col] += v;
          }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float* out = slabs + (size_t)ks * Cout * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NSUB_N; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = j * 32 + (lane & 31);
          out[(size_t)(m0 + row) * Cin + n0 + col] = acc[i * NSUB_N + j][r] + redt[row * BNN + col];
        }
  }
}

// ---------------------------------------------------------------------------------------------
// bwd-weight, wave-autonomous form (the default for S % 32 == 0 and Cin % 32 == 0, Cout % 32 == 0).
//   dW[co][ci] = sum over positions of dY[co][p] * A[ci][p]:  positions are the MFMA K dimension.
// A wave owns a (32*MT) x 32 tile of dW and walks 32-position chunks of its workgroup's position range; both MFMA
// operands come straight from global memory into registers - lane (h, c) loads the 16 consecutive positions
// p0 + 16h .. p0 + 16h + 15 of row c (four 16-byte loads per operand row; a chunk of a row is exactly one 128-byte line),
// and MFMA step kk contracts positions p0 + kk (lanes 0-31) and p0 + 16 + kk (lanes 32-63).  The next chunk's loads
// are in flight during this chunk's MFMAs (two register sets); no LDS and no barrier in the loop, so the 4 waves of a
// workgroup drift apart and loads / MFMAs of different waves overlap.  The 4 private tiles meet in LDS once at the
// end (fixed order 0..3); the workgroup then stores its sum either straight into dW (the whole position range in one
// workgroup: tail layers, 256 positions) or into its slab of a [nslabs][Cout][Cin] workspace that the batched gradient
// reduction (optim.hip: grad_reduce_batch_kernel) sums in slab order.  No float atomics: run-to-run bit-identical.
template <int MT, bool AFFINE>
__device__ __forceinline__ void pw_bww_wave_body(const float* __restrict__ dY, const float* __restrict__ Z,
                                                 const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                 float* __restrict__ out, int Cout, int Cin, int S, int chunks_per_img,
                                                 int total_chunks, int chunks_per_block, int ks, int tile) {
  __shared__ __align__(16) float red[4 * MT * 1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
  const int tiles_n = Cin / 32;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * 32 * MT, n0 = tn * 32;
  const int ch_lo = ks * chunks_per_block, ch_hi = min(total_chunks, ch_lo + chunks_per_block);
  const int mine = max(0, (ch_hi - ch_lo - wv + 3) / 4);  // chunks ch_lo + wv, + 4, ...

  float sc = 1.f, sh = 0.f;
  if (AFFINE) {
    sc = in_scale[n0 + c];
    sh = in_shift[n0 + c];
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x16){0};

  float4 a0[MT][4], b0[4], a1[MT][4], b1[4];
  // chunk index -> row pointers; indices past the range are clamped (their products are skipped below)
  auto issue = [&](int i, float4 (&a)[MT][4], float4 (&b)[4]) {
    const int ch = min(ch_lo + wv + 4 * i, total_chunks - 1);
    const int n = ch / chunks_per_img, s0 = (ch - n * chunks_per_img) * 32 + 16 * h;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float4* p = reinterpret_cast<const float4*>(dY + ((size_t)n * Cout + m0 + mt * 32 + c) * S + s0);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[mt][j] = p[j];
    }
    const float4* q = reinterpret_cast<const float4*>(Z + ((size_t)n * Cin + n0 + c) * S + s0);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = q[j];
  };
  auto mac = [&](float4 (&a)[MT][4], float4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float bv[4] = {b[j].x, b[j].y, b[j].z, b[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = AFFINE ? msl::act(bv[e], sc, sh) : bv[e];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float av = e == 0 ? a[mt][j].x : e == 1 ? a[mt][j].y : e == 2 ? a[mt][j].z : a[mt][j].w;
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x, acc[mt], 0, 0, 0);
        }
      }
    }
  };

  if (mine > 0) issue(0, a0, b0);
#pragma unroll 1
  for (int i = 0; i < mine; i += 2) {
    issue(i + 1, a1, b1);  // clamped: always a valid address
    __builtin_amdgcn_sched_barrier(0);
    mac(a0, b0);
    if (i + 1 < mine) {
      issue(i + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mac(a1, b1);
    }
  }

  // the 4 waves' tiles meet in LDS: red[wv][mt][row][col]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wv * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[mt][r];
  __syncthreads();
  float* dst = out + (size_t)ks * Cout * Cin;
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int q4 = threadIdx.x + e * 256;        // float4 index inside the (32*MT) x 32 tile
    const int row = q4 >> 3, col = (q4 & 7) * 4;  // row = mt*32 + r
    float4 v = *reinterpret_cast<const float4*>(&red[(0 * MT * 32 + row) * 32 + col]);
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const float4 t = *reinterpret_cast<const float4*>(&red[(w2 * MT * 32 + row) * 32 + col]);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    *reinterpret_cast<float4*>(dst + (size_t)(m0 + row) * Cin + n0 + col) = v;
  }
}

template <int MT, bool AFFINE>
__global__ __launch_bounds__(256) void pw_bww_wave_kernel(const float* __restrict__ dY, const float* __restrict__ Z,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, float* __restrict__ out,
                                                          int Cout, int Cin, int S, int chunks_per_img, int total_chunks,
                                                          int chunks_per_block) {
  pw_bww_wave_body<MT, AFFINE>(dY, Z, in_scale, in_shift, out, Cout, Cin, S, chunks_per_img, total_chunks, chunks_per_block,
                               blockIdx.x, blockIdx.y);
}

// The tail of the network (blocks 4-7 at 128^3: 2048 / 256 positions per batch) gives this kernel 64-256 workgroups of
// two to eight chunks each: four launches of ~8 us that are all latency.  ONE launch runs up to four such layers side by
// side: workgroup b belongs to the layer k with first[k] <= b < first[k+1] and is that layer's workgroup
// (ks, tile) = ((b - first[k]) % ksplit[k], (b - first[k]) / ksplit[k]); same arithmetic, same slabs, same bits.
constexpr int PW_BWW_BATCH_MAX = 4;
struct PwBwwBatch {
  const float* dY[PW_BWW_BATCH_MAX];
  const float* Z[PW_BWW_BATCH_MAX];
  const float* in_scale[PW_BWW_BATCH_MAX];
  const float* in_shift[PW_BWW_BATCH_MAX];
  float* out[PW_BWW_BATCH_MAX];
  int Cout[PW_BWW_BATCH_MAX], Cin[PW_BWW_BATCH_MAX], S[PW_BWW_BATCH_MAX], chunks_per_img[PW_BWW_BATCH_MAX],
      total_chunks[PW_BWW_BATCH_MAX], chunks_per_block[PW_BWW_BATCH_MAX], ksplit[PW_BWW_BATCH_MAX],
      first[PW_BWW_BATCH_MAX + 1];
};

__global__ __launch_bounds__(256) void pw_bww_wave_batch_kernel(PwBwwBatch b, int n) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < PW_BWW_BATCH_MAX; ++j)
    if (j < n && (int)blockIdx.x >= b.first[j]) k = j;
  const int local = blockIdx.x - b.first[k];
  pw_bww_wave_body<2, true>(b.dY[k], b.Z[k], b.in_scale[k], b.in_shift[k], b.out[k], b.Cout[k], b.Cin[k], b.S[k],
                            b.chunks_per_img[k], b.total_chunks[k], b.chunks_per_block[k], local % b.ksplit[k],
                            local / b.ksplit[k]);
}

struct BwwWavePlan {
  int mt, tiles, chunks_per_img, total_chunks, ksplit, chunks_per_block;
};

// Supported: whole 32-position chunks and 32-channel tiles.  nslabs == 1 -> the kernel writes dW itself.
static inline bool bww_wave_plan(int N, int Cin, int Cout, int S, BwwWavePlan& p) {
  constexpr int off = 1;
  if (!off || S % 32 != 0 || Cin % 32 != 0 || Cout % 32 != 0) return false;
  p.mt = (Cout % 64 == 0) ? 2 : 1;
  p.tiles = (Cout / (32 * p.mt)) * (Cin / 32);
  p.chunks_per_img = S / 32;
  p.total_chunks = N * p.chunks_per_img;
  // position split: about one workgroup per CU in total, at least 8 chunks (2 per wave) each - every split costs a
  // Cout x Cin slab written and read back
  constexpr int target = 256;
  int ks = std::max(1, std::min(target / std::max(1, p.tiles), p.total_chunks / 8));
  p.chunks_per_block = msl::cdiv(p.total_chunks, ks);
  p.ksplit = msl::cdiv(p.total_chunks, p.chunks_per_block);
  return true;
}

// out[i] = sum_k slabs[k][i], k ascending (fixed order)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                          int count, int nslabs) {
  __shared__ float lds[8 * 33];
  const float t = msl::reduce_slabs_256(slabs, (size_t)count, count, nslabs, lds);
  const int i = blockIdx.x * 32 + (threadIdx.x & 31);
  if ((threadIdx.x >> 5) == 0 && i < count) out[i] = t;
}

struct BwPlan {
  int bnn, ksplit, chunks_per_img, chunks_per_block;
};

BwPlan bw_plan(int N, int Cin, int Cout, int S) {
  BwPlan p;
  p.bnn = (Cin % 64 == 0) ? 64 : 32;
  p.chunks_per_img = msl::cdiv(S, PC);
  const int total = N * p.chunks_per_img;
  const int tiles = (Cout / 64) * (Cin / p.bnn);
  // K split: enough workgroups to matter, but at least 8 position chunks each - every split costs a Cout x Cin slab
  // written and read back (a 256-way split of block 2 moved 8 MB of slabs for 12 MB of operands), and this kernel
  // runs on the weight-gradient stream beside the dependency chain, where fewer, longer workgroups interfere less
  int ks = std::max(1, std::min(total, 512 / std::max(1, tiles)));
  constexpr int min_cpb = 1;
  ks = std::max(1, std::min(ks, total / std::max(1, min_cpb)));
  p.chunks_per_block = msl::cdiv(total, ks);
  p.ksplit = msl::cdiv(total, p.chunks_per_block);
  return p;
}

}  // namespace

extern "C" {

// K-split pays when the plain 64 x 128 tiling cannot fill the chip (tail layers); with >= 256 plain tiles the
// serial-K kernel wins (no cross-wave reduction).
static inline bool use_ksplit(int K, int M, int S, int N) {
  return K >= 128 && K % 128 == 0 && msl::cdiv(S, BN) * msl::cdiv(M, BM) * N < 256;
}

int msl_pwconv_fwd_num_partials(int N, int Cin, int Cout, int S) {
  if (const int cols = strip_cols(N, Cin, Cout, S)) return N * (S / cols);
  if (wave_form(Cin, Cout)) return N * msl::cdiv(S, Cin <= 64 ? 128 * wave_nt(N, Cin, Cout, S) : 32);
  return N * msl::cdiv(S, use_ksplit(Cin, Cout, S, N) ? KS_BN : BN);
}

// z (N,Cin,S) raw + input affine -> y (N,Cout,S) raw + stat partials [2][Cout][NP]
static const msl::BnFold pw_nofold{nullptr, 0, 0, 1.0, nullptr, nullptr, 0.f};

static int pwconv_fwd_impl(const float* z, const float* in_scale, const float* in_shift, const msl::BnFold& fold,
                           const float* w, float* y, double* partials, int N, int Cin, int Cout, int S, void* stream) {
  if (N <= 0 || S <= 0 || Cin % BK != 0 || Cout % 4 != 0) return MSL_ERR_ARG;
  if ((in_scale || fold.partials) && Cin > FOLD_MAXK) return MSL_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (const int cols = strip_cols(N, Cin, Cout, S)) {
    if (in_scale || fold.partials) {
      if (partials) return launch_strip<true, true, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, cols, fold, st);
      return launch_strip<true, false, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, cols, fold, st);
    }
    if (partials) return launch_strip<false, true, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, cols, fold, st);
    return launch_strip<false, false, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, cols, fold, st);
  }
  if (wave_form(Cin, Cout)) {
    if (in_scale || fold.partials) {
      if (partials) return launch_wave<true, true, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, fold, st);
      return launch_wave<true, false, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, fold, st);
    }
    if (partials) return launch_wave<false, true, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, fold, st);
    return launch_wave<false, false, false>(z, in_scale, in_shift, w, y, partials, N, Cout, Cin, S, fold, st);
  }
  if (use_ksplit(Cin, Cout, S, N)) {
    dim3 g2(msl::cdiv(S, KS_BN), msl::cdiv(Cout, KS_BM), N);
#define MSL_KS(A_, S_, T_, X_, Wp_, Y_, P_, M_, K_)                                                                  \
  do {                                                                                                             \
    if ((K_) % 256 == 0) {                                                                                         \
      const size_t lds_ = (size_t)8 * KS_WAVE_LDS * sizeof(float);                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_ksplit_kernel<A_, S_, T_, 8>),    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);                  \
      if (e_ != hipSuccess) return (int)e_;                                                                        \
      MSL_LAUNCH((pw_gemm_ksplit_kernel<A_, S_, T_, 8>), g2, dim3(512), lds_, st, X_, in_scale, in_shift,  \
                         Wp_, Y_, P_, M_, K_, S, fold);                                                            \
    } else {                                                                                                       \
      MSL_LAUNCH((pw_gemm_ksplit_kernel<A_, S_, T_, 4>), g2, dim3(256),                                    \
                         (size_t)4 * KS_WAVE_LDS * sizeof(float), st, X_, in_scale, in_shift, Wp_, Y_, P_, M_, K_, S, fold); \
    }                                                                                                              \
  } while (0)
    if (in_scale || fold.partials) {
      if (partials) MSL_KS(true, true, false, z, w, y, partials, Cout, Cin);
      else MSL_KS(true, false, false, z, w, y, partials, Cout, Cin);
    } else {
      if (partials) MSL_KS(false, true, false, z, w, y, partials, Cout, Cin);
      else MSL_KS(false, false, false, z, w, y, partials, Cout, Cin);
    }
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(msl::cdiv(S, BN), msl::cdiv(Cout, BM), N);
  if (in_scale || fold.partials) {
    if (partials) MSL_LAUNCH((pw_gemm_kernel<true, true, false>), grid, dim3(256), 0, st, z, in_scale, in_shift, w, y, partials, Cout, Cin, S, fold);
    else MSL_LAUNCH((pw_gemm_kernel<true, false, false>), grid, dim3(256), 0, st, z, in_scale, in_shift, w, y, partials, Cout, Cin, S, fold);
  } else {
    if (partials) MSL_LAUNCH((pw_gemm_kernel<false, true, false>), grid, dim3(256), 0, st, z, in_scale, in_shift, w, y, partials, Cout, Cin, S, fold);
    else MSL_LAUNCH((pw_gemm_kernel<false, false, false>), grid, dim3(256), 0, st, z, in_scale, in_shift, w, y, partials, Cout, Cin, S, fold);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_pwconv_fwd(const float* z, const float* in_scale, const float* in_shift, const float* w, float* y,
                   double* partials, int N, int Cin, int Cout, int S, void* stream) {
  return pwconv_fwd_impl(z, in_scale, in_shift, pw_nofold, w, y, partials, N, Cin, Cout, S, stream);
}

// Same, with the input's BatchNorm affine folded in the kernel prologue from the producer's partials.
int msl_pwconv_fwd_fold(const float* z, const double* in_partials, int in_np, double in_count, const float* gamma,
                        const float* beta, float eps, const float* w, float* y, double* partials, int N, int Cin,
                        int Cout, int S, void* stream) {
  if (!in_partials || in_np <= 0) return MSL_ERR_ARG;
  const msl::BnFold fold{in_partials, in_np, Cin, in_count, gamma, beta, eps};
  return pwconv_fwd_impl(z, nullptr, nullptr, fold, w, y, partials, N, Cin, Cout, S, stream);
}

// dy (N,Cout,S) -> g_in (N,Cin,S) = W^T . dy
int msl_pwconv_bwd_data(const float* dy, const float* w, float* g_in, int N, int Cin, int Cout, int S,
                        void* stream) {
  if (N <= 0 || S <= 0 || Cout % BK != 0 || Cin % 4 != 0) return MSL_ERR_ARG;
  if (const int cols = strip_cols(N, Cout, Cin, S))
    return launch_strip<false, false, true>(dy, nullptr, nullptr, w, g_in, nullptr, N, Cin, Cout, S, cols, pw_nofold,
                                            (hipStream_t)stream);
  if (wave_form(Cout, Cin))
    return launch_wave<false, false, true>(dy, nullptr, nullptr, w, g_in, nullptr, N, Cin, Cout, S, pw_nofold,
                                           (hipStream_t)stream);
  if (use_ksplit(Cout, Cin, S, N)) {
    dim3 g2(msl::cdiv(S, KS_BN), msl::cdiv(Cin, KS_BM), N);
    hipStream_t st = (hipStream_t)stream;
    const float *in_scale = nullptr, *in_shift = nullptr;
    double* nopart = nullptr;
    const msl::BnFold fold = pw_nofold;
    MSL_KS(false, false, true, dy, w, g_in, nopart, Cin, Cout);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  dim3 grid(msl::cdiv(S, BN), msl::cdiv(Cin, BM), N);
  MSL_LAUNCH((pw_gemm_kernel<false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, dy, nullptr,
                     nullptr, w, g_in, nullptr, Cin, Cout, S, pw_nofold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

size_t msl_pwconv_bwd_weight_workspace_bytes(int N, int Cin, int Cout, int S) {
  BwwWavePlan wp;
  if (bww_wave_plan(N, Cin, Cout, S, wp)) return (size_t)wp.ksplit * Cout * Cin * sizeof(float);
  BwPlan p = bw_plan(N, Cin, Cout, S);
  return (size_t)p.ksplit * Cout * Cin * sizeof(float);
}

// Number of [Cout][Cin] slabs msl_pwconv_bwd_weight_slabs writes for this shape (1: the result itself).
int msl_pwconv_bwd_weight_nslabs(int N, int Cin, int Cout, int S) {
  BwwWavePlan wp;
  if (bww_wave_plan(N, Cin, Cout, S, wp)) return wp.ksplit;
  return bw_plan(N, Cin, Cout, S).ksplit;
}

// Partial weight gradients: out = [nslabs][Cout][Cin] (nslabs from msl_pwconv_bwd_weight_nslabs); their sum in slab
// order is dW.  With nslabs == 1 `out` may be the gradient tensor itself.
int msl_pwconv_bwd_weight_slabs(const float* dy, const float* z, const float* in_scale, const float* in_shift,
                                float* out, int N, int Cin, int Cout, int S, void* stream) {
  if (N <= 0 || S <= 0 || Cin % 32 != 0 || Cout % 32 != 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  BwwWavePlan wp;
  if (bww_wave_plan(N, Cin, Cout, S, wp)) {
    dim3 grid(wp.ksplit, wp.tiles);
#define MSL_BWW(MT_, A_)                                                                                          \
  MSL_LAUNCH((pw_bww_wave_kernel<MT_, A_>), grid, dim3(256), 0, st, dy, z, in_scale, in_shift, out, Cout, Cin, \
                     S, wp.chunks_per_img, wp.total_chunks, wp.chunks_per_block)
    if (wp.mt == 2) {
      if (in_scale) MSL_BWW(2, true); else MSL_BWW(2, false);
    } else {
      if (in_scale) MSL_BWW(1, true); else MSL_BWW(1, false);
    }
#undef MSL_BWW
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  if (Cout % 64 != 0) return MSL_ERR_ARG;
  BwPlan p = bw_plan(N, Cin, Cout, S);
  dim3 grid(p.ksplit, (Cout / 64) * (Cin / p.bnn));
  if (p.bnn == 64) {
    if (in_scale) MSL_LAUNCH((pw_bwd_weight_kernel<64, true>), grid, dim3(256), 0, st, dy, z, in_scale, in_shift, out, Cout, Cin, S, N, p.chunks_per_img, p.chunks_per_block);
    else MSL_LAUNCH((pw_bwd_weight_kernel<64, false>), grid, dim3(256), 0, st, dy, z, in_scale, in_shift, out, Cout, Cin, S, N, p.chunks_per_img, p.chunks_per_block);
  } else {
    if (in_scale) MSL_LAUNCH((pw_bwd_weight_kernel<32, true>), grid, dim3(256), 0, st, dy, z, in_scale, in_shift, out, Cout, Cin, S, N, p.chunks_per_img, p.chunks_per_block);
    else MSL_LAUNCH((pw_bwd_weight_kernel<32, false>), grid, dim3(256), 0, st, dy, z, in_scale, in_shift, out, Cout, Cin, S, N, p.chunks_per_img, p.chunks_per_block);
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// 1 when msl_pwconv_bwd_weight_slabs_batch takes this shape (the wave-autonomous form with 64-row tiles), else 0
int msl_pwconv_bwd_weight_batchable(int N, int Cin, int Cout, int S) {
  BwwWavePlan wp;
  return (N > 0 && S > 0 && bww_wave_plan(N, Cin, Cout, S, wp) && wp.mt == 2) ? 1 : 0;
}

// msl_pwconv_bwd_weight_slabs for n <= 4 layers in ONE launch (host arrays of n entries each; every layer must be
// msl_pwconv_bwd_weight_batchable and carries its input affine).  out[k]: as msl_pwconv_bwd_weight_slabs for layer k.
int msl_pwconv_bwd_weight_slabs_batch(const float* const* dy, const float* const* z, const float* const* in_scale,
                                      const float* const* in_shift, float* const* out, const int* Cin, const int* Cout,
                                      const int* S, int n, int N, void* stream) {
  if (n < 1 || n > PW_BWW_BATCH_MAX || N <= 0) return MSL_ERR_ARG;
  PwBwwBatch b;
  int total = 0;
  for (int k = 0; k < n; ++k) {
    BwwWavePlan wp;
    if (!dy[k] || !z[k] || !in_scale[k] || !in_shift[k] || !out[k]) return MSL_ERR_ARG;
    if (S[k] <= 0 || !bww_wave_plan(N, Cin[k], Cout[k], S[k], wp) || wp.mt != 2) return MSL_ERR_UNSUPPORTED;
    b.dY[k] = dy[k]; b.Z[k] = z[k]; b.in_scale[k] = in_scale[k]; b.in_shift[k] = in_shift[k]; b.out[k] = out[k];
    b.Cout[k] = Cout[k]; b.Cin[k] = Cin[k]; b.S[k] = S[k];
    b.chunks_per_img[k] = wp.chunks_per_img; b.total_chunks[k] = wp.total_chunks;
    b.chunks_per_block[k] = wp.chunks_per_block; b.ksplit[k] = wp.ksplit;
    b.first[k] = total;
    total += wp.ksplit * wp.tiles;
  }
  for (int k = n; k <= PW_BWW_BATCH_MAX; ++k) b.first[k] = total;
  for (int k = n; k < PW_BWW_BATCH_MAX; ++k) {
    b.dY[k] = b.Z[k] = b.in_scale[k] = b.in_shift[k] = nullptr; b.out[k] = nullptr;
    b.Cout[k] = b.Cin[k] = b.S[k] = b.chunks_per_img[k] = b.total_chunks[k] = b.chunks_per_block[k] = 0; b.ksplit[k] = 1;
  }
  MSL_LAUNCH(pw_bww_wave_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b, n);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// dW (Cout,Cin) = sum_{n,s} dy[n,co,s] * relu(bn(z))[n,ci,s]   (stand-alone form: partial slabs + their reduction; the
// training step uses msl_pwconv_bwd_weight_slabs and reduces the slabs of every layer in one msl_grad_reduce_batch)
int msl_pwconv_bwd_weight(const float* dy, const float* z, const float* in_scale, const float* in_shift,
                          float* dw, float* workspace, int N, int Cin, int Cout, int S, void* stream) {
  const int nslabs = msl_pwconv_bwd_weight_nslabs(N, Cin, Cout, S);
  if (nslabs == 1) return msl_pwconv_bwd_weight_slabs(dy, z, in_scale, in_shift, dw, N, Cin, Cout, S, stream);
  const int rc = msl_pwconv_bwd_weight_slabs(dy, z, in_scale, in_shift, workspace, N, Cin, Cout, S, stream);
  if (rc != MSL_OK) return rc;
  const int count = Cout * Cin;
  MSL_LAUNCH(slab_reduce_kernel, dim3(msl::cdiv(count, 32)), dim3(256), 0, (hipStream_t)stream, workspace, dw,
                     count, nslabs);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
