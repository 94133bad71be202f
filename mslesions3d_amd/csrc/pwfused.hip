// Pointwise backward of a big early block (block 1 at 128^3 x 4: Cout = 64, Cin = 32, 131 072 positions) in ONE pass over
// its operands (reference: autograd of lesions3d/mobilenet.py:45-47  out = relu(bn2(conv2(relu(bn1(z)))))):
//
//   dL/d relu(bn2(y))  --BN2+ReLU backward, applied while loading-->  dL/dy   (never stored)
//   dL/d relu(bn1(z)) = W^T . dL/dy                                    -> g_z   (bwd-data GEMM, 32x32x2 fp32 MFMA)
//   BatchNorm1-backward sums of z (sum gm, sum gm*xhat per channel)     -> fp64 partials, one per workgroup
//   dW = dL/dy . relu(bn1(z))^T                                         -> one [Cout][Cin] slab per workgroup
//
// It replaces four launches that each re-read one of these tensors from HBM - bn_relu_bwd_apply (reads g + y, writes
// dL/dy: 100 MB), the bwd-data strip GEMM (reads dL/dy), bn_relu_bwd_reduce (reads g_z + z) on the dependency chain and the
// pointwise weight gradient (reads dL/dy + z again) on a side stream: 84 MB read + 17 MB written instead of ~270 MB.
// A workgroup walks strips of COLS positions of one image: the applied gradient strip [K][COLS], the activation strip
// [M][COLS] and the raw z strip live in LDS (odd pitch: every MFMA operand read - rows as lanes for the weight gradient,
// columns as lanes for the data gradient - is conflict-free); the weight-gradient tile and the statistics stay in registers
// across the workgroup's strips.  HBM-bound; MFMA only for the two 1x1x1 contractions.  No atomics: fixed summation order.
#include "common.hpp"
#include "../../include/mslesions3d_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// One role's walk over the workgroup's strips (DATA: waves 0-3 = data gradient + BatchNorm1 sums, one 32-column tile each;
// !DATA: waves 4-7 = weight gradient, row tile x column part).  Both roles stage the strip together and meet at the same
// barriers; being separate instantiations, each role's registers are allocated for its own contraction only.
template <int K, int M, int COLS, bool DATA>
__device__ __forceinline__ void pw_bwd_fused_role(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ Wt, const float* __restrict__ z,
    float* __restrict__ gz, double* __restrict__ zpart, float* __restrict__ slabs, int S, int strips_per_img, int total_strips,
    float* lds, const float* coef, const float* cz) {
  constexpr int NT = 512;
  constexpr int P = COLS + 1, KH = K / 2, C4 = COLS / 4, RPP = NT / C4;  // LDS pitch; rows per load pass
  constexpr int GL = K / RPP, ZL = M / RPP;                               // float4 loads per thread: g / y rows, z rows
  constexpr int KT = K / 32, CPART = 4 / KT, CW = COLS / CPART;           // weight gradient: row tiles, column parts
  float* dys = lds;
  float* a1s = dys + K * P;
  float* zs = a1s + M * P;
  const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), h = lane >> 5, c = lane & 31;
  const int r0 = tid / C4, c4 = (tid % C4) * 4, rot = (tid >> 3) & 3;
  const int wt = wq % KT, wp = wq / KT;

  float wr[DATA ? KH : 1];  // this lane's weights for the data gradient: W^T element (m = c, k = h*KH + kk) = W[k][m]
  if (DATA) {
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) wr[kk] = Wt[(size_t)(h * KH + kk) * M + c];
  }
  f32x16 accw = {0};
  float s1r[DATA ? 16 : 1], s2r[DATA ? 16 : 1];
#pragma unroll
  for (int r = 0; r < (DATA ? 16 : 1); ++r) s1r[r] = s2r[r] = 0.f;

  f32x4v gv[GL], yv[GL], zv[ZL];
  // the loads of a strip (indices past the range are clamped: their values are never staged)
  auto issue = [&](int st) {
    const int sc_ = min(st, total_strips - 1);
    const int n = sc_ / strips_per_img, s0 = (sc_ - n * strips_per_img) * COLS;
    const float* gn = g + (size_t)n * K * S + s0;
    const float* yn = y + (size_t)n * K * S + s0;
    const float* zn = z + (size_t)n * M * S + s0;
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      gv[i] = *reinterpret_cast<const f32x4v*>(gn + (size_t)(r0 + i * RPP) * S + c4);
      yv[i] = *reinterpret_cast<const f32x4v*>(yn + (size_t)(r0 + i * RPP) * S + c4);
    }
#pragma unroll
    for (int i = 0; i < ZL; ++i) zv[i] = *reinterpret_cast<const f32x4v*>(zn + (size_t)(r0 + i * RPP) * S + c4);
  };
  issue(blockIdx.x);
  for (int st = blockIdx.x; st < total_strips; st += gridDim.x) {
    const int n = st / strips_per_img, s0 = (st - n * strips_per_img) * COLS;
    __syncthreads();  // the previous strip's LDS reads are done (first pass: the coefficients are published)
    // (odd pitch, 4-byte stores: lane l of a row writes column 4l + e - the 32 lanes of a row would hit 8 banks 4 times; with the
    //  element order rotated by l / 8 they hit 32 different banks)
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int r = r0 + i * RPP;
      const float sc = coef[r], sh = coef[K + r], cC = coef[2 * K + r], cE = coef[3 * K + r];
      float res[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gm = fmaf(yv[i][e], sc, sh) > 0.f ? gv[i][e] : 0.f;
        res[e] = fmaf(sc, gm, fmaf(cC, yv[i][e], cE));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ee = (e + rot) & 3;
        dys[r * P + c4 + ee] = ee == 0 ? res[0] : ee == 1 ? res[1] : ee == 2 ? res[2] : res[3];
      }
    }
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
      const int r = r0 + i * RPP;
      const float sc = cz[r], sh = cz[M + r];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ee = (e + rot) & 3;
        const float zz = ee == 0 ? zv[i][0] : ee == 1 ? zv[i][1] : ee == 2 ? zv[i][2] : zv[i][3];
        zs[r * P + c4 + ee] = zz;
        a1s[r * P + c4 + ee] = msl::act(zz, sc, sh);
      }
    }
    __syncthreads();
    // the staging registers are free again: the NEXT strip's loads are in flight during this strip's MFMAs
    issue(st + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    if (DATA) {
      // ---- data gradient: g_z[m][col] = sum_k W[k][m] * dL/dy[k][col]; wave wq owns columns [32 wq, 32 wq + 32)
      f32x16 acc = {0};
      const float* xb = dys + (h * KH) * P + wq * 32 + c;
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[kk], xb[kk * P], acc, 0, 0, 0);
      float* gzn = gz + (size_t)n * M * S + s0 + wq * 32 + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float a = acc[r];
        gzn[(size_t)row * S] = a;
        // BatchNorm1-backward sums of z (mobilenet.py:45): gm = a where relu(bn1(z)) > 0, xhat = (z - mean) * invstd
        const float zz = zs[row * P + wq * 32 + c];
        const float gm = fmaf(zz, cz[row], cz[M + row]) > 0.f ? a : 0.f;
        s1r[r] += gm;
        s2r[r] = fmaf(gm, (zz - cz[2 * M + row]) * cz[3 * M + row], s2r[r]);
      }
    } else {
      // ---- weight gradient: dW[co][ci] += sum_col dL/dy[co][col] * relu(bn1(z))[ci][col]; columns are the MFMA k axis
      const float* ap = dys + (wt * 32 + c) * P + wp * CW + h;
      const float* bp = a1s + c * P + wp * CW + h;
#pragma unroll
      for (int kk = 0; kk < CW / 2; ++kk) accw = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bp[2 * kk], accw, 0, 0, 0);
    }
  }

  // ---- epilogue: statistics partials (one per workgroup and channel) and the workgroup's weight-gradient slab
  __syncthreads();
  float* red = lds;  // [4 data waves][32 rows][2]
  if (DATA) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sm = msl::half32_sum(s1r[r]), q = msl::half32_sum(s2r[r]);
      if (c == msl::HALF32_SUM_LANE) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[(wq * 32 + row) * 2] = sm;
        red[(wq * 32 + row) * 2 + 1] = q;
      }
    }
  }
  __syncthreads();
  const int NP = gridDim.x;
  if (DATA && tid < M) {
    double sm = 0.0, q = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sm += (double)red[(u * 32 + tid) * 2];
      q += (double)red[(u * 32 + tid) * 2 + 1];
    }
    zpart[(size_t)tid * NP + blockIdx.x] = sm;
    zpart[((size_t)M + tid) * NP + blockIdx.x] = q;
  }
  __syncthreads();
  // the CPART column parts of a row tile meet in LDS (fixed order 0 .. CPART-1): wred[weight wave][row][ci]
  float* wred = lds;
  if (!DATA) {
#pragma unroll
    for (int r = 0; r < 16; ++r) wred[(wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = accw[r];
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * K * M;
  for (int e = tid; e < K * M; e += NT) {
    const int co = e / M, ci = e % M, t = co / 32, row = co % 32;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < CPART; ++u) v += wred[((u * KT + t) * 32 + row) * 32 + ci];
    slab[e] = v;
  }
}

template <int K, int M, int COLS>
__global__ __launch_bounds__(512) void pw_bwd_fused_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ vy,
    const double* __restrict__ ypart, int ynp, double ycount, float* __restrict__ dgamma_y, float* __restrict__ dbeta_y,
    const float* __restrict__ Wt, const float* __restrict__ z, const float* __restrict__ vz, float* __restrict__ gz,
    double* __restrict__ zpart, float* __restrict__ slabs, int S, int strips_per_img, int total_strips) {
  static_assert(M == 32 && K % 32 == 0 && COLS == 128 && 4 % (K / 32) == 0, "one 32-row output tile; four waves x 32 columns");
  constexpr int NT = 512, P = COLS + 1;
  extern __shared__ __align__(16) float lds[];
  float* coef = lds + (K + 2 * M) * P;  // [4][K]  scale, shift, cC, cE of bn2;  then [4][M] scale, shift, mean, invstd of bn1
  float* cz = coef + 4 * K;
  const int tid = threadIdx.x;
  // ---- prologue: BatchNorm2-backward coefficients from the producer's partials (msl_bn_bwd_finalize_coef's arithmetic):
  //      dL/dy = scale * gm + (cC * y + cE),  gm = dL/da where relu(bn2(y)) > 0
  {
    // eight threads per channel, four independent loads each in flight (a serial walk over the partials is ynp dependent
    // memory round trips); the parts are combined by DPP in a fixed order
    static_assert(K * 8 == NT, "eight threads per BatchNorm2 channel");
    const int ch = tid >> 3, part = tid & 7;
    const double* ps = ypart + (size_t)ch * ynp;
    const double* pq = ypart + ((size_t)K + ch) * ynp;
    double s = 0.0, q = 0.0;
    for (int p0 = part; p0 < ynp; p0 += 32) {
      double sv[4], qv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = min(p0 + 8 * u, ynp - 1);
        sv[u] = ps[pc];
        qv[u] = pq[pc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + 8 * u < ynp) {
          s += sv[u];
          q += qv[u];
        }
    }
    s += msl::dpp_mov<0xB1>(s);   // the eight parts of a channel sit in eight neighbouring lanes: quad, then half-row
    s += msl::dpp_mov<0x4E>(s);
    s += msl::dpp_mov<0x141>(s);
    q += msl::dpp_mov<0xB1>(q);
    q += msl::dpp_mov<0x4E>(q);
    q += msl::dpp_mov<0x141>(q);
    if (part == 0) {
      const float k1 = (float)(s / ycount), k2 = (float)(q / ycount);
      const float sc = vy[ch], mu = vy[2 * K + ch], is = vy[3 * K + ch];
      const float t = sc * is * k2;
      coef[ch] = sc;
      coef[K + ch] = vy[K + ch];
      coef[2 * K + ch] = -t;
      coef[3 * K + ch] = fmaf(t, mu, -sc * k1);
      if (blockIdx.x == 0) {
        dbeta_y[ch] = (float)s;
        dgamma_y[ch] = (float)q;
      }
    }
    if (tid < M) {
      cz[tid] = vz[tid];
      cz[M + tid] = vz[M + tid];
      cz[2 * M + tid] = vz[2 * M + tid];
      cz[3 * M + tid] = vz[3 * M + tid];
    }
  }
  if (__builtin_amdgcn_readfirstlane(tid >> 6) < 4)
    pw_bwd_fused_role<K, M, COLS, true>(g, y, Wt, z, gz, zpart, slabs, S, strips_per_img, total_strips, lds, coef, cz);
  else
    pw_bwd_fused_role<K, M, COLS, false>(g, y, Wt, z, gz, zpart, slabs, S, strips_per_img, total_strips, lds, coef, cz);
}

// workgroups (= statistics partials per channel = weight-gradient slabs): <= 256, so that the BatchNorm backward of z can fold
// them in its own prologue (msl_bn_relu_bwd_finalize_apply)
int fused_wgs(int N, int Cin, int Cout, int S) {
  if (N <= 0 || Cin != 32 || Cout != 64 || S % 128 != 0) return 0;
  const long long strips = (long long)N * (S / 128);
  if (strips < 512) return 0;  // small layers: the separate launches are latency-, not bandwidth-bound
  return 256;
}

}  // namespace

extern "C" {

int msl_pwconv_bwd_fused_num_partials(int N, int Cin, int Cout, int S) { return fused_wgs(N, Cin, Cout, S); }

int msl_pwconv_bwd_fused(const float* g_y, const float* y, const float* bn_y_vec, const double* y_partials, int y_np,
                         double y_count, float* dgamma_y, float* dbeta_y, const float* w, const float* z,
                         const float* bn_z_vec, float* g_z, double* z_partials, float* dw_slabs, int N, int Cin, int Cout, int S,
                         void* stream) {
  const int wgs = fused_wgs(N, Cin, Cout, S);
  if (!g_y || !y || !bn_y_vec || !y_partials || y_np <= 0 || !w || !z || !bn_z_vec || !g_z || !z_partials || !dw_slabs)
    return MSL_ERR_ARG;
  if (!wgs) return MSL_ERR_UNSUPPORTED;
  constexpr int K = 64, M = 32, COLS = 128;
  const size_t smem = ((size_t)(K + 2 * M) * (COLS + 1) + 4 * K + 4 * M) * sizeof(float);
  auto k = pw_bwd_fused_kernel<K, M, COLS>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  MSL_LAUNCH(k, dim3(wgs), dim3(512), smem, (hipStream_t)stream, g_y, y, bn_y_vec, y_partials, y_np, y_count, dgamma_y,
                     dbeta_y, w, z, bn_z_vec, g_z, z_partials, dw_slabs, S, S / COLS, N * (S / COLS));
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
