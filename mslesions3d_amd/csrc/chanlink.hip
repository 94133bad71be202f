// Per-channel backward link between two pointwise GEMMs of the MobileNet-3D tail (reference: autograd of
// lesions3d/mobilenet.py:43-47 - Block.forward: relu(bn1(conv1(x))) with conv1 depthwise, x = relu(bn2(conv2(.))) of the
// previous block).  Between the bwd-data GEMM of block i and the bwd-data GEMM of block i-1 everything is per channel:
//
//     dL/da_i --BN1+ReLU backward--> dL/dz_i --depthwise bwd-data--> (+ head share) --BN2+ReLU backward--> dL/dy_{i-1}
//
// and at the tail of the network (blocks 3-7 at 128^3 x 4: 256 ... 16 384 elements per channel) a channel's whole
// population fits the registers of one workgroup.  The three launches of that link (bn_relu_bwd_fused_reg, dw bwd-data,
// bn_relu_bwd_fused_reg: 5-8 us each, all launch + memory latency) become ONE: every global load of the link (g, z, y,
// the head share, taps, BatchNorm vectors) is issued up front - one memory round trip -, the two BatchNorm reductions are
// wave (NW = 1) or 4/8-wave sums, and the transposed depthwise convolution gathers dL/dz from a zero-haloed LDS copy.
// HBM-/latency-bound byte mover: VALU + LDS, no MFMA.  No atomics; fixed summation order -> run-to-run bit-identical.
#include "common.hpp"
#include "../../include/mslesions3d_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// activation tensors are fp32 or bf16 in HBM (T = float / msl::su16, the bf16 activation path); all arithmetic is fp32
template <typename T>
__device__ __forceinline__ f32x4v ldq(const T* p) {
  const float4 v = msl::ld4(p);
  return (f32x4v){v.x, v.y, v.z, v.w};
}
template <typename T>
__device__ __forceinline__ void stq(T* p, f32x4v v) { msl::st4(p, make_float4(v[0], v[1], v[2], v[3])); }
template <typename T>
__device__ __forceinline__ f32x4v rounded(const T* tag, f32x4v v) {  // what a later pass would read back from storage
  const float4 r = msl::as_stored(tag, make_float4(v[0], v[1], v[2], v[3]));
  return (f32x4v){r.x, r.y, r.z, r.w};
}

// sum of (a, b) over the workgroup, result in every thread.  NW == 1: DPP / readlane only.
template <int NW>
__device__ __forceinline__ void chan_sum2(double& a, double& b, double* scratch) {
  a = msl::wave_sum(a);
  b = msl::wave_sum(b);
  if (NW > 1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // (scratch may still be read from the previous reduction)
    if (lane == 0) {
      scratch[2 * w] = a;
      scratch[2 * w + 1] = b;
    }
    __syncthreads();
    a = 0.0;
    b = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      a += scratch[2 * i];
      b += scratch[2 * i + 1];
    }
  }
}

// NW waves = one channel; QO / QI: float4 quads of z (output side of the depthwise layer) / y (input side) per thread.
// Extents are powers of two (lD, lH, lW = log2 of the INPUT extents): every index split is a shift or a mask - with one to
// four waves per SIMD these kernels are bound by the number of instructions a wave issues, not by memory.
// LDS image of dL/dz: [N][OD+2][OH+2][OW+8] floats, a row's data at columns 4 .. OW+3 (16-byte aligned), zero elsewhere.
// BWW: the link also produces the depthwise WEIGHT gradient dW[c][k] = sum_i relu(bn2(y))[i] * dL/dz[(i + 1 - k) / s] (the very
// operand pairs of the transposed convolution: one more FMA per pair) - the weight-gradient launch of the block disappears.
template <typename T, int NW, int QO, int QI, int STRIDE, bool BWW>
__global__ __launch_bounds__(NW * 64) void block_bwd_channel_link_kernel(
    T* __restrict__ g_z, const T* __restrict__ z, const float* __restrict__ vec_z, const float* __restrict__ w_dw,
    const T* __restrict__ y_prev, const float* __restrict__ vec_y, T* __restrict__ g_y,
    float* __restrict__ dgamma_z, float* __restrict__ dbeta_z, float* __restrict__ dgamma_y, float* __restrict__ dbeta_y,
    float* __restrict__ dw_dw, int N, int C, int lD, int lH, int lW, int accumulate) {
  constexpr int NT = NW * 64, LS = STRIDE == 2 ? 1 : 0;
  extern __shared__ __align__(16) float lds[];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int lOD = lD - LS, lOH = lH - LS, lOW = lW - LS;
  const int OD = 1 << lOD, OH = 1 << lOH, OW = 1 << lOW;
  const int lSo = lOD + lOH + lOW, lSi = lD + lH + lW;       // log2 elements per (image, channel)
  const int tot_qo = N << (lSo - 2), tot_qi = N << (lSi - 2);
  const int PW = OW + 8, PP = (OH + 2) * PW, pimg = (OD + 2) * PP, ptot = N * pimg;  // multiples of 4
  // (BWW: the image region is reused for the 27 x NT tap-gradient exchange once the gather is done)
  double* scratch = reinterpret_cast<double*>(lds + (BWW ? max(ptot, 27 * NT + 27 * 4) : ptot));

  // ---- every global load of the link, back to back (clamped addresses, masked later) -------------------------------
  f32x4v gq[QO], zq[QO], yq[QI], aq[QI];
#pragma unroll
  for (int q = 0; q < QO; ++q) {
    const int qi = min(tid + q * NT, tot_qo - 1), n = qi >> (lSo - 2), r = qi & ((1 << (lSo - 2)) - 1);
    const size_t off = (((size_t)n * C + c) << lSo) + 4 * r;
    gq[q] = ldq(g_z + off);
    zq[q] = ldq(z + off);
  }
#pragma unroll
  for (int q = 0; q < QI; ++q) {
    const int qi = min(tid + q * NT, tot_qi - 1), n = qi >> (lSi - 2), r = qi & ((1 << (lSi - 2)) - 1);
    yq[q] = ldq(y_prev + (((size_t)n * C + c) << lSi) + 4 * r);
  }
  if (accumulate) {  // (uniform branch around the whole group: the loads still leave back to back)
#pragma unroll
    for (int q = 0; q < QI; ++q) {
      const int qi = min(tid + q * NT, tot_qi - 1), n = qi >> (lSi - 2), r = qi & ((1 << (lSi - 2)) - 1);
      aq[q] = ldq(g_y + (((size_t)n * C + c) << lSi) + 4 * r);
    }
  } else {
#pragma unroll
    for (int q = 0; q < QI; ++q) aq[q] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  }
  const float sc1 = vec_z[c], sh1 = vec_z[C + c], mu1 = vec_z[2 * C + c], is1 = vec_z[3 * C + c];
  const float sc2 = vec_y[c], sh2 = vec_y[C + c], mu2 = vec_y[2 * C + c], is2 = vec_y[3 * C + c];
  float wt[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wt[k] = w_dw[c * 27 + k];
  // the halo (and everything else) of the LDS image starts at zero
  for (int i = tid; i < (ptot >> 2); i += NT) reinterpret_cast<f32x4v*>(lds)[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};

  // ---- BatchNorm1 + ReLU backward of z_i (mobilenet.py:45): dL/dz = scale * (gm - mean(gm) - xhat * mean(gm * xhat)) ----
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < QO; ++q) {
    const bool in = tid + q * NT < tot_qo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gm = (in && fmaf(zq[q][e], sc1, sh1) > 0.f) ? gq[q][e] : 0.f;
      const float xh = (zq[q][e] - mu1) * is1;
      gq[q][e] = gm;
      zq[q][e] = xh;
      s1 += gm;
      s2 += gm * xh;
    }
  }
  double t1 = (double)s1, t2 = (double)s2;
  chan_sum2<NW>(t1, t2, scratch);  // (NW > 1: its barriers also order the zero fill above before the writes below)
  {
    const double cnt_o = (double)(N << lSo);
    const float k1 = (float)(t1 / cnt_o), k2 = (float)(t2 / cnt_o);
    if (tid == 0) {
      dbeta_z[c] = (float)t1;
      dgamma_z[c] = (float)t2;
    }
    if (NW == 1) __syncthreads();  // single wave: orders the zero fill before the writes below
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const int qi = tid + q * NT;
      if (qi < tot_qo) {
        const int n = qi >> (lSo - 2), r = qi & ((1 << (lSo - 2)) - 1);
        f32x4v dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = sc1 * (gq[q][e] - k1 - zq[q][e] * k2);
        stq(g_z + (((size_t)n * C + c) << lSo) + 4 * r, dz);  // dL/dz_i: the depthwise weight gradient reads it
        dz = rounded(g_z, dz);  // (bf16 storage: the transposed convolution sees what a separate launch would read back)
        const int s = 4 * r, ow = s & (OW - 1), oh = (s >> lOW) & (OH - 1), od = s >> (lOW + lOH);
        *reinterpret_cast<f32x4v*>(lds + n * pimg + (od + 1) * PP + (oh + 1) * PW + 4 + ow) = dz;
      }
    }
  }
  __syncthreads();

  // ---- depthwise bwd-data (transposed 3x3x3, padding 1) + head share, then BatchNorm2 + ReLU backward of y_{i-1} ----------
  s1 = 0.f;
  s2 = 0.f;
  const int W = 1 << lW, H = 1 << lH;
  float dwa[BWW ? 27 : 1];
#pragma unroll
  for (int k = 0; k < (BWW ? 27 : 1); ++k) dwa[k] = 0.f;
#pragma unroll
  for (int q = 0; q < QI; ++q) {
    const int qi = tid + q * NT;
    const bool in = qi < tot_qi;
    const int qc = in ? qi : 0, n = qc >> (lSi - 2), r = qc & ((1 << (lSi - 2)) - 1);
    const int s = 4 * r, iw = s & (W - 1), ih = (s >> lW) & (H - 1), id = s >> (lW + lH);
    f32x4v acc = aq[q];
    f32x4v av;  // the block's input activation relu(bn2(y_{i-1})) at the quad (zero for a masked quad)
#pragma unroll
    for (int e = 0; e < 4; ++e) av[e] = (BWW && in) ? msl::act(yq[q][e], sc2, sh2) : 0.f;
    const float* img = lds + n * pimg + 4;  // (+4: the data columns of a row)
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int td = id + 1 - kd;
      if (STRIDE == 2 && (td & 1)) continue;
      const int pd = (STRIDE == 2 ? (td >> 1) : td) + 1;  // padded plane index, 0 .. OD+1
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int th = ih + 1 - kh;
        if (STRIDE == 2 && (th & 1)) continue;
        const int ph = (STRIDE == 2 ? (th >> 1) : th) + 1;
        const float* row = img + pd * PP + ph * PW;
        const float w0 = wt[kd * 9 + kh * 3], w1 = wt[kd * 9 + kh * 3 + 1], w2 = wt[kd * 9 + kh * 3 + 2];
        if (STRIDE == 1) {
          // g[w] += sum_kw w[kw] * dz[w + 1 - kw]: columns iw-1 .. iw+4 of the row = one 4-byte, one 16-byte, one 4-byte read
          const float lft = row[iw - 1], rgt = row[iw + 4];
          const f32x4v m = *reinterpret_cast<const f32x4v*>(row + iw);
          acc[0] = fmaf(w0, m[1], fmaf(w1, m[0], fmaf(w2, lft, acc[0])));
          acc[1] = fmaf(w0, m[2], fmaf(w1, m[1], fmaf(w2, m[0], acc[1])));
          acc[2] = fmaf(w0, m[3], fmaf(w1, m[2], fmaf(w2, m[1], acc[2])));
          acc[3] = fmaf(w0, rgt, fmaf(w1, m[3], fmaf(w2, m[2], acc[3])));
          if (BWW) {
            float& d0 = dwa[kd * 9 + kh * 3];
            float& d1 = dwa[kd * 9 + kh * 3 + 1];
            float& d2 = dwa[kd * 9 + kh * 3 + 2];
            d0 = fmaf(av[0], m[1], fmaf(av[1], m[2], fmaf(av[2], m[3], fmaf(av[3], rgt, d0))));
            d1 = fmaf(av[0], m[0], fmaf(av[1], m[1], fmaf(av[2], m[2], fmaf(av[3], m[3], d1))));
            d2 = fmaf(av[0], lft, fmaf(av[1], m[0], fmaf(av[2], m[1], fmaf(av[3], m[2], d2))));
          }
        } else {
          // input column w = iw + j: even -> tap 1 of output w/2; odd -> tap 0 of output (w+1)/2 and tap 2 of output (w-1)/2
          const int o = iw >> 1;  // even: an 8-byte aligned pair + one more column
          const float2 r01 = *reinterpret_cast<const float2*>(row + o);
          const float r2 = row[o + 2];
          acc[0] = fmaf(w1, r01.x, acc[0]);
          acc[1] = fmaf(w0, r01.y, fmaf(w2, r01.x, acc[1]));
          acc[2] = fmaf(w1, r01.y, acc[2]);
          acc[3] = fmaf(w0, r2, fmaf(w2, r01.y, acc[3]));
          if (BWW) {
            float& d0 = dwa[kd * 9 + kh * 3];
            float& d1 = dwa[kd * 9 + kh * 3 + 1];
            float& d2 = dwa[kd * 9 + kh * 3 + 2];
            d0 = fmaf(av[1], r01.y, fmaf(av[3], r2, d0));
            d1 = fmaf(av[0], r01.x, fmaf(av[2], r01.y, d1));
            d2 = fmaf(av[1], r01.x, fmaf(av[3], r01.y, d2));
          }
        }
      }
    }
    acc = rounded(g_y, acc);  // (bf16 storage: dL/d relu(bn2(y)) as the separate depthwise launch stores it)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gm = (in && fmaf(yq[q][e], sc2, sh2) > 0.f) ? acc[e] : 0.f;
      const float xh = (yq[q][e] - mu2) * is2;
      aq[q][e] = gm;
      yq[q][e] = xh;
      s1 += gm;
      s2 += gm * xh;
    }
  }
  if (BWW) {
    // 27 tap sums over the workgroup: transposed through LDS ([k][thread], over the image, which every wave has finished
    // reading once the barrier below is passed), 4 x 27 threads add a quarter of the threads each in thread order, 27 finish
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 27; ++k) lds[k * NT + tid] = dwa[k];
    __syncthreads();
    constexpr int PART = NT / 4;
    for (int t = tid; t < 108; t += NT) {  // (NT is a multiple of 64: the four quarters of a tap stay in one lane quad)
      const int k = t >> 2, part = t & 3;
      const f32x4v* src = reinterpret_cast<const f32x4v*>(lds + k * NT + part * PART);
      double sum = 0.0;
#pragma unroll 4
      for (int i = 0; i < PART / 4; ++i) {
        const f32x4v v = src[i];
        sum += (double)v[0];
        sum += (double)v[1];
        sum += (double)v[2];
        sum += (double)v[3];
      }
      // the four quarters of a tap sit in neighbouring lanes: quad sum by DPP, fixed order
      sum += msl::dpp_mov<0xB1>(sum);
      sum += msl::dpp_mov<0x4E>(sum);
      if (part == 0) dw_dw[c * 27 + k] = (float)sum;
    }
  }
  t1 = (double)s1;
  t2 = (double)s2;
  chan_sum2<NW>(t1, t2, scratch);
  const double cnt_i = (double)(N << lSi);
  const float k1 = (float)(t1 / cnt_i), k2 = (float)(t2 / cnt_i);
  if (tid == 0) {
    dbeta_y[c] = (float)t1;
    dgamma_y[c] = (float)t2;
  }
#pragma unroll
  for (int q = 0; q < QI; ++q) {
    const int qi = tid + q * NT;
    if (qi < tot_qi) {
      const int n = qi >> (lSi - 2), r = qi & ((1 << (lSi - 2)) - 1);
      f32x4v dy;
#pragma unroll
      for (int e = 0; e < 4; ++e) dy[e] = sc2 * (aq[q][e] - k1 - yq[q][e] * k2);
      stq(g_y + (((size_t)n * C + c) << lSi) + 4 * r, dy);
    }
  }
}

struct LinkPlan {
  int nw, qo, qi, lD, lH, lW;
  size_t smem;
};

static inline int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

// Waves per channel: at most 4 quads (16 elements) of the larger side per thread - these kernels are bound by the number of
// instructions a wave issues, so the work is spread over up to 16 waves; quads per thread rounded up to 1 / 2 / 4.
bool link_plan(int N, int D, int H, int W, int stride, LinkPlan& p) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return false;
  p.lD = ilog2_exact(D);
  p.lH = ilog2_exact(H);
  p.lW = ilog2_exact(W);
  if (p.lD < 0 || p.lH < 0 || p.lW < 2) return false;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  if (OW < 4 || (stride == 2 && (D < 2 || H < 2))) return false;
  const long long tqo = (long long)N * OD * OH * OW / 4, tqi = (long long)N * D * H * W / 4;
  if (tqi <= 128) p.nw = 1;
  else if (tqi <= 1024) p.nw = 4;
  else if (tqi <= 2048) p.nw = 8;
  else if (tqi <= 4096) p.nw = 16;
  else return false;
  const int nt = p.nw * 64;
  auto up = [](long long v) { return v <= 1 ? 1 : v <= 2 ? 2 : 4; };
  p.qi = up((tqi + nt - 1) / nt);
  p.qo = up((tqo + nt - 1) / nt);
  if (p.qo > 1) p.qo = p.qi;  // instantiated: QO == 1 or QO == QI
  const size_t ptot = (size_t)N * (OD + 2) * (OH + 2) * (OW + 8);
  p.smem = std::max(ptot, (size_t)27 * nt + 27 * 4) * 4 + 2 * p.nw * sizeof(double);
  return p.smem <= 156 * 1024;
}

}  // namespace

extern "C" int msl_block_bwd_channel_link_supported(int N, int D, int H, int W, int stride) {
  LinkPlan p;
  return link_plan(N, D, H, W, stride, p) ? p.nw : 0;  // waves per channel (1, 4, 8, 16); 0: not supported
}

template <typename T>
static int link_launch(T* g_z, const T* z, const float* vec_z, const float* w_dw, const T* y_prev, const float* vec_y, T* g_y,
                       float* dgamma_z, float* dbeta_z, float* dgamma_y, float* dbeta_y, float* dw_dw, int N, int C, int D,
                       int H, int W, int stride, int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || !g_z || !z || !vec_z || !w_dw || !y_prev || !vec_y || !g_y) return MSL_ERR_ARG;
  LinkPlan p;
  if (!link_plan(N, D, H, W, stride, p)) return MSL_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define MSL_LINK_B(NW_, QO_, QI_, S_, B_)                                                                                 \
  do {                                                                                                                   \
    auto k = block_bwd_channel_link_kernel<T, NW_, QO_, QI_, S_, B_>;                                                    \
    if (p.smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.smem); \
    MSL_LAUNCH(k, dim3(C), dim3(NW_ * 64), p.smem, st, g_z, z, vec_z, w_dw, y_prev, vec_y, g_y, dgamma_z, dbeta_z, \
                       dgamma_y, dbeta_y, dw_dw, N, C, p.lD, p.lH, p.lW, accumulate);                                     \
  } while (0)
#define MSL_LINK_S(NW_, QO_, QI_, S_)                  \
  do {                                                 \
    if (dw_dw) MSL_LINK_B(NW_, QO_, QI_, S_, true);    \
    else MSL_LINK_B(NW_, QO_, QI_, S_, false);         \
  } while (0)
#define MSL_LINK_Q(NW_, S_)                                            \
  do {                                                                 \
    if (p.qi == 1) MSL_LINK_S(NW_, 1, 1, S_);                          \
    else if (p.qi == 2 && p.qo == 1) MSL_LINK_S(NW_, 1, 2, S_);        \
    else if (p.qi == 2) MSL_LINK_S(NW_, 2, 2, S_);                     \
    else if (p.qo == 1) MSL_LINK_S(NW_, 1, 4, S_);                     \
    else MSL_LINK_S(NW_, 4, 4, S_);                                    \
  } while (0)
#define MSL_LINK(NW_)                       \
  do {                                      \
    if (stride == 1) MSL_LINK_Q(NW_, 1);    \
    else MSL_LINK_Q(NW_, 2);                \
  } while (0)
  if (p.nw == 1) MSL_LINK(1);
  else if (p.nw == 4) MSL_LINK(4);
  else if (p.nw == 8) MSL_LINK(8);
  else MSL_LINK(16);
#undef MSL_LINK
#undef MSL_LINK_Q
#undef MSL_LINK_S
#undef MSL_LINK_B
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

extern "C" {

int msl_block_bwd_channel_link(float* g_z, const float* z, const float* vec_z, const float* w_dw, const float* y_prev,
                               const float* vec_y, float* g_y, float* dgamma_z, float* dbeta_z, float* dgamma_y,
                               float* dbeta_y, float* dw_dw, int N, int C, int D, int H, int W, int stride, int accumulate,
                               void* stream) {
  return link_launch<float>(g_z, z, vec_z, w_dw, y_prev, vec_y, g_y, dgamma_z, dbeta_z, dgamma_y, dbeta_y, dw_dw, N, C, D, H, W,
                            stride, accumulate, stream);
}

// the same on bf16 activation / activation-gradient tensors (bf16 activation path; BatchNorm vectors, taps, all sums fp32/fp64)
int msl_block_bwd_channel_link_bf16(void* g_z, const void* z, const float* vec_z, const float* w_dw, const void* y_prev,
                                    const float* vec_y, void* g_y, float* dgamma_z, float* dbeta_z, float* dgamma_y,
                                    float* dbeta_y, float* dw_dw, int N, int C, int D, int H, int W, int stride, int accumulate,
                                    void* stream) {
  return link_launch<msl::su16>((msl::su16*)g_z, (const msl::su16*)z, vec_z, w_dw, (const msl::su16*)y_prev, vec_y,
                                (msl::su16*)g_y, dgamma_z, dbeta_z, dgamma_y, dbeta_y, dw_dw, N, C, D, H, W, stride, accumulate,
                                stream);
}

}  // extern "C"
