// Eval-mode stem + block-1 depthwise convolution in ONE pass (inference: BASELINE configs[3]).
//
// Reference: mobilenet.py:26-31 (conv_bn: Conv3d 3x3x3 s2 -> BatchNorm3d -> ReLU) followed by the first half of
// mobilenet.py:34-49 (Block: depthwise Conv3d 3x3x3 s2).  In eval mode the BatchNorm between the two convolutions is a
// constant per-channel affine, so nothing forces the stem activation (32 channels at half resolution: 226 MB at 192^3 x 2,
// written once and read once = 40 % of the pass's HBM traffic) to exist in memory.  (In training it must: its batch statistics
// are a grid-wide dependency.)
//
// A workgroup owns TH output rows x all columns x SD output planes of z1 = dw(relu(bn(stem(x)))) for one image and all 32
// channels, and marches through the 2*SD+1 activation planes it needs:
//   MFMA phase  the (2*TH+1) x AW positions of activation plane p as tiles of 32 columns, dealt to the 8 waves: the stem's
//               im2col GEMM with stem_fwd_rows_kernel's operand layout (the same 14 MFMAs in the same order, so the raw
//               values are bit-identical), B operands read from the plane's INPUT TILE in LDS (3 x planes x 4*TH+3 rows, zeros
//               outside the volume; a tile is requested into registers one plane step before it is stored to LDS: with two
//               waves per SIMD nothing else hides a load), BatchNorm affine + ReLU
//               in registers, result into the shared activation plane in LDS (fp32, one zero column on the left, rows
//               outside the map stay zero);
//   DW phase    every thread owns 1 channel x TH rows x (ZW/16) columns of outputs and adds plane p's 9 taps to the one or
//               two output planes it touches (p even: kd = 1 of plane p/2; p odd: kd = 2 of (p-1)/2, which completes it, and
//               kd = 0 of (p+1)/2), in the order (kd, kh, kw) of dw_s2_rows_eval_kernel: bit-identical outputs.
// Recomputed halo: one plane per SD output planes and one row per TH output rows ((2SD+1)/(2SD) * (2TH+1)/(2TH)).
// HBM traffic: x once (+ L2-served re-reads) and z1 once.
#include "common.hpp"

#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SDW_C = 32;     // stem output channels
constexpr int SDW_NCG = 3;    // output column groups of 16 a thread can own: ZW <= 48, i.e. AW <= 96, input W <= 192
constexpr int SDW_NW = 8;     // waves per workgroup
constexpr int SDW_NT = 64 * SDW_NW;
constexpr int SDW_NLD = 8;    // 16-byte x loads a thread can hold for the next plane's input tile

// AW (the activation map's width = W / 2) is a template parameter: every LDS pitch is then a constant and the tile loop has no
// integer multiplies (v_mul_lo_u32 is quarter rate; 16 of them per tile sat between the MFMAs)
template <int CIN, int TH, int AW, bool BF16>
__global__ __launch_bounds__(SDW_NT) void stem_dw_eval_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                              const float* __restrict__ wdw, void* __restrict__ z, int D, int H,
                                                              int AD, int AH, int ZD, int ZH, int SD) {
  constexpr int K = CIN * 27, KS = (K + 1) / 2, NRW = 2 * TH + 1, NXR = 2 * NRW + 1;
  constexpr int W = 2 * AW, ZW = AW / 2, AP = AW + 4, XP = W + 8, cpr = AW / 32, ipp = NRW * cpr, ncg = ZW / 16;
  static_assert(AW % 32 == 0 && ncg <= SDW_NCG, "activation width: 32, 64 or 96");
  // [CIN][3 planes][NXR rows][XP] input tile of the current activation plane (4 zero columns on the left, zeros outside the
  // volume) | [32][NRW][AP] activation plane | [32] BatchNorm scale | [32] shift
  extern __shared__ __align__(16) float sdw_lds[];
  float* xt = sdw_lds;
  float* a0 = sdw_lds + CIN * 3 * NXR * XP;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int n = blockIdx.z, oh0 = blockIdx.x * TH, od_begin = blockIdx.y * SD;
  const float* xn = x + (size_t)n * CIN * D * H * W;

  // the activation plane starts as zeros: column 0 (left padding) and a row above the map are never written
  for (int i = tid; i < SDW_C * NRW * AP / 4; i += SDW_NT) reinterpret_cast<float4*>(a0)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- MFMA-phase constants of the lane: k = 2 kk + h walks (ci, kd, kh, kw) as stem_fwd_rows_kernel does
  float wa[KS];
  int ldsoff[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 2 * kk + h;
    const bool vk = k < K;
    const int ci = k / 27, t = k % 27, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    wa[kk] = vk ? w[c * K + (vk ? k : 0)] : 0.f;
    ldsoff[kk] = (vk ? ((ci * 3 + kd) * NXR + kh) * XP + kw : 0) + 3 + 2 * c;
  }
  // the weights must STAY in registers: re-loaded inside the tile loop (the compiler's choice under register pressure) their
  // vmcnt waits are also waits for the older input-tile requests
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) msl::pin(wa[kk]);
  // the BatchNorm affine lives in LDS ([32] scale | [32] shift behind the activation plane): 32 more registers per lane made
  // the compiler re-load it from global memory inside the tile loop, and a wait for those loads is also a wait for the
  // older input-tile requests (vmcnt is in order) - the prefetch was waited for at every plane's first tile
  float* aff = a0 + SDW_C * NRW * AP;
  if (tid < SDW_C) aff[tid] = bn_scale[tid];
  else if (tid < 2 * SDW_C) aff[tid] = bn_shift[tid - SDW_C];
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(xn), 0, (int)((unsigned)CIN * D * H * W * 4u), 0x00020000);

  // ---- the thread's share of an input tile: slot e = tid + 512 i -> (ci, kd, row, 16-byte column); plane-invariant parts
  constexpr int q4 = (W >> 2) + 1, per_plane = NXR * q4, total_slots = CIN * 3 * per_plane;
  const int xr0 = 4 * oh0 - 3;  // first input row of the tile
  int sl_goff[SDW_NLD], sl_loff[SDW_NLD], sl_kd[SDW_NLD];  // global float offset without the plane term (-1: zeros), LDS float offset
#pragma unroll
  for (int i = 0; i < SDW_NLD; ++i) {
    const int e = tid + SDW_NT * i;
    const bool in = e < total_slots;
    const int ee = in ? e : 0;
    const int cik = ee / per_plane, rem = ee - cik * per_plane, row = rem / q4, col4 = rem - row * q4;
    const int ci = cik / 3, kd = cik - ci * 3;
    const int ih = xr0 + row;
    sl_kd[i] = in ? kd : -100000;  // (a slot outside the tile never stores)
    sl_loff[i] = (cik * NXR + row) * XP + 4 * col4;
    sl_goff[i] = (in && col4 >= 1 && ih >= 0 && ih < H) ? ((ci * D + kd) * H + ih) * W + 4 * (col4 - 1) : -1;
  }
  u32x4 pre[SDW_NLD];
  auto fetch = [&](int p) {  // the input tile of activation plane p: x planes 2p - 1 + kd
    const int idb = 2 * p - 1;
#pragma unroll
    for (int i = 0; i < SDW_NLD; ++i) {
      const int id = idb + sl_kd[i];
      pre[i] = (u32x4){0u, 0u, 0u, 0u};
      if (sl_goff[i] >= 0 && id >= 0 && id < D) pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (sl_goff[i] + idb * H * W) * 4, 0, 0);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < SDW_NLD; ++i)
      if (sl_kd[i] >= 0) *reinterpret_cast<u32x4*>(xt + sl_loff[i]) = pre[i];
  };

  // ---- DW-phase constants of the thread: channel cs, output columns c16 + 16 g
  const int c16 = tid & 15, cs = tid >> 4;
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = wdw[cs * 27 + k];
#pragma unroll
  for (int k = 0; k < 27; ++k) msl::pin(wk[k]);
  float acc[2][TH][SDW_NCG];  // [output plane parity][row][column group]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int g = 0; g < SDW_NCG; ++g) acc[s][r][g] = 0.f;

  typedef typename std::conditional<BF16, msl::su16, float>::type TZ;
  TZ* zn = reinterpret_cast<TZ*>(z) + (size_t)n * SDW_C * ZD * ZH * ZW;

  // items of the MFMA phase: (activation row hr, 32-column tile); wave wv takes items wv, wv + 8, ...
  const int nplanes = 2 * SD + 1, p_first = 2 * od_begin - 1;
  const int pi0 = p_first < 0 ? 1 : 0;  // plane -1 (above the volume) contributes nothing
  fetch(p_first + pi0);
  stash();
  if (pi0 + 1 < nplanes) fetch(p_first + pi0 + 1);
  __syncthreads();  // the zero fill, the affine and the first input tile
  for (int pi = pi0; pi < nplanes; ++pi) {
    const int p = p_first + pi;
    // ---------------- MFMA phase
#ifndef SDW_ABL_NO_MFMA
    // The BatchNorm + ReLU + LDS store of tile t ("epilogue": ~100 VALU / LDS instructions) is written BEHIND the MFMA chain of
    // tile t + 1, which does not depend on it: the scheduler interleaves the two, so the matrix pipe works through the
    // epilogues (ablation: MFMA chains alone 33 us, epilogues 25 us, back to back before this)
    // one element of a finished tile: BatchNorm affine + ReLU, into the activation plane
    auto epi1 = [&](const f32x16& a, float* dst, const float4 (&s4)[4], const float4 (&t4)[4], int e) {
      const int r4 = e >> 2, j = e & 3;
      float v = a[e];
      if (BF16) v = msl::bf2f(msl::f2bf(v));  // the bf16 path stores the raw stem output as bf16
      const float sv = j == 0 ? s4[r4].x : j == 1 ? s4[r4].y : j == 2 ? s4[r4].z : s4[r4].w;
      const float tv = j == 0 ? t4[r4].x : j == 1 ? t4[r4].y : j == 2 ? t4[r4].z : t4[r4].w;
      dst[(8 * r4 + j) * NRW * AP] = msl::act(v, sv, tv);  // lane rows 4 r4 + j = channel 8 r4 + 4 h + j
    };
    f32x16 a_prev = {0};
    float* dst_prev = a0;
    bool has_prev = false;
    float4 s4[4], t4[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      s4[r4] = *reinterpret_cast<const float4*>(aff + 8 * r4 + 4 * h);
      t4[r4] = *reinterpret_cast<const float4*>(aff + SDW_C + 8 * r4 + 4 * h);
    }
    for (int it = wv; it < ipp; it += SDW_NW) {
      const int hr = it / cpr, ow0 = (it - hr * cpr) * 32;
      if (2 * oh0 - 1 + hr < 0) continue;  // the row above the map: stays zero
      const float* buf = xt + 2 * hr * XP + 2 * ow0;
      // all B operands of the tile leave LDS before its first MFMA
      float bq[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) bq[kk] = buf[ldsoff[kk]];
      __builtin_amdgcn_sched_barrier(0);
      f32x16 a = {0};
      if (has_prev) {  // wave-uniform
        // the previous tile's 16 epilogue elements ride between this tile's dependent MFMAs (order pinned: the scheduler would
        // otherwise put them behind the chain again)
        static_assert(KS >= 8, "epilogue slots");
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kk], bq[kk], a, 0, 0, 0);
          constexpr int per = (16 + KS - 1) / KS;  // elements per slot (2 for KS = 14: slots 0..7)
#pragma unroll
          for (int u = 0; u < per; ++u)
            if (kk * per + u < 16) epi1(a_prev, dst_prev, s4, t4, kk * per + u);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kk], bq[kk], a, 0, 0, 0);
      }
      a_prev = a;
      dst_prev = a0 + (4 * h) * NRW * AP + hr * AP + 1 + ow0 + c;
      has_prev = true;
    }
    if (has_prev) {
#pragma unroll
      for (int e = 0; e < 16; ++e) epi1(a_prev, dst_prev, s4, t4, e);
    }
#endif
    __syncthreads();  // the activation plane is complete, the input tile is consumed
#ifndef SDW_ABL_NO_FETCH
    // the next plane's input tile (requested one plane step ago) goes to LDS, the one after it is requested - BEFORE the DW
    // phase's output stores: a vmcnt wait for loads then never waits for younger stores
    if (pi + 1 < nplanes) {
      stash();
      if (pi + 2 < nplanes) fetch(p + 2);
    }
#endif
    // ---------------- DW phase.  p % 4 fixes the tap plane (kd) and the accumulator slot (output plane parity) of the one
    // or two output planes this activation plane feeds, so each case is straight-line code on fixed registers
    if (p >= 0) {
      const int od_a = p >> 1;          // p even: kd = 1 of plane p/2; p odd: kd = 2 of plane (p-1)/2, which completes it
      const int od_b = (p + 1) >> 1;    // p odd: kd = 0 of plane (p+1)/2
      const bool use_a = od_a >= od_begin && od_a < od_begin + SD;
      const bool use_b = (p & 1) && od_b >= od_begin && od_b < od_begin + SD;
      auto step = [&](auto kda_, auto sa_, auto odd_) {
        constexpr int KDA = decltype(kda_)::value, SA = decltype(sa_)::value, SB = 1 - SA;
        constexpr bool ODD = decltype(odd_)::value;
        const float* ap = a0 + (size_t)cs * NRW * AP;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
#pragma unroll
          for (int g = 0; g < SDW_NCG; ++g) {
            if (g >= ncg) continue;
            const float* tp = ap + 2 * r * AP + 2 * (16 * g + c16);  // tap (kh, kw) at tp[kh * AP + kw]
            float t9[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const float2 t01 = *reinterpret_cast<const float2*>(tp + kh * AP);
              t9[kh * 3] = t01.x; t9[kh * 3 + 1] = t01.y; t9[kh * 3 + 2] = tp[kh * AP + 2];
            }
            if (use_a) {
#pragma unroll
              for (int k = 0; k < 9; ++k) acc[SA][r][g] = fmaf(wk[KDA * 9 + k], t9[k], acc[SA][r][g]);
            }
            if (ODD && use_b) {
#pragma unroll
              for (int k = 0; k < 9; ++k) acc[SB][r][g] = fmaf(wk[k], t9[k], acc[SB][r][g]);
            }
            if (ODD && use_a) {  // plane od_a of the output is complete
              TZ* dst = zn + ((size_t)cs * ZD + od_a) * ZH * ZW + (size_t)(oh0 + r) * ZW + 16 * g + c16;
              if (BF16) *reinterpret_cast<msl::su16*>(dst) = msl::f2bf(acc[SA][r][g]);
              else *reinterpret_cast<float*>(dst) = acc[SA][r][g];
              acc[SA][r][g] = 0.f;
            }
          }
        }
      };
      typedef std::integral_constant<int, 0> I0;
      typedef std::integral_constant<int, 1> I1;
      typedef std::integral_constant<int, 2> I2;
#ifndef SDW_ABL_NO_DW
      switch (p & 3) {  // block-uniform
        case 0: step(I1{}, I0{}, std::false_type{}); break;
        case 2: step(I1{}, I1{}, std::false_type{}); break;
        case 1: step(I2{}, I0{}, std::true_type{}); break;
        default: step(I2{}, I1{}, std::true_type{}); break;
      }
#endif
    }
    __syncthreads();  // the activation plane is consumed, the next input tile is in place
  }
}

struct SdwPlan {
  int th = 0, sd = 0, ap = 0, xp = 0;
  size_t lds = 0;
};

// the tile of a supported shape (th = 0: not supported)
inline SdwPlan sdw_plan(int N, int Cin, int D, int H, int W) {
  SdwPlan p;
  if (N <= 0 || Cin < 1 || Cin > 2 || D < 4 || H < 4 || W < 4 || D % 4 || H % 4 || W % 4) return p;
  const int AW = W / 2, ZD = D / 4, ZH = H / 4;
  if (AW % 32 != 0 || AW > 32 * SDW_NCG) return p;
  if ((long long)Cin * D * H * W >= (1ll << 30)) return p;  // in-image byte offsets are 32-bit
  const int ap = AW + 4;  // column 0 = left padding, AW columns, even, rows 16-byte aligned
  const int xp = W + 8;   // 4 zero columns on the left, W columns; (W + 8) % 32 == 8 for W % 32 == 0
  int th = 0;
  for (int t : {3, 4, 2}) {
    if (ZH % t) continue;
    const int nrw = 2 * t + 1, nxr = 2 * nrw + 1;
    const size_t lds = ((size_t)Cin * 3 * nxr * xp + (size_t)SDW_C * nrw * ap + 2 * SDW_C) * sizeof(float);
    const int slots = Cin * 3 * nxr * (W / 4 + 1);
    if (lds <= 160 * 1024 && slots <= SDW_NLD * SDW_NT) { th = t; p.lds = lds; break; }
  }
  if (!th) return p;
  // output planes per workgroup: as many as keep about one workgroup per CU (fewer planes = more recomputed halo planes)
  int sd = 1;
  for (int s : {8, 6, 4, 3, 2, 1}) {
    if (ZD % s) continue;
    sd = s;
    if ((long long)N * (ZH / th) * (ZD / s) >= 256) break;
  }
  p.th = th; p.sd = sd; p.ap = ap; p.xp = xp;
  return p;
}

}  // namespace

extern "C" {

// 1 if msl_stem_dw_fwd_eval takes this input shape (else: msl_stem_conv_fwd + msl_dwconv_fwd)
int msl_stem_dw_fwd_eval_supported(int N, int Cin, int D, int H, int W) { return sdw_plan(N, Cin, D, H, W).th ? 1 : 0; }

// z1 (N, 32, D/4, H/4, W/4) = depthwise 3x3x3 stride-2 convolution (weights wdw [32][27]) of relu(scale * stem(x) + shift),
// stem = dense 3x3x3 stride-2 convolution 1..2 -> 32 channels (weights w [32][Cin*27]), both padded by 1.  fp32 output, or bf16
// (`bf16_out`: the raw stem output is rounded to bf16 before the affine, as the bf16 path stores it).
static int stem_dw_impl(const float* x, const float* w, const float* bn_scale, const float* bn_shift, const float* wdw, void* z,
                        int N, int Cin, int D, int H, int W, bool bf16_out, void* stream) {
  const SdwPlan p = sdw_plan(N, Cin, D, H, W);
  if (!p.th) return MSL_ERR_UNSUPPORTED;
  const int AD = D / 2, AH = H / 2, AW = W / 2, ZD = D / 4, ZH = H / 4;
  dim3 grid(ZH / p.th, ZD / p.sd, N);
  hipStream_t st = (hipStream_t)stream;
#define MSL_SDW(CI, TH_, AW_, B_)                                                                                      \
  do {                                                                                                                 \
    if (p.lds > 64 * 1024) {                                                                                           \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_dw_eval_kernel<CI, TH_, AW_, B_>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);                     \
      if (e_ != hipSuccess) return (int)e_;                                                                            \
    }                                                                                                                  \
    MSL_LAUNCH((stem_dw_eval_kernel<CI, TH_, AW_, B_>), grid, dim3(SDW_NT), p.lds, st, x, w, bn_scale, bn_shift, wdw, z, \
               D, H, AD, AH, ZD, ZH, p.sd);                                                                            \
  } while (0)
#define MSL_SDW_AW(CI, TH_, B_)                                                                                        \
  do {                                                                                                                 \
    if (AW == 96) MSL_SDW(CI, TH_, 96, B_); else if (AW == 64) MSL_SDW(CI, TH_, 64, B_); else MSL_SDW(CI, TH_, 32, B_); \
  } while (0)
#define MSL_SDW_TH(CI, B_)                                                                                             \
  do {                                                                                                                 \
    if (p.th == 4) MSL_SDW_AW(CI, 4, B_); else if (p.th == 3) MSL_SDW_AW(CI, 3, B_); else MSL_SDW_AW(CI, 2, B_);       \
  } while (0)
  if (Cin == 1) { if (bf16_out) MSL_SDW_TH(1, true); else MSL_SDW_TH(1, false); }
  else { if (bf16_out) MSL_SDW_TH(2, true); else MSL_SDW_TH(2, false); }
#undef MSL_SDW_TH
#undef MSL_SDW_AW
#undef MSL_SDW
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_stem_dw_fwd_eval(const float* x, const float* w, const float* bn_scale, const float* bn_shift, const float* wdw,
                         float* z, int N, int Cin, int D, int H, int W, void* stream) {
  return stem_dw_impl(x, w, bn_scale, bn_shift, wdw, z, N, Cin, D, H, W, false, stream);
}

int msl_stem_dw_fwd_eval_bf16(const float* x, const float* w, const float* bn_scale, const float* bn_shift, const float* wdw,
                              void* z_bf16, int N, int Cin, int D, int H, int W, void* stream) {
  return stem_dw_impl(x, w, bn_scale, bn_shift, wdw, z_bf16, N, Cin, D, H, W, true, stream);
}

}  // extern "C"
