// Depthwise Conv3d(C, C, k3, stride s in {1,2}, pad 1, groups=C, no bias) — forward.
// Reference: Block.conv1 (lesions3d/mobilenet.py:38,44).  Layout NCDHW fp32, W fastest.
//
// The input is the RAW output of the previous conv; its BatchNorm+ReLU is applied while loading
// (relu(fma(x, scale[c], shift[c]))), so the normalised activation never exists in HBM.  The kernel
// also emits fp64 (sum, sumsq) partials of its own raw output for the following BatchNorm.
//
// Three MI355X shapes of the same algorithm (HBM-bound: ~1.5-6.7 flop/B):
//  * "wave" (square power-of-two planes: every depthwise layer of the 128^3 / 64^3 configurations): the input planes
//    of a wave's output slab live in registers, row / column neighbours come from DPP lane shifts or from the lane's
//    own second and third row load - no LDS tile, no barrier, every load of the wave in flight at once
//    (dw_s1_wave_kernel, dw_s2_wave_kernel below).
//  * "stream" (large H x W planes, e.g. 64^2 at 128^3): a workgroup owns a slab of output planes of one
//    (n, c) volume.  Input planes are streamed once through LDS (16-B coalesced global loads along W,
//    next plane prefetched into registers while the current one is consumed), each thread keeps the
//    rolling accumulators of 4 adjacent outputs along W.  For stride 2 the LDS rows are stored
//    de-interleaved (even | odd columns) so the stride-2 taps become conflict-free 16-B LDS reads.
//  * "resident" (small planes): the whole input slab (+halo) sits in LDS, one load phase, one compute phase.
// Workgroup ids are remapped so that the slabs of one volume (which share halo planes) land on the same
// XCD back to back (per-XCD L2).
#include "common.hpp"
#include <algorithm>
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// LDS row geometry
//   stride 1: [ 3 unused | x[-1]=0 | x[0..W) | x[W]=0 | pad ]      x[iw] at column 4+iw, RS = W + 8
//   stride 2: [ E[0..EW) | O[0..OQ) ]  E[j] = x[2j],  O[j] = x[2j-1] (O[0] = x[-1] = 0)
//             EW = roundup4(OW), OQ = roundup4(OW+1), RS = EW + OQ
template <int STRIDE>
__device__ __forceinline__ int row_stride(int W, int OW) {
  if (STRIDE == 1) return W + 8;
  return ((OW + 3) & ~3) + ((OW + 4) & ~3);
}

template <int STRIDE>
__device__ __forceinline__ void store_row4(float* row, int iw, int EW, float v0, float v1, float v2, float v3) {
  if (STRIDE == 1) {
    *reinterpret_cast<float4*>(row + 4 + iw) = make_float4(v0, v1, v2, v3);
  } else {
    const int j = iw >> 1;  // iw % 4 == 0
    *reinterpret_cast<float2*>(row + j) = make_float2(v0, v2);  // E[j], E[j+1]
    row[EW + j + 1] = v1;                                        // O[j+1] = x[2j+1]
    row[EW + j + 2] = v3;                                        // O[j+2] = x[2j+3]
  }
}

// the 6 (stride 1) / 9 (stride 2) input values of one LDS row that feed outputs ow..ow+3
template <int STRIDE>
struct RowTaps {
  float v[3][4];  // v[kw][out]
  __device__ __forceinline__ void load(const float* row, int ow, int EW) {
    if (STRIDE == 1) {
      const float4 m = *reinterpret_cast<const float4*>(row + 4 + ow);
      const float l = row[3 + ow], r = row[8 + ow];
      v[0][0] = l;   v[0][1] = m.x; v[0][2] = m.y; v[0][3] = m.z;
      v[1][0] = m.x; v[1][1] = m.y; v[1][2] = m.z; v[1][3] = m.w;
      v[2][0] = m.y; v[2][1] = m.z; v[2][2] = m.w; v[2][3] = r;
    } else {
      const float4 e = *reinterpret_cast<const float4*>(row + ow);
      const float4 o = *reinterpret_cast<const float4*>(row + EW + ow);
      const float o4 = row[EW + ow + 4];
      v[0][0] = o.x; v[0][1] = o.y; v[0][2] = o.z; v[0][3] = o.w;
      v[1][0] = e.x; v[1][1] = e.y; v[1][2] = e.z; v[1][3] = e.w;
      v[2][0] = o.y; v[2][1] = o.z; v[2][2] = o.w; v[2][3] = o4;
    }
  }
};

__device__ __forceinline__ int xcd_remap(int b, int total) {
  // blocks b and b+8 share an XCD; give each XCD a contiguous range of logical ids
  return (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
}

// Segmented per-channel reduction of per-item (sum, sumsq) held in LDS, fixed order.
__device__ __forceinline__ void emit_stats(const float* it_s, const float* it_q, int G, int L, int c0,
                                           int Ctot, double* partials, int NP, int p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int g = wv; g < G; g += nw) {
    double s = 0.0, q = 0.0;
    for (int i = lane; i < L; i += 64) {
      s += (double)it_s[g * L + i];
      q += (double)it_q[g * L + i];
    }
    s = msl::wave_sum(s);
    q = msl::wave_sum(q);
    if (lane == 0) {
      partials[(size_t)(c0 + g) * NP + p] = s;
      partials[((size_t)Ctot + c0 + g) * NP + p] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "stream" variant.  Workgroup = (n, c, slab of SLAB output planes); the channel is workgroup-uniform, so
// weights and the input affine live in SGPRs.  IPT items per thread (item = 4 adjacent outputs along W of
// one output row); LPT float4 prefetch registers per thread.
// MODE 0: forward.  MODE 1: bwd-weight — the same streamed/staged input, `y` holds dy (read-only) and the 27 tap
// sums dw[c][k] = sum dy[o] * a[o*s-1+k] of this workgroup's slab are emitted as fp64 partials [C*27][NP].
__device__ __forceinline__ float dot4(const float4 d, const float* v) {
  return fmaf(d.w, v[3], fmaf(d.z, v[2], fmaf(d.y, v[1], d.x * v[0])));
}

// Fixed-order reduction of 27 per-thread sums over a segment of `seglen` threads per channel g; red = [27][256].
__device__ __forceinline__ void emit_bww(float* red, const float* a27, int G, int seglen, int c0, double* partials,
                                         int NP, int p) {
#pragma unroll
  for (int k = 0; k < 27; ++k) red[k * 256 + threadIdx.x] = a27[k];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int pair = wv; pair < G * 27; pair += 4) {
    const int g = pair / 27, k = pair % 27;
    double s = 0.0;
    for (int i = lane; i < seglen; i += 64) s += (double)red[k * 256 + g * seglen + i];
    s = msl::wave_sum(s);
    if (lane == 0) partials[((size_t)(c0 + g) * 27 + k) * NP + p] = s;
  }
}

template <int STRIDE, int IPT, int LPT, int MODE>
__global__ __launch_bounds__(256, (IPT == 1 ? (MODE == 0 ? 5 : 4) : 2)) void dw_fwd_stream_kernel(
    const float* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ w, float* __restrict__ y, double* __restrict__ partials, int C, int D, int H,
    int W, int OD, int OH, int OW, int SLAB, int nslabs, int Nbatch, msl::BnFold fold) {
  extern __shared__ __align__(16) float lds[];
  __shared__ float s_aff[2];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int slab = lb % nslabs;
  const int nc = lb / nslabs;  // n * C + c
  const int n = nc / C, c = nc % C;
  const int od0 = slab * SLAB, od1 = min(OD, od0 + SLAB);

  const int RS = row_stride<STRIDE>(W, OW);
  const int EW = (OW + 3) & ~3;
  const int PS = (H + 2) * RS;
  const int OWV = OW >> 2;
  const int nitems = OH * OWV;
  const int W4 = W >> 2;
  const int tot4 = H * W4;  // float4 per input plane

  const bool affine = in_scale != nullptr || fold.partials != nullptr;
  float sc = 1.f, sh = 0.f;
  if (fold.partials) {  // fold the producer's BatchNorm statistics here instead of waiting for a finalize launch
    msl::bn_fold_block(fold, c, 1, &s_aff[0], &s_aff[1]);
    sc = s_aff[0];
    sh = s_aff[1];
  } else if (in_scale) {
    sc = in_scale[c];
    sh = in_shift[c];
  }
  // workgroup-uniform: keep them in scalar registers whatever path produced them
  sc = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc)));
  sh = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh)));
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = MODE == 0 ? w[c * 27 + k] : 0.f;  // uniform -> scalar registers

  // zero the tile once: halo rows/columns are never written again
  for (int i = threadIdx.x; i < PS; i += 256) lds[i] = 0.f;

  int it_lds[IPT], it_out[IPT];
  bool it_ok[IPT];
#pragma unroll
  for (int t = 0; t < IPT; ++t) {
    const int item = threadIdx.x + t * 256;
    it_ok[t] = item < nitems;
    const int it = it_ok[t] ? item : 0;
    const int oh = it / OWV, ow = (it % OWV) * 4;
    it_lds[t] = (STRIDE * oh) * RS + ow;
    it_out[t] = oh * OW + ow;
  }

  float acc_a[IPT][4], acc_b[IPT][4];  // stride 2: a = current od.  stride 1: a = od p-1, b = od p
  float st_s[IPT], st_q[IPT];
  float a27[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) a27[k] = 0.f;
#pragma unroll
  for (int t = 0; t < IPT; ++t) {
    st_s[t] = st_q[t] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc_a[t][v] = acc_b[t][v] = 0.f;
  }

  const int p_begin = STRIDE == 2 ? 2 * od0 - 1 : od0 - 1;
  const int p_end = STRIDE == 2 ? 2 * od1 - 1 : od1;  // inclusive

  // per-prefetch-slot constants (no integer division inside the plane loop)
  float4 pre[LPT];
  int gofs[LPT], lofs[LPT];
  const float* xc = x + (size_t)nc * D * H * W;
  const size_t plane_elems = (size_t)H * W;
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int q = threadIdx.x + i * 256;
    const int qq = q < tot4 ? q : 0;
    const int ih = qq / W4, iw = (qq % W4) * 4;
    gofs[i] = q < tot4 ? qq * 4 : -1;
    lofs[i] = (ih + 1) * RS + (STRIDE == 1 ? 4 + iw : (iw >> 1));
  }
  auto issue_loads = [&](int p) {
    if (p < 0 || p >= D) return;
    const float* xp = xc + (size_t)p * plane_elems;
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      if (gofs[i] >= 0) pre[i] = *reinterpret_cast<const float4*>(xp + gofs[i]);
  };
  auto commit_loads = [&](int p) {
    if (p < 0 || p >= D) return;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      if (gofs[i] < 0) continue;
      float4 v = pre[i];
      if (affine) {
        v.x = msl::act(v.x, sc, sh); v.y = msl::act(v.y, sc, sh);
        v.z = msl::act(v.z, sc, sh); v.w = msl::act(v.w, sc, sh);
      }
      float* dst = lds + lofs[i];
      if (STRIDE == 1) {
        *reinterpret_cast<float4*>(dst) = v;
      } else {
        *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.z);
        dst[EW + 1] = v.y;
        dst[EW + 2] = v.w;
      }
    }
  };
  float* yc = y + (size_t)nc * OD * OH * OW;
  const int oplane = OH * OW;

  issue_loads(p_begin);
  for (int p = p_begin; p <= p_end; ++p) {
    const bool pvalid = p >= 0 && p < D;
    __syncthreads();  // everyone finished reading the previous plane (and the zero fill on the first trip)
    commit_loads(p);
    __syncthreads();
    if (p < p_end) issue_loads(p + 1);  // in flight while this plane is consumed

    if constexpr (MODE == 1) {
      if (pvalid) {
        const float* dyc = yc;
        // output planes this input plane meets, by tap kd: stride 2: odd p -> kd 0 with (p+1)/2, kd 2 with (p-1)/2;
        // even p -> kd 1 with p/2.  stride 1: kd 0,1,2 with p+1, p, p-1.  Only planes of this slab count.
        int odk[3];
        if (STRIDE == 2) {
          odk[0] = (p & 1) ? (p + 1) >> 1 : -1;
          odk[1] = (p & 1) ? -1 : p >> 1;
          odk[2] = (p & 1) ? (p - 1) >> 1 : -1;
        } else {
          odk[0] = p + 1; odk[1] = p; odk[2] = p - 1;
        }
#pragma unroll
        for (int t = 0; t < IPT; ++t) {
          if (!it_ok[t]) continue;
          float4 dv[3];
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            const bool use = odk[kd] >= od0 && odk[kd] < od1;
            dv[kd] = use ? *reinterpret_cast<const float4*>(dyc + (size_t)odk[kd] * oplane + it_out[t])
                         : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            RowTaps<STRIDE> r;
            r.load(lds + it_lds[t] + kh * RS, 0, EW);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              if (STRIDE == 2) {
                if (p & 1) {
                  a27[kh * 3 + kw] += dot4(dv[0], r.v[kw]);
                  a27[18 + kh * 3 + kw] += dot4(dv[2], r.v[kw]);
                } else {
                  a27[9 + kh * 3 + kw] += dot4(dv[1], r.v[kw]);
                }
              } else {
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) a27[kd * 9 + kh * 3 + kw] += dot4(dv[kd], r.v[kw]);
              }
            }
          }
        }
      }
    } else {
      if (STRIDE == 2) {
        if (p & 1) {
          const int od_done = (p - 1) >> 1;
  #pragma unroll
          for (int t = 0; t < IPT; ++t) {
            if (!it_ok[t]) continue;
            float nxt[4] = {0.f, 0.f, 0.f, 0.f};
            if (pvalid) {
  #pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                RowTaps<2> r;
                r.load(lds + it_lds[t] + kh * RS, 0, EW);
  #pragma unroll
                for (int kw = 0; kw < 3; ++kw)
  #pragma unroll
                  for (int v = 0; v < 4; ++v) {
                    acc_a[t][v] = fmaf(wk[18 + kh * 3 + kw], r.v[kw][v], acc_a[t][v]);
                    nxt[v] = fmaf(wk[kh * 3 + kw], r.v[kw][v], nxt[v]);
                  }
              }
            }
            if (od_done >= od0) {
              *reinterpret_cast<float4*>(yc + (size_t)od_done * oplane + it_out[t]) =
                  make_float4(acc_a[t][0], acc_a[t][1], acc_a[t][2], acc_a[t][3]);
  #pragma unroll
              for (int v = 0; v < 4; ++v) {
                st_s[t] += acc_a[t][v];
                st_q[t] = fmaf(acc_a[t][v], acc_a[t][v], st_q[t]);
              }
            }
  #pragma unroll
            for (int v = 0; v < 4; ++v) acc_a[t][v] = nxt[v];
          }
        } else if (pvalid) {
  #pragma unroll
          for (int t = 0; t < IPT; ++t) {
            if (!it_ok[t]) continue;
  #pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              RowTaps<2> r;
              r.load(lds + it_lds[t] + kh * RS, 0, EW);
  #pragma unroll
              for (int kw = 0; kw < 3; ++kw)
  #pragma unroll
                for (int v = 0; v < 4; ++v) acc_a[t][v] = fmaf(wk[9 + kh * 3 + kw], r.v[kw][v], acc_a[t][v]);
            }
          }
        }
      } else {
        const int od_done = p - 1;
  #pragma unroll
        for (int t = 0; t < IPT; ++t) {
          if (!it_ok[t]) continue;
          float nxt[4] = {0.f, 0.f, 0.f, 0.f};
          if (pvalid) {
  #pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              RowTaps<1> r;
              r.load(lds + it_lds[t] + kh * RS, 0, EW);
  #pragma unroll
              for (int kw = 0; kw < 3; ++kw)
  #pragma unroll
                for (int v = 0; v < 4; ++v) {
                  acc_a[t][v] = fmaf(wk[18 + kh * 3 + kw], r.v[kw][v], acc_a[t][v]);
                  acc_b[t][v] = fmaf(wk[9 + kh * 3 + kw], r.v[kw][v], acc_b[t][v]);
                  nxt[v] = fmaf(wk[kh * 3 + kw], r.v[kw][v], nxt[v]);
                }
            }
          }
          if (od_done >= od0) {
            *reinterpret_cast<float4*>(yc + (size_t)od_done * oplane + it_out[t]) =
                make_float4(acc_a[t][0], acc_a[t][1], acc_a[t][2], acc_a[t][3]);
  #pragma unroll
            for (int v = 0; v < 4; ++v) {
              st_s[t] += acc_a[t][v];
              st_q[t] = fmaf(acc_a[t][v], acc_a[t][v], st_q[t]);
            }
          }
  #pragma unroll
          for (int v = 0; v < 4; ++v) {
            acc_a[t][v] = acc_b[t][v];
            acc_b[t][v] = nxt[v];
          }
        }
      }
    }
  }

  if constexpr (MODE == 1) {
    __syncthreads();  // the tile is no longer read; reuse it as the [27][256] reduction buffer
    emit_bww(lds, a27, 1, 256, c, partials, Nbatch * nslabs, n * nslabs + slab);
  } else if (partials) {
    __syncthreads();
    float* it_s = lds;
    float* it_q = lds + nitems;
#pragma unroll
    for (int t = 0; t < IPT; ++t) {
      const int item = threadIdx.x + t * 256;
      if (item < nitems) {
        it_s[item] = st_s[t];
        it_q[item] = st_q[t];
      }
    }
    __syncthreads();
    emit_stats(it_s, it_q, 1, nitems, c, C, partials, Nbatch * nslabs, n * nslabs + slab);
  }
}

// ---------------------------------------------------------------------------------------------
// "resident" variant: the whole input slab (+ zero halo) of G channels in LDS.
template <int STRIDE, int MODE>
__global__ __launch_bounds__(256) void dw_fwd_resident_kernel(
    const float* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ w, float* __restrict__ y, double* __restrict__ partials, int C, int D, int H,
    int W, int OD, int OH, int OW, int G, int SLAB, int nslabs, int Nbatch, int flip, int accumulate,
    msl::BnFold fold) {
  extern __shared__ __align__(16) float lds[];
  __shared__ float s_sc[64], s_sh[64];
  __shared__ float s_w[MODE == 0 ? 64 * 27 : 1];  // the G channels' taps (G <= 64): fetched with the prologue, not per item
  const int wbase = flip ? 26 : 0, wsgn = flip ? -1 : 1;  // flipped taps: stride-1 bwd-data == forward with w[26-k]
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int slab = lb % nslabs;
  const int vg = lb / nslabs;
  const int CG = C / G;
  const int n = vg / CG, c0 = (vg % CG) * G;
  const int od0 = slab * SLAB, od1 = min(OD, od0 + SLAB);
  const int nod = od1 - od0;

  const int RS = row_stride<STRIDE>(W, OW);
  const int EW = (OW + 3) & ~3;
  const int PS = (H + 2) * RS;
  const int NPL = STRIDE * SLAB + (STRIDE == 2 ? 1 : 2);  // planes kept: stride2: 2*SLAB+1, stride1: SLAB+2
  const int p0 = STRIDE * od0 - 1;                         // first plane kept
  const int CS = NPL * PS;                                 // one channel's tile
  const int OWV = OW >> 2;
  const int Lp = OH * OWV;        // items per output plane
  const int L = nod * Lp;         // items per channel in this slab
  const int nitems = G * L;
  const int W4 = W >> 2, plane4 = H * W4;

  if (MODE == 0) {
    for (int i = threadIdx.x; i < G * 27; i += 256) s_w[i] = w[(size_t)c0 * 27 + i];
  }
  for (int i = threadIdx.x; i < G * CS; i += 256) lds[i] = 0.f;
  const bool affine = in_scale != nullptr || fold.partials != nullptr;
  if (fold.partials) {
    msl::bn_fold_block(fold, c0, G, s_sc, s_sh);  // ends with a barrier
  } else {
    if (in_scale && threadIdx.x < G) {
      s_sc[threadIdx.x] = in_scale[c0 + threadIdx.x];
      s_sh[threadIdx.x] = in_shift[c0 + threadIdx.x];
    }
    __syncthreads();
  }
  const int tot4 = G * NPL * plane4;
  // 4 independent 16-byte loads in flight per thread (unconditional, on clamped addresses; see msl::pin)
  for (int q0 = threadIdx.x; q0 < tot4; q0 += 4 * 256) {
    float4 v[4];
    int dst[4];  // LDS row offset, -1 = nothing to store
    int iwv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * 256;
      const bool live = q < tot4;
      const int qq = live ? q : 0;
      const int g = qq / (NPL * plane4);
      const int r1 = qq % (NPL * plane4);
      const int pl = r1 / plane4, rem = r1 % plane4;
      const int p = p0 + pl;
      const bool ok = live && p >= 0 && p < D;
      const int ih = rem / W4;
      iwv[u] = (rem % W4) * 4;
      gv[u] = g;
      dst[u] = ok ? g * CS + pl * PS + (ih + 1) * RS : -1;
      v[u] = *reinterpret_cast<const float4*>(x + (((size_t)n * C + c0 + g) * D + (ok ? p : 0)) * H * W + (size_t)rem * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) msl::pin(v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (dst[u] < 0) continue;
      float4 t = v[u];
      if (affine) {
        const float sc = s_sc[gv[u]], sh = s_sh[gv[u]];
        t.x = msl::act(t.x, sc, sh); t.y = msl::act(t.y, sc, sh);
        t.z = msl::act(t.z, sc, sh); t.w = msl::act(t.w, sc, sh);
      }
      store_row4<STRIDE>(lds + dst[u], iwv[u], EW, t.x, t.y, t.z, t.w);
    }
  }
  __syncthreads();

  if constexpr (MODE == 1) {
    float a27[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) a27[k] = 0.f;
    const float* dyc = y;
    for (int item = threadIdx.x; item < nitems; item += 256) {
      const int g = item / L, r0 = item % L;
      const int odl = r0 / Lp, r1 = r0 % Lp;
      const int oh = r1 / OWV, ow = (r1 % OWV) * 4;
      const float4 dv = *reinterpret_cast<const float4*>(dyc + ((((size_t)n * C + c0 + g) * OD + od0 + odl) * OH + oh) * OW + ow);
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const float* base = lds + g * CS + (STRIDE * odl + kd) * PS + (STRIDE * oh) * RS;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          RowTaps<STRIDE> r;
          r.load(base + kh * RS, ow, EW);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) a27[kd * 9 + kh * 3 + kw] += dot4(dv, r.v[kw]);
        }
      }
    }
    __syncthreads();  // tile reads done: reuse the LDS as the [27][256] reduction buffer
    // G == 1: every thread's items belong to channel c0 (idle threads hold zeros); G > 1: one item per thread and
    // channel g owns the L consecutive threads [g*L, (g+1)*L)
    emit_bww(lds, a27, G, G == 1 ? 256 : L, c0, partials, Nbatch * nslabs, n * nslabs + slab);
    return;
  }
  // stats scratch lives after the tile
  float* it_s = lds + G * CS;
  float* it_q = it_s + nitems;
  for (int item = threadIdx.x; item < nitems; item += 256) {
    const int g = item / L, r0 = item % L;
    const int odl = r0 / Lp, r1 = r0 % Lp;
    const int oh = r1 / OWV, ow = (r1 % OWV) * 4;
    const float* wc = s_w + g * 27;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const float* base = lds + g * CS + (STRIDE * odl + kd) * PS + (STRIDE * oh) * RS;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        RowTaps<STRIDE> r;
        r.load(base + kh * RS, ow, EW);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float ww = wc[wbase + wsgn * (kd * 9 + kh * 3 + kw)];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = fmaf(ww, r.v[kw][v], acc[v]);
        }
      }
    }
    float* yo = y + ((((size_t)n * C + c0 + g) * OD + od0 + odl) * OH + oh) * OW + ow;
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(yo);
      acc[0] += old.x; acc[1] += old.y; acc[2] += old.z; acc[3] += old.w;
    }
    *reinterpret_cast<float4*>(yo) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      s += acc[v];
      q = fmaf(acc[v], acc[v], q);
    }
    it_s[item] = s;
    it_q[item] = q;
  }
  if (partials) {
    __syncthreads();
    emit_stats(it_s, it_q, G, L, c0, C, partials, Nbatch * nslabs, n * nslabs + slab);
  }
}

// ---------------------------------------------------------------------------------------------
// "wave" variant: stride 1 on planes of 4x4 / 8x8 / 16x16 (the tail of the network, where a whole plane is <= 64
// float4).  A wave owns SL output planes of CPW = 64 / (H*W/4) channels and keeps its SL+2 input planes in
// registers: one float4 per lane and plane, every load of the wave in flight at once.  The row above / below is
// the lane W/4 away and the column neighbours sit in the adjacent lanes, so the 3x3 window comes from DPP lane
// shifts (ds_bpermute for the 16x16 plane, whose 4-lane rows cross the 16-lane DPP rows).  No LDS tile, no
// barrier, no integer division per element; waves are independent.
using msl::ld2;
using msl::ld4;
using msl::st2;
using msl::st4;
typedef msl::su16 dwu16;  // bf16 storage (see common.hpp)

template <int S>
__device__ __forceinline__ float lane_minus(float v, int lane) {  // value held by lane - S (caller masks lanes without one)
  if constexpr (S == 1) return msl::dpp_mov<0x111>(v);       // row_shr:1
  else if constexpr (S == 2) return msl::dpp_mov<0x112>(v);  // row_shr:2
  else return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane - S) & 63) << 2, __float_as_int(v)));
}
template <int S>
__device__ __forceinline__ float lane_plus(float v, int lane) {
  if constexpr (S == 1) return msl::dpp_mov<0x101>(v);       // row_shl:1
  else if constexpr (S == 2) return msl::dpp_mov<0x102>(v);  // row_shl:2
  else return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + S) & 63) << 2, __float_as_int(v)));
}

template <int LOGW4, int LOGH, int SL, typename T = float>
__global__ __launch_bounds__(256) void dw_s1_wave_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ w, T* __restrict__ y, double* __restrict__ partials, int C, int D, int nslabs,
    int Nbatch, int flip, int accumulate, msl::BnFold fold) {
  constexpr int W4 = 1 << LOGW4, H = 1 << LOGH, P4 = W4 * H, CPW = 64 / P4, W = 4 * W4, HW = H * W, NPL = SL + 2;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int CG = C / CPW;
  if (gw >= Nbatch * CG * nslabs) return;  // whole wave
  const int slab = gw % nslabs, vg = gw / nslabs;
  const int n = vg / CG, c0 = (vg % CG) * CPW;
  const int cl = CPW == 1 ? 0 : lane >> (LOGW4 + LOGH);
  const int h = (lane >> LOGW4) & (H - 1), w4 = lane & (W4 - 1);
  const int c = c0 + cl;
  const int od0 = slab * SL;

  // every input plane of the wave leaves first (unconditional, on clamped plane indices; masked below)
  const T* xc = x + (size_t)(n * C + c) * D * HW + h * W + w4 * 4;
  float4 pv[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(od0 - 1 + i, 0), D - 1);
    pv[i] = ld4(xc + (size_t)p * HW);
  }
  float wk[27];  // CPW == 1: the channel is wave-uniform and the taps are scalar loads
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[(size_t)c * 27 + (flip ? 26 - k : k)];

  const bool affine = in_scale != nullptr || fold.partials != nullptr;
  float sc = 1.f, sh = 0.f;
  if (fold.partials) {
    float m_, i_;
    double v_;
    if (fold.NP <= 64) {
      msl::bn_fold_serial(fold, c, sc, sh, m_, i_, v_);  // lanes of a channel repeat the same (broadcast) loads
    } else {
      for (int k = 0; k < CPW; ++k) {
        float a, b;
        msl::bn_fold_wave(fold, c0 + k, a, b, m_, i_, v_);
        if (cl == k) {
          sc = a;
          sh = b;
        }
      }
    }
  } else if (in_scale) {
    sc = in_scale[c];
    sh = in_shift[c];
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) msl::pin(pv[i]);

  const bool up_ok = h > 0, dn_ok = h < H - 1, lf_ok = w4 > 0, rt_ok = w4 < W4 - 1;
  float acc[SL][4];
#pragma unroll
  for (int o = 0; o < SL; ++o)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[o][v] = 0.f;

#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform: a zero plane adds nothing
    float4 m = pv[i];
    if (affine) {
      m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
      m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
    }
    // R[kh][0..5] = columns (4*w4 - 1) .. (4*w4 + 4) of input row h - 1 + kh
    float R[3][6];
    R[1][1] = m.x; R[1][2] = m.y; R[1][3] = m.z; R[1][4] = m.w;
    {
      const float a = lane_minus<W4>(m.x, lane), b = lane_minus<W4>(m.y, lane);
      const float cc = lane_minus<W4>(m.z, lane), d = lane_minus<W4>(m.w, lane);
      R[0][1] = up_ok ? a : 0.f; R[0][2] = up_ok ? b : 0.f; R[0][3] = up_ok ? cc : 0.f; R[0][4] = up_ok ? d : 0.f;
    }
    {
      const float a = lane_plus<W4>(m.x, lane), b = lane_plus<W4>(m.y, lane);
      const float cc = lane_plus<W4>(m.z, lane), d = lane_plus<W4>(m.w, lane);
      R[2][1] = dn_ok ? a : 0.f; R[2][2] = dn_ok ? b : 0.f; R[2][3] = dn_ok ? cc : 0.f; R[2][4] = dn_ok ? d : 0.f;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      if constexpr (W4 > 1) {
        const float l = lane_minus<1>(R[kh][4], lane), r = lane_plus<1>(R[kh][1], lane);
        R[kh][0] = lf_ok ? l : 0.f;
        R[kh][5] = rt_ok ? r : 0.f;
      } else {
        R[kh][0] = 0.f;
        R[kh][5] = 0.f;
      }
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int o = i - kd;  // output plane od0 + o reads input plane od0 + o - 1 + kd
      if (o < 0 || o >= SL) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float ww = wk[kd * 9 + kh * 3 + kw];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[o][v] = fmaf(ww, R[kh][v + kw], acc[o][v]);
        }
    }
  }

  double ds = 0.0, dq = 0.0;
  T* yc = y + (size_t)(n * C + c) * D * HW + h * W + w4 * 4;
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = od0 + o;
    if (od >= D) continue;  // wave-uniform (ragged last slab)
    T* yo = yc + (size_t)od * HW;
    if (accumulate) {
      const float4 old = ld4(yo);
      acc[o][0] += old.x; acc[o][1] += old.y; acc[o][2] += old.z; acc[o][3] += old.w;
    }
    st4(yo, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      s += acc[o][v];
      q = fmaf(acc[o][v], acc[o][v], q);
    }
    ds += (double)s;
    dq += (double)q;
  }
  if (partials) {
    if constexpr (P4 == 64) {
      ds = msl::wave_sum(ds);
      dq = msl::wave_sum(dq);
    } else if constexpr (P4 == 16) {
      ds = msl::row16_sum(ds);
      dq = msl::row16_sum(dq);
    } else {
      ds += msl::dpp_mov<0xB1>(ds); ds += msl::dpp_mov<0x4E>(ds);
      dq += msl::dpp_mov<0xB1>(dq); dq += msl::dpp_mov<0x4E>(dq);
    }
    if ((lane & (P4 - 1)) == 0) {
      const int NP = Nbatch * nslabs, pidx = n * nslabs + slab;
      partials[(size_t)c * NP + pidx] = ds;
      partials[((size_t)C + c) * NP + pidx] = dq;
    }
  }
}

// Stride-2 sibling.  A lane owns one "cell" = output row oh x two adjacent outputs (ow = 2*w4, 2*w4+1) and loads the
// three input rows 2oh-1, 2oh, 2oh+1 of its float4 column itself (the repeated row is a cache hit), so the only
// cross-lane value is the column to the left (one DPP shift per row).  Planes of OH * W/4 >= 64 cells are split over
// several waves (WPP), smaller ones share a wave (CPW channels); a wave keeps the 2*SL+1 input planes of its SL
// output planes in registers.  Output plane o of the slab takes plane i = 2o + kd of the slab's planes.
template <int LOGW4, int LOGOH>
constexpr int s2_wave_block() {  // threads per workgroup: up to four waves of a split plane, else four independent waves
  return (1 << (LOGW4 + LOGOH)) == 128 ? 128 : 256;
}

template <int LOGW4, int LOGOH, int SL, typename T = float>
__global__ __launch_bounds__((s2_wave_block<LOGW4, LOGOH>())) void dw_s2_wave_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ w, T* __restrict__ y, double* __restrict__ partials, int C, int D, int OD,
    int nslabs, int Nbatch, msl::BnFold fold) {
  constexpr int W4 = 1 << LOGW4, OH = 1 << LOGOH, LOGC = LOGW4 + LOGOH, CELLS = 1 << LOGC;
  constexpr int CPW = CELLS >= 64 ? 1 : 64 / CELLS, WPP = CELLS >= 64 ? CELLS / 64 : 1;
  constexpr int W = 4 * W4, H = 2 * OH, HW = H * W, OW = 2 * W4, OHW = OH * OW, NPL = 2 * SL + 1;
  // the waves of a split plane form workgroups of up to four (eight-wave workgroups at ~150 VGPRs cost occupancy:
  // 38 us instead of 31 us on block 1); a workgroup's statistics are summed in LDS -> WPP / WPG partials per plane slab
  constexpr int WPG = WPP == 1 ? 4 : (WPP < 4 ? WPP : 4), GPP = WPP == 1 ? 1 : WPP / WPG;
  const int lane = threadIdx.x & 63;
  // the slabs / plane parts of one volume share halo planes and rows: keep them on one XCD (per-XCD L2)
  const int gw = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPG + (threadIdx.x >> 6));
  const int CG = C / CPW;
  if (WPP == 1 && gw >= Nbatch * CG * nslabs) return;  // whole wave (WPP > 1: the grid is exact)
  const int part = gw % WPP, t0 = gw / WPP;  // the waves of one plane are neighbours (they share halo rows)
  const int slab = t0 % nslabs, vg = t0 / nslabs;
  const int n = vg / CG, c0 = (vg % CG) * CPW;
  const int cl = CPW == 1 ? 0 : lane >> LOGC;
  const int cell = CPW == 1 ? part * 64 + lane : lane & (CELLS - 1);
  const int oh = cell >> LOGW4, w4 = cell & (W4 - 1);
  const int c = c0 + cl;
  const int od0 = slab * SL;
  const bool up_ok = oh > 0, lf_ok = w4 > 0;

  const T* xc = x + (size_t)(n * C + c) * D * HW + w4 * 4;
  const int row_m = (up_ok ? 2 * oh - 1 : 0) * W, row_0 = 2 * oh * W;
  float4 pv[NPL][3];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(2 * od0 - 1 + i, 0), D - 1);
    const T* xp = xc + (size_t)p * HW;
    pv[i][0] = ld4(xp + row_m);
    pv[i][1] = ld4(xp + row_0);
    pv[i][2] = ld4(xp + row_0 + W);
  }
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[(size_t)c * 27 + k];

  const bool affine = in_scale != nullptr || fold.partials != nullptr;
  float sc = 1.f, sh = 0.f;
  if (fold.partials) {
    float m_, i_;
    double v_;
    if (fold.NP <= 64) {
      msl::bn_fold_serial(fold, c, sc, sh, m_, i_, v_);
    } else {
      for (int k = 0; k < CPW; ++k) {
        float a, b;
        msl::bn_fold_wave(fold, c0 + k, a, b, m_, i_, v_);
        if (cl == k) {
          sc = a;
          sh = b;
        }
      }
    }
  } else if (in_scale) {
    sc = in_scale[c];
    sh = in_shift[c];
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) msl::pin(pv[i][r]);

  float acc[SL][2];
#pragma unroll
  for (int o = 0; o < SL; ++o) acc[o][0] = acc[o][1] = 0.f;

#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = 2 * od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform
    float Tr[3][5];  // Tr[kh][0..4] = columns 4*w4-1 .. 4*w4+3 of input row 2*oh-1+kh
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float4 m = pv[i][r];
      if (affine) {
        m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
        m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
      }
      if (r == 0) {  // the row above the first output row is padding
        m.x = up_ok ? m.x : 0.f; m.y = up_ok ? m.y : 0.f; m.z = up_ok ? m.z : 0.f; m.w = up_ok ? m.w : 0.f;
      }
      Tr[r][1] = m.x; Tr[r][2] = m.y; Tr[r][3] = m.z; Tr[r][4] = m.w;
      if constexpr (W4 > 1) {
        const float l = lane_minus<1>(m.w, lane);
        Tr[r][0] = lf_ok ? l : 0.f;
      } else {
        Tr[r][0] = 0.f;
      }
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      if ((i - kd) % 2 != 0 || i - kd < 0) continue;
      const int o = (i - kd) / 2;
      if (o >= SL) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float ww = wk[kd * 9 + kh * 3 + kw];
          acc[o][0] = fmaf(ww, Tr[kh][kw], acc[o][0]);
          acc[o][1] = fmaf(ww, Tr[kh][kw + 2], acc[o][1]);
        }
    }
  }

  double ds = 0.0, dq = 0.0;
  T* yc = y + (size_t)(n * C + c) * OD * OHW + oh * OW + 2 * w4;
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = od0 + o;
    if (od >= OD) continue;  // wave-uniform (ragged last slab)
    st2(yc + (size_t)od * OHW, make_float2(acc[o][0], acc[o][1]));
    const float s = acc[o][0] + acc[o][1];
    const float q = fmaf(acc[o][1], acc[o][1], acc[o][0] * acc[o][0]);
    ds += (double)s;
    dq += (double)q;
  }
  if (partials) {
    if constexpr (WPP > 1) {
      __shared__ double red[2][WPG];
      ds = msl::wave_sum(ds);
      dq = msl::wave_sum(dq);
      if (lane == 0) {
        red[0][part % WPG] = ds;
        red[1][part % WPG] = dq;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int k = 0; k < WPG; ++k) {
          s += red[0][k];
          q += red[1][k];
        }
        const int NP = Nbatch * nslabs * GPP, pidx = (n * nslabs + slab) * GPP + part / WPG;
        partials[(size_t)c * NP + pidx] = s;
        partials[((size_t)C + c) * NP + pidx] = q;
      }
      return;
    } else if constexpr (LOGC == 6) {
      ds = msl::wave_sum(ds);
      dq = msl::wave_sum(dq);
    } else if constexpr (LOGC == 5) {
      ds = msl::row16_sum(ds);
      dq = msl::row16_sum(dq);
      const double s_lo = msl::lane_value(ds, 0) + msl::lane_value(ds, 16), s_hi = msl::lane_value(ds, 32) + msl::lane_value(ds, 48);
      const double q_lo = msl::lane_value(dq, 0) + msl::lane_value(dq, 16), q_hi = msl::lane_value(dq, 32) + msl::lane_value(dq, 48);
      ds = lane < 32 ? s_lo : s_hi;
      dq = lane < 32 ? q_lo : q_hi;
    } else if constexpr (LOGC == 4) {
      ds = msl::row16_sum(ds);
      dq = msl::row16_sum(dq);
    } else {
      static_assert(LOGC == 3, "cells per plane: 8, 16, 32 or a multiple of 64");
      ds += msl::dpp_mov<0xB1>(ds); ds += msl::dpp_mov<0x4E>(ds); ds += msl::dpp_mov<0x141>(ds);  // 8 lanes
      dq += msl::dpp_mov<0xB1>(dq); dq += msl::dpp_mov<0x4E>(dq); dq += msl::dpp_mov<0x141>(dq);
    }
    if ((lane & ((CELLS >= 64 ? 64 : CELLS) - 1)) == 0) {
      const int NP = Nbatch * nslabs, pidx = n * nslabs + slab;
      partials[(size_t)c * NP + pidx] = ds;
      partials[((size_t)C + c) * NP + pidx] = dq;
    }
  }
}

// Stride-2 register-marching kernel for planes of ANY width W % 4 == 0 and even H, forward WITHOUT statistics (eval-mode
// inference: the 96^2 / 48^2 / 24^2 / 12^2 planes of a 192^3 volume, which the power-of-two kernels above do not take and
// the LDS-streamed kernel moves at 2.2 TB/s).  Same cell = (output row, two adjacent outputs) per lane and the same
// arithmetic order as dw_s2_wave_kernel; the differences: a wave owns RPW = floor(64 / W4) WHOLE output rows of one plane
// slab (W4 = W / 4 lanes per row; the remaining lanes idle: 48 of 64 at W = 96), so that a lane's left neighbour is
// always the previous lane of the same wave - fetched with a wave-wide DPP shift (wave_shr:1), since rows of 24 lanes
// straddle the 16-lane DPP rows - and the geometry is run-time (one integer division per lane, outside the plane loop).
template <int SL, typename T>
__global__ __launch_bounds__(256) void dw_s2_rows_eval_kernel(const T* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift,
                                                              const float* __restrict__ w, T* __restrict__ y, int C, int D,
                                                              int H, int W, int OD, int nslabs, int RPW, int RG,
                                                              int total_waves) {
  constexpr int NPL = 2 * SL + 1;
  const int W4 = W >> 2, OH = H >> 1, OW = W >> 1, HW = H * W, OHW = OH * OW;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
  if (gw >= total_waves) return;  // whole wave
  const int rg = gw % RG, t0 = gw / RG;  // the row groups of one plane slab are neighbours (they share halo rows)
  const int slab = t0 % nslabs, vg = t0 / nslabs;
  const int n = vg / C, c = vg % C;  // wave-uniform: taps and affine are scalar
  const int lr = lane / W4, w4 = lane - lr * W4;
  const int oh_raw = rg * RPW + lr;
  const bool live = lr < RPW && oh_raw < OH;
  const int oh = live ? oh_raw : 0;     // idle lanes repeat a valid cell's loads and store nothing
  const int od0 = slab * SL;
  const bool up_ok = oh > 0, lf_ok = w4 > 0;

  const T* xc = x + (size_t)(n * C + c) * D * HW + (live ? w4 : 0) * 4;
  const int row_m = (up_ok ? 2 * oh - 1 : 0) * W, row_0 = 2 * oh * W;
  float4 pv[NPL][3];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(2 * od0 - 1 + i, 0), D - 1);
    const T* xp = xc + (size_t)p * HW;
    pv[i][0] = ld4(xp + row_m);
    pv[i][1] = ld4(xp + row_0);
    pv[i][2] = ld4(xp + row_0 + W);
  }
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[(size_t)c * 27 + k];
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) msl::pin(pv[i][r]);

  float acc[SL][2];
#pragma unroll
  for (int o = 0; o < SL; ++o) acc[o][0] = acc[o][1] = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = 2 * od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform
    float Tr[3][5];  // Tr[kh][0..4] = columns 4*w4-1 .. 4*w4+3 of input row 2*oh-1+kh
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float4 m = pv[i][r];
      if (affine) {
        m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
        m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
      }
      if (r == 0) {  // the row above the first output row is padding
        m.x = up_ok ? m.x : 0.f; m.y = up_ok ? m.y : 0.f; m.z = up_ok ? m.z : 0.f; m.w = up_ok ? m.w : 0.f;
      }
      Tr[r][1] = m.x; Tr[r][2] = m.y; Tr[r][3] = m.z; Tr[r][4] = m.w;
      const float l = msl::dpp_mov<0x138>(m.w);  // wave_shr:1 - the previous lane's last column
      Tr[r][0] = lf_ok ? l : 0.f;
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      if ((i - kd) % 2 != 0 || i - kd < 0) continue;
      const int o = (i - kd) / 2;
      if (o >= SL) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float ww = wk[kd * 9 + kh * 3 + kw];
          acc[o][0] = fmaf(ww, Tr[kh][kw], acc[o][0]);
          acc[o][1] = fmaf(ww, Tr[kh][kw + 2], acc[o][1]);
        }
    }
  }
  if (!live) return;
  T* yc = y + (size_t)(n * C + c) * OD * OHW + oh * OW + 2 * w4;
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = od0 + o;
    if (od >= OD) continue;  // wave-uniform (ragged last slab)
    st2(yc + (size_t)od * OHW, make_float2(acc[o][0], acc[o][1]));
  }
}

// Stride-1 sibling of dw_s2_rows_eval_kernel (the 24^2 / 12^2 planes of a 192^3 volume): a lane owns output row h x four
// adjacent outputs and loads the three input rows h-1, h, h+1 of its own float4 column (two of the three are cache hits),
// so the only cross-lane values are the columns left and right of it (wave_shr:1 / wave_shl:1).  Tap order as
// dw_s1_wave_kernel.  W % 4 == 0, any H and D; forward without statistics.
template <int SL, typename T>
__global__ __launch_bounds__(256) void dw_s1_rows_eval_kernel(const T* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift,
                                                              const float* __restrict__ w, T* __restrict__ y, int C, int D,
                                                              int H, int W, int nslabs, int RPW, int RG, int total_waves) {
  constexpr int NPL = SL + 2;
  const int W4 = W >> 2, HW = H * W;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
  if (gw >= total_waves) return;  // whole wave
  const int rg = gw % RG, t0 = gw / RG;
  const int slab = t0 % nslabs, vg = t0 / nslabs;
  const int n = vg / C, c = vg % C;  // wave-uniform
  const int lr = lane / W4, w4 = lane - lr * W4;
  const int h_raw = rg * RPW + lr;
  const bool live = lr < RPW && h_raw < H;
  const int h = live ? h_raw : 0;
  const int od0 = slab * SL;
  const bool up_ok = h > 0, dn_ok = h < H - 1, lf_ok = w4 > 0, rt_ok = w4 < W4 - 1;

  const T* xc = x + (size_t)(n * C + c) * D * HW + (live ? w4 : 0) * 4;
  const int row_m = (up_ok ? h - 1 : h) * W, row_0 = h * W, row_p = (dn_ok ? h + 1 : h) * W;
  float4 pv[NPL][3];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(od0 - 1 + i, 0), D - 1);
    const T* xp = xc + (size_t)p * HW;
    pv[i][0] = ld4(xp + row_m);
    pv[i][1] = ld4(xp + row_0);
    pv[i][2] = ld4(xp + row_p);
  }
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[(size_t)c * 27 + k];
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) msl::pin(pv[i][r]);

  float acc[SL][4];
#pragma unroll
  for (int o = 0; o < SL; ++o)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[o][v] = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform: a zero plane adds nothing
    float R[3][6];  // R[kh][0..5] = columns 4*w4-1 .. 4*w4+4 of input row h-1+kh
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float4 m = pv[i][r];
      if (affine) {
        m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
        m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
      }
      const bool rok = r == 0 ? up_ok : r == 2 ? dn_ok : true;  // rows outside the plane are padding
      m.x = rok ? m.x : 0.f; m.y = rok ? m.y : 0.f; m.z = rok ? m.z : 0.f; m.w = rok ? m.w : 0.f;
      R[r][1] = m.x; R[r][2] = m.y; R[r][3] = m.z; R[r][4] = m.w;
      const float l = msl::dpp_mov<0x138>(m.w), rr = msl::dpp_mov<0x130>(m.x);  // wave_shr:1 / wave_shl:1
      R[r][0] = lf_ok ? l : 0.f;
      R[r][5] = rt_ok ? rr : 0.f;
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int o = i - kd;  // output plane od0 + o reads input plane od0 + o - 1 + kd
      if (o < 0 || o >= SL) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float ww = wk[kd * 9 + kh * 3 + kw];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[o][v] = fmaf(ww, R[kh][v + kw], acc[o][v]);
        }
    }
  }
  if (!live) return;
  T* yc = y + (size_t)(n * C + c) * D * HW + h * W + w4 * 4;
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = od0 + o;
    if (od >= D) continue;  // wave-uniform (ragged last slab)
    st4(yc + (size_t)od * HW, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
  }
}

// Weight gradient on the same register-marching layout (bwd-weight of the layers the two kernels above run forward):
// dw[c][k] = sum over outputs of dy[o] * a[o*s - 1 + k].  The wave holds its input planes AND the dy planes of its
// output slab in registers, every lane accumulates 27 tap sums over its cells, the lanes of a channel are then summed
// (DPP / wave sums in fp64, fixed order) into partials [C*27][NP]; msl_dwconv_bwd_weight_finalize adds the NP partials.
template <int P4>
__device__ __forceinline__ double group_sum(double v, int lane) {  // sum over each aligned group of P4 lanes
  if constexpr (P4 >= 64) {
    return msl::wave_sum(v);
  } else if constexpr (P4 == 32) {
    v = msl::row16_sum(v);
    const double lo = msl::lane_value(v, 0) + msl::lane_value(v, 16), hi = msl::lane_value(v, 32) + msl::lane_value(v, 48);
    return lane < 32 ? lo : hi;
  } else if constexpr (P4 == 16) {
    return msl::row16_sum(v);
  } else if constexpr (P4 == 8) {
    v += msl::dpp_mov<0xB1>(v); v += msl::dpp_mov<0x4E>(v); v += msl::dpp_mov<0x141>(v);
    return v;
  } else {
    static_assert(P4 == 4, "lane groups of 4, 8, 16, 32 or 64");
    v += msl::dpp_mov<0xB1>(v); v += msl::dpp_mov<0x4E>(v);
    return v;
  }
}

template <int LOGW4, int LOGH, int SL, typename T = float>
__global__ __launch_bounds__(256) void dw_s1_wave_bww_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const T* __restrict__ dy, double* __restrict__ partials, int C, int D, int nslabs, int Nbatch) {
  constexpr int W4 = 1 << LOGW4, H = 1 << LOGH, P4 = W4 * H, CPW = 64 / P4, W = 4 * W4, HW = H * W, NPL = SL + 2;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int CG = C / CPW;
  if (gw >= Nbatch * CG * nslabs) return;  // whole wave
  const int slab = gw % nslabs, vg = gw / nslabs;
  const int n = vg / CG, c0 = (vg % CG) * CPW;
  const int cl = CPW == 1 ? 0 : lane >> (LOGW4 + LOGH);
  const int h = (lane >> LOGW4) & (H - 1), w4 = lane & (W4 - 1);
  const int c = c0 + cl;
  const int od0 = slab * SL;

  const size_t cell = (size_t)(n * C + c) * D * HW + h * W + w4 * 4;
  float4 pv[NPL], dv[SL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(od0 - 1 + i, 0), D - 1);
    pv[i] = ld4(x + cell + (size_t)p * HW);
  }
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = min(od0 + o, D - 1);
    dv[o] = ld4(dy + cell + (size_t)od * HW);
  }
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) msl::pin(pv[i]);
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    msl::pin(dv[o]);
    if (od0 + o >= D) dv[o] = make_float4(0.f, 0.f, 0.f, 0.f);  // wave-uniform (ragged last slab)
  }

  const bool up_ok = h > 0, dn_ok = h < H - 1, lf_ok = w4 > 0, rt_ok = w4 < W4 - 1;
  float a27[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) a27[k] = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform
    float4 m = pv[i];
    if (affine) {
      m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
      m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
    }
    float R[3][6];
    R[1][1] = m.x; R[1][2] = m.y; R[1][3] = m.z; R[1][4] = m.w;
    {
      const float a = lane_minus<W4>(m.x, lane), b = lane_minus<W4>(m.y, lane);
      const float cc = lane_minus<W4>(m.z, lane), d = lane_minus<W4>(m.w, lane);
      R[0][1] = up_ok ? a : 0.f; R[0][2] = up_ok ? b : 0.f; R[0][3] = up_ok ? cc : 0.f; R[0][4] = up_ok ? d : 0.f;
    }
    {
      const float a = lane_plus<W4>(m.x, lane), b = lane_plus<W4>(m.y, lane);
      const float cc = lane_plus<W4>(m.z, lane), d = lane_plus<W4>(m.w, lane);
      R[2][1] = dn_ok ? a : 0.f; R[2][2] = dn_ok ? b : 0.f; R[2][3] = dn_ok ? cc : 0.f; R[2][4] = dn_ok ? d : 0.f;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      if constexpr (W4 > 1) {
        const float l = lane_minus<1>(R[kh][4], lane), r = lane_plus<1>(R[kh][1], lane);
        R[kh][0] = lf_ok ? l : 0.f;
        R[kh][5] = rt_ok ? r : 0.f;
      } else {
        R[kh][0] = 0.f;
        R[kh][5] = 0.f;
      }
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int o = i - kd;
      if (o < 0 || o >= SL) continue;
      const float4 d = dv[o];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          a27[kd * 9 + kh * 3 + kw] += fmaf(d.w, R[kh][kw + 3], fmaf(d.z, R[kh][kw + 2], fmaf(d.y, R[kh][kw + 1], d.x * R[kh][kw])));
    }
  }
  const int NP = Nbatch * nslabs, pidx = n * nslabs + slab;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const double t = group_sum<P4>((double)a27[k], lane);
    if ((lane & (P4 - 1)) == 0) partials[((size_t)c * 27 + k) * NP + pidx] = t;
  }
}

template <int LOGW4, int LOGOH, int SL, typename T = float>
__global__ __launch_bounds__((s2_wave_block<LOGW4, LOGOH>())) void dw_s2_wave_bww_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const T* __restrict__ dy, double* __restrict__ partials, int C, int D, int OD, int nslabs, int Nbatch) {
  constexpr int W4 = 1 << LOGW4, OH = 1 << LOGOH, LOGC = LOGW4 + LOGOH, CELLS = 1 << LOGC;
  constexpr int CPW = CELLS >= 64 ? 1 : 64 / CELLS, WPP = CELLS >= 64 ? CELLS / 64 : 1;
  constexpr int W = 4 * W4, H = 2 * OH, HW = H * W, OW = 2 * W4, OHW = OH * OW, NPL = 2 * SL + 1;
  constexpr int WPG = WPP == 1 ? 4 : (WPP < 4 ? WPP : 4), GPP = WPP == 1 ? 1 : WPP / WPG;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPG + (threadIdx.x >> 6));
  const int CG = C / CPW;
  if (WPP == 1 && gw >= Nbatch * CG * nslabs) return;  // whole wave (WPP > 1: the grid is exact)
  const int part = gw % WPP, t0 = gw / WPP;
  const int slab = t0 % nslabs, vg = t0 / nslabs;
  const int n = vg / CG, c0 = (vg % CG) * CPW;
  const int cl = CPW == 1 ? 0 : lane >> LOGC;
  const int cell = CPW == 1 ? part * 64 + lane : lane & (CELLS - 1);
  const int oh = cell >> LOGW4, w4 = cell & (W4 - 1);
  const int c = c0 + cl;
  const int od0 = slab * SL;
  const bool up_ok = oh > 0, lf_ok = w4 > 0;

  const T* xc = x + (size_t)(n * C + c) * D * HW + w4 * 4;
  const int row_m = (up_ok ? 2 * oh - 1 : 0) * W, row_0 = 2 * oh * W;
  float4 pv[NPL][3];
  float2 dv[SL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = min(max(2 * od0 - 1 + i, 0), D - 1);
    const T* xp = xc + (size_t)p * HW;
    pv[i][0] = ld4(xp + row_m);
    pv[i][1] = ld4(xp + row_0);
    pv[i][2] = ld4(xp + row_0 + W);
  }
  const T* dyc = dy + (size_t)(n * C + c) * OD * OHW + oh * OW + 2 * w4;
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    const int od = min(od0 + o, OD - 1);
    dv[o] = ld2(dyc + (size_t)od * OHW);
  }
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) msl::pin(pv[i][r]);
#pragma unroll
  for (int o = 0; o < SL; ++o) {
    asm volatile("" : "+v"(dv[o].x), "+v"(dv[o].y));
    if (od0 + o >= OD) dv[o] = make_float2(0.f, 0.f);  // wave-uniform (ragged last slab)
  }

  float a27[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) a27[k] = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int p = 2 * od0 - 1 + i;
    if (p < 0 || p >= D) continue;  // wave-uniform
    float Tr[3][5];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float4 m = pv[i][r];
      if (affine) {
        m.x = msl::act(m.x, sc, sh); m.y = msl::act(m.y, sc, sh);
        m.z = msl::act(m.z, sc, sh); m.w = msl::act(m.w, sc, sh);
      }
      if (r == 0) {
        m.x = up_ok ? m.x : 0.f; m.y = up_ok ? m.y : 0.f; m.z = up_ok ? m.z : 0.f; m.w = up_ok ? m.w : 0.f;
      }
      Tr[r][1] = m.x; Tr[r][2] = m.y; Tr[r][3] = m.z; Tr[r][4] = m.w;
      if constexpr (W4 > 1) {
        const float l = lane_minus<1>(m.w, lane);
        Tr[r][0] = lf_ok ? l : 0.f;
      } else {
        Tr[r][0] = 0.f;
      }
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      if ((i - kd) % 2 != 0 || i - kd < 0) continue;
      const int o = (i - kd) / 2;
      if (o >= SL) continue;
      const float2 d = dv[o];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) a27[kd * 9 + kh * 3 + kw] += fmaf(d.y, Tr[kh][kw + 2], d.x * Tr[kh][kw]);
    }
  }
  if constexpr (WPP > 1) {
    __shared__ double red[27][WPG];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const double t = msl::wave_sum((double)a27[k]);
      if (lane == 0) red[k][part % WPG] = t;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < WPG; ++q) t += red[threadIdx.x][q];
      const int NP = Nbatch * nslabs * GPP, pidx = (n * nslabs + slab) * GPP + part / WPG;
      partials[((size_t)c * 27 + threadIdx.x) * NP + pidx] = t;
    }
  } else {
    constexpr int GRP = CELLS >= 64 ? 64 : CELLS;
    const int NP = Nbatch * nslabs, pidx = n * nslabs + slab;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const double t = group_sum<GRP>((double)a27[k], lane);
      if ((lane & (GRP - 1)) == 0) partials[((size_t)c * 27 + k) * NP + pidx] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Generic fallback (any W, any plane size): one output per thread straight from global memory.
__global__ __launch_bounds__(256) void dw_fwd_naive_kernel(
    const float* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ w, float* __restrict__ y, int C, int D, int H, int W, int OD, int OH, int OW,
    int stride, msl::BnFold fold) {
  __shared__ float s_aff[2];
  const int nc = blockIdx.y, c = nc % C;
  const int OS = OD * OH * OW;
  const bool affine = in_scale != nullptr || fold.partials != nullptr;
  float sc = 1.f, sh = 0.f;
  if (fold.partials) {
    msl::bn_fold_block(fold, c, 1, &s_aff[0], &s_aff[1]);
    sc = s_aff[0];
    sh = s_aff[1];
  } else if (in_scale) {
    sc = in_scale[c];
    sh = in_shift[c];
  }
  const float* xc = x + (size_t)nc * D * H * W;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < OS; o += gridDim.x * 256) {
    const int ow = o % OW, oh = (o / OW) % OH, od = o / (OW * OH);
    float acc = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int id = od * stride - 1 + kd;
      if (id < 0 || id >= D) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * stride - 1 + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow * stride - 1 + kw;
          if (iw < 0 || iw >= W) continue;
          float v = xc[((size_t)id * H + ih) * W + iw];
          if (affine) v = msl::act(v, sc, sh);
          acc = fmaf(w[c * 27 + kd * 9 + kh * 3 + kw], v, acc);
        }
      }
    }
    y[(size_t)nc * OS + o] = acc;
  }
}

// Small maps without statistics (eval mode; the 6^3 map of a 192^3 volume, which no wave / rows kernel takes): one WAVE per
// (image, channel) parks the whole volume (<= 512 voxels), activated, in a zero-haloed wave-private LDS image and every lane
// computes outputs from there - 27 LDS reads instead of 27 dependent global loads per output, a quarter of the launch's
// workgroups.  Taps in the naive kernel's order (kd, kh, kw); a padded tap adds w * 0 where the naive kernel skips it:
// the same value bit for bit (an fmaf with a zero product leaves a non-negative-zero accumulator unchanged).
constexpr int DW_SMALL_VOX = 512, DW_SMALL_IMG = 1024;
template <typename T>
__global__ __launch_bounds__(256) void dw_small_eval_kernel(const T* __restrict__ x, const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift, const float* __restrict__ w,
                                                            T* __restrict__ y, int NC, int C, int D, int H, int W, int OD,
                                                            int OH, int OW, int stride) {
  __shared__ float img_all[4][DW_SMALL_IMG];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nc = blockIdx.x * 4 + wv;
  if (nc >= NC) return;  // whole wave; the kernel has no workgroup barrier
  float* img = img_all[wv];
  const int c = nc % C, Hp = H + 2, Wp = W + 2, S = D * H * W, OS = OD * OH * OW, vol = (D + 2) * Hp * Wp;
  const bool affine = in_scale != nullptr;
  const float sc = affine ? in_scale[c] : 1.f, sh = affine ? in_shift[c] : 0.f;
  for (int i = lane; i < vol; i += 64) img[i] = 0.f;
  __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in order; this only pins the compiler's order)
  const T* xc = x + (size_t)nc * S;
  for (int i = lane; i < S; i += 64) {
    const int iw = i % W, ih = (i / W) % H, id = i / (W * H);
    float v = msl::ld1(xc + i);
    if (affine) v = msl::act(v, sc, sh);
    img[((id + 1) * Hp + ih + 1) * Wp + iw + 1] = v;
  }
  __builtin_amdgcn_wave_barrier();
  float wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = w[c * 27 + k];
  for (int o = lane; o < OS; o += 64) {
    const int ow = o % OW, oh = (o / OW) % OH, od = o / (OW * OH);
    const float* tp = img + (od * stride * Hp + oh * stride) * Wp + ow * stride;
    float acc = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kd * 9 + kh * 3 + kw], tp[(kd * Hp + kh) * Wp + kw], acc);
    if constexpr (sizeof(T) == 2) y[(size_t)nc * OS + o] = msl::f2bf(acc);
    else y[(size_t)nc * OS + o] = acc;
  }
}

constexpr int STATS_CHUNK = 4096;
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ y,
                                                            double* __restrict__ partials, int C, int S,
                                                            int chunks) {
  __shared__ double scratch[8];
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const size_t base = ((size_t)n * C + c) * S;
  const int lo = chunk * STATS_CHUNK, hi = min(S, lo + STATS_CHUNK);
  float s = 0.f, q = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = y[base + i];
    s += v;
    q = fmaf(v, v, q);
  }
  const int NP = gridDim.z * chunks, p = n * chunks + chunk;
  const double t1 = msl::block_sum((double)s, scratch);
  __syncthreads();
  const double t2 = msl::block_sum((double)q, scratch);
  if (threadIdx.x == 0) {
    partials[(size_t)c * NP + p] = t1;
    partials[((size_t)C + c) * NP + p] = t2;
  }
}

struct DwPlan {
  int variant;  // 0 naive, 1 stream, 2 resident
  int G, SLAB, nslabs, ipt, lpt;
  size_t lds_bytes;
  int num_partials;
};

inline int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

DwPlan make_plan(int N, int C, int D, int H, int W, int stride) {
  DwPlan pl{};
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const bool fast = (W % 4 == 0) && (OW % 4 == 0) && (stride == 1 || (H % 2 == 0 && W % 2 == 0));
  const int S = OD * OH * OW;
  if (!fast) {
    pl.variant = 0;
    pl.num_partials = N * msl::cdiv(S, STATS_CHUNK);
    return pl;
  }
  const int RS = stride == 1 ? W + 8 : ((OW + 3) & ~3) + ((OW + 4) & ~3);
  const int PS = (H + 2) * RS;
  const int OWV = OW / 4, Lp = OH * OWV;
  constexpr int stream_min_hw = 1024;
  if (H * W >= stream_min_hw) {  // stream
    int G = 1;
    int ipt = msl::cdiv(G * Lp, 256);
    int lpt = msl::cdiv(G * H * (W / 4), 256);
    size_t lds = (size_t)G * PS * 4;
    lds = lds > (size_t)2 * G * Lp * 4 ? lds : (size_t)2 * G * Lp * 4;
    if (ipt > 4 || lpt > 12 || lds > 160 * 1024) {
      pl.variant = 0;
      pl.num_partials = N * msl::cdiv(S, STATS_CHUNK);
      return pl;
    }
    pl.variant = 1;
    pl.G = G;
    pl.ipt = ipt <= 1 ? 1 : 4;
    pl.lpt = lpt <= 4 ? 4 : 12;
    const int nvol = N * (C / G);
    int slabs = std::max(1, std::min(OD, 1024 / std::max(1, nvol)));
    int SLAB = msl::cdiv(OD, slabs);
    if (SLAB < 4) SLAB = std::min(4, OD);
    pl.SLAB = SLAB;
    pl.nslabs = msl::cdiv(OD, SLAB);
    pl.lds_bytes = lds;
    pl.num_partials = N * pl.nslabs;
    return pl;
  }
  // resident: choose SLAB (power of two dividing work) and G so that ~256..1024 items and <= 60 KB of LDS
  int SLAB = OD;
  const int nvol1 = N * C;
  while (SLAB > 2 && (nvol1 * msl::cdiv(OD, SLAB) < 1024 || (size_t)(stride * SLAB + 2) * PS * 4 > 40 * 1024) &&
         SLAB % 2 == 0 && SLAB / 2 >= 2)
    SLAB /= 2;
  int NPL = stride * SLAB + (stride == 2 ? 1 : 2);
  size_t per_ch = (size_t)NPL * PS * 4 + (size_t)2 * SLAB * Lp * 4;
  if (per_ch > 150 * 1024) {  // + 7.5 KB of static LDS
    pl.variant = 0;
    pl.num_partials = N * msl::cdiv(S, STATS_CHUNK);
    return pl;
  }
  int G = std::max(1, 256 / std::max(1, SLAB * Lp));
  G = std::min(pow2_floor(G), 64);
  while (G > 1 && (C % G != 0 || per_ch * G > 60 * 1024)) G /= 2;
  pl.variant = 2;
  pl.G = G;
  pl.SLAB = SLAB;
  pl.nslabs = msl::cdiv(OD, SLAB);
  pl.lds_bytes = per_ch * G;
  pl.num_partials = N * pl.nslabs;
  return pl;
}

// Register-marching wave kernel (dw_s1_wave_kernel): stride 1, square planes of 4 / 8 / 16.  Two depth slabs per
// cube of the network's tail (NP = 2N partials per channel, as the resident plan had).
struct WavePlan {
  bool ok;
  int logw4, logh, cpw, SL, nslabs, wpp;  // wpp: waves per plane (stride 2, planes of >= 64 cells)
};

WavePlan make_wave_plan(int N, int C, int D, int H, int W, int stride) {
  constexpr int enabled = 3;  // bit 0: stride 1, bit 1: stride 2
  WavePlan wp{};
  if (!enabled || H != W) return wp;
  if (stride == 2) {
    // (planes up to 64^2: with 32 block 1's 64^2 planes go back to the streamed kernel - slower, tools/ab_dw_wave.sh in round 1)
    constexpr int maxw = 64;
    if ((W != 8 && W != 16 && W != 32 && W != 64) || W > maxw || !(enabled & 2)) return wp;
    wp.logw4 = W == 8 ? 1 : W == 16 ? 2 : W == 32 ? 3 : 4;
    wp.logh = wp.logw4 + 1;  // log2(OH)
    const int cells = (H / 2) * (W / 4), OD = (D - 1) / 2 + 1;
    wp.cpw = cells >= 64 ? 1 : 64 / cells;
    wp.wpp = cells >= 64 ? cells / 64 : 1;
    if (C % wp.cpw != 0) return wp;
    wp.SL = OD >= 8 ? 4 : OD >= 4 ? 2 : 1;
    constexpr int sl2 = 0;  // tuning knob: cap
    if (sl2 > 0) wp.SL = std::min(wp.SL, sl2 >= 4 ? 4 : sl2 >= 2 ? 2 : 1);
    wp.nslabs = msl::cdiv(OD, wp.SL);
    if ((long long)N * (C / wp.cpw) * wp.nslabs * wp.wpp > (1ll << 30)) return wp;
    wp.ok = true;
    return wp;
  }
  wp.wpp = 1;
  if (!(enabled & 1) || stride != 1 || (W != 4 && W != 8 && W != 16)) return wp;
  wp.logw4 = W == 4 ? 0 : W == 8 ? 1 : 2;
  wp.logh = wp.logw4 + 2;
  wp.cpw = 64 / (H * W / 4);
  if (C % wp.cpw != 0) return wp;
  wp.SL = D >= 16 ? 8 : D >= 8 ? 4 : D >= 4 ? 2 : 1;
  constexpr int sl1 = 0;  // tuning knob: cap
  if (sl1 > 0) wp.SL = std::min(wp.SL, sl1 >= 8 ? 8 : sl1 >= 4 ? 4 : sl1 >= 2 ? 2 : 1);
  wp.nslabs = msl::cdiv(D, wp.SL);
  if ((long long)N * (C / wp.cpw) * wp.nslabs > (1ll << 30)) return wp;
  wp.ok = true;
  return wp;
}

template <typename T>
void launch_wave(const WavePlan& wp, const T* x, const float* in_scale, const float* in_shift, const float* w,
                 T* y, double* partials, int N, int C, int D, int flip, int accumulate, const msl::BnFold& fold,
                 hipStream_t st) {
  const int waves = N * (C / wp.cpw) * wp.nslabs;
  const dim3 grid(msl::cdiv(waves, 4)), block(256);
#define MSL_DW_WAVE_SL(LW_, LH_)                                                                                  \
  switch (wp.SL) {                                                                                                \
    case 8: MSL_LAUNCH((dw_s1_wave_kernel<LW_, LH_, 8, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, \
                               partials, C, D, wp.nslabs, N, flip, accumulate, fold); break;                      \
    case 4: MSL_LAUNCH((dw_s1_wave_kernel<LW_, LH_, 4, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, \
                               partials, C, D, wp.nslabs, N, flip, accumulate, fold); break;                      \
    case 2: MSL_LAUNCH((dw_s1_wave_kernel<LW_, LH_, 2, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, \
                               partials, C, D, wp.nslabs, N, flip, accumulate, fold); break;                      \
    default: MSL_LAUNCH((dw_s1_wave_kernel<LW_, LH_, 1, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, \
                                partials, C, D, wp.nslabs, N, flip, accumulate, fold); break;                     \
  }
  if (wp.logw4 == 0) { MSL_DW_WAVE_SL(0, 2) }
  else if (wp.logw4 == 1) { MSL_DW_WAVE_SL(1, 3) }
  else { MSL_DW_WAVE_SL(2, 4) }
#undef MSL_DW_WAVE_SL
}

template <typename T>
void launch_wave_s2(const WavePlan& wp, const T* x, const float* in_scale, const float* in_shift, const float* w,
                    T* y, double* partials, int N, int C, int D, const msl::BnFold& fold, hipStream_t st) {
  const int waves = N * (C / wp.cpw) * wp.nslabs * wp.wpp, OD = (D - 1) / 2 + 1;
  const int wpg = wp.wpp > 1 ? std::min(wp.wpp, 4) : 4;
  const dim3 grid(msl::cdiv(waves, wpg)), block(64 * wpg);
#define MSL_DW_WAVE2_SL(LW_, LH_)                                                                                  \
  switch (wp.SL) {                                                                                                 \
    case 4: MSL_LAUNCH((dw_s2_wave_kernel<LW_, LH_, 4, T>), grid, block, 0, st, x, in_scale, in_shift, w, y,  \
                               partials, C, D, OD, wp.nslabs, N, fold); break;                                     \
    case 2: MSL_LAUNCH((dw_s2_wave_kernel<LW_, LH_, 2, T>), grid, block, 0, st, x, in_scale, in_shift, w, y,  \
                               partials, C, D, OD, wp.nslabs, N, fold); break;                                     \
    default: MSL_LAUNCH((dw_s2_wave_kernel<LW_, LH_, 1, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, \
                                partials, C, D, OD, wp.nslabs, N, fold); break;                                    \
  }
  if (wp.logw4 == 1) { MSL_DW_WAVE2_SL(1, 2) }
  else if (wp.logw4 == 2) { MSL_DW_WAVE2_SL(2, 3) }
  else if (wp.logw4 == 3) { MSL_DW_WAVE2_SL(3, 4) }
  else { MSL_DW_WAVE2_SL(4, 5) }
#undef MSL_DW_WAVE2_SL
}

// eval-mode forward on dw_s2_rows_eval_kernel / dw_s1_rows_eval_kernel: no statistics, planes the power-of-two wave kernels do not take
// (the LDS kernels remain for the shapes these do not take)
bool rows_eval_ok(int N, int C, int D, int H, int W, int stride) {
  constexpr int on = 1;
  if (!on || W % 4 != 0 || W < 8 || W > 256 || D < 2) return false;
  if (stride == 2 && H % 2 != 0) return false;
  if (make_wave_plan(N, C, D, H, W, stride).ok) return false;
  const int OD = (D - 1) / stride + 1, SL = OD >= 8 ? 4 : OD >= 4 ? 2 : 1, RPW = 64 / (W / 4);
  return (long long)N * C * msl::cdiv(OD, SL) * msl::cdiv(stride == 2 ? H / 2 : H, RPW) < (1ll << 30);
}

template <typename T>
void launch_rows_eval(const T* x, const float* in_scale, const float* in_shift, const float* w, T* y, int N, int C, int D,
                      int H, int W, int stride, hipStream_t st) {
  if (stride == 1) {
    const int SL = D >= 8 ? 4 : D >= 4 ? 2 : 1, nslabs = msl::cdiv(D, SL);
    const int RPW = 64 / (W / 4), RG = msl::cdiv(H, RPW);
    const int waves = N * C * nslabs * RG;
    const dim3 grid(msl::cdiv(waves, 4)), block(256);
    switch (SL) {
      case 4: MSL_LAUNCH((dw_s1_rows_eval_kernel<4, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W,
                                 nslabs, RPW, RG, waves); break;
      case 2: MSL_LAUNCH((dw_s1_rows_eval_kernel<2, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W,
                                 nslabs, RPW, RG, waves); break;
      default: MSL_LAUNCH((dw_s1_rows_eval_kernel<1, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W,
                                  nslabs, RPW, RG, waves); break;
    }
    return;
  }
  const int OD = (D - 1) / 2 + 1, SL = OD >= 8 ? 4 : OD >= 4 ? 2 : 1, nslabs = msl::cdiv(OD, SL);
  const int RPW = 64 / (W / 4), RG = msl::cdiv(H / 2, RPW);
  const int waves = N * C * nslabs * RG;
  const dim3 grid(msl::cdiv(waves, 4)), block(256);
  switch (SL) {
    case 4: MSL_LAUNCH((dw_s2_rows_eval_kernel<4, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W, OD,
                               nslabs, RPW, RG, waves); break;
    case 2: MSL_LAUNCH((dw_s2_rows_eval_kernel<2, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W, OD,
                               nslabs, RPW, RG, waves); break;
    default: MSL_LAUNCH((dw_s2_rows_eval_kernel<1, T>), grid, block, 0, st, x, in_scale, in_shift, w, y, C, D, H, W, OD,
                                nslabs, RPW, RG, waves); break;
  }
}

// weight gradient on the wave kernels (the LDS-tiled / generic kernels remain for the other shapes)
bool wave_bww_enabled() {
  constexpr int on = 1;
  return on != 0;
}

int wave_bww_num_partials(const WavePlan& wp, int N) { return N * wp.nslabs * (wp.wpp > 4 ? wp.wpp / 4 : 1); }

template <typename T>
void launch_wave_bww(const WavePlan& wp, int stride, const T* x, const float* in_scale, const float* in_shift,
                     const T* dy, double* partials, int N, int C, int D, hipStream_t st) {
  const int waves = N * (C / wp.cpw) * wp.nslabs * wp.wpp;
  if (stride == 1) {
    const dim3 grid(msl::cdiv(waves, 4)), block(256);
#define MSL_DW_WAVE_BWW1(LW_, LH_)                                                                                \
  switch (wp.SL) {                                                                                                \
    case 8: MSL_LAUNCH((dw_s1_wave_bww_kernel<LW_, LH_, 8, T>), grid, block, 0, st, x, in_scale, in_shift,   \
                               dy, partials, C, D, wp.nslabs, N); break;                                          \
    case 4: MSL_LAUNCH((dw_s1_wave_bww_kernel<LW_, LH_, 4, T>), grid, block, 0, st, x, in_scale, in_shift,   \
                               dy, partials, C, D, wp.nslabs, N); break;                                          \
    case 2: MSL_LAUNCH((dw_s1_wave_bww_kernel<LW_, LH_, 2, T>), grid, block, 0, st, x, in_scale, in_shift,   \
                               dy, partials, C, D, wp.nslabs, N); break;                                          \
    default: MSL_LAUNCH((dw_s1_wave_bww_kernel<LW_, LH_, 1, T>), grid, block, 0, st, x, in_scale, in_shift,  \
                                dy, partials, C, D, wp.nslabs, N); break;                                         \
  }
    if (wp.logw4 == 0) { MSL_DW_WAVE_BWW1(0, 2) }
    else if (wp.logw4 == 1) { MSL_DW_WAVE_BWW1(1, 3) }
    else { MSL_DW_WAVE_BWW1(2, 4) }
#undef MSL_DW_WAVE_BWW1
    return;
  }
  const int wpg = wp.wpp > 1 ? std::min(wp.wpp, 4) : 4, OD = (D - 1) / 2 + 1;
  const dim3 grid(msl::cdiv(waves, wpg)), block(64 * wpg);
#define MSL_DW_WAVE_BWW2(LW_, LH_)                                                                                \
  switch (wp.SL) {                                                                                                \
    case 4: MSL_LAUNCH((dw_s2_wave_bww_kernel<LW_, LH_, 4, T>), grid, block, 0, st, x, in_scale, in_shift,   \
                               dy, partials, C, D, OD, wp.nslabs, N); break;                                      \
    case 2: MSL_LAUNCH((dw_s2_wave_bww_kernel<LW_, LH_, 2, T>), grid, block, 0, st, x, in_scale, in_shift,   \
                               dy, partials, C, D, OD, wp.nslabs, N); break;                                      \
    default: MSL_LAUNCH((dw_s2_wave_bww_kernel<LW_, LH_, 1, T>), grid, block, 0, st, x, in_scale, in_shift,  \
                                dy, partials, C, D, OD, wp.nslabs, N); break;                                     \
  }
  if (wp.logw4 == 1) { MSL_DW_WAVE_BWW2(1, 2) }
  else if (wp.logw4 == 2) { MSL_DW_WAVE_BWW2(2, 3) }
  else if (wp.logw4 == 3) { MSL_DW_WAVE_BWW2(3, 4) }
  else { MSL_DW_WAVE_BWW2(4, 5) }
#undef MSL_DW_WAVE_BWW2
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // namespace

extern "C" {

int msl_dwconv_fwd_num_partials(int N, int C, int D, int H, int W, int stride) {
  const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
  if (wp.ok) return N * wp.nslabs * (wp.wpp > 4 ? wp.wpp / 4 : 1);
  return make_plan(N, C, D, H, W, stride).num_partials;
}

// variant actually chosen for a shape (0 naive, 1 stream, 2 resident, 3 wave) — for tests / DESIGN.md
int msl_dwconv_fwd_variant(int N, int C, int D, int H, int W, int stride) {
  if (make_wave_plan(N, C, D, H, W, stride).ok) return 3;
  return make_plan(N, C, D, H, W, stride).variant;
}

static const msl::BnFold nofold{nullptr, 0, 0, 1.0, nullptr, nullptr, 0.f};

// ---- bf16 storage on the wave kernels (the bf16 activation path, bf16.hip, tries these first) ---------------------------
// number of statistics / weight-gradient partials per channel, or MSL_ERR_UNSUPPORTED when the shape has no wave plan
int msl_dwconv_wave_num_partials(int N, int C, int D, int H, int W, int stride) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
  return wp.ok ? N * wp.nslabs * (wp.wpp > 4 ? wp.wpp / 4 : 1) : MSL_ERR_UNSUPPORTED;
}

// 1 when a statistics-free (eval-mode) forward of this shape runs on dw_s2_rows_eval_kernel / dw_s1_rows_eval_kernel (fp32: inside
// msl_dwconv_fwd; bf16: msl_dwconv_fwd_wave_bf16 with partials == NULL), else 0
int msl_dwconv_fwd_eval_rows_ok(int N, int C, int D, int H, int W, int stride) {
  return (N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && rows_eval_ok(N, C, D, H, W, stride)) ? 1 : 0;
}

// x (N,C,D,H,W) bf16 raw (+ input affine) -> y bf16 raw (+ fp64 statistics partials [2][C][NP] from the fp32 accumulators);
// flip: reversed taps (stride-1 bwd-data), accumulate: y += result
int msl_dwconv_fwd_wave_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                             double* partials, int N, int C, int D, int H, int W, int stride, int flip, int accumulate,
                             void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
  if (!wp.ok && !partials && !flip && !accumulate && rows_eval_ok(N, C, D, H, W, stride)) {  // eval-mode forward
    launch_rows_eval(reinterpret_cast<const dwu16*>(x), in_scale, in_shift, w, reinterpret_cast<dwu16*>(y), N, C, D, H, W,
                     stride, (hipStream_t)stream);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  if (!wp.ok || (stride == 2 && (flip || accumulate))) return MSL_ERR_UNSUPPORTED;
  if (stride == 1)
    launch_wave(wp, reinterpret_cast<const dwu16*>(x), in_scale, in_shift, w, reinterpret_cast<dwu16*>(y), partials, N, C, D, flip,
                accumulate, nofold, (hipStream_t)stream);
  else
    launch_wave_s2(wp, reinterpret_cast<const dwu16*>(x), in_scale, in_shift, w, reinterpret_cast<dwu16*>(y), partials, N, C, D,
                   nofold, (hipStream_t)stream);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ... with the input BatchNorm folded from its producer's statistics partials [2][C][in_np] (same arithmetic and summation
// order as msl_bn_finalize -> the same bits), so that no finalize launch sits between producer and consumer
int msl_dwconv_fwd_wave_bf16_fold(const void* x, const double* in_partials, int in_np, double in_count, const float* gamma,
                                  const float* beta, float eps, const float* w, void* y, double* partials, int N, int C, int D,
                                  int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !in_partials || in_np <= 0) return MSL_ERR_ARG;
  const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
  if (!wp.ok) return MSL_ERR_UNSUPPORTED;
  const msl::BnFold fold{in_partials, in_np, C, in_count, gamma, beta, eps};
  if (stride == 1)
    launch_wave(wp, reinterpret_cast<const dwu16*>(x), (const float*)nullptr, (const float*)nullptr, w, reinterpret_cast<dwu16*>(y),
                partials, N, C, D, 0, 0, fold, (hipStream_t)stream);
  else
    launch_wave_s2(wp, reinterpret_cast<const dwu16*>(x), (const float*)nullptr, (const float*)nullptr, w,
                   reinterpret_cast<dwu16*>(y), partials, N, C, D, fold, (hipStream_t)stream);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// dz (N,C,OD,OH,OW), x (N,C,D,H,W) bf16 (+ input affine) -> fp64 partials [C*27][NP], NP = msl_dwconv_wave_num_partials
int msl_dwconv_bwd_weight_wave_bf16(const void* dz, const void* x, const float* in_scale, const float* in_shift,
                                    double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !partials) return MSL_ERR_ARG;
  const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
  if (!wp.ok) return MSL_ERR_UNSUPPORTED;
  launch_wave_bww(wp, stride, reinterpret_cast<const dwu16*>(x), in_scale, in_shift, reinterpret_cast<const dwu16*>(dz), partials,
                  N, C, D, (hipStream_t)stream);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// bwd-weight on the LDS-tiled kernels (MODE 1).  Returns MSL_ERR_UNSUPPORTED for shapes on the generic path.
// partials: fp64 [C*27][NP], NP = msl_dwconv_bwd_weight_num_partials().

int msl_dwconv_bwd_weight_tiled(const float* dy, const float* x, const float* in_scale, const float* in_shift,
                                double* partials, int N, int C, int D, int H, int W, int stride, void* stream) {
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  if (wave_bww_enabled()) {
    const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
    if (wp.ok) {
      launch_wave_bww(wp, stride, x, in_scale, in_shift, dy, partials, N, C, D, (hipStream_t)stream);
      MSL_LAUNCH_CHECK();
      return MSL_OK;
    }
  }
  DwPlan pl = make_plan(N, C, D, H, W, stride);
  // many channels per workgroup (tiny tail volumes): the per-channel epilogue reduction dominates and the
  // barrier-free wave-per-item kernel of dwconv_bwd.hip is faster
  if (pl.variant == 0 || pl.G > 4) return MSL_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nblocks = N * (C / pl.G) * pl.nslabs;
  const size_t lds = std::max(pl.lds_bytes, (size_t)27 * 256 * sizeof(float));
  float* dyp = const_cast<float*>(dy);
  if (pl.variant == 1) {
#define MSL_DW_BWW(S_, I_, L_)                                                                                  \
  do {                                                                                                          \
    int e_ = set_lds(dw_fwd_stream_kernel<S_, I_, L_, 1>, lds);                                                 \
    if (e_) return e_;                                                                                          \
    MSL_LAUNCH((dw_fwd_stream_kernel<S_, I_, L_, 1>), dim3(nblocks), dim3(256), lds, st, x, in_scale,   \
                       in_shift, nullptr, dyp, partials, C, D, H, W, OD, OH, OW, pl.SLAB, pl.nslabs, N, nofold); \
  } while (0)
    if (stride == 2) {
      if (pl.ipt == 1 && pl.lpt == 4) MSL_DW_BWW(2, 1, 4);
      else MSL_DW_BWW(2, 4, 12);
    } else {
      if (pl.ipt == 1 && pl.lpt == 4) MSL_DW_BWW(1, 1, 4);
      else MSL_DW_BWW(1, 4, 12);
    }
#undef MSL_DW_BWW
  } else {
    if (stride == 2) {
      int e_ = set_lds(dw_fwd_resident_kernel<2, 1>, lds);
      if (e_) return e_;
      MSL_LAUNCH((dw_fwd_resident_kernel<2, 1>), dim3(nblocks), dim3(256), lds, st, x, in_scale, in_shift,
                         nullptr, dyp, partials, C, D, H, W, OD, OH, OW, pl.G, pl.SLAB, pl.nslabs, N, 0, 0, nofold);
    } else {
      int e_ = set_lds(dw_fwd_resident_kernel<1, 1>, lds);
      if (e_) return e_;
      MSL_LAUNCH((dw_fwd_resident_kernel<1, 1>), dim3(nblocks), dim3(256), lds, st, x, in_scale, in_shift,
                         nullptr, dyp, partials, C, D, H, W, OD, OH, OW, pl.G, pl.SLAB, pl.nslabs, N, 0, 0, nofold);
    }
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_dwconv_bwd_weight_tiled_num_partials(int N, int C, int D, int H, int W, int stride) {
  if (wave_bww_enabled()) {
    const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
    if (wp.ok) return wave_bww_num_partials(wp, N);
  }
  DwPlan pl = make_plan(N, C, D, H, W, stride);
  return (pl.variant == 0 || pl.G > 4) ? -1 : pl.num_partials;
}

// Stride-1 bwd-data is the forward convolution of dy with the taps reversed (w[26-k]); reuse the LDS-resident
// forward kernel.  Returns MSL_ERR_UNSUPPORTED when the shape is not on the resident fast path.
int msl_dwconv_s1_bwd_data_resident(const float* dy, const float* w, float* g_in, int N, int C, int D, int H, int W,
                                    int accumulate, void* stream) {
  const WavePlan wp = make_wave_plan(N, C, D, H, W, 1);
  if (wp.ok) {
    launch_wave(wp, dy, nullptr, nullptr, w, g_in, nullptr, N, C, D, 1, accumulate, nofold, (hipStream_t)stream);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  DwPlan pl = make_plan(N, C, D, H, W, 1);
  if (pl.variant != 2) return MSL_ERR_UNSUPPORTED;
  const int nblocks = N * (C / pl.G) * pl.nslabs;
  int e_ = set_lds(dw_fwd_resident_kernel<1, 0>, pl.lds_bytes);
  if (e_) return e_;
  MSL_LAUNCH((dw_fwd_resident_kernel<1, 0>), dim3(nblocks), dim3(256), pl.lds_bytes, (hipStream_t)stream, dy,
                     nullptr, nullptr, w, g_in, nullptr, C, D, H, W, D, H, W, pl.G, pl.SLAB, pl.nslabs, N, 1, accumulate, nofold);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// bf16 storage, statistics-free forward of a map of at most 512 voxels (dw_small_eval_kernel); MSL_ERR_UNSUPPORTED if larger
int msl_dwconv_fwd_small_eval_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y, int N,
                                   int C, int D, int H, int W, int stride, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  if (D * H * W > DW_SMALL_VOX || (D + 2) * (H + 2) * (W + 2) > DW_SMALL_IMG) return MSL_ERR_UNSUPPORTED;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  MSL_LAUNCH(dw_small_eval_kernel<dwu16>, dim3(msl::cdiv(N * C, 4)), dim3(256), 0, (hipStream_t)stream, (const dwu16*)x,
             in_scale, in_shift, w, (dwu16*)y, N * C, C, D, H, W, OD, OH, OW, stride);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

static int dwconv_fwd_impl(const float* x, const float* in_scale, const float* in_shift, const msl::BnFold& fold,
                           const float* w, float* y, double* partials, int N, int C, int D, int H, int W, int stride,
                           int force_naive, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return MSL_ERR_ARG;
  const int OD = (D - 1) / stride + 1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (!force_naive) {
    const WavePlan wp = make_wave_plan(N, C, D, H, W, stride);
    if (wp.ok) {
      if (stride == 2) launch_wave_s2(wp, x, in_scale, in_shift, w, y, partials, N, C, D, fold, st);
      else launch_wave(wp, x, in_scale, in_shift, w, y, partials, N, C, D, 0, 0, fold, st);
      MSL_LAUNCH_CHECK();
      return MSL_OK;
    }
    if (!partials && !fold.partials && rows_eval_ok(N, C, D, H, W, stride)) {  // eval-mode forward, no statistics
      launch_rows_eval(x, in_scale, in_shift, w, y, N, C, D, H, W, stride, st);
      MSL_LAUNCH_CHECK();
      return MSL_OK;
    }
  }
  if (!force_naive && !partials && !fold.partials && D * H * W <= DW_SMALL_VOX && (D + 2) * (H + 2) * (W + 2) <= DW_SMALL_IMG) {
    MSL_LAUNCH(dw_small_eval_kernel<float>, dim3(msl::cdiv(N * C, 4)), dim3(256), 0, st, x, in_scale, in_shift, w, y, N * C, C,
               D, H, W, OD, OH, OW, stride);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
  DwPlan pl = make_plan(N, C, D, H, W, stride);
  if (force_naive && pl.variant != 0) {
    pl.variant = 0;
    pl.num_partials = N * msl::cdiv(OD * OH * OW, STATS_CHUNK);
  }
  if (pl.variant == 0) {
    const int S = OD * OH * OW;
    MSL_LAUNCH(dw_fwd_naive_kernel, dim3(std::min(msl::cdiv(S, 256), 256), N * C), dim3(256), 0, st, x,
                       in_scale, in_shift, w, y, C, D, H, W, OD, OH, OW, stride, fold);
    MSL_LAUNCH_CHECK();
    if (partials) {
      const int chunks = msl::cdiv(S, STATS_CHUNK);
      MSL_LAUNCH(channel_stats_kernel, dim3(chunks, C, N), dim3(256), 0, st, y, partials, C, S, chunks);
      MSL_LAUNCH_CHECK();
    }
    return MSL_OK;
  }
  const int nblocks = N * (C / pl.G) * pl.nslabs;
  if (pl.variant == 1) {
#define MSL_DW_STREAM(S_, I_, L_)                                                                              \
  do {                                                                                                         \
    int e_ = set_lds(dw_fwd_stream_kernel<S_, I_, L_, 0>, pl.lds_bytes);                                          \
    if (e_) return e_;                                                                                         \
    MSL_LAUNCH((dw_fwd_stream_kernel<S_, I_, L_, 0>), dim3(nblocks), dim3(256), pl.lds_bytes, st, x,      \
                       in_scale, in_shift, w, y, partials, C, D, H, W, OD, OH, OW, pl.SLAB, pl.nslabs, N, fold);  \
  } while (0)
    if (stride == 2) {
      if (pl.ipt == 1 && pl.lpt == 4) MSL_DW_STREAM(2, 1, 4);
      else MSL_DW_STREAM(2, 4, 12);
    } else {
      if (pl.ipt == 1 && pl.lpt == 4) MSL_DW_STREAM(1, 1, 4);
      else MSL_DW_STREAM(1, 4, 12);
    }
#undef MSL_DW_STREAM
  } else {
    if (stride == 2) {
      int e_ = set_lds(dw_fwd_resident_kernel<2, 0>, pl.lds_bytes);
      if (e_) return e_;
      MSL_LAUNCH((dw_fwd_resident_kernel<2, 0>), dim3(nblocks), dim3(256), pl.lds_bytes, st, x, in_scale,
                         in_shift, w, y, partials, C, D, H, W, OD, OH, OW, pl.G, pl.SLAB, pl.nslabs, N, 0, 0, fold);
    } else {
      int e_ = set_lds(dw_fwd_resident_kernel<1, 0>, pl.lds_bytes);
      if (e_) return e_;
      MSL_LAUNCH((dw_fwd_resident_kernel<1, 0>), dim3(nblocks), dim3(256), pl.lds_bytes, st, x, in_scale,
                         in_shift, w, y, partials, C, D, H, W, OD, OH, OW, pl.G, pl.SLAB, pl.nslabs, N, 0, 0, fold);
    }
  }
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_dwconv_fwd(const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                   double* partials, int N, int C, int D, int H, int W, int stride, int force_naive,
                   void* stream) {
  return dwconv_fwd_impl(x, in_scale, in_shift, nofold, w, y, partials, N, C, D, H, W, stride, force_naive, stream);
}

// Same, but the input's BatchNorm affine is folded inside the kernel from the producer's statistic partials
// (in_partials [2][C][in_np], element count in_count, gamma/beta/eps of that BatchNorm): no finalize launch in between.
int msl_dwconv_fwd_fold(const float* x, const double* in_partials, int in_np, double in_count, const float* gamma,
                        const float* beta, float eps, const float* w, float* y, double* partials, int N, int C, int D,
                        int H, int W, int stride, void* stream) {
  if (!in_partials || in_np <= 0) return MSL_ERR_ARG;
  const msl::BnFold fold{in_partials, in_np, C, in_count, gamma, beta, eps};
  return dwconv_fwd_impl(x, nullptr, nullptr, fold, w, y, partials, N, C, D, H, W, stride, 0, stream);
}

}  // extern "C"
