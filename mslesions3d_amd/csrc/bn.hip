// BatchNorm3d (train-mode batch statistics) pieces of the MobileNet-3D backbone.
// Reference semantics: nn.BatchNorm3d + ReLU as used in lesions3d/mobilenet.py:26-31,:39-45
// (eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for the running estimate).
//
// Design: the conv kernels emit per-workgroup partial (sum, sum of squares) per channel in fp64
// (`partials[2][C][NP]`); `bn_finalize` folds them in a fixed order into the per-channel affine
// (scale = gamma*invstd, shift = beta - mean*scale) that the NEXT kernel applies while loading
// (relu(fma(x, scale, shift))).  The normalised activation is never written to HBM except for the
// three feature maps the detection heads read.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(64) void bn_finalize_kernel(msl::BnFold f, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var,
                                                         long long* __restrict__ num_batches_tracked, float momentum,
                                                         float* __restrict__ scale, float* __restrict__ shift,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float sc, sh, mu, is;
  double var;
  if (f.NP <= 64) {
    if (lane != 0) return;
    msl::bn_fold_serial(f, c, sc, sh, mu, is, var);
  } else {
    msl::bn_fold_wave(f, c, sc, sh, mu, is, var);
    if (lane != 0) return;
  }
  scale[c] = sc;
  shift[c] = sh;
  save_mean[c] = mu;
  save_invstd[c] = is;
  if (running_mean) {
    const float unbiased = (float)(f.count > 1.0 ? var * f.count / (f.count - 1.0) : var);
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
  }
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

// All BatchNorm layers of the network in one launch: entry e of the table describes one layer; workgroup b serves
// channel (b - first_block[e]) of the entry it falls into.
struct BnFinalizeEntry {
  msl::BnFold fold;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  float momentum;
  float* scale;
  float* shift;
  float* save_mean;
  float* save_invstd;
  int first_block;
};

__global__ __launch_bounds__(64) void bn_finalize_batch_kernel(const BnFinalizeEntry* __restrict__ table, int n_entries) {
  int e = 0;
  while (e + 1 < n_entries && (int)blockIdx.x >= table[e + 1].first_block) ++e;
  const BnFinalizeEntry t = table[e];
  const int c = blockIdx.x - t.first_block, lane = threadIdx.x;
  float sc, sh, mu, is;
  double var;
  if (t.fold.NP <= 64) {
    if (lane != 0) return;
    msl::bn_fold_serial(t.fold, c, sc, sh, mu, is, var);
  } else {
    msl::bn_fold_wave(t.fold, c, sc, sh, mu, is, var);
    if (lane != 0) return;
  }
  t.scale[c] = sc;
  t.shift[c] = sh;
  t.save_mean[c] = mu;
  t.save_invstd[c] = is;
  if (t.running_mean) {
    const float unbiased = (float)(t.fold.count > 1.0 ? var * t.fold.count / (t.fold.count - 1.0) : var);
    t.running_mean[c] = (1.0f - t.momentum) * t.running_mean[c] + t.momentum * mu;
    t.running_var[c] = (1.0f - t.momentum) * t.running_var[c] + t.momentum * unbiased;
  }
  if (c == 0 && t.num_batches_tracked) *t.num_batches_tracked += 1;
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean,
                                      const float* __restrict__ running_var, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

// every BatchNorm of the network in one launch (table as for bn_finalize_batch_kernel; one thread per channel): the
// arithmetic of bn_eval_affine_kernel, bit for bit
__global__ __launch_bounds__(128) void bn_eval_affine_batch_kernel(const BnFinalizeEntry* __restrict__ table, int n_entries,
                                                                   int total_channels) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= total_channels) return;
  int e = 0;
  while (e + 1 < n_entries && i >= table[e + 1].first_block) ++e;
  const BnFinalizeEntry& t = table[e];
  const int c = i - t.first_block;
  const float invstd = 1.0f / sqrtf(t.running_var[c] + t.fold.eps);
  const float sc = t.fold.gamma[c] * invstd;
  t.scale[c] = sc;
  t.shift[c] = t.fold.beta[c] - t.running_mean[c] * sc;
}

// out = relu(y*scale+shift), written in plain NCDHW and/or in a zero-haloed layout
// (N,C,D+2,H+2,W+2) that lets the head convolutions run without bounds checks.
__global__ __launch_bounds__(256) void bn_relu_materialize_kernel(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, msl::BnFold fold,
    float* __restrict__ out_plain, float* __restrict__ out_pad, int C, int D, int H, int W) {
  __shared__ float s_aff[2];
  const int S = D * H * W;
  const int nc = blockIdx.y;  // n*C + c
  const int c = nc % C;
  float sc, sh;
  if (fold.partials) {
    msl::bn_fold_block(fold, c, 1, &s_aff[0], &s_aff[1]);
    sc = s_aff[0];
    sh = s_aff[1];
  } else {
    sc = scale[c];
    sh = shift[c];
  }
  const size_t base = (size_t)nc * S;
  const int Hp = H + 2, Wp = W + 2;
  const size_t pbase = (size_t)nc * (D + 2) * Hp * Wp;
  if ((W & 3) == 0) {  // four consecutive voxels of a row per thread: one 16-byte load, one index decode
    const int S4 = S >> 2, W4 = W >> 2;
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < S4; i4 += gridDim.x * blockDim.x) {
      const float4 r = *reinterpret_cast<const float4*>(y + base + (size_t)i4 * 4);
      const float4 v = make_float4(msl::act(r.x, sc, sh), msl::act(r.y, sc, sh), msl::act(r.z, sc, sh), msl::act(r.w, sc, sh));
      if (out_plain) *reinterpret_cast<float4*>(out_plain + base + (size_t)i4 * 4) = v;
      if (out_pad) {
        const int w = (i4 % W4) * 4, h = (i4 / W4) % H, d = i4 / (W4 * H);
        float* o = out_pad + pbase + ((size_t)(d + 1) * Hp + (h + 1)) * Wp + (w + 1);  // (+1: not 16-byte aligned)
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      }
    }
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
    const float v = msl::act(y[base + i], sc, sh);
    if (out_plain) out_plain[base + i] = v;
    if (out_pad) {
      const int w = i % W, h = (i / W) % H, d = i / (W * H);
      out_pad[pbase + ((size_t)(d + 1) * Hp + (h + 1)) * Wp + (w + 1)] = v;
    }
  }
}

// ---- backward ---------------------------------------------------------------------------------
// For a_out = relu(bn(y)), given g = dL/da_out:
//   gm = g * [a_out > 0];  dbeta = sum gm;  dgamma = sum gm * xhat,  xhat = (y - mean) * invstd
//   dL/dy = gamma*invstd * (gm - dbeta/n - xhat * dgamma/n)
constexpr int BWD_CHUNK = 4096;  // elements of one (n,c) row handled by one workgroup

__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    double* __restrict__ partials, int C, int S, int chunks) {
  __shared__ double scratch[8];
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const size_t base = ((size_t)n * C + c) * S;
  const int lo = chunk * BWD_CHUNK, hi = min(S, lo + BWD_CHUNK);
  float s1 = 0.f, s2 = 0.f;
  if ((S & 3) == 0) {
    // BWD_CHUNK = 4 x 1024: the thread's four float4 pairs are loaded back to back (clamped addresses, masked)
    float4 gv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = lo + threadIdx.x * 4 + u * 1024;
      const size_t o = base + (i < hi ? i : lo);
      gv[u] = *reinterpret_cast<const float4*>(g + o);
      yv[u] = *reinterpret_cast<const float4*>(y + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = lo + threadIdx.x * 4 + u * 1024 < hi;
      const float ga[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, ya[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gm = (in && fmaf(ya[k], sc, sh) > 0.f) ? ga[k] : 0.f;
        s1 += gm;
        s2 += gm * ((ya[k] - mu) * is);
      }
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      const float yy = y[base + i];
      const float gm = fmaf(yy, sc, sh) > 0.f ? g[base + i] : 0.f;
      s1 += gm;
      s2 += gm * ((yy - mu) * is);
    }
  }
  const int NP = gridDim.z * chunks;
  const int p = n * chunks + chunk;
  const double t1 = msl::block_sum((double)s1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum((double)s2, scratch);
  if (threadIdx.x == 0) {
    partials[(size_t)c * NP + p] = t1;
    partials[((size_t)C + c) * NP + p] = t2;
  }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ partials, int NP,
                                                             double count, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta,
                                                             float* __restrict__ c1, float* __restrict__ c2,
                                                             int C, const float* __restrict__ fwd_vec,
                                                             float* __restrict__ coef) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const double* ps = partials + (size_t)c * NP;
  const double* pq = partials + ((size_t)C + c) * NP;
  double s = 0.0, q = 0.0;
  int p = lane;
  for (; p + 7 * 64 < NP; p += 8 * 64) {  // 16 loads in flight
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = ps[p + u * 64];
      b[u] = pq[p + u * 64];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += a[u];
      q += b[u];
    }
  }
  for (; p < NP; p += 64) {
    s += ps[p];
    q += pq[p];
  }
  s = msl::wave_sum(s);
  q = msl::wave_sum(q);
  if (lane == 0) {
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    const float k1 = (float)(s / count), k2 = (float)(q / count);
    c1[c] = k1;
    c2[c] = k2;
    if (coef) {
      // dL/dy = scale * (gm - c1 - xhat * c2),  xhat = (y - mean) * invstd   ==   scale * gm + (cC * y + cE)
      const float sc = fwd_vec[c], mu = fwd_vec[2 * C + c], is = fwd_vec[3 * C + c];
      const float t = sc * is * k2;
      coef[c] = -t;
      coef[C + c] = fmaf(t, mu, -sc * k1);
    }
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ c1, const float* __restrict__ c2, float* __restrict__ dy, int C, int S,
    const double* __restrict__ partials, int NP, double count, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ c1_out, float* __restrict__ c2_out) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  float k1, k2;
  if (partials) {
    // finalize folded in (NP is small here): every workgroup of the channel sums the same partials in the same order
    // (wave 0, lanes strided, DPP tree), one of them publishes dgamma / dbeta / c1 / c2 - no separate launch on the chain
    __shared__ float kk[2];
    if (threadIdx.x < 64) {
      double s = 0.0, q = 0.0;
      for (int p = threadIdx.x; p < NP; p += 64) {
        s += partials[(size_t)c * NP + p];
        q += partials[((size_t)C + c) * NP + p];
      }
      s = msl::wave_sum(s);
      q = msl::wave_sum(q);
      if (threadIdx.x == 0) {
        kk[0] = (float)(s / count);
        kk[1] = (float)(q / count);
        if (blockIdx.x == 0 && n == 0) {
          dbeta[c] = (float)s;
          dgamma[c] = (float)q;
          c1_out[c] = kk[0];
          c2_out[c] = kk[1];
        }
      }
    }
    __syncthreads();
    k1 = kk[0];
    k2 = kk[1];
  } else {
    k1 = c1[c];
    k2 = c2[c];
  }
  const size_t base = ((size_t)n * C + c) * S;
  if ((S & 3) == 0) {
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < S; i += gridDim.x * 256 * 4) {
      const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
      const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
      const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gm = fmaf(ya[k], sc, sh) > 0.f ? ga[k] : 0.f;
        o[k] = sc * (gm - k1 - ((ya[k] - mu) * is) * k2);
      }
      *reinterpret_cast<float4*>(dy + base + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
      const float yy = y[base + i];
      const float gm = fmaf(yy, sc, sh) > 0.f ? g[base + i] : 0.f;
      dy[base + i] = sc * (gm - k1 - ((yy - mu) * is) * k2);
    }
  }
}

// Whole BatchNorm+ReLU backward of one channel in ONE workgroup (reduce -> coefficients -> apply) for the layers
// whose per-channel data is small (<= 64 K elements: blocks 2-7).  Replaces three launches (two of them pure
// launch latency) by one; the second pass re-reads g and y from L2.
__global__ __launch_bounds__(256) void bn_relu_bwd_fused_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dy, int N, int C, int S, double count) {
  __shared__ double scratch[8];
  __shared__ float coef[2];
  const int c = blockIdx.x;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const bool vec = (S & 3) == 0;
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * S;
    if (vec) {
      for (int i = threadIdx.x * 4; i < S; i += 1024) {
        const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
        const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gm = fmaf(ya[k], sc, sh) > 0.f ? ga[k] : 0.f;
          s1 += gm;
          s2 += gm * ((ya[k] - mu) * is);
        }
      }
    } else {
      for (int i = threadIdx.x; i < S; i += 256) {
        const float yy = y[base + i];
        const float gm = fmaf(yy, sc, sh) > 0.f ? g[base + i] : 0.f;
        s1 += gm;
        s2 += gm * ((yy - mu) * is);
      }
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
    if (vec) {
      for (int i = threadIdx.x * 4; i < S; i += 1024) {
        const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
        const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gm = fmaf(ya[k], sc, sh) > 0.f ? ga[k] : 0.f;
          o[k] = sc * (gm - k1 - ((ya[k] - mu) * is) * k2);
        }
        *reinterpret_cast<float4*>(dy + base + i) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else {
      for (int i = threadIdx.x; i < S; i += 256) {
        const float yy = y[base + i];
        const float gm = fmaf(yy, sc, sh) > 0.f ? g[base + i] : 0.f;
        dy[base + i] = sc * (gm - k1 - ((yy - mu) * is) * k2);
      }
    }
  }
}

// Register-resident form of the above for N*S <= NT*IPT*4 elements per channel (S % 4 == 0): the channel's g and y are
// read ONCE, all loads of a thread in flight together, kept in registers across the reduction, and dL/dy is written from
// them - one memory round trip, a block-wide reduction, one burst of stores.
template <int NT, int IPT>
__global__ __launch_bounds__(NT) void bn_relu_bwd_fused_reg_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dy, int N, int C, int S, double count) {
  __shared__ double scratch[16];
  __shared__ float coef[2];
  const int c = blockIdx.x;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const int S4 = S >> 2, total4 = N * S4;
  float4 gv[IPT], yv[IPT];
  size_t off[IPT];
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    const int t = threadIdx.x + u * NT;
    const int tt = t < total4 ? t : 0;
    const int n = tt / S4, i4 = tt - n * S4;
    off[u] = ((size_t)n * C + c) * S + (size_t)i4 * 4;
    gv[u] = *reinterpret_cast<const float4*>(g + off[u]);
    yv[u] = *reinterpret_cast<const float4*>(y + off[u]);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < IPT; ++u) {
    const bool in = threadIdx.x + u * NT < total4;
    float* ga = reinterpret_cast<float*>(&gv[u]);
    const float* ya = reinterpret_cast<const float*>(&yv[u]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ga[k] = (in && fmaf(ya[k], sc, sh) > 0.f) ? ga[k] : 0.f;  // gm
      s1 += ga[k];
      s2 += ga[k] * ((ya[k] - mu) * is);
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
    if (threadIdx.x + u * NT < total4) {
      const float* ga = reinterpret_cast<const float*>(&gv[u]);
      const float* ya = reinterpret_cast<const float*>(&yv[u]);
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = sc * (ga[k] - k1 - ((ya[k] - mu) * is) * k2);
      *reinterpret_cast<float4*>(dy + off[u]) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

}  // namespace

extern "C" {

int msl_bn_finalize(const double* partials, int num_partials, double count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, int C, void* stream) {
  if (C <= 0 || num_partials <= 0) return MSL_ERR_ARG;
  const msl::BnFold f{partials, num_partials, C, count, gamma, beta, eps};
  MSL_LAUNCH(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, f, running_mean, running_var,
                     num_batches_tracked, momentum, scale, shift, save_mean, save_invstd);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// ---- batched finalize: the host fills a table (one entry per BatchNorm) in HOST memory with msl_bn_finalize_table_set,
// copies it to the device once, and launches every layer's finalize with one call per step.
size_t msl_bn_finalize_entry_bytes(void) { return sizeof(BnFinalizeEntry); }

int msl_bn_finalize_table_set(void* host_table, int index, int first_block, const double* partials, int num_partials,
                              double count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, long long* num_batches_tracked, float momentum, float eps,
                              float* scale, float* shift, float* save_mean, float* save_invstd, int C) {
  if (!host_table || index < 0 || C <= 0 || num_partials <= 0) return MSL_ERR_ARG;
  BnFinalizeEntry& t = reinterpret_cast<BnFinalizeEntry*>(host_table)[index];
  t.fold = msl::BnFold{partials, num_partials, C, count, gamma, beta, eps};
  t.running_mean = running_mean;
  t.running_var = running_var;
  t.num_batches_tracked = num_batches_tracked;
  t.momentum = momentum;
  t.scale = scale;
  t.shift = shift;
  t.save_mean = save_mean;
  t.save_invstd = save_invstd;
  t.first_block = first_block;
  return MSL_OK;
}

int msl_bn_finalize_batch(const void* device_table, int n_entries, int total_channels, void* stream) {
  if (!device_table || n_entries <= 0 || total_channels <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_finalize_batch_kernel, dim3(total_channels), dim3(64), 0, (hipStream_t)stream,
                     reinterpret_cast<const BnFinalizeEntry*>(device_table), n_entries);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// eval-mode (scale, shift) of every BatchNorm listed in the table (msl_bn_finalize_table_set) in ONE launch
int msl_bn_eval_affine_batch(const void* device_table, int n_entries, int total_channels, void* stream) {
  if (!device_table || n_entries <= 0 || total_channels <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_eval_affine_batch_kernel, dim3(msl::cdiv(total_channels, 128)), dim3(128), 0, (hipStream_t)stream,
                     reinterpret_cast<const BnFinalizeEntry*>(device_table), n_entries, total_channels);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, int C, void* stream) {
  if (C <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_eval_affine_kernel, dim3(msl::cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma,
                     beta, running_mean, running_var, eps, scale, shift, C);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_relu_materialize(const float* y, const float* scale, const float* shift, float* out_plain,
                            float* out_pad, int N, int C, int D, int H, int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MSL_ERR_ARG;
  const int S = D * H * W;
  dim3 grid(min(msl::cdiv((W & 3) == 0 ? S / 4 : S, 256), 64), N * C);
  const msl::BnFold nofold{nullptr, 0, C, 1.0, nullptr, nullptr, 0.f};
  MSL_LAUNCH(bn_relu_materialize_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, nofold,
                     out_plain, out_pad, C, D, H, W);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// same, but the affine is folded from the producer's statistic partials inside the kernel (see common.hpp BnFold)
int msl_bn_relu_materialize_fold(const float* y, const double* partials, int num_partials, double count,
                                 const float* gamma, const float* beta, float eps, float* out_plain, float* out_pad,
                                 int N, int C, int D, int H, int W, void* stream) {
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || num_partials <= 0) return MSL_ERR_ARG;
  const int S = D * H * W;
  dim3 grid(min(msl::cdiv((W & 3) == 0 ? S / 4 : S, 256), 64), N * C);
  const msl::BnFold f{partials, num_partials, C, count, gamma, beta, eps};
  MSL_LAUNCH(bn_relu_materialize_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, nullptr, nullptr, f,
                     out_plain, out_pad, C, D, H, W);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_relu_bwd_num_partials(int N, int S) { return N * msl::cdiv(S, BWD_CHUNK); }

int msl_bn_relu_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift,
                           const float* mean, const float* invstd, double* partials, int N, int C, int S,
                           void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  const int chunks = msl::cdiv(S, BWD_CHUNK);
  MSL_LAUNCH(bn_relu_bwd_reduce_kernel, dim3(chunks, C, N), dim3(256), 0, (hipStream_t)stream, g, y,
                     scale, shift, mean, invstd, partials, C, S, chunks);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_bwd_finalize(const double* partials, int num_partials, double count, float* dgamma, float* dbeta,
                        float* c1, float* c2, int C, void* stream) {
  if (C <= 0 || num_partials <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partials, num_partials,
                     count, dgamma, dbeta, c1, c2, C, (const float*)nullptr, (float*)nullptr);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// Same, on a whole (8, C) BatchNorm vector block [scale, shift, mean, invstd | c1, c2, cC, cE]: rows 4-7 are written;
// cC, cE fold the backward into  dL/dy = scale * gm + (cC * y + cE)  for consumers that apply it while loading.
int msl_bn_bwd_finalize_coef(const double* partials, int num_partials, double count, float* dgamma, float* dbeta,
                             float* bn_vec, int C, void* stream) {
  if (C <= 0 || num_partials <= 0 || !bn_vec) return MSL_ERR_ARG;
  MSL_LAUNCH(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partials, num_partials,
                     count, dgamma, dbeta, bn_vec + 4 * C, bn_vec + 5 * C, C, (const float*)bn_vec, bn_vec + 6 * C);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// single-launch variant (reduce + finalize + apply); intended for N*S <= 65536 elements per channel
int msl_bn_relu_bwd_fused(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, float* dgamma, float* dbeta, float* dy, int N, int C, int S,
                          void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total4 = (long long)N * (S >> 2);
#define MSL_BN_REG(NT_, IPT_)                                                                                          \
  MSL_LAUNCH((bn_relu_bwd_fused_reg_kernel<NT_, IPT_>), dim3(C), dim3(NT_), 0, st, g, y, scale, shift, mean, \
                     invstd, dgamma, dbeta, dy, N, C, S, (double)N * S)
  if ((S & 3) == 0 && total4 <= 4096) {
    if (total4 <= 64) MSL_BN_REG(64, 1);
    else if (total4 <= 256) MSL_BN_REG(256, 1);
    else if (total4 <= 512) MSL_BN_REG(256, 2);
    else if (total4 <= 1024) MSL_BN_REG(512, 2);
    else if (total4 <= 2048) MSL_BN_REG(1024, 2);
    else MSL_BN_REG(1024, 4);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
#undef MSL_BN_REG
  MSL_LAUNCH(bn_relu_bwd_fused_kernel, dim3(C), dim3(256), 0, st, g, y, scale, shift, mean,
                     invstd, dgamma, dbeta, dy, N, C, S, (double)N * S);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_bn_relu_bwd_apply(const float* g, const float* y, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* c1, const float* c2, float* dy,
                          int N, int C, int S, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return MSL_ERR_ARG;
  const int gx = min(msl::cdiv(S, 1024), 64);
  MSL_LAUNCH(bn_relu_bwd_apply_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, g, y, scale,
                     shift, mean, invstd, c1, c2, dy, C, S, (const double*)nullptr, 0, 1.0, (float*)nullptr,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// msl_bn_bwd_finalize + msl_bn_relu_bwd_apply in one launch, for reduce partials that are few (<= 256 per channel):
// bn_vec is the (>= 6, C) vector block [scale, shift, mean, invstd, c1, c2, ...]; rows 4, 5 are written.
int msl_bn_relu_bwd_finalize_apply(const double* partials, int num_partials, double count, const float* g, const float* y,
                                   float* bn_vec, float* dgamma, float* dbeta, float* dy, int N, int C, int S,
                                   void* stream) {
  if (N <= 0 || C <= 0 || S <= 0 || num_partials <= 0 || !partials || !bn_vec) return MSL_ERR_ARG;
  const int gx = min(msl::cdiv(S, 1024), 64);
  MSL_LAUNCH(bn_relu_bwd_apply_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, g, y, bn_vec,
                     bn_vec + C, bn_vec + 2 * C, bn_vec + 3 * C, (const float*)nullptr, (const float*)nullptr, dy, C, S,
                     partials, num_partials, count, dgamma, dbeta, bn_vec + 4 * C, bn_vec + 5 * C);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
