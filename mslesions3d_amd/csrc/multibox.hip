// Prior (anchor) generation, box transforms, IoU, prior<->object matching with target encoding, and the
// MultiBox loss (forward + closed-form backward).
// Reference: LSSD3D.create_prior_boxes (lesions3d/ssd3d.py:286-342), box utilities
// (lesions3d/utils.py:42-149), MultiBoxLoss (lesions3d/ssd3d.py:741-941).
//
// Everything here is tiny (P = 9344 priors at 128^3) and latency-bound: the ~40 small torch ops + host
// syncs per image of the reference become 6 launches for the whole batch with no host round trip.
// Integer outputs (matched object per prior, classes) must be bit-exact, so the IoU arithmetic follows the
// reference's operation order with FMA contraction disabled, and ties are resolved explicitly:
// first maximum wins; in the force-match the highest-numbered object wins (last writer, ssd3d.py:865).
#include "common.hpp"
#pragma clang fp contract(off)

namespace {

struct Box {
  float v[6];
};

__device__ __forceinline__ Box load_box(const float* p) {
  Box b;
#pragma unroll
  for (int i = 0; i < 6; ++i) b.v[i] = p[i];
  return b;
}

// utils.py:42-51
__device__ __forceinline__ Box c_to_xyz(const Box& c) {
  Box o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float h = c.v[3 + i] / 2.0f;
    o.v[i] = c.v[i] - h;
    o.v[3 + i] = c.v[i] + h;
  }
  return o;
}
// utils.py:92-102
__device__ __forceinline__ Box xyz_to_c(const Box& b) {
  Box o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.v[i] = (b.v[3 + i] + b.v[i]) / 2.0f;
    o.v[3 + i] = b.v[3 + i] - b.v[i];
  }
  return o;
}
// utils.py:71-89
__device__ __forceinline__ Box encode_box(const Box& c, const Box& p) {
  Box o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.v[i] = (c.v[i] - p.v[i]) / (p.v[3 + i] / 10.0f);
    o.v[3 + i] = logf(c.v[3 + i] / p.v[3 + i]) * 5.0f;
  }
  return o;
}
// utils.py:54-68
__device__ __forceinline__ Box decode_box(const Box& g, const Box& p) {
  Box o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.v[i] = g.v[i] * p.v[3 + i] / 10.0f + p.v[i];
    o.v[3 + i] = expf(g.v[3 + i] / 5.0f) * p.v[3 + i];
  }
  return o;
}
__device__ __forceinline__ float box_vol(const Box& b) {
  return (b.v[3] - b.v[0]) * (b.v[4] - b.v[1]) * (b.v[5] - b.v[2]);
}
// utils.py:105-149
__device__ __forceinline__ float box_inter(const Box& a, const Box& b) {
  float e[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float lo = fmaxf(a.v[i], b.v[i]);
    const float hi = fminf(a.v[3 + i], b.v[3 + i]);
    e[i] = fmaxf(hi - lo, 0.0f);
  }
  return e[0] * e[1] * e[2];
}
__device__ __forceinline__ float box_iou(const Box& a, float vol_a, const Box& b, float vol_b) {
  const float inter = box_inter(a, b);
  const float uni = vol_a + vol_b - inter;
  return inter / uni;
}

// ---- priors -------------------------------------------------------------------------------------
__global__ void make_priors_kernel(float* __restrict__ out, int row_off, int D0, int D1, int D2, double scale,
                                   int bpl) {
  const int total = D0 * D1 * D2 * bpl;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = t % bpl;
  int r = t / bpl;
  const int k = r % D2;
  r /= D2;
  const int j = r % D1, i = r / D1;
  const double cx = ((double)j + 0.5) / D1, cy = ((double)i + 0.5) / D0, cz = ((double)k + 0.5) / D2;  // ssd3d.py:307-309
  const double sz = b == 0 ? scale : scale + scale / (double)b;                                    // ssd3d.py:330-331
  float* o = out + (size_t)(row_off + t) * 6;
  const double vals[6] = {cx, cy, cz, sz, sz, sz};
#pragma unroll
  for (int q = 0; q < 6; ++q) o[q] = fminf(fmaxf((float)vals[q], 0.0f), 1.0f);  // ssd3d.py:337
}

// ---- element-wise box transforms (the utils.* API) --------------------------------------------
__global__ void box_transform_kernel(const float* __restrict__ a, const float* __restrict__ pri, float* __restrict__ out,
                                     int n, int op) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Box x = load_box(a + (size_t)i * 6);
  Box o;
  if (op == 0) o = c_to_xyz(x);
  else if (op == 1) o = xyz_to_c(x);
  else if (op == 2) o = encode_box(x, load_box(pri + (size_t)i * 6));
  else o = decode_box(x, load_box(pri + (size_t)i * 6));
#pragma unroll
  for (int q = 0; q < 6; ++q) out[(size_t)i * 6 + q] = o.v[q];
}

__global__ void iou_matrix_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  int n1, int n2, int inter_only) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n2) return;
  const Box x = load_box(a + (size_t)i * 6), y = load_box(b + (size_t)j * 6);
  out[(size_t)i * n2 + j] = inter_only ? box_inter(x, y) : box_iou(x, box_vol(x), y, box_vol(y));
}

// ---- matching -----------------------------------------------------------------------------------
// best object per prior: first maximum over objects in ascending order (ssd3d.py:801, :833-837)
__global__ __launch_bounds__(256) void match_prior_best_kernel(const float* __restrict__ gt_boxes,
                                                               const int* __restrict__ obj_off,
                                                               const float* __restrict__ priors_c,
                                                               float* __restrict__ overlap, int* __restrict__ obj,
                                                               int P, int* __restrict__ npos) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (npos && n == 0 && p == 0) *npos = 0;  // the positives counter of match_encode_kernel (three launches later)
  if (p >= P) return;
  const int o0 = obj_off[n], o1 = obj_off[n + 1];
  const Box pb = c_to_xyz(load_box(priors_c + (size_t)p * 6));
  const float pv = box_vol(pb);
  float best = 0.f;
  int bi = 0;
  for (int o = o0; o < o1; ++o) {
    const Box g = load_box(gt_boxes + (size_t)o * 6);
    const float v = box_iou(g, box_vol(g), pb, pv);
    if (o == o0 || v > best) {
      best = v;
      bi = o - o0;
    }
  }
  overlap[(size_t)n * P + p] = best;
  obj[(size_t)n * P + p] = bi;
}

// best prior per object: first maximum over priors (ssd3d.py:812).  One workgroup (1024 threads) per object; a thread
// visits priors tid, tid + 1024, ... four at a time with their 24 loads in flight together (the serial form spent 36
// dependent memory round trips per thread: 15 us for a dozen objects).
constexpr int MOB_T = 1024;
__global__ __launch_bounds__(MOB_T) void match_object_best_kernel(const float* __restrict__ gt_boxes,
                                                                  const float* __restrict__ priors_c,
                                                                  int* __restrict__ prior_for_obj, int P) {
  __shared__ float sv[MOB_T];
  __shared__ int si[MOB_T];
  const int o = blockIdx.x;
  const Box g = load_box(gt_boxes + (size_t)o * 6);
  const float gv = box_vol(g);
  float best = 0.f;
  int bi = -1;
  for (int p0 = threadIdx.x; p0 < P; p0 += 4 * MOB_T) {
    Box c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = load_box(priors_c + (size_t)min(p0 + u * MOB_T, P - 1) * 6);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * MOB_T;
      const Box pb = c_to_xyz(c[u]);
      const float v = box_iou(g, gv, pb, box_vol(pb));
      if (p < P && (bi < 0 || v > best)) {  // ascending p within a thread: strict > keeps the first
        best = v;
        bi = p;
      }
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = MOB_T / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v2 = sv[threadIdx.x + s];
      const int i2 = si[threadIdx.x + s];
      const float v1 = sv[threadIdx.x];
      const int i1 = si[threadIdx.x];
      const bool take = i2 >= 0 && (i1 < 0 || v2 > v1 || (v2 == v1 && i2 < i1));
      if (take) {
        sv[threadIdx.x] = v2;
        si[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) prior_for_obj[o] = si[0];
}

// force-match (ssd3d.py:865, :868): every object claims its best prior; duplicates -> highest object wins
__global__ __launch_bounds__(256) void match_force_kernel(const int* __restrict__ obj_off,
                                                          const int* __restrict__ prior_for_obj,
                                                          float* __restrict__ overlap, int* __restrict__ obj, int P) {
  const int n = blockIdx.x;
  const int o0 = obj_off[n], o1 = obj_off[n + 1];
  for (int o = o0 + threadIdx.x; o < o1; o += 256) {
    const int pf = prior_for_obj[o];
    bool winner = true;
    for (int o2 = o + 1; o2 < o1; ++o2) winner = winner && (prior_for_obj[o2] != pf);
    overlap[(size_t)n * P + pf] = 1.0f;
    if (winner) obj[(size_t)n * P + pf] = o - o0;
  }
}

// labels + threshold band + target encoding (ssd3d.py:871-887)
__global__ __launch_bounds__(256) void match_encode_kernel(const float* __restrict__ gt_boxes,
                                                           const long long* __restrict__ gt_labels,
                                                           const int* __restrict__ obj_off,
                                                           const float* __restrict__ priors_c,
                                                           const float* __restrict__ overlap,
                                                           const int* __restrict__ obj, float thr_lo, float thr_hi,
                                                           int soft, long long* __restrict__ true_classes,
                                                           float* __restrict__ true_locs,
                                                           long long* __restrict__ matched, int P, int* __restrict__ npos,
                                                           int npos_reset) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (npos && npos_reset && n == 0 && p == 0) *npos = 0;  // (a batch without objects: nobody adds, nobody zeroed it before)
  if (p >= P) return;
  const size_t idx = (size_t)n * P + p;
  const int o0 = obj_off[n], o1 = obj_off[n + 1];
  if (o1 == o0) {  // ssd3d.py:854-855
    true_classes[idx] = 0;
    if (matched) matched[idx] = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) true_locs[idx * 6 + q] = 0.f;
    return;
  }
  const int o = obj[idx];
  const float ov = overlap[idx];
  long long lab = gt_labels[o0 + o];
  if (ov < thr_lo) lab = 0;
  else if (soft && ov < thr_hi) lab = -1;
  true_classes[idx] = lab;
  if (matched) matched[idx] = o;
  if (npos) {  // number of positive priors of the batch (ssd3d.py:890-893 n_positives.sum()): an exact integer count
    const unsigned long long m = __ballot(lab > 0);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(npos, __popcll(m));
  }
  const Box t = encode_box(xyz_to_c(load_box(gt_boxes + (size_t)(o0 + o) * 6)), load_box(priors_c + (size_t)p * 6));
#pragma unroll
  for (int q = 0; q < 6; ++q) true_locs[idx * 6 + q] = t.v[q];
}

// ---- loss ---------------------------------------------------------------------------------------
constexpr int MAXC = 16;

__global__ __launch_bounds__(256) void multibox_loss_partial_kernel(const float* __restrict__ locs,
                                                                    const float* __restrict__ scores,
                                                                    const long long* __restrict__ true_classes,
                                                                    const float* __restrict__ true_locs,
                                                                    double* __restrict__ partials, int total, int ncls,
                                                                    int* __restrict__ nan_flag) {
  __shared__ double scratch[8];
  double ce = 0.0, l1 = 0.0, np = 0.0;
  bool bad_loc = false, bad_score = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (nan_flag) {  // the NaN guards of ssd3d.py:258-261 for free: this kernel reads every loc and score anyway
      for (int c = 0; c < ncls; ++c) bad_score |= isnan(scores[(size_t)i * ncls + c]);
#pragma unroll
      for (int q = 0; q < 6; ++q) bad_loc |= isnan(locs[(size_t)i * 6 + q]);
    }
    const long long tc = true_classes[i];
    if (tc >= 0) {
      const float* x = scores + (size_t)i * ncls;
      float m = x[0];
      for (int c = 1; c < ncls; ++c) m = fmaxf(m, x[c]);
      float se = 0.f;
      for (int c = 0; c < ncls; ++c) se += expf(x[c] - m);
      ce += (double)((m + logf(se)) - x[tc]);
    }
    if (tc > 0) {
      np += 1.0;
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) a += fabsf(locs[(size_t)i * 6 + q] - true_locs[(size_t)i * 6 + q]);
      l1 += (double)a;
    }
  }
  if (nan_flag) {
    if (__any(bad_loc) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
    if (__any(bad_score) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 2);
  }
  const double t0 = msl::block_sum(ce, scratch);
  __syncthreads();
  const double t1 = msl::block_sum(l1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum(np, scratch);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = t0;
    partials[blockIdx.x * 3 + 1] = t1;
    partials[blockIdx.x * 3 + 2] = t2;
  }
}

// out[0] = conf_loss, out[1] = loc_loss, out[2] = number of positives (as float)
__global__ void multibox_loss_finalize_kernel(const double* __restrict__ partials, int nblocks, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double ce = 0.0, l1 = 0.0, np = 0.0;
  for (int i = 0; i < nblocks; ++i) {
    ce += partials[i * 3 + 0];
    l1 += partials[i * 3 + 1];
    np += partials[i * 3 + 2];
  }
  const float npf = (float)np;
  out[0] = (float)ce / npf;                 // ssd3d.py:933
  out[1] = (float)l1 / (npf * 6.0f);         // nn.L1Loss mean over positives x 6 (ssd3d.py:896); 0/0 -> NaN
  out[2] = npf;
}

template <bool FROM_PARTIALS>
__global__ __launch_bounds__(256) void multibox_loss_bwd_kernel(const float* __restrict__ locs,
                                                                const float* __restrict__ scores,
                                                                const long long* __restrict__ true_classes,
                                                                const float* __restrict__ true_locs,
                                                                float* __restrict__ loss_out,
                                                                const double* __restrict__ partials, int nparts,
                                                                const float* __restrict__ upstream,
                                                                float* __restrict__ dlocs, float* __restrict__ dscores,
                                                                int total, int ncls) {
  __shared__ float s_np;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float npf;
  if (FROM_PARTIALS) {
    // every workgroup folds the (few) partials itself; workgroup 0 also publishes the losses -> no finalize launch
    if (threadIdx.x < 64) {
      double ce = 0.0, l1 = 0.0, np = 0.0;
      for (int k = threadIdx.x; k < nparts; k += 64) {
        ce += partials[k * 3 + 0];
        l1 += partials[k * 3 + 1];
        np += partials[k * 3 + 2];
      }
      ce = msl::wave_sum(ce);
      l1 = msl::wave_sum(l1);
      np = msl::wave_sum(np);
      if (threadIdx.x == 0) {
        const float f = (float)np;
        s_np = f;
        if (blockIdx.x == 0) {
          loss_out[0] = (float)ce / f;
          loss_out[1] = (float)l1 / (f * 6.0f);
          loss_out[2] = f;
        }
      }
    }
    __syncthreads();
    npf = s_np;
  } else {
    npf = loss_out[2];
  }
  if (i >= total) return;
  const float gc = upstream[0] / npf, gl = upstream[1] / (npf * 6.0f);
  const long long tc = true_classes[i];
  const float* x = scores + (size_t)i * ncls;
  float* ds = dscores + (size_t)i * ncls;
  if (tc >= 0) {
    float m = x[0];
    for (int c = 1; c < ncls; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) se += expf(x[c] - m);
    for (int c = 0; c < ncls; ++c) {
      const float sm = expf(x[c] - m) / se;
      ds[c] = gc * (sm - (c == (int)tc ? 1.0f : 0.0f));
    }
  } else {
    for (int c = 0; c < ncls; ++c) ds[c] = 0.f;
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    float g = 0.f;
    if (tc > 0) {
      const float d = locs[(size_t)i * 6 + q] - true_locs[(size_t)i * 6 + q];
      g = d > 0.f ? gl : (d < 0.f ? -gl : 0.f);
    }
    dlocs[(size_t)i * 6 + q] = g;
  }
}

// Loss of the training hot loop in ONE launch (ssd3d.py:890-941 + its autograd + the scatter of the head gradients): with
// the number of positives known from the matching (msl_multibox_match_count) the per-prior gradients do not have to wait
// for the loss sums, so a thread computes the cross entropy / L1 terms of its prior, their gradients
//     dL/dscores = upstream[0] / n_pos * (softmax - onehot)   (ignored priors 0),   dL/dlocs = upstream[1] / (6 n_pos) * sign
// and writes them straight into the zero-haloed (N, 16*MT, D+2, H+2, W+2) gradient images the head convolutions' backward
// reads (row a*6 + q for the box regressions of anchor a, 12 + a*ncls + c for its class scores): no (N,P,.) gradient tensors,
// no pack launch.  The (sum ce, sum l1) partials of the workgroups are folded into loss_out by the step's batched gradient
// reduction (optim.hip kind 4) - nothing on the dependency chain waits for the loss VALUE.
struct LossPackDst {
  float* dO[4];
  int D[4], H[4], W[4], prior_off[4];
};
__global__ __launch_bounds__(256) void multibox_loss_pack_kernel(
    const float* __restrict__ locs, const float* __restrict__ scores, const long long* __restrict__ true_classes,
    const float* __restrict__ true_locs, const int* __restrict__ npos, const float* __restrict__ upstream,
    double* __restrict__ partials, int* __restrict__ nan_flag, LossPackDst dst, int nscales, int N, int P, int ncls, int CO) {
  __shared__ double scratch[8];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float npf = (float)*npos;
  const float gc = upstream[0] / npf, gl = upstream[1] / (npf * 6.0f);
  double ce = 0.0, l1 = 0.0;
  bool bad_loc = false, bad_score = false;
  if (i < N * P) {
    const int n = i / P, p = i - n * P;
    const long long tc = true_classes[i];
    const float* x = scores + (size_t)i * ncls;
    float xs[MAXC], gs[MAXC], lq[6], gq[6];
    for (int c = 0; c < ncls; ++c) {
      xs[c] = x[c];
      bad_score |= isnan(xs[c]);
      gs[c] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      lq[q] = locs[(size_t)i * 6 + q];
      bad_loc |= isnan(lq[q]);
      gq[q] = 0.f;
    }
    if (tc >= 0) {
      float m = xs[0];
      for (int c = 1; c < ncls; ++c) m = fmaxf(m, xs[c]);
      float se = 0.f;
      for (int c = 0; c < ncls; ++c) se += expf(xs[c] - m);
      ce = (double)((m + logf(se)) - xs[tc]);
      for (int c = 0; c < ncls; ++c) gs[c] = gc * (expf(xs[c] - m) / se - (c == (int)tc ? 1.0f : 0.0f));
    }
    if (tc > 0) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float d = lq[q] - true_locs[(size_t)i * 6 + q];
        a += fabsf(d);
        gq[q] = d > 0.f ? gl : (d < 0.f ? -gl : 0.f);
      }
      l1 = (double)a;
    }
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (j < nscales && p >= dst.prior_off[j]) k = j;
    const int local = p - dst.prior_off[k], pos = local >> 1, a = local & 1;  // two anchors per location (ssd3d.py:213)
    const int W = dst.W[k], H = dst.H[k], D = dst.D[k];
    const int w = pos % W, h = (pos / W) % H, d = pos / (W * H);
    const int Hp = H + 2, Wp = W + 2;
    const size_t volp = (size_t)(D + 2) * Hp * Wp;
    float* base = dst.dO[k] + (size_t)n * CO * volp + ((size_t)(d + 1) * Hp + h + 1) * Wp + w + 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) base[(size_t)(a * 6 + q) * volp] = gq[q];
    for (int c = 0; c < ncls; ++c) base[(size_t)(12 + a * ncls + c) * volp] = gs[c];
  }
  if (nan_flag) {  // the NaN guards of ssd3d.py:258-261 for free: this kernel reads every loc and score anyway
    if (__any(bad_loc) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
    if (__any(bad_score) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 2);
  }
  const double t0 = msl::block_sum(ce, scratch);
  __syncthreads();
  const double t1 = msl::block_sum(l1, scratch);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 2 + 0] = t0;
    partials[blockIdx.x * 2 + 1] = t1;
  }
}

constexpr int LOSS_BLOCKS = 64;

// ---- optional loss variants (the reference keeps them as commented code, ssd3d.py:760,926-932) ----
// flags: 1 = hard-negative mining, 2 = smooth-L1 localisation, 4 = focal confidence.  The live path above (flags 0)
// stays on its own kernels.
constexpr int VAR_HNM = 1, VAR_SMOOTH = 2, VAR_FOCAL = 4;

// confidence loss of one prior with target class tc >= 0
__device__ __forceinline__ float conf_value(const float* __restrict__ x, long long tc, int ncls, bool focal) {
  if (focal) {  // sigmoid focal loss on the foreground logit: 0.25 * (1 - p_t)^2 * BCE   (MONAI FocalLoss, gamma 2)
    const float z = tc > 0 ? x[1] : -x[1];                          // p_t = sigmoid(z)
    const float sp = fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));      // BCE = softplus(-z)
    const float q = 1.f / (1.f + expf(z));                          // 1 - p_t
    return 0.25f * q * q * sp;
  }
  float m = x[0];
  for (int c = 1; c < ncls; ++c) m = fmaxf(m, x[c]);
  float se = 0.f;
  for (int c = 0; c < ncls; ++c) se += expf(x[c] - m);
  return (m + logf(se)) - x[tc];
}

__device__ __forceinline__ unsigned sortable_key(float v) {  // larger float <=> larger key (negative zero / tiny negatives too)
  const unsigned b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// forward partial sums.  With VAR_HNM the negatives' losses go to conf_all (positives / ignored priors = 0, exactly the
// reference's conf_loss_neg) and only the positives' confidence loss is summed here.
__global__ __launch_bounds__(256) void multibox_loss_partial_var_kernel(
    const float* __restrict__ locs, const float* __restrict__ scores, const long long* __restrict__ true_classes,
    const float* __restrict__ true_locs, double* __restrict__ partials, float* __restrict__ conf_all, int total,
    int ncls, int flags, int* __restrict__ nan_flag) {
  __shared__ double scratch[8];
  const bool hnm = flags & VAR_HNM, smooth = flags & VAR_SMOOTH, focal = flags & VAR_FOCAL;
  double ce = 0.0, l1 = 0.0, np = 0.0;
  bool bad_loc = false, bad_score = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (nan_flag) {
      for (int c = 0; c < ncls; ++c) bad_score |= isnan(scores[(size_t)i * ncls + c]);
#pragma unroll
      for (int q = 0; q < 6; ++q) bad_loc |= isnan(locs[(size_t)i * 6 + q]);
    }
    const long long tc = true_classes[i];
    float cv = 0.f;
    if (tc >= 0) cv = conf_value(scores + (size_t)i * ncls, tc, ncls, focal);
    if (hnm) {
      conf_all[i] = tc == 0 ? cv : 0.f;
      if (tc > 0) ce += (double)cv;
    } else if (tc >= 0) {
      ce += (double)cv;
    }
    if (tc > 0) {
      np += 1.0;
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float d = locs[(size_t)i * 6 + q] - true_locs[(size_t)i * 6 + q];
        const float ad = fabsf(d);
        a += smooth ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : ad;
      }
      l1 += (double)a;
    }
  }
  if (nan_flag) {
    if (__any(bad_loc) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
    if (__any(bad_score) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 2);
  }
  const double t0 = msl::block_sum(ce, scratch);
  __syncthreads();
  const double t1 = msl::block_sum(l1, scratch);
  __syncthreads();
  const double t2 = msl::block_sum(np, scratch);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = t0;
    partials[blockIdx.x * 3 + 1] = t1;
    partials[blockIdx.x * 3 + 2] = t2;
  }
}

// Hard-negative mining of one image per workgroup (ssd3d.py:907-908,926-929): among the P entries of conf_loss_neg keep
// the k = neg_pos_ratio * n_positives largest.  An 8-bit radix select over the sortable keys finds the k-th largest
// value T; everything above T is kept, and of the entries equal to T the first (k - #above) in index order (what a
// stable descending sort keeps).  Their sum goes to hnm_sum[n] (fp64, fixed order) and conf_all becomes the 0/1 mask
// the backward kernel reads.
__global__ __launch_bounds__(1024) void multibox_hnm_select_kernel(const long long* __restrict__ true_classes,
                                                                   float* __restrict__ conf_all,
                                                                   double* __restrict__ hnm_sum, int P, int ratio) {
  __shared__ int hist[256];
  __shared__ int s_i[4];       // 0: n_pos, 1: chosen bin, 2: remaining, 3: running tie rank
  __shared__ int s_wave[16];
  __shared__ double scratch[16];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long* tc = true_classes + (size_t)n * P;
  float* v = conf_all + (size_t)n * P;
  if (tid < 4) s_i[tid] = 0;
  __syncthreads();
  int np = 0;
  for (int j = tid; j < P; j += 1024) np += tc[j] > 0;
  for (int o = 32; o > 0; o >>= 1) np += __shfl_down(np, o);
  if (lane == 0 && np) atomicAdd(&s_i[0], np);
  __syncthreads();
  const long long kk = (long long)ratio * s_i[0];
  const int k = kk > P ? P : (int)kk;
  if (k <= 0 || k >= P) {  // nothing / everything is kept
    double acc = 0.0;
    for (int j = tid; j < P; j += 1024) {
      if (k > 0) acc += (double)v[j];
      v[j] = k > 0 ? 1.f : 0.f;
    }
    const double t = msl::block_sum(acc, scratch);
    if (tid == 0) hnm_sum[n] = t;
    return;
  }
  unsigned prefix = 0;
  if (tid == 0) s_i[2] = k;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    for (int b = tid; b < 256; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int j = tid; j < P; j += 1024) {
      const unsigned key = sortable_key(v[j]);
      if (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int remaining = s_i[2], cum = 0, b = 255;
      for (; b > 0; --b) {
        if (cum + hist[b] >= remaining) break;
        cum += hist[b];
      }
      s_i[1] = b;
      s_i[2] = remaining - cum;  // still needed from bin b
    }
    __syncthreads();
    prefix |= (unsigned)s_i[1] << shift;
  }
  const unsigned tkey = prefix;
  const int need = s_i[2];  // entries equal to T that are kept (>= 1)
  double acc = 0.0;
  for (int j0 = 0; j0 < P; j0 += 1024) {  // index order: the tie rank must count every earlier tie
    const int j = j0 + tid;
    const float val = j < P ? v[j] : 0.f;
    const unsigned key = j < P ? sortable_key(val) : 0u;
    const bool tie = j < P && key == tkey;
    const unsigned long long bal = __ballot(tie);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int off = s_i[3];
    for (int w = 0; w < wv; ++w) off += s_wave[w];
    const bool keep = j < P && (key > tkey || (tie && off + before < need));
    if (keep) acc += (double)val;
    if (j < P) v[j] = keep ? 1.f : 0.f;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_i[3] += tot;
    }
    __syncthreads();
  }
  const double t = msl::block_sum(acc, scratch);
  if (tid == 0) hnm_sum[n] = t;
}

// backward of the variants; every workgroup folds the forward partials (+ the per-image mined sums) itself and workgroup
// 0 publishes loss_out = [conf, loc, n_positives], as multibox_loss_bwd_kernel<true> does.  write_grads = 0: losses only.
__global__ __launch_bounds__(256) void multibox_loss_bwd_var_kernel(
    const float* __restrict__ locs, const float* __restrict__ scores, const long long* __restrict__ true_classes,
    const float* __restrict__ true_locs, float* __restrict__ loss_out, const double* __restrict__ partials, int nparts,
    const double* __restrict__ hnm_sum, int nimages, const float* __restrict__ sel, const float* __restrict__ upstream,
    float* __restrict__ dlocs, float* __restrict__ dscores, int total, int ncls, int flags, int write_grads) {
  __shared__ float s_np;
  const bool hnm = flags & VAR_HNM, smooth = flags & VAR_SMOOTH, focal = flags & VAR_FOCAL;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 64) {
    double ce = 0.0, l1 = 0.0, np = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) {
      ce += partials[k * 3 + 0];
      l1 += partials[k * 3 + 1];
      np += partials[k * 3 + 2];
    }
    if (hnm)
      for (int k = threadIdx.x; k < nimages; k += 64) ce += hnm_sum[k];
    ce = msl::wave_sum(ce);
    l1 = msl::wave_sum(l1);
    np = msl::wave_sum(np);
    if (threadIdx.x == 0) {
      const float f = (float)np;
      s_np = f;
      if (blockIdx.x == 0) {
        loss_out[0] = (float)ce / f;
        loss_out[1] = (float)l1 / (f * 6.0f);
        loss_out[2] = f;
      }
    }
  }
  __syncthreads();
  const float npf = s_np;
  if (i >= total || !write_grads) return;
  const float gc = upstream[0] / npf, gl = upstream[1] / (npf * 6.0f);
  const long long tc = true_classes[i];
  const float* x = scores + (size_t)i * ncls;
  float* ds = dscores + (size_t)i * ncls;
  const bool active = tc > 0 || (tc == 0 && (!hnm || sel[i] != 0.f));
  if (!active) {
    for (int c = 0; c < ncls; ++c) ds[c] = 0.f;
  } else if (focal) {
    const float sgn = tc > 0 ? 1.f : -1.f;
    const float z = sgn * x[1];
    const float sp = fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));
    const float q = 1.f / (1.f + expf(z));
    const float dz = -0.25f * q * q * (2.f * (1.f - q) * sp + q);
    for (int c = 0; c < ncls; ++c) ds[c] = c == 1 ? gc * sgn * dz : 0.f;
  } else {
    float m = x[0];
    for (int c = 1; c < ncls; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) se += expf(x[c] - m);
    for (int c = 0; c < ncls; ++c) {
      const float sm = expf(x[c] - m) / se;
      ds[c] = gc * (sm - (c == (int)tc ? 1.0f : 0.0f));
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    float g = 0.f;
    if (tc > 0) {
      const float d = locs[(size_t)i * 6 + q] - true_locs[(size_t)i * 6 + q];
      if (smooth && fabsf(d) < 1.f) g = gl * d;
      else g = d > 0.f ? gl : (d < 0.f ? -gl : 0.f);
    }
    dlocs[(size_t)i * 6 + q] = g;
  }
}

}  // namespace

extern "C" {

int msl_make_priors(float* out, int row_off, int D0, int D1, int D2, double scale, int boxes_per_location,
                    void* stream) {
  if (D0 <= 0 || D1 <= 0 || D2 <= 0 || boxes_per_location < 1) return MSL_ERR_ARG;
  const int total = D0 * D1 * D2 * boxes_per_location;
  MSL_LAUNCH(make_priors_kernel, dim3(msl::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out,
                     row_off, D0, D1, D2, scale, boxes_per_location);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// op: 0 cxcycz_to_xyz, 1 xyz_to_cxcycz, 2 cxcycz_to_gcxgcygcz (encode), 3 gcxgcygcz_to_cxcycz (decode)
int msl_box_transform(const float* boxes, const float* priors, float* out, int n, int op, void* stream) {
  if (n < 0 || op < 0 || op > 3) return MSL_ERR_ARG;
  if (n == 0) return MSL_OK;
  MSL_LAUNCH(box_transform_kernel, dim3(msl::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, boxes, priors,
                     out, n, op);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_iou_matrix(const float* set1, const float* set2, float* out, int n1, int n2, int intersection_only,
                   void* stream) {
  if (n1 < 0 || n2 < 0) return MSL_ERR_ARG;
  if (n1 == 0 || n2 == 0) return MSL_OK;
  MSL_LAUNCH(iou_matrix_kernel, dim3(msl::cdiv(n2, 256), n1), dim3(256), 0, (hipStream_t)stream, set1, set2,
                     out, n1, n2, intersection_only);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// gt_boxes (T,6) corner form, gt_labels (T,), obj_off (N+1,) prefix offsets into them, all on device.
// scratch: overlap (N,P) f32, obj (N,P) i32, prior_for_obj (T,) i32.  soft != 0 -> two-threshold band.
static int match_impl(const float* gt_boxes, const long long* gt_labels, const int* obj_off, int total_objects,
                      const float* priors_c, int N, int P, float thr_lo, float thr_hi, int soft, float* overlap,
                      int* obj, int* prior_for_obj, long long* true_classes, float* true_locs,
                      long long* matched, int* npos, void* stream) {
  if (N <= 0 || P <= 0 || total_objects < 0) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 gp(msl::cdiv(P, 256), N);
  if (total_objects > 0) {
    MSL_LAUNCH(match_prior_best_kernel, gp, dim3(256), 0, st, gt_boxes, obj_off, priors_c, overlap, obj, P, npos);
    MSL_LAUNCH_CHECK();
    MSL_LAUNCH(match_object_best_kernel, dim3(total_objects), dim3(MOB_T), 0, st, gt_boxes, priors_c,
                       prior_for_obj, P);
    MSL_LAUNCH_CHECK();
    MSL_LAUNCH(match_force_kernel, dim3(N), dim3(256), 0, st, obj_off, prior_for_obj, overlap, obj, P);
    MSL_LAUNCH_CHECK();
  }
  MSL_LAUNCH(match_encode_kernel, gp, dim3(256), 0, st, gt_boxes, gt_labels, obj_off, priors_c, overlap, obj,
                     thr_lo, thr_hi, soft, true_classes, true_locs, matched, P, npos, total_objects > 0 ? 0 : 1);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_multibox_match(const float* gt_boxes, const long long* gt_labels, const int* obj_off, int total_objects,
                       const float* priors_c, int N, int P, float thr_lo, float thr_hi, int soft, float* overlap,
                       int* obj, int* prior_for_obj, long long* true_classes, float* true_locs,
                       long long* matched, void* stream) {
  return match_impl(gt_boxes, gt_labels, obj_off, total_objects, priors_c, N, P, thr_lo, thr_hi, soft, overlap, obj,
                    prior_for_obj, true_classes, true_locs, matched, nullptr, stream);
}

// the same, and *npos = number of positive priors of the batch (for msl_multibox_loss_pack)
int msl_multibox_match_count(const float* gt_boxes, const long long* gt_labels, const int* obj_off, int total_objects,
                             const float* priors_c, int N, int P, float thr_lo, float thr_hi, int soft, float* overlap,
                             int* obj, int* prior_for_obj, long long* true_classes, float* true_locs,
                             long long* matched, int* npos, void* stream) {
  if (!npos) return MSL_ERR_ARG;
  return match_impl(gt_boxes, gt_labels, obj_off, total_objects, priors_c, N, P, thr_lo, thr_hi, soft, overlap, obj,
                    prior_for_obj, true_classes, true_locs, matched, npos, stream);
}

size_t msl_multibox_loss_workspace_bytes(void) { return (size_t)LOSS_BLOCKS * 3 * sizeof(double); }

// loss_out[0] = conf, [1] = loc, [2] = number of positive priors
int msl_multibox_loss_fwd(const float* locs, const float* scores, const long long* true_classes,
                          const float* true_locs, double* workspace, float* loss_out, int N, int P, int ncls,
                          void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || ncls > MAXC) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  MSL_LAUNCH(multibox_loss_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, locs, scores, true_classes,
                     true_locs, workspace, N * P, ncls, (int*)nullptr);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(multibox_loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, LOSS_BLOCKS, loss_out);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// upstream[0] = dL/dconf, upstream[1] = dL/dloc (device)
int msl_multibox_loss_bwd(const float* locs, const float* scores, const long long* true_classes,
                          const float* true_locs, const float* loss_out, const float* upstream, float* dlocs,
                          float* dscores, int N, int P, int ncls, void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || ncls > MAXC) return MSL_ERR_ARG;
  MSL_LAUNCH(multibox_loss_bwd_kernel<false>, dim3(msl::cdiv(N * P, 256)), dim3(256), 0, (hipStream_t)stream,
                     locs, scores, true_classes, true_locs, const_cast<float*>(loss_out), nullptr, 0, upstream, dlocs,
                     dscores, N * P, ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// forward + backward of the loss in two launches (training hot loop): the backward kernel folds the forward's
// partials itself and publishes loss_out = [conf, loc, n_positives]
int msl_multibox_loss_fwd_bwd(const float* locs, const float* scores, const long long* true_classes,
                              const float* true_locs, double* workspace, float* loss_out, const float* upstream,
                              float* dlocs, float* dscores, int* nan_flag, int N, int P, int ncls, void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || ncls > MAXC) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  MSL_LAUNCH(multibox_loss_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, locs, scores, true_classes,
                     true_locs, workspace, N * P, ncls, nan_flag);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(multibox_loss_bwd_kernel<true>, dim3(msl::cdiv(N * P, 256)), dim3(256), 0, st, locs, scores,
                     true_classes, true_locs, loss_out, workspace, LOSS_BLOCKS, upstream, dlocs, dscores, N * P, ncls);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// Loss + loss gradient + head-gradient images in one launch (training hot loop; see multibox_loss_pack_kernel).  npos: the
// counter msl_multibox_match_count filled; partials: 2 * msl_multibox_loss_pack_num_partials(N, P) doubles, folded into
// loss_out = [conf, loc, n_positives] by a kind-4 entry of msl_grad_reduce_batch; dO_pad / D / H / W / prior_off: host arrays of
// n <= 4 scales (the zero-haloed gradient images of msl_head_grad_pack, whose padding rows must be zero and stay untouched).
int msl_multibox_loss_pack_num_partials(int N, int P) { return msl::cdiv(N * P, 256); }

int msl_multibox_loss_pack(const float* locs, const float* scores, const long long* true_classes, const float* true_locs,
                           const int* npos, const float* upstream, double* partials, int* nan_flag, float* const* dO_pad,
                           const int* D, const int* H, const int* W, const int* prior_off, int n, int N, int P, int ncls,
                           void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || ncls > MAXC || n <= 0 || n > 4 || !npos || !upstream || !partials) return MSL_ERR_ARG;
  LossPackDst dst;
  for (int k = 0; k < n; ++k) {
    dst.dO[k] = dO_pad[k];
    dst.D[k] = D[k];
    dst.H[k] = H[k];
    dst.W[k] = W[k];
    dst.prior_off[k] = prior_off[k];
  }
  const int CO = 16 * ((12 + 2 * ncls + 15) / 16);
  MSL_LAUNCH(multibox_loss_pack_kernel, dim3(msl::cdiv(N * P, 256)), dim3(256), 0, (hipStream_t)stream, locs, scores,
                     true_classes, true_locs, npos, upstream, partials, nan_flag, dst, n, N, P, ncls, CO);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// Optional loss variants (flags: 1 hard-negative mining with neg_pos_ratio, 2 smooth-L1, 4 focal; 0 = the live path).
// var_ws: msl_multibox_loss_var_workspace_bytes(N, P) bytes = [N*P floats: mined mask][N doubles: mined sums].
// dlocs / dscores may be null (forward only).  loss_out = [conf, loc, n_positives].
size_t msl_multibox_loss_var_workspace_bytes(int N, int P) {
  return (((size_t)N * P * sizeof(float) + 15) & ~(size_t)15) + (size_t)N * sizeof(double);
}

int msl_multibox_loss_var(const float* locs, const float* scores, const long long* true_classes, const float* true_locs,
                          double* workspace, void* var_ws, float* loss_out, const float* upstream, float* dlocs,
                          float* dscores, int* nan_flag, int N, int P, int ncls, int flags, int neg_pos_ratio,
                          void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || ncls > MAXC || flags < 0 || flags > 7 || neg_pos_ratio < 0) return MSL_ERR_ARG;
  if ((flags & VAR_FOCAL) && ncls != 2) return MSL_ERR_UNSUPPORTED;  // defined for background + one class
  if ((dlocs == nullptr) != (dscores == nullptr) || (dlocs && !upstream) || !var_ws) return MSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* sel = (float*)var_ws;
  double* sums = (double*)((char*)var_ws + (((size_t)N * P * sizeof(float) + 15) & ~(size_t)15));
  MSL_LAUNCH(multibox_loss_partial_var_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, locs, scores, true_classes,
                     true_locs, workspace, sel, N * P, ncls, flags, nan_flag);
  MSL_LAUNCH_CHECK();
  if (flags & VAR_HNM) {
    MSL_LAUNCH(multibox_hnm_select_kernel, dim3(N), dim3(1024), 0, st, true_classes, sel, sums, P, neg_pos_ratio);
    MSL_LAUNCH_CHECK();
  }
  const int write = dlocs != nullptr;
  MSL_LAUNCH(multibox_loss_bwd_var_kernel, dim3(write ? msl::cdiv(N * P, 256) : 1), dim3(256), 0, st, locs,
                     scores, true_classes, true_locs, loss_out, workspace, LOSS_BLOCKS, sums, N, sel, upstream, dlocs,
                     dscores, N * P, ncls, flags, write);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
