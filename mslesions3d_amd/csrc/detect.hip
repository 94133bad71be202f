// Detection decoding: softmax -> box decode -> per-class score filter -> sort -> cap -> 3D greedy NMS ->
// placeholder / global top-k.   Reference: LSSD3D.detect_objects (lesions3d/ssd3d.py:344-460).
//
// The reference runs a Python loop with one host sync per candidate box.  Here the whole batch is 5
// launches and no host sync until the final counts are read:
//   prepare  : softmax + decode for every prior (element-wise)
//   select   : only the best cap = 10*top_k candidates of a class matter, so a 2048-bin histogram of the scores picks
//              the threshold bin and the candidates at or above it are compacted into a shortlist (O(P))
//   rank     : stable descending order of the shortlist by counting
//              (rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)}) from LDS tiles - exact, deterministic, no sort
//              network; every candidate that beats a shortlisted one is shortlisted too (equal scores share a bin), so
//              the rank inside the shortlist IS the global rank.  O(K^2) with K ~ cap instead of O(P^2): 1.04 ms ->
//              a few us at 192^3 (P = 31 536)
//   mask     : 64-bit suppression masks  IoU(i, j) > max_overlap  for the (<= 10*top_k) sorted candidates
//   scan     : one wavefront walks the candidates in score order with the suppression set held as one
//              64-bit word per lane (ssd3d.py:414-426 semantics, including "suppress all, then clear self")
//   finalize : per image concat of classes, placeholder if empty (ssd3d.py:437-440), top-k (ssd3d.py:449-453)
// Ties between equal scores are resolved by ascending prior index (stable order).  IoU arithmetic keeps the
// reference's operation order with FMA contraction off, so keep-lists are bit-exact for equal inputs.
#include "common.hpp"
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float iou6(const float* a, const float* b) {
  float e[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float lo = fmaxf(a[i], b[i]);
    const float hi = fminf(a[3 + i], b[3 + i]);
    e[i] = fmaxf(hi - lo, 0.0f);
  }
  const float inter = e[0] * e[1] * e[2];
  const float va = (a[3] - a[0]) * (a[4] - a[1]) * (a[5] - a[2]);
  const float vb = (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]);
  const float uni = va + vb - inter;
  return inter / uni;
}

#define MSL_FN __device__ __forceinline__
#include "softmax_exp.h"

// probs (N, ncls-1, P): foreground class probabilities; boxes (N, P, 6) decoded corner boxes
__global__ __launch_bounds__(256) void detect_prepare_kernel(const float* __restrict__ locs,
                                                             const float* __restrict__ scores,
                                                             const float* __restrict__ priors_c,
                                                             float* __restrict__ probs, float* __restrict__ boxes,
                                                             int N, int P, int ncls, int* __restrict__ sel, int sel_count,
                                                             int* __restrict__ ncand, int ncand_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // the counters of the selection passes start at zero (was two memset launches in front of this kernel)
  for (int j = i; j < sel_count; j += gridDim.x * 256) sel[j] = 0;
  if (i < ncand_count) ncand[i] = 0;
  if (i >= N * P) return;
  const int n = i / P, p = i % P;
  const float* x = scores + (size_t)i * ncls;
  // bit for bit what torch's CPU softmax yields (softmax_exp.h)
  msl_softmax_foreground(x, ncls, probs + (size_t)n * (ncls - 1) * P + p, P);
  const float* g = locs + (size_t)i * 6;
  const float* pr = priors_c + (size_t)p * 6;
  float cxyz[3], sz[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    cxyz[k] = g[k] * pr[3 + k] / 10.0f + pr[k];     // utils.py:67
    sz[k] = expf(g[3 + k] / 5.0f) * pr[3 + k];      // utils.py:68
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float h = sz[k] / 2.0f;                    // utils.py:50-51
    boxes[(size_t)i * 6 + k] = cxyz[k] - h;
    boxes[(size_t)i * 6 + 3 + k] = cxyz[k] + h;
  }
}

constexpr int HB = 2048;  // histogram bins over the probability range [0, 1]
__device__ __forceinline__ int score_bin(float sc) { return min(HB - 1, max(0, (int)(sc * (float)HB))); }

// select workspace per (n, class): [hist HB | threshold bin | shortlist count | pad 2 | shortlist P]
__host__ __device__ __forceinline__ size_t sel_stride(int P) { return (size_t)HB + 4 + P; }

// grid (ceil(P/256), N*(ncls-1)): histogram of the candidates' scores (+ their count)
__global__ __launch_bounds__(256) void detect_hist_kernel(const float* __restrict__ probs, float min_score, int P,
                                                          int* __restrict__ sel, int* __restrict__ ncand) {
  __shared__ int h[HB];
  const int nc = blockIdx.y;
  for (int b = threadIdx.x; b < HB; b += 256) h[b] = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float si = i < P ? probs[(size_t)nc * P + i] : 0.f;
  const bool cand = i < P && si > min_score;  // strict (ssd3d.py:388); NaN is never a candidate
  if (cand) atomicAdd(&h[score_bin(si)], 1);
  const unsigned long long any = __ballot(cand);
  if ((threadIdx.x & 63) == 0 && any) atomicAdd(&ncand[nc], __popcll(any));
  __syncthreads();
  int* hist = sel + (size_t)nc * sel_stride(P);
  for (int b = threadIdx.x; b < HB; b += 256)
    if (h[b]) atomicAdd(&hist[b], h[b]);  // integer counts: the result does not depend on the order
}

// grid (N*(ncls-1)), one wave: the highest bin b* with  #(candidates in bins >= b*) >= cap  (bin 0 if fewer candidates)
__global__ __launch_bounds__(64) void detect_threshold_kernel(int* __restrict__ sel, int P, int cap) {
  int* hist = sel + (size_t)blockIdx.x * sel_stride(P);
  const int lane = threadIdx.x;
  constexpr int PER = HB / 64;  // lane l owns bins [l*PER, (l+1)*PER)
  int own = 0;
  for (int k = 0; k < PER; ++k) own += hist[lane * PER + k];
  // suffix sum over lanes: above = candidates in higher lanes' bins
  int above = 0;
  for (int l = 63; l >= 0; --l) {
    const int v = __shfl(own, l, 64);
    if (l > lane) above += v;
  }
  int thr = -1;
  if (above < cap && above + own >= cap) {  // the threshold bin is in this lane's range (exactly one lane)
    int acc = above;
    for (int k = PER - 1; k >= 0; --k) {
      acc += hist[lane * PER + k];
      if (acc >= cap) {
        thr = lane * PER + k;
        break;
      }
    }
  }
  const unsigned long long found = __ballot(thr >= 0);
  if (found == 0ull) {
    if (lane == 0) hist[HB] = 0;  // fewer than cap candidates: everybody is shortlisted
  } else if (thr >= 0) {
    hist[HB] = thr;
  }
  if (lane == 0) hist[HB + 1] = 0;  // shortlist counter
}

// grid (ceil(P/256), N*(ncls-1)): candidates in bins >= b* -> shortlist (any order: the ranking below is by value)
__global__ __launch_bounds__(256) void detect_compact_kernel(const float* __restrict__ probs, float min_score, int P,
                                                             int* __restrict__ sel) {
  const int nc = blockIdx.y;
  int* ws = sel + (size_t)nc * sel_stride(P);
  const int thr = ws[HB];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float si = i < P ? probs[(size_t)nc * P + i] : 0.f;
  const bool take = i < P && si > min_score && score_bin(si) >= thr;
  const unsigned long long m = __ballot(take);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(&ws[HB + 1], __popcll(m));
  base = __shfl(base, 0, 64);
  if (take) ws[HB + 4 + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

// grid (ceil(P/256), N*(ncls-1)); blocks beyond the shortlist exit at once.  rank among the shortlist == global rank.
__global__ __launch_bounds__(256) void detect_rank_kernel(const float* __restrict__ probs, int P, int cap,
                                                          const int* __restrict__ sel, int* __restrict__ sorted_idx) {
  __shared__ float ts[256];
  __shared__ int ti[256];
  const int nc = blockIdx.y;
  const int* ws = sel + (size_t)nc * sel_stride(P);
  const int K = ws[HB + 1];
  if (blockIdx.x * 256 >= K) return;
  const int* list = ws + HB + 4;
  const float* s = probs + (size_t)nc * P;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool live = a < K;
  const int i = live ? list[a] : 0;
  const float si = live ? s[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < K; j0 += 256) {
    const int b = j0 + threadIdx.x;
    const int jb = b < K ? list[b] : 0;
    ti[threadIdx.x] = jb;
    ts[threadIdx.x] = b < K ? s[jb] : 0.f;
    __syncthreads();
    const int lim = min(256, K - j0);
    for (int t = 0; t < lim; ++t) {
      const float sj = ts[t];
      rank += (sj > si) || (sj == si && ti[t] < i);
    }
    __syncthreads();
  }
  if (live && rank < cap) sorted_idx[(size_t)nc * cap + rank] = i;
}

// mask[nc][i][w] bit b = IoU(sorted i, sorted 64w+b) > max_overlap.   grid (ceil(cap/4), N*(ncls-1)), 4 rows/block
__global__ __launch_bounds__(256) void detect_mask_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ sorted_idx,
                                                          const int* __restrict__ ncand, float max_overlap, int P,
                                                          int cap, int Wn, int ncls1,
                                                          unsigned long long* __restrict__ mask) {
  const int nc = blockIdx.y, n = nc / ncls1;
  const int M = min(ncand[nc], cap);
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= M) return;
  const int* sidx = sorted_idx + (size_t)nc * cap;
  const float* bn = boxes + (size_t)n * P * 6;
  float bi[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) bi[k] = bn[(size_t)sidx[i] * 6 + k];
  for (int w = 0; w < Wn; ++w) {
    const int j = w * 64 + lane;
    bool over = false;
    if (j < M) {
      float bj[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) bj[k] = bn[(size_t)sidx[j] * 6 + k];
      over = iou6(bi, bj) > max_overlap;  // strict; NaN -> false (ssd3d.py:422)
    }
    const unsigned long long bits = __ballot(over);
    if (lane == 0) mask[((size_t)nc * cap + i) * Wn + w] = bits;
  }
}

// One workgroup per (n, class).  The greedy rule (ssd3d.py:417-426: candidates in score order; one that no KEPT earlier
// candidate overlaps is kept) has a unique solution: keep_i = !any_{j < i}(keep_j && M[j][i]).  The overlap mask is bitwise
// symmetric (iou6 commutes in every operation), so M[j][i] is bit j of candidate i's OWN row and each candidate can test
// itself.  Wave 0 walks blocks of 64 candidates, one per lane:
//   * against the earlier blocks (final): any(row_i[w] & keep[w]), w < block - independent LDS reads;
//   * inside the block: start from "everyone not yet suppressed is kept" and re-evaluate
//       keep_i = !(row_i[block] & keep_block & bits below i)
//     with one ballot per round until the word stops changing.  After round t the first t lanes are final, so at most 64
//     rounds; in practice the length of the longest suppression chain inside the block (a handful).
// The one-candidate-at-a-time walk this replaces took 76-111 us at 500 candidates (about 200 cycles of scalar / lane-read
// latency per candidate); a version that only serialised over the KEPT candidates was no faster.
constexpr int SCAN_LDS = 6144;  // 64-bit words (48 KB): mask rows staged by all 256 threads when they fit
__global__ __launch_bounds__(256) void detect_scan_kernel(const unsigned long long* __restrict__ mask,
                                                          const int* __restrict__ ncand, int cap, int Wn,
                                                          unsigned long long* __restrict__ keep_bits,
                                                          int* __restrict__ nkept) {
  __shared__ unsigned long long lm[SCAN_LDS];
  __shared__ unsigned long long kf[64];  // final keep words of the blocks done so far
  const int nc = blockIdx.x, lane = threadIdx.x & 63;
  const int M = min(ncand[nc], cap);
  const unsigned long long* mk = mask + (size_t)nc * cap * Wn;
  const bool staged = (size_t)M * Wn <= SCAN_LDS;
  if (staged) {
    for (int e = threadIdx.x; e < M * Wn; e += 256) lm[e] = mk[e];
    __syncthreads();
  }
  if (threadIdx.x >= 64) return;
  const int nblk = (M + 63) >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;  // the candidates of this block in front of this lane's
  int cnt = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int i = (blk << 6) + lane;
    const bool valid = i < M;
    const int ir = valid ? i : 0;
    const unsigned long long* row = staged ? &lm[ir * Wn] : nullptr;
    bool free_ = valid;  // not suppressed by a kept candidate of an earlier block
    for (int w = 0; w < blk; ++w) {
      const unsigned long long rw = staged ? row[w] : mk[(size_t)ir * Wn + w];
      free_ = free_ && (rw & kf[w]) == 0ull;
    }
    const unsigned long long rb = staged ? row[blk] : mk[(size_t)ir * Wn + blk];
    unsigned long long cur = __ballot(free_);
    for (int round = 0; round < 64; ++round) {  // wave-uniform exit
      const unsigned long long nw = __ballot(free_ && (rb & cur & lower) == 0ull);
      if (nw == cur) break;
      cur = nw;
    }
    if (lane == 0) kf[blk] = cur;  // (a wave's LDS operations execute in order: the next block's reads see it)
    if (lane == 0) keep_bits[(size_t)nc * Wn + blk] = cur;
    cnt += __popcll(cur);
  }
  if (lane >= nblk && lane < Wn) keep_bits[(size_t)nc * Wn + lane] = 0ull;
  if (lane == 0) nkept[nc] = cnt;
}

constexpr int FIN_LDS = 8192;

// one workgroup per image
__global__ __launch_bounds__(256) void detect_finalize_kernel(const float* __restrict__ probs,
                                                              const float* __restrict__ boxes,
                                                              const int* __restrict__ sorted_idx,
                                                              const unsigned long long* __restrict__ keep_bits,
                                                              const int* __restrict__ nkept, int P, int cap, int Wn,
                                                              int ncls1, int top_k, float* __restrict__ out_boxes,
                                                              float* __restrict__ out_scores,
                                                              long long* __restrict__ out_labels,
                                                              long long* __restrict__ out_prior,
                                                              int* __restrict__ out_count,
                                                              float* __restrict__ tmp_scores, int* __restrict__ tmp_ref) {
  const int n = blockIdx.x;
  __shared__ int wpre[64];         // kept candidates before word w of the current class (Wn <= 64)
  __shared__ float lts[FIN_LDS];   // the concatenated scores, when they fit
  // 1. concat kept candidates of all classes, class-major, each class in score order
  int offs = 0;
  const int cat_cap = ncls1 * cap;
  float* ts = tmp_scores + (size_t)n * cat_cap;
  int* tr = tmp_ref + (size_t)n * cat_cap;  // class * P + prior
  for (int c = 0; c < ncls1; ++c) {
    const int nc = n * ncls1 + c;
    const unsigned long long* kb = keep_bits + (size_t)nc * Wn;
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix of the words' popcounts: one load per lane, shuffles (Wn <= 64)
      const int lane = threadIdx.x;
      const int pc = lane < Wn ? __popcll(kb[lane]) : 0;
      int incl = pc;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      wpre[lane] = incl - pc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Wn * 64; i += 256) {
      const unsigned long long word = kb[i >> 6];
      if ((word >> (i & 63)) & 1ull) {
        const int pos = wpre[i >> 6] + __popcll(word & ((1ull << (i & 63)) - 1ull));
        const int prior = sorted_idx[(size_t)nc * cap + i];
        ts[offs + pos] = probs[(size_t)nc * P + prior];
        tr[offs + pos] = c * P + prior;
      }
    }
    offs += nkept[nc];
  }
  __syncthreads();
  const int total = offs;
  float* ob = out_boxes + (size_t)n * top_k * 6;
  float* os = out_scores + (size_t)n * top_k;
  long long* ol = out_labels + (size_t)n * top_k;
  long long* op = out_prior + (size_t)n * top_k;
  if (total == 0) {  // ssd3d.py:437-440
    if (threadIdx.x == 0) {
      ob[0] = 0.f; ob[1] = 0.f; ob[2] = 0.f; ob[3] = 1.f; ob[4] = 1.f; ob[5] = 1.f;
      os[0] = 0.f;
      ol[0] = 0;
      op[0] = -1;
      out_count[n] = 1;
    }
    return;
  }
  // ssd3d.py:449-453: stable descending sort of the concatenation, keep top_k.  With ONE foreground class (the reference's
  // two-class task) the concatenation is that class's keep-list, already in descending score order (ties by ascending prior
  // index): the stable sort is the identity and the best top_k are the first top_k - no ranking pass.
  const bool resort = total > top_k && ncls1 > 1;
  const bool in_lds = resort && total <= FIN_LDS;
  if (in_lds) {
    for (int i = threadIdx.x; i < total; i += 256) lts[i] = ts[i];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < total; i += 256) {
    int pos = i;
    if (resort) {
      // rank = number of candidates that sort in front (stable, descending).  Two explicit loops: a pointer chosen at
      // run time between LDS and global memory is a generic pointer, and 2 x 500 flat loads per thread made this kernel
      // 150 us; the LDS form reads a broadcast word per step.
      pos = 0;
      if (in_lds) {
        const float si = lts[i];
        int j = 0;
        for (; j + 4 <= total; j += 4) {
          const float s0 = lts[j], s1 = lts[j + 1], s2 = lts[j + 2], s3 = lts[j + 3];
          pos += ((s0 > si) || (s0 == si && j < i)) + ((s1 > si) || (s1 == si && j + 1 < i)) +
                 ((s2 > si) || (s2 == si && j + 2 < i)) + ((s3 > si) || (s3 == si && j + 3 < i));
        }
        for (; j < total; ++j) {
          const float sj = lts[j];
          pos += (sj > si) || (sj == si && j < i);
        }
      } else {
        const float si = ts[i];
        for (int j = 0; j < total; ++j) {
          const float sj = ts[j];
          pos += (sj > si) || (sj == si && j < i);
        }
      }
    }
    if (pos < top_k) {
      const int ref = tr[i], c = ref / P, prior = ref % P;
      os[pos] = ts[i];
      ol[pos] = c + 1;
      op[pos] = prior;
#pragma unroll
      for (int k = 0; k < 6; ++k) ob[pos * 6 + k] = boxes[((size_t)n * P + prior) * 6 + k];
    }
  }
  if (threadIdx.x == 0) out_count[n] = min(total, top_k);
}

}  // namespace

extern "C" {

// ints of `select_ws`: per (image, foreground class) a 2048-bin score histogram, the threshold bin, the shortlist counter and
// a shortlist of up to P prior indices
size_t msl_detect_select_ws_ints(int N, int P, int ncls) { return (size_t)N * (ncls - 1) * sel_stride(P); }

// Workspace tensors are caller-allocated (shapes in include/mslesions3d_hip.h).  cap = 10 * top_k <= 4096.
int msl_detect_objects(const float* locs, const float* scores, const float* priors_c, int N, int P, int ncls,
                       float min_score, float max_overlap, int top_k, float* probs, float* boxes,
                       int* sorted_idx, int* ncand, unsigned long long* mask, unsigned long long* keep_bits,
                       int* nkept, float* tmp_scores, int* tmp_ref, int* select_ws, float* out_boxes, float* out_scores,
                       long long* out_labels, long long* out_prior, int* out_count, void* stream) {
  if (N <= 0 || P <= 0 || ncls < 2 || top_k <= 0 || !select_ws) return MSL_ERR_ARG;
  const int cap = 10 * top_k;
  if (cap > 4096) return MSL_ERR_UNSUPPORTED;
  const int Wn = msl::cdiv(cap, 64), ncls1 = ncls - 1;
  hipStream_t st = (hipStream_t)stream;
  if (msl_detect_select_ws_ints(N, P, ncls) >= (1ull << 31) || N * ncls1 > N * P) return MSL_ERR_UNSUPPORTED;
  MSL_LAUNCH(detect_prepare_kernel, dim3(msl::cdiv(N * P, 256)), dim3(256), 0, st, locs, scores, priors_c, probs, boxes,
                     N, P, ncls, select_ws, (int)msl_detect_select_ws_ints(N, P, ncls), ncand, N * ncls1);
  MSL_LAUNCH_CHECK();
  const dim3 gp(msl::cdiv(P, 256), N * ncls1);
  MSL_LAUNCH(detect_hist_kernel, gp, dim3(256), 0, st, probs, min_score, P, select_ws, ncand);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_threshold_kernel, dim3(N * ncls1), dim3(64), 0, st, select_ws, P, cap);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_compact_kernel, gp, dim3(256), 0, st, probs, min_score, P, select_ws);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_rank_kernel, gp, dim3(256), 0, st, probs, P, cap, select_ws, sorted_idx);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_mask_kernel, dim3(msl::cdiv(cap, 4), N * ncls1), dim3(256), 0, st, boxes, sorted_idx, ncand, max_overlap, P, cap, Wn, ncls1, mask);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_scan_kernel, dim3(N * ncls1), dim3(256), 0, st, mask, ncand, cap, Wn, keep_bits, nkept);
  MSL_LAUNCH_CHECK();
  MSL_LAUNCH(detect_finalize_kernel, dim3(N), dim3(256), 0, st, probs, boxes, sorted_idx, keep_bits, nkept, P, cap, Wn, ncls1, top_k, out_boxes, out_scores, out_labels, out_prior, out_count, tmp_scores, tmp_ref);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
