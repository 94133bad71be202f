// Stem: dense Conv3d(C_in -> 32, k3, stride (2,2,2) for cubes / (1,2,2) otherwise, pad 1, no bias)
// Reference: conv_bn (lesions3d/mobilenet.py:26-31) used as features[0] (lesions3d/ssd3d.py:60-61).
//
// HBM-bound layer (C_in = 1: 33.5 MB in, 134 MB out at 128^3 x 4).  One thread owns ONE output voxel
// and produces all 32 output channels from a single 27*C_in-value register patch, so the input is read
// once; every per-channel store is a 256-B row segment per wave (lanes run along W, the fastest axis).
// Weights are wave-uniform -> scalar loads feeding v_fma with an SGPR operand (no LDS, no VGPRs).
// BN statistics: per-workgroup (sum, sumsq) per channel via an LDS transpose, emitted as fp64 partials.
#include "common.hpp"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int STEM_COUT = 32;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w, float* __restrict__ y,
                                                       double* __restrict__ partials, int D, int H, int W,
                                                       int OD, int OH, int OW, int sd, int sh, int sw) {
  constexpr int K = CIN * 27;
  __shared__ float red[8][256];
  const int n = blockIdx.y;
  const int OS = OD * OH * OW;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const bool valid = o < OS;
  const int oo = valid ? o : 0;
  const int ow = oo % OW, oh = (oo / OW) % OH, od = oo / (OW * OH);

  float in[K];
  {
    const int id0 = od * sd - 1, ih0 = oh * sh - 1, iw0 = ow * sw - 1;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float* xc = x + ((size_t)n * CIN + ci) * D * H * W;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int id = id0 + kd;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = ih0 + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = iw0 + kw;
            const bool ok = valid && id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
            in[ci * 27 + kd * 9 + kh * 3 + kw] = ok ? xc[((size_t)id * H + ih) * W + iw] : 0.f;
          }
        }
      }
    }
  }

  const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;  // 8 groups of 32 threads
  double chs[4], chq[4];
  float* yo = y + (size_t)n * STEM_COUT * OS + oo;
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(w[(gp * 8 + c) * K + k], in[k], a);
      acc[c] = valid ? a : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (valid) yo[(size_t)(gp * 8 + c) * OS] = acc[c];
      red[c][threadIdx.x] = acc[c];
    }
    __syncthreads();
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = red[grp][i * 32 + lane32];
      s += v;
      q = fmaf(v, v, q);
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) {
      ds += __shfl_xor(ds, m, 64);
      dq += __shfl_xor(dq, m, 64);
    }
    chs[gp] = ds;
    chq[gp] = dq;
    __syncthreads();
  }
  if (partials && lane32 == 0) {
    const int NP = gridDim.x * gridDim.y;
    const int p = n * gridDim.x + blockIdx.x;
#pragma unroll
    for (int gp = 0; gp < 4; ++gp) {
      const int c = gp * 8 + grp;
      partials[(size_t)c * NP + p] = chs[gp];
      partials[((size_t)STEM_COUT + c) * NP + p] = chq[gp];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Stem forward on the matrix cores.  The VALU form above spends 27*Cin FMAs per output and channel (864 per voxel at
// Cin = 1): 23 us of pure VALU issue at 128^3 x 4 next to a 34 us HBM floor.  As a GEMM  Y[32 x pos] = W[32 x K] .
// im2col[K x pos]  (K = 27*Cin, padded to even) on v_mfma_f32_32x32x2_f32 the same work is 12 us of MFMA time and the
// kernel becomes a streaming one.  A wave owns 64 consecutive outputs of one row (two 32-column tiles):
//   A operand  lane (h, c): W[c][2kk + h]                               - registers, loaded once per wave
//   B operand  lane (h, c): x at tap k = 2kk + h of output column c     - one global load per kk, straight to registers
// (interior tiles take a mask-free path; volume borders a clamped-and-masked one).  Per-channel (sum, sumsq) are
// carried per lane across the wave's chunks and reduced once at the end (DPP), fp64 partials [2][32][NP].
// BF16OUT: y is bf16 (round to nearest even at the store; the statistics are taken from the fp32 accumulators).
template <int CIN, bool BF16OUT = false>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            void* __restrict__ y, double* __restrict__ partials,
                                                            int D, int H, int W, int OD, int OH, int OW, int sd, int sh,
                                                            int sw, int chunks_per_row, int chunks_per_n, int iters) {
  constexpr int K = CIN * 27, KS = (K + 1) / 2;
  __shared__ float red[4][2][32];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int n = blockIdx.y;
  const int OS = OD * OH * OW;
  const float* xn = x + (size_t)n * CIN * D * H * W;
  constexpr int EB = BF16OUT ? 2 : 4;  // bytes per output element
  char* yn = (char*)y + (size_t)n * STEM_COUT * OS * EB;

  float wa[KS];
  int tapoff[KS];  // element offset of the lane's tap kk inside the image
  int lowtap[KS];  // bit 0 / 1 / 2: the tap sits on the low (-1) side in d / h / w: outside the volume on a low-border tile
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 2 * kk + h;
    const bool vk = k < K;
    const int ci = k / 27, t = k % 27, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    wa[kk] = vk ? w[c * K + (vk ? k : 0)] : 0.f;
    tapoff[kk] = vk ? ((ci * D + kd) * H + kh) * W + kw : 0;
    lowtap[kk] = vk ? (kd == 0 ? 1 : 0) | (kh == 0 ? 2 : 0) | (kw == 0 ? 4 : 0) : 0;
  }
  float ssum[16], qsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ssum[r] = qsum[r] = 0.f;
  // Buffer addressing for the streaming accesses: address = descriptor base + per-lane 32-bit byte offset (VGPR, fixed
  // for the whole kernel) + per-tile scalar byte offset (SGPR) - nothing per access runs on the vector ALU.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(xn), 0, (int)((unsigned)CIN * D * H * W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(yn), 0, (int)((unsigned)STEM_COUT * OS * (unsigned)EB), 0x00020000);
  auto store_out = [&](float v, int voff, int soff) {
    if (BF16OUT) __builtin_amdgcn_raw_buffer_store_b16((short)msl::f2bf(v), ry, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, voff, soff, 0);
  };
  int lanetap4[KS];  // byte offset of (tap kk, column c) relative to the tile's first tap
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) lanetap4[kk] = (c * sw + tapoff[kk]) * 4;

  // The wave walks `iters` consecutive 64-column chunks = 2 * iters tiles of 32 output columns; the im2col column of
  // tile j+1 is requested before the MFMAs of tile j, so loads, matrix work and stores of consecutive tiles overlap.
  // Everything per tile is kept off the VALU as far as possible (it is the VALU, not the matrix pipe or HBM, that
  // bounds a naive version: ~450 vector instructions per tile against 14 MFMAs): tile coordinates advance with scalar
  // increments, addresses are (scalar base)[unsigned 32-bit lane offset].
  struct Tile {
    bool live;
    int od, oh, seg, half;  // wave-uniform
  };
  Tile cur;
  {
    // workgroups b and b+8 share an XCD (one L2): contiguous chunk ranges per XCD, so the input rows shared by
    // neighbouring output rows are fetched into one L2
    const int lblock = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int chunk0 = (lblock * 4 + wv) * iters;
    cur.live = chunk0 < chunks_per_n;
    const int cc = cur.live ? chunk0 : 0;
    cur.seg = cc % chunks_per_row;
    const int r0 = cc / chunks_per_row;
    cur.oh = r0 % OH;
    cur.od = r0 / OH;
    cur.half = 0;
  }
  int tiles_left = 2 * iters;  // of this wave
  auto advance = [&](Tile t) {  // the next tile of the walk
    --tiles_left;
    if (t.half == 0) {
      t.half = 1;
    } else {
      t.half = 0;
      if (++t.seg == chunks_per_row) {
        t.seg = 0;
        if (++t.oh == OH) {
          t.oh = 0;
          ++t.od;
        }
      }
    }
    t.live = t.live && tiles_left > 0 && t.od < OD;  // once past the end (or never started) a wave stays idle
    if (!t.live) t.od = t.oh = t.seg = 0;
    return t;
  };
  auto load_tile = [&](const Tile& t, float (&b)[KS]) {
    const int owt = t.seg * 64 + t.half * 32;
    const int id0 = t.od * sd - 1, ih0 = t.oh * sh - 1;
    const int rowbase = (id0 * H + ih0) * W - 1;
    const int ow = owt + c;
    const bool rows_in = id0 >= 0 && id0 + 2 < D && ih0 >= 0 && ih0 + 2 < H;
    const bool high_in = id0 + 2 < D && ih0 + 2 < H && (owt + 31) * sw + 1 < W;  // nothing beyond the high faces
    if (t.live && rows_in && owt >= 1 && (owt + 31) * sw + 1 < W) {  // wave-uniform: every tap of every lane is inside
      const int soff = __builtin_amdgcn_readfirstlane((rowbase + owt * sw) * 4);
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) b[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, lanetap4[kk], soff, 0));
    } else if (t.live && high_in) {
      // Low-border tile (first plane / first row / first 32 columns: with stride 2 and even sizes that is every other
      // tile of a row): only the taps on the -1 side can be outside, and which ones is known per lane and k-step.  Same
      // loads as the interior path; an outside tap is fetched one voxel further in along every axis (always inside the
      // image) and replaced by zero - no per-tap coordinate arithmetic (49 -> 44 us alone at 128^3 x 4).
      // Ablation of the result (tools/probes/r02_stemabl.sh): of 45 us the 14 stride-2 gathers of a tile cost 17.6, the
      // MFMAs 12, the 16 stores 8, the statistics 1.6 - the kernel is bound by vector-memory INSTRUCTION issue (~24
      // cycles per gather, ~9 per store and CU), not by bytes: bf16 output takes the same time.  Tried and dropped: a
      // third register set (two tiles of look-ahead: 191 VGPRs, occupancy 2, 52 us); the transposed product with four
      // 16-byte stores per lane (32-byte segments per channel: 48 us); 384-2048 workgroups per image (55-75 us).
      const int base4 = (rowbase + owt * sw) * 4;          // may be negative: folded into the per-lane offset
      const int shift4 = ((H + 1) * W + 1) * 4;
      const int lowmask = (id0 < 0 ? 1 : 0) | (ih0 < 0 ? 2 : 0) | ((owt == 0 && c == 0) ? 4 : 0);
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const bool out = (lowtap[kk] & lowmask) != 0;
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, lanetap4[kk] + base4 + (out ? shift4 : 0), 0, 0));
        b[kk] = out ? 0.f : v;
      }
    } else {
      const bool pv = t.live && ow < OW;
      const int iw0 = ow * sw - 1;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int k = 2 * kk + h, t27 = k % 27, kd = t27 / 9, kh = (t27 / 3) % 3, kw = t27 % 3;
        const int id = id0 + kd, ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = pv && k < K && id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W;
        const float v = xn[ok ? rowbase + ow * sw + tapoff[kk] : 0];
        b[kk] = ok ? v : 0.f;
      }
    }
  };
  const int loff4 = (4 * h * OS + c) * EB;  // per lane, bytes: rows 4h.., column c
  auto finish_tile = [&](const Tile& t, const float (&b)[KS]) {
    f32x16 acc = {0};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kk], b[kk], acc, 0, 0, 0);
    // D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int owt = t.seg * 64 + t.half * 32;
    const int yoff = __builtin_amdgcn_readfirstlane(((t.od * OH + t.oh) * OW + owt) * EB);  // scalar
    if (t.live && owt + 32 <= OW) {  // wave-uniform: whole tile inside
#pragma unroll
      for (int r = 0; r < 16; ++r) store_out(acc[r], loff4, yoff + ((r & 3) + 8 * (r >> 2)) * OS * EB);
    } else {
      const bool pv = t.live && owt + c < OW;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (pv) store_out(acc[r], loff4, yoff + ((r & 3) + 8 * (r >> 2)) * OS * EB);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r];  // exact zero in dead columns (their im2col column is zero)
      ssum[r] += v;
      qsum[r] = fmaf(v, v, qsum[r]);
    }
  };
  {
    float b0[KS], b1[KS];
    load_tile(cur, b0);
    while (tiles_left > 0) {  // wave-uniform
      const Tile t1 = advance(cur);
      load_tile(t1, b1);
      __builtin_amdgcn_sched_barrier(0);
      finish_tile(cur, b0);
      cur = advance(t1);
      load_tile(cur, b0);
      __builtin_amdgcn_sched_barrier(0);
      finish_tile(t1, b1);
    }
  }
  if (partials) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s = msl::half32_sum(ssum[r]), q = msl::half32_sum(qsum[r]);
      if (c == msl::HALF32_SUM_LANE) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[wv][0][row] = s;
        red[wv][1][row] = q;
      }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int ch = threadIdx.x;
      const double s = ((double)red[0][0][ch] + (double)red[1][0][ch]) + ((double)red[2][0][ch] + (double)red[3][0][ch]);
      const double q = ((double)red[0][1][ch] + (double)red[1][1][ch]) + ((double)red[2][1][ch] + (double)red[3][1][ch]);
      const int NP = gridDim.x * gridDim.y, p = n * gridDim.x + blockIdx.x;
      partials[(size_t)ch * NP + p] = s;
      partials[((size_t)STEM_COUT + ch) * NP + p] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stem forward, row-staged form (stride 2 along W, W % 4 == 0, OW % 32 == 0, Cin <= 2).  The register-fed kernel above is
// bound by vector-memory INSTRUCTION issue: 14 stride-2 dword gathers per 32-column tile (17.6 of its 45 us at 128^3 x 4).
// Here a wave fetches the 9 * Cin input rows of a 64-column chunk CONTIGUOUSLY - 33 lanes x 16 bytes per row, columns
// [2*ow0 - 4, 2*ow0 + 128) - parks them in its own LDS region (two buffers; the next chunk's rows are in flight during
// this chunk's MFMAs and stores) and reads every im2col operand from LDS (stride-2 ds_read_b32: 2-way bank conflicts, cheap).
// 9 contiguous loads per 64 columns instead of 28 gathers; rows / columns outside the volume are zeros written to LDS, so
// there is no border path at all.  Same tile walk, statistics and partial layout as stem_fwd_mfma_kernel.
constexpr int SFR_RP = 136;  // LDS row pitch in floats (132 used; 136 % 32 == 8)
template <int CIN, bool BF16OUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CIN == 1 ? 4 : 2))) void stem_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            void* __restrict__ y, double* __restrict__ partials, int D, int H,
                                                            int W, int OD, int OH, int OW, int sd, int sh, int chunks_per_row,
                                                            int chunks_per_n, int iters) {
  constexpr int K = CIN * 27, KS = (K + 1) / 2, NR = CIN * 9;
  extern __shared__ __align__(16) float rows_all[];  // [4 waves][2][NR * SFR_RP]
  __shared__ float red[4][2][32];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int n = blockIdx.y;
  const int OS = OD * OH * OW;
  const float* xn = x + (size_t)n * CIN * D * H * W;
  constexpr int EB = BF16OUT ? 2 : 4;
  char* yn = (char*)y + (size_t)n * STEM_COUT * OS * EB;
  float* myrows = rows_all + (size_t)wv * 2 * NR * SFR_RP;

  float wa[KS];
  int ldsoff[KS];  // LDS element of the lane's tap kk for output column 0 of the chunk's first half
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 2 * kk + h;
    const bool vk = k < K;
    const int ci = k / 27, t = k % 27, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    wa[kk] = vk ? w[c * K + (vk ? k : 0)] : 0.f;
    ldsoff[kk] = (vk ? (ci * 9 + kd * 3 + kh) * SFR_RP + kw : 0) + 3 + 2 * c;
  }
  float ssum[16], qsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ssum[r] = qsum[r] = 0.f;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(xn), 0, (int)((unsigned)CIN * D * H * W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(yn), 0, (int)((unsigned)STEM_COUT * OS * (unsigned)EB), 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int lblock = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int chunk0 = (lblock * 4 + wv) * iters;
  struct Chunk {
    bool live;
    int od, oh, ow0;  // wave-uniform
  };
  auto chunk_at = [&](int it) {
    Chunk q;
    const int ch = chunk0 + it;
    q.live = it < iters && ch < chunks_per_n;
    const int cc = __builtin_amdgcn_readfirstlane(q.live ? ch : 0);
    const int seg = cc % chunks_per_row, r0 = cc / chunks_per_row;
    q.ow0 = seg * 64;
    q.oh = r0 % OH;
    q.od = r0 / OH;
    return q;
  };
  // the rows of a chunk: lane l < 33 holds floats [4l, 4l + 4) of every row (columns 2*ow0 - 4 + 4l ..)
  auto issue = [&](const Chunk& q, u32x4 (&rr)[NR]) {
    const int id0 = q.od * sd - 1, ih0 = q.oh * sh - 1, col0 = 2 * q.ow0 - 4;
    const bool lane_in = lane < 33 && !(q.ow0 == 0 && lane == 0);  // columns -4 .. -1 of the first chunk: padding
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ci = r / 9, kd = (r % 9) / 3, kh = r % 3;
      const int id = id0 + kd, ih = ih0 + kh;
      const bool row_in = q.live && id >= 0 && id < D && ih >= 0 && ih < H;  // wave-uniform
      rr[r] = (u32x4){0u, 0u, 0u, 0u};
      if (row_in) {
        const int off = (((ci * D + id) * H + ih) * W + col0 + 4 * lane) * 4;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, lane_in ? off : 0, 0, 0);
        if (lane_in) rr[r] = v;
      }
    }
  };
  auto park = [&](const u32x4 (&rr)[NR], float* buf) {
    if (lane < 33) {
#pragma unroll
      for (int r = 0; r < NR; ++r) *reinterpret_cast<u32x4*>(buf + r * SFR_RP + 4 * lane) = rr[r];
    }
  };
  auto store_out = [&](float v, int voff, int soff) {
    if (BF16OUT) __builtin_amdgcn_raw_buffer_store_b16((short)msl::f2bf(v), ry, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, voff, soff, 0);
  };
  const int loff4 = (4 * h * OS + c) * EB;  // per lane, bytes: rows 4h.., column c
  u32x4 rr[NR];
  Chunk cur = chunk_at(0);
  issue(cur, rr);
  for (int it = 0; it < iters; ++it) {
    if (!cur.live) break;  // wave-uniform; a wave's chunks are consecutive
    float* buf = myrows + (it & 1) * NR * SFR_RP;
    park(rr, buf);  // (a wave's LDS operations execute in order: no barrier; the other buffer is still being read by nobody)
    const Chunk nxt = chunk_at(it + 1);
    issue(nxt, rr);  // in flight during this chunk's MFMAs and stores
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int owt = cur.ow0 + 32 * half;
      if (owt >= OW) continue;  // wave-uniform (OW % 32 == 0: a tile is inside or outside)
      f32x16 acc = {0};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kk], buf[ldsoff[kk] + 64 * half], acc, 0, 0, 0);
      const int yoff = __builtin_amdgcn_readfirstlane(((cur.od * OH + cur.oh) * OW + owt) * EB);
#pragma unroll
      for (int r = 0; r < 16; ++r) store_out(acc[r], loff4, yoff + ((r & 3) + 8 * (r >> 2)) * OS * EB);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        ssum[r] += v;
        qsum[r] = fmaf(v, v, qsum[r]);
      }
    }
    cur = nxt;
  }
  if (partials) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s = msl::half32_sum(ssum[r]), q = msl::half32_sum(qsum[r]);
      if (c == msl::HALF32_SUM_LANE) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[wv][0][row] = s;
        red[wv][1][row] = q;
      }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int ch = threadIdx.x;
      const double s = ((double)red[0][0][ch] + (double)red[1][0][ch]) + ((double)red[2][0][ch] + (double)red[3][0][ch]);
      const double q = ((double)red[0][1][ch] + (double)red[1][1][ch]) + ((double)red[2][1][ch] + (double)red[3][1][ch]);
      const int NP = gridDim.x * gridDim.y, p = n * gridDim.x + blockIdx.x;
      partials[(size_t)ch * NP + p] = s;
      partials[((size_t)STEM_COUT + ch) * NP + p] = q;
    }
  }
}

constexpr int STEM_FWD_BLOCKS_PER_IMAGE = 256;

// ---------------------------------------------------------------------------------------------
// Stem bwd-weight: dW[co][k] = sum_{n,o} dy[n,co,o] * x[n,ci,o*s-1+tap], k = ci*27 + tap.
// A GEMM with M = 32 (co), N = 27*Cin (k, padded to 32*NT), K = all output positions (1M at 128^3 x 4), on
// v_mfma_f32_32x32x2_f32.  Each wave walks chunks of <= 64 consecutive outputs of one output row: the dy tile
// (32 x 64, coalesced along W) and the 9*Cin input rows it needs are staged in the wave's own LDS region;
// the im2col operand is gathered from those rows (row pitch == 3 mod 32 -> the 27 taps hit distinct banks).
// The read of dy (134 MB) dominates; accumulators stay in registers over all chunks of a wave.
constexpr int SB_DY_LD = 65;     // dy tile pitch
constexpr int SB_ROW_LD = 131;   // input row pitch (2*64 + 1 = 129 needed; 131 % 32 == 3)

// APPLY: `dy` holds dL/d relu(bn(y)) (not yet through the BatchNorm backward); the kernel reads y as well and
// applies  dL/dy = scale * (gm - c1 - xhat * c2)  while staging the tile, so the 402 MB apply pass over the stem
// gradient never runs.  bnv = [scale, shift, mean, invstd, c1, c2] x 32 channels.
//
// MODE 2 (FUSED) goes one step further for a stem that feeds a stride-2 depthwise layer: `dy` is that layer's
// dL/dz (N,32,OD1,OH1,OW1) and w1p its taps, tap-major (27,32); the stem-activation gradient  g = dwconv^T(dL/dz)
// is rebuilt per chunk, so the 134 MB gradient tensor is neither written nor read.  bnv then has 8 rows
// [scale, shift, mean, invstd, c1, c2, cC, cE] (msl_bn_bwd_finalize_coef): dL/dy = scale * gm + (cC * y + cE).
struct StemFusedSrc {
  int OD1, OH1, OW1;  // dims of dL/dz
};

// BF16ACT: dy (MODE 2: dL/dz of block 1) and yraw are bf16 tensors (bf16 activation path); everything else as in fp32.
template <int CIN, int MODE, bool BF16ACT = false>
__global__ __launch_bounds__(256) void stem_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ slabs, int N, int D, int H, int W,
                                                              int OD, int OH, int OW, int sd, int sh, int sw,
                                                              int chunks_per_row, int total_chunks, int iters,
                                                              const float* __restrict__ yraw,
                                                              const float* __restrict__ bnv,
                                                              const float* __restrict__ w1p, StemFusedSrc fs) {
  constexpr bool APPLY = MODE >= 1;
  constexpr bool FUSED = MODE == 2;
  constexpr unsigned ES = BF16ACT ? 2u : 4u;  // bytes per element of dy / yraw
  constexpr int NT = (CIN * 27 + 31) / 32;
  constexpr int WAVE_LDS = 32 * SB_DY_LD + CIN * 9 * SB_ROW_LD;
  extern __shared__ __align__(16) float lds[];
  // readfirstlane: the wave index (hence the chunk coordinates, the tap parities, every weight address) is wave-uniform,
  // but only this tells the compiler so - scalar registers, scalar loads and scalar branches instead of vector ones
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* dyt = lds + wv * WAVE_LDS;
  float* rows = dyt + 32 * SB_DY_LD;
  const int OS = OD * OH * OW;

  // this lane's im2col column(s): tap k = nt*32 + (lane & 31)
  int tapoff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int k = nt * 32 + (lane & 31);
    if (k < CIN * 27) {
      const int ci = k / 27, t = k % 27;
      tapoff[nt] = (ci * 9 + t / 3) * SB_ROW_LD + (t % 3);
    } else {
      tapoff[nt] = 0;  // padding column of the GEMM: reads staged data, accumulates a value that the reduce kernel drops
    }
  }
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x16){0};

  if constexpr (FUSED) {
    // Software pipeline over (chunk, group of 8 channels): while group k is turned into gradient values, the loads of
    // group k+1 (8 y + 8 x 4 dz values per lane) are in flight; the next chunk's input rows and first group are
    // requested before this chunk's MFMAs.  Per axis the voxel index i takes taps
    //   even i: (k = 1, o = i/2)      odd i: (k = 0, o = (i+1)/2) and (k = 2, o = (i-1)/2)
    // so a chunk row has 1, 2 or 4 live (kd, kh) pairs (wave-uniform); all four slots are always loaded (dead ones
    // re-read a live address) and dead slots are skipped with a scalar branch.  Along W lane l loads
    // dz[(ow >> 1) + (ow & 1)]; an odd voxel takes its kw = 2 tap (dz[ow >> 1]) from its even neighbour by DPP.
    const int OS1 = fs.OD1 * fs.OH1 * fs.OW1;
    // workgroups b and b+8 run on the same XCD (one L2): give each XCD a contiguous range of chunks, so that the dz rows
    // shared by neighbouring stem rows are fetched into one L2 instead of eight
    const int lblock = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    struct Ctx {
      bool live, in, okl, odd;   // in/okl/odd: per lane
      bool vD[2], vH[2];         // wave-uniform from here on
      int kD[2], kH[2];
      int n, od, oh, ow0;
      int ly4, ldz4;             // per-lane BYTE offsets into a y row / a dz row
      unsigned yrow4;            // byte offset of y_raw (n, 0, od, oh, ow0)
      unsigned dzrow4[4];        // byte offsets of the dz rows of the four (td, th) slots at (n, 0, ., ., 0)
    };
    // Buffer addressing: descriptor base + per-lane VGPR byte offset + scalar byte offset.  The 160 loads of a chunk then
    // need one 32-bit scalar add each instead of a 64-bit pointer computation (the kernel was bound by its SCALAR
    // instruction stream: ~1000 scalar instructions per chunk, most of them address arithmetic).
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)msl::uniform_base(yraw), 0, (int)((unsigned)N * 32u * (unsigned)OS * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(
        (void*)msl::uniform_base(dy), 0, (int)((unsigned)N * 32u * (unsigned)OS1 * ES), 0x00020000);
    const unsigned OS4 = (unsigned)OS * ES, OS14 = (unsigned)OS1 * ES;
    auto ldact = [](const __amdgpu_buffer_rsrc_t& rs, int voff, int soff) {  // one element of dy / yraw as fp32
      if constexpr (BF16ACT) return msl::bf2f((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
      else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    };
    // the wave walks `iters` CONSECUTIVE chunks: coordinates are decoded once and then advanced with scalar increments
    // (four integer divisions per chunk otherwise), and consecutive rows share their dz rows in this CU's L1
    int w_seg, w_oh, w_od, w_n;
    {
      const int chunk0 = (lblock * 4 + wv) * iters;
      const int cc = chunk0 < total_chunks ? chunk0 : 0;
      w_seg = cc % chunks_per_row;
      int r = cc / chunks_per_row;
      w_oh = r % OH;
      r /= OH;
      w_od = r % OD;
      w_n = r / OD;
    }
    auto make_ctx = [&](int it) {
      Ctx c;
      const int chunk = (lblock * 4 + wv) * iters + it;
      c.live = it < iters && chunk < total_chunks;
      if (it > 0) {  // advance by one chunk
        if (++w_seg == chunks_per_row) {
          w_seg = 0;
          if (++w_oh == OH) {
            w_oh = 0;
            if (++w_od == OD) {
              w_od = 0;
              ++w_n;
            }
          }
        }
      }
      c.oh = c.live ? w_oh : 0;
      c.od = c.live ? w_od : 0;
      c.n = c.live ? w_n : 0;
      const int seg = c.live ? w_seg : 0;
      c.ow0 = seg * 64;
      const int npos = c.live ? min(64, OW - c.ow0) : 0;
      c.in = lane < npos;
      const int ly = c.in ? lane : 0;
      c.ly4 = ly * (int)ES;
      const int ow = c.ow0 + ly;
      c.odd = ow & 1;
      const int jl = (ow >> 1) + (ow & 1);
      c.okl = jl < fs.OW1;
      c.ldz4 = (c.okl ? jl : 0) * (int)ES;
      const bool pd = c.od & 1, ph = c.oh & 1;
      int idD[2], idH[2];
      idD[0] = pd ? (c.od + 1) >> 1 : c.od >> 1; c.kD[0] = pd ? 0 : 1; c.vD[0] = idD[0] < fs.OD1;
      idD[1] = c.od >> 1;                         c.kD[1] = 2;          c.vD[1] = pd;
      idH[0] = ph ? (c.oh + 1) >> 1 : c.oh >> 1; c.kH[0] = ph ? 0 : 1; c.vH[0] = idH[0] < fs.OH1;
      idH[1] = c.oh >> 1;                         c.kH[1] = 2;          c.vH[1] = ph;
      if (!c.vD[0]) idD[0] = c.od >> 1;  // dead slots re-read a live row
      if (!c.vH[0]) idH[0] = c.oh >> 1;
      c.yrow4 = ((unsigned)(c.n * 32) * (unsigned)OS + (unsigned)((c.od * OH + c.oh) * OW + c.ow0)) * ES;
      const unsigned dzn = (unsigned)(c.n * 32) * (unsigned)OS1;
#pragma unroll
      for (int t = 0; t < 4; ++t) c.dzrow4[t] = (dzn + (unsigned)((idD[t >> 1] * fs.OH1 + idH[t & 1]) * fs.OW1)) * ES;
      return c;
    };
    auto issue = [&](const Ctx& c, int cg, float (&yv)[8], float (&dz)[8][4]) {
      unsigned o = c.yrow4 + (unsigned)(cg * 8) * OS4;  // scalar, advanced by one channel per load
#pragma unroll
      for (int k = 0; k < 8; ++k, o += OS4)
        yv[k] = ldact(ry, c.ly4, (int)o);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        unsigned q = c.dzrow4[t] + (unsigned)(cg * 8) * OS14;
#pragma unroll
        for (int k = 0; k < 8; ++k, q += OS14)
          dz[k][t] = ldact(rdz, c.ldz4, (int)q);
      }
    };
    auto consume = [&](const Ctx& c, int cg, const float (&yv)[8], const float (&dz)[8][4]) {
      float g[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (c.vD[t >> 1] && c.vH[t & 1]) {  // wave-uniform
          const float* wk = w1p + (c.kD[t >> 1] * 9 + c.kH[t & 1] * 3) * 32 + cg * 8;  // tap-major (27, 32)
          float dn[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) dn[k] = msl::dpp_mov<0xA0>(dz[k][t]);  // quad_perm [0,0,2,2]: the even neighbour's dz
          // the taps are scalar operands; the lane classes are execution masks, not selects:
          //   even voxel: kw = 1 on its own dz;  odd voxel: kw = 2 on the neighbour's, kw = 0 on its own (if in range)
          if (c.odd) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = fmaf(wk[64 + k], dn[k], g[k]);
            if (c.okl) {
#pragma unroll
              for (int k = 0; k < 8; ++k) g[k] = fmaf(wk[k], dz[k][t], g[k]);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = fmaf(wk[32 + k], dz[k][t], g[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int co = cg * 8 + k;
        const float sc = bnv[co], sf = bnv[32 + co];
        const float gm = fmaf(yv[k], sc, sf) > 0.f ? g[k] : 0.f;
        const float v = fmaf(sc, gm, fmaf(bnv[192 + co], yv[k], bnv[224 + co]));  // scale*gm + (cC*y + cE)
        dyt[co * SB_DY_LD + lane] = c.in ? v : 0.f;
      }
    };
    float xreg[CIN * 9][3];
    bool xok[CIN * 9][3];
    auto issue_x = [&](const Ctx& c) {
      const int iw0 = c.ow0 * sw - 1;
      const int span = 64 * sw + 1;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          const int id = c.od * sd - 1 + rr / 3, ih = c.oh * sh - 1 + rr % 3;
          const bool rok = c.live && id >= 0 && id < D && ih >= 0 && ih < H;
          const float* src = x + (((size_t)c.n * CIN + ci) * D + (rok ? id : 0)) * H * W + (size_t)(rok ? ih : 0) * W;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int j = lane + 64 * t, iw = iw0 + j;
            xok[ci * 9 + rr][t] = rok && j <= span && iw >= 0 && iw < W;
            xreg[ci * 9 + rr][t] = src[xok[ci * 9 + rr][t] ? iw : 0];
          }
        }
    };
    float ya[8], da[8][4], yb[8], db[8][4];
    Ctx cur = make_ctx(0);
    issue_x(cur);
    issue(cur, 0, ya, da);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r2 = 0; r2 < CIN * 9; ++r2)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float v = xok[r2][t] ? xreg[r2][t] : 0.f;
          if (t == 2) msl::pin(v);  // else the load is sunk into the conditional store
          const int j = lane + 64 * t;
          if (j < SB_ROW_LD) rows[r2 * SB_ROW_LD + j] = v;
        }
      issue(cur, 1, yb, db);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 0, ya, da);
      issue(cur, 2, ya, da);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 1, yb, db);
      issue(cur, 3, yb, db);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 2, ya, da);
      const Ctx nxt = make_ctx(it + 1);
      issue_x(nxt);
      issue(nxt, 0, ya, da);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 3, yb, db);
#pragma unroll 4
      for (int s = 0; s < 32; ++s) {
        const int pos = 2 * s + (lane >> 5);
        const float a = dyt[(lane & 31) * SB_DY_LD + pos];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float b = rows[tapoff[nt] + pos * sw];
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
        }
      }
      cur = nxt;
    }
  } else
  for (int it = 0; it < iters; ++it) {
    const int chunk = (blockIdx.x * iters + it) * 4 + wv;
    const bool live = chunk < total_chunks;
    const int cc = live ? chunk : 0;
    const int seg = cc % chunks_per_row;
    int r = cc / chunks_per_row;
    const int oh = r % OH;
    r /= OH;
    const int od = r % OD, n = r / OD;
    const int ow0 = seg * 64;
    const int npos = live ? min(64, OW - ow0) : 0;
    // (the tiles are private to the wave and a wave's LDS operations execute in order: no barrier in this loop)
    // Issue every global load of the chunk first (32 dy + 9*CIN*3 input values per lane in flight), then
    // store to LDS: a load->store->load chain would expose the full memory latency 59 times per chunk.
    // Every load is unconditional on a clamped (always valid) address and masked afterwards: a predicated load
    // becomes a branch per load, and the compiler then waits for each one before issuing the next.  The
    // sched_barrier keeps the machine scheduler from re-interleaving the loads with their uses to save VGPRs.
    float dreg[32], yreg[APPLY ? 32 : 1];
    const bool in = lane < npos;
    {
      const size_t off = (size_t)n * 32 * OS + ((size_t)od * OH + oh) * OW + ow0 + (in ? lane : 0);
      if (BF16ACT) {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(dy) + off;
#pragma unroll
        for (int co = 0; co < 32; ++co) dreg[co] = msl::bf2f(src[(size_t)co * OS]);
        if (APPLY) {
          const unsigned short* ysrc = reinterpret_cast<const unsigned short*>(yraw) + off;
#pragma unroll
          for (int co = 0; co < 32; ++co) yreg[co] = msl::bf2f(ysrc[(size_t)co * OS]);
        }
      } else {
        const float* src = dy + off;
#pragma unroll
        for (int co = 0; co < 32; ++co) dreg[co] = src[(size_t)co * OS];
        if (APPLY) {
          const float* ysrc = yraw + off;
#pragma unroll
          for (int co = 0; co < 32; ++co) yreg[co] = ysrc[(size_t)co * OS];
        }
      }
    }
    float xreg[CIN * 9][3];
    bool xok[CIN * 9][3];
    {
      const int iw0 = ow0 * sw - 1;
      const int span = 64 * sw + 1;  // j in [0, span]
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          const int id = od * sd - 1 + rr / 3, ih = oh * sh - 1 + rr % 3;
          const bool rok = live && id >= 0 && id < D && ih >= 0 && ih < H;
          const float* src = x + (((size_t)n * CIN + ci) * D + (rok ? id : 0)) * H * W + (size_t)(rok ? ih : 0) * W;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int j = lane + 64 * t, iw = iw0 + j;
            xok[ci * 9 + rr][t] = rok && j <= span && iw >= 0 && iw < W;
            xreg[ci * 9 + rr][t] = src[xok[ci * 9 + rr][t] ? iw : 0];
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // everything above is in flight before anything below waits
    if (APPLY) {
#pragma unroll
      for (int co = 0; co < 32; ++co) {
        const float yv = yreg[co];
        const float sc = bnv[co], sf = bnv[32 + co], mu = bnv[64 + co], is = bnv[96 + co];
        const float gm = fmaf(yv, sc, sf) > 0.f ? dreg[co] : 0.f;
        dreg[co] = sc * (gm - bnv[128 + co] - ((yv - mu) * is) * bnv[160 + co]);
      }
    }
#pragma unroll
    for (int co = 0; co < 32; ++co) dreg[co] = in ? dreg[co] : 0.f;
#pragma unroll
    for (int r2 = 0; r2 < CIN * 9; ++r2)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        xreg[r2][t] = xok[r2][t] ? xreg[r2][t] : 0.f;
        if (t == 2) asm volatile("" : "+v"(xreg[r2][t]));  // pin it here: else the load is sunk into the conditional store
      }
#pragma unroll
    for (int co = 0; co < 32; ++co) dyt[co * SB_DY_LD + lane] = dreg[co];
#pragma unroll
    for (int r2 = 0; r2 < CIN * 9; ++r2)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int j = lane + 64 * t;
        if (j < SB_ROW_LD) rows[r2 * SB_ROW_LD + j] = xreg[r2][t];
      }
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int pos = 2 * s + (lane >> 5);
      const float a = dyt[(lane & 31) * SB_DY_LD + pos];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float b = rows[tapoff[nt] + pos * sw];
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
      }
    }
  }
  // fixed-order reduction over the 4 waves through LDS, then one slab per block: slab[32][32*NT]
  __syncthreads();
  float* red = lds;
  for (int w2 = 3; w2 >= 1; --w2) {
    if (wv == w2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
          float* p = red + row * (32 * NT) + nt * 32 + (lane & 31);
          if (w2 == 3) *p = acc[nt][rg];
          else *p += acc[nt][rg];
        }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float* out = slabs + (size_t)blockIdx.x * 32 * 32 * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        const int idx = row * (32 * NT) + nt * 32 + (lane & 31);
        out[idx] = acc[nt][rg] + red[idx];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused stem weight gradient, TILE-STAGED form (MODE 2 semantics; stride 2 on every axis, W == 128 -> OW == 64,
// OH % 8 == 0, Cin <= 2): what the 128^3 training configurations run.
// The wave-private kernel above gives a lane ONE output position: 5 dword loads per (position, channel), 4 of them the
// dz slots of which 2.25 are live on average - 187 vector-memory instructions and 4 load->use round trips per 64
// positions, 2 waves per SIMD: bound by instruction issue and latency at 2 TB/s.  Here a WORKGROUP owns a tile of 256
// positions = 4 output rows (n, od, oh = p + 2*(4*g4 + r), r = 0..3) of ONE oh parity p, so the live (kd, kh) slot set
// is uniform over the tile, and a lane owns FOUR consecutive positions ow = 4m .. 4m+3 of one row:
//   * y: one 16-byte load per (lane, channel); dz: one 8-byte load (dz[2m], dz[2m+1]) per (lane, channel, slot), the
//     third value dz[2m+2] is the next lane's first (DPP row_shl:1 inside the 16-lane row; the row end shifts in the
//     zero the border needs) - 45 instead of 748 vector loads per 256 positions and wave;
//   * produce phase: wave w rebuilds g = dwconv^T(dz) for channels 8w..8w+7, applies the folded BatchNorm backward and
//     writes the rows [c][256] of the shared g tile (ds_write_b128); the 36*Cin input rows of the tile go to LDS as
//     contiguous 16-byte row loads (pitch 131: the 27 taps of a position sit on distinct banks);
//   * MFMA phase: wave w takes output row w of the tile (64 positions = 32 k-steps of v_mfma_f32_32x32x2_f32, M = 32
//     channels, N = 27*Cin taps): the g operand comes as one ds_read_b128 per 4 k-steps (k-step s = 4q + j multiplies
//     positions 8q + j and 8q + 4 + j), the im2col operand as one stride-2 ds_read_b32 per MFMA;
//   * the next tile's loads are issued between the two phases and land during the MFMAs.
// Same slab layout / reduction as stem_bwd_weight_kernel (one [32][32*NT] slab per workgroup).
constexpr int SBT_LDA = 260;  // g-tile pitch in floats: rows 16-byte aligned and (pitch / 4) odd -> conflict-free ds_read_b128
constexpr int SBT_RP = 131;   // input-row pitch (1 + 128 used); 131 % 32 == 3

// NW = waves per workgroup (4 or 8): wave w produces channels [CPW*w, CPW*(w+1)), CPW = 32 / NW, and multiplies positions
// [PPW*w, PPW*(w+1)), PPW = 256 / NW, of the tile.
template <int CIN, bool BF16ACT, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2))) void stem_bww_tile_kernel(const void* __restrict__ dzp, const void* __restrict__ yp,
                                                            const float* __restrict__ x, float* __restrict__ slabs, int N,
                                                            int D, int H, int OD, int OH, int tiles_total, int iters,
                                                            const float* __restrict__ bnv, const float* __restrict__ w1p) {
  constexpr int CPW = 32 / NW, PPW = 256 / NW;
  constexpr unsigned ES = BF16ACT ? 2u : 4u;
  constexpr int NT = (CIN * 27 + 31) / 32;
  constexpr int W = 128, OW = 64, OW1 = 32;
  constexpr int NROWS = 4 * CIN * 9;             // input rows of a tile
  constexpr int NXL = (NROWS / 2 + NW - 1) / NW;       // row-pair loads per wave and tile
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __align__(16) float lds[];
  float* At = lds;                               // [32][SBT_LDA]
  float* rows = lds + 32 * SBT_LDA;              // [NROWS][SBT_RP], row = (r * CIN + ci) * 9 + kd * 3 + kh
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane >> 4, m = lane & 15, h = lane >> 5;
  const int OD1 = (OD - 1) / 2 + 1, OH1 = OH >> 1;
  const unsigned OS = (unsigned)OD * OH * OW, OS1 = (unsigned)OD1 * OH1 * OW1;
  const int gpc = OH >> 3;                       // tiles per (n, od, parity)

  int tapoff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int k = nt * 32 + (lane & 31);
    if (k < CIN * 27) {
      const int ci = k / 27, t = k % 27;
      tapoff[nt] = (ci * 9 + t / 3) * SBT_RP + (t % 3);
    } else {
      tapoff[nt] = 0;  // padding column of the GEMM: accumulates a value the reduction drops
    }
  }
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x16){0};
  for (int R = threadIdx.x; R < NROWS; R += NW * 64) rows[R * SBT_RP] = 0.f;  // column iw = -1 of every row

  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(yp), 0, (int)((unsigned)N * 32u * OS * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(dzp), 0, (int)((unsigned)N * 32u * OS1 * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)msl::uniform_base(x), 0, (int)((unsigned)N * CIN * D * H * W * 4u), 0x00020000);

  // the input rows this lane fetches: pair u of the wave = rows 2*(wv + 4u) + h, constant over the tiles
  // lane-constant parts: byte offset of (ci, kd, 4 * out + kh, column) inside x, LDS element of the row, and which border
  // can put the row outside the volume (then the lane stores zeros)
  int xr_off[NXL], xr_dst[NXL];
  bool xr_valid[NXL], xr_kd0[NXL], xr_kd2[NXL], xr_top[NXL];
#pragma unroll
  for (int u = 0; u < NXL; ++u) {
    const int R = 2 * (wv + NW * u) + h;
    xr_valid[u] = R < NROWS;
    const int Rc = xr_valid[u] ? R : 0;
    const int out = Rc / (CIN * 9), rem = Rc % (CIN * 9), ci = rem / 9, kd = (rem % 9) / 3, kh = rem % 3;
    xr_off[u] = (((ci * D + kd) * H + 4 * out + kh) * W + 4 * (lane & 31)) * 4;
    xr_dst[u] = Rc * SBT_RP + 1 + 4 * (lane & 31);
    xr_kd0[u] = kd == 0;
    xr_kd2[u] = kd == 2;
    xr_top[u] = out == 0 && kh == 0;
  }
  // input row (id, ih) = (2 od - 1 + kd, 4 (4 g4 + out) + 2 p - 1 + kh): outside the volume only for kd == 0 at od == 0,
  // kd == 2 at the last plane of an odd D, and kh == 0 in the first row of the first tile of parity 0
#define XR_OK(u, t) (!((xr_kd0[u] & ((t).od == 0)) | (xr_kd2[u] & (2 * (t).od + 1 >= D)) | (xr_top[u] & (((t).g4 | (t).p) == 0))))

  // workgroups b and b+8 run on the same XCD (one L2): give each XCD a contiguous range of tiles
  const int lblock = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int tile0 = lblock * iters;
  struct Tile {
    int n, od, p, g4;  // wave-uniform
  };
  auto tile_first = [&]() {  // the workgroup's tiles are consecutive: decode once, then count
    Tile t;
    const int T = __builtin_amdgcn_readfirstlane(tile0 < tiles_total ? tile0 : tiles_total - 1);
    t.g4 = T % gpc;
    int q = T / gpc;
    t.p = q & 1;
    q >>= 1;
    t.od = q % OD;
    t.n = q / OD;
    return t;
  };
  auto tile_next = [&](Tile t) {
    if (++t.g4 == gpc) {
      t.g4 = 0;
      if ((t.p ^= 1) == 0 && ++t.od == OD) {
        t.od = 0;
        ++t.n;
      }
    }
    return t;
  };
  u32x4 Y[CPW];                       // fp32: 4 values; bf16: .xy hold 4 values
  u32x2 DZ[CPW][4];                   // slot t = td * 2 + th; fp32: 2 values; bf16: .x holds 2 values
  u32x4 X[NXL];
  // Two input channels: 9 row-pair loads per wave would keep 36 more registers alive across the MFMA phase (spills at
  // 2 waves per SIMD), so there the rows are requested inside the produce phase, once the dz registers are dead.
  constexpr bool XLATE = CIN > 1;
  auto issue_x = [&](const Tile& t) {
#pragma unroll
    for (int u = 0; u < NXL; ++u) {
      const int xt = (((t.n * CIN * D + 2 * t.od - 1) * H + 16 * t.g4 + 2 * t.p - 1) * W) * 4;  // scalar
      X[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, XR_OK(u, t) ? xr_off[u] + xt : 0, 0, 0);
    }
  };
  auto issue = [&](const Tile& t) {
    const int qr = 4 * t.g4 + r;
    const int yv = (int)(((unsigned)(t.p + 2 * qr) * OW + 4u * m) * ES);
    unsigned ys = (((unsigned)(t.n * 32 + CPW * wv)) * OS + (unsigned)t.od * OH * OW) * ES;
#pragma unroll
    for (int k = 0; k < CPW; ++k, ys += OS * ES) {
      if constexpr (BF16ACT) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ry, yv, (int)ys, 0);
        Y[k] = (u32x4){v.x, v.y, 0u, 0u};
      } else {
        Y[k] = __builtin_amdgcn_raw_buffer_load_b128(ry, yv, (int)ys, 0);
      }
    }
    const bool pd = t.od & 1;
    int idD[2], idH[2];
    idD[0] = pd ? (t.od + 1) >> 1 : t.od >> 1;
    if (idD[0] >= OD1) idD[0] = t.od >> 1;  // dead slot: re-read a live plane
    idD[1] = t.od >> 1;
    idH[0] = t.p ? qr + 1 : qr;
    if (idH[0] >= OH1) idH[0] = qr;         // (per lane) masked in consume
    idH[1] = qr;
#pragma unroll
    for (int td = 0; td < 2; ++td) {
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        const int dv = (int)(((unsigned)idH[th] * OW1 + 2u * m) * ES);
        unsigned ds = (((unsigned)(t.n * 32 + CPW * wv)) * OS1 + (unsigned)idD[td] * OH1 * OW1) * ES;
#pragma unroll
        for (int k = 0; k < CPW; ++k, ds += OS1 * ES) {
          if constexpr (BF16ACT) {
            DZ[k][td * 2 + th] = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rdz, dv, (int)ds, 0), 0u};
          } else {
            DZ[k][td * 2 + th] = __builtin_amdgcn_raw_buffer_load_b64(rdz, dv, (int)ds, 0);
          }
        }
      }
    }
    if constexpr (!XLATE) issue_x(t);
  };
  auto val = [](unsigned bits, int e) {  // element e of a pair of packed bf16
    return __uint_as_float(e ? (bits & 0xffff0000u) : (bits << 16));
  };
  auto consume = [&](const Tile& t) {
    const int qr = 4 * t.g4 + r;
    const bool pd = t.od & 1;
    const bool okH0 = (t.p ? qr + 1 : qr) < OH1;          // per lane
    const bool edgeH = t.p && t.g4 == gpc - 1;            // wave-uniform: only the last tile of parity 1 has such lanes
    const bool vD[2] = {(pd ? (t.od + 1) >> 1 : t.od >> 1) < OD1, pd};
    const bool vH[2] = {true, (bool)t.p};
    const int kD[2] = {pd ? 0 : 1, 2}, kH[2] = {t.p ? 0 : 1, 2};
    // slot-outer, channel-inner: the three taps of a slot for the wave's 8 channels are three s_load_dwordx8 (tap-major
    // weights), i.e. one scalar round trip per live slot and one for the BatchNorm vectors - not one per (slot, channel)
    // packed pairs: ge = (g0, g2) and go = (g1, g3) of the lane's four positions, so that a slot is three v_pk_fma_f32 with
    // a broadcast scalar tap:  ge += w1 * (a, b);  go += w2 * (a, b);  go += w0 * (b, nb)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f ge[CPW], go[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) ge[k] = go[k] = (v2f){0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int td = s >> 1, th = s & 1;
      if (vD[td] && vH[th]) {  // wave-uniform
        const float* wk = w1p + (kD[td] * 9 + kH[th] * 3) * 32 + CPW * wv;  // tap-major (27, 32)
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          float a, b;
          if constexpr (BF16ACT) { a = val(DZ[k][s].x, 0); b = val(DZ[k][s].x, 1); }
          else { a = __uint_as_float(DZ[k][s].x); b = __uint_as_float(DZ[k][s].y); }
          if (th == 0 && edgeH) { a = okH0 ? a : 0.f; b = okH0 ? b : 0.f; }
          const float nb = msl::dpp_mov<0x101>(a);  // row_shl:1: dz[2m+2] = the next lane's first value, 0 past the row
          const float w0 = wk[k], w1 = wk[32 + k], w2 = wk[64 + k];
          const v2f ab = {a, b}, bn = {b, nb};
          ge[k] = __builtin_elementwise_fma((v2f){w1, w1}, ab, ge[k]);
          go[k] = __builtin_elementwise_fma((v2f){w2, w2}, ab, go[k]);
          go[k] = __builtin_elementwise_fma((v2f){w0, w0}, bn, go[k]);
        }
      }
    }
    if constexpr (XLATE) {
      issue_x(t);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = CPW * wv + k;
      const float sc = bnv[c], sf = bnv[32 + c], cC = bnv[192 + c], cE = bnv[224 + c];
      float y0, y1, y2, y3;
      if constexpr (BF16ACT) { y0 = val(Y[k].x, 0); y1 = val(Y[k].x, 1); y2 = val(Y[k].y, 0); y3 = val(Y[k].y, 1); }
      else { y0 = __uint_as_float(Y[k].x); y1 = __uint_as_float(Y[k].y); y2 = __uint_as_float(Y[k].z); y3 = __uint_as_float(Y[k].w); }
      // scale*gm + (cC*y + cE), two values per v_pk_fma_f32
      const v2f sc2 = {sc, sc}, sf2 = {sf, sf}, cC2 = {cC, cC}, cE2 = {cE, cE};
      const v2f ya = {y0, y1}, yb = {y2, y3};
      const v2f ta = __builtin_elementwise_fma(ya, sc2, sf2), tb = __builtin_elementwise_fma(yb, sc2, sf2);
      const v2f ga = {ta.x > 0.f ? ge[k].x : 0.f, ta.y > 0.f ? go[k].x : 0.f};
      const v2f gb = {tb.x > 0.f ? ge[k].y : 0.f, tb.y > 0.f ? go[k].y : 0.f};
      const v2f va = __builtin_elementwise_fma(sc2, ga, __builtin_elementwise_fma(cC2, ya, cE2));
      const v2f vb = __builtin_elementwise_fma(sc2, gb, __builtin_elementwise_fma(cC2, yb, cE2));
      *reinterpret_cast<float4*>(At + c * SBT_LDA + 4 * lane) = make_float4(va.x, va.y, vb.x, vb.y);
    }
#pragma unroll
    for (int u = 0; u < NXL; ++u) {
      const bool ok = XR_OK(u, t);
      if (xr_valid[u]) {
        float* dst = rows + xr_dst[u];
        dst[0] = ok ? __uint_as_float(X[u].x) : 0.f;
        dst[1] = ok ? __uint_as_float(X[u].y) : 0.f;
        dst[2] = ok ? __uint_as_float(X[u].z) : 0.f;
        dst[3] = ok ? __uint_as_float(X[u].w) : 0.f;
      }
    }
  };

  Tile cur = tile_first();
  issue(cur);
  for (int it = 0; it < iters; ++it) {
    if (tile0 + it >= tiles_total) break;  // uniform over the workgroup
    consume(cur);
    __syncthreads();
    if (it + 1 < iters && tile0 + it + 1 < tiles_total) {  // uniform
      cur = tile_next(cur);
      issue(cur);  // in flight during the MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* arow = At + (lane & 31) * SBT_LDA + PPW * wv + 4 * h;
    const float* brow = rows + ((PPW * wv) >> 6) * (CIN * 9 * SBT_RP) + 2 * ((PPW * wv) & 63) + 8 * h;
#pragma unroll
    for (int q = 0; q < PPW / 8; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * q);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float b = brow[tapoff[nt] + 2 * (8 * q + j)];
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b, acc[nt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // fixed-order reduction over the 4 waves through LDS, then one slab per block: slab[32][32*NT]
  __syncthreads();
  float* red = lds;
  for (int w2 = NW - 1; w2 >= 1; --w2) {
    if (wv == w2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
          float* p = red + row * (32 * NT) + nt * 32 + (lane & 31);
          if (w2 == NW - 1) *p = acc[nt][rg];
          else *p += acc[nt][rg];
        }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float* out = slabs + (size_t)blockIdx.x * 32 * 32 * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        const int idx = row * (32 * NT) + nt * 32 + (lane & 31);
        out[idx] = acc[nt][rg] + red[idx];
      }
  }
}

#undef XR_OK

__global__ __launch_bounds__(256) void stem_bwd_weight_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                     int K, int NT, int nslabs) {
  __shared__ float lds[8 * 33];
  // reduce the padded [32][32*NT] slab image, then drop the padding columns
  const int count = 32 * 32 * NT;
  const float t = msl::reduce_slabs_256(slabs, (size_t)count, count, nslabs, lds);
  const int i = blockIdx.x * 32 + (threadIdx.x & 31);
  if ((threadIdx.x >> 5) == 0 && i < count) {
    const int co = i / (32 * NT), k = i % (32 * NT);
    if (k < K) dw[co * K + k] = t;
  }
}

constexpr int STEM_BW_BLOCKS = 512;
// workgroups of the stem weight gradient (= partial slabs): two per CU
static inline int stem_bw_blocks() {
  constexpr int v = STEM_BW_BLOCKS;
  return v > 0 ? v : STEM_BW_BLOCKS;
}

}  // namespace

extern "C" {

static inline int stem_fwd_blocks(int OD, int OH, int OW) {
  constexpr int per_image = STEM_FWD_BLOCKS_PER_IMAGE;
  return std::min(per_image, msl::cdiv(OD * OH * msl::cdiv(OW, 64), 4));
}

int msl_stem_conv_fwd_num_partials(int N, int OD, int OH, int OW) { return N * stem_fwd_blocks(OD, OH, OW); }

// x (N,Cin,D,H,W) -> y (N,32,OD,OH,OW) raw conv output (fp32, or bf16 when `bf16_out`) + fp64 stat partials [2][32][NP].
static int stem_fwd_impl(const float* x, const float* w, void* y, double* partials, int N, int Cin, int D, int H, int W,
                         int sd, int sh, int sw, bool bf16_out, void* stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || sd < 1 || sd > 2 || sh < 1 || sh > 2 || sw < 1 || sw > 2)
    return MSL_ERR_ARG;
  if ((long long)Cin * D * H * W >= (1ll << 30)) return MSL_ERR_UNSUPPORTED;  // in-image byte offsets are 32-bit
  const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
  const int chunks_per_row = msl::cdiv(OW, 64), chunks_per_n = OD * OH * chunks_per_row;
  const int nb = stem_fwd_blocks(OD, OH, OW);
  const int iters = msl::cdiv(chunks_per_n, nb * 4);
  dim3 grid(nb, N);
  hipStream_t st = (hipStream_t)stream;
  constexpr int rows_on = 1;
  if (rows_on && sw == 2 && W % 4 == 0 && OW % 32 == 0 && Cin <= 2) {
    // row-staged form (contiguous 16-byte row loads, operands from LDS)
    const size_t lds = (size_t)4 * 2 * Cin * 9 * SFR_RP * sizeof(float);
#define MSL_STEM_FR(CI, B_)                                                                                             \
  do {                                                                                                                  \
    if (lds > 64 * 1024) {                                                                                              \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_rows_kernel<CI, B_>),                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e_ != hipSuccess) return (int)e_;                                                                             \
    }                                                                                                                   \
    MSL_LAUNCH((stem_fwd_rows_kernel<CI, B_>), grid, dim3(256), lds, st, x, w, y, partials, D, H, W, OD, OH, OW, sd, \
                       sh, chunks_per_row, chunks_per_n, iters);                                                        \
  } while (0)
    if (Cin == 1) { if (bf16_out) MSL_STEM_FR(1, true); else MSL_STEM_FR(1, false); }
    else { if (bf16_out) MSL_STEM_FR(2, true); else MSL_STEM_FR(2, false); }
#undef MSL_STEM_FR
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
#define MSL_STEM_FW(CI)                                                                                                 \
  do {                                                                                                                  \
    if (bf16_out)                                                                                                       \
      MSL_LAUNCH((stem_fwd_mfma_kernel<CI, true>), grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, OW, \
                         sd, sh, sw, chunks_per_row, chunks_per_n, iters);                                              \
    else                                                                                                                \
      MSL_LAUNCH((stem_fwd_mfma_kernel<CI, false>), grid, dim3(256), 0, st, x, w, y, partials, D, H, W, OD, OH, \
                         OW, sd, sh, sw, chunks_per_row, chunks_per_n, iters);                                          \
  } while (0)
  switch (Cin) {
    case 1: MSL_STEM_FW(1); break;
    case 2: MSL_STEM_FW(2); break;
    case 3: MSL_STEM_FW(3); break;
    case 4: MSL_STEM_FW(4); break;
    default: return MSL_ERR_UNSUPPORTED;
  }
#undef MSL_STEM_FW
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_stem_conv_fwd(const float* x, const float* w, float* y, double* partials, int N, int Cin, int D,
                      int H, int W, int sd, int sh, int sw, void* stream) {
  return stem_fwd_impl(x, w, y, partials, N, Cin, D, H, W, sd, sh, sw, false, stream);
}

// the same with a bf16 output tensor (bf16 activation path: csrc/bf16.hip)
int msl_stem_conv_fwd_bf16(const float* x, const float* w, void* y_bf16, double* partials, int N, int Cin, int D,
                           int H, int W, int sd, int sh, int sw, void* stream) {
  return stem_fwd_impl(x, w, y_bf16, partials, N, Cin, D, H, W, sd, sh, sw, true, stream);
}

size_t msl_stem_conv_bwd_weight_workspace_bytes(int Cin) {
  const int NT = (Cin * 27 + 31) / 32;
  return (size_t)stem_bw_blocks() * 32 * 32 * NT * sizeof(float);
}

// number of [32][32*NT] slabs the stem weight-gradient kernels leave in `workspace` (NT = ceil(Cin*27 / 32))
int msl_stem_conv_bwd_weight_nslabs(int N, int D, int H, int W, int sd, int sh, int sw) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || sd < 1 || sh < 1 || sw < 1) return MSL_ERR_ARG;
  const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
  return std::min(stem_bw_blocks(), msl::cdiv(N * OD * OH * msl::cdiv(OW, 64), 4));
}

// dw (32,Cin,3,3,3) = correlation of dy (N,32,OD,OH,OW) with x (N,Cin,D,H,W).  For all three forms: dw == NULL leaves
// the partial slabs in `workspace` (deferred reduction, msl_grad_reduce_batch kind 2).
static int stem_bww_impl(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                         int sd, int sh, int sw, const float* yraw, const float* bnv, const float* w1, void* stream,
                         bool bf16act = false);

int msl_stem_conv_bwd_weight(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D,
                             int H, int W, int sd, int sh, int sw, void* stream) {
  return stem_bww_impl(dy, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, nullptr, nullptr, nullptr, stream);
}

// The stem feeds a stride-2 depthwise layer (taps w1_t (27,32), tap-major as written by
// msl_dwconv_s2_bwd_bnreduce_bww) and `dz` is THAT layer's dL/dz (N,32,ceil(OD/2),ceil(OH/2),ceil(OW/2)): the
// gradient of the stem activation is rebuilt from dz inside the kernel.  bn_vec: the (8,32) block of
// msl_bn_bwd_finalize_coef.
int msl_stem_conv_bwd_weight_fused(const float* dz, const float* w1_t, const float* yraw, const float* bn_vec,
                                   const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                                   int sd, int sh, int sw, void* stream) {
  if (!dz || !w1_t || !yraw || !bn_vec) return MSL_ERR_ARG;
  {  // the kernel addresses y and dz with 32-bit byte offsets
    const long long OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
    if ((long long)N * 32 * OD * OH * OW * 4 >= (1ll << 32)) return MSL_ERR_UNSUPPORTED;
  }
  return stem_bww_impl(dz, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, yraw, bn_vec, w1_t, stream);
}

// g = dL/d relu(bn(y)): BatchNorm backward applied on load.  bn_vec = (6, 32) fp32 rows [scale, shift, mean, invstd,
// c1 = dbeta/n, c2 = dgamma/n] as produced by msl_bn_finalize + msl_bn_bwd_finalize.
int msl_stem_conv_bwd_weight_bnapply(const float* g, const float* yraw, const float* bn_vec, const float* x, float* dw,
                                     float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                     void* stream) {
  if (!yraw || !bn_vec) return MSL_ERR_ARG;
  return stem_bww_impl(g, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, yraw, bn_vec, nullptr, stream);
}

// bf16 activation path: dz (dL/dz of block 1) and yraw are bf16 tensors; otherwise msl_stem_conv_bwd_weight_fused
int msl_stem_conv_bwd_weight_fused_bf16(const void* dz, const float* w1_t, const void* yraw, const float* bn_vec, const float* x,
                                        float* dw, float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                        void* stream) {
  if (!dz || !w1_t || !yraw || !bn_vec) return MSL_ERR_ARG;
  {
    const long long OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
    if ((long long)N * 32 * OD * OH * OW * 2 >= (1ll << 32)) return MSL_ERR_UNSUPPORTED;
  }
  return stem_bww_impl((const float*)dz, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, (const float*)yraw, bn_vec, w1_t, stream,
                       true);
}

// bf16 activation path: g (dL/d relu(bn(y))) and yraw are bf16 tensors; same slabs / reduction as msl_stem_conv_bwd_weight_bnapply
int msl_stem_conv_bwd_weight_bnapply_bf16(const void* g, const void* yraw, const float* bn_vec, const float* x, float* dw,
                                          float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                          void* stream) {
  if (!yraw || !bn_vec) return MSL_ERR_ARG;
  return stem_bww_impl((const float*)g, x, dw, workspace, N, Cin, D, H, W, sd, sh, sw, (const float*)yraw, bn_vec, nullptr,
                       stream, true);
}

static int stem_bww_impl(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                         int sd, int sh, int sw, const float* yraw, const float* bnv, const float* w1, void* stream,
                         bool bf16act) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || sd < 1 || sd > 2 || sh < 1 || sh > 2 || sw < 1 || sw > 2)
    return MSL_ERR_ARG;
  const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
  const int chunks_per_row = msl::cdiv(OW, 64);
  const int total_chunks = N * OD * OH * chunks_per_row;
  const int nblocks = std::min(stem_bw_blocks(), msl::cdiv(total_chunks, 4));
  const int iters = msl::cdiv(total_chunks, nblocks * 4);
  hipStream_t st = (hipStream_t)stream;
  const int NT = (Cin * 27 + 31) / 32;
  const size_t lds = (size_t)4 * (32 * SB_DY_LD + Cin * 9 * SB_ROW_LD) * sizeof(float);
  const StemFusedSrc fs{(OD - 1) / 2 + 1, (OH - 1) / 2 + 1, (OW - 1) / 2 + 1};
  constexpr int tile_on = 1;
  if (tile_on && w1 && sd == 2 && sh == 2 && sw == 2 && W == 128 && H % 16 == 0 && Cin <= 2 &&
      (long long)N * Cin * D * H * W * 4 < (1ll << 32)) {
    // tile-staged form (stem_bww_tile_kernel): a workgroup per 4 output rows of one parity; same slab count as below
    const int tiles_total = N * OD * (OH / 4);
    const int nb = std::min(stem_bw_blocks(), tiles_total);
    const int it = msl::cdiv(tiles_total, nb);
    const size_t tl = (size_t)(32 * SBT_LDA + 36 * Cin * SBT_RP) * sizeof(float);
    constexpr int tile_nw = 4;
#define MSL_STEM_BWT2(CI, B_, NW_)                                                                                   \
  do {                                                                                                               \
    if (tl > 64 * 1024) {                                                                                            \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bww_tile_kernel<CI, B_, NW_>),          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);                      \
      if (e_ != hipSuccess) return (int)e_;                                                                          \
    }                                                                                                                \
    MSL_LAUNCH((stem_bww_tile_kernel<CI, B_, NW_>), dim3(nb), dim3(NW_ * 64), tl, st, (const void*)dy,       \
                       (const void*)yraw, x, workspace, N, D, H, OD, OH, tiles_total, it, bnv, w1);                  \
  } while (0)
#define MSL_STEM_BWT(CI, B_)                          \
  do {                                                \
    if (tile_nw == 8) MSL_STEM_BWT2(CI, B_, 8);       \
    else MSL_STEM_BWT2(CI, B_, 4);                    \
  } while (0)
    if (Cin == 1) { if (bf16act) MSL_STEM_BWT(1, true); else MSL_STEM_BWT(1, false); }
    else { if (bf16act) MSL_STEM_BWT(2, true); else MSL_STEM_BWT(2, false); }
#undef MSL_STEM_BWT
#undef MSL_STEM_BWT2
    MSL_LAUNCH_CHECK();
    if (!dw) return MSL_OK;
    MSL_LAUNCH(stem_bwd_weight_reduce_kernel, dim3(msl::cdiv(32 * 32 * NT, 32)), dim3(256), 0, st, workspace, dw,
                       Cin * 27, NT, nb);
    MSL_LAUNCH_CHECK();
    return MSL_OK;
  }
#define MSL_STEM_BW1(CI, AP)                                                                                         \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_weight_kernel<CI, AP>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e_ != hipSuccess) return (int)e_;                                                                          \
    }                                                                                                                \
    MSL_LAUNCH((stem_bwd_weight_kernel<CI, AP>), dim3(nblocks), dim3(256), lds, st, dy, x, workspace, N, D,  \
                       H, W, OD, OH, OW, sd, sh, sw, chunks_per_row, total_chunks, iters, yraw, bnv, w1, fs);        \
  } while (0)
#define MSL_STEM_BW1B(CI, AP)                                                                                        \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_weight_kernel<CI, AP, true>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e_ != hipSuccess) return (int)e_;                                                                          \
    }                                                                                                                \
    MSL_LAUNCH((stem_bwd_weight_kernel<CI, AP, true>), dim3(nblocks), dim3(256), lds, st, dy, x, workspace, N, \
                       D, H, W, OD, OH, OW, sd, sh, sw, chunks_per_row, total_chunks, iters, yraw, bnv, w1, fs);     \
  } while (0)
#define MSL_STEM_BW(CI)                                \
  do {                                                 \
    if (bf16act) {                                     \
      if (w1) MSL_STEM_BW1B(CI, 2);                    \
      else if (yraw) MSL_STEM_BW1B(CI, 1);             \
      else MSL_STEM_BW1B(CI, 0);                       \
    } else if (w1) MSL_STEM_BW1(CI, 2);                \
    else if (yraw) MSL_STEM_BW1(CI, 1);                \
    else MSL_STEM_BW1(CI, 0);                          \
  } while (0)
  switch (Cin) {
    case 1: MSL_STEM_BW(1); break;
    case 2: MSL_STEM_BW(2); break;
    case 3: MSL_STEM_BW(3); break;
    case 4: MSL_STEM_BW(4); break;
    default: return MSL_ERR_UNSUPPORTED;
  }
#undef MSL_STEM_BW
#undef MSL_STEM_BW1
#undef MSL_STEM_BW1B
  MSL_LAUNCH_CHECK();
  if (!dw) return MSL_OK;  // deferred: the caller folds the slabs with msl_grad_reduce_batch (kind 2)
  const int K = Cin * 27;
  MSL_LAUNCH(stem_bwd_weight_reduce_kernel, dim3(msl::cdiv(32 * 32 * NT, 32)), dim3(256), 0, st, workspace, dw, K,
                     NT, nblocks);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

}  // extern "C"
