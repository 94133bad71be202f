// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the 3D-SSD hot path.
// Wavefront = 64 lanes everywhere.  All reductions here have a FIXED combination order, so every
// kernel that uses them is run-to-run bit-reproducible (no atomics on floating point anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSL_WAVE 64
#define MSL_OK 0
#define MSL_ERR_ARG (-1)
#define MSL_ERR_UNSUPPORTED (-2)

#define MSL_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

// ---- kernel launches -------------------------------------------------------------------------------------------
// Every launch of the library goes through MSL_LAUNCH.  Normally that IS hipLaunchKernelGGL.  If the calling thread has
// armed a stop event (msl_arm_stop_event, optim.hip), the launch that the arming names carries the event as the stopEvent
// of hipExtLaunchKernel: the event completes with that kernel's own completion signal and the stream gets NO separate
// record (barrier) packet.  Measured on MI355X (tools/probes/evrec.hip, chain of dependent kernels, a second stream waiting
// for each): fork by hipEventRecord +4.9 us per kernel on the recording stream, by a hipEventDisableSystemFence event
// +2.4 us, by stop event +1.2 us.
#include <hip/hip_ext.h>
#include <tuple>
#include <utility>
namespace msl {
struct StopEventArm {
  hipEvent_t ev = nullptr;  // armed event (nullptr: none)
  int skip = 0;             // launches to let pass before the one that takes the event
  unsigned launches = 0;    // launches issued by this thread so far (the recorder learns launches-per-entry-point from it)
};
StopEventArm& stop_event_arm();  // thread-local, defined in optim.hip

inline hipEvent_t take_stop_event() {
  StopEventArm& a = stop_event_arm();
  ++a.launches;
  if (a.ev == nullptr) return nullptr;
  if (a.skip > 0) {
    --a.skip;
    return nullptr;
  }
  hipEvent_t e = a.ev;
  a.ev = nullptr;
  return e;
}

template <typename... KArgs, size_t... I>
inline void launch_with_stop_event(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned smem, hipStream_t st,
                                   hipEvent_t stop, std::tuple<KArgs...>& vals, std::index_sequence<I...>) {
  void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {(void*)&std::get<I>(vals)...};
  (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, smem, st, nullptr, stop, 0);
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned smem, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  hipEvent_t stop = take_stop_event();
  if (stop == nullptr) {
    hipLaunchKernelGGL(kernel, grid, block, smem, st, std::forward<Args>(args)...);
  } else {
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};  // the kernel's own parameter types, addressable
    launch_with_stop_event(kernel, grid, block, smem, st, stop, vals, std::index_sequence_for<KArgs...>{});
  }
}
}  // namespace msl
#define MSL_LAUNCH(kernel, ...) msl::launch(kernel, __VA_ARGS__)

namespace msl {

// ---- cross-lane sums on the VALU (DPP) -------------------------------------------------------------------------
// __shfl_* lowers to ds_bpermute_b32 (an LDS-pipe instruction per step); a DPP add is one VALU instruction.  The
// combination order below is fixed, so results stay run-to-run bit-identical.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over each aligned group of 16 lanes (a DPP "row"); every lane of the group gets the result.
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// Sum over each 32-lane half of the wave; valid in lanes 16-31 (lower half) and 48-63 (upper half): the second
// 16-lane row of each half adds the first row's total, broadcast from its lane 15 (row_bcast15 - 5 VALU instructions).
constexpr int HALF32_SUM_LANE = 16;  // (lane & 31) of a lane that holds the result
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
  return v + prev;
}

// Sum over the 64 lanes; every lane gets the result.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// row[col .. col+3] of a row of S floats, zero past the end.  The loads are unconditional on clamped (always valid)
// addresses and masked afterwards: a predicated load compiles to a branch per load, and the compiler then waits for
// each one before it issues the next - the loads of a tile must leave back to back.
// VEC: S % 4 == 0, col % 4 == 0 and the row 16-byte aligned (then col < S implies col + 3 < S).
template <bool VEC>
__device__ __forceinline__ float4 load4_zfill(const float* __restrict__ row, int col, int S) {
  float4 v;
  if (VEC) {
    const bool in = col < S;
    v = *reinterpret_cast<const float4*>(row + (in ? col : 0));
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
  } else {
    const int last = S - 1;
    v.x = row[min(col, last)]; v.y = row[min(col + 1, last)]; v.z = row[min(col + 2, last)]; v.w = row[min(col + 3, last)];
    v.x = col < S ? v.x : 0.f; v.y = col + 1 < S ? v.y : 0.f; v.z = col + 2 < S ? v.z : 0.f; v.w = col + 3 < S ? v.w : 0.f;
  }
  return v;
}

// Scalar-base addressing.  A pointer that IS wave-uniform is made so for the compiler too (two v_readfirstlane) and kept
// in the GLOBAL address space (an integer round trip would otherwise yield a generic pointer and flat_* instructions);
// base + zero-extended 32-bit BYTE offset then selects the saddr mode: no 64-bit vector address arithmetic per access.
typedef __attribute__((address_space(1))) char gbyte;
__device__ __forceinline__ gbyte* uniform_base(const void* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gbyte*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float load_f32(const gbyte* base, unsigned byte_off) {
  return *(const __attribute__((address_space(1))) float*)(base + byte_off);
}
__device__ __forceinline__ void store_f32(gbyte* base, unsigned byte_off, float v) {
  *(__attribute__((address_space(1))) float*)(base + byte_off) = v;
}

// Force a loaded value into its register HERE.  Without it the optimiser sinks a load into the conditional block
// that is its only user, which turns "issue U loads, then use them" back into U serial round trips.
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// Sum over the whole block (blockDim.x multiple of 64, <= 1024).  Result valid in thread 0.
// `scratch` must hold >= blockDim.x/64 doubles; the caller must __syncthreads() before reusing it.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < nw; ++i) t += scratch[i];
  }
  return t;
}

// relu(x * s + t) when affine is requested, else x.  fmaf is what every consumer uses, so all
// kernels that re-create an activation from (raw conv output, scale, shift) agree bit for bit.
// NaN must survive the ReLU (torch's relu propagates it; fmaxf would swallow it and hide a diverged run from
// the NaN guards of ssd3d.py:95-98,:258-261).
__device__ __forceinline__ float act(float x, float s, float t) {
  const float v = fmaf(x, s, t);
  return v < 0.0f ? 0.0f : v;
}

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- bf16 storage (bf16 activation path) ------------------------------------------------------------------------
// Round-to-nearest-even through the compiler's __bf16 (v_cvt_pk_bf16_f32 on gfx950: NaN stays NaN); widening is exact.
__device__ __forceinline__ unsigned short f2bf(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// ---- activation storage: fp32 (the reference's precision) or bf16 (the bf16 activation path) -------------------------------
// Kernels templated on the storage type T walk 4 (or 2, or 1) consecutive elements per lane either way; arithmetic is fp32.
typedef unsigned short su16;
typedef su16 su16x4 __attribute__((ext_vector_type(4)));
typedef su16 su16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const su16* p) { return bf2f(*p); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const su16* p) {
  const su16x4 v = *reinterpret_cast<const su16x4*>(p);
  return make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const su16* p) {
  const su16x2 v = *reinterpret_cast<const su16x2*>(p);
  return make_float2(bf2f(v[0]), bf2f(v[1]));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(su16* p, float4 v) {
  su16x4 o;
  o[0] = f2bf(v.x); o[1] = f2bf(v.y); o[2] = f2bf(v.z); o[3] = f2bf(v.w);
  *reinterpret_cast<su16x4*>(p) = o;
}
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(su16* p, float2 v) {
  su16x2 o;
  o[0] = f2bf(v.x); o[1] = f2bf(v.y);
  *reinterpret_cast<su16x2*>(p) = o;
}
// the value a later pass will read back from storage (bf16: rounded once)
__device__ __forceinline__ float4 as_stored(const float*, float4 v) { return v; }
__device__ __forceinline__ float4 as_stored(const su16*, float4 v) {
  return make_float4(bf2f(f2bf(v.x)), bf2f(f2bf(v.y)), bf2f(f2bf(v.z)), bf2f(f2bf(v.w)));
}

// ---- BatchNorm fold -------------------------------------------------------------------------------------------
// Turning the fp64 (sum, sumsq) partials of a producer into its per-channel affine is cheap, but as a kernel of its
// own it sits on the forward dependency chain (15 launches of ~6 us).  So the CONSUMER kernels fold the partials of
// the channels they read in their prologue, and msl_bn_finalize (same arithmetic, same summation order -> the same
// bits) runs off the chain on a side stream to update the running statistics and to store the vectors for backward.
// Canonical order: NP <= 64 -> one thread adds the partials serially; NP > 64 -> 64 lanes take p = lane, lane+64, ...
// and their sums are combined by the wave_sum tree.
struct BnFold {
  const double* partials;  // [2][C][NP]; nullptr = no fold (use the in_scale / in_shift vectors)
  int NP, C;
  double count;
  const float* gamma;
  const float* beta;
  float eps;
};

__device__ __forceinline__ void bn_affine_from_sums(const BnFold& f, int c, double s, double q, float& scale,
                                                    float& shift, float& mean_o, float& invstd_o, double& var_o) {
  const double mean = s / f.count;
  double var = q / f.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)f.eps);
  const double sc = (double)f.gamma[c] * invstd;
  scale = (float)sc;
  shift = (float)((double)f.beta[c] - mean * sc);
  mean_o = (float)mean;
  invstd_o = (float)invstd;
  var_o = var;
}

// NP <= 64: call from ONE thread.
__device__ __forceinline__ void bn_fold_serial(const BnFold& f, int c, float& scale, float& shift, float& mean_o,
                                               float& invstd_o, double& var_o) {
  const double* ps = f.partials + (size_t)c * f.NP;
  const double* pq = f.partials + ((size_t)f.C + c) * f.NP;
  double s = 0.0, q = 0.0;
  int p = 0;
  for (; p + 8 <= f.NP; p += 8) {  // 16 loads in flight; the additions keep the serial order
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = ps[p + u];
      b[u] = pq[p + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += a[u];
      q += b[u];
    }
  }
  for (; p < f.NP; ++p) {
    s += ps[p];
    q += pq[p];
  }
  bn_affine_from_sums(f, c, s, q, scale, shift, mean_o, invstd_o, var_o);
}

// NP > 64: call from all 64 lanes of one wave (same c); results valid in every lane.
__device__ __forceinline__ void bn_fold_wave(const BnFold& f, int c, float& scale, float& shift, float& mean_o,
                                             float& invstd_o, double& var_o) {
  const int lane = threadIdx.x & 63;
  const double* ps = f.partials + (size_t)c * f.NP;
  const double* pq = f.partials + ((size_t)f.C + c) * f.NP;
  double s = 0.0, q = 0.0;
  int p = lane;
  for (; p + 7 * 64 < f.NP; p += 8 * 64) {  // 16 loads in flight; the additions keep the serial order
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
  for (; p < f.NP; p += 64) {
    s += ps[p];
    q += pq[p];
  }
  s = wave_sum(s);
  q = wave_sum(q);
  bn_affine_from_sums(f, c, s, q, scale, shift, mean_o, invstd_o, var_o);
}

// Fold `nch` channels starting at c0 into LDS arrays (all threads of the workgroup call this; ends with a barrier).
__device__ __forceinline__ void bn_fold_block(const BnFold& f, int c0, int nch, float* lds_scale, float* lds_shift) {
  float a, b, m, i;
  double v;
  if (f.NP <= 64) {
    for (int k = threadIdx.x; k < nch; k += blockDim.x) {
      bn_fold_serial(f, c0 + k, a, b, m, i, v);
      lds_scale[k] = a;
      lds_shift[k] = b;
    }
  } else {
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = wv; k < nch; k += nw) {
      bn_fold_wave(f, c0 + k, a, b, m, i, v);
      if ((threadIdx.x & 63) == 0) {
        lds_scale[k] = a;
        lds_shift[k] = b;
      }
    }
  }
  __syncthreads();
}


// out[i] = sum_k slabs[k * stride + i], k = 0..nslabs-1.  32 outputs x 8 slab groups per workgroup; every thread
// walks its slabs with independent loads (deep memory-level parallelism), then the 8 group sums are folded
// in a fixed order -> bit-reproducible.  Launch with cdiv(count, 32) workgroups of 256 threads.
__device__ __forceinline__ float reduce_slabs_256(const float* __restrict__ slabs, size_t stride, int count,
                                                  int nslabs, float* lds /* [8][33] */) {
  const int li = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  float s = 0.f;
  if (i < count) {
    const float* p = slabs + i;
    int k = g;
#pragma unroll 1
    for (; k + 56 < nslabs; k += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 8 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nslabs; k += 8) s += p[(size_t)k * stride];
  }
  lds[g * 33 + li] = s;
  __syncthreads();
  float t = 0.f;
  if (g == 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t += lds[u * 33 + li];
  }
  return t;  // valid for g == 0 && i < count
}

}  // namespace msl
