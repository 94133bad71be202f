// Fused Adam over the flat parameter arena + NaN flag.
// Reference: LSSD3D.configure_optimizers (lesions3d/ssd3d.py:704-722): torch.optim.Adam(weight_decay=5e-4)
// (L2 added to the gradient, not AdamW), bias parameters at 2*lr; NaN guards of ssd3d.py:258-261,:479.
//
// All trainable parameters live in ONE contiguous fp32 buffer ordered by backward completion (heads first,
// stem last) with a byte mask marking the 2*lr (bias) elements, and matching flat grad / exp_avg /
// exp_avg_sq buffers, so the optimiser is a single HBM-streaming launch
// (7 x 3.8 MB) and the data-parallel all-reduce works on contiguous buckets.  Hyper-parameters are read
// from a small device buffer so the launch can sit inside a captured HIP graph.
#include "common.hpp"

namespace {

// hp: [0] step_size for biases (= 2*lr / bias_correction1), [1] step_size others, [2] sqrt(bias_correction2),
//     [3] beta1, [4] beta2, [5] eps, [6] weight_decay, [7] gradient scale (1/world_size for DP mean)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ hp, const unsigned char* __restrict__ is_bias, int n) {
  const float ss_b = hp[0], ss_o = hp[1], bc2s = hp[2], b1 = hp[3], b2 = hp[4], eps = hp[5], wd = hp[6], gs = hp[7];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float pi = p[i];
    float gi = g[i] * gs;
    gi = gi + wd * pi;                               // weight_decay: grad.add(param, alpha=wd)
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);  // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;  // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - (is_bias[i] ? ss_b : ss_o) * (mi / denom);  // param.addcdiv_(exp_avg, denom, -step_size)
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- batched gradient reduction ---------------------------------------------------------------------------------
// Every weight-gradient kernel of the step that splits its position range over workgroups leaves PARTIAL results
// (fp32 slabs or fp64 per-element partials) in its workspace; ONE launch of this kernel folds all of them into the flat
// gradient arena in a fixed order (slab 0, 1, 2, ... per output element: run-to-run bit-identical, no float atomics).
// It replaces ~20 tiny per-layer reduce launches of the step (slab_reduce / dw finalize / head reduce / stem reduce).
// A workgroup produces 32 output elements of one table entry; it finds its entry by scanning the (<= 64) first_block
// fields, which are wave-uniform scalar loads.
struct GradReduceEntry {   // 64 bytes, filled on the host by msl_grad_reduce_table_set
  const void* src;
  float* dst;
  float* dst2;
  long long stride;        // slab-major kinds: elements between two slabs
  int kind, nslabs, count, first_block;
  int p0, p1, p2, pad;
};
enum { GR_SLAB_F32 = 0, GR_ELEM_F64 = 1, GR_STEM_F32 = 2, GR_HEAD_F32 = 3, GR_LOSS = 4 };

// Output address of element i of a slab-major entry (nullptr: padding, nothing to store).
__device__ __forceinline__ float* grad_reduce_dst(const GradReduceEntry& en, int i) {
  if (en.kind == GR_SLAB_F32) return en.dst + i;
  if (en.kind == GR_STEM_F32) {  // padded [32][32*NT] image -> dw[co][K]   (p0 = K, p1 = NT)
    const int co = i / (32 * en.p1), k = i % (32 * en.p1);
    return k < en.p0 ? en.dst + (size_t)co * en.p0 + k : nullptr;
  }
  // GR_HEAD_F32: i walks a slab in STORAGE order [ct][tap][mt][co16][ci16] (coalesced slab reads; the scattered accesses
  // are the `count` writes, not the nslabs * count reads)   (p0 = C, p1 = MT, p2 = co_total)
  const int C = en.p0, MT = en.p1;
  const int cil = i & 15, col = (i >> 4) & 15, r = i >> 8;
  const int mt = r % MT, tap = (r / MT) % 27, ct = r / (MT * 27);
  const int co = mt * 16 + col, ci = ct * 16 + cil;
  return co >= en.p2 ? nullptr
         : co < 12   ? en.dst + ((size_t)co * C + ci) * 27 + tap
                     : en.dst2 + ((size_t)(co - 12) * C + ci) * 27 + tap;
}

// Few slabs (<= GR_FEW): a thread owns 4 consecutive outputs (16-byte slab reads, all slabs' loads in flight at once),
// 1024 outputs per workgroup - the workgroup count, not the bytes, is what this launch costs.
constexpr int GR_FEW = 16;
__host__ __device__ __forceinline__ bool grad_reduce_few(int kind, int nslabs, int count, long long stride) {
  return kind != GR_ELEM_F64 && nslabs <= GR_FEW && count % 4 == 0 && stride % 4 == 0;
}

__global__ __launch_bounds__(256) void grad_reduce_batch_kernel(const GradReduceEntry* __restrict__ table, int n_entries,
                                                                const int* __restrict__ block_entry) {
  __shared__ float lds[8 * 33];
  __shared__ double ldsd[8 * 33];
  int e = 0;
  if (block_entry) {
    e = block_entry[blockIdx.x];  // one load instead of a walk over up to 64 first_block fields (dependent scalar loads)
  } else {
    for (int k = 1; k < n_entries; ++k)
      if ((int)blockIdx.x >= table[k].first_block) e = k;
  }
  const GradReduceEntry en = table[e];
  const int blk = blockIdx.x - en.first_block;
  const int li = threadIdx.x & 31, g = threadIdx.x >> 5;
  if (en.kind == GR_LOSS) {
    // loss values of msl_multibox_loss_pack: src = fp64 [nslabs][2] (sum ce, sum l1) per workgroup, dst2 = the positives counter
    // (int) -> dst = [conf, loc, n_positives] (ssd3d.py:896, :933); one wave, lanes strided over the partials, fixed order
    if (threadIdx.x < 64) {
      const double* src = (const double*)en.src;
      double ce = 0.0, l1 = 0.0;
      for (int k = threadIdx.x; k < en.nslabs; k += 64) {
        ce += src[2 * k];
        l1 += src[2 * k + 1];
      }
      ce = msl::wave_sum(ce);
      l1 = msl::wave_sum(l1);
      if (threadIdx.x == 0) {
        const float f = (float)*reinterpret_cast<const int*>(en.dst2);
        en.dst[0] = (float)ce / f;
        en.dst[1] = (float)l1 / (f * 6.0f);
        en.dst[2] = f;
      }
    }
    return;
  }
  if (en.kind == GR_ELEM_F64) {
    // partials of one element are contiguous: 32 lanes walk them (coalesced), 8 elements per pass, 4 passes
    const double* src = (const double*)en.src;
    for (int pass = 0; pass < 4; ++pass) {
      const int i = blk * 32 + pass * 8 + g;
      double s = 0.0;
      if (i < en.count) {
        const double* ps = src + (size_t)i * en.nslabs;
        int p = li;
        for (; p + 7 * 32 < en.nslabs; p += 8 * 32) {  // 8 independent loads in flight; additions in index order
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = ps[p + u * 32];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < en.nslabs; p += 32) s += ps[p];
      }
      ldsd[g * 33 + li] = s;
      __syncthreads();
      if (li == 0 && i < en.count) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) t += ldsd[g * 33 + u];
        en.dst[i] = (float)t;
      }
      __syncthreads();
    }
    return;
  }
  if (en.pad == 0 && grad_reduce_few(en.kind, en.nslabs, en.count, en.stride)) {
    const int i = blk * 1024 + threadIdx.x * 4;
    if (i >= en.count) return;
    const float* p = (const float*)en.src + i;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= en.nslabs; k += 4) {  // 4 independent 16-byte loads in flight; additions in slab order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u) * en.stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      }
    }
    for (; k < en.nslabs; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * en.stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (en.kind == GR_SLAB_F32) {
      *reinterpret_cast<float4*>(en.dst + i) = s;
    } else {
      const float vals[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float* o = grad_reduce_dst(en, i + u);
        if (o) *o = vals[u];
      }
    }
    return;
  }
  // many slabs: 32 outputs x 8 slab groups per workgroup; output i of this entry sits at src[k * stride + i] in slab k
  const int i = blk * 32 + li;
  float* out = i < en.count ? grad_reduce_dst(en, i) : nullptr;
  float s = 0.f;
  if (i < en.count) {
    const float* p = (const float*)en.src + i;
    int k = g;
#pragma unroll 1
    for (; k + 56 < en.nslabs; k += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 8 * u) * en.stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < en.nslabs; k += 8) s += p[(size_t)k * en.stride];
  }
  lds[g * 33 + li] = s;
  __syncthreads();
  if (g == 0 && out) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += lds[u * 33 + li];
    *out = t;
  }
}

__global__ __launch_bounds__(256) void nan_flag_kernel(const float* __restrict__ x, size_t n, int* __restrict__ flag,
                                                       int bit) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= isnan(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

// two tensors, one launch (the forward pass checks its two outputs: ssd3d.py:258-261)
__global__ __launch_bounds__(256) void nan_flag2_kernel(const float* __restrict__ a, size_t na, int bit_a, const float* __restrict__ b,
                                                        size_t nb, int bit_b, int* __restrict__ flag) {
  bool bad_a = false, bad_b = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na; i += (size_t)gridDim.x * 256) bad_a |= isnan(a[i]);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (size_t)gridDim.x * 256) bad_b |= isnan(b[i]);
  const int bits = (__any(bad_a) ? bit_a : 0) | (__any(bad_b) ? bit_b : 0);
  if (bits && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
}

}  // namespace

msl::StopEventArm& msl::stop_event_arm() {
  static thread_local msl::StopEventArm arm;
  return arm;
}

extern "C" {

int msl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* hp,
                  const unsigned char* is_bias, int n, void* stream) {
  if (n <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(adam_kernel, dim3(min(msl::cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, params,
                     grads, exp_avg, exp_avg_sq, hp, is_bias, n);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

size_t msl_grad_reduce_entry_bytes(void) { return sizeof(GradReduceEntry); }

// Fill entry `index` of a HOST table (uploaded by the caller) and return the number of workgroups it needs (< 0: bad
// argument); `first_block` = sum of the counts returned for the entries before it.
//   kind 0  fp32 slabs  [nslabs][stride >= count]           -> dst[i]                       (pointwise weight gradients)
//   kind 1  fp64 partials [count][nslabs]                    -> dst[i]                       (depthwise weight gradients)
//   kind 2  fp32 slabs of the padded stem image [32][32*p1]  -> dst[co*p0 + k], k < p0       (stem; count = 1024 * p1)
//   kind 3  fp32 head slabs (msl_head_conv_bwd_weight)       -> dst = dloc_w, dst2 = dcl_w   (p0 = C, p1 = MT, p2 = 12+2*ncls;
//                                                                                            count = stride = (C/16)*27*MT*256)
//   kind 4  loss partials of msl_multibox_loss_pack: src fp64 [nslabs][2], dst2 = its positives counter (int*), count = 1
//                                                            -> dst = loss_out [conf, loc, n_positives]
int msl_grad_reduce_table_set(void* host_table, int index, int first_block, int kind, const void* src, float* dst,
                              float* dst2, int nslabs, int count, long long stride, int p0, int p1, int p2) {
  if (!host_table || index < 0 || kind < 0 || kind > 4 || !src || !dst || nslabs <= 0 || count <= 0) return MSL_ERR_ARG;
  GradReduceEntry e{src, dst, dst2, stride, kind, nslabs, count, first_block, p0, p1, p2, 0};
  ((GradReduceEntry*)host_table)[index] = e;
  if (kind == GR_LOSS) return dst2 ? 1 : MSL_ERR_ARG;
  if (grad_reduce_few(kind, nslabs, count, stride) && ((uintptr_t)src % 16 == 0) && (kind != GR_SLAB_F32 || (uintptr_t)dst % 16 == 0))
    return msl::cdiv(count, 1024);
  ((GradReduceEntry*)host_table)[index].pad = 1;  // force the many-slab form (unaligned pointers)
  return msl::cdiv(count, 32);
}

int msl_grad_reduce_batch(const void* table, int n_entries, int total_blocks, void* stream) {
  if (!table || n_entries <= 0 || n_entries > 64 || total_blocks <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(grad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const GradReduceEntry*)table, n_entries, (const int*)nullptr);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// the same with a device array block_entry[total_blocks] = table entry of every workgroup (the caller knows it from the counts
// msl_grad_reduce_table_set returned): the workgroups then do not search the table
int msl_grad_reduce_batch_indexed(const void* table, int n_entries, const int* block_entry, int total_blocks, void* stream) {
  if (!table || !block_entry || n_entries <= 0 || n_entries > 64 || total_blocks <= 0) return MSL_ERR_ARG;
  MSL_LAUNCH(grad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const GradReduceEntry*)table, n_entries, block_entry);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// *flag |= bit if any element of x is NaN (flag must be zeroed by the caller once per step)
int msl_nan_flag(const float* x, size_t n, int* flag, int bit, void* stream) {
  if (n == 0) return MSL_OK;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  MSL_LAUNCH(nan_flag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, flag, bit);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

int msl_nan_flag2(const float* a, size_t na, int bit_a, const float* b, size_t nb, int bit_b, int* flag, void* stream) {
  const size_t n = na > nb ? na : nb;
  if (n == 0) return MSL_OK;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  MSL_LAUNCH(nan_flag2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, na, bit_a, b, nb, bit_b, flag);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// 32-bit fill on the stream (replaces torch's .zero_() inside recorded launch programs)
int msl_fill_u32(void* dst, unsigned int value, size_t count, void* stream) {
  if (count == 0) return MSL_OK;
  hipError_t e = hipMemsetD32Async((hipDeviceptr_t)dst, (int)value, count, (hipStream_t)stream);
  return e == hipSuccess ? MSL_OK : (int)e;
}

// ---- stream fork / join primitives (so that a multi-stream schedule can live inside a recorded launch program)
int msl_event_create(void** out) {
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) return (int)e;
  *out = (void*)ev;
  return MSL_OK;
}
// An event that orders streams of ONE device only: no system-scope release / acquire at the record (hip_runtime_api.h,
// hipEventDisableSystemFence: "device memory may not be visible to the host and other devices").  Kernel packets keep
// their own agent-scope fences, which is what a consumer on another stream of the same GPU needs.  Never hand such an
// event to the host (hipEventSynchronize), to a copy engine moving data to the host, or to another device's stream.
int msl_event_create_device(void** out) {
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
  if (e != hipSuccess) return (int)e;
  *out = (void*)ev;
  return MSL_OK;
}
// (hipEventReleaseToDevice was tried for the timing pairs: the 4.6 us a pair adds to a launch did not change)
int msl_event_create_timed(void** out) {
  hipEvent_t ev;
  hipError_t e = hipEventCreate(&ev);
  if (e != hipSuccess) return (int)e;
  *out = (void*)ev;
  return MSL_OK;
}
// elapsed milliseconds between two completed timing events
int msl_event_elapsed_ms(void* start, void* stop, float* out_ms) {
  return (int)hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop);
}
int msl_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
int msl_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
int msl_stream_wait_event(void* stream, void* ev) {
  return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
}

// ---- stop events: a fork without a record packet (common.hpp, MSL_LAUNCH) ---------------------------------------------------
// msl_arm_stop_event(ev, skip): the (skip + 1)-th kernel launch this THREAD issues from now on completes `ev` (as if
// msl_event_record(ev, that launch's stream) followed it).  The caller must know how many kernels the entry points in
// between launch: msl_thread_launch_count() before and after a call tells (the launch-program recorder does that).
int msl_arm_stop_event(void* ev, int skip) {
  if (ev == nullptr || skip < 0) return MSL_ERR_ARG;
  msl::StopEventArm& a = msl::stop_event_arm();
  if (a.ev != nullptr) return MSL_ERR_ARG;  // the previous arming was never consumed: a miscounted program
  a.ev = (hipEvent_t)ev;
  a.skip = skip;
  return MSL_OK;
}
// 1 if an armed stop event is still waiting for its launch (after the entry point it was meant for: a bug)
int msl_stop_event_pending(void) { return msl::stop_event_arm().ev != nullptr ? 1 : 0; }
int msl_thread_launch_count(void) { return (int)(msl::stop_event_arm().launches & 0x7FFFFFFFu); }

int msl_abi_version(void) { return 1; }

}  // extern "C"
