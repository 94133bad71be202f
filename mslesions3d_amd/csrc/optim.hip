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

__global__ __launch_bounds__(256) void nan_flag_kernel(const float* __restrict__ x, size_t n, int* __restrict__ flag,
                                                       int bit) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= isnan(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

}  // namespace

extern "C" {

int msl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* hp,
                  const unsigned char* is_bias, int n, void* stream) {
  if (n <= 0) return MSL_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(min(msl::cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, params,
                     grads, exp_avg, exp_avg_sq, hp, is_bias, n);
  MSL_LAUNCH_CHECK();
  return MSL_OK;
}

// *flag |= bit if any element of x is NaN (flag must be zeroed by the caller once per step)
int msl_nan_flag(const float* x, size_t n, int* flag, int bit, void* stream) {
  if (n == 0) return MSL_OK;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(nan_flag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, flag, bit);
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

// ---- HIP graph capture of a launch sequence (single-GPU training step) ------------------------------------------
int msl_graph_begin(void* stream) { return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed); }
int msl_graph_end(void* stream, void** exec_out) {
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return (int)e;
  *exec_out = (void*)exec;
  return MSL_OK;
}
int msl_graph_launch(void* exec, void* stream) { return (int)hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream); }
int msl_graph_destroy(void* exec) { return (int)hipGraphExecDestroy((hipGraphExec_t)exec); }

int msl_abi_version(void) { return 1; }

}  // extern "C"
