// exp() exactly as torch's CPU softmax computes it.  Included by detect.hip (MSL_FN = __device__ __forceinline__) and, as
// plain C++, by tests/test_host_cpu.py (MSL_FN = static inline), which checks it bit for bit against torch.softmax.
//
// ssd3d.py:363 `F.softmax(predicted_scores, dim=2)` runs ATen's vec_softmax_lastdim kernel: e_c = Vectorized<float>::exp(x_c - max)
// (= Sleef's expf, 1.0-ulp variant), the sum in class order, p_c = e_c * (1 / sum) - a multiplication by the reciprocal, not a
// division.  Sleef's expf: Cody-Waite reduction by ln 2 in two fused steps, degree-5 Horner polynomial in fused multiply-adds,
// 1 + (s*s*u + s), scaling by two exact powers of two.  The candidate order of the NMS is decided by the last bit of these
// probabilities ("bit-exact NMS keep-lists", BASELINE.json north_star), hence a restatement instead of the device libm.
// Compile without floating-point contraction (the Makefile does for detect.hip): every fused operation is spelled out.
#pragma once

MSL_FN float msl_pow2i(int q) { return __builtin_bit_cast(float, (q + 0x7f) << 23); }

MSL_FN float msl_softmax_exp(float d) {
  const int q = (int)__builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
  const float qf = (float)q;
  float s = __builtin_fmaf(qf, -0.693145751953125f, d);
  s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
  float u = 0.000198527617612853646278381f;
  u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
  u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
  u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
  u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
  u = __builtin_fmaf(u, s, 0.5f);
  u = 1.0f + __builtin_fmaf(s * s, u, s);
  u = u * msl_pow2i(q >> 1) * msl_pow2i(q - (q >> 1));
  if (d < -104.0f) u = 0.0f;
  if (d > 100.0f) u = __builtin_inff();
  return u;
}

// probabilities of classes 1..ncls-1 of one prior (x: ncls logits) -> out[c-1]
MSL_FN void msl_softmax_foreground(const float* x, int ncls, float* out, long out_stride) {
  float m = x[0];
  for (int c = 1; c < ncls; ++c) m = __builtin_fmaxf(m, x[c]);
  float se = 0.f;
  for (int c = 0; c < ncls; ++c) se += msl_softmax_exp(x[c] - m);
  const float rcp = 1.0f / se;
  for (int c = 1; c < ncls; ++c) out[(c - 1) * out_stride] = msl_softmax_exp(x[c] - m) * rcp;
}
