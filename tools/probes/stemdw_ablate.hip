// Where does stem_dw_eval_kernel spend its time?  The kernel of csrc/stemdw.hip with one part compiled out
// (-DSDW_ABL_NO_DW / -DSDW_ABL_NO_MFMA / -DSDW_ABL_NO_FETCH), timed alone at 192^3 x 2.
// Build: for v in "" -DSDW_ABL_NO_DW -DSDW_ABL_NO_MFMA -DSDW_ABL_NO_FETCH; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I../../mslesions3d_amd/csrc -o stemdw_ablate$v stemdw_ablate.hip; done
#include "stemdw.hip"
#include <cstdio>
#include <vector>
msl::StopEventArm& msl::stop_event_arm() { static thread_local msl::StopEventArm a; return a; }
int main() {
  const int N = 2, D = 192;
  float *x, *w, *wd, *sc, *sh, *z;
  hipMalloc(&x, (size_t)N * D * D * D * 4); hipMalloc(&w, 32 * 27 * 4); hipMalloc(&wd, 32 * 27 * 4); hipMalloc(&sc, 128); hipMalloc(&sh, 128);
  hipMalloc(&z, (size_t)N * 32 * (D / 4) * (D / 4) * (D / 4) * 4);
  hipMemset(x, 0, (size_t)N * D * D * D * 4); hipMemset(w, 0, 32 * 27 * 4); hipMemset(wd, 0, 32 * 27 * 4); hipMemset(sc, 0, 128); hipMemset(sh, 0, 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) msl_stem_dw_fwd_eval(x, w, sc, sh, wd, z, N, 1, D, D, D, nullptr);
  hipEventRecord(a, nullptr);
  for (int i = 0; i < 20; ++i) msl_stem_dw_fwd_eval(x, w, sc, sh, wd, z, N, 1, D, D, D, nullptr);
  hipEventRecord(b, nullptr); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%.1f us per launch\n", ms * 1000 / 20);
  return 0;
}
