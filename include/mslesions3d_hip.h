/* mslesions3d_hip.h — C ABI of libmsl3d_hip.so: the MI355X (gfx950) kernels behind the 3D-SSD hot path.
 *
 * The reference (Medical-Image-Analysis-Laboratory/MSLesions3D) has no FFI: its hot path is Python calling
 * stock torch ops.  The entry points below are what a maintainer of the reference would bind
 * (ctypes) to replace those op sequences; each one names the reference lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the comment says "host"; tensors are dense fp32, NCDHW,
 *    W fastest; `stream` is a hipStream_t passed as void* (NULL = default stream); nothing here synchronises.
 *  - return value: 0 on success, >0 a hipError_t from the launch, -1 bad argument, -2 unsupported shape.
 *  - "raw" = convolution output BEFORE BatchNorm; an (in_scale, in_shift) pair is the folded BatchNorm of the
 *    producer (scale = gamma/sqrt(var+eps), shift = beta - mean*scale); consumers apply relu(x*scale+shift)
 *    while loading.  in_scale == NULL means "input is already an activation / a plain tensor".
 *  - stat partials: fp64 [2][C][NP] (sum, then sum of squares) written by a conv kernel, NP from the matching
 *    *_num_partials(); msl_bn_finalize folds them in a fixed order (bit-reproducible, no float atomics).
 */
#ifndef MSLESIONS3D_HIP_H
#define MSLESIONS3D_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int msl_abi_version(void);

/* ---- BatchNorm3d (train-mode batch statistics) + ReLU : mobilenet.py:29-30, :39, :41, :44-45 ------------- */
int msl_bn_finalize(const double* partials, int num_partials, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                    float eps, float* scale, float* shift, float* save_mean, float* save_invstd, int C,
                    void* stream);
/* every BatchNorm of the network in one launch: fill a host table entry per layer, copy it to the device once */
size_t msl_bn_finalize_entry_bytes(void);
int msl_bn_finalize_table_set(void* host_table, int index, int first_block, const double* partials, int num_partials,
                              double count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, long long* num_batches_tracked, float momentum, float eps,
                              float* scale, float* shift, float* save_mean, float* save_invstd, int C);
int msl_bn_finalize_batch(const void* device_table, int n_entries, int total_channels, void* stream);
/* eval mode: scale/shift from the running statistics */
/* eval-mode (scale, shift) of every BatchNorm in a table built with msl_bn_finalize_table_set, one launch */
int msl_bn_eval_affine_batch(const void* device_table, int n_entries, int total_channels, void* stream);
int msl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, int C, void* stream);
/* relu(y*scale+shift) into a plain (N,C,D,H,W) tensor and/or a zero-haloed (N,C,D+2,H+2,W+2) one (either may be NULL) */
int msl_bn_relu_materialize(const float* y, const float* scale, const float* shift, float* out_plain,
                            float* out_pad, int N, int C, int D, int H, int W, void* stream);
int msl_bn_relu_materialize_fold(const float* y, const double* partials, int num_partials, double count,
                                 const float* gamma, const float* beta, float eps, float* out_plain, float* out_pad,
                                 int N, int C, int D, int H, int W, void* stream);
/* backward of a = relu(bn(y)):  reduce -> finalize -> apply (dy may alias g) */
int msl_bn_relu_bwd_num_partials(int N, int S);
int msl_bn_relu_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift,
                           const float* mean, const float* invstd, double* partials, int N, int C, int S,
                           void* stream);
int msl_bn_bwd_finalize(const double* partials, int num_partials, double count, float* dgamma, float* dbeta,
                        float* c1, float* c2, int C, void* stream);
/* same on an (8, C) vector block [scale, shift, mean, invstd | c1, c2, cC, cE]: writes rows 4-7, where
 * dL/dy = scale * gm + (cC * y + cE) is the folded form used by msl_stem_conv_bwd_weight_fused */
int msl_bn_bwd_finalize_coef(const double* partials, int num_partials, double count, float* dgamma, float* dbeta,
                             float* bn_vec, int C, void* stream);
int msl_bn_relu_bwd_apply(const float* g, const float* y, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* c1, const float* c2, float* dy,
                          int N, int C, int S, void* stream);
/* msl_bn_bwd_finalize + msl_bn_relu_bwd_apply in one launch (for <= 256 partials per channel): bn_vec = the (>= 6, C)
 * block [scale, shift, mean, invstd, c1, c2, ...], rows 4-5 are written */
int msl_bn_relu_bwd_finalize_apply(const double* partials, int num_partials, double count, const float* g, const float* y,
                                   float* bn_vec, float* dgamma, float* dbeta, float* dy, int N, int C, int S,
                                   void* stream);

/* the three steps above in one launch (one workgroup per channel); for N*S <= 65536 elements per channel */
int msl_bn_relu_bwd_fused(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, float* dgamma, float* dbeta, float* dy, int N, int C, int S,
                          void* stream);

/* Per-channel backward link of a Block (autograd of mobilenet.py:43-47 between two pointwise GEMMs), one launch instead of
 * msl_bn_relu_bwd_fused (bn1) + msl_dwconv_bwd_data + msl_bn_relu_bwd_fused (previous block's bn2):
 *   g_z (N,C,OD,OH,OW): in dL/d relu(bn1(z)), out dL/dz (in place; the depthwise weight gradient reads it)
 *   g_y (N,C,D,H,W):    out dL/dy_prev; with accumulate != 0 its content (the heads' share, ssd3d.py:143-169 backward) is
 *                       added to the depthwise bwd-data result before the BatchNorm backward
 *   vec_z / vec_y: (>= 4, C) BatchNorm vector blocks [scale, shift, mean, invstd] of bn1 (on z) / of the previous bn2 (on y_prev)
 *   dw_dw (C,27) or NULL: if given, the depthwise weight gradient dL/dw_dw (mobilenet.py:38 backward) is produced as well
 *                       (same operand pairs as the transposed convolution) - msl_dwconv_bwd_weight is then not needed
 *   (D,H,W) = input extents of the depthwise layer (powers of two, W >= 4, OW >= 4), stride 1 or 2 (k3, p1).  A channel's
 *   whole population lives in one workgroup: supported while N*D*H*W <= 16384 per channel; else MSL_ERR_UNSUPPORTED (-2).
 *   _supported returns the number of waves per channel the launch would use (1, 4, 8 or 16), 0 if unsupported: with
 *   many waves per channel the kernel is bound by instruction issue and dw_dw is better left to its own launch. */
int msl_block_bwd_channel_link_supported(int N, int D, int H, int W, int stride);
int msl_block_bwd_channel_link(float* g_z, const float* z, const float* vec_z, const float* w_dw, const float* y_prev,
                               const float* vec_y, float* g_y, float* dgamma_z, float* dbeta_z, float* dgamma_y,
                               float* dbeta_y, float* dw_dw, int N, int C, int D, int H, int W, int stride, int accumulate,
                               void* stream);

/* Pointwise backward of a big early block in one pass (autograd of mobilenet.py:45-47 for Block.conv2 / bn2 / bn1):
 *   g_y (N,Cout,S) = dL/d relu(bn2(y)) (raw), y (N,Cout,S) raw conv2 output, bn_y_vec (>= 4, Cout) [scale, shift, mean, invstd],
 *   y_partials fp64 [2][Cout][y_np] = the BatchNorm2-backward sums the producer of g_y emitted (msl_dwconv_bwd_data_bnreduce),
 *   y_count = N*S  ->  dgamma_y / dbeta_y, and dL/dy applied on the fly (never stored);
 *   g_z (N,Cin,S) = W^T . dL/dy (= dL/d relu(bn1(z)), raw), z_partials fp64 [2][Cin][NP] = BatchNorm1-backward sums of z
 *   (for msl_bn_relu_bwd_finalize_apply), dw_slabs [NP][Cout][Cin] = partial sums of dL/dW (msl_grad_reduce kind 0);
 *   NP = msl_pwconv_bwd_fused_num_partials (0: shape not taken - Cin = 32, Cout = 64, S % 128 == 0, >= 512 strips of 128
 *   positions - use msl_bn_relu_bwd_* + msl_pwconv_bwd_data + msl_pwconv_bwd_weight_slabs; the entry returns -2). */
int msl_pwconv_bwd_fused_num_partials(int N, int Cin, int Cout, int S);
int msl_pwconv_bwd_fused(const float* g_y, const float* y, const float* bn_y_vec, const double* y_partials, int y_np,
                         double y_count, float* dgamma_y, float* dbeta_y, const float* w, const float* z,
                         const float* bn_z_vec, float* g_z, double* z_partials, float* dw_slabs, int N, int Cin, int Cout, int S,
                         void* stream);

/* ---- stem: Conv3d(Cin->32,k3,stride (sd,sh,sw),p1,no bias) : mobilenet.py:26-31 via ssd3d.py:60-61 ------- */
int msl_stem_conv_fwd_num_partials(int N, int OD, int OH, int OW);
int msl_stem_conv_fwd(const float* x, const float* w, float* y, double* partials, int N, int Cin, int D, int H,
                      int W, int sd, int sh, int sw, void* stream);
size_t msl_stem_conv_bwd_weight_workspace_bytes(int Cin);
/* slabs ([32][32*ceil(Cin*27/32)] floats each) the three stem weight-gradient forms leave in `workspace`; with dw == NULL they
 * skip their own reduction (deferred: msl_grad_reduce_batch kind 2) */
int msl_stem_conv_bwd_weight_nslabs(int N, int D, int H, int W, int sd, int sh, int sw);
int msl_stem_conv_bwd_weight(const float* dy, const float* x, float* dw, float* workspace, int N, int Cin, int D,
                             int H, int W, int sd, int sh, int sw, void* stream);

/* same, with the BatchNorm+ReLU backward of the stem applied while loading: g = dL/d relu(bn(y)), bn_vec (6,32) */
int msl_stem_conv_bwd_weight_bnapply(const float* g, const float* yraw, const float* bn_vec, const float* x, float* dw,
                                     float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                     void* stream);

/* same again for a stem that feeds a stride-2 depthwise layer with taps w1_t (27,32), tap-major: dz is THAT layer's dL/dz
 * (N,32,ceil(OD/2),ceil(OH/2),ceil(OW/2)) and the gradient of the stem activation (autograd's grad of ssd3d.py:81's
 * first feature) is rebuilt from it inside the kernel, never stored.  Pair: msl_dwconv_s2_bwd_bnreduce_bww. */
int msl_stem_conv_bwd_weight_fused(const float* dz, const float* w1_t, const float* yraw, const float* bn_vec,
                                   const float* x, float* dw, float* workspace, int N, int Cin, int D, int H, int W,
                                   int sd, int sh, int sw, void* stream);

/* ---- depthwise Conv3d(C,C,k3,stride s,p1,groups=C) : Block.conv1, mobilenet.py:38,44 ---------------------- */
int msl_dwconv_fwd_num_partials(int N, int C, int D, int H, int W, int stride);
int msl_dwconv_fwd_variant(int N, int C, int D, int H, int W, int stride); /* 0 naive, 1 stream, 2 resident, 3 wave */
int msl_dwconv_fwd(const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                   double* partials, int N, int C, int D, int H, int W, int stride, int force_naive, void* stream);
/* stride-1 bwd-data as a forward pass with reversed taps on the LDS-resident kernel (-2 if the shape is not on that path) */
int msl_dwconv_s1_bwd_data_resident(const float* dy, const float* w, float* g_in, int N, int C, int D, int H, int W,
                                    int accumulate, void* stream);
/* forward with the input's BatchNorm folded in-kernel from the producer's partials (no finalize launch on the chain) */
int msl_dwconv_fwd_fold(const float* x, const double* in_partials, int in_np, double in_count, const float* gamma,
                        const float* beta, float eps, const float* w, float* y, double* partials, int N, int C, int D,
                        int H, int W, int stride, void* stream);
int msl_dwconv_bwd_data(const float* dy, const float* w, float* g_in, int N, int C, int D, int H, int W,
                        int stride, int accumulate, void* stream);
/* bwd-weight on the LDS-tiled forward machinery (-2 / -1 when the shape is on the generic path) */
int msl_dwconv_bwd_weight_tiled(const float* dy, const float* x, const float* in_scale, const float* in_shift,
                                double* partials, int N, int C, int D, int H, int W, int stride, void* stream);
int msl_dwconv_bwd_weight_tiled_num_partials(int N, int C, int D, int H, int W, int stride);
/* stride-2 bwd-data that also emits the BatchNorm-backward partials of the layer it feeds (fp64 [2][C][NP]) */
int msl_dwconv_bwd_data_bnreduce_num_partials(int N, int C, int D, int H, int W);
int msl_dwconv_bwd_data_bnreduce(const float* dy, const float* w, float* g_in, const float* y_prev,
                                 const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                 const float* bn_invstd, double* partials, int N, int C, int D, int H, int W, int stride,
                                 int accumulate, void* stream);
/* stride-2 backward in ONE pass over (dy = dL/dz, y_prev) that does NOT write the input gradient: emits the
 * BatchNorm-backward partials of the producer layer (fp64 [2][C][NP]) and this layer's weight-gradient partials
 * (fp64 [C*27][NP]); msl_dwconv_bwd_weight_finalize sums the latter into dw (C,27).  NP from ..._num_partials (-1 =
 * shape unsupported).  w_taps_t (may be NULL) receives the (27,C) transpose of w. */
int msl_dwconv_s2_bwd_bnreduce_bww_num_partials(int N, int C, int D, int H, int W);
int msl_dwconv_s2_bwd_bnreduce_bww(const float* dy, const float* w, const float* y_prev, const float* bn_scale,
                                   const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                                   double* bn_partials, double* w_partials, float* w_taps_t, int N, int C, int D, int H,
                                   int W, void* stream);
/* the same pass when the input gradient is a tensor of its own (every stride-2 block but the one behind a fused stem): also
 * writes g_in (N,C,D,H,W) = dL/d relu(bn(y_prev)) (accumulate != 0: on top of the heads' share already there) - one launch
 * instead of msl_dwconv_bwd_data_bnreduce + msl_dwconv_bwd_weight; partial counts as msl_dwconv_s2_bwd_bnreduce_bww_num_partials */
int msl_dwconv_s2_bwd_data_bnreduce_bww(const float* dy, const float* w, float* g_in, const float* y_prev,
                                        const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                        const float* bn_invstd, double* bn_partials, double* w_partials, int N, int C, int D,
                                        int H, int W, int accumulate, void* stream);
int msl_dwconv_bwd_weight_finalize(const double* w_partials, int num_partials, float* dw, int C, void* stream);
int msl_dwconv_bwd_weight_num_partials(int N, int C, int D, int H, int W, int stride);
int msl_dwconv_bwd_weight(const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                          double* partials, int N, int C, int D, int H, int W, int stride, void* stream);

/* ---- pointwise Conv3d(Cin,Cout,k1) = per-image GEMM on MFMA : Block.conv2, mobilenet.py:40,45 ------------- */
int msl_pwconv_fwd_num_partials(int N, int Cin, int Cout, int S);
int msl_pwconv_fwd(const float* z, const float* in_scale, const float* in_shift, const float* w, float* y,
                   double* partials, int N, int Cin, int Cout, int S, void* stream);
int msl_pwconv_fwd_fold(const float* z, const double* in_partials, int in_np, double in_count, const float* gamma,
                        const float* beta, float eps, const float* w, float* y, double* partials, int N, int Cin,
                        int Cout, int S, void* stream);
int msl_pwconv_bwd_data(const float* dy, const float* w, float* g_in, int N, int Cin, int Cout, int S,
                        void* stream);
size_t msl_pwconv_bwd_weight_workspace_bytes(int N, int Cin, int Cout, int S);
int msl_pwconv_bwd_weight(const float* dy, const float* z, const float* in_scale, const float* in_shift, float* dw,
                          float* workspace, int N, int Cin, int Cout, int S, void* stream);
/* the same as partial results: `out` = [nslabs][Cout][Cin] slabs whose sum (slab order) is dW; nslabs == 1: dW itself.
 * The training step folds the slabs of every layer in one msl_grad_reduce_batch (kind 0). */
int msl_pwconv_bwd_weight_nslabs(int N, int Cin, int Cout, int S);
int msl_pwconv_bwd_weight_slabs(const float* dy, const float* z, const float* in_scale, const float* in_shift,
                                float* out, int N, int Cin, int Cout, int S, void* stream);
/* the same for n <= 4 layers in ONE launch (the tail blocks, whose launches are pure latency): the eight arrays have n
 * entries and live on the host; every layer must be batchable (msl_pwconv_bwd_weight_batchable == 1) and carries its
 * input affine.  out[k] as `out` above for layer k.  Same arithmetic and slabs as n single calls: mobilenet.py:40 */
int msl_pwconv_bwd_weight_batchable(int N, int Cin, int Cout, int S);
int msl_pwconv_bwd_weight_slabs_batch(const float* const* dy, const float* const* z, const float* const* in_scale,
                                      const float* const* in_shift, float* const* out, const int* Cin, const int* Cout,
                                      const int* S, int n, int N, void* stream);

/* ---- detection heads: loc (C->12) + cls (C->2*ncls) k3 p1 convs, permute/view/cat fused : ssd3d.py:113-169 - */
size_t msl_head_packed_weight_elems(int C, int ncls);
int msl_head_pack_weights(const float* loc_w, const float* cl_w, float* Wf, float* Wb, int C, int ncls,
                          void* stream);
/* the same for all (<= 4) scales of the model in one launch; the five arrays have n entries and live on the host */
int msl_head_pack_weights_batch(const float* const* loc_w, const float* const* cl_w, float* const* Wf, float* const* Wb,
                                const int* C, int n, int ncls, void* stream);
size_t msl_head_fwd_workspace_bytes(int N, int C, int D, int H, int W, int ncls);
int msl_head_conv_fwd(const float* a_pad, const float* Wf, const float* loc_b, const float* cl_b, float* locs,
                      float* scores, float* workspace, int N, int C, int D, int H, int W, int Ptot, int prior_off,
                      int ncls, void* stream);
int msl_head_grad_pack(const float* dlocs, const float* dscores, float* dO_pad, int N, int D, int H, int W,
                       int Ptot, int prior_off, int ncls, void* stream);
/* the same for all (n <= 4) scales in one launch; dO_pad / D / H / W / prior_off are host arrays of n entries */
int msl_head_grad_pack_batch(const float* dlocs, const float* dscores, float* const* dO_pad, const int* D, const int* H,
                             const int* W, const int* prior_off, int n, int N, int Ptot, int ncls, void* stream);
int msl_head_conv_bwd_data(const float* dO_pad, const float* Wb, float* g_a, int N, int C, int D, int H, int W,
                           int ncls, void* stream);
size_t msl_head_bwd_weight_workspace_bytes(int N, int C, int D, int H, int W, int ncls);
/* slabs msl_head_conv_bwd_weight leaves in `workspace`; dloc_w == NULL skips its own reduction (deferred: kind 3) */
int msl_head_conv_bwd_weight_nslabs(int N, int C, int D, int H, int W, int ncls);
int msl_head_conv_bwd_weight(const float* dO_pad, const float* a_pad, float* dloc_w, float* dcl_w, float* dloc_b,
                             float* dcl_b, float* workspace, int N, int C, int D, int H, int W, int ncls,
                             void* stream);

/* ---- bf16 activation path (BASELINE configs[2]/[3]; the reference is fp32: a build-side extension) ------
 * Activations are bf16 in HBM (raw conv outputs, NCDHW; `void*` = bf16 storage); weights, BatchNorm vectors and statistics
 * and all accumulators stay fp32.  The forward kernels of the backbone + heads (mobilenet.py:26-49, ssd3d.py:113-169): */
int msl_stem_conv_fwd_bf16(const float* x, const float* w, void* y_bf16, double* partials, int N, int Cin, int D,
                           int H, int W, int sd, int sh, int sw, void* stream);

/* Eval-mode stem + block-1 depthwise convolution in one pass (inference; mobilenet.py:26-31 followed by the depthwise half of
 * mobilenet.py:34-49): z1 (N,32,D/4,H/4,W/4) = dw3x3x3 s2 (wdw [32][27]) of relu(bn_scale * stem(x) + bn_shift), the stem
 * activation never reaching HBM (in eval mode the BatchNorm between the two is a constant affine).  Values are those of
 * msl_stem_conv_fwd followed by msl_dwconv_fwd, bit for bit.  _supported: 1 if the shape is taken (Cin <= 2, D, H, W multiples
 * of 4, W/2 a multiple of 32 and <= 96), else callers use the two separate entry points. */
int msl_stem_dw_fwd_eval_supported(int N, int Cin, int D, int H, int W);
int msl_stem_dw_fwd_eval(const float* x, const float* w, const float* bn_scale, const float* bn_shift, const float* wdw,
                         float* z, int N, int Cin, int D, int H, int W, void* stream);
int msl_stem_dw_fwd_eval_bf16(const float* x, const float* w, const float* bn_scale, const float* bn_shift, const float* wdw,
                              void* z_bf16, int N, int Cin, int D, int H, int W, void* stream);
int msl_dwconv_fwd_bf16_num_partials(int N, int C, int D, int H, int W, int stride);
/* the register-marching wave kernels of the fp32 path on bf16 storage (square power-of-two planes; MSL_ERR_UNSUPPORTED
 * otherwise - msl_dwconv_*_bf16 try these first and fall back to the LDS-tiled any-shape kernels) */
int msl_dwconv_wave_num_partials(int N, int C, int D, int H, int W, int stride);
/* 1 when a statistics-free (eval-mode) forward of this shape runs on the register-marching rows kernels (W % 4 == 0 planes that
 * are not powers of two, e.g. the 96^2 ... 12^2 planes of a 192^3 volume): fp32 inside msl_dwconv_fwd, bf16 inside
 * msl_dwconv_fwd_wave_bf16 / msl_dwconv_fwd_bf16 when partials == NULL.  mobilenet.py:37-39 */
int msl_dwconv_fwd_eval_rows_ok(int N, int C, int D, int H, int W, int stride);
/* bf16 storage, statistics-free forward of a map of at most 512 voxels (one wave per (image, channel), volume in LDS);
 * msl_dwconv_fwd_bf16 takes this route by itself in eval mode */
int msl_dwconv_fwd_small_eval_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y, int N,
                                   int C, int D, int H, int W, int stride, void* stream);
int msl_dwconv_fwd_wave_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                             double* partials, int N, int C, int D, int H, int W, int stride, int flip, int accumulate,
                             void* stream);
/* stride-2 bwd-data on bf16 gradients (one thread per 2x2x4 input patch; W % 4 == 0); y_prev != NULL: also the
 * BatchNorm-backward partials [2][C][NP] of the layer written, NP = msl_dwconv_bwd_data_bnreduce_num_partials */
int msl_dwconv_bwd_data_s2_patch_bf16(const void* dy, const float* w, void* g_in, const void* y_prev, const float* bn_vec,
                                      double* partials, int N, int C, int D, int H, int W, int accumulate, void* stream);
int msl_dwconv_fwd_wave_bf16_fold(const void* x, const double* in_partials, int in_np, double in_count, const float* gamma,
                                  const float* beta, float eps, const float* w, void* y, double* partials, int N, int C, int D,
                                  int H, int W, int stride, void* stream);
int msl_dwconv_bwd_weight_wave_bf16(const void* dz, const void* x, const float* in_scale, const float* in_shift,
                                    double* partials, int N, int C, int D, int H, int W, int stride, void* stream);
int msl_dwconv_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int C, int D, int H, int W, int stride, void* stream);
int msl_pwconv_fwd_bf16_num_partials(int N, int S);
/* pointwise GEMM on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) */
/* input BatchNorm folded from the producer's partials (in_np <= 64, Cin <= 1024; MSL_ERR_UNSUPPORTED otherwise) */
int msl_pwconv_fwd_bf16_fold(const void* z, const double* in_partials, int in_np, double in_count, const float* gamma,
                             const float* beta, float eps, const float* w, void* y, double* partials, int N, int Cin, int Cout,
                             int S, void* stream);
int msl_pwconv_fwd_bf16(const void* z, const float* in_scale, const float* in_shift, const float* w, void* y,
                        double* partials, int N, int Cin, int Cout, int S, void* stream);
/* relu(bn(y)) of a head feature map -> bf16 CHANNELS-LAST zero-haloed copy (N,D+2,H+2,W+2,C), halo zeroed by the caller */
int msl_bn_relu_materialize_bf16(const void* y, const float* scale, const float* shift, float* plain, void* pad_cl, int N,
                                 int C, int D, int H, int W, void* stream);
/* ... fp32 zero-haloed NCDHW copy (N,C,D+2,H+2,W+2) instead: the bf16 TRAINING step runs its heads on the fp32 kernels */
int msl_bn_relu_materialize_bf16_pad32(const void* y, const float* scale, const float* shift, float* pad, int N, int C, int D,
                                       int H, int W, void* stream);
/* the same in training mode with the affine folded from the producer's statistics partials ([2][C][in_np]) */
int msl_bn_relu_materialize_bf16_pad32_fold(const void* y, const double* in_partials, int in_np, double in_count,
                                            const float* gamma, const float* beta, float eps, float* pad, int N, int C, int D,
                                            int H, int W, void* stream);
size_t msl_head_packed_weight_bf16_elems(int C);
int msl_head_pack_weights_bf16(const float* loc_w, const float* cl_w, void* Wp, int C, int ncls, void* stream);
/* both head convolutions of a scale on v_mfma_f32_16x16x32_bf16, fp32 rows out */
int msl_head_conv_fwd_bf16(const void* a_cl, const void* Wp, const float* loc_b, const float* cl_b, float* locs,
                           float* scores, int N, int C, int D, int H, int W, int Ptot, int prior_off, int ncls,
                           void* stream);
/* ... and the backward kernels of the bf16 training step (BASELINE configs[2]); gradients of activations are bf16 too,
 * weight gradients and all reductions fp32 / fp64 (same partial-sum layouts as the fp32 kernels, folded by
 * msl_grad_reduce_batch / msl_bn_bwd_finalize): */
int msl_pwconv_bwd_data_bf16(const void* dy, const float* w, void* g_in, int N, int Cin, int Cout, int S, void* stream);
int msl_pwconv_bwd_weight_bf16_nslabs(int N, int Cin, int Cout, int S);
int msl_pwconv_bwd_weight_slabs_bf16(const void* dy, const void* z, const float* in_scale, const float* in_shift, float* out,
                                     int N, int Cin, int Cout, int S, void* stream);
int msl_dwconv_bwd_data_bf16(const void* dy, const float* w, void* g_in, int N, int C, int D, int H, int W, int stride,
                             int accumulate, void* stream);
int msl_dwconv_bwd_weight_bf16(const void* dz, const void* x, const float* in_scale, const float* in_shift, double* partials,
                               int N, int C, int D, int H, int W, int stride, void* stream);
int msl_bn_relu_bwd_bf16_num_partials(int N, int S);
int msl_bn_relu_bwd_reduce_bf16(const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, double* partials, int N, int C, int S, void* stream);
int msl_bn_relu_bwd_apply_bf16(const void* g, const void* y, const float* vec, void* dy, int N, int C, int S, void* stream);
int msl_bn_relu_bwd_fused_bf16(const void* g, const void* y, const float* vec, float* dgamma, float* dbeta, void* dy, int N,
                               int C, int S, void* stream);
/* msl_block_bwd_channel_link on bf16 activation / activation-gradient tensors (g_z, z, y_prev, g_y); vectors, taps, sums fp32 */
int msl_block_bwd_channel_link_bf16(void* g_z, const void* z, const float* vec_z, const float* w_dw, const void* y_prev,
                                    const float* vec_y, void* g_y, float* dgamma_z, float* dbeta_z, float* dgamma_y,
                                    float* dbeta_y, float* dw_dw, int N, int C, int D, int H, int W, int stride, int accumulate,
                                    void* stream);
int msl_head_conv_bwd_data_bf16(const float* dO_pad, const float* Wb, void* g_a_bf16, int N, int C, int D, int H, int W,
                                int ncls, void* stream);
int msl_head_conv_bwd_weight_bf16(const float* dO_pad, const void* a_cl, float* dloc_w, float* dcl_w, float* dloc_b,
                                  float* dcl_b, float* workspace, int N, int C, int D, int H, int W, int ncls,
                                  void* stream);
/* the fused stem backward (msl_dwconv_s2_bwd_bnreduce_bww -> msl_bn_bwd_finalize_coef -> msl_stem_conv_bwd_weight_fused) on
 * bf16 dL/dz and raw stem output: dL/d(stem activation) is never stored */
int msl_dwconv_s2_bwd_bnreduce_bww_bf16(const void* dy, const float* w, const void* y_prev, const float* bn_vec,
                                        double* bn_partials, double* w_partials, float* w_taps_t, int N, int C, int D, int H,
                                        int W, void* stream);
int msl_stem_conv_bwd_weight_fused_bf16(const void* dz, const float* w1_t, const void* yraw, const float* bn_vec, const float* x,
                                        float* dw, float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                        void* stream);
int msl_stem_conv_bwd_weight_bnapply_bf16(const void* g, const void* yraw, const float* bn_vec, const float* x, float* dw,
                                          float* workspace, int N, int Cin, int D, int H, int W, int sd, int sh, int sw,
                                          void* stream);

/* ---- priors, box math, matching, MultiBox loss : ssd3d.py:286-342, utils.py:42-149, ssd3d.py:741-941 ------- */
int msl_make_priors(float* out, int row_off, int D0, int D1, int D2, double scale, int boxes_per_location,
                    void* stream);
/* op 0 cxcycz_to_xyz (utils.py:42) 1 xyz_to_cxcycz (:92) 2 cxcycz_to_gcxgcygcz (:71) 3 gcxgcygcz_to_cxcycz (:54) */
int msl_box_transform(const float* boxes, const float* priors, float* out, int n, int op, void* stream);
/* find_jaccard_overlap3d (utils.py:125) / find_intersection3d (utils.py:105) */
int msl_iou_matrix(const float* set1, const float* set2, float* out, int n1, int n2, int intersection_only,
                   void* stream);
/* MultiBoxLoss.forward matching part (ssd3d.py:786-887) for the whole batch.  gt_* are the per-image lists
 * concatenated; obj_off (N+1) int32 prefix offsets.  scratch: overlap (N,P) f32, obj (N,P) i32,
 * prior_for_obj (T) i32.  Outputs: true_classes (N,P) i64 in {-1,0,label}, true_locs (N,P,6), matched (N,P) i64
 * (may be NULL).  soft != 0 selects the two-threshold band of ssd3d.py:878-881. */
int msl_multibox_match(const float* gt_boxes, const long long* gt_labels, const int* obj_off, int total_objects,
                       const float* priors_c, int N, int P, float thr_lo, float thr_hi, int soft, float* overlap,
                       int* obj, int* prior_for_obj, long long* true_classes, float* true_locs, long long* matched,
                       void* stream);
/* the same; additionally *npos = number of positive priors of the batch (ssd3d.py:890-893), for msl_multibox_loss_pack */
int msl_multibox_match_count(const float* gt_boxes, const long long* gt_labels, const int* obj_off, int total_objects,
                             const float* priors_c, int N, int P, float thr_lo, float thr_hi, int soft, float* overlap,
                             int* obj, int* prior_for_obj, long long* true_classes, float* true_locs, long long* matched,
                             int* npos, void* stream);
size_t msl_multibox_loss_workspace_bytes(void);
/* loss part (ssd3d.py:890-941): loss_out[0] = conf_loss, [1] = loc_loss, [2] = number of positives */
int msl_multibox_loss_fwd(const float* locs, const float* scores, const long long* true_classes,
                          const float* true_locs, double* workspace, float* loss_out, int N, int P, int ncls,
                          void* stream);
int msl_multibox_loss_bwd(const float* locs, const float* scores, const long long* true_classes,
                          const float* true_locs, const float* loss_out, const float* upstream, float* dlocs,
                          float* dscores, int N, int P, int ncls, void* stream);

/* loss forward + backward for the training loop (2 launches; publishes loss_out as above).  nan_flag (may be NULL):
 * |= 1 if locs holds a NaN, |= 2 if scores does (the guards of ssd3d.py:258-261 without a pass of their own) */
int msl_multibox_loss_fwd_bwd(const float* locs, const float* scores, const long long* true_classes,
                              const float* true_locs, double* workspace, float* loss_out, const float* upstream,
                              float* dlocs, float* dscores, int* nan_flag, int N, int P, int ncls, void* stream);

/* Training hot loop, ONE launch: the loss terms of ssd3d.py:890-941, their gradients (the number of positives comes from
 * msl_multibox_match_count, so no thread waits for the loss sums) and the scatter of those gradients into the zero-haloed
 * head-gradient images of msl_head_grad_pack (dO_pad / D / H / W / prior_off: host arrays of n <= 4 scales; rows
 * a*6+q = box regressions of anchor a, 12 + a*ncls + c = class scores; padding rows are not written and must be zero).
 * partials: 2 * msl_multibox_loss_pack_num_partials(N, P) doubles; a kind-4 entry of msl_grad_reduce_batch folds them into
 * loss_out = [conf_loss, loc_loss, n_positives].  upstream / nan_flag as msl_multibox_loss_fwd_bwd. */
int msl_multibox_loss_pack_num_partials(int N, int P);
int msl_multibox_loss_pack(const float* locs, const float* scores, const long long* true_classes, const float* true_locs,
                           const int* npos, const float* upstream, double* partials, int* nan_flag, float* const* dO_pad,
                           const int* D, const int* H, const int* W, const int* prior_off, int n, int N, int P, int ncls,
                           void* stream);

/* Optional loss variants the reference keeps as commented code (ssd3d.py:758-760 loss choices, :926-932 hard-negative
 * mining); flags: 1 = hard-negative mining (keep the neg_pos_ratio * n_positives largest negative losses per image),
 * 2 = smooth-L1 (nn.SmoothL1Loss, beta 1) for the localisation loss, 4 = focal confidence loss (MONAI FocalLoss gamma 2,
 * weight 0.25, background excluded: two classes only, else MSL_ERR_UNSUPPORTED); 0 = the live path.  Forward, and
 * backward too when dlocs / dscores are given (upstream as above).  var_ws: ..._var_workspace_bytes(N, P) bytes. */
size_t msl_multibox_loss_var_workspace_bytes(int N, int P);
int msl_multibox_loss_var(const float* locs, const float* scores, const long long* true_classes, const float* true_locs,
                          double* workspace, void* var_ws, float* loss_out, const float* upstream, float* dlocs,
                          float* dscores, int* nan_flag, int N, int P, int ncls, int flags, int neg_pos_ratio,
                          void* stream);

/* ---- LSSD3D.detect_objects (ssd3d.py:344-460): softmax, decode, filter, sort, 3D NMS, top-k ---------------
 * cap = 10*top_k (<= 4096), Wn = ceil(cap/64), K1 = ncls-1.  Caller-allocated scratch:
 *   probs (N,K1,P) f32; boxes (N,P,6) f32; sorted_idx (N,K1,cap) i32; ncand (N*K1) i32;
 *   mask (N,K1,cap,Wn) u64; keep_bits (N,K1,Wn) u64; nkept (N*K1) i32; tmp_scores (N,K1*cap) f32; tmp_ref same i32;
 *   select_ws: msl_detect_select_ws_ints(N,P,ncls) i32 (score histogram + shortlist of the candidates that can reach the
 *   best cap: the stable order is then found among ~cap candidates instead of all P)
 * Outputs: out_boxes (N,top_k,6), out_scores (N,top_k), out_labels / out_prior (N,top_k) i64, out_count (N) i32. */
size_t msl_detect_select_ws_ints(int N, int P, int ncls);
int msl_detect_objects(const float* locs, const float* scores, const float* priors_c, int N, int P, int ncls,
                       float min_score, float max_overlap, int top_k, float* probs, float* boxes, int* sorted_idx,
                       int* ncand, unsigned long long* mask, unsigned long long* keep_bits, int* nkept,
                       float* tmp_scores, int* tmp_ref, int* select_ws, float* out_boxes, float* out_scores,
                       long long* out_labels, long long* out_prior, int* out_count, void* stream);

/* ---- optimiser + NaN guard : ssd3d.py:704-722, :258-261 ---------------------------------------------------- */
/* hp (device, 8 floats): step_size(bias), step_size(other), sqrt(bias_correction2), beta1, beta2, eps,
 * weight_decay, gradient scale.  is_bias (n bytes): 1 for elements of '.bias' parameters (2*lr group). */
int msl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* hp,
                  const unsigned char* is_bias, int n, void* stream);
int msl_nan_flag(const float* x, size_t n, int* flag, int bit, void* stream);
/* the same for two tensors in one launch: *flag |= bit_a if a holds a NaN, |= bit_b if b does */
int msl_nan_flag2(const float* a, size_t na, int bit_a, const float* b, size_t nb, int bit_b, int* flag, void* stream);
/* Batched gradient reduction (autograd's accumulation of the conv weight gradients, ssd3d.py:467-531 backward): the
 * weight-gradient kernels leave partial sums (fp32 slabs / fp64 partials) in their workspaces; ONE launch folds the
 * listed ones into the flat gradient arena in a fixed order.  The table is built on the host with _table_set (which
 * returns the number of workgroups of the entry; first_block = running sum) and uploaded by the caller.
 * kind 0: fp32 slabs [nslabs][stride] -> dst[i];  1: fp64 partials [count][nslabs] -> dst[i];  2: stem slabs, padded
 * [32][32*p1] image -> dst[co*p0 + k];  3: head slabs -> dst = dloc_w, dst2 = dcl_w (p0 = C, p1 = MT, p2 = 12 + 2*ncls);
 * 4: the loss partials of msl_multibox_loss_pack (src fp64 [nslabs][2], dst2 = its int positives counter, count = 1)
 * -> dst = loss_out [conf_loss, loc_loss, n_positives]. */
size_t msl_grad_reduce_entry_bytes(void);
int msl_grad_reduce_table_set(void* host_table, int index, int first_block, int kind, const void* src, float* dst,
                              float* dst2, int nslabs, int count, long long stride, int p0, int p1, int p2);
int msl_grad_reduce_batch(const void* table, int n_entries, int total_blocks, void* stream);
/* the same; block_entry[total_blocks] (device) names every workgroup's table entry, so no workgroup searches the table */
int msl_grad_reduce_batch_indexed(const void* table, int n_entries, const int* block_entry, int total_blocks, void* stream);
/* stream fork/join (hipEvent with timing disabled): record on the producer stream, wait on the consumer stream */
int msl_event_create(void** out);
int msl_event_create_device(void** out);                      /* orders streams of ONE device only (hipEventDisableSystemFence): never for host waits or other devices */
/* Fork without a record packet: the (skip + 1)-th kernel launch the calling THREAD issues from now on completes `ev`
 * (stopEvent of hipExtLaunchKernel) exactly as if msl_event_record(ev, <that launch's stream>) followed it.
 * msl_thread_launch_count(): kernel launches issued by the calling thread so far (launches per entry point = difference
 * around a call); msl_stop_event_pending(): 1 while an armed event has not met its launch. */
int msl_arm_stop_event(void* ev, int skip);
int msl_stop_event_pending(void);
int msl_thread_launch_count(void);      /* not an error code: the count, modulo 2^31 */
int msl_event_create_timed(void** out);                       /* timing-enabled event (launch-duration measurements) */
int msl_event_elapsed_ms(void* start, void* stop, float* out_ms);
int msl_event_destroy(void* ev);
int msl_event_record(void* ev, void* stream);
int msl_stream_wait_event(void* stream, void* ev);
/* native replay of a recorded launch sequence (generated trampolines, csrc/gen_runner.py): fn_ids from
 * msl_program_fn_id(); slots = n x stride raw 64-bit argument values; *failed_at = index of the failing call */
int msl_program_fn_id(const char* name);
int msl_run_program(const int* fn_ids, const unsigned long long* slots, int stride, int n, int* failed_at);
/* the same issued by two host threads: calls with lane[i] == 1 (the side streams' work) go to a persistent worker thread,
 * lane 0 (the dependency chain's stream) stays with the caller; wait_for[i] >= 0 names the msl_event_record entry that
 * must have been issued before the msl_stream_wait_event of entry i.  Every stream must belong to one lane.  Callers are
 * serialised internally (one worker thread, one posted program at a time): safe from several host threads. */
int msl_run_program_mt(const int* fn_ids, const unsigned long long* slots, int stride, int n, const int* lane,
                       const int* wait_for, int device, int* failed_at);
/* asynchronous 32-bit fill (used to clear flags / counters inside a launch sequence) */
int msl_fill_u32(void* dst, unsigned int value, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif
