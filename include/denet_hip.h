/*
 * denet_hip.h — C ABI of libdenet_hip.so, the MI355X (gfx950) implementation of the DeNet training hot path.
 *
 * Every entry point replaces one device-side operation that the reference (lachlants/denet) reaches through
 * Theano GpuOps / cuDNN, or one host-side C++ function it loads through common.import_c. The reference
 * interface each function stands in for is cited as `denet/...:line`.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the parameter name ends in `_host`;
 *   - activations are NHWC fp32 with a *physical* channel count (multiple of 32 for convolution operands,
 *     4 for the network input), filters are KRSC fp32 holding the already flipped (correlation) taps;
 *   - functions enqueue work on `stream` and return immediately: they never allocate, never synchronise;
 *   - return value 0 = ok, DENET_ERR_ARG (-1000) = rejected arguments, other negative = -(hipError_t);
 *     denet_last_error() returns a thread-local description of the last failure;
 *   - buffers are owned by the caller; `workspace` sizes come from the matching *_workspace_bytes().
 */
#ifndef DENET_HIP_H
#define DENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the HIP stream handle, declared here so a plain-C host can include this file without the HIP headers (an
 * identical re-declaration of hip_runtime_api.h's typedef, which C11 and C++ both allow) */
typedef struct ihipStream_t* hipStream_t;

#define DENET_ERR_ARG (-1000)
#define DENET_TAP_THEANO 0 /* denet/layer/denet_sparse.py:72-84  (i*extent)/(gs-1), round half to even */
#define DENET_TAP_CUDA 1   /* denet/layer/denet_sparse_op.py:65-71 i*extent*(1/(gs-1)), lroundf           */
#define DENET_SOLVER_SGD 0      /* denet/model/model_cnn.py:282-287 */
#define DENET_SOLVER_NESTEROV 1 /* denet/model/model_cnn.py:289-294 ("torch" == "nesterov") */

/* ---- runtime ---------------------------------------------------------------------------------------- */
const char* denet_last_error(void);
int denet_abi_version(void);
int denet_device_info(int device, int* cu_count, int* clock_khz, char* arch, int arch_len);

/* host helper of the RoI list editing (denet/layer/denet_sparse.py:184-187 `random.sample(list, n)`): advances a copy
 * of CPython's MT19937 state exactly like random.sample(range(n), k) and returns the k chosen indices.
 * mt_host: 624 state words, pos_host: the position word (random.getstate()[1][624]); all HOST pointers.        */
int denet_host_py_random_sample(unsigned* mt_host, int* pos_host, int n, int k, int* pool_ws_host, int* out_host);
/* the whole training-time RoI list editing of a batch (denet/layer/denet_sparse.py:184-201: random.sample trim,
 * 4 random.uniform draws per random box, ground truth over the tail), call for call on the same generator state.
 * det:[B,S,5] rows of denet_samples_finish_host; gt: concatenated [n,4] doubles, gt_off:[B+1]; ws: 2*S ints.
 * out_pr:[B,S], out_box:[B,S,4] doubles (the reference's Python list values), out_box_f32: the uploaded array. */
int denet_host_edit_samples(unsigned* mt_host, int* pos_host, const float* det_host, const int* count_host, int B, int S,
                            int n_keep, const double* gt_host, const int* gt_off_host, int sample_gt, int* ws_host,
                            double* out_pr_host, double* out_box_host, float* out_box_f32_host);
/* the same editing with the generator's outputs drawn AHEAD of the hand-off (the list editing has to wait for the device's
 * proposal, the numbers it will draw do not). denet_host_mt_prefetch advances a COPY of the state by n 32-bit outputs into
 * out_host[n] and records the state words after every refill: snaps_host[j][624], snap_first_host[j] = index of the first output
 * drawn from snapshot j (snapshot 0 = the state at entry, which keeps the entry position; later ones start at position 0), so
 * that the state after any number of consumed outputs can be handed back to `random`. denet_host_edit_samples_stream is
 * denet_host_edit_samples reading that stretch through *cursor_host; *exhausted_host = 1: the stretch ran dry, the outputs are
 * incomplete and the caller repeats the batch on the live generator. */
int denet_host_mt_prefetch(unsigned* mt_host, int* pos_host, long n, unsigned* out_host, unsigned* snaps_host,
                           long* snap_first_host, int max_snaps, int* n_snaps_host);
int denet_host_edit_samples_stream(const unsigned* stream_host, long n_stream, long* cursor_host, int* exhausted_host,
                                   const float* det_host, const int* count_host, int B, int S, int n_keep, const double* gt_host,
                                   const int* gt_off_host, int sample_gt, int* ws_host, double* out_pr_host, double* out_box_host,
                                   float* out_box_f32_host);
/* the host's whole share of the hand-off in one call (the device stands idle meanwhile): denet_samples_finish_host into
 * det_out_host [B][S][5], then denet_host_edit_samples_stream on it. Same outputs as the two calls. */
int denet_host_handoff_stream(const unsigned* stream_host, long n_stream, long* cursor_host, int* exhausted_host,
                              const int* box_host, const float* absd_host, const int* count_host, int H, int W, int B, int S,
                              int n_keep, const double* gt_host, const int* gt_off_host, int sample_gt, int* ws_host,
                              float* det_out_host, double* out_pr_host, double* out_box_host, float* out_box_f32_host);
/* ... and the part of it the device is waiting for - the bbox array alone, from the packed proposal's integer boxes: the same
 * selection, random boxes and float32 values, without the score arithmetic and the double-precision lists (the caller produces
 * those later with denet_host_handoff_stream from the same cursor). */
int denet_host_handoff_boxes_stream(const unsigned* stream_host, long n_stream, long* cursor_host, int* exhausted_host,
                                    const int* box_host, const int* count_host, int H, int W, int B, int S, int n_keep,
                                    const double* gt_host, const int* gt_off_host, int sample_gt, int* ws_host, float* out_box_f32_host);
/* the same with the uniform doubles of the stretch read from a table made ahead of the hand-off (uniforms_host [n_stream],
 * denet_host_mt_uniforms; NULL: computed in place): a random box is then four table reads. The values random.uniform would
 * return (denet/layer/denet_sparse.py:189-194) - same bits either way. */
int denet_host_handoff_boxes_stream_u(const unsigned* stream_host, long n_stream, long* cursor_host, int* exhausted_host,
                                      const int* box_host, const int* count_host, int H, int W, int B, int S, int n_keep,
                                      const double* gt_host, const int* gt_off_host, int sample_gt, int* ws_host, float* out_box_f32_host,
                                      const double* uniforms_host);
/* uniforms_host[p] = the double random.random() returns when the generator stands at output p of the stretch (genrand_res53 of
 * outputs p and p + 1; CPython Modules/_randommodule.c), p < n - 1; [n - 1] = 0. */
int denet_host_mt_uniforms(const unsigned* stream_host, long n, double* uniforms_host);
/* detection targets of a batch (denet/layer/denet_detect.py:147-235) in RoI-major layout: fp32 IoU matrix in the
 * operation order of common/theano_util.py:38-59, class / class x fitness-bin targets for IoU > t0, box-regression
 * target of the arg-max ground truth for IoU > t1, rows normalised and divided by S. gt: concatenated [n,4] doubles,
 * gt_off:[B+1], gt_class:[n]; roi:[B,S,4] doubles; det:[B*S,s0]; valid:[B*S] and reg:[B*S,8] or both NULL.      */
int denet_host_detect_targets(const double* gt_host, const int* gt_off_host, const int* gt_class_host,
                              const double* roi_host, int B, int S, int s0, int null_class, int fitness_num,
                              int jointfit, double t0, double t1, float* det_host, float* valid_host, float* reg_host,
                              float* indfit_host /* [B*S,fitness_num] or NULL */);

/* ---- convolution  (denet/layer/convolution.py:80-83 -> cuDNN conv fwd; model_cnn.py:318 tensor.grad ->
 *      cuDNN bwd-data / bwd-filter).  x:[N,H,W,C]  w:[K,R,S,C]  y:[N,OH,OW,K]; `S` may be padded beyond the
 *      real tap count S_real when C < 32 (first layer: C=4, S=8, S_real=7); `add` (optional, shape of the
 *      output) is summed in the epilogue: residual / skip accumulation without an extra pass.          */
int denet_conv_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, int N, int H, int W,
                   int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream);
/* the same with an activation in the epilogue: y = act(conv(x, w) + bias + add), relu != 0: max(., 0). Used at
 * inference, where a batch-norm layer behind the convolution is folded into its filters (denet_bn_fold) and the ReLU /
 * residual add of batch_norm_relu.py:34-48, resnet.py:109-113 ride in the convolution's epilogue. */
int denet_conv_fwd_act(const float* x, const float* w, const float* bias, const float* add, float* y, int relu, int N, int H,
                       int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW,
                       hipStream_t stream);
/* the same (no activation), and the epilogue also writes the per-channel sums of y for the batch norm that follows
 * (denet/layer/batch_norm.py:50-53, batch_norm_relu.py:34-54): stats_partial [ceil(N*OH*OW/128)][2][K] doubles (sum |
 * sum of squares per row tile), *stats_rows = number of rows. denet_bn_fwd_train_pre consumes them.               */
int denet_conv_fwd_stats(const float* x, const float* w, const float* bias, const float* add, float* y,
                         double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                         int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream);
int denet_conv_dgrad(const float* dy, const float* w, const float* add, float* dx, int N, int H, int W, int C, int K,
                     int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream);
size_t denet_conv_wgrad_workspace_bytes(int N, int C, int K, int R, int S, int OH, int OW);
/* kernel instantiation picked by this thread's last conv launch: mode 0 fwd / 1 dgrad / 2 wgrad, tile BMxBN, LDS
 * buffers, grid.y (wgrad split slices / dgrad stride classes) — profiling bookkeeping only                      */
/* Winograd F(m x m, 3x3) paths of the same stride-1 pad-1 3x3 convolution, tile = m in {2, 4}: forward, data gradient and
 * filter gradient with 2.25x (m = 2) / 4x (m = 4) fewer multiplications and extra HBM-bound transforms; they pay for
 * many channels. H, W multiples of m; C, K multiples of 32. workspace: denet_conv_wino_workspace_bytes (transformed
 * filters, input tiles and products); split_ws: split-K slices of the filter-gradient product (as denet_conv_wgrad).
 * denet_conv_wino_tune measures the launch configuration of the component GEMMs once per geometry (it synchronises). */
size_t denet_conv_wino_workspace_bytes(int tile, int N, int H, int W, int C, int K);
int denet_conv_wino_tune(float* workspace, size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N,
                         int H, int W, int C, int K, hipStream_t stream);
/* u_cached: transformed filters from denet_conv_wino_filter (dgrad = 0 / 1), or NULL to transform inside the call;
 * v_keep: [(m+2)^2 * tiles * C] buffer receiving the transformed input of the forward pass, v_cached: the same buffer
 * handed to the filter gradient of that layer (same tile), or NULL.                                             */
int denet_conv_wino_filter(const float* w, float* u, int tile, int dgrad, int C, int K, hipStream_t stream);
int denet_conv_wino_fwd(const float* x, const float* w, const float* u_cached, float* v_keep, const float* bias,
                        const float* add, float* y, float* workspace, size_t workspace_bytes, int tile, int N, int H, int W,
                        int C, int K, hipStream_t stream);
int denet_conv_wino_fwd_act(const float* x, const float* w, const float* u_cached, float* v_keep, const float* bias,
                            const float* add, float* y, int relu, float* workspace, size_t workspace_bytes, int tile, int N,
                            int H, int W, int C, int K, hipStream_t stream);
/* denet_conv_wino_fwd with the batch-norm column sums written by the output transform: stats_partial [rows][2][K] doubles,
 * *stats_rows = rows, or 0 (no sums written) when 256 is not a multiple of K/4                                    */
int denet_conv_wino_fwd_stats(const float* x, const float* w, const float* u_cached, float* v_keep, const float* bias,
                              const float* add, float* y, double* stats_partial, size_t stats_bytes, int* stats_rows,
                              float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                              hipStream_t stream);
/* ..._up: the same on the 2 x 2 nearest-neighbour up-sampling of x_small [N][H/2][W/2][C] (H, W: the layer's input size) - the
 * pool-inverse layer in front of the convolution (denet/layer/pool_inv.py:10-41) evaluated inside the input transform */
int denet_conv_wino_fwd_stats_up(const float* x_small, const float* w, const float* u_cached, float* v_keep, const float* bias,
                              const float* add, float* y, double* stats_partial, size_t stats_bytes, int* stats_rows,
                              float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                              hipStream_t stream);
/* OPT-IN, not the fp32 path: C[M][N] = A[M][K] B[N][K]^T (+ bias[N]) with every product as a 3-term bf16 split on the bf16
 * matrix cores (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation): the forward pass of a 1x1 stride-1 convolution
 * (convolution.py:80-83; the detection head) at ~1e-6 relative error instead of the exact fp32 FMA chain of denet_conv_fwd. */
int denet_gemm_bf16x3_ok(int M, int N, int K);
int denet_gemm_bf16x3_nt(const float* a, const float* b, const float* bias, float* c, int M, int N, int K, hipStream_t stream);
/* dst [C][R] = src [R][C]^T (fp32): operands of the data / filter gradient in the K-contiguous form of the call above */
int denet_transpose_f32(const float* src, float* dst, int R, int C, hipStream_t stream);
/* Fused F(2x2,3x3) convolution for Ci = 64 (stride 1, pad 1; H, W even; Co a multiple of 64): transforms and the
 * 16 component products in one kernel, x -> y only. u = denet_conv_wino_filter(tile 2) output: dgrad = 0 for the forward
 * pass (denet/layer/convolution.py:80-83), dgrad = 1 for the data gradient (x = dy; model_cnn.py:318). Optional bias [Co],
 * add [N,H,W,Co], relu (y = max(y, 0)) and batch-norm column sums stats_partial [rows][2][Co] (doubles; Co = 64: a row per
 * workgroup of the launch, else one per block; rows <= N*ceil(H/16)*ceil(W/16) = what the buffer must hold; *stats_rows =
 * rows). Work items are 16x16-pixel blocks; maps that are no multiple of 16 waste the overhang. */
int denet_conv_wino2f_ok(int N, int H, int W, int Ci, int Co);
int denet_conv_wino2f(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                      double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int Ci, int Co,
                      hipStream_t stream);
/* The filter gradient of the same layers (C = K = 64), fused the same way: dw [64][3][3][64] from x and dy [N,H,W,64]
 * (model_cnn.py:318). workspace: denet_conv_wino2f_wgrad_workspace_bytes (per-workgroup partial sums, added in a fixed order). */
int denet_conv_wino2f_wgrad_ok(int N, int H, int W, int C, int K);
size_t denet_conv_wino2f_wgrad_workspace_bytes(int N, int H, int W);
int denet_conv_wino2f_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H,
                            int W, int C, int K, hipStream_t stream);
/* ---- The first layer, `C.B[64,7,2]` on the 3-channel image (csrc/stem.hip; reference: denet/layer/convolution.py:80-83 with the
 *      desc of examples/resnet34-imagenet.sh:7, filter gradient: tensor.grad, model_cnn.py:318), as kernels of their own: the
 *      reduction walks the 147 real taps (padded to 160) instead of the 7 x 8 x 4 = 224 of the padded KRSC layout. denet_conv_fwd /
 *      denet_conv_fwd_stats / denet_conv_wgrad hand this geometry over themselves when denet_conv_stem_ok(pass: 0 forward, 1
 *      filter gradient) says so (x [N][H][W][4], w / dw [64][7][8][4], stride 2, pad 3, even H and W; DENET_STEM=0 keeps the
 *      generic kernels). Forward: optional bias and batch-norm column sums, stats_partial [rows][2][64] doubles with rows =
 *      the launch's workgroups (*stats_rows). Filter gradient: workspace = one partial per workgroup, added in a fixed order. */
int denet_conv_stem_ok(int pass, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW);
int denet_conv_stem_fwd(const float* x, const float* w, const float* bias, float* y, double* stats_partial, size_t stats_bytes,
                        int* stats_rows, int N, int H, int W, hipStream_t stream);
/* ..._from with x_nchw != 0: x is the image batch in the reference's own layout [N][3][H][W] (dataset/__init__.py:359) - the
 * training step then never makes an NHWC copy of its input */
int denet_conv_stem_fwd_from(const float* x, int x_nchw, const float* w, const float* bias, float* y, double* stats_partial,
                             size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream);
int denet_conv_stem_fwd_act(const float* x, int x_nchw, const float* w, const float* bias, float* y, int relu, double* stats_partial,
                            size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream);   /* + y = max(y, 0) */
int denet_conv_stem_wgrad_from(const float* x, int x_nchw, const float* dy, float* dw, float* workspace, size_t workspace_bytes,
                               int N, int H, int W, hipStream_t stream);
size_t denet_conv_stem_wgrad_workspace_bytes(void);
int denet_conv_stem_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int N, int H, int W,
                          hipStream_t stream);
/* ---- Winograd passes whose input is formed on the fly from the batch-norm layer next to them (csrc/winograd.hip,
 *      wino_prep_kernel). Reference: the BN -> conv chains of the residual blocks (denet/layer/resnet.py:60-90,
 *      batch_norm_relu.py:34-54): a pointwise pass writes a tensor the next convolution's input transform re-reads at once.
 *      denet_bn_link describes the batch norm; all values are bit-identical to the separate passes.
 *        forward  (denet_conv_wino_fwd_fold):  x = pre-normalisation tensor, aux = residual input or NULL, gamma / beta /
 *                 mean / invstd = the layer's coefficients (denet_bn_stats_final), relu; out receives the activation.
 *        backward (denet_conv_wino_dgrad_fold): x = the convolution's output (= the batch norm's input), aux = gradient of
 *                 the batch norm's output, y = its forward output for the ReLU mask (NULL: recomputed from x, needs beta),
 *                 coef = [2][C] from denet_bn_bwd_sums; out = NULL or the masked gradient (residual branch). The gradient
 *                 of the convolution's output itself is never written: the call forms the transformed input of the
 *                 data-gradient products and dm_out = A dy A^T for denet_conv_wino_wgrad_dm.                              */
typedef struct denet_bn_link {
    const float* x;
    const float* aux;
    const float* y;
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    const float* coef;
    float* out;
    int relu;
} denet_bn_link;
int denet_conv_wino_fwd_fold(const denet_bn_link* bn, const float* w, const float* u_cached, float* v_keep, const float* bias,
                             const float* add, float* y, int relu, double* stats_partial, size_t stats_bytes, int* stats_rows,
                             float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                             hipStream_t stream);
/* transform_done_event: NULL or a hipEvent_t recorded on `stream` right behind the transform kernel (dm_out complete), so that
 * the filter-gradient chain of a second stream can start while this call's products still run.
 * sums_of (optional, also denet_conv_wino_dgrad_sums / denet_conv_wino2f_sums): dx is the gradient of the OUTPUT of another batch
 * norm (x = that layer's input, y = its forward output or NULL, gamma / beta / mean / invstd, relu); the output transform then
 * also writes that layer's two backward reductions sum(g), sum(g * xhat) (g = dx masked by its ReLU) as stats_partial
 * [rows][2][C] doubles for denet_bn_bwd_final - instead of the pass over dx, x and y that bn_bwd_partial_kernel makes.
 * *stats_rows = 0: channel count not supported, nothing written. */
int denet_conv_wino_dgrad_fold(const denet_bn_link* bn, float* dm_out, const float* w, const float* u_cached, const float* add,
                               float* dx, const denet_bn_link* sums_of, double* stats_partial, size_t stats_bytes, int* stats_rows,
                               float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                               void* transform_done_event, hipStream_t stream);
int denet_conv_wino_dgrad_sums(const float* dy, const float* w, const float* u_cached, const float* add, float* dx,
                               const denet_bn_link* sums_of, double* stats_partial, size_t stats_bytes, int* stats_rows,
                               float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                               hipStream_t stream);
/* denet_conv_dgrad_sums: the implicit-GEMM data gradient with the same request, any stride (rows = stride^2 *
 * ceil(N*(H/stride)*(W/stride) / 128): one per parity class of input pixels and row tile). */
int denet_conv_dgrad_sums(const float* dy, const float* w, const float* add, float* dx, const denet_bn_link* sums_of,
                          double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K, int R,
                          int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream);
/* the data gradient of a 1x1 stride-1 convolution (convolution.py:80-83 under theano.grad) as a forward product over the
 * TRANSPOSED filter wt [C][K] (denet_transpose_f32 of w [K][C]): dx [N][H][W][C] = dy [N][H][W][K] . wt^T (+ add). The forward
 * loop reads both operands reduction-contiguous: 118 -> 137 TFLOP/s on the 4736 <- 1536 head layer; bit-identical to
 * denet_conv_dgrad. sums_of / stats_partial / stats_rows as in denet_conv_dgrad_sums, or all null.                    */
/* denet_conv_dgrad / _sums with the filter TRANSPOSED, wt [R][S][C][K] (denet_transpose_f32(w, wt, K, R*S*C)): the same products in
 * the same order (bit-identical), the filter operand reduction-contiguous like the forward pass's. tensor.grad w.r.t. the input
 * of conv2d, model_cnn.py:318 / convolution.py:80-83. sums_of / stats_partial / stats_rows as in denet_conv_dgrad_sums, or null. */
int denet_conv_dgrad_t(const float* dy, const float* wt, const float* add, float* dx, const denet_bn_link* sums_of,
                       double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K, int R, int S,
                       int S_real, int stride, int pad, int OH, int OW, hipStream_t stream);
/* The data gradient of a 3x3 STRIDE-2 pad-1 convolution (the first convolution of a ResNet stage: convolution.py:80-83 under
 * tensor.grad, model_cnn.py:318 -> cuDNN bwd-data) with the four parity classes of input pixels in one workgroup (csrc/dgrad_s2.hip):
 * x [N,H,W,C] -> y [N,H/2,W/2,K], even H and W, C and K multiples of 64. w_packed = denet_conv_dgrad_s2_pack(w [K][3][3][C]) =
 * [K/16][9][C][16]. add / sums_of / stats_partial / stats_rows as denet_conv_dgrad_sums (rows = denet_conv_dgrad_s2_stats_rows: one
 * per block of 8 x 8 output positions), or null. Same sums as denet_conv_dgrad in another order (not bit-identical to it). */
int denet_conv_dgrad_s2_ok(int N, int H, int W, int C, int K);
int denet_conv_dgrad_s2_stats_rows(int N, int H, int W);
int denet_conv_dgrad_s2_pack(const float* w, float* packed, int C, int K, hipStream_t stream);
int denet_conv_dgrad_s2(const float* dy, const float* w_packed, const float* add, float* dx, const denet_bn_link* sums_of,
                        double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                        hipStream_t stream);
int denet_conv_dgrad_1x1t(const float* dy, const float* wt, const float* add, float* dx, const denet_bn_link* sums_of,
                          double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                          hipStream_t stream);
int denet_conv_wino2f_sums(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                           double* stats_partial, size_t stats_bytes, int* stats_rows, const denet_bn_link* sums_of, int N, int H,
                           int W, int Ci, int Co, hipStream_t stream);
/* Tile-parallel fused F(4x4,3x3) convolution (csrc/wino4t.hip): input transform, the 36 component products and the output
 * transform in one kernel, x -> y only (3x3, stride 1, pad 1; H, W multiples of 4; C a multiple of 16; K of 64). Same operator and
 * arguments as denet_conv_wino2f_sums (forward pass: denet/layer/convolution.py:80-83; data gradient with x = dy and the
 * data-gradient filters: model_cnn.py:318), with C = the pass's reduction channels, K = the channels it writes. u_packed: the
 * F(4x4) transformed filters of denet_conv_wino_filter(tile 4) [36][K][C] re-laid as [C/16][36][K][16] by denet_conv_wino4t_pack.
 * stats_partial [rows][2][K] doubles, rows = denet_conv_wino4t_stats_rows (one per block of 4 x 8 tiles). */
int denet_conv_wino4t_ok(int N, int H, int W, int C, int K);
int denet_conv_wino4t_stats_rows(int N, int H, int W);
int denet_conv_wino4t_pack(const float* u, float* packed, int C, int K, hipStream_t stream);
int denet_conv_wino4t_sums(const float* x, const float* u_packed, const float* bias, const float* add, float* y, int relu,
                           double* stats_partial, size_t stats_bytes, int* stats_rows, const denet_bn_link* sums_of, int N, int H,
                           int W, int C, int K, hipStream_t stream);
int denet_conv_wino_wgrad_dm(const float* x, const float* dm, const float* v_cached, float* dw, float* workspace,
                             size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N, int H, int W, int C,
                             int K, hipStream_t stream);
int denet_conv_wino_dgrad(const float* dy, const float* w, const float* u_cached, const float* add, float* dx,
                          float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                          hipStream_t stream);
int denet_conv_wino_wgrad(const float* x, const float* dy, const float* v_cached, float* dw, float* workspace,
                          size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N, int H, int W, int C,
                          int K, hipStream_t stream);
/* measured launch configuration: times the candidate tile shapes / loop structures (wgrad: split-K round counts) of one
 * convolution pass on the caller's own buffers, remembers the fastest for this geometry and leaves the pass's result in
 * `out`. mode 0 = fwd (a = x, b = w), 1 = dgrad (a = dy, b = w), 2 = wgrad (a = x, b = dy, out = dw). This is the ONE
 * entry point that synchronises `stream`. denet_conv_tuned reports the remembered choice (returns 1 if none).    */
int denet_conv_tune(int mode, const float* a, const float* b, const float* bias, const float* add, float* out,
                    float* workspace, size_t workspace_bytes, int N, int H, int W, int C, int K, int R, int S, int S_real,
                    int stride, int pad, int OH, int OW, hipStream_t stream);
int denet_conv_tuned(int mode, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad,
                     int* tile, int* nbuf, int* rounds);
/* persistence of the measured configurations: records of 14 ints (11 key fields + tile, nbuf, rounds). export returns the
 * number of entries (writes at most `capacity`); import adds / replaces entries; geometries present are not measured again */
/* a single wavefront that idles for `microseconds` on `stream`: lets a host find out whether two streams run concurrently
 * (different hardware queues) or were multiplexed onto one queue by the runtime                                      */
int denet_spin(int microseconds, hipStream_t stream);
int denet_tune_export(int* records, int capacity);
int denet_tune_import(const int* records, int count);
int denet_tune_clear(void);
int denet_conv_last_config(int* mode, int* bm, int* bn, int* nbuf, int* grid_y);
/* live timing of the igemm kernel alone (bench.py roofline leg): denet_conv_profile(1) starts recording one HIP event
 * pair per convolution launch on the launch stream, (0) stops and frees; _read returns the duration of launch i and the
 * instantiation <mode,BM,BN,2,2,NBUF> it used (the name rocprofv3 --kernel-trace shows). denet_conv_profile(2) records the
 * instantiations only - no events, no effect on timing or stream order (the per-layer "which kernel ran" audit of the parity
 * tests and of bench.py's kernels_used); _read then returns a duration of 0.                                     */
/* the fused F(4x4,3x3) product + output-transform kernel (csrc/wino4f.hip) inside the denet_conv_wino_* passes: 0 = never,
 * 32 / 64 = that tile block wherever the geometry allows (33 / 34: 32-tile blocks as 4-wave workgroups on 64 / 32 output channels, several per CU), -1 = the default
 * policy (DENET_WINO4F, DENET_WINO4F_TB); returns the previous setting. Same operator as the un-fused passes (convolution.py:80-83), other rounding.                       */
int denet_conv_wino4f_mode(int mode);
/* the same for the F(4x4) FILTER-gradient products inside denet_conv_wino_wgrad / _wgrad_dm (csrc/wino4g.hip: component products
 * on the contraction-major operands + the adjoint filter transform, model_cnn.py:318): 0 = never, 1 = wherever the geometry
 * allows, -1 = default (DENET_WINO4G); returns the previous setting.                                                     */
int denet_conv_wino4g_mode(int mode);
int denet_conv_profile(int enable);
int denet_conv_profile_count(void);
int denet_conv_profile_read(int i, float* ms, int* mode, int* bm, int* bn, int* nbuf);
int denet_conv_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int N,
                     int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW,
                     hipStream_t stream);

/* ---- batch norm (+ReLU, +residual)  (denet/layer/batch_norm.py:50-79 dnn_batch_normalization_train/test,
 *      denet/layer/batch_norm_relu.py:34-54 BatchNormReluOp + grad, denet/layer/resnet.py:109-113).
 *      x,y,res,dy,dx,dres: [M,C] (M = N*H*W).  run_mean/run_stdinv are updated in place (momentum form of
 *      batch_norm.py:75-76; the running statistic is the INVERSE standard deviation).                    */
/* In-launch second stage of a batch norm's reductions (csrc/bn_final.h). A convolution pass that writes partial column sums
 * (denet_conv_fwd_stats, denet_conv_wino_fwd_stats*, denet_conv_wino_fwd_fold, denet_conv_wino2f*, denet_conv_stem_fwd*: the
 * statistics of the batch norm BEHIND it; denet_conv_*dgrad*_sums / _fold, denet_conv_dgrad_1x1t: the backward sums of the batch norm
 * IN FRONT) can reduce them itself in its last workgroup, which takes denet_bn_stats_final / denet_bn_bwd_final off the stream.
 * The caller ARMS the batch norm (this thread, the next producing call only), runs the pass and DISARMS: disarm returns 1 if the
 * pass took the final over (the outputs are then written by the pass; the caller must not launch the separate final), 0 if the
 * kernel that ran cannot (the caller proceeds as before). Results are bit-identical to the separate launches.
 *   arm_stats: what denet_bn_stats_final writes - save_mean, save_invstd and (optional) the running statistics update
 *              (batch_norm.py:50-53, 75-76); arm_sums: what denet_bn_bwd_final writes - dgamma, dbeta, coef [2][C].
 *   counters:  `ncounters` zeroed unsigned ints owned by the caller, one per column group of the producing kernel (64 suffice),
 *              used by ONE pass at a time and left zero by it. Nothing is taken unless denet_bn_final_mode enables the kind. */
int denet_bn_final_arm_stats(long M, int C, float momentum, float eps, float* run_mean, float* run_stdinv, float* save_mean,
                             float* save_invstd, unsigned* counters, int ncounters);
int denet_bn_final_arm_sums(long M, int C, float* dgamma, float* dbeta, float* coef, unsigned* counters, int ncounters);
int denet_bn_final_disarm(void);
/* which reductions a pass may take over: bit 0 forward statistics, bit 1 backward sums; bits < 0 only queries. Returns the previous
 * setting. Default 0 (DENET_BN_FINAL_FOLD): measured slower than the separate launches on MI355X - kept as a tested option. */
int denet_bn_final_mode(int bits);
size_t denet_bn_workspace_bytes(long M, int C);
int denet_bn_fwd_train(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                       float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd, void* workspace,
                       long M, int C, float momentum, float eps, int relu, hipStream_t stream);
/* batch norm (training) whose statistics pass is replaced by the sums a convolution epilogue wrote: partial [rows][2][C] */
int denet_bn_fwd_train_pre(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                           float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd, const double* partial,
                           int rows, long M, int C, float momentum, float eps, int relu, hipStream_t stream);
/* the pieces of the training passes on their own, for consumers that evaluate the pointwise part themselves
 * (denet_conv_wino_fwd_fold / _dgrad_fold): statistics -> coefficients (+ running statistics), the pointwise forward pass
 * given the coefficients, the two reductions of the backward pass (dgamma, dbeta, coef [2][C] = mean(g), mean(g*xhat)) and its
 * pointwise part given coef. denet_bn_stats_final + denet_bn_apply == denet_bn_fwd_train_pre,
 * denet_bn_bwd_sums + denet_bn_bwd_apply == denet_bn_bwd (same kernels). */
int denet_bn_stats_final(const double* partial, int rows, long M, int C, float momentum, float eps, float* run_mean,
                         float* run_stdinv, float* save_mean, float* save_invstd, hipStream_t stream);
int denet_bn_apply(const float* x, const float* res, float* y, const float* gamma, const float* beta, const float* save_mean,
                   const float* save_invstd, long M, int C, int relu, hipStream_t stream);
int denet_bn_bwd_sums(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                      void* workspace, long M, int C, int relu, hipStream_t stream);
int denet_bn_bwd_final(const double* partial, int rows, long M, int C, float* dgamma, float* dbeta, float* coef,
                       hipStream_t stream);
int denet_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const float* coef, float* dx, float* dres, long M,
                       int C, int relu, hipStream_t stream);
int denet_bn_fwd_test(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                      const float* run_mean, const float* run_stdinv, float* coef, int coef_ready, long M, int C, float eps,
                      int relu, hipStream_t stream);
/* inference: conv(x, w) + b followed by test-mode batch norm == conv(x, w_out) + b_out (batch_norm.py:50-52 incl. its
 * double epsilon). w: [K][per_k] KRSC filters, conv_bias: [K] or NULL. */
int denet_bn_fold(const float* w, const float* conv_bias, const float* gamma, const float* beta, const float* run_mean,
                  const float* run_stdinv, float eps, float* w_out, float* b_out, int K, long per_k, hipStream_t stream);
/* relu mask of the backward: y > 0 if y is given; if y is NULL it is recomputed from x (needs beta) — the fused
 * residual blocks pass y, plain BNA layers pass NULL and save one pass over the activation                       */
int denet_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dgamma, float* dbeta,
                 void* workspace, long M, int C, int relu, hipStream_t stream);
/* A max pool directly behind a BN + ReLU layer (batch_norm_relu.py:34-48 -> pool.py:38, the ResNet stem), training: the
 * normalised tensor is never written. Forward: statistics from `partial` / `rows` (denet_conv_fwd_stats) or, partial = NULL,
 * measured here (workspace = denet_bn_workspace_bytes(N*H*W, C)); writes y_pool [N,OH,OW,C] and the argmax taps (first maximum
 * in scan order, as denet_maxpool_fwd). Backward: dy_pool + argmax -> dx [N,H,W,C], dgamma, dbeta; bit-identical to
 * denet_maxpool_bwd followed by denet_bn_bwd. */
int denet_bn_relu_pool_fwd_train(const float* x, float* y_pool, unsigned char* argmax, const float* gamma, const float* beta,
                                 float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd, const double* partial,
                                 int rows, void* workspace, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                                 float momentum, float eps, hipStream_t stream);
int denet_bn_relu_pool_bwd(const float* x, const float* dy_pool, const unsigned char* argmax, const float* gamma,
                           const float* beta, const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                           float* dbeta, void* workspace, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                           hipStream_t stream);
/* The same backward pass with its two reductions taken over the POOLED tensors (a quarter of the elements, no window gather):
 * every window sends its gradient to its argmax pixel, whose ReLU output is y_pool - so sum g = sum dy_pool * [y_pool > 0] and
 * sum g * xhat = sum dy_pool * [y_pool > 0] * xhat_pool with xhat_pool = (x - mean) * invstd at the argmax, written by
 * ..._fwd_train_xhat. ..._bwd_sums reduces them (zeros / ones: [C] constant vectors; coef [2][C] = the means over the N*H*W input
 * pixels); they may instead come from the data-gradient pass that wrote dy_pool (denet_conv_wino2f_sums with sums_of = {x:
 * xhat_pool, y: y_pool, mean: zeros, invstd: ones, relu: 1}) through denet_bn_bwd_final. ..._bwd_apply is the pointwise pass. */
int denet_bn_relu_pool_fwd_train_xhat(const float* x, float* y_pool, unsigned char* argmax, float* xhat_pool, const float* gamma,
                                      const float* beta, float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd,
                                      const double* partial, int rows, void* workspace, int N, int H, int W, int C, int OH, int OW,
                                      int k, int stride, int pad, float momentum, float eps, hipStream_t stream);
int denet_bn_relu_pool_bwd_sums(const float* xhat_pool, const float* y_pool, const float* dy_pool, const float* zeros,
                                const float* ones, float* dgamma, float* dbeta, float* coef, void* workspace, int N, int H, int W,
                                int C, int OH, int OW, hipStream_t stream);
int denet_bn_relu_pool_bwd_apply(const float* x, const float* dy_pool, const unsigned char* argmax, const float* gamma,
                                 const float* beta, const float* save_mean, const float* save_invstd, const float* coef, float* dx,
                                 int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad, hipStream_t stream);

/* ---- pooling  (denet/layer/pool.py:28-40 dnn_pool max / average_inc_pad;
 *      denet/layer/pool_inv_op.py:38-63 k_pool_inv, :144-169 k_pool_inv_grad)                             */
int denet_maxpool_fwd(const float* x, float* y, unsigned char* argmax, int N, int H, int W, int C, int OH, int OW,
                      int k, int stride, int pad, hipStream_t stream);
int denet_maxpool_bwd(const float* dy, const unsigned char* argmax, float* dx, int N, int H, int W, int C, int OH,
                      int OW, int k, int stride, int pad, hipStream_t stream);
int denet_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                      hipStream_t stream);
int denet_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, int k, int stride,
                      int pad, hipStream_t stream);
int denet_pool_inv_fwd(const float* x, float* y, int N, int H, int W, int C, int fy, int fx, hipStream_t stream);
int denet_pool_inv_bwd(const float* dy, float* dx, int N, int H, int W, int C, int fy, int fx, hipStream_t stream);

/* ---- device-side rendering of the data pipeline's augmentation plan (csrc/image.hip)
 *      Replaces the pixel work of denet/dataset/augment.py (add_border :51-61, crop, scale :21-47 = Pillow thumbnail /
 *      resize, image_to_array :9-17, photometric :271-285, colorspace :288-293) and of
 *      denet/dataset/image_loader.py:71-105 (mean/std, mirror); the random decisions stay on the host
 *      (denet_amd/dataset/augment.py plan_*). Resampling is Pillow's two-pass fixed-point convolution
 *      (src/libImaging/Resample.c, Pillow 12.2.0), bit for bit: coefficient tables from the host, integer passes on
 *      the device. Images are RGBX u8 between the steps; the result is one fp32 NHWC image of the training batch.   */
int denet_host_resample_coeffs(int in_size, double in0, double in1, int out_size, int filter, int* bounds_host,
                               int* kk_host, long kk_capacity);
int denet_image_crop(const unsigned char* src, unsigned char* dst_rgbx, int sw, int sh, int src_bpp, int px, int py, int x0,
                     int y0, int w, int h, hipStream_t stream);
int denet_image_reduce(const unsigned char* in_rgbx, unsigned char* out_rgbx, int in_w, int in_h, int fx, int fy,
                       hipStream_t stream);
int denet_image_resample_pass(const unsigned char* in_rgbx, unsigned char* out_rgbx, int in_w, int in_h, int out_n,
                              int horizontal, const int* bounds_dev, const int* kk_dev, int ksize, hipStream_t stream);
int denet_image_render_batch(int B, const unsigned char* const* src_host, const int* src_wh, const int* op_off, const int* ops,
                             const int* ops_wh, const double* ops_in1, const int* photo_n, const int* photo_ops,
                             const double* photo_alpha, const double* noise, const unsigned char* has_noise,
                             const float* mean_std, const unsigned char* mirror, int crop, int cp, float* out_dev,
                             unsigned char* pinned_host, size_t pinned_bytes, unsigned char* staging_dev,
                             unsigned char* scratch0, unsigned char* scratch1, size_t scratch_bytes,
                             unsigned long long* sums_dev, hipStream_t stream);
int denet_image_finish(const unsigned char* img_rgbx, float* out_nhwc, int w, int h, int cp, int n_ops,
                       const int* ops_host, const double* alphas_host, const double* noise_host,
                       const float* mean_std_host, int mirror, unsigned long long* sums_ws, hipStream_t stream);

/* ---- shape / stochastic layers of the operator surface (csrc/augment.hip), NHWC fp32, C % 4 == 0
 *      B    zero border, (left, right, top, bottom)              denet/layer/border.py:18-33
 *      CM   per-image random crop + column mirror + row flip     denet/layer/crop_mirror.py:26-56 (train=0: centre
 *           crop, no mirror / flip)
 *      D    dropout, y = x * mask / (1 - rate)                   denet/layer/dropout.py:20-24 (call again with dy
 *           and the same seed for the gradient)
 *      SKIP "concat" combine mode: channel concatenation         denet/layer/skip.py:93-96
 *      DC   bias epilogue of the deconvolution                   denet/layer/deconvolution.py:66-67
 *    The reference's random stream is Theano's MRG_RandomStreams (third party, not reproducible here); the masks
 *    and crop geometry are a pure function of `seed` and the LOGICAL element / image index instead (splitmix64
 *    finaliser, restated in oracle/layers.py), so nothing is stored between the forward and backward pass.  */
int denet_border_fwd(const float* x, float* y, int N, int H, int W, int C, int left, int right, int top, int bottom,
                     hipStream_t stream);
int denet_border_bwd(const float* dy, float* dx, int N, int H, int W, int C, int left, int right, int top, int bottom,
                     hipStream_t stream);
int denet_crop_mirror_fwd(const float* x, float* y, int N, int H, int W, int C, int crop_h, int crop_w, float mirror_pr,
                          float flip_pr, int train, uint64_t seed, hipStream_t stream);
int denet_crop_mirror_bwd(const float* dy, float* dx, int N, int H, int W, int C, int crop_h, int crop_w,
                          float mirror_pr, float flip_pr, int train, uint64_t seed, hipStream_t stream);
int denet_dropout(const float* x, float* y, int N, int HW, int C, int C_logical, float rate, uint64_t seed,
                  hipStream_t stream);
int denet_concat_fwd(const float* a, const float* b, float* y, long rows, int CA, int CAP, int CB, int CBP, int CYP,
                     hipStream_t stream);
int denet_concat_bwd(const float* dy, float* da, float* db, long rows, int CA, int CAP, int CB, int CBP, int CYP,
                     hipStream_t stream);
int denet_add_bias(const float* x, const float* bias, float* y, long rows, int C, hipStream_t stream);

/* ---- element-wise / boundary helpers
 *      layout conversion of the NCHW batches handed to ModelCNN.train_step (denet/model/model_cnn.py:407),
 *      residual / skip add (denet/layer/skip.py:81-86), `A` relu (denet/layer/activation.py:31-34),
 *      conv-bias gradient, solver update (denet/model/model_cnn.py:282-294, 321-331).                    */
int denet_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t stream);
int denet_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t stream);
int denet_add(const float* a, const float* b, float* y, long n, int relu, hipStream_t stream);
int denet_relu_fwd(const float* x, float* y, long n, hipStream_t stream);
int denet_relu_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream);
size_t denet_colsum_workspace_bytes(long M, int C);
int denet_colsum(const float* x, float* out, void* workspace, long M, int C, hipStream_t stream);
int denet_solver_step(float* params, float* moments, const float* grads, long n, long n_decay, float lr,
                      float momentum, int iteration, float decay, float grad_scale, int mode, hipStream_t stream);
/* adam (denet/model/model_cnn.py:296-305): first / second moment buffers m, v; momentum = (beta1, beta2); the bias
 * corrections 1/(1-beta^(iteration+1)) are evaluated on the host in double; eps = 1e-8                         */
int denet_solver_adam(float* params, float* m, float* v, const float* grads, long n, long n_decay, float lr,
                      float beta1, float beta2, int iteration, float decay, float grad_scale, hipStream_t stream);
int denet_scale(float* x, long n, float s, hipStream_t stream);

/* ---- DeNet corner map  (denet/layer/denet_corner.py:50-53 log_softmax([x,-x]); :126-134 cost)
 *      conv:[B,H,W,CP] (first Cn channels are corner logits)  corner_pr/target:[B,2,Cn,H,W] (the reference's
 *      own layout, it is what build_samples consumes).  cost[0] receives cost_factor*corner_cost.         */
int denet_corner_fwd(const float* conv, float* corner_pr, int B, int H, int W, int CP, int Cn, hipStream_t stream);
size_t denet_loss_workspace_bytes(void);
int denet_corner_loss(const float* corner_pr, const float* target, float* dconv, float* cost, void* workspace, int B,
                      int H, int W, int CP, int Cn, float cost_factor, hipStream_t stream);

/* ---- sparse RoI feature sampling  (denet/layer/denet_sparse_op.py:42-85 k_sparse_sample<gs>, :171-212
 *      k_sparse_sample_grad<gs>; Theano fallback denet/layer/denet_sparse.py:70-96)
 *      fmap:[B,H,W,CP] channels [coff,coff+F) are sampled; bbox:[B*rois,4] normalised x0,y0,x1,y1;
 *      out:[B*rois,KP] = gs*gs*F features, box height, box width, zero padding; taps:[B*rois,gs*gs] receives
 *      the sampled cell index ys*W+xs (bit-exact parity surface).  The gradient is a deterministic segmented
 *      sum (the taps of every image grouped by cell with a stable counting sort, summed in ascending
 *      (roi, tap) order) instead of the reference's atomicAdd scatter.                                     */
int denet_sparse_fwd(const float* fmap, const float* bbox, float* out, int* taps, int B, int H, int W, int CP,
                     int coff, int F, int rois_per_image, int gs, int KP, int tap_rule, hipStream_t stream);
/* denet_sparse_sort: the grouping of the tap list by cell alone (it depends only on `taps`; a caller may queue it on a
 * side stream during the forward pass); denet_sparse_bwd with taps == NULL then consumes the lists. sort_ws holds
 * denet_sparse_sort_workspace_bytes(...) bytes; H*W <= 32768 cells.                                               */
size_t denet_sparse_sort_workspace_bytes(int B, int H, int W, int rois_per_image, int gs);
/* 1: denet_sparse_sort serves this problem with ONE kernel (H*W <= 4096 cells, <= 65535 taps per image, DENET_SORT_ONE_KERNEL
 * not 0) - the host keeps it on the compute stream; 0: the three-kernel form, which a host may queue on a side stream */
int denet_sparse_sort_is_single(int B, int H, int W, int rois_per_image, int gs);
int denet_sparse_sort(const int* taps, void* sort_ws, size_t sort_ws_bytes, int B, int H, int W, int rois_per_image,
                      int gs, hipStream_t stream);
int denet_sparse_bwd(const float* dy, const int* taps, void* sort_ws, size_t sort_ws_bytes, float* dfmap, int B, int H,
                     int W, int CP, int coff, int F, int rois_per_image, int gs, int KP, int zero_from,
                     hipStream_t stream);

/* ---- detection cost  (denet/layer/denet_detect.py:238-313 get_errors/cost; theano_util.py:27-34)
 *      logits:[M,CP] (ncls class logits, nreg box regressors, nfit independent-fitness logits); det_target:[M,ncls];
 *      bbox_valid:[M]; bbox_target:[M,8] = target cx,cy,w,h, sample cx,cy,w,h; fit_target:[M,nfit] or NULL (:103-108,
 *      :298-301); costs[0] = DET cost, costs[1] = BBOX cost + independent-fitness cost.                      */
int denet_detect_loss(const float* logits, const float* det_target, const float* bbox_valid, const float* bbox_target,
                      const float* roi_bbox, const float* fit_target, float* dlogits, float* costs, void* workspace, int M,
                      int batch, int CP, int ncls, int nreg, int nfit, float cost_factor, float bbox_factor,
                      float fit_factor, int bounded_iou, hipStream_t stream);

/* ---- inference tail  (denet/layer/denet_detect.py:76-100 class log-softmax + box decoding, :330-349
 *      joint-fitness marginalisation; denet/layer/denet_detect.cc:99-173 build_detections_nms, :73-97 hard NMS,
 *      :35-71 Gaussian soft-NMS).  logits:[M,CP]; det_pr/fitness:[M,class_num+1] log domain; bbox:[M,4];
 *      count:[B] valid RoIs per image; keep:[B,class_num,S] bytes (1 = surviving detection). The soft-NMS variant is
 *      sequential in its selections: denet_soft_nms_batch runs one wave per (class, image) on the device, the *_host
 *      entries are the same method on host copies (tests, single classes).                                    */
int denet_detect_decode(const float* logits, const float* roi_bbox, float* det_pr, float* fitness, float* bbox, int M,
                        int CP, int class_num, int jointfit, int nreg, int nfit /* independent-fitness logits, :396-401 */,
                        float overlap_threshold, hipStream_t stream);
int denet_detect_nms(const float* det_pr, const float* fitness, const float* bbox, const int* count,
                     unsigned char* keep, int B, int S, int class_num, float pr_threshold, float nms_threshold,
                     hipStream_t stream);
int denet_soft_nms_host(const float* score_host, const float* box_host, int n, float nms_threshold,
                        int* out_order_host, float* out_score_host, int* out_n_host);
long denet_soft_nms_batch_host(const float* det_pr, const float* fitness, const float* bbox, const int* counts, int B, int S,
                               int class_num, float pr_threshold, float nms_threshold, float* out_score, int* out_cls,
                               int* out_row, int* out_count, long capacity);
/* the same tail on the device, bit-identical to denet_soft_nms_batch_host: a wave per (class, image) runs the sequential
 * selection (arg-max and rescoring parallel over the candidates); all pointers device memory, out_* sized B*class_num*S,
 * *out_total = number of detections, out_count[b] per image; output order = (image, class, selection order), :153-160 */
size_t denet_soft_nms_workspace_bytes(int B, int S, int class_num);
int denet_soft_nms_batch(const float* det_pr, const float* fitness, const float* bbox, const int* count, int B, int S,
                         int class_num, float pr_threshold, float nms_threshold, float* out_score, int* out_cls, int* out_row,
                         int* out_count, int* out_total, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- corner selection + RoI proposal  (denet/layer/denet_sparse.cc:489-557 run_build_samples, :321-471
 *      search_corners, :271-308 get_sample; Python-facing wrapper build_samples :559-668, which the reference
 *      calls on the HOST with a D2H copy of the corner map, denet/layer/denet_sparse.py:129-139).
 *      corner_pr:[B,2,Cn,H,W] device, Cn = 4 (TL,TR,BL,BR) or 5 (+ centre, DNC.C). Per image the best `sample_count` candidate boxes, ranked exactly like
 *      the reference (score descending == |pr_f-pr_t| ascending; equal scores ordered by generation index):
 *        out_box:[B,sample_count,4] int32 corner cells x0,y0,x1,y1;  out_absd:[B,sample_count] fp32 |pr_f-pr_t|;
 *        out_count:[B].
 *      denet_samples_finish_host converts HOST copies of these into the reference's tuples
 *      (pr, x0/W, y0/H, (x1+1)/W, (y1+1)/H) with the reference's own host arithmetic (denet_sparse.cc:306-307);
 *      samples_host:[B,sample_count,5].  cluster_threshold < 1: ask for sample_count = 10 * sample_num^2 (up to 61 440; beyond 7936 the final sort runs through a global buffer) and
 *      pass each image with more than sample_num^2 candidates through denet_host_cluster_samples (apply_cluster,
 *      denet_sparse.cc:165-242, host code in the reference as well): out_host [output_num][5], out_count rows.   */
size_t denet_build_samples_workspace_bytes(int B, int Cn, int H, int W, int max_corners, int sample_count);
int denet_build_samples(const float* corner_pr, int* out_box, float* out_absd, int* out_count, void* workspace,
                        size_t workspace_bytes, int B, int Cn, int H, int W, float corner_threshold, int sample_count,
                        int max_corners, int local_max, hipStream_t stream);
/* The training-time RoI list editing (DeNetSparseLayer.get_target, denet/layer/denet_sparse.py:184-201) on the device, for the
 * batches in which no image proposes more than n_keep RoIs (no random.sample; the host checks the counts it has copied): proposals
 * (denet_samples_finish_host's box arithmetic), random boxes from generator outputs drawn ahead on the host
 * (denet_host_mt_prefetch; 8 per box, their position follows from the counts), ground truth in the last slots. box [B][S][4]
 * int32 and count [B] are denet_build_samples' device outputs, H x W the corner map, mt_out [n_out] the uploaded outputs, cursor0
 * the number already used, gt [n][4] doubles + gt_off [B+1]. out_bbox [B][S][4] floats = what build_bbox_array would upload, bit
 * for bit what denet_host_edit_samples(_stream) writes to out_box_f32 - the host runs that later, beside the device's gather and
 * head, for the Python-side list. status [2] (device, zeroed here): [0] != 0: not this call's case, out_bbox incomplete;
 * [1]: outputs consumed. */
int denet_edit_samples_device(const int* box, const int* count, int H, int W, const uint32_t* mt_out, long n_out, long cursor0,
                              const double* gt, const int* gt_off, int sample_gt, int B, int S, int n_keep, float* out_bbox,
                              int* status, hipStream_t stream);
/* diagnostics of the LAST denet_build_samples call on `workspace` (same geometry): corners kept per (image, type) after the
 * max_corners truncation (denet_sparse.cc:526-530) -> ncorner_out [B*Cn] int32, candidate boxes generated per image by the
 * pair search (:337-373) -> candidates_out [B] uint32; device buffers, copies on `stream`. */
int denet_build_samples_stats(const void* workspace, size_t workspace_bytes, int B, int Cn, int H, int W, int max_corners,
                              int sample_count, int* ncorner_out, unsigned* candidates_out, hipStream_t stream);
int denet_host_cluster_samples(const float* samples_host, int n, float threshold, int output_num, float* out_host,
                               int* out_count);
int denet_samples_finish_host(const int* box_host, const float* absd_host, const int* count_host, int B,
                              int sample_count, int H, int W, float* samples_host);

#ifdef __cplusplus
}
#endif
#endif /* DENET_HIP_H */
