/*
 * contrastyou_hip.h -- C ABI of libcontrastyou_hip.so, the MI355X (gfx950)
 * implementation of the Contrast-You SemiSupervisedEpocher hot path.
 *
 * The reference (jizongFox/Contrast-You) is 100 % Python on torch: it has no
 * FFI of its own.  Every entry point below therefore replaces a torch library
 * call made by a reference leaf; the file:line of that call site is cited per
 * function.  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the parameter name starts with h_ (host);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises, nothing allocates (callers pass workspaces);
 *   - activations are NHWC ("channels_last"): element (n,h,w,c) lives at
 *     ((n*H + h)*W + w)*ld + c, ld >= C the pixel pitch in elements;
 *   - dtype codes: CY_F32 (verification mode, f32 MFMA, exact fmaf chains)
 *     CY_BF16 (production mode, bf16 MFMA with f32 accumulation) and CY_F16
 *     (the reference's fp16 autocast mode, f16 MFMA with f32 accumulation);
 *   - return value 0 = enqueued, <0 = argument/shape error (nothing was
 *     launched), see CY_ERR_*.
 */
#ifndef CONTRASTYOU_HIP_H
#define CONTRASTYOU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CY_OK 0
#define CY_ERR_ARG (-1)
#define CY_ERR_SHAPE (-2)
#define CY_ERR_DTYPE (-3)
#define CY_ERR_LAUNCH (-4)
#define CY_ERR_WORKSPACE (-5)

#define CY_F32 0
#define CY_BF16 1
#define CY_F16 2 /* IEEE half: the reference's own autocast dtype (contrastyou/amp/amp.py:13-45); same kernels as
                  * CY_BF16 with mfma_f32_32x32x16_f16, f32 accumulation; needs loss scaling (torch GradScaler) */

/* how source 1 of a 3x3 conv is addressed (arch/unet.py:67-70 MaxPool2d,
 * :38 Upsample(scale_factor=2) nearest, both folded into the conv's loads) */
#define CY_SRC_DIRECT 0 /* src1 is [N,H,W,C1] */
#define CY_SRC_POOL2 1  /* src1 is [N,2H,2W,C1]; 2x2 max taken on load */
#define CY_SRC_UP2 2    /* src1 is [N,H/2,W/2,C1]; nearest x2 on load */

/* ABI version; bumped whenever a signature changes. */
int cy_abi_version(void);
/* name of the offload arch the library was built for ("gfx950"). */
const char* cy_build_arch(void);
/* id of the HIP-graph capture `stream` currently belongs to, 0 when it is not capturing.  Host-side
 * only (hipStreamGetCaptureInfo): lets the binding tell apart events recorded in different captures
 * (an event recorded inside one capture must not be waited on inside another). */
unsigned long long cy_stream_capture_id(void* stream);
/* profiling aid: a one-thread kernel on `stream` stores the device wall clock (100 MHz ticks) into
 * buf[slot] (device memory); works inside a stream capture. */
int cy_debug_stamp(unsigned long long* buf, int slot, void* stream);
/* measurement aid: a one-lane kernel that holds `stream` for `micros` (<= 50000) microseconds of the device wall
 * clock, so that a host can enqueue work behind it and time back-to-back GPU execution with events. */
int cy_debug_spin(int micros, void* stream);
/* measurement aid: timing-only HIP events (hipEventDisableSystemFence: recording one does not write back and
 * invalidate the L2s, which a default event does -- inside the interval it measures).  elapsed_us synchronises on
 * e1 first. */
int cy_debug_event_create(void** ev);
int cy_debug_event_record(void* ev, void* stream);
int cy_debug_event_elapsed_us(void* e0, void* e1, float* us);
int cy_debug_event_destroy(void* ev);
/* Hand-over from a captured HIP graph to a stream outside it (the data-parallel optimizer starts the all-reduce of
 * the gradient buckets whose layers are done while the rest of the backward graph runs): `stream` waits until the
 * int32 counter in device memory, which a kernel of the graph increments, is >= at_least. */
int cy_stream_wait_value(void* stream, const int* counter, int at_least);

/* ------------------------------------------------------------------------
 * 3x3 convolution, stride 1, pad 1, no bias  (nn.Conv2d at
 * contrastyou/arch/unet.py:21,24,39) as an implicit GEMM on MFMA.
 *
 * Geometry: output [N,H,W,Cout].  Logical input = channel concat of source 1
 * (C1 channels, addressed per `mode1`) and optional source 2 (C2 channels,
 * always direct) -- torch.cat((skip, up), dim=1) at arch/unet.py:142,151,160,
 * 169 without the copy.  If `prologue` != 0 source 1 is a RAW conv output of
 * the previous layer and relu(scale[c]*x + shift[c]) (training-mode
 * BatchNorm2d + ReLU, arch/unet.py:22-23) is applied while loading it.
 * ------------------------------------------------------------------------ */
typedef struct cy_conv_desc {
  int32_t N, H, W;      /* output spatial geometry */
  int32_t C1, C2, Cout; /* channels of source 1, source 2 (0 = none), output */
  int32_t mode1;        /* CY_SRC_* for source 1 */
  int32_t prologue;     /* 1: relu(scale*x+shift) on source 1 while loading */
  int32_t in_dtype;     /* dtype of sources and packed weights */
  int32_t out_dtype;    /* dtype of the output tensor */
  int32_t ld1, ld2;     /* pixel pitch (elements) of source 1 / 2 */
  int32_t ldo;          /* pixel pitch of `out` */
  int32_t split_c;      /* >0: couts >= split_c go to out2 (pitch ldo2) */
  int32_t ldo2;
} cy_conv_desc;

/* Packed weight geometry for a (Cout,Cin) 3x3 kernel: [9][co_pad][ci_pad]. */
int cy_conv3x3_packed_dims(int Cout, int Cin, int* co_pad, int* ci_pad);

/* Elements of the packed image of a (Cout, Cin) kernel in `dtype`: the [9][co_pad][ci_pad] image, followed for
 * 16-bit dtypes with Cout % 64 == 0 and Cin % 16 == 0 by the stage-contiguous image of the LDS-DMA kernel
 * (csrc/cy_conv_flow.h): [Cout/64][Cin/16][tap][2][64 couts][8 channels] = 9*Cout*Cin elements. */
long long cy_conv3x3_packed_elems(int Cout, int Cin, int dtype);

/* Repack reference-layout weights w[Cout][Cin][3][3] (f32) into the forward
 * image wf (cy_conv3x3_packed_elems(Cout, Cin, dtype) elements: [tap][co_pad][ci_pad], then the stage-contiguous
 * image) and (if wd != NULL) the data-gradient image wd (cy_conv3x3_packed_elems(Cin, Cout, dtype) elements:
 * wd[tap][ci_pad'][co_pad'] = w[co][ci][2-kh][2-kw], then its stage-contiguous image), both of `dtype`. */
int cy_conv3x3_pack_weights(const float* w, void* wf, void* wd, int Cout, int Cin, int dtype,
                            void* stream);

/* All 3x3 weights of a network in ONE launch (the per-layer call above costs a launch per layer and
 * step: the weights change with every optimizer step).  `items` is a DEVICE array describing the
 * layers: source pointer, geometry, the element offsets of the layer's forward / data-gradient image
 * inside the two output arenas, and `first` = index of the layer's first work unit in the flattened
 * range [0, total).  A work unit is a 32 (co) x 32 (ci) tile; a layer has
 * ceil(max(co_pad, co_pad2) / 32) * ceil(max(ci_pad, ci_pad2) / 32) of them, co-major.
 * Offsets instead of pointers keep the table valid when the arenas are re-allocated per step. */
typedef struct cy_pack_item {
  const float* w;          /* [Cout][Cin][3][3] f32 */
  long long off_f, off_d;  /* element offsets into wf_arena / wd_arena */
  long long first;
  int Cout, Cin;
  int co_pad, ci_pad;      /* forward image  [9][co_pad][ci_pad]   (cy_conv3x3_packed_dims(Cout, Cin)) */
  int ci_pad2, co_pad2;    /* dgrad image    [9][ci_pad2][co_pad2] (cy_conv3x3_packed_dims(Cin, Cout)) */
  long long off_ff, off_fd; /* element offsets of the stage-contiguous images in the same arenas, or -1 (none) */
} cy_pack_item;
int cy_conv3x3_pack_weights_batched(const cy_pack_item* items, int n_items, long long total,
                                    void* wf_arena, void* wd_arena, int dtype, void* stream);

/* Number of per-tile statistic partials the forward kernel writes for `d`
 * (stats buffer is float[num_partials][2][Cout]: sum, sum of squares). */
int cy_conv3x3_num_partials(const cy_conv_desc* d);

/* The launch plan the library derives for `d`: which kernel instantiation cy_conv3x3_fwd will run.
 * Parity tests assert on it, so that every instantiation that appears in profiles/ is known to be
 * covered by a test that provably took it (the plan depends on N, H, W, channel counts and dtype). */
typedef struct cy_conv_plan {
  int32_t kernel;     /* 0: conv3x3_igemm_kernel, 1: conv3x3_plane_kernel, 4: conv3x3_stream_kernel, 5: conv3x3_flow_kernel */
  int32_t th, tw, bn; /* output tile (rows x columns) and output channels per workgroup */
  int32_t ksplit;     /* >1: split-K over input-channel chunks + conv_splitk_finish_kernel */
  int32_t one_per_cu; /* plane kernel, 128 couts: the one-workgroup-per-CU build with halo prefetch */
  int32_t partials;   /* = cy_conv3x3_num_partials(d) */
  int32_t workgroups; /* grid size of the conv launch */
} cy_conv_plan;
int cy_conv3x3_plan(const cy_conv_desc* d, cy_conv_plan* plan);

/* out = conv3x3(concat(src1', src2), w).  stats may be NULL.  Layers with too few
 * output tiles to fill the chip run split-K over input-channel chunks and need a
 * workspace of cy_conv3x3_fwd_ws_bytes(d) bytes (0 for the others; ws may then be NULL). */
size_t cy_conv3x3_fwd_ws_bytes(const cy_conv_desc* d);
int cy_conv3x3_fwd(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                   const float* shift, const void* w_packed, void* out, void* out2, float* stats,
                   void* ws, size_t ws_bytes, void* stream);

/* development aid: workgroup 0 of every following streaming-kernel launch (library built with -DCY_STREAM_STAMPS)
 * records shader-clock stamps of its waves into dev_buf[8][128] (DEVICE memory; NULL switches it off again). */
int cy_debug_conv_stamps(unsigned long long* dev_buf);
/* development aid: per-wave phase totals (shader clocks) of the twelve-wave weight-gradient kernel's tile loop
 * (library built with -DCY_WGRAD_STAMPS), [12 waves][8] = {request, mfma loop, commit, barrier, tiles}. */
int cy_debug_wgrad_stamps(unsigned long long* dev_buf);

/* Weight gradient dw[Cout][Cin][3][3] (f32, reference layout) =
 *   sum_p dy[p][co] * in[p+tap][ci]  with `in` addressed exactly as in
 * cy_conv3x3_fwd (same desc; desc.out_dtype/ldo describe dy).  ws is a
 * workspace of at least cy_conv3x3_wgrad_ws_bytes(d) bytes.  accumulate != 0:
 * dw += result (autograd's gradient accumulation folded into the reduction). */
size_t cy_conv3x3_wgrad_ws_bytes(const cy_conv_desc* d);
int cy_conv3x3_wgrad(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                     const float* shift, const void* dy, float* dw, int accumulate, void* ws,
                     size_t ws_bytes, void* stream);

/* Launch plan of the weight gradient for `d` (+ n_b images of a second segment, 0 for none). */
typedef struct cy_wgrad_plan {
  int32_t twelve;        /* 2: wgrad12s_kernel (wave-specialised, 64x64 blocks), 1: wgrad12_kernel, 0: wgrad_kernel */
  int32_t wco, wci, wk;  /* wave layout: 32x32 (co,ci) blocks per workgroup and pixel splits inside it */
  int32_t th, tw;        /* spatial tile */
  int32_t splits;        /* pixel splits = f32 slabs summed by wgrad_reduce_kernel */
  int32_t workgroups;
} cy_wgrad_plan;
int cy_conv3x3_wgrad_plan(const cy_conv_desc* d, int n_b, cy_wgrad_plan* plan);

/* The same layer's weight gradient over TWO batches in one launch: segment a is described by `d`
 * (d->N images), segment b has n_b images of the same geometry with its own tensors and -- if
 * d->prologue -- its own BN coefficients.  dw (+)= dw_a + dw_b (fixed summation order).  This is what
 * the two network passes of SemiSupervisedEpocher's two-stage step (epocher.py:348-360) contribute to
 * one weight; one launch instead of two halves the slab traffic and the launch count of the encoder.
 * bf16 only (CY_ERR_DTYPE otherwise). */
size_t cy_conv3x3_wgrad_pair_ws_bytes(const cy_conv_desc* d, int n_b);
int cy_conv3x3_wgrad_pair(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                          const float* shift, const void* dy, int n_b, const void* src1_b,
                          const void* src2_b, const float* scale_b, const float* shift_b,
                          const void* dy_b, float* dw, int accumulate, void* ws, size_t ws_bytes,
                          void* stream);

/* ---- BatchNorm sums without finalize launches (ABI v11; csrc/cy_bn_acc.h) -------------------------------------
 * A training-mode nn.BatchNorm2d (arch/unet.py:22,25,40) needs per-channel sums over the whole batch.  Instead of
 * per-workgroup partial rows + a finalize launch per layer (cy_bn_finalize / cy_bn_bwd_finalize below, still there),
 * the producing kernels ADD their partials into an accumulator -- 64-bit integer atomics on a two-limb fixed-point
 * split, so the result does not depend on the order of arrival -- and the consuming kernels derive the coefficients
 * themselves; the consumer's first workgroup leaves them in memory (`coef`) for the backward pass.
 * Accumulator: int64 words [R][4][C] ({sum hi, lo, sum of squares hi, lo} per replica) + R flag words; the caller
 * zeroes it before the producing launch.  R = cy_bn_acc_replicas(C, workgroups adding into a channel). */
typedef struct cy_bn_acc {
  void* acc;
  int32_t R, C;
} cy_bn_acc;
typedef struct cy_bn_fold {
  const void* acc;     /* the accumulator the previous conv filled */
  int32_t R, C;
  const float* gamma;  /* [C] or NULL (1) */
  const float* beta;   /* [C] or NULL (0) */
  double count;        /* N*H*W */
  float eps;
  int32_t reserved;
  float* coef;         /* [5][C] f32: scale, shift, mean, invstd, unbiased variance (for cy_bn_running_update) */
} cy_bn_fold;
int cy_bn_acc_replicas(int C, int workgroups);
size_t cy_bn_acc_bytes(int C, int R);
/* workgroups of cy_conv3x3_fwd(d) / cy_conv3x3_first_fwd that add into one channel's sums (for cy_bn_acc_replicas) */
int cy_conv3x3_stat_workgroups(const cy_conv_desc* d);
/* cy_conv3x3_fwd with either side of the convolution on accumulators: `in_fold` != NULL (and d->prologue): the
 * BN+ReLU coefficients of source 1 come from the previous layer's accumulator (kernels that cannot fold in place
 * run cy_bn_fold first); `out_acc` != NULL: the statistics of the output are added into it. */
int cy_conv3x3_fwd_bn(const cy_conv_desc* d, const void* src1, const void* src2, const cy_bn_fold* in_fold,
                      const float* scale, const float* shift, const void* w_packed, void* out, void* out2,
                      float* stats, const cy_bn_acc* out_acc, void* ws, size_t ws_bytes, void* stream);
int cy_conv3x3_first_fwd_acc(const float* x, const float* w, void* out, const cy_bn_acc* out_acc, int N, int Cin,
                             int H, int W, int Cout, int out_dtype, void* stream);
/* accumulator -> f->coef as its own launch (what a folding consumer does in its first workgroup) */
int cy_bn_fold_coef(const cy_bn_fold* f, void* stream);
/* out = relu(scale*y + shift) (+ 2x2 max) with the coefficients taken from the accumulator */
int cy_bn_relu_apply_fold(const void* y, const cy_bn_fold* f, void* out, long npix, int y_dtype, int out_dtype,
                          void* stream);
int cy_bn_relu_apply_pool_fold(const void* y, const cy_bn_fold* f, void* out, void* pooled, int N, int H, int W,
                               int y_dtype, int out_dtype, void* stream);
/* running statistics of up to 32 BatchNorm layers in ONE launch: running = (1 - momentum) * running + momentum *
 * batch moment, from the `coef` arrays their consumers left (rows 2 and 4).  `h_items` is a HOST array: the
 * pointers travel as kernel arguments (capturable, no device table). */
typedef struct cy_bn_run_item {
  const float* coef;
  float* running_mean;
  float* running_var;
  int32_t C;
  float momentum;
} cy_bn_run_item;
int cy_bn_running_update(const cy_bn_run_item* h_items, int n, void* stream);
/* backward: the sums of dz and dz*xhat into an accumulator (as cy_bn_relu_bwd_reduce / cy_maxpool2_bwd_bn write
 * partial rows); coef = the forward pass's [5][C] */
int cy_bn_relu_bwd_reduce_acc(const void* da, int ld_da, const void* y, const float* coef, const cy_bn_acc* acc,
                              long npix, int C, int dtype, void* stream);
int cy_bn_relu_bwd_workgroups(long npix, int C);
int cy_maxpool2_bwd_bn_acc(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                           const float* coef, const cy_bn_acc* acc, int N, int H, int W, int C, int dtype,
                           void* stream);
/* cy_upsample2_bwd whose output is the dA of relu(bn(y)) (the last BatchNorm of the block the Upsample read): that
 * layer's backward sums are added into `acc` from the values in registers (what cy_maxpool2_bwd_bn_acc does for the
 * encoder).  (H, W) are the low-resolution dims; y is [N,H,W,C].  cy_upsample2_bwd_bn_workgroups: the adders per
 * channel (for cy_bn_acc_replicas), or CY_ERR_SHAPE where the form does not apply (256 % (C/8) != 0). */
int cy_upsample2_bwd_bn_workgroups(int N, int H, int W, int C);
int cy_upsample2_bwd_bn_acc(const void* dup, int ld_dup, void* dx, const void* y, const float* coef,
                            const cy_bn_acc* acc, int N, int H, int W, int C, int dtype, void* stream);
/* dy = scale*dz + k1*y + k0 with (k1, k0) derived from the accumulator; the first workgroup adds (accumulate != 0) or
 * stores dgamma / dbeta (either may be NULL) */
int cy_bn_relu_bwd_apply_fold(const void* da, int ld_da, const void* y, const float* coef, const cy_bn_acc* acc,
                              double count, int batch_stats, float* dgamma, float* dbeta, int accumulate, void* dy,
                              long npix, int C, int dtype, void* stream);

/* The data gradient of a 3x3 conv with the backward of the BatchNorm + ReLU behind that conv in its load path
 * (autograd of arch/unet.py:21-23 in one launch instead of cy_bn_relu_bwd_apply + cy_conv3x3_fwd):
 *   dy = scale * dA * [scale*y + shift > 0] + k1*y + k0     (formed in LDS from dA and y, (k1, k0) from `acc`)
 *   out = conv3x3(dy, w_packed)                             (w_packed = the data-gradient image of the layer)
 * and dy is also written to bn->dy for the weight gradient.  d describes the data-gradient convolution with
 * d->prologue == 2: C1 = the forward layer's output channels, C2 = 0, direct source; dA, y and dy share the pitch
 * d->ld1.  The first workgroup adds (accumulate) or stores dgamma / dbeta.  Only launch plans with room for the second
 * halo buffer take this path: ask cy_conv3x3_dgrad_bn_ok(d) (1 / 0) and fall back to the two launches otherwise. */
typedef struct cy_bn_bwd_in {
  const void* y;        /* the forward layer's raw conv output [N,H,W,C1] */
  const float* coef;    /* the forward pass's [5][C1] */
  const cy_bn_acc* acc; /* sums of dz and dz*xhat (cy_bn_relu_bwd_reduce_acc / cy_maxpool2_bwd_bn_acc) */
  double count;         /* N*H*W */
  int32_t batch_stats, accumulate;
  float* dgamma;        /* [C1] or NULL */
  float* dbeta;
  void* dy;             /* out: [N,H,W,C1] */
} cy_bn_bwd_in;
int cy_conv3x3_dgrad_bn_ok(const cy_conv_desc* d);
int cy_conv3x3_dgrad_bn(const cy_conv_desc* d, const void* dA, const cy_bn_bwd_in* bn, const void* w_packed, void* out,
                        void* out2, void* ws, size_t ws_bytes, void* stream);

/* A data gradient whose output (all couts, or the second part of a split output: channels [c0, c0 + C)) is the dA of
 * a BatchNorm + ReLU: the epilogue adds that layer's backward sums (dz = dA * [scale*y + shift > 0], dz * xhat) into
 * `acc` from the values it has in registers -- y of that layer is read once, at the output positions -- and the reduce
 * launch (cy_bn_relu_bwd_reduce_acc) is not needed.  Flow-kernel plans without split-K whose cout blocks align with the
 * range: ask cy_conv3x3_dgrad_dz_ok(d, c0, C). */
typedef struct cy_bn_dz_out {
  const void* y;        /* [N,H,W,C], contiguous */
  const float* coef;    /* that layer's forward [5][C] */
  const cy_bn_acc* acc;
  int32_t c0, C;
} cy_bn_dz_out;
int cy_conv3x3_dgrad_dz_ok(const cy_conv_desc* d, int c0, int C);
int cy_conv3x3_dgrad_dz(const cy_conv_desc* d, const void* dy, const void* w_packed, void* out, void* out2,
                        const cy_bn_dz_out* dz, void* ws, size_t ws_bytes, void* stream);

/* First layer (input_dim 1..4, arch/unet.py:72): x is the f32 NCHW image
 * [N,Cin,H,W]; w is the reference-layout f32 weight [Cout][Cin][3][3]. */
int cy_conv3x3_first_num_partials(int N, int H, int W, int Cout);
int cy_conv3x3_first_fwd(const float* x, const float* w, void* out, float* stats, int N, int Cin,
                         int H, int W, int Cout, int out_dtype, void* stream);
size_t cy_conv3x3_first_wgrad_ws_bytes(int N, int Cin, int H, int W, int Cout);
int cy_conv3x3_first_wgrad(const float* x, const void* dy, float* dw, int accumulate, int N, int Cin,
                           int H, int W, int Cout, int dy_dtype, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d (training statistics) + ReLU   (arch/unet.py:22-23,25-26,40-41)
 * ------------------------------------------------------------------------ */
/* Reduce conv-epilogue partials to per-channel mean / biased var, fold them
 * with gamma/beta into scale/shift, and update running stats
 * (momentum, unbiased var; torch.nn.BatchNorm2d semantics).
 * use_batch_stats=0 (module.eval()): scale/shift come from running stats.
 * update_running=0: track_running_stats disabled (utils/utils.py:225-237). */
int cy_bn_finalize(const float* partials, int num_partials, int C, double count,
                   const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int use_batch_stats,
                   int update_running, float* scale, float* shift, float* mean, float* invstd,
                   void* stream);

/* out = relu(scale[c]*y + shift[c]) over npix pixels of C channels. */
int cy_bn_relu_apply(const void* y, const float* scale, const float* shift, void* out,
                     long npix, int C, int y_dtype, int out_dtype, void* stream);

/* The same, and the 2x2 max of the result (nn.MaxPool2d(2) of the block output, contrastyou/arch/unet.py:108-121)
 * into `pooled` [N,H,W,C] in one pass; y / out are [N,2H,2W,C] (H, W: the POOLED dims).  out and pooled share
 * out_dtype = y_dtype; the pooled values are maxima of the stored (rounded) outputs. */
int cy_bn_relu_apply_pool(const void* y, const float* scale, const float* shift, void* out, void* pooled,
                          int N, int H, int W, int C, int y_dtype, int out_dtype, void* stream);

/* Backward of y -> relu(bn(y)).  Step 1: per-channel sums of
 * dz = da*(scale*y+shift > 0) and dz*xhat into partials
 * float[num_partials][2][C];  num_partials from cy_bn_bwd_num_partials. */
int cy_bn_bwd_num_partials(long npix, int C);
int cy_bn_relu_bwd_reduce(const void* da, int ld_da, const void* y, const float* scale,
                          const float* shift, const float* mean, const float* invstd,
                          float* partials, long npix, int C, int dtype, void* stream);
/* Step 2: dgamma/dbeta from the partials (fixed summation order; accumulate != 0
 * adds into them) and the per-channel coefficients coef[2][C] = {k1, k0} of step 3. */
int cy_bn_bwd_finalize(const float* partials, int num_partials, int C, const float* scale,
                       const float* mean, const float* invstd, double count, int batch_stats,
                       float* dgamma, float* dbeta, int accumulate, float* coef, void* stream);
/* Step 3: dy = scale*dz + k1*y + k0, i.e. scale*(dz - dbeta/M - xhat*dgamma/M) for batch
 * statistics and scale*dz (k1 = k0 = 0) for eval-mode BN. */
int cy_bn_relu_bwd_apply(const void* da, int ld_da, const void* y, const float* scale,
                         const float* shift, const float* coef, void* dy, long npix, int C,
                         int dtype, void* stream);

/* ------------------------------------------------------------------------
 * MaxPool2d(2,2) backward (arch/unet.py:67-70) and nearest-x2 backward
 * (arch/unet.py:38): the forward halves are folded into the conv loads.
 * ------------------------------------------------------------------------ */
/* dx[N,2H,2W,C] = route(dpool[N,H,W,C]) to the first arg-max of each 2x2
 * window of x (+ add[N,2H,2W,C] if add != NULL, pitch ld_add). */
int cy_maxpool2_bwd(const void* x, const void* dpool, const void* add, int ld_add, void* dx, int N,
                    int H, int W, int C, int dtype, void* stream);
/* dx[N,H,W,C] = sum of the 2x2 block of dup[N,2H,2W,C] (pitch ld_dup). */
/* cy_maxpool2_bwd whose output is the dA of a BatchNorm+ReLU (the block's last one, y = its raw conv output): also
 * writes that layer's backward partial sums (what cy_bn_relu_bwd_reduce would compute from dx and y in a second
 * pass), cy_maxpool2_bwd_bn_num_partials rows of [2][C]; feed them to cy_bn_bwd_finalize. */
int cy_maxpool2_bwd_bn_num_partials(int N, int H, int W, int C);
int cy_maxpool2_bwd_bn(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* partials, int N, int H, int W, int C, int dtype, void* stream);
int cy_upsample2_bwd(const void* dup, int ld_dup, void* dx, int N, int H, int W, int C, int dtype,
                     void* stream);

/* ------------------------------------------------------------------------
 * 1x1 classifier head  nn.Conv2d(C, K, 1) + bias  (arch/unet.py:102)
 * logits are f32 [N,H,W,K].
 * ------------------------------------------------------------------------ */
int cy_head1x1_fwd(const void* x, const float* w, const float* b, float* logits, long npix, int C,
                   int K, int x_dtype, void* stream);
size_t cy_head1x1_bwd_ws_bytes(long npix, int C, int K);
int cy_head1x1_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                   float* db, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                   void* stream);
/* the same with the parameter gradients ADDED into dw [K][C] / db [K] (the live .grad buffers of the head: autograd's
 * accumulation folded into the reduction, as for the conv and BatchNorm parameters) */
int cy_head1x1_bwd_into(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                        float* db, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------
 * Supervised loss  KL_div(softmax(logits), one_hot(target))
 * (semi_seg/epochers/epocher.py:317-318, contrastyou/losses/kl.py:112-125,
 *  contrastyou/utils/general.py:114-120):
 *   loss = mean_p -log((softmax(z_p)[t_p] + eps) / (1 + eps))
 * ------------------------------------------------------------------------ */
size_t cy_softmax_kl_ws_bytes(long npix);
int cy_softmax_kl_fwd(const float* logits, const int64_t* target, float* loss, long npix, int K,
                      float eps, void* ws, size_t ws_bytes, void* stream);
/* dlogits = gscale[0]/npix * dloss/dz  (gscale: device scalar, upstream grad) */
int cy_softmax_kl_bwd(const float* logits, const int64_t* target, const float* gscale,
                      float* dlogits, long npix, int K, float eps, void* stream);

/* ------------------------------------------------------------------------
 * Encoder projection head (contrastyou/projectors/heads.py:14-22,81-96,
 * projectors/nn.py:47-54): AdaptiveAvgPool2d(1) -> Linear -> LeakyReLU(0.01)
 * -> Linear -> x / max(||x||_2, 1e-12).
 * ------------------------------------------------------------------------ */
/* pooled[n][c] = mean_{hw} x[n,h,w,c]   (f32 out) */
int cy_avgpool_fwd(const void* x, float* pooled, int N, int HW, int C, int dtype, void* stream);
/* dx[n,h,w,c] = dpooled[n][c] / HW */
int cy_avgpool_bwd(const float* dpooled, void* dx, int N, int HW, int C, int dtype, void* stream);
/* y[m][o] = act(sum_i x[m][i]*w[o][i] + b[o]);  act: 0 none, 1 LeakyReLU(slope) */
int cy_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int I, int O,
                  int act, float slope, void* stream);
/* given dy (pre-activation gradient is formed inside from y when act=1):
 * dx[m][i], dw[o][i], db[o].  Any of dx/dw/db may be NULL. */
int cy_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                  float* dw, float* db, int M, int I, int O, int act, float slope, void* stream);
/* the same, with dw / db ADDED into (the parameters' .grad buffers: no separate accumulation launch) */
int cy_linear_bwd_into(const float* x, const float* w, const float* y, const float* dy, float* dx,
                       float* dw, float* db, int M, int I, int O, int act, float slope, void* stream);
/* The whole ProjectionHead (contrastyou/projectors/heads.py:12-22,81-96, head_type "mlp", normalize) in one launch:
 * feat NHWC [B][HW][C] (dtype) -> pooled [B][C] -> y1 = lrelu(W1 pooled + b1) [B][hid] -> y2 = W2 y1 + b2 [B][out]
 * -> z = y2 / max(|y2|, eps), norms [B]; pooled / y1 / y2 / norms are what the backward needs.
 * Backward: dfeat (dtype, may be NULL), and dW1 [hid][C], db1, dW2 [out][hid], db2 written or ADDED into
 * (accumulate); ws: cy_proj_head_bwd_ws_bytes.  C % 8 == 0, hid % 4 == 0; C, hid, out <= 512. */
int cy_proj_head_fwd(const void* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                     float* pooled, float* y1, float* y2, float* z, float* norms, int B, int HW, int C, int hid,
                     int out, float slope, float eps, int dtype, void* stream);
size_t cy_proj_head_bwd_ws_bytes(int B, int hid, int out);
int cy_proj_head_bwd(const float* dz, const float* pooled, const float* y1, const float* y2, const float* norms,
                     const float* w1, const float* w2, void* dfeat, float* dw1, float* db1, float* dw2, float* db2,
                     int accumulate, void* ws, size_t ws_bytes, int B, int HW, int C, int hid, int out, float slope,
                     float eps, int dtype, void* stream);
/* z = x / max(||x||,eps) row-wise; norms[m] saved for backward. */
int cy_l2norm_fwd(const float* x, float* z, float* norms, int M, int D, float eps, void* stream);
int cy_l2norm_bwd(const float* x, const float* norms, const float* dz, float* dx, int M, int D,
                  float eps, void* stream);

/* ------------------------------------------------------------------------
 * InfoNCE / SupCon loss  (contrastyou/losses/contrastive.py:14-20,31-100)
 * P = [z1; z2] is [R=2n][D] f32 (unit rows), labels[n] int32 (row i and
 * i+n share labels[i]); t = temperature.
 *   S = P P^T / t;  M = max(S) (incl. diagonal);  E = exp(S - M)
 *   loss = -mean_i [ sum_{j in pos(i)} ((S_ij - M) - log(sum_{k!=i} E_ik + 1e-16)) / |pos(i)| ]
 * row_stats (f32 [R][4]) keeps {denominator_i, |pos(i)|, sum_pos S_ij, M}
 * for the backward pass.  Rows with |pos(i)| == 0 give NaN exactly like the
 * reference (which then raises RuntimeError, contrastive.py:98).
 * ------------------------------------------------------------------------ */
/* Either labels (int32 [n]) or pos_mask (uint8 [n][n], 1 = positive pair; the
 * `mask=` argument of SupConLoss1.forward) must be non-NULL.
 * S: f32 [R][R] buffer that receives P P^T / t (kept for backward). */
int cy_supcon_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* S,
                  float* loss, float* row_stats, int n, int D, float t, void* stream);
/* dP[R][D] = gscale[0] * dloss/dP.  G: f32 [R][R] scratch. */
int cy_supcon_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* S,
                  const float* row_stats, const float* gscale, float* G, float* dP, int n, int D,
                  float t, void* stream);
/* SupConLoss1(exclude_other_pos=True) (losses/contrastive.py:87-91): per positive pair the denominator holds that pair and
 * the negatives only, the negatives' sum rescaled by 1 / (neg / (pos + neg) + 1e-4).  On the materialised similarity
 * matrix S [2n][2n]; row_stats [2n][4] and tmp [4 * 2n + 1] are kept for the backward (tmp[8n] = the global maximum). */
int cy_supcon_excl_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* S, float* loss,
                       float* row_stats, float* tmp, int n, int D, float t, void* stream);
int cy_supcon_excl_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* S,
                       const float* row_stats, const float* tmp, const float* gscale, float* G, float* dP, int n, int D,
                       float t, void* stream);
/* materialise sim_logits (S-M), sim_exp, pos_mask, neg_mask, each f32 [R][R]
 * (contrastive.py:79-82; read only for the TensorBoard figures). any may be NULL */
/* The same loss and gradient with S (and the gradient matrix) formed tile by tile in the MFMA accumulators and
 * never written: forward = row sums with the global shift M = max_i |P_i|^2 / t (the maximum of S lies on its
 * diagonal), backward = (1/t) G P with the S tile recomputed.  D <= 256, D % 8 == 0.  row_stats as above;
 * diag_out (optional) receives S_ii = |P_i|^2 / t, what SupConLoss1's unit-norm assertion looks at.
 * ws: cy_supcon_fused_ws_bytes(n, D) bytes, the same buffer for both calls. */
size_t cy_supcon_fused_ws_bytes(int n, int D);
int cy_supcon_fused_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* loss,
                        float* row_stats, float* diag_out, void* ws, size_t ws_bytes, int n, int D, float t,
                        void* stream);
int cy_supcon_fused_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* row_stats,
                        const float* gscale, float* dP, void* ws, size_t ws_bytes, int n, int D, float t,
                        void* stream);
int cy_supcon_matrices(const float* S, const float* row_stats, const int32_t* labels,
                       const uint8_t* pos_mask, float* sim_logits, float* sim_exp,
                       float* pos_out, float* neg_out, int n, void* stream);
/* f32 GEMM on the f32 MFMA: C[M][N] = alpha * A[M][K] * op(B); b_trans=1: B is
 * [N][K] (C = A B^T), b_trans=0: B is [K][N]. */
int cy_sgemm(const float* A, const float* B, float* C, int M, int N, int K, float alpha,
             int b_trans, void* stream);

/* ------------------------------------------------------------------------
 * In-step affine augmentation (semi_seg/augment.py:297-311, configured at
 * semi_seg/epochers/epocher.py:226-238; third-party `rising`, un-vendored):
 * nearest-neighbour resampling with a per-sample 2x3 matrix theta (f32 [N][6],
 * output-normalised -> input-normalised coordinates, align_corners=False,
 * zero padding) and optional gamma x**g (g f32[N], NULL = none) applied
 * BEFORE the geometry as the reference does for mode="image".
 * Tensors are NHWC of `dtype` (an [N,1,H,W] image is its own NHWC view).
 * ------------------------------------------------------------------------ */
int cy_affine_nearest_fwd(const void* x, void* out, const float* theta, const float* gamma, int N,
                          int C, int H, int W, int dtype, void* stream);
/* dx = gather-form adjoint (deterministic, no atomics). */
int cy_affine_nearest_bwd(const void* dout, void* dx, const float* theta, int N, int C, int H,
                          int W, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Segmentation metric helper: per-sample per-class intersection/union counts
 * of argmax(logits) vs target (contrastyou/meters/general_dice_meter.py:
 * 37-72,93-110).  counts is int64 [N][K][2] = {intersection, union}.
 * ------------------------------------------------------------------------ */
int cy_dice_counts(const float* logits, const int64_t* target, int64_t* counts, int N, int HW,
                   int K, void* stream);

/* ------------------------------------------------------------------------
 * Mean-teacher / consistency hooks (semi_seg/hooks/mt.py:49-82,
 * semi_seg/hooks/consistency.py:22-38)
 * ------------------------------------------------------------------------ */
/* teacher = (alpha*teacher + (1-alpha)*student) * (1 - weight_decay) over a
 * flat f32 buffer of n elements. */
int cy_ema_update(float* teacher, const float* student, long n, float alpha, float weight_decay,
                  void* stream);
/* loss = mean((softmax(a)-softmax(b))^2) over [npix][K] f32 logits */
size_t cy_softmax_mse_ws_bytes(long npix);
int cy_softmax_mse_fwd(const float* a, const float* b, float* loss, long npix, int K, void* ws,
                       size_t ws_bytes, void* stream);
/* da (and db if not NULL) = gscale[0] * dloss/d{a,b} */
int cy_softmax_mse_bwd(const float* a, const float* b, const float* gscale, float* da, float* db,
                       long npix, int K, void* stream);

/* ------------------------------------------------------------------------
 * RAdam step over a flat f32 parameter buffer (torch.optim.RAdam semantics,
 * contrastyou/trainer/base.py:66-75, config/base.yaml:10-13).
 * `step` is the 1-based step count AFTER this update.
 * ------------------------------------------------------------------------ */
int cy_radam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                  void* stream);

/* ------------------------------------------------------------------------
 * Dense (pixel-wise) contrastive projector
 * (contrastyou/projectors/heads.py:31-41,99-123: Conv1x1(C,hid) -> LeakyReLU ->
 * Conv1x1(hid,out) -> AdaptiveAvgPool2d((sh,sw)) -> L2 norm) and point
 * sampling (semi_seg/hooks/infonce.py:31-46).
 * The second 1x1 conv commutes with the average pool, so the fused kernel
 * produces  hpool[bin][hid] = mean over the bin's pixels of
 * lrelu(W1 x + b1)  straight from the NHWC feature map x [N][H][W] (pixel
 * stride ldx, dtype); the [bins][hid] x [hid][out] product and the
 * normalisation use cy_linear_* / cy_l2norm_*.
 * bins: int32 [nb][3] = (image, bin row, bin col), distinct; NULL = all
 * N*sh*sw bins in row-major order.  Bin extents follow torch's adaptive
 * pooling: rows [floor(i*H/sh), ceil((i+1)*H/sh)).  Requires sh<=H, sw<=W.
 * ------------------------------------------------------------------------ */
int cy_dense_proj_fwd(const void* x, const float* w1, const float* b1, const int32_t* bins, int nb,
                      float* hpool, int N, int H, int W, int C, int ldx, int hid, int sh, int sw,
                      float slope, int dtype, void* stream);
size_t cy_dense_proj_bwd_ws_bytes(int nb, int C, int hid);
/* dx (dtype, NHWC, ldx) is ACCUMULATED into (zero it first; may be NULL);
 * dw1 [hid][C], db1 [hid] f32, written or accumulated; hid <= 256. */
int cy_dense_proj_bwd(const void* x, const float* w1, const float* b1, const int32_t* bins, int nb,
                      const float* dhpool, void* dx, float* dw1, float* db1, int accumulate, int N,
                      int H, int W, int C, int ldx, int hid, int sh, int sw, float slope, int dtype,
                      void* ws, size_t ws_bytes, void* stream);
/* nn.AdaptiveAvgPool2d((sh,sw)) on an NHWC map -> f32 [nb][C] (same bin list
 * convention); backward for the all-bins case (gather form, deterministic). */
int cy_adaptive_avgpool_fwd(const void* x, const int32_t* bins, int nb, float* out, int N, int H,
                            int W, int C, int ldx, int sh, int sw, int dtype, void* stream);
int cy_adaptive_avgpool_bwd(const float* dpool, void* dx, int N, int H, int W, int C, int ldx,
                            int sh, int sw, int dtype, void* stream);
/* nn.AdaptiveMaxPool2d(size) (projectors/nn.py:16-23: pool_name="adaptive_max" of both projection heads) on an NHWC map:
 * out [N*sh*sw][C] f32 and arg [N*sh*sw][C] = pixel index (inside the image) of the first maximum in row-major order;
 * backward in gather form (bins may overlap: deterministic, no atomics). */
int cy_adaptive_maxpool_fwd(const void* x, float* out, int32_t* arg, int N, int H, int W, int C, int ldx, int sh,
                            int sw, int dtype, void* stream);
int cy_adaptive_maxpool_bwd(const float* dpool, const int32_t* arg, void* dx, int N, int H, int W, int C, int ldx,
                            int sh, int sw, int dtype, void* stream);
/* out[m][:] = src[idx[m]][:] (f32 rows of length D); backward scatters rows
 * (idx distinct). */
int cy_gather_rows_fwd(const float* src, const int32_t* idx, float* out, int M, int D,
                       void* stream);
int cy_gather_rows_bwd(const float* dout, const int32_t* idx, float* dsrc, int M, int D,
                       void* stream);

/* ------------------------------------------------------------------------
 * Cluster heads and discrete mutual-information losses
 * (contrastyou/projectors/heads.py:44-78,125-173; contrastyou/losses/discreteMI.py:
 * 90-170,201-261).  Probability maps are NHWC f32 [N][H][W][k], k <= 64.
 * ------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------
 * Stacked cluster heads in one pass (DenseClusterHead / ClusterHead, head_type="linear",
 * projectors/heads.py:125-173): probs[S][M][k] = softmax over each sub-head's k of (W x + b) / T with
 * K = S*k <= 128 outputs and C in {32, 64} inputs, rows x[M][C] (bf16 / f16 / f32, NHWC pixels); exact f32
 * MFMA; the logits are never written.  Backward from (probs, dprobs): dx[M][C] (dtype of x; NULL: skipped),
 * dw[K][C], db[K] (NULL: skipped; ws from cy_cluster_head_bwd_ws_bytes).
 * ------------------------------------------------------------------------ */
int cy_cluster_head_fwd(const void* x, const float* w, const float* b, float* probs, long M, int C, int K, int S,
                        int k, float invT, int dtype, void* stream);
size_t cy_cluster_head_bwd_ws_bytes(long M, int C);
int cy_cluster_head_bwd(const void* x, const float* w, const float* probs, const float* dprobs, void* dx, float* dw,
                        float* db, long M, int C, int K, int S, int k, float invT, int dtype, void* ws,
                        size_t ws_bytes, void* stream);

/* logits [M][S*k] -> probs [S][M][k] = softmax(logits*invT) within each of the
 * S sub-heads (SoftmaxWithT, projectors/nn.py:35-44); S=1 is a row softmax. */
int cy_group_softmax_fwd(const float* logits, float* probs, long M, int S, int k, float invT,
                         void* stream);
int cy_group_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, long M, int S,
                         int k, float invT, void* stream);
/* J[T*T][k][k], T = 2*pad+1:  J[(u,v)][i][j] = scale * sum_{n,a,b}
 * x1[n][a+u-pad][b+v-pad][i] * x2[n][a][b][j] (zero outside the image);
 * scale = 1/(N*H*W) if normalise else 1.  pad=0, normalise=1 is
 * compute_joint_2D_with_padding_zeros; pad>0 is the F.conv2d of
 * compute_joint_2D; H=W=1 gives compute_joint on [N][k] vectors. */
size_t cy_joint_ws_bytes(int N, int H, int W, int k, int pad);
int cy_joint_fwd(const float* x1, const float* x2, float* J, int N, int H, int W, int k, int pad,
                 int normalise, void* ws, size_t ws_bytes, void* stream);
/* dx1, dx2 (either may be NULL) = gscale[0] * adjoint of cy_joint_fwd applied to dJ */
int cy_joint_bwd(const float* x1, const float* x2, const float* dJ, const float* gscale, float* dx1,
                 float* dx2, int N, int H, int W, int k, int pad, int normalise, void* stream);
/* loss on a joint.  mode 0: IIDSegmentationLoss padding 0; mode 1: padding>0
 * (subtract min, +1e-8, per-displacement and global normalisation, loss/T^2);
 * mode 2: IIDLoss on vectors (symmetrise if `symmetric`, normalise to 1).
 * out2[0] = loss(lamda), out2[1] = loss(lamda=1); P (may be NULL) = the
 * normalised joint; dJ (may be NULL) = dloss/dJ. */
size_t cy_iid_loss_ws_bytes(int TT, int k);
int cy_iid_loss(const float* J, float* out2, float* P, float* dJ, int TT, int k, int mode,
                int symmetric, float lamda, float eps, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * UNet2.Block = Conv2d(3x3, bias) -> GroupNorm(G, C) -> SiLU
 * (contrastyou/arch/unet2.py:208-224).  The conv runs on cy_conv3x3_* without
 * bias; these kernels fold the bias into the normalisation:
 * out = silu(gamma * ((y + bias) - mean_g) * rstd_g + beta), NHWC `dtype`.
 * mean_rstd: f32 [N][G][2], written by fwd, read by bwd.
 * ------------------------------------------------------------------------ */
size_t cy_gn_ws_bytes(int N, int C);
int cy_gn_silu_fwd(const void* y, int ldy, const float* bias, const float* gamma, const float* beta,
                   void* out, int ldo, float* mean_rstd, int N, int HW, int C, int G, float eps,
                   int dtype, void* ws, size_t ws_bytes, void* stream);
/* du = dloss/d(conv output); dgamma, dbeta, dbias f32 [C] (NULL to skip). */
int cy_gn_silu_bwd(const void* y, int ldy, const void* dz, int ldd, const float* bias,
                   const float* gamma, const float* beta, const float* mean_rstd, void* du, int ldu,
                   float* dgamma, float* dbeta, float* dbias, int accumulate, int N, int HW, int C,
                   int G, int dtype, void* ws, size_t ws_bytes, void* stream);
/* The same with the time-embedding modulation of ResnetBlock / Block (contrastyou/arch/unet2.py:216-220,240-246):
 * out = SiLU(GroupNorm(y + bias) * (mod_scale[n][c] + 1) + mod_shift[n][c]); mod_* are f32 [N][C].  Backward also gives
 * dmod_scale / dmod_shift [N][C] (NULL to skip). */
int cy_gn_silu_mod_fwd(const void* y, int ldy, const float* bias, const float* gamma, const float* beta,
                       const float* mod_scale, const float* mod_shift, void* out, int ldo, float* mean_rstd, int N,
                       int HW, int C, int G, float eps, int dtype, void* ws, size_t ws_bytes, void* stream);
int cy_gn_silu_mod_bwd(const void* y, int ldy, const void* dz, int ldd, const float* bias, const float* gamma,
                       const float* beta, const float* mod_scale, const float* mod_shift, const float* mean_rstd,
                       void* du, int ldu, float* dgamma, float* dbeta, float* dbias, float* dmod_scale,
                       float* dmod_shift, int accumulate, int N, int HW, int C, int G, int dtype, void* ws,
                       size_t ws_bytes, void* stream);
/* F.interpolate(x, size=(h,w), mode="bilinear", align_corners=False) on an NHWC
 * tensor (semi_seg/hooks/cc.py:132, ccblock.py:300). */
int cy_bilinear_fwd(const void* x, void* out, int N, int H, int W, int C, int h, int w, int dtype,
                    void* stream);

/* ------------------------------------------------------------------------
 * The glue of the reference's second backbone, `UNet2` (contrastyou/arch/unet2.py): everything that is not a
 * 3x3 conv + GroupNorm + SiLU block.  f32, NHWC maps viewed as [pixels][channels] matrices.
 *   7x7 stem (:45), Downsample = Conv2d(4, 2, 1) (:180-181), 1x1 projections      -> cy_im2col + cy_gemm_strided
 *   Upsample = ConvTranspose2d(4, 2, 1) (:176-177) and every data gradient        -> cy_gemm_strided + cy_col2im
 *   LayerNorm over channels (:183-194)                                            -> cy_chan_layernorm_*
 *   LinearAttention (:245-271): q.softmax(-2) * scale, k.softmax(-1), two einsums  -> cy_head_softmax_*,
 *                                                                                    cy_col_softmax_*, cy_gemm_strided
 *   Attention (:274-304): softmax(q k^T) v over the positions of the bottleneck   -> cy_gemm_strided, cy_row_softmax_*
 * ------------------------------------------------------------------------ */
typedef struct {
  long rs, cs; /* element (i, j) of a matrix at i*rs + j*cs (in elements) */
  long s1, s2; /* batch (b1, b2) starts at b1*s1 + b2*s2 */
} cy_mat_layout;
/* C[b] = alpha * A[b] (M x K) * B[b] (K x N) (+ bias[n]) (+ C[b] when accumulate) for nb1 x nb2 batches, every
 * operand addressed through its layout (so transposes, channel slices of a wider map and per-head blocks need no
 * copies), on the f32 MFMA.  ksplit > 1 splits K into that many ranges whose partial products go through `ws`
 * (cy_gemm_strided_ws_bytes) and are summed in a fixed order: run-to-run identical results. */
size_t cy_gemm_strided_ws_bytes(int M, int N, int nbatch, int ksplit);
int cy_gemm_strided(const float* A, const cy_mat_layout* la, const float* B, const cy_mat_layout* lb, float* C,
                    const cy_mat_layout* lc, const float* bias, int M, int N, int K, int nb1, int nb2, float alpha,
                    int accumulate, int ksplit, float* ws, size_t ws_bytes, void* stream);
/* cols[(n, ho, wo)][(kh, kw, c)] = x[n, ho*stride - pad + kh, wo*stride - pad + kw, c] (zero outside);
 * Ho = (H + 2 pad - KH) / stride + 1.  x [N,H,W,C], cols [N*Ho*Wo][KH*KW*C]. */
int cy_im2col(const float* x, float* cols, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
              void* stream);
/* the adjoint of cy_im2col as a gather: out[n,h,w,c] = bias[c] (or 0) + the sum of the cols entries that were read
 * from (n,h,w,c).  H, W are those of `out`. */
int cy_col2im(const float* cols, const float* bias, float* out, int N, int H, int W, int C, int KH, int KW, int stride,
              int pad, void* stream);
/* out[n] (+)= sum_m x[m][n], two ordered stages (bias gradients) */
size_t cy_colsum_ws_bytes(long M, int N);
int cy_colsum(const float* x, float* out, long M, int N, int accumulate, float* ws, size_t ws_bytes, void* stream);
/* y[p][c] = (x[p][c] - mean_p) / sqrt(var_p + eps) * g[c] + b[c] over the C channels of each of M pixels */
int cy_chan_layernorm_fwd(const float* x, const float* g, const float* b, float* y, long M, int C, float eps,
                          void* stream);
size_t cy_chan_layernorm_bwd_ws_bytes(long M, int C);
/* dx, and dg / db (both or neither) */
int cy_chan_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, float* db, long M,
                          int C, float eps, float* ws, size_t ws_bytes, void* stream);
/* y[p][h][d] = scale * softmax_d(x[p][off + h*dh + d]) for rows of ld floats (q of the linear attention) */
int cy_head_softmax_fwd(const float* x, int ld, int off, float* y, long M, int heads, int dh, float scale,
                        void* stream);
/* its backward, written into columns off.. of rows of ld_dx floats */
int cy_head_softmax_bwd(const float* y, const float* dy, float* dx, int ld_dx, int off, long M, int heads, int dh,
                        float scale, void* stream);
/* y[b][p][c] = softmax over the n positions p of image b of x[(b*n + p)*ld + off + c] (k of the linear attention) */
size_t cy_col_softmax_ws_bytes(int B, int n, int Ch);
int cy_col_softmax_fwd(const float* x, int ld, int off, float* y, int B, int n, int Ch, float* ws, size_t ws_bytes,
                       void* stream);
/* dx[(b*n + p)*ld_dx + off + c] = y * (dy - t[b][c]) with t[b][c] = sum_p y * dy supplied by the caller */
int cy_col_softmax_bwd(const float* y, const float* dy, const float* t, float* dx, int ld_dx, int off, int B, int n,
                       int Ch, void* stream);
/* softmax of every row of x [rows][n], in place; backward: dp <- p * (dp - sum_j p * dp) */
int cy_row_softmax_fwd(float* x, long rows, int n, void* stream);
int cy_row_softmax_bwd(const float* p, float* dp, long rows, int n, void* stream);
/* UNet2's time embedding (contrastyou/arch/unet2.py:51-58,161-173,230-231): SinusoidalPosEmb(dim) of time [B] ->
 * out [B][dim] (sin half, cos half); elementwise SiLU (kind 0) / exact GELU (kind 1) of the embedding MLPs, f32 */
int cy_sinusoidal_emb(const float* time, float* out, int B, int dim, void* stream);
int cy_act_fwd(const float* x, float* y, long n, int kind, void* stream);
int cy_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTRASTYOU_HIP_H */
