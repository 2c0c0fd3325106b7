// In-step affine augmentation, EMA teacher update and the RAdam step.
//   semi_seg/augment.py:297-311 + semi_seg/epochers/epocher.py:226-238  (rising BaseAffine,
//       Mirror, GammaCorrection -- third-party, un-vendored; geometry is passed in as theta)
//   semi_seg/hooks/mt.py:49-82                                         (EMAUpdater)
//   contrastyou/trainer/base.py:66-75 (torch.optim.RAdam)
#include "cy_common.h"

namespace {

// output pixel (oy,ox) -> input pixel (iy,ix) under theta (affine_grid / grid_sample conventions,
// align_corners=False, nearest = round-half-to-even); returns false when the source is outside the
// image (zero padding).  Nearest-neighbour sampling is discontinuous in the source coordinate, so the
// coordinate arithmetic is part of the definition: it is done in f64, one correctly rounded operation
// at a time in the order written here (no FMA contraction), which any IEEE-754 host reproduces bit for
// bit (oracle.losses.affine_source_index).
__device__ __forceinline__ bool affine_src(const float* th, int oy, int ox, int H, int W, int* iy,
                                           int* ix) {
#pragma clang fp contract(off)
  const double t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
  const double xo = __dsub_rn(__ddiv_rn((double)(2 * ox + 1), (double)W), 1.0);
  const double yo = __dsub_rn(__ddiv_rn((double)(2 * oy + 1), (double)H), 1.0);
  const double xi = __dadd_rn(__dadd_rn(__dmul_rn(t0, xo), __dmul_rn(t1, yo)), t2);
  const double yi = __dadd_rn(__dadd_rn(__dmul_rn(t3, xo), __dmul_rn(t4, yo)), t5);
  const double px = __dmul_rn(__dsub_rn(__dmul_rn(__dadd_rn(xi, 1.0), (double)W), 1.0), 0.5);
  const double py = __dmul_rn(__dsub_rn(__dmul_rn(__dadd_rn(yi, 1.0), (double)H), 1.0), 0.5);
  const double rx = rint(px), ry = rint(py);
  if (!(rx >= 0.0 && rx <= (double)(W - 1) && ry >= 0.0 && ry <= (double)(H - 1))) return false;
  *ix = (int)rx;
  *iy = (int)ry;
  return true;
}

// thread = (output pixel, group of 8 channels)
template <typename T>
__global__ void __launch_bounds__(256)
    affine_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, const float* __restrict__ theta,
                      const float* __restrict__ gamma, int N, int C, int H, int W) {
  const int G = (C + 7) / 8;
  const long total = (long)N * H * W * G;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int g = (int)(e % G);
    const long p = e / G;
    const int ox = (int)(p % W);
    const int oy = (int)((p / W) % H);
    const int n = (int)(p / ((long)W * H));
    int iy, ix;
    const bool ok = affine_src(theta + n * 6, oy, ox, H, W, &iy, &ix);
    const int c0 = g * 8;
    T* op = out + p * C + c0;
    const T* ip = x + ((long)(n * H + iy) * W + ix) * C + c0;
    const float gm = gamma ? gamma[n] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (c0 + j < C) {
        float v = ok ? to_f32<T>(ip[j]) : 0.f;
        if (gamma && ok) v = powf(v, gm);
        op[j] = from_f32<T>(v);
      }
    }
  }
}

// adjoint in gather form: every (input pixel, 8-channel group) scans the output window that can
// map onto the pixel and sums the matching gradients in scan order (deterministic, no atomics)
template <typename T>
__global__ void __launch_bounds__(256)
    affine_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx,
                      const float* __restrict__ theta, int N, int C, int H, int W) {
  const int G = (C + 7) / 8;
  const long total = (long)N * H * W * G;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int g = (int)(e % G);
    const long p = e / G;
    const int ix = (int)(p % W);
    const int iy = (int)((p / W) % H);
    const int n = (int)(p / ((long)W * H));
    const float* th = theta + n * 6;
    // inverse map of the pixel centre: pixel -> normalised -> theta^-1 -> output pixel
    const float xi = (2.f * ix + 1.f) / (float)W - 1.f;
    const float yi = (2.f * iy + 1.f) / (float)H - 1.f;
    const float det = th[0] * th[4] - th[1] * th[3];
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;  // empty window when theta is singular
    if (fabsf(det) > 1e-12f) {
      const float i00 = th[4] / det, i01 = -th[1] / det, i10 = -th[3] / det, i11 = th[0] / det;
      const float bx = xi - th[2], by = yi - th[5];
      const float xo = i00 * bx + i01 * by, yo = i10 * bx + i11 * by;
      const float cx = ((xo + 1.f) * (float)W - 1.f) * 0.5f;
      const float cy = ((yo + 1.f) * (float)H - 1.f) * 0.5f;
      // half extent (in output pixels) of the pre-image of one input pixel, plus a margin
      const int rx = (int)ceilf(0.5f * (fabsf(i00) + fabsf(i01) * (float)W / (float)H)) + 1;
      const int ry = (int)ceilf(0.5f * (fabsf(i10) * (float)H / (float)W + fabsf(i11))) + 1;
      x0 = (int)floorf(cx) - rx, x1 = (int)ceilf(cx) + rx;
      y0 = (int)floorf(cy) - ry, y1 = (int)ceilf(cy) + ry;
      if (x0 < 0) x0 = 0;
      if (y0 < 0) y0 = 0;
      if (x1 > W - 1) x1 = W - 1;
      if (y1 > H - 1) y1 = H - 1;
    }
    const int c0 = g * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = y0; oy <= y1; ++oy)
      for (int ox = x0; ox <= x1; ++ox) {
        int sy2, sx2;
        if (affine_src(th, oy, ox, H, W, &sy2, &sx2) && sy2 == iy && sx2 == ix) {
          const T* gp = dout + ((long)(n * H + oy) * W + ox) * C + c0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c0 + j < C) acc[j] += to_f32<T>(gp[j]);
        }
      }
    T* dp = dx + p * C + c0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) dp[j] = from_f32<T>(acc[j]);
  }
}

__global__ void __launch_bounds__(256)
    ema_kernel(float* __restrict__ teacher, const float* __restrict__ student, long n, float alpha,
               float keep) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
    teacher[i] = (alpha * teacher[i] + (1.f - alpha) * student[i]) * keep;
}

struct RAdamScalars {
  float lr, beta1, beta2, eps, wd;
  float bias_c1;    // 1 - beta1^t
  float rect;       // rectification term, <0: un-rectified branch
  float sqrt_bc2;   // sqrt(1 - beta2^t)
};

__global__ void __launch_bounds__(256)
    radam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, long n, RAdamScalars s) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    float grad = g[i];
    const float pv = p[i];
    if (s.wd != 0.f) grad = fmaf(s.wd, pv, grad);
    const float mi = m[i] + (grad - m[i]) * (1.f - s.beta1);  // lerp, as torch does
    const float vi = s.beta2 * v[i] + (1.f - s.beta2) * grad * grad;
    m[i] = mi;
    v[i] = vi;
    const float bc = mi / s.bias_c1;
    float upd;
    if (s.rect >= 0.f)
      upd = bc * s.lr * s.rect * (s.sqrt_bc2 / (sqrtf(vi) + s.eps));
    else
      upd = bc * s.lr;
    p[i] = pv - upd;
  }
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int cy_affine_nearest_fwd(const void* x, void* out, const float* theta, const float* gamma, int N,
                          int C, int H, int W, int dtype, void* stream) {
  if (!x || !out || !theta || N <= 0 || C <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * H * W * ((C + 7) / 8));
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(affine_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x,
                       (bf16*)out, theta, gamma, N, C, H, W);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(affine_fwd_kernel<f16>, dim3(grid), dim3(256), 0, st, (const f16*)x,
                       (f16*)out, theta, gamma, N, C, H, W);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(affine_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x,
                       (float*)out, theta, gamma, N, C, H, W);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_affine_nearest_bwd(const void* dout, void* dx, const float* theta, int N, int C, int H,
                          int W, int dtype, void* stream) {
  if (!dout || !dx || !theta || N <= 0 || C <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * H * W * ((C + 7) / 8));
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(affine_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dout,
                       (bf16*)dx, theta, N, C, H, W);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(affine_bwd_kernel<f16>, dim3(grid), dim3(256), 0, st, (const f16*)dout,
                       (f16*)dx, theta, N, C, H, W);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(affine_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dout,
                       (float*)dx, theta, N, C, H, W);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_ema_update(float* teacher, const float* student, long n, float alpha, float weight_decay,
                  void* stream) {
  if (!teacher || !student || n <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, teacher,
                     student, n, alpha, 1.f - weight_decay);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_radam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                  void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return CY_ERR_ARG;
  RAdamScalars s;
  s.lr = lr, s.beta1 = beta1, s.beta2 = beta2, s.eps = eps, s.wd = weight_decay;
  const double b1t = pow((double)beta1, (double)step), b2t = pow((double)beta2, (double)step);
  const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
  const double rho_inf = 2.0 / (1.0 - (double)beta2) - 1.0;
  const double rho_t = rho_inf - 2.0 * (double)step * b2t / bc2;
  s.bias_c1 = (float)bc1;
  s.sqrt_bc2 = (float)sqrt(bc2);
  if (rho_t > 5.0)
    s.rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf /
                         ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  else
    s.rect = -1.f;
  hipLaunchKernelGGL(radam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, n, s);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

// profiling aid: one thread writes the device's constant-rate wall clock (100 MHz) into buf[slot];
// launched between the kernels of a stream (also inside a captured graph) it timestamps that point
// of the stream on the GPU's own clock.  tools/step_timeline.py builds the step's timeline from it.
static __global__ void stamp_kernel(unsigned long long* buf, int slot) { buf[slot] = wall_clock64(); }

int cy_debug_stamp(unsigned long long* buf, int slot, void* stream) {
  if (!buf || slot < 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, slot);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

// Measurement aid (bench.py's instrumented pass): holds the stream for `micros` microseconds of the GPU's wall
// clock, so that the host can enqueue a whole step behind it and the HIP events around each launch then bracket
// back-to-back GPU execution instead of the host's enqueue latency.  One lane; bounded (<= 50 ms).
static __global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

int cy_debug_spin(int micros, void* stream) {
  if (micros < 0 || micros > 50000) return CY_ERR_ARG;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long)micros * 100ull);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

// Timing-only HIP events for bench.py's per-launch durations: created with hipEventDisableSystemFence, i.e. WITHOUT
// the system-scope release + L2 writeback / invalidate a default event performs when it is recorded -- around a
// single launch that flush lands inside the measured interval (all of the launch's output is still dirty in the
// L2s) and makes the following kernel re-read from HBM what it would have found there.
int cy_debug_event_create(void** ev) {
  if (!ev) return CY_ERR_ARG;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return CY_ERR_LAUNCH;
  *ev = (void*)e;
  return CY_OK;
}

int cy_debug_event_record(void* ev, void* stream) {
  if (!ev) return CY_ERR_ARG;
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? CY_OK : CY_ERR_LAUNCH;
}

int cy_debug_event_elapsed_us(void* e0, void* e1, float* us) {
  if (!e0 || !e1 || !us) return CY_ERR_ARG;
  if (hipEventSynchronize((hipEvent_t)e1) != hipSuccess) return CY_ERR_LAUNCH;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) return CY_ERR_LAUNCH;
  *us = ms * 1000.f;
  return CY_OK;
}

int cy_debug_event_destroy(void* ev) {
  if (!ev) return CY_ERR_ARG;
  return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? CY_OK : CY_ERR_LAUNCH;
}

// Cross-stream hand-over out of a captured HIP graph (data-parallel training: the gradient buckets whose layers
// are done are all-reduced while the rest of the backward graph still runs).  An external event-record node is what
// CUDA offers for this; hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue under capture on
// this runtime, so the hand-over is a counter in device memory: a kernel of the graph increments it, a stream outside
// the graph waits until it has reached the number of the current step (hipStreamWaitValue32, >=).
int cy_stream_wait_value(void* stream, const int* counter, int at_least) {
  if (!counter) return CY_ERR_ARG;
  return hipStreamWaitValue32((hipStream_t)stream, (void*)counter, (uint32_t)at_least, hipStreamWaitValueGte,
                              0xffffffffu) == hipSuccess
             ? CY_OK
             : CY_ERR_LAUNCH;
}

}  // extern "C"
