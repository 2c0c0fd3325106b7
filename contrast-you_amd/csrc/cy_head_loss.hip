// Classifier head and pixel-wise losses / metrics.
//   nn.Conv2d(C, K, 1) + bias                     contrastyou/arch/unet.py:102
//   KL_div(softmax(logits), one_hot(target))      semi_seg/epochers/epocher.py:317-318,
//                                                 contrastyou/losses/kl.py:112-125,
//                                                 contrastyou/utils/general.py:114-120
//   MSE(softmax(a), softmax(b))                   semi_seg/hooks/consistency.py:15,36, mt.py:98,186
//   UniversalDice intersections / unions          contrastyou/meters/general_dice_meter.py:37-110
// All of it is HBM-streaming work over [pixels][K] f32 logits (K = classes).
#include "cy_common.h"

// the wide head on the matrix-core kernels of cy_cluster_head.hip (same shared object)
bool cy_head_wide_ok(int C, int K);
size_t cy_head_wide_bwd_ws_bytes(long M, int C);
int cy_head_wide_fwd(const void* x, const float* w, const float* b, float* logits, long M, int C, int K, int dtype,
                     void* stream);
int cy_head_wide_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db, int accumulate,
                     long M, int C, int K, int dtype, void* ws, size_t ws_bytes, void* stream);

namespace {

constexpr int KMAX = 16;    // segmentation classes
constexpr int KWIDE = 128;  // stacked cluster-head outputs

template <typename T> __device__ __forceinline__ void load8h(const T* p, float* f) {
  if constexpr (sizeof(T) == 2) {
    Chunk<T>::unpack(ld16(p), f);
  } else {
    Chunk<float>::unpack(ld16(p), f);
    Chunk<float>::unpack(ld16(p + 4), f + 4);
  }
}
template <typename T> __device__ __forceinline__ void store8h(T* p, const float* f) {
  if constexpr (sizeof(T) == 2) {
    st16(p, Chunk<T>::pack(f));
  } else {
    st16(p, Chunk<float>::pack(f));
    st16(p + 4, Chunk<float>::pack(f + 4));
  }
}

// ---------------------------------------------------------------- 1x1 head forward
template <typename T, int KM>
__global__ void __launch_bounds__(256)
    head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                    const float* __restrict__ b, float* __restrict__ logits, long npix, int C,
                    int K) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [K][C] + [K]
  for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < K; i += 256) sw[K * C + i] = b ? b[i] : 0.f;
  __syncthreads();
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    float acc[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) acc[k] = k < K ? sw[K * C + k] : 0.f;
    for (int c0 = 0; c0 < C; c0 += 8) {
      float f[8];
      load8h<T>(x + p * C + c0, f);
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        if (k < K) {  // weights as two 16-byte LDS broadcasts per 8 FMAs
          const f32x4 wa = *reinterpret_cast<const f32x4*>(sw + k * C + c0);
          const f32x4 wb = *reinterpret_cast<const f32x4*>(sw + k * C + c0 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[k] = fmaf(f[j], wa[j], acc[k]);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[k] = fmaf(f[4 + j], wb[j], acc[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) logits[p * K + k] = acc[k];
  }
}

// dx[p][c] = sum_k dl[p][k] w[k][c]
template <typename T>
__global__ void __launch_bounds__(256)
    head_bwd_dx_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                       T* __restrict__ dx, long npix, int C, int K) {
  extern __shared__ float sw[];  // [K][C]
  for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
  __syncthreads();
  const int G = C / 8;
  const long total = npix * G;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int g = (int)(e % G);
    const long p = e / G;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float d = dl[p * K + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(d, sw[k * C + g * 8 + j], o[j]);
    }
    store8h<T>(dx + p * C + g * 8, o);
  }
}

// wide-K variant (stacked cluster heads): one thread per (pixel, 32-channel group) reads its
// dl row as 16-byte vectors (contiguous per thread) instead of K scalar loads shared by 4 lanes
template <typename T>
__global__ void __launch_bounds__(256)
    head_bwd_dx_wide_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                            T* __restrict__ dx, long npix, int C, int K) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [K][C]
  for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
  __syncthreads();
  const int G = C / 32;
  const long total = npix * G;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int g = (int)(e % G);
    const long p = e / G;
    float o[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) o[j] = 0.f;
    const float* dp = dl + p * K;
    for (int k0 = 0; k0 < K; k0 += 4) {
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(dp + k0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* wr = sw + (k0 + kk) * C + g * 32;
#pragma unroll
        for (int j4 = 0; j4 < 8; ++j4) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * j4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * j4 + j] = fmaf(d4[kk], wv[j], o[4 * j4 + j]);
        }
      }
    }
#pragma unroll
    for (int j8 = 0; j8 < 4; ++j8) store8h<T>(dx + p * C + g * 32 + j8 * 8, o + 8 * j8);
  }
}

// partial dW[k][c], db[k] per block:  ws[block][K*C + K]
// KG output channels are accumulated per pass over the pixels (x and dl are re-read K/KG times):
// 4 for the few segmentation classes, 16 for the 100 stacked cluster-head outputs.
template <typename T, int KG>
__global__ void __launch_bounds__(256)
    head_bwd_dw_kernel(const T* __restrict__ x, const float* __restrict__ dl,
                       float* __restrict__ ws, long npix, int C, int K) {
  extern __shared__ float sred[];  // [rows][G][K*8] then [rows][K]
  const int G = C / 8;
  const int gpp = G < 256 ? G : 256;
  const int rows = 256 / gpp;
  const int tid = threadIdx.x;
  const int g = tid % gpp, prow = tid / gpp;
  const bool active = tid < rows * gpp && g < G;
  const long per = (npix + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  float* wsb = ws + (size_t)blockIdx.x * (K * C + K);
  for (int k0 = 0; k0 < K; k0 += KG) {
    float acc[KG][8], accb[KG];
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      accb[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    }
    if (active) {
      for (long p = p0 + prow; p < p1; p += rows) {
        float f[8];
        load8h<T>(x + p * C + g * 8, f);
        float d[KG];
        if (k0 + KG <= K && (K & 3) == 0) {  // whole group in range: 16-byte loads of the dl row
#pragma unroll
          for (int k4 = 0; k4 < KG / 4; ++k4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dl + p * K + k0 + 4 * k4);
            d[4 * k4] = v[0], d[4 * k4 + 1] = v[1], d[4 * k4 + 2] = v[2], d[4 * k4 + 3] = v[3];
          }
        } else {
#pragma unroll
          for (int k = 0; k < KG; ++k) d[k] = (k0 + k < K) ? dl[p * K + k0 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KG; ++k) {
          accb[k] += d[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(d[k], f[j], acc[k][j]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < KG; kb += 4) {  // block reduction over the pixel rows, 4 outputs at a time
      __syncthreads();
      if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int j = 0; j < 8; ++j) sred[((prow * gpp + g) * 4 + k) * 8 + j] = acc[kb + k][j];
        }
        if (g == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) sred[rows * gpp * 32 + prow * 4 + k] = accb[kb + k];
        }
      }
      __syncthreads();
      for (int e = tid; e < 4 * gpp * 8; e += 256) {
        const int k = e / (gpp * 8), cl = e % (gpp * 8);
        const int gg = cl / 8, j = cl % 8;
        if (k0 + kb + k < K && cl < C) {
          float s = 0.f;
          for (int q = 0; q < rows; ++q) s += sred[((q * gpp + gg) * 4 + k) * 8 + j];
          wsb[(k0 + kb + k) * C + cl] = s;
        }
      }
      if (tid < 4 && k0 + kb + tid < K) {
        float s = 0.f;
        for (int q = 0; q < rows; ++q) s += sred[rows * gpp * 32 + q * 4 + tid];
        wsb[K * C + k0 + kb + tid] = s;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    head_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                           float* __restrict__ db, int nblk, int KC, int K, int accumulate) {
  // 4 outputs per block, 64 slices of the block partials each, fixed slice order
  __shared__ double sred[64][4];
  const int el = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int i = blockIdx.x * 4 + el;
  double s = 0.0;
  if (i < KC + K)
    for (int q = sl; q < nblk; q += 64) s += (double)ws[(size_t)q * (KC + K) + i];
  sred[sl][el] = s;
  __syncthreads();
  if (sl == 0 && i < KC + K) {
    double t = 0.0;
    for (int q = 0; q < 64; ++q) t += sred[q][el];
    if (i < KC) {
      if (dw) dw[i] = accumulate ? dw[i] + (float)t : (float)t;
    } else if (db) {
      db[i - KC] = accumulate ? db[i - KC] + (float)t : (float)t;
    }
  }
}

inline int head_dw_blocks(long npix) {
  long b = (npix + 255) / 256;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- softmax helpers
__device__ __forceinline__ void softmax_k(const float* z, float* p, int K) {
  float m = z[0];
#pragma unroll
  for (int k = 1; k < KMAX; ++k)
    if (k < K) m = fmaxf(m, z[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      p[k] = expf(z[k] - m);
      s += p[k];
    }
  const float inv = 1.f / s;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) p[k] *= inv;
}

__device__ __forceinline__ void load_logits(const float* l, long p, int K, float* z) {
  if (K == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(l + p * 4);
    z[0] = v[0], z[1] = v[1], z[2] = v[2], z[3] = v[3];
  } else {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) z[k] = l[p * K + k];
  }
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  return sh[0];
}

// ---------------------------------------------------------------- softmax + KL(one-hot)
__global__ void __launch_bounds__(256)
    softmax_kl_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                          double* __restrict__ partial, long npix, int K, float eps) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    float z[KMAX], pr[KMAX];
    load_logits(logits, p, K, z);
    softmax_k(z, pr, K);
    const int t = (int)target[p];
    float pt = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && k == t) pt = pr[k];
    // -t*log((p+eps)/(t+eps)) with t = 1 for the target class, 0 elsewhere
    acc += (double)(-logf((pt + eps) / (1.f + eps)));
  }
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
    mean_finalize_kernel(const double* __restrict__ partial, int nblk, double denom,
                         float* __restrict__ loss) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) loss[0] = (float)(tot / denom);
}

__global__ void __launch_bounds__(256)
    softmax_kl_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                          const float* __restrict__ gscale, float* __restrict__ dlogits, long npix,
                          int K, float eps) {
  const float gs = gscale[0] / (float)npix;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    float z[KMAX], pr[KMAX];
    load_logits(logits, p, K, z);
    softmax_k(z, pr, K);
    const int t = (int)target[p];
    float pt = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && k == t) pt = pr[k];
    const float coef = -gs * pt / (pt + eps);
    if (K == 4) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = coef * ((k == t ? 1.f : 0.f) - pr[k]);
      *reinterpret_cast<f32x4*>(dlogits + p * 4) = o;
    } else {
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) dlogits[p * K + k] = coef * ((k == t ? 1.f : 0.f) - pr[k]);
    }
  }
}

// ---------------------------------------------------------------- MSE of two softmaxes
__global__ void __launch_bounds__(256)
    softmax_mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                           double* __restrict__ partial, long npix, int K) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    float za[KMAX], zb[KMAX], pa[KMAX], pb[KMAX];
    load_logits(a, p, K, za);
    load_logits(b, p, K, zb);
    softmax_k(za, pa, K);
    softmax_k(zb, pb, K);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float d = pa[k] - pb[k];
        s = fmaf(d, d, s);
      }
    acc += (double)s;
  }
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
    softmax_mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                           const float* __restrict__ gscale, float* __restrict__ da,
                           float* __restrict__ db, long npix, int K) {
  const float gs = 2.f * gscale[0] / ((float)npix * (float)K);
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    float za[KMAX], zb[KMAX], pa[KMAX], pb[KMAX];
    load_logits(a, p, K, za);
    load_logits(b, p, K, zb);
    softmax_k(za, pa, K);
    softmax_k(zb, pb, K);
    // dL/dpa_k = gs*(pa_k-pb_k); dz = p * (g - sum_j g_j p_j)
    float dota = 0.f, dotb = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float d = pa[k] - pb[k];
        dota = fmaf(d, pa[k], dota);
        dotb = fmaf(d, pb[k], dotb);
      }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float d = pa[k] - pb[k];
        if (da) da[p * K + k] = gs * pa[k] * (d - dota);
        if (db) db[p * K + k] = -gs * pb[k] * (d - dotb);
      }
  }
}

// ---------------------------------------------------------------- dice counts
__global__ void __launch_bounds__(256)
    dice_counts_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                       unsigned long long* __restrict__ counts, int HW, int K) {
  // grid (blocks_per_sample, N)
  __shared__ unsigned int sc[KMAX * 2];
  const int n = blockIdx.y;
  if (threadIdx.x < KMAX * 2) sc[threadIdx.x] = 0u;
  __syncthreads();
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const long gp = (long)n * HW + p;
    float z[KMAX];
    load_logits(logits, gp, K, z);
    int best = 0;
    float bv = z[0];
#pragma unroll
    for (int k = 1; k < KMAX; ++k)
      if (k < K && z[k] > bv) {
        bv = z[k];
        best = k;
      }
    const int t = (int)target[gp];
    if (best == t) atomicAdd(&sc[best * 2 + 0], 1u);
    atomicAdd(&sc[best * 2 + 1], 1u);
    if (t >= 0 && t < K) atomicAdd(&sc[t * 2 + 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x < K * 2 && sc[threadIdx.x])
    atomicAdd(&counts[(size_t)n * K * 2 + threadIdx.x], (unsigned long long)sc[threadIdx.x]);
}

inline int loss_blocks(long npix) {
  long b = (npix + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int cy_head1x1_fwd(const void* x, const float* w, const float* b, float* logits, long npix, int C,
                   int K, int x_dtype, void* stream) {
  if (!x || !w || !logits || npix <= 0) return CY_ERR_ARG;
  if (C % 8 || K < 1 || K > KWIDE || (size_t)(K * C + K) * 4 > 60000) return CY_ERR_SHAPE;
  if (x_dtype != CY_BF16 && x_dtype != CY_F32 && x_dtype != CY_F16) return CY_ERR_DTYPE;
  // stacked cluster sub-heads over 32 / 64 channels: f32 matrix cores, the logits tile written in contiguous runs
  // (the VALU kernel below stores one float per lane at a 4 K-byte stride: 0.8 TB/s at K = 100)
  if (cy_head_wide_ok(C, K)) return cy_head_wide_fwd(x, w, b, logits, npix, C, K, x_dtype, stream);
  hipStream_t st = (hipStream_t)stream;
  const int grid = loss_blocks(npix) * 2;
  const size_t smem = (size_t)(K * C + K) * sizeof(float);
  const bool h = x_dtype == CY_BF16;
  if (K <= KMAX) {  // segmentation classes
    if (h)
      hipLaunchKernelGGL((head_fwd_kernel<bf16, KMAX>), dim3(grid), dim3(256), smem, st,
                         (const bf16*)x, w, b, logits, npix, C, K);
    else if (x_dtype == CY_F16)
      hipLaunchKernelGGL((head_fwd_kernel<f16, KMAX>), dim3(grid), dim3(256), smem, st,
                         (const f16*)x, w, b, logits, npix, C, K);
    else
      hipLaunchKernelGGL((head_fwd_kernel<float, KMAX>), dim3(grid), dim3(256), smem, st,
                         (const float*)x, w, b, logits, npix, C, K);
  } else {  // stacked cluster sub-heads (DenseClusterHead: 5 x 20 outputs)
    if (h)
      hipLaunchKernelGGL((head_fwd_kernel<bf16, KWIDE>), dim3(grid), dim3(256), smem, st,
                         (const bf16*)x, w, b, logits, npix, C, K);
    else if (x_dtype == CY_F16)
      hipLaunchKernelGGL((head_fwd_kernel<f16, KWIDE>), dim3(grid), dim3(256), smem, st,
                         (const f16*)x, w, b, logits, npix, C, K);
    else
      hipLaunchKernelGGL((head_fwd_kernel<float, KWIDE>), dim3(grid), dim3(256), smem, st,
                         (const float*)x, w, b, logits, npix, C, K);
  }
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_head1x1_bwd_ws_bytes(long npix, int C, int K) {
  if (cy_head_wide_ok(C, K)) return cy_head_wide_bwd_ws_bytes(npix, C);
  return (size_t)head_dw_blocks(npix) * (K * C + K) * sizeof(float);
}

static int head1x1_bwd_impl(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                            float* db, int accumulate, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                            void* stream);

int cy_head1x1_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                   float* db, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                   void* stream) {
  return head1x1_bwd_impl(x, w, dlogits, dx, dw, db, 0, npix, C, K, x_dtype, ws, ws_bytes, stream);
}

int cy_head1x1_bwd_into(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                        float* db, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                        void* stream) {
  return head1x1_bwd_impl(x, w, dlogits, dx, dw, db, 1, npix, C, K, x_dtype, ws, ws_bytes, stream);
}

static int head1x1_bwd_impl(const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                            float* db, int accumulate, long npix, int C, int K, int x_dtype, void* ws, size_t ws_bytes,
                            void* stream) {
  if (!x || !w || !dlogits || npix <= 0) return CY_ERR_ARG;
  if (C % 8 || K < 1 || K > KWIDE || (size_t)(K * C + K) * 4 > 60000) return CY_ERR_SHAPE;
  if (x_dtype != CY_BF16 && x_dtype != CY_F32 && x_dtype != CY_F16) return CY_ERR_DTYPE;
  if (cy_head_wide_ok(C, K))
    return cy_head_wide_bwd(x, w, dlogits, dx, dw, db, accumulate, npix, C, K, x_dtype, ws, ws_bytes, stream);
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    const long total = npix * (C / 8);
    long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    const size_t smem = (size_t)K * C * sizeof(float);
    if (K > KMAX && K % 4 == 0 && C % 32 == 0) {
      long bw = (npix * (C / 32) + 255) / 256;
      if (bw > 8192) bw = 8192;
      if (x_dtype == CY_BF16)
        hipLaunchKernelGGL(head_bwd_dx_wide_kernel<bf16>, dim3((int)bw), dim3(256), smem, st, dlogits,
                           w, (bf16*)dx, npix, C, K);
      else if (x_dtype == CY_F16)
        hipLaunchKernelGGL(head_bwd_dx_wide_kernel<f16>, dim3((int)bw), dim3(256), smem, st, dlogits,
                           w, (f16*)dx, npix, C, K);
      else
        hipLaunchKernelGGL(head_bwd_dx_wide_kernel<float>, dim3((int)bw), dim3(256), smem, st, dlogits,
                           w, (float*)dx, npix, C, K);
    } else if (x_dtype == CY_BF16)
      hipLaunchKernelGGL(head_bwd_dx_kernel<bf16>, dim3((int)b), dim3(256), smem, st, dlogits, w,
                         (bf16*)dx, npix, C, K);
    else if (x_dtype == CY_F16)
      hipLaunchKernelGGL(head_bwd_dx_kernel<f16>, dim3((int)b), dim3(256), smem, st, dlogits, w,
                         (f16*)dx, npix, C, K);
    else
      hipLaunchKernelGGL(head_bwd_dx_kernel<float>, dim3((int)b), dim3(256), smem, st, dlogits, w,
                         (float*)dx, npix, C, K);
    CY_CHECK_LAUNCH();
  }
  if (dw || db) {
    if (!ws || ws_bytes < cy_head1x1_bwd_ws_bytes(npix, C, K)) return CY_ERR_WORKSPACE;
    const int nblk = head_dw_blocks(npix);
    const int G = C / 8;
    const int gpp = G < 256 ? G : 256;
    if (G > 256) return CY_ERR_SHAPE;
    const int rows = 256 / gpp;
    const size_t smem = (size_t)(rows * gpp * 32 + rows * 4) * sizeof(float);
    const bool h = x_dtype == CY_BF16;
    if (K <= KMAX) {
      if (h)
        hipLaunchKernelGGL((head_bwd_dw_kernel<bf16, 4>), dim3(nblk), dim3(256), smem, st,
                           (const bf16*)x, dlogits, (float*)ws, npix, C, K);
      else if (x_dtype == CY_F16)
        hipLaunchKernelGGL((head_bwd_dw_kernel<f16, 4>), dim3(nblk), dim3(256), smem, st,
                           (const f16*)x, dlogits, (float*)ws, npix, C, K);
      else
        hipLaunchKernelGGL((head_bwd_dw_kernel<float, 4>), dim3(nblk), dim3(256), smem, st,
                           (const float*)x, dlogits, (float*)ws, npix, C, K);
    } else {
      if (h)
        hipLaunchKernelGGL((head_bwd_dw_kernel<bf16, 16>), dim3(nblk), dim3(256), smem, st,
                           (const bf16*)x, dlogits, (float*)ws, npix, C, K);
      else if (x_dtype == CY_F16)
        hipLaunchKernelGGL((head_bwd_dw_kernel<f16, 16>), dim3(nblk), dim3(256), smem, st,
                           (const f16*)x, dlogits, (float*)ws, npix, C, K);
      else
        hipLaunchKernelGGL((head_bwd_dw_kernel<float, 16>), dim3(nblk), dim3(256), smem, st,
                           (const float*)x, dlogits, (float*)ws, npix, C, K);
    }
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3(cy_cdiv(K * C + K, 4)), dim3(256), 0, st,
                       (const float*)ws, dw, db, nblk, K * C, K, accumulate);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

size_t cy_softmax_kl_ws_bytes(long npix) { return (size_t)loss_blocks(npix) * sizeof(double); }

int cy_softmax_kl_fwd(const float* logits, const int64_t* target, float* loss, long npix, int K,
                      float eps, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !target || !loss || !ws || npix <= 0) return CY_ERR_ARG;
  if (K < 1 || K > KMAX) return CY_ERR_SHAPE;
  if (ws_bytes < cy_softmax_kl_ws_bytes(npix)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = loss_blocks(npix);
  hipLaunchKernelGGL(softmax_kl_fwd_kernel, dim3(nblk), dim3(256), 0, st, logits, target,
                     (double*)ws, npix, K, eps);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk,
                     (double)npix, loss);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_softmax_kl_bwd(const float* logits, const int64_t* target, const float* gscale,
                      float* dlogits, long npix, int K, float eps, void* stream) {
  if (!logits || !target || !gscale || !dlogits || npix <= 0) return CY_ERR_ARG;
  if (K < 1 || K > KMAX) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_kl_bwd_kernel, dim3(loss_blocks(npix) * 2), dim3(256), 0,
                     (hipStream_t)stream, logits, target, gscale, dlogits, npix, K, eps);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_softmax_mse_ws_bytes(long npix) { return (size_t)loss_blocks(npix) * sizeof(double); }

int cy_softmax_mse_fwd(const float* a, const float* b, float* loss, long npix, int K, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!a || !b || !loss || !ws || npix <= 0) return CY_ERR_ARG;
  if (K < 1 || K > KMAX) return CY_ERR_SHAPE;
  if (ws_bytes < cy_softmax_mse_ws_bytes(npix)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = loss_blocks(npix);
  hipLaunchKernelGGL(softmax_mse_fwd_kernel, dim3(nblk), dim3(256), 0, st, a, b, (double*)ws, npix,
                     K);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk,
                     (double)npix * (double)K, loss);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_softmax_mse_bwd(const float* a, const float* b, const float* gscale, float* da, float* db,
                       long npix, int K, void* stream) {
  if (!a || !b || !gscale || (!da && !db) || npix <= 0) return CY_ERR_ARG;
  if (K < 1 || K > KMAX) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_mse_bwd_kernel, dim3(loss_blocks(npix) * 2), dim3(256), 0,
                     (hipStream_t)stream, a, b, gscale, da, db, npix, K);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_dice_counts(const float* logits, const int64_t* target, int64_t* counts, int N, int HW,
                   int K, void* stream) {
  if (!logits || !target || !counts || N <= 0 || HW <= 0) return CY_ERR_ARG;
  if (K < 1 || K > KMAX) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, (size_t)N * K * 2 * sizeof(int64_t), st) != hipSuccess)
    return CY_ERR_LAUNCH;
  int bps = (HW + 255) / 256;
  if (bps > 64) bps = 64;
  hipLaunchKernelGGL(dice_counts_kernel, dim3(bps, N), dim3(256), 0, st, logits, target,
                     (unsigned long long*)counts, HW, K);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
