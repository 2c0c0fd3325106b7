// GroupNorm + SiLU after a biased 3x3 convolution, and bilinear image resize.
//   UNet2.Block: Conv2d(3x3, bias) -> GroupNorm(8, C) -> SiLU      contrastyou/arch/unet2.py:208-224
//   F.interpolate(image, size, mode="bilinear")                    semi_seg/hooks/cc.py:132,
//                                                                  semi_seg/hooks/ccblock.py:300
//
// The convolution itself runs on the implicit-GEMM kernels of cy_conv3x3.hip WITHOUT the bias;
// the bias is folded into the normalisation here: u = y + b_c, x^ = (u - mean_g) * rstd_g,
// v = gamma_c x^ + beta_c, z = v * sigmoid(v).  All kernels stream NHWC tensors once
// (HBM-bound); statistics are two-stage, fixed-order reductions.
#include "cy_common.h"

namespace {

constexpr int GN_SPLIT = 32;  // pixel splits per image in the reduction kernels

template <typename T> __device__ __forceinline__ void load8(const T* p, float* f) {
  if constexpr (sizeof(T) == 2) {
    Chunk<T>::unpack(ld16(p), f);
  } else {
    Chunk<float>::unpack(ld16(p), f);
    Chunk<float>::unpack(ld16(p + 4), f + 4);
  }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f) {
  if constexpr (sizeof(T) == 2) {
    st16(p, Chunk<T>::pack(f));
  } else {
    st16(p, Chunk<float>::pack(f));
    st16(p + 4, Chunk<float>::pack(f + 4));
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// Generic per-(image, split) channel reduction of NV values per element.
// grid (GN_SPLIT, N); thread layout: gpp 8-channel groups x rows pixel lanes.
// F(f_y[8], f_d[8], c0, acc[NV][8]) accumulates.
template <typename T, int NV, typename F>
__device__ __forceinline__ void channel_reduce(const T* __restrict__ y, int ldy,
                                               const T* __restrict__ d, int ldd, int HW, int C,
                                               float* __restrict__ part, F f) {
  extern __shared__ float sred[];  // [rows][gpp][NV*8]
  const int G8 = C / 8;
  const int gpp = G8 < 256 ? G8 : 256;
  const int rows = 256 / gpp;
  const int tid = threadIdx.x;
  const int g = tid % gpp, prow = tid / gpp;
  const bool active = prow < rows;
  const int n = blockIdx.y;
  const int per = (HW + GN_SPLIT - 1) / GN_SPLIT;
  const int q0 = blockIdx.x * per, q1 = min(HW, q0 + per);
  float* dst = part + ((size_t)n * GN_SPLIT + blockIdx.x) * NV * C;
  const int gg = g;  // C <= 2048 (checked by the host) => every 8-channel group has its own thread
  float acc[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[v][j] = 0.f;
  if (active) {
    // four pixels per round, every load requested before the first use (a thread walks a dozen pixels or more: the
    // dependent form paid one memory latency per pixel); same order of additions
    int q = q0 + prow;
    for (; q + 3 * rows < q1; q += 4 * rows) {
      float fy[4][8], fd[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long pix = (long)n * HW + q + u * rows;
        load8<T>(y + pix * ldy + gg * 8, fy[u]);
        if (d) load8<T>(d + pix * ldd + gg * 8, fd[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) f(fy[u], fd[u], n, gg * 8, acc);
    }
    for (; q < q1; q += rows) {
      const long pix = (long)n * HW + q;
      float fy[8], fd[8];
      load8<T>(y + pix * ldy + gg * 8, fy);
      if (d) load8<T>(d + pix * ldd + gg * 8, fd);
      f(fy, fd, n, gg * 8, acc);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) sred[((prow * gpp + g) * NV + v) * 8 + j] = acc[v][j];
  }
  __syncthreads();
  for (int e = tid; e < NV * 8 * gpp; e += 256) {
    const int gl = e / (NV * 8), r = e - gl * NV * 8, v = r / 8, j = r - v * 8;
    float s = 0.f;
    for (int q = 0; q < rows; ++q) s += sred[((q * gpp + gl) * NV + v) * 8 + j];
    dst[v * C + gl * 8 + j] = s;
  }
}

// ---------------------------------------------------------------- forward statistics
template <typename T>
__global__ void __launch_bounds__(256)
    gn_stats_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ bias, int HW, int C,
                    float* __restrict__ part) {
  channel_reduce<T, 2>(y, ldy, (const T*)nullptr, 0, HW, C, part,
                       [&](const float* fy, const float*, int, int c0, float(*acc)[8]) {
#pragma unroll
                         for (int j = 0; j < 8; ++j) {
                           const float u = fy[j] + (bias ? bias[c0 + j] : 0.f);
                           acc[0][j] += u;
                           acc[1][j] = fmaf(u, u, acc[1][j]);
                         }
                       });
}

// one block per (n, group): mean / rstd over the group's channels and the pixel splits (fixed-shape tree), then the
// per-(n, channel) coefficients of  v = a * y + b  (a = gamma * rstd, b = beta + gamma * (bias - mean) * rstd) that the
// apply kernel reads -- it used to divide, and to fetch five scalars, per ELEMENT (45 us for a 26 MB map), and this
// kernel was one thread per (n, group) walking 512 dependent loads (37 us)
__global__ void __launch_bounds__(256)
    gn_stats_finalize_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                             const float* __restrict__ gamma, const float* __restrict__ beta,
                             float* __restrict__ mean_rstd, float* __restrict__ xcoef, int N, int C, int G, int HW,
                             float eps, const float* __restrict__ mod_s, const float* __restrict__ mod_t) {
  __shared__ double s1s[256], s2s[256];
  const int n = blockIdx.x / G, g = blockIdx.x - n * G, cg = C / G;
  const int tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int e = tid; e < GN_SPLIT * cg; e += 256) {
    const int sp = e / cg, c = g * cg + (e - sp * cg);
    const float* p = part + ((size_t)n * GN_SPLIT + sp) * 2 * C;
    s1 += (double)p[c];
    s2 += (double)p[C + c];
  }
  s1s[tid] = s1, s2s[tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) s1s[tid] += s1s[tid + o], s2s[tid] += s2s[tid + o];
    __syncthreads();
  }
  const double m = (double)cg * HW;
  const double mean = s1s[0] / m;
  double var = s2s[0] / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0) {
    mean_rstd[2 * (n * G + g)] = fmean;
    mean_rstd[2 * (n * G + g) + 1] = frstd;
  }
  for (int cl = tid; cl < cg; cl += 256) {
    const int c = g * cg + cl;
    // time-embedding modulation (ResnetBlock, arch/unet2.py:216-220: v * (scale + 1) + shift after the norm) = a
    // per-(image, channel) affine: gamma' = gamma (1 + s), beta' = beta (1 + s) + t
    float gm = gamma[c], bt = beta[c];
    if (mod_s) {
      const float s1 = 1.f + mod_s[(size_t)n * C + c];
      gm *= s1;
      bt = fmaf(bt, s1, mod_t[(size_t)n * C + c]);
    }
    const float a = gm * frstd;
    xcoef[((size_t)n * 2 + 0) * C + c] = a;
    xcoef[((size_t)n * 2 + 1) * C + c] = fmaf((bias ? bias[c] : 0.f) - fmean, a, bt);
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

template <typename T>
__global__ void __launch_bounds__(256)
    gn_silu_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ xcoef, T* __restrict__ out,
                         int ldo, int N, int HW, int C) {
  const int G8 = C / 8;
  const long total = (long)N * HW * G8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int g8 = (int)(i % G8);
    const long pix = i / G8;
    const int n = (int)(pix / HW);
    float f[8];
    load8<T>(y + pix * ldy + g8 * 8, f);
    const float* ca = xcoef + ((size_t)n * 2 + 0) * C + g8 * 8;
    const float* cb = xcoef + ((size_t)n * 2 + 1) * C + g8 * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ca), a1 = *reinterpret_cast<const f32x4*>(ca + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(cb), b1 = *reinterpret_cast<const f32x4*>(cb + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = fmaf(a0[j], f[j], b0[j]), v1 = fmaf(a1[j], f[4 + j], b1[j]);
      f[j] = v0 * sigmoidf_(v0);
      f[4 + j] = v1 * sigmoidf_(v1);
    }
    store8<T>(out + pix * ldo + g8 * 8, f);
  }
}

// ---------------------------------------------------------------- backward
// xt[n][0..3][C] = (a, b, ah, bh) with  v = a*y + b  (the GroupNorm output before SiLU) and  x^ = ah*y + bh, from the
// statistics the forward pass saved: the two big backward kernels read four coefficient rows instead of dividing and
// fetching five scalars per element
__global__ void __launch_bounds__(256)
    gn_bwd_coef_kernel(const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                       const float* __restrict__ mean_rstd, float* __restrict__ xt, int N, int C, int G,
                       const float* __restrict__ mod_s, const float* __restrict__ mod_t) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * C) return;
  const int n = e / C, c = e - n * C;
  const float* mr = mean_rstd + 2 * ((size_t)n * G + c / (C / G));
  const float rstd = mr[1], bh = ((bias ? bias[c] : 0.f) - mr[0]) * rstd;
  float* t = xt + (size_t)n * 4 * C + c;
  float gm = gamma[c], bt = beta[c];
  if (mod_s) {
    const float s1 = 1.f + mod_s[e];
    gm *= s1;
    bt = fmaf(bt, s1, mod_t[e]);
  }
  t[0] = gm * rstd;
  t[C] = fmaf(gm, bh, bt);
  t[2 * C] = rstd;
  t[3 * C] = bh;
}

// per (n, split, channel): A = sum dv, B = sum dv*x^, X = sum x^
template <typename T>
__global__ void __launch_bounds__(256)
    gn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dz, int ldd,
                         const float* __restrict__ xt, int HW, int C, float* __restrict__ part) {
  channel_reduce<T, 3>(y, ldy, dz, ldd, HW, C, part,
                       [&](const float* fy, const float* fd, int n, int c0, float(*acc)[8]) {
                         const float* t = xt + (size_t)n * 4 * C + c0;
#pragma unroll
                         for (int j = 0; j < 8; ++j) {
                           const float xh = fmaf(t[2 * C + j], fy[j], t[3 * C + j]);
                           const float v = fmaf(t[j], fy[j], t[C + j]);
                           const float sg = sigmoidf_(v);
                           const float dv = fd[j] * (sg + v * sg * (1.f - sg));
                           acc[0][j] += dv;
                           acc[1][j] = fmaf(dv, xh, acc[1][j]);
                           acc[2][j] += xh;
                         }
                       });
}

// one block per (n, g): coefficients  du = k1*dv + k0 + k2*x^  and per-image parameter grads
__global__ void __launch_bounds__(64)
    gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                           const float* __restrict__ mean_rstd, float* __restrict__ coef,
                           float* __restrict__ pgrad, int N, int C, int G, int HW, const float* __restrict__ mod_s) {
  __shared__ double sA[64], sB[64];
  __shared__ double sa[2048 / 8], sb[2048 / 8], sx[2048 / 8];  // per-channel sums (cg <= 256)
  const int n = blockIdx.x / G, g = blockIdx.x - n * G, cg = C / G;
  const int tid = threadIdx.x;
  double tA = 0.0, tB = 0.0;
  for (int cl = tid; cl < cg; cl += 64) {
    const int c = g * cg + cl;
    double a = 0.0, b = 0.0, x = 0.0;
    for (int sp = 0; sp < GN_SPLIT; ++sp) {
      const float* p = part + ((size_t)n * GN_SPLIT + sp) * 3 * C;
      a += (double)p[c];
      b += (double)p[C + c];
      x += (double)p[2 * C + c];
    }
    sa[cl] = a, sb[cl] = b, sx[cl] = x;
    const double gme = (double)(mod_s ? gamma[c] * (1.f + mod_s[(size_t)n * C + c]) : gamma[c]);  // (the effective gamma)
    tA += gme * a;
    tB += gme * b;
  }
  sA[tid] = tA, sB[tid] = tB;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (tid < o) sA[tid] += sA[tid + o], sB[tid] += sB[tid + o];
    __syncthreads();
  }
  const double m = (double)cg * HW;
  const double m1 = sA[0] / m, m2 = sB[0] / m;
  const double rstd = mean_rstd[2 * (n * G + g) + 1];
  for (int cl = tid; cl < cg; cl += 64) {
    const int c = g * cg + cl;
    float* k = coef + (size_t)n * 3 * C + c;  // [n][k0 | k1 | k2][C]
    const double gme = (double)(mod_s ? gamma[c] * (1.f + mod_s[(size_t)n * C + c]) : gamma[c]);
    k[0] = (float)(-rstd * m1);
    k[C] = (float)(rstd * gme);
    k[2 * C] = (float)(-rstd * m2);
    float* pg = pgrad + ((size_t)n * C + c) * 3;
    pg[0] = (float)sb[cl];                                                       // dgamma_n (of the effective gamma)
    pg[1] = (float)sa[cl];                                                       // dbeta_n  (of the effective beta)
    pg[2] = (float)(rstd * (gme * sa[cl] - HW * m1 - m2 * sx[cl]));               // dbias_n
  }
}

__global__ void __launch_bounds__(256)
    gn_param_grad_kernel(const float* __restrict__ pgrad, float* __restrict__ dgamma,
                         float* __restrict__ dbeta, float* __restrict__ dbias, int N, int C,
                         int accumulate, const float* __restrict__ mod_s, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float* __restrict__ dmod_s, float* __restrict__ dmod_t) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s[3] = {0.0, 0.0, 0.0};
  for (int n = 0; n < N; ++n) {
    // with the modulation v = (gamma x^ + beta)(1 + s) + t: dgamma = sum_n (1 + s) dgamma'_n, dbeta likewise,
    // ds_n = gamma dgamma'_n + beta dbeta'_n, dt_n = dbeta'_n
    const float* pg = pgrad + ((size_t)n * C + c) * 3;
    const double w = mod_s ? 1.0 + (double)mod_s[(size_t)n * C + c] : 1.0;
    s[0] += w * (double)pg[0];
    s[1] += w * (double)pg[1];
    s[2] += (double)pg[2];
    if (dmod_s) dmod_s[(size_t)n * C + c] = gamma[c] * pg[0] + beta[c] * pg[1];
    if (dmod_t) dmod_t[(size_t)n * C + c] = pg[1];
  }
  float* dst[3] = {dgamma, dbeta, dbias};
  for (int v = 0; v < 3; ++v)
    if (dst[v]) dst[v][c] = accumulate ? dst[v][c] + (float)s[v] : (float)s[v];
}

template <typename T>
__global__ void __launch_bounds__(256)
    gn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dz, int ldd,
                        const float* __restrict__ xt, const float* __restrict__ coef, T* __restrict__ du, int ldu,
                        int N, int HW, int C) {
  const int G8 = C / 8;
  const long total = (long)N * HW * G8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int g8 = (int)(i % G8);
    const long pix = i / G8;
    const int n = (int)(pix / HW);
    float fy[8], fd[8];
    load8<T>(y + pix * ldy + g8 * 8, fy);
    load8<T>(dz + pix * ldd + g8 * 8, fd);
    const float* t = xt + (size_t)n * 4 * C + g8 * 8;
    const float* k = coef + (size_t)n * 3 * C + g8 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = fmaf(t[2 * C + j], fy[j], t[3 * C + j]);
      const float v = fmaf(t[j], fy[j], t[C + j]);
      const float sg = sigmoidf_(v);
      const float dv = fd[j] * (sg + v * sg * (1.f - sg));
      fy[j] = fmaf(k[C + j], dv, fmaf(k[2 * C + j], xh, k[j]));
    }
    store8<T>(du + pix * ldu + g8 * 8, fy);
  }
}

// ---------------------------------------------------------------- bilinear resize (align_corners=False)
template <typename T>
__global__ void __launch_bounds__(256)
    bilinear_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int H, int W, int C, int h,
                    int w) {
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  const long total = (long)N * h * w * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % C);
    long t = e / C;
    const int ox = (int)(t % w);
    t /= w;
    const int oy = (int)(t % h), n = (int)(t / h);
    float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f);
    float fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const T* b = x + (long)n * H * W * C + c;
    const float v00 = to_f32<T>(b[((long)y0 * W + x0) * C]), v01 = to_f32<T>(b[((long)y0 * W + x1) * C]);
    const float v10 = to_f32<T>(b[((long)y1 * W + x0) * C]), v11 = to_f32<T>(b[((long)y1 * W + x1) * C]);
    const float top = v00 * (1.f - lx) + v01 * lx, bot = v10 * (1.f - lx) + v11 * lx;
    out[e] = from_f32<T>(top * (1.f - ly) + bot * ly);
  }
}

inline size_t reduce_smem(int C, int NV) {
  const int G8 = C / 8, gpp = G8 < 256 ? G8 : 256, rows = 256 / gpp;
  return (size_t)rows * gpp * NV * 8 * sizeof(float);
}

}  // namespace

extern "C" {

size_t cy_gn_ws_bytes(int N, int C) {
  /* reduction partials [N][SPLIT][3][C] + coef [N][C][3] + pgrad [N][C][3] */
  return ((size_t)N * GN_SPLIT * 3 * C + 10 * (size_t)N * C) * sizeof(float);
}

static int gn_silu_fwd_impl(const void* y, int ldy, const float* bias, const float* gamma, const float* beta,
                            const float* mod_s, const float* mod_t, void* out, int ldo, float* mean_rstd, int N, int HW,
                            int C, int G, float eps, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if ((mod_s == nullptr) != (mod_t == nullptr)) return CY_ERR_ARG;
  if (!y || !gamma || !beta || !out || !mean_rstd || N <= 0 || HW <= 0) return CY_ERR_ARG;
  if (C % 8 || G <= 0 || C % G || ldy % 8 || ldo % 8 || ldy < C || ldo < C || C / G > 256 || C > 2048)
    return CY_ERR_SHAPE;
  if (dtype != CY_BF16 && dtype != CY_F32 && dtype != CY_F16) return CY_ERR_DTYPE;
  if (!ws || ws_bytes < cy_gn_ws_bytes(N, C)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  float* xcoef = part + (size_t)N * GN_SPLIT * 3 * C;  // [N][2][C] behind the partial sums (cy_gn_ws_bytes has 6*N*C there)
  const size_t smem = reduce_smem(C, 2);
  const int grid = grid_for((long)N * HW * (C / 8));
  if (dtype == CY_BF16) {
    hipLaunchKernelGGL(gn_stats_kernel<bf16>, dim3(GN_SPLIT, N), dim3(256), smem, st, (const bf16*)y,
                       ldy, bias, HW, C, part);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(N * G), dim3(256), 0, st, (const float*)part, bias, gamma, beta,
                       mean_rstd, xcoef, N, C, G, HW, eps, mod_s, mod_t);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_silu_apply_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)y, ldy,
                       (const float*)xcoef, (bf16*)out, ldo, N, HW, C);
  } else if (dtype == CY_F16) {
    hipLaunchKernelGGL(gn_stats_kernel<f16>, dim3(GN_SPLIT, N), dim3(256), smem, st, (const f16*)y,
                       ldy, bias, HW, C, part);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(N * G), dim3(256), 0, st, (const float*)part, bias, gamma, beta,
                       mean_rstd, xcoef, N, C, G, HW, eps, mod_s, mod_t);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_silu_apply_kernel<f16>, dim3(grid), dim3(256), 0, st, (const f16*)y, ldy,
                       (const float*)xcoef, (f16*)out, ldo, N, HW, C);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(GN_SPLIT, N), dim3(256), smem, st,
                       (const float*)y, ldy, bias, HW, C, part);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(N * G), dim3(256), 0, st, (const float*)part, bias, gamma, beta,
                       mean_rstd, xcoef, N, C, G, HW, eps, mod_s, mod_t);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_silu_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)y,
                       ldy, (const float*)xcoef, (float*)out, ldo, N, HW, C);
  }
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_gn_silu_fwd(const void* y, int ldy, const float* bias, const float* gamma, const float* beta,
                   void* out, int ldo, float* mean_rstd, int N, int HW, int C, int G, float eps,
                   int dtype, void* ws, size_t ws_bytes, void* stream) {
  return gn_silu_fwd_impl(y, ldy, bias, gamma, beta, nullptr, nullptr, out, ldo, mean_rstd, N, HW, C, G, eps, dtype, ws,
                          ws_bytes, stream);
}

int cy_gn_silu_mod_fwd(const void* y, int ldy, const float* bias, const float* gamma, const float* beta,
                       const float* mod_scale, const float* mod_shift, void* out, int ldo, float* mean_rstd, int N,
                       int HW, int C, int G, float eps, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!mod_scale || !mod_shift) return CY_ERR_ARG;
  return gn_silu_fwd_impl(y, ldy, bias, gamma, beta, mod_scale, mod_shift, out, ldo, mean_rstd, N, HW, C, G, eps, dtype,
                          ws, ws_bytes, stream);
}

static int gn_silu_bwd_impl(const void* y, int ldy, const void* dz, int ldd, const float* bias,
                            const float* gamma, const float* beta, const float* mod_s, const float* mod_t,
                            const float* mean_rstd, void* du, int ldu, float* dgamma, float* dbeta, float* dbias,
                            float* dmod_s, float* dmod_t, int accumulate, int N, int HW, int C, int G, int dtype,
                            void* ws, size_t ws_bytes, void* stream) {
  if ((mod_s == nullptr) != (mod_t == nullptr) || (!mod_s && (dmod_s || dmod_t))) return CY_ERR_ARG;
  if (!y || !dz || !gamma || !beta || !mean_rstd || !du || N <= 0 || HW <= 0) return CY_ERR_ARG;
  if (C % 8 || G <= 0 || C % G || ldy % 8 || ldd % 8 || ldu % 8 || ldy < C || ldd < C || ldu < C ||
      C / G > 256 || C > 2048)
    return CY_ERR_SHAPE;
  if (dtype != CY_BF16 && dtype != CY_F32 && dtype != CY_F16) return CY_ERR_DTYPE;
  if (!ws || ws_bytes < cy_gn_ws_bytes(N, C)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  float* coef = part + (size_t)N * GN_SPLIT * 3 * C;
  float* pgrad = coef + 3 * (size_t)N * C;
  float* xt = pgrad + 3 * (size_t)N * C;  // [N][4][C]
  const size_t smem = reduce_smem(C, 3);
  const int grid = grid_for((long)N * HW * (C / 8));
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(cy_cdiv((long)N * C, 256)), dim3(256), 0, st, bias, gamma, beta,
                     mean_rstd, xt, N, C, G, mod_s, mod_t);
  CY_CHECK_LAUNCH();
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16>, dim3(GN_SPLIT, N), dim3(256), smem, st,
                       (const bf16*)y, ldy, (const bf16*)dz, ldd, (const float*)xt, HW, C, part);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<f16>, dim3(GN_SPLIT, N), dim3(256), smem, st,
                       (const f16*)y, ldy, (const f16*)dz, ldd, (const float*)xt, HW, C, part);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(GN_SPLIT, N), dim3(256), smem, st,
                       (const float*)y, ldy, (const float*)dz, ldd, (const float*)xt, HW, C, part);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N * G), dim3(64), 0, st, (const float*)part, gamma,
                     mean_rstd, coef, pgrad, N, C, G, HW, mod_s);
  CY_CHECK_LAUNCH();
  if (dgamma || dbeta || dbias || dmod_s || dmod_t) {
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(cy_cdiv(C, 256)), dim3(256), 0, st,
                       (const float*)pgrad, dgamma, dbeta, dbias, N, C, accumulate, mod_s, gamma, beta, dmod_s, dmod_t);
    CY_CHECK_LAUNCH();
  }
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)y, ldy,
                       (const bf16*)dz, ldd, (const float*)xt, (const float*)coef, (bf16*)du, ldu, N, HW, C);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<f16>, dim3(grid), dim3(256), 0, st, (const f16*)y, ldy,
                       (const f16*)dz, ldd, (const float*)xt, (const float*)coef, (f16*)du, ldu, N, HW, C);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)y, ldy,
                       (const float*)dz, ldd, (const float*)xt, (const float*)coef, (float*)du, ldu, N, HW, C);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_gn_silu_bwd(const void* y, int ldy, const void* dz, int ldd, const float* bias,
                   const float* gamma, const float* beta, const float* mean_rstd, void* du, int ldu,
                   float* dgamma, float* dbeta, float* dbias, int accumulate, int N, int HW, int C,
                   int G, int dtype, void* ws, size_t ws_bytes, void* stream) {
  return gn_silu_bwd_impl(y, ldy, dz, ldd, bias, gamma, beta, nullptr, nullptr, mean_rstd, du, ldu, dgamma, dbeta, dbias,
                          nullptr, nullptr, accumulate, N, HW, C, G, dtype, ws, ws_bytes, stream);
}

int cy_gn_silu_mod_bwd(const void* y, int ldy, const void* dz, int ldd, const float* bias, const float* gamma,
                       const float* beta, const float* mod_scale, const float* mod_shift, const float* mean_rstd,
                       void* du, int ldu, float* dgamma, float* dbeta, float* dbias, float* dmod_scale,
                       float* dmod_shift, int accumulate, int N, int HW, int C, int G, int dtype, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!mod_scale || !mod_shift) return CY_ERR_ARG;
  return gn_silu_bwd_impl(y, ldy, dz, ldd, bias, gamma, beta, mod_scale, mod_shift, mean_rstd, du, ldu, dgamma, dbeta,
                          dbias, dmod_scale, dmod_shift, accumulate, N, HW, C, G, dtype, ws, ws_bytes, stream);
}

int cy_bilinear_fwd(const void* x, void* out, int N, int H, int W, int C, int h, int w, int dtype,
                    void* stream) {
  if (!x || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || h <= 0 || w <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * h * w * C);
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(bilinear_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x,
                       (bf16*)out, N, H, W, C, h, w);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(bilinear_kernel<f16>, dim3(grid), dim3(256), 0, st, (const f16*)x,
                       (f16*)out, N, H, W, C, h, w);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(bilinear_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x,
                       (float*)out, N, H, W, C, h, w);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
