// BatchNorm2d (training statistics) + ReLU forward/backward around the MFMA
// convolutions, and the backward halves of MaxPool2d(2) / nearest Upsample(x2)
// whose forward halves live in the conv load path.
// Reference call sites: contrastyou/arch/unet.py:22-23,25-26,40-41 (BN+ReLU),
// :67-70 (MaxPool2d), :38 (Upsample).  All kernels are HBM-streaming: 16-byte
// NHWC chunks per lane, per-channel coefficients from L1/L2, fixed-order
// (deterministic) reductions through per-block partials.
#include "cy_bn_acc.h"

#include <cstdlib>

namespace {

// ------------------------------------------------------------------ finalize
__global__ void __launch_bounds__(1024)
    bn_finalize_kernel(const float* __restrict__ partials, int P, int C, double count,
                       const float* __restrict__ gamma, const float* __restrict__ beta,
                       float* running_mean, float* running_var, float momentum, float eps,
                       int use_batch_stats, int update_running, float* scale, float* shift,
                       float* mean_out, float* invstd_out) {
  // block = 4 channels x 256 partial slices (many small blocks: the reduction is latency-bound);
  // fixed summation order (slice-major) => deterministic
  __shared__ double sred[2][256][4];
  const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  // (the per-channel parameters are requested BEFORE the partial sums are reduced: one memory round trip
  //  instead of two in a kernel that is nothing but latency)
  float rm_old = 0.f, rv_old = 1.f, gm = 1.f, bt = 0.f;
  if (sl == 0 && c < C) {
    if (running_mean) rm_old = running_mean[c], rv_old = running_var[c];
    if (gamma) gm = gamma[c];
    if (beta) bt = beta[c];
  }
  double s1 = 0.0, s2 = 0.0;
  if (c < C && use_batch_stats) {
#pragma unroll 8  // (the loads of eight rounds in flight; same order of additions)
    for (int p = sl; p < P; p += 256) {
      s1 += (double)partials[((size_t)p * 2 + 0) * C + c];
      s2 += (double)partials[((size_t)p * 2 + 1) * C + c];
    }
  }
  sred[0][sl][cl] = s1;
  sred[1][sl][cl] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed-shape tree over the slices (deterministic)
    if (sl < o) {
      sred[0][sl][cl] += sred[0][sl + o][cl];
      sred[1][sl][cl] += sred[1][sl + o][cl];
    }
    __syncthreads();
  }
  if (sl == 0 && c < C) {
    double mean, var;
    if (use_batch_stats) {
      const double t1 = sred[0][0][cl], t2 = sred[1][0][cl];
      mean = t1 / count;
      var = t2 / count - mean * mean;
      if (var < 0.0) var = 0.0;
      if (update_running) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)rm_old + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)rv_old + momentum * unb);
      }
    } else {
      mean = rm_old;
      var = rv_old;
    }
    const double istd = 1.0 / sqrt(var + (double)eps);
    const double g = (double)gm;
    const double b = (double)bt;
    scale[c] = (float)(g * istd);
    shift[c] = (float)(b - mean * g * istd);
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)istd;
  }
}

// ------------------------------------------------------------------ apply
// FOLD (cy_bn_acc.h): the coefficients are derived from the producing conv's accumulator by every workgroup (one channel
// per thread, through LDS); workgroup 0 leaves them in memory for the backward pass.  The host keeps the grid small
// enough that the accumulator reads (R * C * 32 bytes per workgroup, from L2) stay a small share of the tensor traffic.
// threads of a workgroup of the elementwise kernels: the folding forms run 1024 (a quarter of the workgroups, i.e. of the
// accumulator gathers -- R * C * 32 bytes each, through one L2 -- and a gather that is one round trip: tools/bench_ew_cold.py)
template <bool FOLD> constexpr int ew_threads() { return FOLD ? 1024 : 256; }

template <bool FOLD>
__device__ __forceinline__ void bn_fold_table(const BnFold& f, float* s_coef) {
  if constexpr (FOLD) {
    constexpr int NT = ew_threads<FOLD>();
    unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(s_coef + 2 * f.C);  // [4 C + 1] (host: C % 8 == 0)
    const int c = threadIdx.x;  // (host: C <= 1024 = NT)  gamma / beta requested ahead of the gather
    const float pg = (c < f.C && f.gamma) ? f.gamma[c] : 1.f, pb = (c < f.C && f.beta) ? f.beta[c] : 0.f;
    bn_acc_gather<true>(f.acc, f.R, f.C, s_sum, threadIdx.x, NT);
    if (c < f.C) {
      float a, b;
      bn_fold_finish_gb(f, c, blockIdx.x == 0, bn_sum_read(s_sum, f.C, 0, c), bn_sum_read(s_sum, f.C, 1, c), (double)pg,
                        (double)pb, a, b);
      s_coef[c] = a;
      s_coef[f.C + c] = b;
    }
    __syncthreads();
  }
}

template <typename TI, typename TO, bool FOLD>
__global__ void __launch_bounds__(ew_threads<FOLD>())
    bn_relu_apply_kernel(const TI* __restrict__ y, const float* __restrict__ scale,
                         const float* __restrict__ shift, TO* __restrict__ out, long npix, int C, const BnFold fold) {
  extern __shared__ __attribute__((aligned(16))) float s_coef[];  // FOLD: [2][C] floats, then [4 C + 1] 64-bit sums
  // unit of work: 8 channels of one pixel (one 16-byte bf16 chunk / two f32 chunks).  The item after the current one
  // is always in flight; the FIRST one is requested before the coefficients are derived (FOLD: the accumulator's round
  // trip and the data's overlap -- on the small maps the kernel is nothing but that chain of latencies).
  constexpr long NT = ew_threads<FOLD>();
  const int G = C / 8;
  const long total = npix * G;
  const long stride = (long)gridDim.x * NT;
  long i = blockIdx.x * NT + threadIdx.x;
  u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
  auto request = [&](long idx) {
    const int g = (int)(idx % G);
    const TI* yp = y + (idx / G) * C + g * 8;
    r0 = ld16(yp);
    if constexpr (sizeof(TI) == 4) r1 = ld16(yp + 4);
  };
  if (i < total) request(i);
  bn_fold_table<FOLD>(fold, s_coef);
  while (i < total) {
    const int g = (int)(i % G);
    const long p = i / G;
    float f[8];
    if constexpr (sizeof(TI) == 2) {
      Chunk<TI>::unpack(r0, f);
    } else {
      Chunk<float>::unpack(r0, f);
      Chunk<float>::unpack(r1, f + 4);
    }
    const long nxt = i + stride;
    if (nxt < total) request(nxt);
    f32x4 s0, s1, h0, h1;
    if constexpr (FOLD) {
      s0 = *reinterpret_cast<const f32x4*>(s_coef + g * 8), s1 = *reinterpret_cast<const f32x4*>(s_coef + g * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(s_coef + C + g * 8), h1 = *reinterpret_cast<const f32x4*>(s_coef + C + g * 8 + 4);
    } else {
      s0 = *reinterpret_cast<const f32x4*>(scale + g * 8), s1 = *reinterpret_cast<const f32x4*>(scale + g * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(shift + g * 8), h1 = *reinterpret_cast<const f32x4*>(shift + g * 8 + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = fmaxf(fmaf(s0[j], f[j], h0[j]), 0.f);
      f[4 + j] = fmaxf(fmaf(s1[j], f[4 + j], h1[j]), 0.f);
    }
    TO* op = out + p * C + g * 8;
    if constexpr (sizeof(TO) == 2) {
      st16(op, Chunk<TO>::pack(f));
    } else {
      st16(op, Chunk<float>::pack(f));
      st16(op + 4, Chunk<float>::pack(f + 4));
    }
    i = nxt;
  }
}

// out = relu(scale*y + shift) AND its 2x2 max (nn.MaxPool2d(2) of the block output, reference unet.py:108-121) in one
// pass: unit of work = 8 channels of one 2x2 quad (H, W are the POOLED dims).  The pooled copy costs a quarter of
// the writes and lets the next block's first conv (and its weight gradient) read a plain tensor -- i.e. run on the
// DMA-fed kernels, which cannot take a maximum on load.
template <typename TI, typename TO, bool FOLD>
__global__ void __launch_bounds__(ew_threads<FOLD>())
    bn_relu_apply_pool_kernel(const TI* __restrict__ y, const float* __restrict__ scale,
                              const float* __restrict__ shift, TO* __restrict__ out, TO* __restrict__ pooled,
                              int N, int H, int W, int C, const BnFold fold) {
  extern __shared__ __attribute__((aligned(16))) float s_coef[];  // FOLD: [2][C] floats, then [4 C + 1] 64-bit sums
  bn_fold_table<FOLD>(fold, s_coef);
  const int G = C / 8;
  const long total = (long)N * H * W * G;
  const int W2 = 2 * W;
  constexpr long NT = ew_threads<FOLD>();
  for (long i = blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int g = (int)(i % G);
    long p = i / G;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long n = p / H;
    f32x4 s0, s1, h0, h1;
    if constexpr (FOLD) {
      s0 = *reinterpret_cast<const f32x4*>(s_coef + g * 8), s1 = *reinterpret_cast<const f32x4*>(s_coef + g * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(s_coef + C + g * 8), h1 = *reinterpret_cast<const f32x4*>(s_coef + C + g * 8 + 4);
    } else {
      s0 = *reinterpret_cast<const f32x4*>(scale + g * 8), s1 = *reinterpret_cast<const f32x4*>(scale + g * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(shift + g * 8), h1 = *reinterpret_cast<const f32x4*>(shift + g * 8 + 4);
    }
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = 0.f;  // relu outputs are >= 0
    float f[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long px = (n * 2 * H + 2 * h + (q >> 1)) * W2 + 2 * w + (q & 1);
      const TI* yp = y + px * C + g * 8;
      if constexpr (sizeof(TI) == 2) {
        Chunk<TI>::unpack(ld16(yp), f[q]);
      } else {
        Chunk<float>::unpack(ld16(yp), f[q]);
        Chunk<float>::unpack(ld16(yp + 4), f[q] + 4);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[q][j] = fmaxf(fmaf(s0[j], f[q][j], h0[j]), 0.f);
        f[q][4 + j] = fmaxf(fmaf(s1[j], f[q][4 + j], h1[j]), 0.f);
      }
      const long px = (n * 2 * H + 2 * h + (q >> 1)) * W2 + 2 * w + (q & 1);
      TO* op = out + px * C + g * 8;
      if constexpr (sizeof(TO) == 2) {
        // the pooled value is the max of the STORED (rounded) values, like MaxPool2d of the stored tensor
        const u32x4 pk = Chunk<TO>::pack(f[q]);
        st16(op, pk);
        float r[8];
        Chunk<TO>::unpack(pk, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], r[j]);
      } else {
        st16(op, Chunk<float>::pack(f[q]));
        st16(op + 4, Chunk<float>::pack(f[q] + 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[q][j]);
      }
    }
    TO* pp = pooled + ((n * H + h) * W + w) * C + g * 8;
    if constexpr (sizeof(TO) == 2) {
      st16(pp, Chunk<TO>::pack(m));
    } else {
      st16(pp, Chunk<float>::pack(m));
      st16(pp + 4, Chunk<float>::pack(m + 4));
    }
  }
}

// ------------------------------------------------------------------ backward
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

// partial sums of dz and dz*xhat.  thread = (pixel lane, group of 8 channels)
// DEEP: four pixels per round with all eight loads requested first -- for the small maps, whose few workgroups
// are latency-bound (14 x 14 x 16 images: 21 -> 14 us); on the large maps the 64 extra registers cost occupancy and
// bandwidth (23 vs 20 us at 112 x 112 x 16), so the host picks per launch
template <typename T, bool DEEP>
__global__ void __launch_bounds__(256)
    bn_relu_bwd_reduce_kernel(const T* __restrict__ da, int ld_da, const T* __restrict__ y,
                              const float* __restrict__ scale, const float* __restrict__ shift,
                              const float* __restrict__ mean, const float* __restrict__ invstd,
                              float* __restrict__ partials, long npix, int C, unsigned long long* sacc, int sR) {
  extern __shared__ float sred[];  // [2][rows][C]
  const int G = C / 8;
  const int tid = threadIdx.x;
  const int nb = gridDim.x;
  const long per = (npix + nb - 1) / nb;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  // channel groups may exceed the block: loop over group "pages"
  const int gpp = G < 256 ? G : 256;  // groups per page
  const int rows = 256 / gpp;
  const int g_in_page = tid % gpp;
  const int prow = tid / gpp;
  const bool active = tid < rows * gpp;
  for (int page = 0; page * gpp < G; ++page) {
    const int g = page * gpp + g_in_page;
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
    if (active && g < G) {
      float sc[8], sh[8], mu[8], is[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = scale[g * 8 + j];
        sh[j] = shift[g * 8 + j];
        mu[j] = mean[g * 8 + j];
        is[j] = invstd[g * 8 + j];
      }
      // four pixels per round, all eight loads requested before the first use: a thread walks up to a few dozen
      // pixels and the dependent form paid one memory latency per pixel (14 us on a 3 MB map).  Same order of
      // additions per thread as the one-pixel loop.
      long p = p0 + prow;
      if constexpr (DEEP)
      for (; p + 3L * rows < p1; p += 4L * rows) {
        float d[4][8], v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          load8<T>(da + (p + (long)u * rows) * ld_da + g * 8, d[u]);
          load8<T>(y + (p + (long)u * rows) * C + g * 8, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float z = fmaf(sc[j], v[u][j], sh[j]);
            const float dz = z > 0.f ? d[u][j] : 0.f;
            a1[j] += dz;
            a2[j] += dz * ((v[u][j] - mu[j]) * is[j]);
          }
      }
      for (; p < p1; p += rows) {
        float d[8], v[8];
        load8<T>(da + p * ld_da + g * 8, d);
        load8<T>(y + p * C + g * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = fmaf(sc[j], v[j], sh[j]);
          const float dz = z > 0.f ? d[j] : 0.f;
          a1[j] += dz;
          a2[j] += dz * ((v[j] - mu[j]) * is[j]);
        }
      }
    }
    __syncthreads();
    if (active && g < G) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sred[(0 * rows + prow) * (gpp * 8) + g_in_page * 8 + j] = a1[j];
        sred[(1 * rows + prow) * (gpp * 8) + g_in_page * 8 + j] = a2[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * gpp * 8; i += 256) {
      const int which = i / (gpp * 8);
      const int cl = i % (gpp * 8);
      const int c = page * gpp * 8 + cl;
      if (c < C) {
        float s = 0.f;
        for (int q = 0; q < rows; ++q) s += sred[(which * rows + q) * (gpp * 8) + cl];
        if (sacc) bn_acc_add(sacc, sR, C, (int)blockIdx.x & (sR - 1), which, c, s);
        else partials[((size_t)blockIdx.x * 2 + which) * C + c] = s;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    bn_bwd_finalize_kernel(const float* __restrict__ partials, int P, int C,
                           const float* __restrict__ scale, const float* __restrict__ mean,
                           const float* __restrict__ invstd, double count, int batch_stats,
                           float* dgamma, float* dbeta, int accumulate, float* __restrict__ coef) {
  __shared__ double sred[2][64][4];
  const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float sc_ = 0.f, mu_ = 0.f, is_ = 0.f, dg_old = 0.f, db_old = 0.f;  // requested before the reduction (see bn_finalize_kernel)
  if (sl == 0 && c < C) {
    if (batch_stats) sc_ = scale[c], mu_ = mean[c], is_ = invstd[c];
    if (accumulate) {
      if (dbeta) db_old = dbeta[c];
      if (dgamma) dg_old = dgamma[c];
    }
  }
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
#pragma unroll 8  // (the loads of eight rounds in flight; same order of additions)
    for (int p = sl; p < P; p += 64) {
      s1 += (double)partials[((size_t)p * 2 + 0) * C + c];
      s2 += (double)partials[((size_t)p * 2 + 1) * C + c];
    }
  }
  sred[0][sl][cl] = s1;
  sred[1][sl][cl] = s2;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (sl < o) {
      sred[0][sl][cl] += sred[0][sl + o][cl];
      sred[1][sl][cl] += sred[1][sl + o][cl];
    }
    __syncthreads();
  }
  if (sl == 0 && c < C) {
    const double t1 = sred[0][0][cl], t2 = sred[1][0][cl];
    if (dbeta) dbeta[c] = accumulate ? db_old + (float)t1 : (float)t1;
    if (dgamma) dgamma[c] = accumulate ? dg_old + (float)t2 : (float)t2;
    // dy = scale*dz + k1*y + k0   (k1 = k0 = 0 when BN used running statistics)
    double k1 = 0.0, k0 = 0.0;
    if (batch_stats) {
      const double sc = sc_, mu = mu_, is = is_;
      k1 = -sc * is * t2 / count;
      k0 = -sc * t1 / count - k1 * mu;
    }
    coef[c] = (float)k1;
    coef[C + c] = (float)k0;
  }
}

// FOLD (cy_bn_acc.h): (k1, k0) are derived from the backward sums' accumulator by every workgroup; workgroup 0 adds the
// parameter gradients (what bn_bwd_finalize_kernel did in a launch of its own)
template <typename T, bool FOLD>
__global__ void __launch_bounds__(ew_threads<FOLD>())
    bn_relu_bwd_apply_kernel(const T* __restrict__ da, int ld_da, const T* __restrict__ y,
                             const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ coef, T* __restrict__ dy, long npix, int C, const BnBwdFold fold) {
  extern __shared__ __attribute__((aligned(16))) float s_coef[];  // FOLD: [4][C] = scale, shift, k1, k0, then [4 C + 1] 64-bit sums
  const int G = C / 8;
  const bool pow2 = (G & (G - 1)) == 0;
  const int gshift = 31 - __clz(G);
  constexpr long NT = ew_threads<FOLD>();
  const long total = npix * G;
  const long stride = (long)gridDim.x * NT;
  long i = blockIdx.x * NT + threadIdx.x;
  // the next item's (da, y) are always in flight; the first one's before the coefficients are derived (see
  // bn_relu_apply_kernel)
  u32x4 rd0 = {0u, 0u, 0u, 0u}, rd1 = {0u, 0u, 0u, 0u}, rv0 = {0u, 0u, 0u, 0u}, rv1 = {0u, 0u, 0u, 0u};
  auto request = [&](long idx) {
    const int g = pow2 ? (int)(idx & (G - 1)) : (int)(idx % G);
    const long p = pow2 ? (idx >> gshift) : (idx / G);
    const T* dp = da + p * ld_da + g * 8;
    const T* vp = y + p * C + g * 8;
    // (streaming reads: this pass is the last reader of dA and, in the backward pass, of y -- 5 % on cold data)
    rd0 = ld16_nt(dp), rv0 = ld16_nt(vp);
    if constexpr (sizeof(T) == 4) rd1 = ld16_nt(dp + 4), rv1 = ld16_nt(vp + 4);
  };
  if (i < total) request(i);
  if constexpr (FOLD) {
    unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(s_coef + 4 * C);
    // (host: C <= 1024 = NT) the forward coefficients and moments of this thread's channel and, for the leader, the gradient
    // words it adds to: all requested ahead of the gather, used after it -- one round trip to memory, not three
    const bool leader = blockIdx.x == 0;
    const int c = threadIdx.x;
    const bool on = c < C;
    const float psc = on ? scale[c] : 0.f, psh = on ? shift[c] : 0.f;
    const float pm = (on && fold.batch_stats) ? fold.coef[2 * C + c] : 0.f, pi = (on && fold.batch_stats) ? fold.coef[3 * C + c] : 0.f;
    const float pdg = (on && leader && fold.accumulate && fold.dgamma) ? fold.dgamma[c] : 0.f;
    const float pdb = (on && leader && fold.accumulate && fold.dbeta) ? fold.dbeta[c] : 0.f;
    bn_acc_gather<true>(fold.acc, fold.R, C, s_sum, threadIdx.x, (int)NT);
    if (on) {
      float k1v, k0v;
      bn_bwd_fold_finish(fold, s_sum, c, leader, psc, pm, pi, pdg, pdb, k1v, k0v);
      s_coef[c] = psc;
      s_coef[C + c] = psh;
      s_coef[2 * C + c] = k1v;
      s_coef[3 * C + c] = k0v;
    }
    __syncthreads();
  }
  while (i < total) {
    const int g = pow2 ? (int)(i & (G - 1)) : (int)(i % G);
    const long p = pow2 ? (i >> gshift) : (i / G);
    float d[8], v[8], o[8];
    if constexpr (sizeof(T) == 2) {
      Chunk<T>::unpack(rd0, d);
      Chunk<T>::unpack(rv0, v);
    } else {
      Chunk<float>::unpack(rd0, d), Chunk<float>::unpack(rd1, d + 4);
      Chunk<float>::unpack(rv0, v), Chunk<float>::unpack(rv1, v + 4);
    }
    const long nxt = i + stride;
    if (nxt < total) request(nxt);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a, b, c1, c0;
      if constexpr (FOLD) {
        a = *reinterpret_cast<const f32x4*>(s_coef + g * 8 + 4 * h), b = *reinterpret_cast<const f32x4*>(s_coef + C + g * 8 + 4 * h);
        c1 = *reinterpret_cast<const f32x4*>(s_coef + 2 * C + g * 8 + 4 * h), c0 = *reinterpret_cast<const f32x4*>(s_coef + 3 * C + g * 8 + 4 * h);
      } else {
        a = *reinterpret_cast<const f32x4*>(scale + g * 8 + 4 * h), b = *reinterpret_cast<const f32x4*>(shift + g * 8 + 4 * h);
        c1 = *reinterpret_cast<const f32x4*>(coef + g * 8 + 4 * h), c0 = *reinterpret_cast<const f32x4*>(coef + C + g * 8 + 4 * h);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float yv = v[4 * h + j];
        const float z = fmaf(a[j], yv, b[j]);
        const float dz = z > 0.f ? d[4 * h + j] : 0.f;
        o[4 * h + j] = fmaf(a[j], dz, fmaf(c1[j], yv, c0[j]));
      }
    }
    store8<T>(dy + p * C + g * 8, o);
    i = nxt;
  }
}

// ------------------------------------------------------------------ pool / upsample backward
// STATS: the gradient this kernel writes is the dA of the block's last BatchNorm: its backward sums (sum dz, sum dz*xhat
// with dz = dA where relu(bn(y)) is active) are taken from the values in registers -- one extra read of y instead of the
// separate reduce pass over (dA, y).  One partial row per workgroup, lanes of a channel group combined in a fixed order;
// the sums use the ROUNDED gradient, i.e. what bn_relu_bwd_reduce_kernel would read back.  Needs 256 % (C/8) == 0.
template <typename T, bool STATS>
__global__ void __launch_bounds__(256)
    maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dpool,
                        const T* __restrict__ add, int ld_add, T* __restrict__ dx, int N, int H,
                        int W, int C, const T* __restrict__ y, const float* __restrict__ scale,
                        const float* __restrict__ shift, const float* __restrict__ mean,
                        const float* __restrict__ invstd, float* __restrict__ partials, unsigned long long* sacc, int sR) {
  // (H,W) are the POOLED dims; x/dx/add/y are [N,2H,2W,C]
  const int G = C / 8;
  const long total = (long)N * H * W * G;
  const int W2 = 2 * W;
  float a1[8], a2[8], sc[8], sh[8], mu[8], is[8];
  if constexpr (STATS) {
    const int g = threadIdx.x % G;  // (the grid stride is a multiple of G: a thread keeps its channel group)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a1[j] = a2[j] = 0.f;
      sc[j] = scale[g * 8 + j], sh[j] = shift[g * 8 + j], mu[j] = mean[g * 8 + j], is[j] = invstd[g * 8 + j];
    }
  }
  constexpr long NT = 256;
  for (long i = blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int g = (int)(i % G);
    long p = i / G;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    const long base = ((long)(n * 2 * H + 2 * h) * W2 + 2 * w);
    const long q[4] = {base, base + 1, base + W2, base + W2 + 1};
    float xv[4][8], g8[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) load8<T>(x + q[k] * C + g * 8, xv[k]);
    load8<T>(dpool + ((long)(n * H + h) * W + w) * C + g * 8, g8);
    float o[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int best = 0;
      float bv = xv[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        if (xv[k][j] > bv) {
          bv = xv[k][j];
          best = k;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == best) ? g8[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (add) {
        float a8[8];
        load8<T>(add + q[k] * ld_add + g * 8, a8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[k][j] += a8[j];
      }
      store8<T>(dx + q[k] * C + g * 8, o[k]);
      if constexpr (STATS) {
        float yv[8], d[8];
        load8<T>(y + q[k] * C + g * 8, yv);
        if constexpr (sizeof(T) == 2) {
          Chunk<T>::unpack(Chunk<T>::pack(o[k]), d);  // the stored (rounded) gradient
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = o[k][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = fmaf(sc[j], yv[j], sh[j]);
          const float dz = z > 0.f ? d[j] : 0.f;
          a1[j] += dz;
          a2[j] += dz * ((yv[j] - mu[j]) * is[j]);
        }
      }
    }
  }
  if constexpr (STATS) {
    __shared__ float sred[256 * 16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sred[threadIdx.x * 16 + j] = a1[j];
      sred[threadIdx.x * 16 + 8 + j] = a2[j];
    }
    __syncthreads();
    const int per_g = 256 / G;
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
      const int which = e / C, c = e - which * C;
      const int g = c >> 3, jj = c & 7;
      float s = 0.f;
      for (int k = 0; k < per_g; ++k) s += sred[(g + k * G) * 16 + which * 8 + jj];
      if (sacc) bn_acc_add(sacc, sR, C, (int)blockIdx.x & (sR - 1), which, c, s);
      else partials[((size_t)blockIdx.x * 2 + which) * C + c] = s;
    }
  }
}

// STATS (as maxpool2_bwd_kernel): the gradient this kernel writes is the dA of relu(bn(y)) -- the last BatchNorm of the block
// whose output the Upsample consumed -- and its backward sums are added into an accumulator from the values in registers
// (the rounded gradient, as a separate reduce pass would read it back).  Needs 256 % (C/8) == 0.
template <typename T, bool STATS>
__global__ void __launch_bounds__(256)
    upsample2_bwd_kernel(const T* __restrict__ dup, int ld_dup, T* __restrict__ dx, int N, int H,
                         int W, int C, const T* __restrict__ y, const float* __restrict__ coef, unsigned long long* sacc,
                         int sR) {
  const int G = C / 8;
  const long total = (long)N * H * W * G;
  const int W2 = 2 * W;
  float a1[8], a2[8], sc[8], sh[8], mu[8], is[8];
  if constexpr (STATS) {
    const int g = threadIdx.x % G;  // (the grid stride is a multiple of G: a thread keeps its channel group)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a1[j] = a2[j] = 0.f;
      sc[j] = coef[g * 8 + j], sh[j] = coef[C + g * 8 + j], mu[j] = coef[2 * C + g * 8 + j], is[j] = coef[3 * C + g * 8 + j];
    }
  }
  constexpr long NT = 256;
  for (long i = blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int g = (int)(i % G);
    long p = i / G;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    const long base = ((long)(n * 2 * H + 2 * h) * W2 + 2 * w);
    float a[8], b[8], c[8], d[8];
    load8<T>(dup + base * ld_dup + g * 8, a);
    load8<T>(dup + (base + 1) * ld_dup + g * 8, b);
    load8<T>(dup + (base + W2) * ld_dup + g * 8, c);
    load8<T>(dup + (base + W2 + 1) * ld_dup + g * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (a[j] + b[j]) + (c[j] + d[j]);
    store8<T>(dx + ((long)(n * H + h) * W + w) * C + g * 8, a);
    if constexpr (STATS) {
      float yv[8], dv[8];
      load8<T>(y + ((long)(n * H + h) * W + w) * C + g * 8, yv);
      if constexpr (sizeof(T) == 2) {
        Chunk<T>::unpack(Chunk<T>::pack(a), dv);  // the stored (rounded) gradient
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = a[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = fmaf(sc[j], yv[j], sh[j]);
        const float dz = z > 0.f ? dv[j] : 0.f;
        a1[j] += dz;
        a2[j] += dz * ((yv[j] - mu[j]) * is[j]);
      }
    }
  }
  if constexpr (STATS) {
    __shared__ float sred[256 * 16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sred[threadIdx.x * 16 + j] = a1[j];
      sred[threadIdx.x * 16 + 8 + j] = a2[j];
    }
    __syncthreads();
    const int per_g = 256 / G;
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
      const int which = e / C, c = e - which * C;
      const int g = c >> 3, jj = c & 7;
      float t = 0.f;
      for (int k = 0; k < per_g; ++k) t += sred[(g + k * G) * 16 + which * 8 + jj];
      bn_acc_add(sacc, sR, C, (int)blockIdx.x & (sR - 1), which, c, t);
    }
  }
}

// accumulator -> coefficients as a launch of its own (for consumers that cannot fold in place)
__global__ void __launch_bounds__(256) bn_fold_kernel(const BnFold f) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_sum_k[];  // [4 C + 1]
  bn_acc_gather(f.acc, f.R, f.C, s_sum_k, threadIdx.x, 256);
  for (int c = threadIdx.x; c < f.C; c += 256) {
    float a, b;
    bn_fold_channel_lds(f, s_sum_k, c, true, a, b);
  }
}

// running statistics of several layers in one launch; the table travels as a kernel argument
struct BnRunArgs {
  cy_bn_run_item it[32];
};
__global__ void __launch_bounds__(256) bn_running_update_kernel(const BnRunArgs a) {
  const cy_bn_run_item& it = a.it[blockIdx.x];
  for (int c = threadIdx.x; c < it.C; c += 256) {
    const double m = (double)it.momentum;
    it.running_mean[c] = (float)((1.0 - m) * (double)it.running_mean[c] + m * (double)it.coef[2 * it.C + c]);
    it.running_var[c] = (float)((1.0 - m) * (double)it.running_var[c] + m * (double)it.coef[4 * it.C + c]);
  }
}

// grid of the folding elementwise kernels: every workgroup reads the accumulator, so few, long-running workgroups of
// 1024 threads (one per CU)
inline int fold_grid(long items) {
  long b = (items + 1023) / 1024;
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

inline int stream_grid(long total_threads) {
  long b = (total_threads + 255) / 256;
  if (b > 2048 * 4) b = 2048 * 4;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int cy_bn_finalize(const float* partials, int num_partials, int C, double count,
                   const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int use_batch_stats,
                   int update_running, float* scale, float* shift, float* mean, float* invstd,
                   void* stream) {
  if (C <= 0 || !scale || !shift || !mean || !invstd) return CY_ERR_ARG;
  if (use_batch_stats && (!partials || num_partials <= 0 || count <= 0)) return CY_ERR_ARG;
  if ((!use_batch_stats || update_running) && (!running_mean || !running_var)) return CY_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cy_cdiv(C, 4)), dim3(1024), 0, (hipStream_t)stream,
                     partials, num_partials, C, count, gamma, beta, running_mean, running_var,
                     momentum, eps, use_batch_stats, update_running, scale, shift, mean, invstd);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int bn_relu_apply_impl(const void* y, const float* scale, const float* shift, const cy_bn_fold* f, void* out,
                              long npix, int C, int y_dtype, int out_dtype, void* stream) {
  if (!y || !out || npix <= 0) return CY_ERR_ARG;
  if (!f && (!scale || !shift)) return CY_ERR_ARG;
  if (f && (!f->acc || !f->coef || f->C != C || f->R < 1 || (f->R & (f->R - 1)) || f->count <= 0)) return CY_ERR_ARG;
  if (C % 8 || (f && C > 1024)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const BnFold bf = f ? bn_fold_from_abi(f) : BnFold{};
  const int grid = f ? fold_grid(npix * (C / 8)) : stream_grid(npix * (C / 8));
  const size_t smem = f ? (size_t)2 * C * sizeof(float) + ((size_t)4 * C + 2) * 8 : 0;
#define CY_APPLY(TI, TO)                                                                                        \
  do {                                                                                                          \
    if (f) hipLaunchKernelGGL((bn_relu_apply_kernel<TI, TO, true>), dim3(grid), dim3(1024), smem, st, (const TI*)y, \
                              scale, shift, (TO*)out, npix, C, bf);                                             \
    else hipLaunchKernelGGL((bn_relu_apply_kernel<TI, TO, false>), dim3(grid), dim3(256), 0, st, (const TI*)y,  \
                            scale, shift, (TO*)out, npix, C, bf);                                               \
  } while (0)
  if (y_dtype == CY_BF16 && out_dtype == CY_BF16) CY_APPLY(bf16, bf16);
  else if (y_dtype == CY_F32 && out_dtype == CY_F32) CY_APPLY(float, float);
  else if (y_dtype == CY_BF16 && out_dtype == CY_F32) CY_APPLY(bf16, float);
  else if (y_dtype == CY_F16 && out_dtype == CY_F16) CY_APPLY(f16, f16);
  else if (y_dtype == CY_F16 && out_dtype == CY_F32) CY_APPLY(f16, float);
  else return CY_ERR_DTYPE;
#undef CY_APPLY
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_relu_apply(const void* y, const float* scale, const float* shift, void* out, long npix,
                     int C, int y_dtype, int out_dtype, void* stream) {
  return bn_relu_apply_impl(y, scale, shift, nullptr, out, npix, C, y_dtype, out_dtype, stream);
}

int cy_bn_relu_apply_fold(const void* y, const cy_bn_fold* f, void* out, long npix, int y_dtype, int out_dtype,
                          void* stream) {
  if (!f) return CY_ERR_ARG;
  return bn_relu_apply_impl(y, nullptr, nullptr, f, out, npix, f->C, y_dtype, out_dtype, stream);
}

static int bn_relu_apply_pool_impl(const void* y, const float* scale, const float* shift, const cy_bn_fold* f, void* out,
                                   void* pooled, int N, int H, int W, int C, int y_dtype, int out_dtype, void* stream) {
  if (!y || !out || !pooled || N <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  if (!f && (!scale || !shift)) return CY_ERR_ARG;
  if (f && (!f->acc || !f->coef || f->C != C || f->R < 1 || (f->R & (f->R - 1)) || f->count <= 0)) return CY_ERR_ARG;
  if (C % 8 || (f && C > 1024)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const BnFold bf = f ? bn_fold_from_abi(f) : BnFold{};
  const long items = (long)N * H * W * (C / 8);
  const int grid = f ? fold_grid(items * 4) : stream_grid(items);
  const size_t smem = f ? (size_t)2 * C * sizeof(float) + ((size_t)4 * C + 2) * 8 : 0;
#define CY_APPLY_POOL(TI, TO)                                                                                      \
  do {                                                                                                             \
    if (f) hipLaunchKernelGGL((bn_relu_apply_pool_kernel<TI, TO, true>), dim3(grid), dim3(1024), smem, st,          \
                              (const TI*)y, scale, shift, (TO*)out, (TO*)pooled, N, H, W, C, bf);                  \
    else hipLaunchKernelGGL((bn_relu_apply_pool_kernel<TI, TO, false>), dim3(grid), dim3(256), 0, st, (const TI*)y, \
                            scale, shift, (TO*)out, (TO*)pooled, N, H, W, C, bf);                                  \
  } while (0)
  if (y_dtype == CY_BF16 && out_dtype == CY_BF16) CY_APPLY_POOL(bf16, bf16);
  else if (y_dtype == CY_F32 && out_dtype == CY_F32) CY_APPLY_POOL(float, float);
  else if (y_dtype == CY_F16 && out_dtype == CY_F16) CY_APPLY_POOL(f16, f16);
  else return CY_ERR_DTYPE;
#undef CY_APPLY_POOL
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_relu_apply_pool(const void* y, const float* scale, const float* shift, void* out, void* pooled,
                          int N, int H, int W, int C, int y_dtype, int out_dtype, void* stream) {
  return bn_relu_apply_pool_impl(y, scale, shift, nullptr, out, pooled, N, H, W, C, y_dtype, out_dtype, stream);
}

int cy_bn_relu_apply_pool_fold(const void* y, const cy_bn_fold* f, void* out, void* pooled, int N, int H, int W,
                               int y_dtype, int out_dtype, void* stream) {
  if (!f) return CY_ERR_ARG;
  return bn_relu_apply_pool_impl(y, nullptr, nullptr, f, out, pooled, N, H, W, f->C, y_dtype, out_dtype, stream);
}

int cy_bn_fold_coef(const cy_bn_fold* f, void* stream) {
  if (!f || !f->acc || !f->coef || f->C <= 0 || f->R < 1 || (f->R & (f->R - 1)) || f->count <= 0) return CY_ERR_ARG;
  if (f->C > 2048) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(1), dim3(256), ((size_t)4 * f->C + 2) * 8, (hipStream_t)stream, bn_fold_from_abi(f));
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_running_update(const cy_bn_run_item* h_items, int n, void* stream) {
  if (!h_items || n <= 0) return CY_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!h_items[i].coef || !h_items[i].running_mean || !h_items[i].running_var || h_items[i].C <= 0) return CY_ERR_ARG;
  for (int i0 = 0; i0 < n; i0 += 32) {
    BnRunArgs a;
    const int m = n - i0 < 32 ? n - i0 : 32;
    for (int i = 0; i < m; ++i) a.it[i] = h_items[i0 + i];
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, a);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_bn_bwd_num_partials(long npix, int C) {
  (void)C;
  // 128 pixels per workgroup; small maps (fewer than 512 such workgroups) get 32 pixels per workgroup and the
  // deep-prefetch form of the kernel
  long b = (npix + 127) / 128;
  if (b < 512) b = (npix + 31) / 32 < 512 ? (npix + 31) / 32 : 511;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

static int bn_relu_bwd_reduce_impl(const void* da, int ld_da, const void* y, const float* scale,
                                   const float* shift, const float* mean, const float* invstd,
                                   float* partials, unsigned long long* sacc, int sR, long npix, int C, int dtype, void* stream);

int cy_bn_relu_bwd_reduce(const void* da, int ld_da, const void* y, const float* scale,
                          const float* shift, const float* mean, const float* invstd,
                          float* partials, long npix, int C, int dtype, void* stream) {
  if (!partials) return CY_ERR_ARG;
  return bn_relu_bwd_reduce_impl(da, ld_da, y, scale, shift, mean, invstd, partials, nullptr, 0, npix, C, dtype, stream);
}

int cy_bn_relu_bwd_workgroups(long npix, int C) { return cy_bn_bwd_num_partials(npix, C); }

int cy_bn_relu_bwd_reduce_acc(const void* da, int ld_da, const void* y, const float* coef, const cy_bn_acc* acc,
                              long npix, int C, int dtype, void* stream) {
  if (!coef || !acc || !acc->acc || acc->C != C || acc->R < 1 || (acc->R & (acc->R - 1))) return CY_ERR_ARG;
  return bn_relu_bwd_reduce_impl(da, ld_da, y, coef, coef + C, coef + 2 * C, coef + 3 * C, nullptr,
                                 (unsigned long long*)acc->acc, acc->R, npix, C, dtype, stream);
}

static int bn_relu_bwd_reduce_impl(const void* da, int ld_da, const void* y, const float* scale,
                                   const float* shift, const float* mean, const float* invstd,
                                   float* partials, unsigned long long* sacc, int sR, long npix, int C, int dtype, void* stream) {
  if (!da || !y || !scale || !shift || !mean || !invstd || (!partials && !sacc)) return CY_ERR_ARG;
  if (C % 8 || ld_da % 8 || ld_da < C) return CY_ERR_SHAPE;
  const int G = C / 8;
  const int gpp = G < 256 ? G : 256;
  const int rows = 256 / gpp;
  const size_t smem = (size_t)2 * rows * gpp * 8 * sizeof(float);
  const int grid = cy_bn_bwd_num_partials(npix, C);
  hipStream_t st = (hipStream_t)stream;
  const bool deep = grid < 512;
#define CY_BN_RED(TT, DD)                                                                                          \
  hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<TT, DD>), dim3(grid), dim3(256), smem, st, (const TT*)da, ld_da,   \
                     (const TT*)y, scale, shift, mean, invstd, partials, npix, C, sacc, sR)
  if (dtype == CY_BF16) {
    if (deep) CY_BN_RED(bf16, true); else CY_BN_RED(bf16, false);
  } else if (dtype == CY_F16) {
    if (deep) CY_BN_RED(f16, true); else CY_BN_RED(f16, false);
  } else if (dtype == CY_F32) {
    if (deep) CY_BN_RED(float, true); else CY_BN_RED(float, false);
  } else {
    return CY_ERR_DTYPE;
  }
#undef CY_BN_RED
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_bwd_finalize(const float* partials, int num_partials, int C, const float* scale,
                       const float* mean, const float* invstd, double count, int batch_stats,
                       float* dgamma, float* dbeta, int accumulate, float* coef, void* stream) {
  if (!partials || !coef || num_partials <= 0 || C <= 0) return CY_ERR_ARG;
  if (batch_stats && (!scale || !mean || !invstd || count <= 0)) return CY_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cy_cdiv(C, 4)), dim3(256), 0,
                     (hipStream_t)stream, partials, num_partials, C, scale, mean, invstd, count,
                     batch_stats, dgamma, dbeta, accumulate, coef);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_relu_bwd_apply(const void* da, int ld_da, const void* y, const float* scale,
                         const float* shift, const float* coef, void* dy, long npix, int C,
                         int dtype, void* stream) {
  if (!da || !y || !scale || !shift || !coef || !dy) return CY_ERR_ARG;
  if (C % 8 || ld_da % 8 || ld_da < C) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(npix * (C / 8));
  const BnBwdFold nf = {};
#define CY_BWD_APPLY(TT)                                                                                          \
  hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<TT, false>), dim3(grid), dim3(256), 0, st, (const TT*)da, ld_da,    \
                     (const TT*)y, scale, shift, coef, (TT*)dy, npix, C, nf)
  if (dtype == CY_BF16) CY_BWD_APPLY(bf16);
  else if (dtype == CY_F16) CY_BWD_APPLY(f16);
  else if (dtype == CY_F32) CY_BWD_APPLY(float);
  else return CY_ERR_DTYPE;
#undef CY_BWD_APPLY
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_bn_relu_bwd_apply_fold(const void* da, int ld_da, const void* y, const float* coef, const cy_bn_acc* acc,
                              double count, int batch_stats, float* dgamma, float* dbeta, int accumulate, void* dy,
                              long npix, int C, int dtype, void* stream) {
  if (!da || !y || !coef || !dy || !acc || !acc->acc || acc->C != C || acc->R < 1 || (acc->R & (acc->R - 1)) || count <= 0)
    return CY_ERR_ARG;
  if (C % 8 || ld_da % 8 || ld_da < C || C > 1024) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  BnBwdFold f;
  f.acc = (const unsigned long long*)acc->acc, f.R = acc->R, f.C = C, f.coef = coef;
  f.inv_count = 1.0 / count, f.batch_stats = batch_stats, f.accumulate = accumulate, f.dgamma = dgamma, f.dbeta = dbeta;
  const int grid = fold_grid(npix * (C / 8));
  const size_t smem = (size_t)4 * C * sizeof(float) + ((size_t)4 * C + 2) * 8;
#define CY_BWD_APPLY(TT)                                                                                          \
  hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<TT, true>), dim3(grid), dim3(1024), smem, st, (const TT*)da, ld_da,  \
                     (const TT*)y, coef, coef + C, (const float*)nullptr, (TT*)dy, npix, C, f)
  if (dtype == CY_BF16) CY_BWD_APPLY(bf16);
  else if (dtype == CY_F16) CY_BWD_APPLY(f16);
  else if (dtype == CY_F32) CY_BWD_APPLY(float);
  else return CY_ERR_DTYPE;
#undef CY_BWD_APPLY
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_maxpool2_bwd(const void* x, const void* dpool, const void* add, int ld_add, void* dx, int N,
                    int H, int W, int C, int dtype, void* stream) {
  if (!x || !dpool || !dx || N <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  if (C % 8 || (add && (ld_add % 8 || ld_add < C))) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((long)N * H * W * (C / 8));
#define CY_POOL_BWD(TT)                                                                                          \
  hipLaunchKernelGGL((maxpool2_bwd_kernel<TT, false>), dim3(grid), dim3(256), 0, st, (const TT*)x,              \
                     (const TT*)dpool, (const TT*)add, ld_add, (TT*)dx, N, H, W, C, (const TT*)nullptr, nullptr, \
                     nullptr, nullptr, nullptr, nullptr, nullptr, 0)
  if (dtype == CY_BF16) CY_POOL_BWD(bf16);
  else if (dtype == CY_F16) CY_POOL_BWD(f16);
  else if (dtype == CY_F32) CY_POOL_BWD(float);
  else return CY_ERR_DTYPE;
#undef CY_POOL_BWD
  CY_CHECK_LAUNCH();
  return CY_OK;
}

/* partial rows cy_maxpool2_bwd_bn writes (a function of the geometry only), or CY_ERR_SHAPE when the fused form
 * does not apply (the channel groups must divide a workgroup) */
int cy_maxpool2_bwd_bn_num_partials(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C % 8 || C / 8 > 256 || 256 % (C / 8)) return CY_ERR_SHAPE;
  long b = ((long)N * H * W * (C / 8) + 255) / 256;
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

static int maxpool2_bwd_bn_impl(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                float* partials, unsigned long long* sacc, int sR, int N, int H, int W, int C, int dtype,
                                void* stream);

int cy_maxpool2_bwd_bn(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* partials, int N, int H, int W, int C, int dtype, void* stream) {
  if (!partials) return CY_ERR_ARG;
  return maxpool2_bwd_bn_impl(x, dpool, add, ld_add, dx, y, scale, shift, mean, invstd, partials, nullptr, 0, N, H, W, C,
                              dtype, stream);
}

int cy_maxpool2_bwd_bn_acc(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                           const float* coef, const cy_bn_acc* acc, int N, int H, int W, int C, int dtype,
                           void* stream) {
  if (!coef || !acc || !acc->acc || acc->C != C || acc->R < 1 || (acc->R & (acc->R - 1))) return CY_ERR_ARG;
  return maxpool2_bwd_bn_impl(x, dpool, add, ld_add, dx, y, coef, coef + C, coef + 2 * C, coef + 3 * C, nullptr,
                              (unsigned long long*)acc->acc, acc->R, N, H, W, C, dtype, stream);
}

static int maxpool2_bwd_bn_impl(const void* x, const void* dpool, const void* add, int ld_add, void* dx, const void* y,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                float* partials, unsigned long long* sacc, int sR, int N, int H, int W, int C, int dtype,
                                void* stream) {
  if (!x || !dpool || !dx || !y || !scale || !shift || !mean || !invstd || (!partials && !sacc)) return CY_ERR_ARG;
  if (add && (ld_add % 8 || ld_add < C)) return CY_ERR_SHAPE;
  const int grid = cy_maxpool2_bwd_bn_num_partials(N, H, W, C);
  if (grid < 0) return grid;
  hipStream_t st = (hipStream_t)stream;
#define CY_POOL_BWD(TT)                                                                                      \
  hipLaunchKernelGGL((maxpool2_bwd_kernel<TT, true>), dim3(grid), dim3(256), 0, st, (const TT*)x,           \
                     (const TT*)dpool, (const TT*)add, ld_add, (TT*)dx, N, H, W, C, (const TT*)y, scale, shift, \
                     mean, invstd, partials, sacc, sR)
  if (dtype == CY_BF16) CY_POOL_BWD(bf16);
  else if (dtype == CY_F16) CY_POOL_BWD(f16);
  else if (dtype == CY_F32) CY_POOL_BWD(float);
  else return CY_ERR_DTYPE;
#undef CY_POOL_BWD
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int upsample2_bwd_impl(const void* dup, int ld_dup, void* dx, const void* y, const float* coef,
                              unsigned long long* sacc, int sR, int N, int H, int W, int C, int dtype, void* stream) {
  if (!dup || !dx || N <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  if (C % 8 || ld_dup % 8 || ld_dup < C) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int grid = stream_grid((long)N * H * W * (C / 8));
  if (sacc) {
    if (C / 8 > 256 || 256 % (C / 8)) return CY_ERR_SHAPE;
    grid = cy_upsample2_bwd_bn_workgroups(N, H, W, C);
  }
#define CY_UP_BWD(TT)                                                                                             \
  do {                                                                                                            \
    if (sacc) hipLaunchKernelGGL((upsample2_bwd_kernel<TT, true>), dim3(grid), dim3(256), 0, st, (const TT*)dup,   \
                                 ld_dup, (TT*)dx, N, H, W, C, (const TT*)y, coef, sacc, sR);                      \
    else hipLaunchKernelGGL((upsample2_bwd_kernel<TT, false>), dim3(grid), dim3(256), 0, st, (const TT*)dup,       \
                            ld_dup, (TT*)dx, N, H, W, C, (const TT*)nullptr, (const float*)nullptr,               \
                            (unsigned long long*)nullptr, 0);                                                     \
  } while (0)
  if (dtype == CY_BF16) CY_UP_BWD(bf16);
  else if (dtype == CY_F16) CY_UP_BWD(f16);
  else if (dtype == CY_F32) CY_UP_BWD(float);
  else return CY_ERR_DTYPE;
#undef CY_UP_BWD
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_upsample2_bwd(const void* dup, int ld_dup, void* dx, int N, int H, int W, int C, int dtype,
                     void* stream) {
  return upsample2_bwd_impl(dup, ld_dup, dx, nullptr, nullptr, nullptr, 0, N, H, W, C, dtype, stream);
}

int cy_upsample2_bwd_bn_workgroups(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C % 8 || C / 8 > 256 || 256 % (C / 8)) return CY_ERR_SHAPE;
  long b = ((long)N * H * W * (C / 8) + 255) / 256;
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

int cy_upsample2_bwd_bn_acc(const void* dup, int ld_dup, void* dx, const void* y, const float* coef,
                            const cy_bn_acc* acc, int N, int H, int W, int C, int dtype, void* stream) {
  if (!y || !coef || !acc || !acc->acc || acc->C != C || acc->R < 1 || (acc->R & (acc->R - 1))) return CY_ERR_ARG;
  return upsample2_bwd_impl(dup, ld_dup, dx, y, coef, (unsigned long long*)acc->acc, acc->R, N, H, W, C, dtype, stream);
}

}  // extern "C"
