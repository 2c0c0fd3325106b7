// The parts of the reference's second backbone that are not 3x3 conv + GroupNorm + SiLU blocks
// (contrastyou/arch/unet2.py): 7x7 stem, 4x4 stride-2 convolution (Downsample :180-181), 4x4 stride-2 transposed
// convolution (Upsample :176-177), 1x1 projections, channel LayerNorm (:183-194), linear attention (:245-271) and
// softmax attention over the bottleneck map (:274-304).
// Everything is f32 on NHWC maps viewed as [pixels][channels] matrices.  The contractions all go through ONE strided,
// batched GEMM on the exact f32 MFMA (v_mfma_f32_32x32x2_f32) -- a convolution is im2col + GEMM (+ col2im for the
// transposed one and for data gradients), the attention products address the q / k / v slices of the qkv map in
// place through row / column / batch strides -- with deterministic split-K (partial tiles + an ordered sum).
// None of this is on the SemiSupervisedEpocher + InfoNCE hot path: sized for correctness and full-chip launches,
// not tuned per shape.
#include "cy_common.h"

namespace {

struct MatL {
  long rs, cs, s1, s2;  // element (i, j) of batch (b1, b2) at base + b1*s1 + b2*s2 + i*rs + j*cs
};

// ---------------------------------------------------------------- strided batched GEMM (f32 MFMA)
// C = alpha * A[M][K] * B[K][N] (+ bias[n]) (+ C).  64x64 block tile, four waves each a 32x32 accumulator, K staged 16
// at a time through LDS.  A_MFAST / B_NFAST: which index is contiguous in memory (the tile loaders walk that one
// fastest).  ksplit > 1: blockIdx.z = batch * ksplit + s computes K range s into ws[z][M][N]; gemm_reduce_kernel sums.
template <bool A_MFAST, bool B_NFAST>
__global__ void __launch_bounds__(256)
    gemm_strided_kernel(const float* __restrict__ A, MatL la, const float* __restrict__ B, MatL lb,
                        float* __restrict__ C, MatL lc, const float* __restrict__ bias, int M, int N, int K,
                        int nb2, float alpha, int accumulate, int ksplit, int kchunk, float* __restrict__ ws) {
  __shared__ float sA[64][17];
  __shared__ float sB[16][65];  // [k][n]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z, batch = z / ksplit, ks = z - batch * ksplit;
  const int b1 = batch / nb2, b2 = batch - b1 * nb2;
  A += b1 * la.s1 + b2 * la.s2;
  B += b1 * lb.s1 + b2 * lb.s2;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int i = lane & 31, kk = lane >> 5;
  const int k_begin = ks * kchunk;
  const int k_end = k_begin + kchunk < K ? k_begin + kchunk : K;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
    __syncthreads();
    for (int e = tid; e < 64 * 16; e += 256) {
      const int rr = A_MFAST ? (e & 63) : (e >> 4), kc = A_MFAST ? (e >> 6) : (e & 15);
      const int gm = m0 + rr, gk = k0 + kc;
      sA[rr][kc] = (gm < M && gk < k_end) ? A[gm * la.rs + gk * la.cs] : 0.f;
    }
    for (int e = tid; e < 64 * 16; e += 256) {
      const int nn = B_NFAST ? (e & 63) : (e >> 4), kc = B_NFAST ? (e >> 6) : (e & 15);
      const int gn = n0 + nn, gk = k0 + kc;
      sB[kc][nn] = (gn < N && gk < k_end) ? B[gk * lb.rs + gn * lb.cs] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      const float av = sA[wm * 32 + i][s + kk];
      const float bv = sB[s + kk][wn * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  float* Cb = C + b1 * lc.s1 + b2 * lc.s2;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kk;
    const int gm = m0 + wm * 32 + row, gn = n0 + wn * 32 + i;
    if (gm < M && gn < N) {
      if (ksplit > 1) {
        ws[((long)z * M + gm) * N + gn] = acc[reg];
      } else {
        float v = alpha * acc[reg];
        if (bias) v += bias[gn];
        float* p = Cb + gm * lc.rs + gn * lc.cs;
        *p = accumulate ? *p + v : v;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    gemm_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, MatL lc, const float* __restrict__ bias,
                       int M, int N, int nb2, int nbatch, float alpha, int accumulate, int ksplit) {
  const long total = (long)nbatch * M * N;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int n = (int)(idx % N);
    const long r = idx / N;
    const int m = (int)(r % M);
    const int batch = (int)(r / M);
    float s = 0.f;
    for (int q = 0; q < ksplit; ++q) s += ws[(((long)batch * ksplit + q) * M + m) * N + n];  // fixed order
    float v = alpha * s;
    if (bias) v += bias[n];
    const int b1 = batch / nb2, b2 = batch - b1 * nb2;
    float* p = C + b1 * lc.s1 + b2 * lc.s2 + m * lc.rs + n * lc.cs;
    *p = accumulate ? *p + v : v;
  }
}

// ---------------------------------------------------------------- im2col / col2im (NHWC, f32)
// cols[(n, ho, wo)][(kh, kw, c)] = x[n, ho*stride - pad + kh, wo*stride - pad + kw, c]   (zero outside)
__global__ void __launch_bounds__(256)
    im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int N, int H, int W, int C, int Ho, int Wo,
                  int KH, int KW, int stride, int pad) {
  const long total = (long)N * Ho * Wo * KH * KW * C;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int kw = (int)(r % KW);
    r /= KW;
    const int kh = (int)(r % KH);
    r /= KH;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
    cols[idx] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((n * H + h) * W + w) * C + c] : 0.f;
  }
}

// the adjoint, as a gather (deterministic): out[n, h, w, c] = bias[c] + sum over the (kh, kw) for which
// ho = (h + pad - kh) / stride, wo = (w + pad - kw) / stride are integers in range of cols[(n, ho, wo)][(kh, kw, c)]
__global__ void __launch_bounds__(256)
    col2im_kernel(const float* __restrict__ cols, const float* __restrict__ bias, float* __restrict__ out, int N,
                  int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad) {
  const long total = (long)N * H * W * C;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long n = r / H;
    float s = bias ? bias[c] : 0.f;
    for (int kh = 0; kh < KH; ++kh) {
      const int th = h + pad - kh;
      if (th < 0 || th % stride) continue;
      const int ho = th / stride;
      if (ho >= Ho) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int tw = w + pad - kw;
        if (tw < 0 || tw % stride) continue;
        const int wo = tw / stride;
        if (wo >= Wo) continue;
        s += cols[(((n * Ho + ho) * Wo + wo) * KH * KW + kh * KW + kw) * C + c];
      }
    }
    out[idx] = s;
  }
}

// ---------------------------------------------------------------- column sums (bias gradients)
// part[s][n] = sum over the rows of slice s of x[m][n]; gemm_reduce-style ordered sum afterwards
__global__ void __launch_bounds__(256)
    colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long M, int N, long rows_per) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const long m0 = (long)blockIdx.y * rows_per;
  const long m1 = m0 + rows_per < M ? m0 + rows_per : M;
  float s = 0.f;
  for (long m = m0; m < m1; ++m) s += x[m * N + n];
  part[(long)blockIdx.y * N + n] = s;
}

__global__ void __launch_bounds__(256)
    slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int S, long n, int accumulate) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    float s = 0.f;
    for (int q = 0; q < S; ++q) s += part[(long)q * n + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

// ---------------------------------------------------------------- channel LayerNorm (unet2.py:183-194)
// y[p][c] = (x[p][c] - mean_p) / sqrt(var_p + eps) * g[c] + b[c], mean / biased variance over the C channels of pixel p
__global__ void __launch_bounds__(256)
    chan_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                       float* __restrict__ y, long M, int C, float eps) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < M; p += (long)gridDim.x * 256L) {
    const float* xp = x + p * C;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += xp[c];
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = xp[c] - mu;
      var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    float* yp = y + p * C;
    for (int c = 0; c < C; ++c) yp[c] = (xp[c] - mu) * rstd * g[c] + b[c];
  }
}

// dx[p][c] = rstd * (t[c] - mean_c(t) - xhat[c] * mean_c(t * xhat)),  t = dy * g;  stats[p] = (mean, rstd) for the
// parameter-gradient pass
__global__ void __launch_bounds__(256)
    chan_ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ dy,
                          float* __restrict__ dx, float* __restrict__ stats, long M, int C, float eps) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < M; p += (long)gridDim.x * 256L) {
    const float* xp = x + p * C;
    const float* dp = dy + p * C;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += xp[c];
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = xp[c] - mu;
      var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float t = dp[c] * g[c];
      s1 += t;
      s2 += t * (xp[c] - mu) * rstd;
    }
    s1 /= (float)C;
    s2 /= (float)C;
    float* op = dx + p * C;
    for (int c = 0; c < C; ++c) {
      const float xh = (xp[c] - mu) * rstd;
      op[c] = rstd * (dp[c] * g[c] - s1 - xh * s2);
    }
    stats[2 * p] = mu;
    stats[2 * p + 1] = rstd;
  }
}

// part[0][s][c] = sum_p dy * xhat, part[1][s][c] = sum_p dy over the pixels of slice s (lane = channel, fixed order)
__global__ void __launch_bounds__(256)
    chan_ln_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                             const float* __restrict__ stats, float* __restrict__ part, long M, int C, long rows_per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long p0 = (long)blockIdx.y * rows_per;
  const long p1 = p0 + rows_per < M ? p0 + rows_per : M;
  float a = 0.f, bsum = 0.f;
  for (long p = p0; p < p1; ++p) {
    const float d = dy[p * C + c];
    a += d * (x[p * C + c] - stats[2 * p]) * stats[2 * p + 1];
    bsum += d;
  }
  part[((long)0 * gridDim.y + blockIdx.y) * C + c] = a;  // [which][slice][C]
  part[((long)1 * gridDim.y + blockIdx.y) * C + c] = bsum;
}

// ---------------------------------------------------------------- softmaxes
// q of the linear attention: softmax over the dh channels of a head, times `scale` (unet2.py:262,265).
// x: rows of ld floats, heads*dh of them from column `off` on.  One thread per (row, head).
__global__ void __launch_bounds__(256)
    head_softmax_fwd_kernel(const float* __restrict__ x, int ld, int off, float* __restrict__ y, long M, int heads,
                            int dh, float scale) {
  const long total = M * heads;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int h = (int)(idx % heads);
    const long p = idx / heads;
    const float* xp = x + p * ld + off + h * dh;
    float m = xp[0];
    for (int d = 1; d < dh; ++d) m = fmaxf(m, xp[d]);
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s += expf(xp[d] - m);
    const float r = scale / s;
    float* yp = y + (p * heads + h) * dh;
    for (int d = 0; d < dh; ++d) yp[d] = expf(xp[d] - m) * r;
  }
}

// y = scale * P, P = softmax: dx = P * scale * (dy - sum_d P * dy), written into rows of ld_dx floats at column off
__global__ void __launch_bounds__(256)
    head_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                            int ld_dx, int off, long M, int heads, int dh, float scale) {
  const long total = M * heads;
  const float inv = 1.0f / scale;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int h = (int)(idx % heads);
    const long p = idx / heads;
    const float* yp = y + (p * heads + h) * dh;
    const float* dp = dy + (p * heads + h) * dh;
    float dot = 0.f;
    for (int d = 0; d < dh; ++d) dot += yp[d] * inv * dp[d];
    float* op = dx + p * ld_dx + off + h * dh;
    for (int d = 0; d < dh; ++d) op[d] = yp[d] * (dp[d] - dot);  // (P * scale = y)
  }
}

// k of the linear attention: softmax over the n POSITIONS of image b, per channel (unet2.py:263).
// stage 1: per (slice, image) running (max, sum of exp) per channel over the slice's positions
__global__ void __launch_bounds__(256)
    col_softmax_partial_kernel(const float* __restrict__ x, int ld, int off, float* __restrict__ part, int n, int Ch,
                               int rows_per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Ch) return;
  const int S = gridDim.y, s = blockIdx.y, b = blockIdx.z;
  const int p0 = s * rows_per;
  const int p1 = p0 + rows_per < n ? p0 + rows_per : n;
  float m = -INFINITY, z = 0.f;
  for (int p = p0; p < p1; ++p) {
    const float v = x[((long)b * n + p) * ld + off + c];
    if (v > m) {
      z = z * expf(m - v) + 1.f;
      m = v;
    } else {
      z += expf(v - m);
    }
  }
  float* o = part + (((long)b * S + s) * Ch + c) * 2;
  o[0] = m;
  o[1] = z;
}

// stage 2: combine the slices in order -> stat[b][c] = (max, sum)
__global__ void __launch_bounds__(256)
    col_softmax_final_kernel(const float* __restrict__ part, float* __restrict__ stat, int B, int S, int Ch) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * Ch) return;
  const int b = idx / Ch, c = idx - b * Ch;
  float m = -INFINITY;
  for (int s = 0; s < S; ++s) m = fmaxf(m, part[(((long)b * S + s) * Ch + c) * 2]);
  float z = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* o = part + (((long)b * S + s) * Ch + c) * 2;
    if (o[1] > 0.f) z += o[1] * expf(o[0] - m);
  }
  stat[2 * idx] = m;
  stat[2 * idx + 1] = z;
}

// stage 3: y[b][p][c] = exp(x - max) / sum
__global__ void __launch_bounds__(256)
    col_softmax_apply_kernel(const float* __restrict__ x, int ld, int off, const float* __restrict__ stat,
                             float* __restrict__ y, long B, int n, int Ch) {
  const long total = B * n * Ch;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int c = (int)(idx % Ch);
    const long r = idx / Ch;
    const long b = r / n;
    const float* st = stat + (b * Ch + c) * 2;
    y[idx] = expf(x[r * ld + off + c] - st[0]) / st[1];
  }
}

// dx[b][p][c] = y * (dy - t[b][c]),  t[b][c] = sum_p y * dy (the caller has it in closed form), into rows of ld_dx
__global__ void __launch_bounds__(256)
    col_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ t,
                           float* __restrict__ dx, int ld_dx, int off, long B, int n, int Ch) {
  const long total = B * n * Ch;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int c = (int)(idx % Ch);
    const long r = idx / Ch;
    const long b = r / n;
    dx[r * ld_dx + off + c] = y[idx] * (dy[idx] - t[b * Ch + c]);
  }
}

// rows of the attention matrix (unet2.py:296-297): softmax over the last axis, in place.  One workgroup per row.
__device__ __forceinline__ float block_reduce(float v, float* sred, bool is_max) {
  const int tid = threadIdx.x;
  sred[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sred[tid] = is_max ? fmaxf(sred[tid], sred[tid + o]) : sred[tid] + sred[tid + o];
    __syncthreads();
  }
  const float r = sred[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) row_softmax_fwd_kernel(float* __restrict__ x, int n) {
  __shared__ float sred[256];
  float* row = x + (long)blockIdx.x * n;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, row[j]);
  m = block_reduce(m, sred, true);
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) s += expf(row[j] - m);
  s = block_reduce(s, sred, false);
  const float r = 1.0f / s;
  for (int j = threadIdx.x; j < n; j += 256) row[j] = expf(row[j] - m) * r;
}

// ds = p * (dp - sum_j p * dp), written over dp
__global__ void __launch_bounds__(256)
    row_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int n) {
  __shared__ float sred[256];
  const float* pr = p + (long)blockIdx.x * n;
  float* dr = dp + (long)blockIdx.x * n;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) s += pr[j] * dr[j];
  s = block_reduce(s, sred, false);
  for (int j = threadIdx.x; j < n; j += 256) dr[j] = pr[j] * (dr[j] - s);
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

inline MatL to_l(const cy_mat_layout* l) { return MatL{l->rs, l->cs, l->s1, l->s2}; }

// ---- the time embedding of UNet2 (arch/unet2.py:51-58,161-173,230-231): a [B, dim] table and two tiny MLPs ----------
// out[b][i] = sin(t_b e_i), out[b][half + i] = cos(t_b e_i), e_i = exp(-i log(10000) / (half - 1))   (SinusoidalPosEmb)
__global__ void __launch_bounds__(256) sinusoidal_emb_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * half) return;
  const int b = e / half, i = e - b * half;
  const float f = expf((float)i * -(logf(10000.f) / (float)(half - 1)));
  const float a = t[b] * f;
  out[(size_t)b * dim + i] = sinf(a);
  out[(size_t)b * dim + half + i] = cosf(a);
}

// kind 0: SiLU (x sigmoid(x)), kind 1: GELU (exact: x Phi(x), nn.GELU() default)
__device__ __forceinline__ float act_f(float x, int kind) {
  if (kind == 0) return x / (1.f + expf(-x));
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float act_df(float x, int kind) {
  if (kind == 0) {
    const float s = 1.f / (1.f + expf(-x));
    return s * (1.f + x * (1.f - s));
  }
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}
__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) y[i] = act_f(x[i], kind);
}
__global__ void __launch_bounds__(256)
    act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n, int kind) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) dx[i] = dy[i] * act_df(x[i], kind);
}

}  // namespace

extern "C" {

int cy_sinusoidal_emb(const float* time, float* out, int B, int dim, void* stream) {
  if (!time || !out || B <= 0) return CY_ERR_ARG;
  if (dim < 4 || dim % 2) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(sinusoidal_emb_kernel, dim3(cy_cdiv(B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, time, out, B, dim);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_act_fwd(const float* x, float* y, long n, int kind, void* stream) {
  if (!x || !y || n <= 0) return CY_ERR_ARG;
  if (kind != 0 && kind != 1) return CY_ERR_SHAPE;
  long b = (n + 255) / 256;
  hipLaunchKernelGGL(act_fwd_kernel, dim3((int)(b > 4096 ? 4096 : b)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream) {
  if (!x || !dy || !dx || n <= 0) return CY_ERR_ARG;
  if (kind != 0 && kind != 1) return CY_ERR_SHAPE;
  long b = (n + 255) / 256;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((int)(b > 4096 ? 4096 : b)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, kind);
  CY_CHECK_LAUNCH();
  return CY_OK;
}


size_t cy_gemm_strided_ws_bytes(int M, int N, int nbatch, int ksplit) {
  return ksplit > 1 ? (size_t)nbatch * ksplit * M * N * sizeof(float) : 0;
}

int cy_gemm_strided(const float* A, const cy_mat_layout* la, const float* B, const cy_mat_layout* lb, float* C,
                    const cy_mat_layout* lc, const float* bias, int M, int N, int K, int nb1, int nb2, float alpha,
                    int accumulate, int ksplit, float* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || !la || !lb || !lc) return CY_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || nb1 <= 0 || nb2 <= 0 || ksplit <= 0) return CY_ERR_SHAPE;
  const long nbatch = (long)nb1 * nb2;
  if (nbatch * ksplit > 65535) return CY_ERR_SHAPE;
  if (ksplit > 1 && (!ws || ws_bytes < cy_gemm_strided_ws_bytes(M, N, (int)nbatch, ksplit))) return CY_ERR_WORKSPACE;
  int kchunk = (K + ksplit - 1) / ksplit;
  kchunk = (kchunk + 15) / 16 * 16;
  hipStream_t st = (hipStream_t)stream;
  if (cy_cdiv(N, 64) > 65535) return CY_ERR_SHAPE;
  const dim3 grid(cy_cdiv(M, 64), cy_cdiv(N, 64), (unsigned)(nbatch * ksplit));
  const bool am = la->rs == 1 && la->cs != 1, bn = lb->cs == 1;
#define CY_GEMM(AM, BN)                                                                                         \
  hipLaunchKernelGGL((gemm_strided_kernel<AM, BN>), grid, dim3(256), 0, st, A, to_l(la), B, to_l(lb), C, to_l(lc), \
                     bias, M, N, K, nb2, alpha, accumulate, ksplit, kchunk, ws)
  if (am && bn) CY_GEMM(true, true);
  else if (am) CY_GEMM(true, false);
  else if (bn) CY_GEMM(false, true);
  else CY_GEMM(false, false);
#undef CY_GEMM
  CY_CHECK_LAUNCH();
  if (ksplit > 1) {
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(grid_for(nbatch * M * N)), dim3(256), 0, st, ws, C, to_l(lc), bias, M,
                       N, nb2, (int)nbatch, alpha, accumulate, ksplit);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_im2col(const float* x, float* cols, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
              void* stream) {
  if (!x || !cols) return CY_ERR_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return CY_ERR_SHAPE;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((long)N * Ho * Wo * KH * KW * C)), dim3(256), 0,
                     (hipStream_t)stream, x, cols, N, H, W, C, Ho, Wo, KH, KW, stride, pad);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_col2im(const float* cols, const float* bias, float* out, int N, int H, int W, int C, int KH, int KW, int stride,
              int pad, void* stream) {
  if (!cols || !out) return CY_ERR_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return CY_ERR_SHAPE;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, cols,
                     bias, out, N, H, W, C, Ho, Wo, KH, KW, stride, pad);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int slices_for(long M) {
  long s = (M + 255) / 256;
  return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

size_t cy_colsum_ws_bytes(long M, int N) { return (size_t)slices_for(M) * N * sizeof(float); }

int cy_colsum(const float* x, float* out, long M, int N, int accumulate, float* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || !ws) return CY_ERR_ARG;
  if (M <= 0 || N <= 0) return CY_ERR_SHAPE;
  if (ws_bytes < cy_colsum_ws_bytes(M, N)) return CY_ERR_WORKSPACE;
  const int S = slices_for(M);
  const long rows_per = (M + S - 1) / S;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cy_cdiv(N, 256), S), dim3(256), 0, st, x, ws, M, N, rows_per);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(N)), dim3(256), 0, st, ws, out, S, (long)N, accumulate);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_chan_layernorm_fwd(const float* x, const float* g, const float* b, float* y, long M, int C, float eps,
                          void* stream) {
  if (!x || !g || !b || !y) return CY_ERR_ARG;
  if (M <= 0 || C <= 0) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(chan_ln_fwd_kernel, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, x, g, b, y, M, C, eps);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_chan_layernorm_bwd_ws_bytes(long M, int C) {
  return ((size_t)2 * M + (size_t)slices_for(M) * 2 * C) * sizeof(float);
}

int cy_chan_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, float* db, long M,
                          int C, float eps, float* ws, size_t ws_bytes, void* stream) {
  if (!x || !g || !dy || !dx || !ws) return CY_ERR_ARG;
  if (M <= 0 || C <= 0) return CY_ERR_SHAPE;
  if (ws_bytes < cy_chan_layernorm_bwd_ws_bytes(M, C)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* stats = ws;
  float* part = ws + 2 * M;
  hipLaunchKernelGGL(chan_ln_bwd_dx_kernel, dim3(grid_for(M)), dim3(256), 0, st, x, g, dy, dx, stats, M, C, eps);
  CY_CHECK_LAUNCH();
  if (dg && db) {
    const int S = slices_for(M);
    const long rows_per = (M + S - 1) / S;
    hipLaunchKernelGGL(chan_ln_bwd_param_kernel, dim3(cy_cdiv(C, 256), S), dim3(256), 0, st, x, dy, stats, part, M, C,
                       rows_per);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(C)), dim3(256), 0, st, part, dg, S, (long)C, 0);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(C)), dim3(256), 0, st, part + (size_t)S * C, db, S, (long)C, 0);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_head_softmax_fwd(const float* x, int ld, int off, float* y, long M, int heads, int dh, float scale,
                        void* stream) {
  if (!x || !y) return CY_ERR_ARG;
  if (M <= 0 || heads <= 0 || dh <= 0 || off < 0 || ld < off + heads * dh) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(head_softmax_fwd_kernel, dim3(grid_for(M * heads)), dim3(256), 0, (hipStream_t)stream, x, ld, off,
                     y, M, heads, dh, scale);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_head_softmax_bwd(const float* y, const float* dy, float* dx, int ld_dx, int off, long M, int heads, int dh,
                        float scale, void* stream) {
  if (!y || !dy || !dx) return CY_ERR_ARG;
  if (M <= 0 || heads <= 0 || dh <= 0 || off < 0 || ld_dx < off + heads * dh || scale == 0.f) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(head_softmax_bwd_kernel, dim3(grid_for(M * heads)), dim3(256), 0, (hipStream_t)stream, y, dy, dx,
                     ld_dx, off, M, heads, dh, scale);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int col_slices(int n) {
  int s = (n + 127) / 128;
  return s < 1 ? 1 : (s > 256 ? 256 : s);
}

size_t cy_col_softmax_ws_bytes(int B, int n, int Ch) {
  return ((size_t)B * col_slices(n) * Ch * 2 + (size_t)B * Ch * 2) * sizeof(float);
}

int cy_col_softmax_fwd(const float* x, int ld, int off, float* y, int B, int n, int Ch, float* ws, size_t ws_bytes,
                       void* stream) {
  if (!x || !y || !ws) return CY_ERR_ARG;
  if (B <= 0 || n <= 0 || Ch <= 0 || off < 0 || ld < off + Ch || B > 65535) return CY_ERR_SHAPE;
  if (ws_bytes < cy_col_softmax_ws_bytes(B, n, Ch)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int S = col_slices(n);
  const int rows_per = (n + S - 1) / S;
  float* part = ws;
  float* stat = ws + (size_t)B * S * Ch * 2;
  hipLaunchKernelGGL(col_softmax_partial_kernel, dim3(cy_cdiv(Ch, 256), S, B), dim3(256), 0, st, x, ld, off, part, n,
                     Ch, rows_per);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(col_softmax_final_kernel, dim3(cy_cdiv(B * Ch, 256)), dim3(256), 0, st, part, stat, B, S, Ch);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(col_softmax_apply_kernel, dim3(grid_for((long)B * n * Ch)), dim3(256), 0, st, x, ld, off, stat, y,
                     (long)B, n, Ch);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_col_softmax_bwd(const float* y, const float* dy, const float* t, float* dx, int ld_dx, int off, int B, int n,
                       int Ch, void* stream) {
  if (!y || !dy || !t || !dx) return CY_ERR_ARG;
  if (B <= 0 || n <= 0 || Ch <= 0 || off < 0 || ld_dx < off + Ch) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(col_softmax_bwd_kernel, dim3(grid_for((long)B * n * Ch)), dim3(256), 0, (hipStream_t)stream, y,
                     dy, t, dx, ld_dx, off, (long)B, n, Ch);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_row_softmax_fwd(float* x, long rows, int n, void* stream) {
  if (!x) return CY_ERR_ARG;
  if (rows <= 0 || n <= 0 || rows > 0x7fffffffL) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(row_softmax_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, n);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_row_softmax_bwd(const float* p, float* dp, long rows, int n, void* stream) {
  if (!p || !dp) return CY_ERR_ARG;
  if (rows <= 0 || n <= 0 || rows > 0x7fffffffL) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, n);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
