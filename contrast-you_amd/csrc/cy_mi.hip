// Discrete mutual-information losses over cluster-probability maps.
//   DenseClusterHead / ClusterHead softmax(T)     contrastyou/projectors/heads.py:44-78,125-173,
//                                                 contrastyou/projectors/nn.py:35-44
//   compute_joint (vectors)                       contrastyou/losses/discreteMI.py:201-222
//   compute_joint_2D (displaced, padding > 0)     contrastyou/losses/discreteMI.py:225-243
//   compute_joint_2D_with_padding_zeros           contrastyou/losses/discreteMI.py:246-261
//   IIDLoss / IIDSegmentationLoss                 contrastyou/losses/discreteMI.py:90-170
//
// The k x k joint of two [pixels][k] probability maps is a skinny contraction over N*H*W pixels
// (k = 20): HBM-bound, 2*k*4 bytes read per pixel, k*k FMAs.  It is NOT reshaped into an MFMA
// GEMM: each thread keeps a 4x4 register tile of the joint and walks a slice of the pixels of an
// LDS-staged tile; block partials are reduced in a fixed order (deterministic).
// Layouts: probability maps are NHWC f32 [N][H][W][k]; joints are [T*T][k][k], T = 2*padding+1.
#include "cy_common.h"

namespace {

constexpr int J_P = 64;      // pixels per staged tile
constexpr int J_KMAX = 64;   // max clusters

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---------------------------------------------------------------- grouped softmax
// in  [M][S*k] logits  ->  out [S][M][k] probabilities of softmax(logits * invT) per group.
// A block owns GS_PB consecutive rows: the [GS_PB][S*k] logit tile is one contiguous span of
// memory and each [GS_PB][k] output tile is another, so all global traffic is coalesced; the
// (row, group) -> thread transposition happens in LDS (row pitch S*k+1: conflict-free).
constexpr int GS_PB = 64;

__global__ void __launch_bounds__(256)
    group_softmax_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, long M, int S,
                             int k, float invT) {
  extern __shared__ float tile[];  // [GS_PB][S*k + 1]
  const int SK = S * k, pitch = SK + 1;
  const long m0 = (long)blockIdx.x * GS_PB;
  const int rows = (int)min((long)GS_PB, M - m0);
  const float* src = in + m0 * SK;
  for (int e = threadIdx.x; e < rows * SK; e += 256) tile[(e / SK) * pitch + e % SK] = src[e] * invT;
  __syncthreads();
  for (int e = threadIdx.x; e < rows * S; e += 256) {
    float* v = tile + (e % rows) * pitch + (e / rows) * k;
    float mx = -INFINITY;
    for (int i = 0; i < k; ++i) mx = fmaxf(mx, v[i]);
    float sum = 0.f;
    for (int i = 0; i < k; ++i) {
      v[i] = __expf(v[i] - mx);
      sum += v[i];
    }
    const float r = 1.f / sum;
    for (int i = 0; i < k; ++i) v[i] *= r;
  }
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    float* dst = out + ((long)s * M + m0) * k;
    for (int e = threadIdx.x; e < rows * k; e += 256) dst[e] = tile[(e / k) * pitch + s * k + e % k];
  }
}

// dlogits[m][s*k+i] = invT * p_i * (dp_i - sum_j p_j dp_j)
__global__ void __launch_bounds__(256)
    group_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                             float* __restrict__ dlogits, long M, int S, int k, float invT) {
  extern __shared__ float tile[];  // p then dp: 2 x [GS_PB][S*k + 1]
  const int SK = S * k, pitch = SK + 1;
  float* tp = tile;
  float* td = tile + GS_PB * pitch;
  const long m0 = (long)blockIdx.x * GS_PB;
  const int rows = (int)min((long)GS_PB, M - m0);
  for (int s = 0; s < S; ++s) {
    const float* sp = p + ((long)s * M + m0) * k;
    const float* sd = dp + ((long)s * M + m0) * k;
    for (int e = threadIdx.x; e < rows * k; e += 256) {
      tp[(e / k) * pitch + s * k + e % k] = sp[e];
      td[(e / k) * pitch + s * k + e % k] = sd[e];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < rows * S; e += 256) {
    const int off = (e % rows) * pitch + (e / rows) * k;
    float dot = 0.f;
    for (int i = 0; i < k; ++i) dot = fmaf(tp[off + i], td[off + i], dot);
    for (int i = 0; i < k; ++i) td[off + i] = invT * tp[off + i] * (td[off + i] - dot);
  }
  __syncthreads();
  float* dst = dlogits + m0 * SK;
  for (int e = threadIdx.x; e < rows * SK; e += 256) dst[e] = td[(e / SK) * pitch + e % SK];
}

// ---------------------------------------------------------------- joint forward
// part[blockIdx.y = displacement][blockIdx.x][k*k]
__global__ void __launch_bounds__(256)
    joint_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                     float* __restrict__ part, int N, int H, int W, int k, int pad) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int kp = (k + 3) & ~3, k4 = kp >> 2;
  const int tps = k4 * k4;           // threads per pixel slice
  const int nsl = 256 / tps;         // slices (>= 1 because k <= 64)
  float* xs1 = sm;                   // [J_P][kp]
  float* xs2 = sm + J_P * kp;        // [J_P][kp]
  float* red = sm + 2 * J_P * kp;    // [nsl][kp*kp]
  const int tid = threadIdx.x;
  const int T = 2 * pad + 1;
  const int du = (int)blockIdx.y / T - pad, dv = (int)blockIdx.y % T - pad;
  const int sl = tid / tps, tt = tid - sl * tps;
  const int ti = tt / k4, tj = tt - ti * k4;
  const bool active = sl < nsl;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  const long npix = (long)N * H * W;
  const long per = ((npix + gridDim.x - 1) / gridDim.x + J_P - 1) / J_P * J_P;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  for (long pt = p0; pt < p1; pt += J_P) {
    __syncthreads();
    if (pad == 0 && kp == k) {  // the tile is one contiguous span of both maps: 16-byte copies
      const long lim = (p1 - pt) * k;  // floats of this tile that exist
      for (int e4 = tid; e4 < J_P * k / 4; e4 += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (4L * e4 < lim) {
          a = *reinterpret_cast<const f32x4*>(x1 + pt * k + 4 * e4);
          b = *reinterpret_cast<const f32x4*>(x2 + pt * k + 4 * e4);
        }
        *reinterpret_cast<f32x4*>(xs1 + 4 * e4) = a;
        *reinterpret_cast<f32x4*>(xs2 + 4 * e4) = b;
      }
    } else
    for (int e = tid; e < J_P * kp; e += 256) {
      const int p = e / kp, c = e - p * kp;
      const long pix = pt + p;
      float a = 0.f, b = 0.f;
      if (pix < p1 && c < k) {
        b = x2[pix * k + c];
        if (pad == 0) {
          a = x1[pix * k + c];
        } else {
          const int w = (int)(pix % W);
          const long t = pix / W;
          const int h = (int)(t % H);
          const int hh = h + du, ww = w + dv;
          if (hh >= 0 && hh < H && ww >= 0 && ww < W) a = x1[(pix + (long)du * W + dv) * k + c];
        }
      }
      xs1[e] = a;
      xs2[e] = b;
    }
    __syncthreads();
    if (active) {
      for (int p = sl; p < J_P; p += nsl) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xs1 + p * kp + 4 * ti);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xs2 + p * kp + 4 * tj);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[sl * kp * kp + (4 * ti + i) * kp + 4 * tj + j] = acc[i][j];
  }
  __syncthreads();
  float* dst = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (k * k);
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    float s = 0.f;
    for (int q = 0; q < nsl; ++q) s += red[q * kp * kp + i * kp + j];
    dst[e] = s;
  }
}

// Displaced joints (padding > 0) in ONE pass over the maps: the kernel above is launched per displacement and reads
// both maps (2p+1)^2 times (9 displacements: 266 GB/s of algorithmic traffic).  Here a block stages R image rows of
// x2 and the R + 2p rows of x1 around them, with p zero columns on both sides (so a displaced read is an address
// shift, no bounds test), and every thread keeps one 4x4 tile of the joint per displacement: up to JM_D = 9 tiles,
// 144 accumulators; (2p+1)^2 > 9 runs in groups of nine (blockIdx.y).  Same partial layout and fixed-order
// reduction as above.
constexpr int JM_D = 9;

__global__ void __launch_bounds__(256, 2)
    joint_fwd_multi_kernel(const float* __restrict__ x1, const float* __restrict__ x2, float* __restrict__ part,
                           int N, int H, int W, int k, int pad, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int kp = (k + 3) & ~3, k4 = kp >> 2;
  const int tps = k4 * k4, nsl = 256 / tps;
  const int Wp = W + 2 * pad, Rp = R + 2 * pad;
  float* xs1 = sm;                       // [Rp][Wp][kp]
  float* xs2 = xs1 + (size_t)Rp * Wp * kp;  // [R][W][kp]
  float* red = sm;                          // [nsl][kp*kp]: over the tiles once they are done (72 KB at W = 224,
                                            // k = 20: two workgroups per CU, one stages while the other computes)
  const int tid = threadIdx.x;
  const int T = 2 * pad + 1;
  const int d0 = (int)blockIdx.y * JM_D;
  const int nd = T * T - d0 < JM_D ? T * T - d0 : JM_D;
  const int sl = tid / tps, tt = tid - sl * tps;
  const int ti = tt / k4, tj = tt - ti * k4;
  const bool active = sl < nsl;
  float acc[JM_D][4][4];
#pragma unroll
  for (int d = 0; d < JM_D; ++d)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[d][a][b] = 0.f;
  int doff[JM_D];  // LDS offset (floats) of displacement d relative to the undisplaced pixel
#pragma unroll
  for (int d = 0; d < JM_D; ++d) {
    const int dd = d0 + (d < nd ? d : 0);
    doff[d] = ((dd / T - pad) * Wp + (dd % T - pad)) * kp;
  }
  const int tiles_h = (H + R - 1) / R;
  const int ntile = N * tiles_h;
  if (kp == k)  // the zero columns left and right of every staged x1 row
    for (int e = tid; e < Rp * 2 * pad * kp; e += 256) {
      const int r = e / (2 * pad * kp), q = e - r * (2 * pad * kp);
      const int col = q / kp < pad ? q / kp : W + q / kp;  // 0..pad-1 | W+pad..W+2pad-1
      xs1[(r * Wp + col) * kp + q % kp] = 0.f;
    }
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int n = t / tiles_h, h0 = (t - n * tiles_h) * R;
    const int rows = H - h0 < R ? H - h0 : R;
    __syncthreads();
    if (kp == k) {
      // an image row is one contiguous span of W*k floats in memory and in the staged tile: 16-byte copies, no
      // index arithmetic (the pad columns were zeroed once, rows outside the image are zeroed here)
      const int row4 = W * k / 4;
      for (int r = 0; r < Rp; ++r) {
        const int hh = h0 - pad + r;
        const bool ok = r < rows + 2 * pad && hh >= 0 && hh < H;
        const float* src = x1 + ((long)n * H + (ok ? hh : 0)) * W * k;
        float* dst = xs1 + (r * Wp + pad) * kp;
        for (int e4 = tid; e4 < row4; e4 += 256) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) v = *reinterpret_cast<const f32x4*>(src + 4 * e4);
          *reinterpret_cast<f32x4*>(dst + 4 * e4) = v;
        }
      }
      const float* src2 = x2 + ((long)n * H + h0) * W * k;
      for (int e4 = tid; e4 < R * row4; e4 += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e4 < rows * row4) v = *reinterpret_cast<const f32x4*>(src2 + 4 * e4);
        *reinterpret_cast<f32x4*>(xs2 + 4 * e4) = v;
      }
    } else {
      for (int e = tid; e < Rp * Wp * kp; e += 256) {
        const int c = e % kp;
        const int q = e / kp;
        const int cw = q % Wp, r = q / Wp;
        const int hh = h0 - pad + r, ww = cw - pad;
        float v = 0.f;
        if (c < k && r < rows + 2 * pad && hh >= 0 && hh < H && ww >= 0 && ww < W)
          v = x1[(((long)n * H + hh) * W + ww) * k + c];
        xs1[e] = v;
      }
      for (int e = tid; e < R * W * kp; e += 256) {
        const int c = e % kp;
        const int q = e / kp;
        const int w = q % W, r = q / W;
        xs2[e] = (c < k && r < rows) ? x2[(((long)n * H + h0 + r) * W + w) * k + c] : 0.f;
      }
    }
    __syncthreads();
    if (active) {
      int r = 0, w = sl;
      while (w >= W) w -= W, ++r;
      for (; r < rows;) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(xs2 + (r * W + w) * kp + 4 * tj);
        const float* a0 = xs1 + ((r + pad) * Wp + (w + pad)) * kp + 4 * ti;
#pragma unroll
        for (int d = 0; d < JM_D; ++d) {
          if (d < nd) {  // wave-uniform
            const f32x4 a = *reinterpret_cast<const f32x4*>(a0 + doff[d]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[d][i][j] = fmaf(a[i], b[j], acc[d][i][j]);
          }
        }
        w += nsl;
        while (w >= W) w -= W, ++r;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < JM_D; ++d) {  // (no early exit: the accumulators must stay statically indexed = in registers)
    if (d >= nd) continue;          // block-uniform
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[sl * kp * kp + (4 * ti + i) * kp + 4 * tj + j] = acc[d][i][j];
    }
    __syncthreads();
    float* dst = part + ((size_t)(d0 + d) * gridDim.x + blockIdx.x) * (k * k);
    for (int e = tid; e < k * k; e += 256) {
      const int i = e / k, j = e - i * k;
      float s = 0.f;
      for (int q = 0; q < nsl; ++q) s += red[q * kp * kp + i * kp + j];
      dst[e] = s;
    }
  }
}

// J[e] = scale * sum_b part[d][b][r]: 4 elements per block, 64 slices of the block partials each,
// fixed slice order (deterministic)
__global__ void __launch_bounds__(256)
    joint_reduce_kernel(const float* __restrict__ part, float* __restrict__ J, int TT, int nblk,
                        int kk, double scale) {
  __shared__ double sred[64][4];
  const int el = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int e = blockIdx.x * 4 + el;
  double s = 0.0;
  if (e < TT * kk) {
    const int d = e / kk, r = e - d * kk;
    for (int b = sl; b < nblk; b += 64) s += (double)part[((size_t)d * nblk + b) * kk + r];
  }
  sred[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < TT * kk) {
    double t = 0.0;
    for (int q = 0; q < 64; ++q) t += sred[q][el];
    J[e] = (float)(t * scale);
  }
}

// ---------------------------------------------------------------- joint backward
// which = 0: dx1[pix][i] = g*scale * sum_{d,j} dJ[d][i][j] * x2[pix - disp(d)][j]
// which = 1: dx2[pix][j] = g*scale * sum_{d,i} dJ[d][i][j] * x1[pix + disp(d)][i]
__global__ void __launch_bounds__(256)
    joint_bwd_kernel(const float* __restrict__ other, const float* __restrict__ dJ,
                     const float* __restrict__ gscale, float* __restrict__ dx, int N, int H, int W,
                     int k, int pad, float scale, int which) {
  extern __shared__ float sdj[];  // [T*T][k][k]
  const int T = 2 * pad + 1, TT = T * T;
  for (int e = threadIdx.x; e < TT * k * k; e += 256) sdj[e] = dJ[e];
  __syncthreads();
  const float g = gscale[0] * scale;
  const long total = (long)N * H * W * k;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % k);
    const long pix = e / k;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    float s = 0.f;
    for (int d = 0; d < TT; ++d) {
      const int du = d / T - pad, dv = d % T - pad;
      const int hh = which ? h + du : h - du, ww = which ? w + dv : w - dv;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      const long q = which ? pix + (long)du * W + dv : pix - (long)du * W - dv;
      const float* o = other + q * k;
      const float* m = sdj + d * k * k;
      if (which) {
        for (int i = 0; i < k; ++i) s = fmaf(m[i * k + c], o[i], s);
      } else {
        for (int j = 0; j < k; ++j) s = fmaf(m[c * k + j], o[j], s);
      }
    }
    dx[e] = g * s;
  }
}

// k % 4 == 0: one thread per (pixel, 4 output channels), 16-byte global and LDS accesses
__global__ void __launch_bounds__(256)
    joint_bwd4_kernel(const float* __restrict__ other, const float* __restrict__ dJ,
                      const float* __restrict__ gscale, float* __restrict__ dx, int N, int H, int W,
                      int k, int pad, float scale, int which) {
  extern __shared__ __attribute__((aligned(16))) float sdj[];  // [T*T][k][k]
  const int T = 2 * pad + 1, TT = T * T, k4 = k >> 2;
  for (int e = threadIdx.x; e < TT * k * k; e += 256) sdj[e] = dJ[e];
  __syncthreads();
  const float g = gscale[0] * scale;
  const long total = (long)N * H * W * k4;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % k4) * 4;
    const long pix = e / k4;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < TT; ++d) {
      const int du = d / T - pad, dv = d % T - pad;
      const int hh = which ? h + du : h - du, ww = which ? w + dv : w - dv;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      const long q = which ? pix + (long)du * W + dv : pix - (long)du * W - dv;
      const float* o = other + q * k;
      const float* m = sdj + d * k * k;
      for (int j4 = 0; j4 < k4; ++j4) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(o + 4 * j4);
        if (which) {  // s[c..c+3] += sum_i M[i][c..c+3] * o[i]
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f32x4 mv = *reinterpret_cast<const f32x4*>(m + (4 * j4 + t) * k + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = fmaf(mv[u], ov[t], s[u]);
          }
        } else {  // s[c+u] += sum_j M[c+u][j] * o[j]
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 mv = *reinterpret_cast<const f32x4*>(m + (c + u) * k + 4 * j4);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[u] = fmaf(mv[t], ov[t], s[u]);
          }
        }
      }
    }
    f32x4 r = {g * s[0], g * s[1], g * s[2], g * s[3]};
    *reinterpret_cast<f32x4*>(dx + pix * k + c) = r;
  }
}

// ---------------------------------------------------------------- loss on the joint (one block)
__device__ __forceinline__ double block_sum(double v, double* sred) {
  const int tid = threadIdx.x;
  __syncthreads();
  sred[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sred[tid] += sred[tid + o];
    __syncthreads();
  }
  const double r = sred[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_min(double v, double* sred) {
  const int tid = threadIdx.x;
  __syncthreads();
  sred[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sred[tid] = fmin(sred[tid], sred[tid + o]);
    __syncthreads();
  }
  const double r = sred[0];
  __syncthreads();
  return r;
}

// mode 0: segmentation, padding 0 (no normalisation)      mode 1: segmentation, padding > 0
// mode 2: vectors (IIDLoss): symmetrise + normalise to 1
// ws: 4 arrays of TT*kk floats (p1, p2/p, dP, tmp) + (2*TT*k) marginals
// out[0] = loss, out[1] = loss with lambda = 1 (mode 2 only); P = final joint; dJ = dloss/dJ.
__global__ void __launch_bounds__(256)
    iid_loss_kernel(const float* __restrict__ J, float* __restrict__ out, float* __restrict__ P,
                    float* __restrict__ dJ, float* __restrict__ ws, int TT, int k, int mode,
                    int symmetric, float lamda, float eps) {
  __shared__ double sred[256];
  const int tid = threadIdx.x;
  const int kk = k * k, n = TT * kk;
  float* p1 = ws;            // per-displacement normalised (mode 1) / raw
  float* p = ws + n;         // final joint
  float* dP = ws + 2 * n;    // dloss/dp, then dloss/dp2
  float* row = ws + 4 * n;   // r_i  [TT][k]
  float* col = row + TT * k; // c_j  [TT][k]
  float* S = col + TT * k;   // [TT] displacement sums (mode 1)

  // ---- forward normalisations
  double Z = 1.0;
  if (mode == 1) {
    double mn = 1e300;
    for (int e = tid; e < n; e += 256) mn = fmin(mn, (double)J[e]);
    mn = block_min(mn, sred);
    for (int d = 0; d < TT; ++d) {
      double s = 0.0;
      for (int e = tid; e < kk; e += 256) s += (double)J[d * kk + e] - mn + 1e-8;
      s = block_sum(s, sred);
      if (tid == 0) S[d] = (float)s;
      for (int e = tid; e < kk; e += 256) p1[d * kk + e] = (float)(((double)J[d * kk + e] - mn + 1e-8) / s);
    }
  } else {
    for (int e = tid; e < n; e += 256) p1[e] = J[e];
  }
  __syncthreads();
  {  // symmetrise into p (unnormalised), then global normalisation for modes 1, 2
    double z = 0.0;
    for (int e = tid; e < n; e += 256) {
      const int d = e / kk, r = e - d * kk, i = r / k, j = r - i * k;
      float v = p1[e];
      if (symmetric) v = 0.5f * (v + p1[d * kk + j * k + i]);
      p[e] = v;
      z += (double)v;
    }
    if (mode != 0) {
      Z = block_sum(z, sred);
      for (int e = tid; e < n; e += 256) p[e] = (float)((double)p[e] / Z);
    }
  }
  __syncthreads();
  // ---- marginals per displacement
  for (int e = tid; e < TT * k; e += 256) {
    const int d = e / k, a = e - d * k;
    double r = 0.0, c = 0.0;
    for (int b = 0; b < k; ++b) {
      r += (double)p[d * kk + a * k + b];
      c += (double)p[d * kk + b * k + a];
    }
    row[e] = (float)r;
    col[e] = (float)c;
  }
  __syncthreads();
  // ---- loss and dloss/dp
  const double invTT = mode == 1 ? 1.0 / (double)TT : 1.0;
  double l = 0.0, l1 = 0.0;
  for (int e = tid; e < n; e += 256) {
    const int d = e / kk, r = e - d * kk, a = r / k, b = r - a * k;
    const double pv = p[e], ra = row[d * k + a], cb = col[d * k + b];
    const double lp = log(pv + eps), lr = log(ra + eps), lc = log(cb + eps);
    l += -pv * (lp - lamda * lc - lamda * lr);
    l1 += -pv * (lp - lc - lr);
    const double g = -(lp - lamda * lc - lamda * lr) - pv / (pv + eps) + lamda * cb / (cb + eps) +
                     lamda * ra / (ra + eps);
    dP[e] = (float)(g * invTT);
    if (P) P[e] = (float)pv;
  }
  l = block_sum(l, sred);
  l1 = block_sum(l1, sred);
  if (tid == 0) {
    out[0] = (float)(l * invTT);
    out[1] = (float)(l1 * invTT);
  }
  if (!dJ) return;
  // ---- backward through the normalisations
  if (mode != 0) {  // p = p2 / Z
    double dot = 0.0;
    for (int e = tid; e < n; e += 256) dot += (double)dP[e] * (double)p[e];
    dot = block_sum(dot, sred);
    for (int e = tid; e < n; e += 256) dP[e] = (float)(((double)dP[e] - dot) / Z);
  }
  __syncthreads();
  float* dp1 = ws + 3 * n;
  for (int e = tid; e < n; e += 256) {
    const int d = e / kk, r = e - d * kk, i = r / k, j = r - i * k;
    dp1[e] = symmetric ? 0.5f * (dP[e] + dP[d * kk + j * k + i]) : dP[e];
  }
  __syncthreads();
  if (mode == 1) {
    for (int d = 0; d < TT; ++d) {
      double dot = 0.0;
      for (int e = tid; e < kk; e += 256) dot += (double)dp1[d * kk + e] * (double)p1[d * kk + e];
      dot = block_sum(dot, sred);
      const double s = S[d];
      for (int e = tid; e < kk; e += 256) dJ[d * kk + e] = (float)(((double)dp1[d * kk + e] - dot) / s);
    }
  } else {
    for (int e = tid; e < n; e += 256) dJ[e] = dp1[e];
  }
}

inline int joint_blocks(long npix) {
  long b = (npix + 1023) / 1024;  // >= 3 blocks per CU at the config sizes
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" {

int cy_group_softmax_fwd(const float* logits, float* probs, long M, int S, int k, float invT,
                         void* stream) {
  if (!logits || !probs || M <= 0 || S <= 0 || k <= 0) return CY_ERR_ARG;
  const size_t smem = (size_t)GS_PB * (S * k + 1) * sizeof(float);
  if (smem > 64 * 1024) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(group_softmax_fwd_kernel, dim3(cy_cdiv(M, GS_PB)), dim3(256), smem,
                     (hipStream_t)stream, logits, probs, M, S, k, invT);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_group_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, long M, int S,
                         int k, float invT, void* stream) {
  if (!probs || !dprobs || !dlogits || M <= 0 || S <= 0 || k <= 0) return CY_ERR_ARG;
  const size_t smem = (size_t)2 * GS_PB * (S * k + 1) * sizeof(float);
  if (smem > 64 * 1024) return CY_ERR_SHAPE;
  hipLaunchKernelGGL(group_softmax_bwd_kernel, dim3(cy_cdiv(M, GS_PB)), dim3(256), smem,
                     (hipStream_t)stream, probs, dprobs, dlogits, M, S, k, invT);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_joint_ws_bytes(int N, int H, int W, int k, int pad) {
  const int T = 2 * pad + 1;
  return (size_t)T * T * joint_blocks((long)N * H * W) * k * k * sizeof(float);
}

/* J[T*T][k][k] = scale * sum over pixels of x1[shifted] (x) x2 ; scale = 1/(N*H*W) when
 * normalise != 0 (the padding-0 segmentation joint), else 1. */
int cy_joint_fwd(const float* x1, const float* x2, float* J, int N, int H, int W, int k, int pad,
                 int normalise, void* ws, size_t ws_bytes, void* stream) {
  if (!x1 || !x2 || !J || N <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  if (k < 1 || k > J_KMAX || pad < 0 || pad >= H || pad >= W) return CY_ERR_SHAPE;
  if (!ws || ws_bytes < cy_joint_ws_bytes(N, H, W, k, pad)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long npix = (long)N * H * W;
  const int nblk = joint_blocks(npix), T = 2 * pad + 1;
  const int kp = (k + 3) & ~3, k4 = kp / 4, nsl = 256 / (k4 * k4);
  const size_t smem = (size_t)(2 * J_P * kp + nsl * kp * kp) * sizeof(float);
  // padding > 0: every displacement from one staged tile of R rows, if it fits the LDS
  int R = 256 / W;
  R = R < 1 ? 1 : (R > H ? H : R);
  size_t smem_multi = ((size_t)(R + 2 * pad) * (W + 2 * pad) * kp + (size_t)R * W * kp) * sizeof(float);
  if (smem_multi < (size_t)nsl * kp * kp * sizeof(float)) smem_multi = (size_t)nsl * kp * kp * sizeof(float);
  if (pad > 0 && smem_multi <= 150 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(joint_fwd_multi_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return CY_ERR_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(joint_fwd_multi_kernel, dim3(nblk, cy_cdiv(T * T, JM_D)), dim3(256), smem_multi, st, x1, x2,
                       (float*)ws, N, H, W, k, pad, R);
  } else {
    hipLaunchKernelGGL(joint_fwd_kernel, dim3(nblk, T * T), dim3(256), smem, st, x1, x2, (float*)ws,
                       N, H, W, k, pad);
  }
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(joint_reduce_kernel, dim3(cy_cdiv((long)T * T * k * k, 4)), dim3(256), 0, st,
                     (const float*)ws, J, T * T, nblk, k * k,
                     normalise ? 1.0 / (double)npix : 1.0);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_joint_bwd(const float* x1, const float* x2, const float* dJ, const float* gscale, float* dx1,
                 float* dx2, int N, int H, int W, int k, int pad, int normalise, void* stream) {
  if (!x1 || !x2 || !dJ || !gscale || N <= 0 || H <= 0 || W <= 0) return CY_ERR_ARG;
  if (k < 1 || k > J_KMAX || pad < 0) return CY_ERR_SHAPE;
  const int T = 2 * pad + 1;
  const size_t smem = (size_t)T * T * k * k * sizeof(float);
  if (smem > 64 * 1024) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long npix = (long)N * H * W;
  const float scale = normalise ? (float)(1.0 / (double)npix) : 1.f;
  const bool v4 = (k & 3) == 0;
  const int grid = grid_for(v4 ? npix * (k / 4) : npix * k);
  for (int which = 0; which < 2; ++which) {
    float* dst = which ? dx2 : dx1;
    const float* other = which ? x1 : x2;
    if (!dst) continue;
    if (v4)
      hipLaunchKernelGGL(joint_bwd4_kernel, dim3(grid), dim3(256), smem, st, other, dJ, gscale, dst, N,
                         H, W, k, pad, scale, which);
    else
      hipLaunchKernelGGL(joint_bwd_kernel, dim3(grid), dim3(256), smem, st, other, dJ, gscale, dst, N,
                         H, W, k, pad, scale, which);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

size_t cy_iid_loss_ws_bytes(int TT, int k) {
  return ((size_t)4 * TT * k * k + 2 * (size_t)TT * k + TT) * sizeof(float);
}

int cy_iid_loss(const float* J, float* out2, float* P, float* dJ, int TT, int k, int mode,
                int symmetric, float lamda, float eps, void* ws, size_t ws_bytes, void* stream) {
  if (!J || !out2 || TT <= 0 || k <= 0) return CY_ERR_ARG;
  if (mode < 0 || mode > 2 || k > J_KMAX) return CY_ERR_SHAPE;
  if (!ws || ws_bytes < cy_iid_loss_ws_bytes(TT, k)) return CY_ERR_WORKSPACE;
  hipLaunchKernelGGL(iid_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, J, out2, P, dJ,
                     (float*)ws, TT, k, mode, symmetric, lamda, eps);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
