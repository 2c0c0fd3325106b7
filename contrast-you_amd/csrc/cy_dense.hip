// Dense (pixel-wise) contrastive projector and point sampling.
//   DenseProjectionHead: Conv1x1(C,hid) -> LeakyReLU(0.01) -> Conv1x1(hid,out)
//                        -> AdaptiveAvgPool2d((s,s)) -> L2 normalise over channels
//                                                 contrastyou/projectors/heads.py:31-41,99-123
//   region_extractor:    5 (h,w) points per image of the s x s map
//                                                 semi_seg/hooks/infonce.py:31-46
//
// The second 1x1 convolution and the average pooling are both linear, so they commute:
//     pool(W2 * lrelu(W1 x + b1) + b2) = W2 * pool(lrelu(W1 x + b1)) + b2.
// This file therefore computes  hpool[bin][hid] = mean_{px in bin} lrelu(W1 x[px] + b1)  directly
// from the NHWC feature map -- the [pixels][hid] intermediate (1.6 GB f32 at 32 x 256 x 224^2) is
// never written -- and leaves the [bins][hid] x [hid][out] product and the normalisation to the
// small dense kernels of cy_contrast.hip.  A bin list restricts the work to the bins the loss
// actually samples (5 per image in the dense InfoNCE hook); null means all N*s*s bins.
//
// Adaptive pooling bins (torch semantics): rows [floor(i*H/s), ceil((i+1)*H/s)), same for
// columns; neighbouring bins overlap by at most one pixel when s <= H.
#include <cstdlib>

#include "cy_conv_tile.h"

namespace {

constexpr int DP_PT = 16;  // pixels per register tile
constexpr int DP_CB = 32;  // channels per staged chunk

struct BinRect {
  int n, r0, r1, c0, c1, colour;
};

__device__ __forceinline__ BinRect bin_rect(const int32_t* bins, int b, int sh, int sw, int H,
                                            int W) {
  int n, i, j;
  if (bins) {
    n = bins[3 * b], i = bins[3 * b + 1], j = bins[3 * b + 2];
  } else {
    j = b % sw;
    const int t = b / sw;
    i = t % sh;
    n = t / sh;
  }
  BinRect r;
  r.n = n;
  r.r0 = (i * H) / sh;
  r.r1 = ((i + 1) * H + sh - 1) / sh;
  r.c0 = (j * W) / sw;
  r.c1 = ((j + 1) * W + sw - 1) / sw;
  r.colour = (i & 1) | ((j & 1) << 1);
  return r;
}

__device__ __forceinline__ long bin_pixel(const BinRect& R, int q, int bw, int H, int W) {
  const int qr = q / bw, qc = q - qr * bw;
  return ((long)R.n * H + R.r0 + qr) * W + R.c0 + qc;
}

// xs[p][c] (f32) <- x[pixel pt0+p of the bin][cc + c], zeros outside the bin / channel range
template <typename T>
__device__ __forceinline__ void stage_pixels(const T* __restrict__ x, int ldx, const BinRect& R,
                                             int bw, int npx, int H, int W, int pt0, int cc, int C,
                                             float* xs, int tid) {
  constexpr int EPC = ElemTr<T>::EPC;
  constexpr int CPP = DP_CB / EPC;  // 16-byte chunks per pixel
  if (tid < DP_PT * CPP) {
    const int p = tid / CPP, k = tid - p * CPP;
    const int q = pt0 + p, c = cc + k * EPC;
    float f[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] = 0.f;
    if (q < npx && c < C) Chunk<T>::unpack(ld16(x + bin_pixel(R, q, bw, H, W) * ldx + c), f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) xs[p * DP_CB + k * EPC + e] = f[e];
  }
}

__device__ __forceinline__ void load_wrow(const float* __restrict__ w1, int o, bool ov, int cc,
                                          int C, float* wreg) {
#pragma unroll
  for (int c = 0; c < DP_CB; ++c) wreg[c] = (ov && cc + c < C) ? w1[(size_t)o * C + cc + c] : 0.f;
}

__device__ __forceinline__ void fma_tile(const float* xs, const float* wreg, float* acc) {
#pragma unroll
  for (int p = 0; p < DP_PT; ++p) {
#pragma unroll
    for (int c4 = 0; c4 < DP_CB / 4; ++c4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + p * DP_CB + c4 * 4);  // broadcast
      acc[p] = fmaf(xv[0], wreg[4 * c4], acc[p]);
      acc[p] = fmaf(xv[1], wreg[4 * c4 + 1], acc[p]);
      acc[p] = fmaf(xv[2], wreg[4 * c4 + 2], acc[p]);
      acc[p] = fmaf(xv[3], wreg[4 * c4 + 3], acc[p]);
    }
  }
}

// ---------------------------------------------------------------- forward: one block per bin
template <typename T>
__global__ void __launch_bounds__(256)
    dense_proj_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w1,
                          const float* __restrict__ b1, const int32_t* __restrict__ bins,
                          float* __restrict__ hpool, int H, int W, int C, int ldx, int hid, int sh,
                          int sw, float slope) {
  __shared__ __attribute__((aligned(16))) float xs[DP_PT * DP_CB];
  const int tid = threadIdx.x;
  const BinRect R = bin_rect(bins, blockIdx.x, sh, sw, H, W);
  const int bw = R.c1 - R.c0, npx = (R.r1 - R.r0) * bw;
  const float inv = 1.f / (float)npx;
  const bool single = C <= DP_CB;
  for (int o0 = 0; o0 < hid; o0 += 256) {
    const int o = o0 + tid;
    const bool ov = o < hid;
    const float bias = ov ? b1[o] : 0.f;
    float wreg[DP_CB];
    if (single) load_wrow(w1, o, ov, 0, C, wreg);
    float sum = 0.f;
    for (int pt0 = 0; pt0 < npx; pt0 += DP_PT) {
      float acc[DP_PT];
#pragma unroll
      for (int p = 0; p < DP_PT; ++p) acc[p] = bias;
      for (int cc = 0; cc < C; cc += DP_CB) {
        __syncthreads();
        stage_pixels<T>(x, ldx, R, bw, npx, H, W, pt0, cc, C, xs, tid);
        if (!single) load_wrow(w1, o, ov, cc, C, wreg);
        __syncthreads();
        fma_tile(xs, wreg, acc);
      }
#pragma unroll
      for (int p = 0; p < DP_PT; ++p)
        if (pt0 + p < npx) sum += acc[p] > 0.f ? acc[p] : slope * acc[p];
    }
    if (ov) hpool[(size_t)blockIdx.x * hid + o] = sum * inv;
  }
}

// ---------------------------------------------------------------- backward
// Persistent blocks over the bins of ONE colour class ((i&1, j&1): bins of a class never share a
// pixel), so the read-modify-write of dx needs no atomics and the result is deterministic.
// Thread o owns hidden unit o (hid <= 256): recomputes pre[p][o], forms
//   g[p][o] = dhpool[bin][o]/|bin| * lrelu'(pre[p][o]),
// accumulates dW1[o][cb..cb+32) and db1[o] in registers across all bins of the block, and the
// block reduces dx[p][c] = sum_o g[p][o] W1[o][c] through LDS.
// Two instantiations keep the live register set small: DW = true produces dW1/db1 partials,
// DW = false produces dx (each recomputes the cheap pre-activations).
template <typename T, bool DW>
__global__ void __launch_bounds__(256)
    dense_proj_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w1,
                          const float* __restrict__ b1, const int32_t* __restrict__ bins, int nb,
                          const float* __restrict__ dhpool, T* __restrict__ dx,
                          float* __restrict__ part, int H, int W, int C, int ldx, int hid, int sh,
                          int sw, float slope, int colour) {
  __shared__ __attribute__((aligned(16))) float xs[DP_PT * DP_CB];
  __shared__ __attribute__((aligned(16))) float gs[DW ? 4 : DP_PT * 256];
  __shared__ float ps[DW ? 4 : DP_PT * 8 * 32];
  const int tid = threadIdx.x;
  const int o = tid;
  const bool ov = o < hid;
  const float bias = ov ? b1[o] : 0.f;
  const bool single = C <= DP_CB;
  const int cl = tid & 31, og = tid >> 5;
  float* mypart = part + (size_t)blockIdx.x * ((size_t)hid * C + hid);
  float wreg[DP_CB];
  if (single) load_wrow(w1, o, ov, 0, C, wreg);

  for (int cb = 0; cb < C; cb += DP_CB) {
    float dwacc[DW ? DP_CB : 1];
#pragma unroll
    for (int c = 0; c < (DW ? DP_CB : 1); ++c) dwacc[c] = 0.f;
    float dbacc = 0.f;
    float wcol[DW ? 1 : 32];  // W1[og*32 + i][cb + cl]
    if constexpr (!DW) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int oo = og * 32 + i;
        wcol[i] = (oo < hid && cb + cl < C) ? w1[(size_t)oo * C + cb + cl] : 0.f;
      }
    }
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
      const BinRect R = bin_rect(bins, b, sh, sw, H, W);
      if (colour >= 0 && R.colour != colour) continue;  // uniform over the block
      const int bw = R.c1 - R.c0, npx = (R.r1 - R.r0) * bw;
      const float gb = ov ? dhpool[(size_t)b * hid + o] / (float)npx : 0.f;
      for (int pt0 = 0; pt0 < npx; pt0 += DP_PT) {
        float acc[DP_PT];
#pragma unroll
        for (int p = 0; p < DP_PT; ++p) acc[p] = bias;
        for (int cc = 0; cc < C; cc += DP_CB) {
          __syncthreads();
          stage_pixels<T>(x, ldx, R, bw, npx, H, W, pt0, cc, C, xs, tid);
          if (!single) load_wrow(w1, o, ov, cc, C, wreg);
          __syncthreads();
          fma_tile(xs, wreg, acc);
        }
        if (DW && !single) {  // bring channel block cb back for the dW products
          __syncthreads();
          stage_pixels<T>(x, ldx, R, bw, npx, H, W, pt0, cb, C, xs, tid);
          __syncthreads();
        }
        // g overwrites the pre-activations (one live register tile)
#pragma unroll
        for (int p = 0; p < DP_PT; ++p) {
          acc[p] = (pt0 + p < npx) ? gb * (acc[p] > 0.f ? 1.f : slope) : 0.f;
          if constexpr (!DW) gs[p * 256 + tid] = acc[p];
        }
        if constexpr (DW) {
          if (cb == 0) {
#pragma unroll
            for (int p = 0; p < DP_PT; ++p) dbacc += acc[p];
          }
#pragma unroll
          for (int p = 0; p < DP_PT; ++p) {
#pragma unroll
            for (int c4 = 0; c4 < DP_CB / 4; ++c4) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + p * DP_CB + c4 * 4);
              dwacc[4 * c4] = fmaf(acc[p], xv[0], dwacc[4 * c4]);
              dwacc[4 * c4 + 1] = fmaf(acc[p], xv[1], dwacc[4 * c4 + 1]);
              dwacc[4 * c4 + 2] = fmaf(acc[p], xv[2], dwacc[4 * c4 + 2]);
              dwacc[4 * c4 + 3] = fmaf(acc[p], xv[3], dwacc[4 * c4 + 3]);
            }
          }
        } else {
          __syncthreads();
          // partial over this thread's 32 hidden units, for channel cb+cl, of every pixel of the tile
#pragma unroll
          for (int p = 0; p < DP_PT; ++p) acc[p] = 0.f;
#pragma unroll
          for (int i4 = 0; i4 < 8; ++i4) {
#pragma unroll
            for (int p = 0; p < DP_PT; ++p) {
              const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + p * 256 + og * 32 + i4 * 4);
              acc[p] = fmaf(gv[0], wcol[4 * i4], acc[p]);
              acc[p] = fmaf(gv[1], wcol[4 * i4 + 1], acc[p]);
              acc[p] = fmaf(gv[2], wcol[4 * i4 + 2], acc[p]);
              acc[p] = fmaf(gv[3], wcol[4 * i4 + 3], acc[p]);
            }
          }
#pragma unroll
          for (int p = 0; p < DP_PT; ++p) ps[(p * 8 + og) * 32 + cl] = acc[p];
          __syncthreads();
#pragma unroll
          for (int pr = 0; pr < DP_PT; pr += 8) {
            const int p = pr + og;
            if (pt0 + p < npx && cb + cl < C) {
              float s = 0.f;
#pragma unroll
              for (int q = 0; q < 8; ++q) s += ps[(p * 8 + q) * 32 + cl];
              T* d = dx + bin_pixel(R, pt0 + p, bw, H, W) * ldx + cb + cl;
              *d = from_f32<T>(to_f32<T>(*d) + s);
            }
          }
        }
      }
    }
    if constexpr (DW) {
      if (ov) {
#pragma unroll
        for (int c = 0; c < DP_CB; ++c)
          if (cb + c < C) mypart[(size_t)o * C + cb + c] = dwacc[c];
        if (cb == 0) mypart[(size_t)hid * C + o] = dbacc;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    slot_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                       float* __restrict__ db, int slots, int nw, int nbias, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nw + nbias) return;
  double s = 0.0;
  for (int q = 0; q < slots; ++q) s += (double)part[(size_t)q * (nw + nbias) + e];
  float* dst = e < nw ? (dw ? dw + e : nullptr) : (db ? db + (e - nw) : nullptr);
  if (dst) *dst = accumulate ? *dst + (float)s : (float)s;
}

// ---------------------------------------------------------------- plain adaptive average pool
template <typename T>
__global__ void __launch_bounds__(256)
    adaptive_pool_fwd_kernel(const T* __restrict__ x, const int32_t* __restrict__ bins,
                             float* __restrict__ out, int H, int W, int C, int ldx, int sh, int sw) {
  const BinRect R = bin_rect(bins, blockIdx.x, sh, sw, H, W);
  const int bw = R.c1 - R.c0, npx = (R.r1 - R.r0) * bw;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int q = 0; q < npx; ++q) s += to_f32<T>(x[bin_pixel(R, q, bw, H, W) * ldx + c]);
    out[(size_t)blockIdx.x * C + c] = s / (float)npx;
  }
}

// gather form (all bins): dx[n,r,c,:] = sum over the <=2x2 bins containing (r,c) of dpool/|bin|
template <typename T>
__global__ void __launch_bounds__(256)
    adaptive_pool_bwd_kernel(const float* __restrict__ dpool, T* __restrict__ dx, int N, int H,
                             int W, int C, int ldx, int sh, int sw) {
  const long total = (long)N * H * W * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % C);
    const long pix = e / C;
    const int w = (int)(pix % W);
    const long t = pix / W;
    const int h = (int)(t % H), n = (int)(t / H);
    float s = 0.f;
    const int i_hi = min(sh - 1, (int)(((long)(h + 1) * sh - 1) / H));
    const int j_hi = min(sw - 1, (int)(((long)(w + 1) * sw - 1) / W));
    for (int i = max(0, i_hi - 2); i <= i_hi; ++i) {
      const int r0 = (i * H) / sh, r1 = ((i + 1) * H + sh - 1) / sh;
      if (h < r0 || h >= r1) continue;
      for (int j = max(0, j_hi - 2); j <= j_hi; ++j) {
        const int c0 = (j * W) / sw, c1 = ((j + 1) * W + sw - 1) / sw;
        if (w < c0 || w >= c1) continue;
        s += dpool[(((size_t)n * sh + i) * sw + j) * C + c] / (float)((r1 - r0) * (c1 - c0));
      }
    }
    dx[pix * ldx + c] = from_f32<T>(s);
  }
}

// ---------------------------------------------------------------- adaptive max pool (nn.AdaptiveMaxPool2d, projectors/nn.py:16-23)
// out[bin][c] = max over the bin (torch's bin boundaries: bin_rect), arg = pixel index inside the image of the FIRST
// maximum in row-major order (torch's `val > max` scan); a NaN wins like in torch
template <typename T>
__global__ void __launch_bounds__(256)
    adaptive_maxpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int32_t* __restrict__ arg, int H,
                                int W, int C, int ldx, int sh, int sw) {
  const BinRect R = bin_rect(nullptr, blockIdx.x, sh, sw, H, W);
  const int bw = R.c1 - R.c0, npx = (R.r1 - R.r0) * bw;
  for (int c = threadIdx.x; c < C; c += 256) {
    float best = -INFINITY;
    int bi = -1;
    for (int q = 0; q < npx; ++q) {
      const size_t px = bin_pixel(R, q, bw, H, W);
      const float v = to_f32<T>(x[px * ldx + c]);
      if (bi < 0 || v > best || v != v) {
        if (!(best != best)) best = v, bi = (int)(px % ((size_t)H * W));  // (a NaN already held stays)
      }
    }
    out[(size_t)blockIdx.x * C + c] = best;
    arg[(size_t)blockIdx.x * C + c] = bi;
  }
}

// gather form: dx[n,r,c,:] = sum over the <= 3 x 3 bins containing (r, c) of dpool where that bin's arg-max is (r, c)
template <typename T>
__global__ void __launch_bounds__(256)
    adaptive_maxpool_bwd_kernel(const float* __restrict__ dpool, const int32_t* __restrict__ arg, T* __restrict__ dx,
                                int N, int H, int W, int C, int ldx, int sh, int sw) {
  const long total = (long)N * H * W * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % C);
    const long pix = e / C;
    const int w = (int)(pix % W);
    const long t = pix / W;
    const int h = (int)(t % H), n = (int)(t / H);
    const int me = h * W + w;
    float s = 0.f;
    const int i_hi = min(sh - 1, (int)(((long)(h + 1) * sh - 1) / H));
    const int j_hi = min(sw - 1, (int)(((long)(w + 1) * sw - 1) / W));
    for (int i = max(0, i_hi - 2); i <= i_hi; ++i) {
      const int r0 = (i * H) / sh, r1 = ((i + 1) * H + sh - 1) / sh;
      if (h < r0 || h >= r1) continue;
      for (int j = max(0, j_hi - 2); j <= j_hi; ++j) {
        const int c0 = (j * W) / sw, c1 = ((j + 1) * W + sw - 1) / sw;
        if (w < c0 || w >= c1) continue;
        const size_t b = (((size_t)n * sh + i) * sw + j) * C + c;
        if (arg[b] == me) s += dpool[b];
      }
    }
    dx[pix * ldx + c] = from_f32<T>(s);
  }
}

// ---------------------------------------------------------------- row gather / scatter
__global__ void __launch_bounds__(256)
    gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                       float* __restrict__ out, int M, int D) {
  const long total = (long)M * D;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int m = (int)(e / D), d = (int)(e - (long)m * D);
    out[e] = src[(size_t)idx[m] * D + d];
  }
}
__global__ void __launch_bounds__(256)
    scatter_rows_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                        float* __restrict__ dsrc, int M, int D) {
  const long total = (long)M * D;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int m = (int)(e / D), d = (int)(e - (long)m * D);
    dsrc[(size_t)idx[m] * D + d] = dout[e];  // indices are distinct (checked by the host)
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

inline int dp_bwd_blocks(int nb) { return nb < 256 ? nb : 256; }

}  // namespace

#include "cy_dense_mfma.h"

extern "C" {

int cy_dense_proj_fwd(const void* x, const float* w1, const float* b1, const int32_t* bins, int nb,
                      float* hpool, int N, int H, int W, int C, int ldx, int hid, int sh, int sw,
                      float slope, int dtype, void* stream) {
  if (!x || !w1 || !b1 || !hpool || N <= 0 || nb <= 0) return CY_ERR_ARG;
  if (C % 8 || ldx < C || ldx % 8 || hid <= 0 || sh <= 0 || sw <= 0 || sh > H || sw > W)
    return CY_ERR_SHAPE;
  if (!bins && nb != N * sh * sw) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dpm_applicable(dtype, C, hid, slope)) {
    if (dtype == CY_BF16)
      return dpm_launch_fwd<bf16>((const bf16*)x, w1, b1, bins, nb, hpool, H, W, ldx, C, hid, sh, sw, slope, st);
    return dpm_launch_fwd<f16>((const f16*)x, w1, b1, bins, nb, hpool, H, W, ldx, C, hid, sh, sw, slope, st);
  }
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(dense_proj_fwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x, w1,
                       b1, bins, hpool, H, W, C, ldx, hid, sh, sw, slope);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(dense_proj_fwd_kernel<f16>, dim3(nb), dim3(256), 0, st, (const f16*)x, w1,
                       b1, bins, hpool, H, W, C, ldx, hid, sh, sw, slope);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(dense_proj_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x,
                       w1, b1, bins, hpool, H, W, C, ldx, hid, sh, sw, slope);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_dense_proj_bwd_ws_bytes(int nb, int C, int hid) {
  const size_t valu = (size_t)dp_bwd_blocks(nb) * ((size_t)hid * C + hid) * sizeof(float);
  // workgroup partials + the job table (cells: N (2 sh - 1)(2 sw - 1) < 4 nb records)
  const size_t mfma = (size_t)DPM_WGS * DPM_PART * sizeof(float) + (size_t)4 * (nb > 0 ? nb : 1) * sizeof(DpRec);
  return valu > mfma ? valu : mfma;
}

int cy_dense_proj_bwd(const void* x, const float* w1, const float* b1, const int32_t* bins, int nb,
                      const float* dhpool, void* dx, float* dw1, float* db1, int accumulate, int N,
                      int H, int W, int C, int ldx, int hid, int sh, int sw, float slope, int dtype,
                      void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w1 || !b1 || !dhpool || N <= 0 || nb <= 0) return CY_ERR_ARG;
  if (C % 8 || ldx < C || ldx % 8 || hid <= 0 || hid > 256 || sh <= 0 || sw <= 0 || sh > H ||
      sw > W)
    return CY_ERR_SHAPE;
  if (!bins && nb != N * sh * sw) return CY_ERR_SHAPE;
  if (dtype != CY_BF16 && dtype != CY_F32 && dtype != CY_F16) return CY_ERR_DTYPE;
  if (!ws || ws_bytes < cy_dense_proj_bwd_ws_bytes(nb, C, hid)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (dpm_applicable(dtype, C, hid, slope)) {
    if (dtype == CY_BF16)
      return dpm_launch_bwd<bf16>((const bf16*)x, w1, b1, bins, nb, dhpool, (bf16*)dx, dw1, db1, accumulate, N, H, W,
                                  ldx, C, hid, sh, sw, slope, (float*)ws, st);
    return dpm_launch_bwd<f16>((const f16*)x, w1, b1, bins, nb, dhpool, (f16*)dx, dw1, db1, accumulate, N, H, W, ldx,
                               C, hid, sh, sw, slope, (float*)ws, st);
  }
  const int G = dp_bwd_blocks(nb);
  const size_t slot = (size_t)hid * C + hid;
  // dW1/db1: one launch over all bins (no pixel is written); dx: one launch per colour class
  if (dw1 || db1) {
    if (dtype == CY_BF16)
      hipLaunchKernelGGL((dense_proj_bwd_kernel<bf16, true>), dim3(G), dim3(256), 0, st, (const bf16*)x,
                         w1, b1, bins, nb, dhpool, (bf16*)nullptr, (float*)ws, H, W, C, ldx, hid, sh, sw,
                         slope, -1);
    else if (dtype == CY_F16)
      hipLaunchKernelGGL((dense_proj_bwd_kernel<f16, true>), dim3(G), dim3(256), 0, st, (const f16*)x,
                         w1, b1, bins, nb, dhpool, (f16*)nullptr, (float*)ws, H, W, C, ldx, hid, sh, sw,
                         slope, -1);
    else
      hipLaunchKernelGGL((dense_proj_bwd_kernel<float, true>), dim3(G), dim3(256), 0, st,
                         (const float*)x, w1, b1, bins, nb, dhpool, (float*)nullptr, (float*)ws, H, W, C,
                         ldx, hid, sh, sw, slope, -1);
    CY_CHECK_LAUNCH();
  }
  for (int colour = 0; dx && colour < 4; ++colour) {
    if (dtype == CY_BF16)
      hipLaunchKernelGGL((dense_proj_bwd_kernel<bf16, false>), dim3(G), dim3(256), 0, st,
                         (const bf16*)x, w1, b1, bins, nb, dhpool, (bf16*)dx, (float*)ws, H, W, C, ldx,
                         hid, sh, sw, slope, colour);
    else if (dtype == CY_F16)
      hipLaunchKernelGGL((dense_proj_bwd_kernel<f16, false>), dim3(G), dim3(256), 0, st,
                         (const f16*)x, w1, b1, bins, nb, dhpool, (f16*)dx, (float*)ws, H, W, C, ldx,
                         hid, sh, sw, slope, colour);
    else
      hipLaunchKernelGGL((dense_proj_bwd_kernel<float, false>), dim3(G), dim3(256), 0, st,
                         (const float*)x, w1, b1, bins, nb, dhpool, (float*)dx, (float*)ws, H, W, C, ldx,
                         hid, sh, sw, slope, colour);
    CY_CHECK_LAUNCH();
  }
  if (dw1 || db1) {
    hipLaunchKernelGGL(slot_reduce_kernel, dim3(cy_cdiv((long)slot, 256)), dim3(256), 0, st,
                       (const float*)ws, dw1, db1, G, hid * C, hid, accumulate);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_adaptive_avgpool_fwd(const void* x, const int32_t* bins, int nb, float* out, int N, int H,
                            int W, int C, int ldx, int sh, int sw, int dtype, void* stream) {
  if (!x || !out || N <= 0 || nb <= 0) return CY_ERR_ARG;
  if (ldx < C || sh <= 0 || sw <= 0 || sh > H || sw > W) return CY_ERR_SHAPE;
  if (!bins && nb != N * sh * sw) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x,
                       bins, out, H, W, C, ldx, sh, sw);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel<f16>, dim3(nb), dim3(256), 0, st, (const f16*)x,
                       bins, out, H, W, C, ldx, sh, sw);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel<float>, dim3(nb), dim3(256), 0, st,
                       (const float*)x, bins, out, H, W, C, ldx, sh, sw);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_adaptive_avgpool_bwd(const float* dpool, void* dx, int N, int H, int W, int C, int ldx,
                            int sh, int sw, int dtype, void* stream) {
  if (!dpool || !dx || N <= 0) return CY_ERR_ARG;
  if (ldx < C || sh <= 0 || sw <= 0 || sh > H || sw > W) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * H * W * C);
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, dpool,
                       (bf16*)dx, N, H, W, C, ldx, sh, sw);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel<f16>, dim3(grid), dim3(256), 0, st, dpool,
                       (f16*)dx, N, H, W, C, ldx, sh, sw);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, dpool,
                       (float*)dx, N, H, W, C, ldx, sh, sw);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_adaptive_maxpool_fwd(const void* x, float* out, int32_t* arg, int N, int H, int W, int C, int ldx, int sh,
                            int sw, int dtype, void* stream) {
  if (!x || !out || !arg || N <= 0) return CY_ERR_ARG;
  if (ldx < C || sh <= 0 || sw <= 0 || sh > H || sw > W) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int nb = N * sh * sw;
#define CY_AMP(TT)                                                                                             \
  hipLaunchKernelGGL(adaptive_maxpool_fwd_kernel<TT>, dim3(nb), dim3(256), 0, st, (const TT*)x, out, arg, H, W, \
                     C, ldx, sh, sw)
  if (dtype == CY_BF16) CY_AMP(bf16);
  else if (dtype == CY_F16) CY_AMP(f16);
  else if (dtype == CY_F32) CY_AMP(float);
  else return CY_ERR_DTYPE;
#undef CY_AMP
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_adaptive_maxpool_bwd(const float* dpool, const int32_t* arg, void* dx, int N, int H, int W, int C, int ldx,
                            int sh, int sw, int dtype, void* stream) {
  if (!dpool || !arg || !dx || N <= 0) return CY_ERR_ARG;
  if (ldx < C || sh <= 0 || sw <= 0 || sh > H || sw > W) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * H * W * C);
#define CY_AMP(TT)                                                                                              \
  hipLaunchKernelGGL(adaptive_maxpool_bwd_kernel<TT>, dim3(grid), dim3(256), 0, st, dpool, arg, (TT*)dx, N, H, W, \
                     C, ldx, sh, sw)
  if (dtype == CY_BF16) CY_AMP(bf16);
  else if (dtype == CY_F16) CY_AMP(f16);
  else if (dtype == CY_F32) CY_AMP(float);
  else return CY_ERR_DTYPE;
#undef CY_AMP
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_gather_rows_fwd(const float* src, const int32_t* idx, float* out, int M, int D,
                       void* stream) {
  if (!src || !idx || !out || M <= 0 || D <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)M * D)), dim3(256), 0,
                     (hipStream_t)stream, src, idx, out, M, D);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_gather_rows_bwd(const float* dout, const int32_t* idx, float* dsrc, int M, int D,
                       void* stream) {
  if (!dout || !idx || !dsrc || M <= 0 || D <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)M * D)), dim3(256), 0,
                     (hipStream_t)stream, dout, idx, dsrc, M, D);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
