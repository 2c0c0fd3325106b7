// Stacked cluster heads on the matrix cores: `DenseClusterHead` / `ClusterHead` with head_type="linear"
// (contrastyou/projectors/heads.py:125-173): S sub-heads x k clusters = K <= 128 outputs of one 1x1 conv /
// linear layer over C <= 64 input channels, followed by softmax(logits / T) inside each sub-head
// (SoftmaxWithT, projectors/nn.py:35-44).
//
// The VALU pair it replaces (head1x1 + group_softmax, DESIGN round 1: 0.9 + 0.32 ms forward, 1.46 + 0.65 ms
// backward for 16 x 224 x 224 pixels, K = 100) wrote the f32 logits (400 B per pixel) and read them back; here
// a wave owns 32 pixels at a time, the logits live in the exact-f32 MFMA accumulators and one LDS tile, and only
// the probabilities (forward) or dx / the weight-gradient slabs (backward) reach HBM:
//   forward   logits[K x 32 px] = W[K x C] x^T  (v_mfma_f32_32x32x2_f32, weights as the row operand)
//             -> LDS tile [px][K] -> per (pixel, sub-head) softmax -> probs[S][pixels][k]
//   backward  dlogits from (probs, dprobs) per (pixel, sub-head) into the LDS tile, then
//             dx[C x 32 px] = W^T dlogits^T,  dW[K x C] += dlogits^T x,  db[K] += column sums
//             (per-wave accumulators over all its tiles; slabs summed in fixed order by a second kernel).
#include "cy_common.h"

namespace {

constexpr int CH_KP = 128;          // couts, padded
// pitch of a logits tile row (floats): the smallest p >= K with p % 8 == 4 -- K = 100 instead of the padded 128 lets eight
// waves' tiles fit the LDS next to the weights (C = 32), i.e. two waves per SIMD instead of one; rows are 16-byte
// aligned, so the epilogue, the softmax rows and the linear mode's output move the tile in 16-byte LDS accesses (it went
// through LDS one float at a time with the odd pitch of rounds 2-3), and eight consecutive rows start in the eight
// 16-byte bank groups
__host__ __device__ inline int ch_ldl(int K) { return ((K + 3) & ~7) + 4; }
constexpr int CH_LDL = CH_KP + 1;   // backward: padded couts take part in the products

template <typename T> __device__ __forceinline__ void ch_load8(const T* p, float* f) {
  if constexpr (sizeof(T) == 2) {
    Chunk<T>::unpack(ld16(p), f);
  } else {
    Chunk<float>::unpack(ld16(p), f);
    Chunk<float>::unpack(ld16(p + 4), f + 4);
  }
}

constexpr int CH_KMAX = 32;  // clusters per sub-head, at most
// k consecutive floats of a [..][k] tensor <-> registers, all issued before the first use (a per-element loop
// exposes one memory latency per element); KQ = k rounded up to a multiple of 4 is a template parameter so that
// only the last quad carries `j < k` predicates (32 predicated elements cost 60 SGPR pairs and spilled them);
// 16-byte accesses when k % 4 == 0 (rows are then 16-byte aligned)
template <int KQ> __device__ __forceinline__ void ch_row_load(const float* __restrict__ p, float* v, int k) {
  if ((k & 3) == 0) {
#pragma unroll
    for (int q = 0; q < KQ / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      v[4 * q] = t[0], v[4 * q + 1] = t[1], v[4 * q + 2] = t[2], v[4 * q + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < KQ; ++j) v[j] = (j < KQ - 4 || j < k) ? p[j < k ? j : 0] : 0.f;
  }
}
template <int KQ> __device__ __forceinline__ void ch_row_store(float* __restrict__ p, const float* v, int k) {
  if ((k & 3) == 0) {
#pragma unroll
    for (int q = 0; q < KQ / 4; ++q)
      *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else {
#pragma unroll
    for (int j = 0; j < KQ; ++j)
      if (j < KQ - 4 || j < k) p[j] = v[j];
  }
}

// softmax(./T) of the rows of one tile: lane -> (pixel, sub-head) pairs
template <int KQ>
__device__ __forceinline__ void ch_softmax_rows(const float* sL, int ldl, float* __restrict__ probs, long M, long p0,
                                                int S, int k, float invT, int lane) {
  for (int t = lane; t < 32 * S; t += 64) {
    const int px = t & 31, s = t >> 5;
    if (p0 + px >= M) continue;
    const float* lr = sL + px * ldl + s * k;
    float v[KQ];
    if ((k & 3) == 0) {  // (rows and sub-head offsets are multiples of four floats)
#pragma unroll
      for (int q = 0; q < KQ / 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(lr + 4 * q);
        v[4 * q] = t4[0], v[4 * q + 1] = t4[1], v[4 * q + 2] = t4[2], v[4 * q + 3] = t4[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < KQ; ++j) v[j] = lr[j];  // (columns beyond k: the next sub-head's / padding, masked below)
    }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < KQ; ++j) m = (j < KQ - 4 || j < k) ? fmaxf(m, v[j]) : m;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
      v[j] = (j < KQ - 4 || j < k) ? expf((v[j] - m) * invT) : 0.f;
      sum += v[j];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < KQ; ++j) v[j] *= inv;
    ch_row_store<KQ>(probs + ((size_t)s * M + p0 + px) * k, v, k);
  }
}

// dlogits rows of one tile from (probs, dprobs)
template <int KQ>
__device__ __forceinline__ void ch_dlogit_rows(float* sL, int ldl, const float* __restrict__ probs,
                                               const float* __restrict__ dprobs, long M, long p0, int S, int k,
                                               float invT, int lane) {
  for (int t = lane; t < 32 * S; t += 64) {
    const int px = t & 31, s = t >> 5;
    float* lr = sL + px * ldl + s * k;
    float pv[KQ], gv[KQ];
    if (p0 + px < M) {
      ch_row_load<KQ>(probs + ((size_t)s * M + p0 + px) * k, pv, k);
      ch_row_load<KQ>(dprobs + ((size_t)s * M + p0 + px) * k, gv, k);
    } else {
#pragma unroll
      for (int j = 0; j < KQ; ++j) pv[j] = gv[j] = 0.f;
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < KQ; ++j) dot = fmaf(pv[j], gv[j], dot);  // (elements beyond k are zero)
#pragma unroll
    for (int j = 0; j < KQ; ++j)
      if (j < KQ - 4 || j < k) lr[j] = pv[j] * (gv[j] - dot) * invT;
  }
}

#define CY_CH_KQ_SWITCH(k_, CALL)             \
  switch (((k_) + 3) / 4) {                   \
    case 1: { constexpr int KQ = 4; CALL; } break;   \
    case 2: { constexpr int KQ = 8; CALL; } break;   \
    case 3: { constexpr int KQ = 12; CALL; } break;  \
    case 4: { constexpr int KQ = 16; CALL; } break;  \
    case 5: { constexpr int KQ = 20; CALL; } break;  \
    case 6: { constexpr int KQ = 24; CALL; } break;  \
    case 7: { constexpr int KQ = 28; CALL; } break;  \
    default: { constexpr int KQ = 32; CALL; } break; \
  }

// x rows p0 .. p0+31 -> sX[32][C + 1] f32 (zeros beyond M); one wave
template <typename T>
__device__ __forceinline__ void ch_stage_x(const T* __restrict__ x, float* sX, long p0, long M, int C, int lane) {
  const int g8 = C / 8;
  for (int e = lane; e < 32 * g8; e += 64) {
    const int rr = e / g8, c8 = (e % g8) * 8;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p0 + rr < M) ch_load8<T>(x + (size_t)(p0 + rr) * C + c8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) sX[rr * (C + 1) + c8 + j] = f[j];
  }
}

__device__ __forceinline__ void ch_stage_w(const float* __restrict__ w, float* sW, int K, int C, int tid, int nthr) {
  for (int e = tid; e < CH_KP * C; e += nthr) {
    const int kk = e / C, c = e % C;
    sW[kk * (C + 1) + c] = kk < K ? w[(size_t)kk * C + c] : 0.f;
  }
}

// LINEAR (cy_head1x1_fwd with more than 16 outputs, e.g. DenseClusterHead(normalize=True)): no softmax, the tile's logits
// [32 px][K] -- one contiguous run of the [M][K] output -- go out as they stand in LDS, 256 contiguous bytes per store
template <typename T, bool LINEAR = false>
__global__ void __launch_bounds__(512)
    cluster_head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                            float* __restrict__ probs, long M, int C, int K, int S, int k, float invT) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sW = sm;                                   // [128][C + 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = blockDim.x >> 6, ldl = ch_ldl(K);
  float* sX = sW + CH_KP * (C + 1) + wave * (32 * (C + 1) + 32 * ldl);  // this wave's [32][C + 1]
  float* sL = sX + 32 * (C + 1);                                         // and [32][K | 1]
  float* sBias = sW + CH_KP * (C + 1) + nwave * (32 * (C + 1) + 32 * ldl);  // [128]
  const int i = lane & 31, kk = lane >> 5;
  ch_stage_w(w, sW, K, C, tid, blockDim.x);
  if (tid < CH_KP) sBias[tid] = (b && tid < K) ? b[tid] : 0.f;
  __syncthreads();
  const long ntile = (M + 31) / 32;
  for (long tile = (long)blockIdx.x * nwave + wave; tile < ntile; tile += (long)gridDim.x * nwave) {
    const long p0 = tile * 32;
    ch_stage_x<T>(x, sX, p0, M, C, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
    const float* pb = sX + i * (C + 1) + kk;
    for (int ks = 0; ks < C; ks += 2) {
      const float bv = pb[ks];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sW[(nb * 32 + i) * (C + 1) + ks + kk], bv, acc[nb], 0, 0, 0);
    }
    const int K4 = (K + 3) & ~3;  // (<= ldl: a quad that starts below K4 lies inside the row; columns >= K are never read)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = four consecutive couts
        const int co0 = nb * 32 + 8 * g + 4 * kk;
        if (co0 < K4) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + co0);
          *reinterpret_cast<f32x4*>(sL + i * ldl + co0) =
              f32x4{acc[nb][4 * g] + b4[0], acc[nb][4 * g + 1] + b4[1], acc[nb][4 * g + 2] + b4[2], acc[nb][4 * g + 3] + b4[3]};
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (LINEAR) {
      const long rows = M - p0 < 32 ? M - p0 : 32;
      float* o = probs + (size_t)p0 * K;
      if ((K & 3) == 0) {  // quads: 16-byte LDS reads, 16-byte stores (the tile starts at a multiple of 32 K floats)
        const int qpr = K >> 2, nq = (int)rows * qpr;
        int px = 0, cq = lane;
        while (cq >= qpr) cq -= qpr, ++px;
        for (int e = lane; e < nq; e += 64) {
          *reinterpret_cast<f32x4*>(o + 4 * e) = *reinterpret_cast<const f32x4*>(sL + px * ldl + 4 * cq);
          cq += 64;
          while (cq >= qpr) cq -= qpr, ++px;
        }
      } else {
        int px = 0, co = lane;  // element e = px * K + co of the tile, e = lane, lane + 64, ...
        while (co >= K) co -= K, ++px;
        for (int e = lane; e < (int)rows * K; e += 64) {
          o[e] = sL[px * ldl + co];
          co += 64;
          while (co >= K) co -= K, ++px;
        }
      }
    } else {
      CY_CH_KQ_SWITCH(k, ch_softmax_rows<KQ>(sL, ldl, probs, M, p0, S, k, invT, lane));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// slab layout per wave: [K_pad 128][C] dW then [128] db
// LINEAR (cy_head1x1_bwd, wide): `dprobs` is dlogits [M][K] itself, copied into the tile
template <typename T, bool LINEAR = false>
__global__ void __launch_bounds__(256)
    cluster_head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ probs,
                            const float* __restrict__ dprobs, T* __restrict__ dx, float* __restrict__ slabs, long M,
                            int C, int K, int S, int k, float invT, int need_dx, int need_dw) {
  extern __shared__ float sm[];
  float* sW = sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sX = sW + CH_KP * (C + 1) + wave * (32 * (C + 1) + 32 * CH_LDL);
  float* sL = sX + 32 * (C + 1);
  const int i = lane & 31, kk = lane >> 5;
  const int ncb = C / 32;  // 32-channel blocks (host: C in {32, 64})
  ch_stage_w(w, sW, K, C, tid, 256);
  __syncthreads();
  f32x16 accw[4][2];  // dW blocks [cout block][channel block]
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 16; ++q) accw[nb][cb][q] = 0.f;
  float dbv[2] = {0.f, 0.f};  // couts lane and lane + 64
  const long ntile = (M + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntile; tile += (long)gridDim.x * 4) {
    const long p0 = tile * 32;
    // dlogits of the tile: softmax backward per (pixel, sub-head); padding couts and pixels beyond M are zero
    for (int e = lane; e < 32 * (CH_KP - K); e += 64)
      sL[(e % 32) * CH_LDL + K + e / 32] = 0.f;
    if constexpr (LINEAR) {
      // the tile is one contiguous, 16-byte aligned run of 32 K floats: every lane requests its (up to 16) quads before
      // the first one is used (a load -> LDS store loop pays one memory latency per iteration), then scatters them
      // into the padded rows; element e = 4 (lane + 64 t) + j sits at (e / K, e % K), advanced by 256 per round
      const long rows = M - p0 < 32 ? M - p0 : 32;
      const int n = (int)rows * K;
      const float* d = dprobs + (size_t)p0 * K;
      f32x4 v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e0 = 4 * (lane + 64 * t);
        if (e0 + 4 <= n) {
          v[t] = *reinterpret_cast<const f32x4*>(d + e0);
        } else {
          v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (e0 + j < n) v[t][j] = d[e0 + j];
        }
      }
      const int d256 = 256 / K, r256 = 256 - d256 * K;
      int px[4], co[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * lane + j;
        px[j] = e / K;
        co[j] = e - px[j] * K;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (px[j] < 32) sL[px[j] * CH_LDL + co[j]] = v[t][j];
          px[j] += d256, co[j] += r256;
          if (co[j] >= K) co[j] -= K, ++px[j];
        }
      }
    } else {
      CY_CH_KQ_SWITCH(k, ch_dlogit_rows<KQ>(sL, CH_LDL, probs, dprobs, M, p0, S, k, invT, lane));
    }
    if (need_dw) ch_stage_x<T>(x, sX, p0, M, C, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (need_dx) {  // dx[c][px] = sum_co W[co][c] dl[px][co]: rows = channels, columns = pixels
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (cb >= ncb) break;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        for (int ks = 0; ks < CH_KP; ks += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sW[(ks + kk) * (C + 1) + cb * 32 + i],
                                                     sL[i * CH_LDL + ks + kk], acc, 0, 0, 0);
        if (p0 + i < M) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {  // four runs of four consecutive channels
            const int c0 = cb * 32 + 8 * g + 4 * kk;
            T* d = dx + (size_t)(p0 + i) * C + c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = from_f32<T>(acc[4 * g + j]);
          }
        }
      }
    }
    if (need_dw) {  // dW[co][c] += sum_px dl[px][co] x[px][c]: rows = couts, columns = channels, K = pixels
      for (int ks = 0; ks < 32; ks += 2) {
        const float* lrow = sL + (ks + kk) * CH_LDL;
        const float* xrow = sX + (ks + kk) * (C + 1);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if (cb >= ncb) break;
          const float bv = xrow[cb * 32 + i];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            accw[nb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(lrow[nb * 32 + i], bv, accw[nb][cb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float sacc = 0.f;
        for (int px = 0; px < 32; ++px) sacc += sL[px * CH_LDL + hf * 64 + lane];
        dbv[hf] += sacc;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (need_dw) {
    float* slab = slabs + ((size_t)blockIdx.x * 4 + wave) * ((size_t)CH_KP * C + CH_KP);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (cb >= ncb) break;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int co = nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
          slab[(size_t)co * C + cb * 32 + i] = accw[nb][cb][reg];
        }
      }
    slab[(size_t)CH_KP * C + lane] = dbv[0];
    slab[(size_t)CH_KP * C + 64 + lane] = dbv[1];
  }
}

// dw / db = the sum of the workgroups' slabs, in two ordered stages: blockIdx.y sums a contiguous group of slabs into
// stage[g][per], a second launch sums the CH_RG groups.  (One stage with one thread per element walked up to 2048
// slabs through dependent loads on 17 workgroups: 635 us.)
constexpr int CH_RG = 64;

__global__ void __launch_bounds__(256)
    cluster_head_slab_partial_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ stage, int per) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= per) return;
  const int gsz = (nslab + CH_RG - 1) / CH_RG;
  const int q0 = blockIdx.y * gsz;
  const int q1 = q0 + gsz < nslab ? q0 + gsz : nslab;
  float s = 0.f;
#pragma unroll 8
  for (int q = q0; q < q1; ++q) s += slabs[(size_t)q * per + e];
  stage[(size_t)blockIdx.y * per + e] = s;
}

__global__ void __launch_bounds__(256)
    cluster_head_slab_reduce_kernel(const float* __restrict__ stage, float* __restrict__ dw, float* __restrict__ db,
                                    int K, int C, int accumulate) {
  const int per = CH_KP * C + CH_KP;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per; e += gridDim.x * 256) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < CH_RG; ++q) s += stage[(size_t)q * per + e];
    if (e < CH_KP * C) {
      if (e / C < K && dw) dw[e] = accumulate ? dw[e] + s : s;
    } else if (e - CH_KP * C < K && db) {
      db[e - CH_KP * C] = accumulate ? db[e - CH_KP * C] + s : s;
    }
  }
}

inline int ch_blocks(long M) {
  long nb = ((M + 31) / 32 + 3) / 4;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}
inline size_t ch_fwd_smem_w(int C, int K, int nw) {
  return ((size_t)CH_KP * (C + 1) + nw * (32 * (C + 1) + 32 * ch_ldl(K)) + CH_KP) * sizeof(float);
}
// eight waves where their tiles fit the 160 KB next to the weights (C = 32 up to K = 108), four otherwise
inline int ch_fwd_waves(int C, int K) { return ch_fwd_smem_w(C, K, 8) <= 160 * 1024 ? 8 : 4; }
inline size_t ch_fwd_smem(int C, int K) { return ch_fwd_smem_w(C, K, ch_fwd_waves(C, K)); }
inline size_t ch_smem(int C) { return ((size_t)CH_KP * (C + 1) + 4 * (32 * (C + 1) + 32 * CH_LDL) + CH_KP) * sizeof(float); }

template <typename K_>
int ch_set_smem(K_ kern, int C) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)ch_smem(C)) == hipSuccess
             ? CY_OK
             : CY_ERR_LAUNCH;
}

}  // namespace

extern "C" {

/* K = S * k <= 128 outputs, C in {32, 64}, k <= 32 */
static int ch_check(long M, int C, int K, int S, int k) {
  if (M <= 0 || S <= 0 || k <= 0 || k > CH_KMAX || K != S * k || K > CH_KP) return CY_ERR_SHAPE;
  if (C != 32 && C != 64) return CY_ERR_SHAPE;
  return CY_OK;
}

}  // extern "C"

template <bool LINEAR>
static int ch_fwd_impl(const void* x, const float* w, const float* b, float* out, long M, int C, int K, int S, int k,
                       float invT, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int nw = ch_fwd_waves(C, K);
  long gb = ((M + 31) / 32 + nw - 1) / nw;
  const int grid = (int)(gb > 256 ? 256 : gb);  // one block per CU (LDS): grid-stride over the pixel tiles
  const size_t smem = ch_fwd_smem(C, K);
#define CY_CH_FWD(TT)                                                                                         \
  do {                                                                                                        \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_head_fwd_kernel<TT, LINEAR>),               \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)            \
      return CY_ERR_LAUNCH;                                                                                   \
    hipLaunchKernelGGL((cluster_head_fwd_kernel<TT, LINEAR>), dim3(grid), dim3(64 * nw), smem, st, (const TT*)x, w, b, \
                       out, M, C, K, S, k, invT);                                                             \
  } while (0)
  if (dtype == CY_BF16) CY_CH_FWD(bf16);
  else if (dtype == CY_F16) CY_CH_FWD(f16);
  else if (dtype == CY_F32) CY_CH_FWD(float);
  else return CY_ERR_DTYPE;
#undef CY_CH_FWD
  CY_CHECK_LAUNCH();
  return CY_OK;
}

extern "C" {

int cy_cluster_head_fwd(const void* x, const float* w, const float* b, float* probs, long M, int C, int K, int S,
                        int k, float invT, int dtype, void* stream) {
  if (!x || !w || !probs) return CY_ERR_ARG;
  int rc = ch_check(M, C, K, S, k);
  if (rc != CY_OK) return rc;
  return ch_fwd_impl<false>(x, w, b, probs, M, C, K, S, k, invT, dtype, stream);
}

size_t cy_cluster_head_bwd_ws_bytes(long M, int C) {
  return ((size_t)ch_blocks(M) * 4 + CH_RG) * ((size_t)CH_KP * C + CH_KP) * sizeof(float);
}

}  // extern "C"

template <bool LINEAR>
static int ch_bwd_impl(const void* x, const float* w, const float* probs, const float* dprobs, void* dx, float* dw,
                       float* db, int accumulate, long M, int C, int K, int S, int k, float invT, int dtype, void* ws,
                       size_t ws_bytes, void* stream) {
  int rc;
  const int need_dx = dx != nullptr, need_dw = dw != nullptr || db != nullptr;
  if (need_dw && (!ws || ws_bytes < cy_cluster_head_bwd_ws_bytes(M, C))) return CY_ERR_WORKSPACE;
  if (!need_dx && !need_dw) return CY_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = ch_blocks(M);
  const size_t smem = ch_smem(C);
#define CY_CH_BWD(TT)                                                                                          \
  do {                                                                                                         \
    if ((rc = ch_set_smem(cluster_head_bwd_kernel<TT, LINEAR>, 64)) != CY_OK) return rc;                       \
    hipLaunchKernelGGL((cluster_head_bwd_kernel<TT, LINEAR>), dim3(grid), dim3(256), smem, st, (const TT*)x, w, probs, \
                       dprobs, (TT*)dx, (float*)ws, M, C, K, S, k, invT, need_dx, need_dw);                    \
  } while (0)
  if (dtype == CY_BF16) CY_CH_BWD(bf16);
  else if (dtype == CY_F16) CY_CH_BWD(f16);
  else if (dtype == CY_F32) CY_CH_BWD(float);
  else return CY_ERR_DTYPE;
#undef CY_CH_BWD
  CY_CHECK_LAUNCH();
  if (need_dw) {
    const int per = CH_KP * C + CH_KP;
    float* stage = (float*)ws + (size_t)grid * 4 * per;
    hipLaunchKernelGGL(cluster_head_slab_partial_kernel, dim3(cy_cdiv(per, 256), CH_RG), dim3(256), 0, st,
                       (const float*)ws, grid * 4, stage, per);
    CY_CHECK_LAUNCH();
    hipLaunchKernelGGL(cluster_head_slab_reduce_kernel, dim3(cy_cdiv(per, 256)), dim3(256), 0, st,
                       (const float*)stage, dw, db, K, C, accumulate);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

extern "C" {

int cy_cluster_head_bwd(const void* x, const float* w, const float* probs, const float* dprobs, void* dx, float* dw,
                        float* db, long M, int C, int K, int S, int k, float invT, int dtype, void* ws,
                        size_t ws_bytes, void* stream) {
  if (!x || !w || !probs || !dprobs) return CY_ERR_ARG;
  int rc = ch_check(M, C, K, S, k);
  if (rc != CY_OK) return rc;
  if (dw && !db) return CY_ERR_WORKSPACE;
  return ch_bwd_impl<false>(x, w, probs, dprobs, dx, dw, db, 0, M, C, K, S, k, invT, dtype, ws, ws_bytes, stream);
}

}  // extern "C"

// ---- the wide 1x1 head (cy_head1x1_fwd / _bwd with 16 < K <= 128 outputs over 32 or 64 channels) on the kernels above:
// called from cy_head_loss.hip, same shared object
bool cy_head_wide_ok(int C, int K) { return (C == 32 || C == 64) && K > 16 && K <= CH_KP; }
size_t cy_head_wide_bwd_ws_bytes(long M, int C) { return cy_cluster_head_bwd_ws_bytes(M, C); }
int cy_head_wide_fwd(const void* x, const float* w, const float* b, float* logits, long M, int C, int K, int dtype,
                     void* stream) {
  return ch_fwd_impl<true>(x, w, b, logits, M, C, K, 1, K, 1.f, dtype, stream);
}
int cy_head_wide_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db, int accumulate,
                     long M, int C, int K, int dtype, void* ws, size_t ws_bytes, void* stream) {
  return ch_bwd_impl<true>(x, w, nullptr, dlogits, dx, dw, db, accumulate, M, C, K, 1, K, 1.f, dtype, ws, ws_bytes, stream);
}
