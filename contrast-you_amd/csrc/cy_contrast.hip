// Projection head pieces and the InfoNCE / SupCon loss.
//   contrastyou/projectors/heads.py:14-22,81-96 + projectors/nn.py:47-54
//     AdaptiveAvgPool2d(1) -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise
//   contrastyou/losses/contrastive.py:14-20,31-100  (exp_sim_temperature, SupConLoss1)
// Everything here is f32: the embedding x embedding similarity and the
// G*P products run on the exact f32 MFMA (v_mfma_f32_32x32x2_f32).
#include "cy_common.h"

namespace {

// ---------------------------------------------------------------- sgemm (f32 MFMA)
// C[M][N] = alpha * A[M][K] * op(B).  64x64 block tile, 4 waves each a 32x32
// MFMA accumulator, K staged 16 at a time through LDS.
template <bool BT>
__global__ void __launch_bounds__(256)
    sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                 int M, int N, int K, float alpha) {
  __shared__ float sA[64][17];
  __shared__ float sB[16][65];  // [k][n]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int i = lane & 31, kk = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    __syncthreads();
    // A tile: 64 rows x 16 k
    for (int e = tid; e < 64 * 16; e += 256) {
      const int rr = e >> 4, kc = e & 15;
      const int gm = m0 + rr, gk = k0 + kc;
      sA[rr][kc] = (gm < M && gk < K) ? A[(size_t)gm * K + gk] : 0.f;
    }
    if (BT) {
      for (int e = tid; e < 64 * 16; e += 256) {
        const int nn = e >> 4, kc = e & 15;
        const int gn = n0 + nn, gk = k0 + kc;
        sB[kc][nn] = (gn < N && gk < K) ? B[(size_t)gn * K + gk] : 0.f;
      }
    } else {
      for (int e = tid; e < 64 * 16; e += 256) {
        const int kc = e >> 6, nn = e & 63;
        const int gn = n0 + nn, gk = k0 + kc;
        sB[kc][nn] = (gn < N && gk < K) ? B[(size_t)gk * N + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
      const float av = sA[wm * 32 + i][ks + kk];
      const float bv = sB[ks + kk][wn * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kk;
    const int gm = m0 + wm * 32 + row, gn = n0 + wn * 32 + i;
    if (gm < M && gn < N) C[(size_t)gm * N + gn] = alpha * acc[reg];
  }
}

int launch_sgemm(const float* A, const float* B, float* C, int M, int N, int K, float alpha,
                 int b_trans, hipStream_t st) {
  dim3 grid(cy_cdiv(N, 64), cy_cdiv(M, 64));
  if (b_trans)
    hipLaunchKernelGGL(sgemm_kernel<true>, grid, dim3(256), 0, st, A, B, C, M, N, K, alpha);
  else
    hipLaunchKernelGGL(sgemm_kernel<false>, grid, dim3(256), 0, st, A, B, C, M, N, K, alpha);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

// ---------------------------------------------------------------- SupCon
__device__ __forceinline__ bool is_pos(const int32_t* labels, const uint8_t* pm, int n, int i,
                                       int j) {
  const int a = i % n, b = j % n;
  if (labels) return labels[a] == labels[b];
  return pm[(size_t)a * n + b] != 0;
}

// one wave per row: row max (incl. diagonal), sum_{k!=i} exp(s-m), sum over positives, count
__global__ void __launch_bounds__(256)
    supcon_rowstats_kernel(const float* __restrict__ S, const int32_t* __restrict__ labels,
                           const uint8_t* __restrict__ pm, float* __restrict__ tmp, int n) {
  const int R = 2 * n;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* sr = S + (size_t)row * R;
  float m = -INFINITY;
  for (int j = lane; j < R; j += 64) m = fmaxf(m, sr[j]);
  m = wave_max(m);
  float l = 0.f, ps = 0.f, cnt = 0.f;
  for (int j = lane; j < R; j += 64) {
    if (j == row) continue;
    const float s = sr[j];
    l += expf(s - m);
    if (is_pos(labels, pm, n, row, j)) {
      ps += s;
      cnt += 1.f;
    }
  }
  l = wave_sum(l);
  ps = wave_sum(ps);
  cnt = wave_sum(cnt);
  if (lane == 0) {
    tmp[row * 4 + 0] = m;
    tmp[row * 4 + 1] = l;
    tmp[row * 4 + 2] = ps;
    tmp[row * 4 + 3] = cnt;
  }
}

// single block: global max, per-row denominators, mean loss
__global__ void __launch_bounds__(256)
    supcon_finalize_kernel(float* __restrict__ row_stats, float* __restrict__ loss, int R) {
  __shared__ float smax[256];
  __shared__ double ssum[256];
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int i = tid; i < R; i += 256) m = fmaxf(m, row_stats[i * 4 + 0]);
  smax[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) smax[tid] = fmaxf(smax[tid], smax[tid + o]);
    __syncthreads();
  }
  const float M = smax[0];
  double acc = 0.0;
  for (int i = tid; i < R; i += 256) {
    const float mi = row_stats[i * 4 + 0], li = row_stats[i * 4 + 1];
    const float ps = row_stats[i * 4 + 2], cnt = row_stats[i * 4 + 3];
    const float Di = li * expf(mi - M) + 1e-16f;
    const float li_loss = ps / cnt - M - logf(Di);  // cnt==0 -> NaN like the reference
    acc += (double)li_loss;
    row_stats[i * 4 + 0] = Di;
    row_stats[i * 4 + 1] = cnt;
    row_stats[i * 4 + 2] = ps;
    row_stats[i * 4 + 3] = M;
  }
  ssum[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) ssum[tid] += ssum[tid + o];
    __syncthreads();
  }
  if (tid == 0) loss[0] = (float)(-ssum[0] / (double)R);
}

// G + G^T with G_ij = -(g/R) [pos_ij/cnt_i - e_ij/D_i], zero diagonal
__global__ void __launch_bounds__(256)
    supcon_gsym_kernel(const float* __restrict__ S, const float* __restrict__ rs,
                       const int32_t* __restrict__ labels, const uint8_t* __restrict__ pm,
                       const float* __restrict__ gscale, float* __restrict__ G, int n) {
  const int R = 2 * n;
  const long total = (long)R * R;
  const float gs = gscale[0] / (float)R;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int i = (int)(e / R), j = (int)(e % R);
    float v = 0.f;
    if (i != j) {
      const float M = rs[3];
      const float eij = expf(S[e] - M);
      const float Di = rs[i * 4 + 0], Dj = rs[j * 4 + 0];
      const float ci = rs[i * 4 + 1], cj = rs[j * 4 + 1];
      const float pos = is_pos(labels, pm, n, i, j) ? 1.f : 0.f;
      v = -gs * (pos * (1.f / ci + 1.f / cj) - eij * (1.f / Di + 1.f / Dj));
    }
    G[e] = v;
  }
}

// ---- SupConLoss1(exclude_other_pos=True), contrastyou/losses/contrastive.py:87-91 --------------------------------------
// Per positive pair (i, j) the denominator holds that pair and the negatives only, the negatives' sum rescaled by
// kappa_i = 1 / (neg_i / (pos_i + neg_i) + 1e-4):
//   loss_i = (1 / c_i) sum_{j in pos_i} [ s_ij - log(e_ij + a_i) ],   a_i = kappa_i sum_{k in neg_i} e_ik + 1e-16,  e = exp(s - M)
// with M the global maximum (detached, as in the reference).  On the materialised similarity matrix: row maxima from
// supcon_rowstats_kernel, then one wave per row.  rs[row] = (a_i, c_i, kappa_i T_i / c_i, loss_i), T_i = sum_pos 1 / (e_ij + a_i).
__global__ void __launch_bounds__(256)
    supcon_excl_rows_kernel(const float* __restrict__ S, const int32_t* __restrict__ labels,
                            const uint8_t* __restrict__ pm, const float* __restrict__ tmp, float* __restrict__ rs,
                            float* __restrict__ Mout, int n) {
  const int R = 2 * n;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float M = -INFINITY;
  for (int i = lane; i < R; i += 64) M = fmaxf(M, tmp[i * 4]);  // (every wave: the row maxima are 4 R bytes)
  M = wave_max(M);
  const float* sr = S + (size_t)row * R;
  float ln = 0.f, cnt = 0.f;
  for (int j = lane; j < R; j += 64) {
    if (j == row) continue;
    if (is_pos(labels, pm, n, row, j)) cnt += 1.f;
    else ln += expf(sr[j] - M);
  }
  ln = wave_sum(ln);
  cnt = wave_sum(cnt);
  const float negc = (float)(R - 1) - cnt;
  const float kappa = 1.f / (negc / (cnt + negc) + 1e-4f);
  const float ai = ln * kappa + 1e-16f;
  float T = 0.f, ls = 0.f;
  for (int j = lane; j < R; j += 64) {
    if (j == row || !is_pos(labels, pm, n, row, j)) continue;
    const float sm = sr[j] - M;
    const float d = expf(sm) + ai;
    T += 1.f / d;
    ls += sm - logf(d);
  }
  T = wave_sum(T);
  ls = wave_sum(ls);
  if (lane == 0) {
    rs[row * 4 + 0] = ai;
    rs[row * 4 + 1] = cnt;
    rs[row * 4 + 2] = kappa * T / cnt;
    rs[row * 4 + 3] = ls / cnt;  // cnt == 0 -> NaN like the reference
    if (row == 0) Mout[0] = M;
  }
}

__global__ void __launch_bounds__(256)
    supcon_excl_loss_kernel(float* __restrict__ rs, float* __restrict__ loss, const float* __restrict__ Mptr, int R) {
  __shared__ double ssum[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < R; i += 256) acc += (double)rs[i * 4 + 3];
  ssum[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) ssum[threadIdx.x] += ssum[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = (float)(-ssum[0] / (double)R);
    rs[3] = Mptr[0];  // (where cy_supcon_matrices looks for the shift; row 0's loss term has been summed)
  }
}

// G + G^T with G_ij = -(g/R) d loss_i / d s_ij:  pos: (1/c_i) a_i / (e_ij + a_i);  neg: -e_ij kappa_i T_i / c_i
__global__ void __launch_bounds__(256)
    supcon_excl_gsym_kernel(const float* __restrict__ S, const float* __restrict__ rs, const float* __restrict__ Mptr,
                            const int32_t* __restrict__ labels, const uint8_t* __restrict__ pm,
                            const float* __restrict__ gscale, float* __restrict__ G, int n) {
  const int R = 2 * n;
  const long total = (long)R * R;
  const float gs = gscale[0] / (float)R;
  const float M = Mptr[0];
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int i = (int)(e / R), j = (int)(e % R);
    float v = 0.f;
    if (i != j) {
      const float eij = expf(S[e] - M);
      if (is_pos(labels, pm, n, i, j)) {
        const float ai = rs[i * 4 + 0], aj = rs[j * 4 + 0];
        v = -gs * (ai / ((eij + ai) * rs[i * 4 + 1]) + aj / ((eij + aj) * rs[j * 4 + 1]));
      } else {
        v = gs * eij * (rs[i * 4 + 2] + rs[j * 4 + 2]);
      }
    }
    G[e] = v;
  }
}

__global__ void __launch_bounds__(256)
    supcon_matrices_kernel(const float* __restrict__ S, const float* __restrict__ rs,
                           const int32_t* __restrict__ labels, const uint8_t* __restrict__ pm,
                           float* sl, float* se, float* po, float* ne, int n) {
  const int R = 2 * n;
  const long total = (long)R * R;
  const float M = rs[3];
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int i = (int)(e / R), j = (int)(e % R);
    const float s = S[e] - M;
    if (sl) sl[e] = s;
    if (se) se[e] = expf(s);
    const bool pos = is_pos(labels, pm, n, i, j);
    if (po) po[e] = (i != j && pos) ? 1.f : 0.f;
    if (ne) ne[e] = (i != j && !pos) ? 1.f : 0.f;
  }
}

// ---------------------------------------------------------------- SupCon, fused: S never leaves the CU
// The 2n x 2n similarity matrix S = P P^T / t (64 MB at the C5 size of 4096 embeddings) and the gradient
// matrix G of the same size are formed tile by tile in the MFMA accumulators and consumed there:
//   forward   row sums  l_i = sum_{k != i} exp(S_ik - M),  ps_i = sum_{pos} S_ik,  cnt_i   per 64-row block and
//             column split, with the reference's global shift M = max S known beforehand: S_ij <= max(S_ii, S_jj)
//             (Cauchy-Schwarz), so M = max_i |P_i|^2 / t from a row-norm pre-pass -- no online rescaling;
//   backward  dP_i = (1/t) sum_j G_ij P_j with G_ij = -(g/R)[pos_ij (1/c_i + 1/c_j) - e_ij (1/D_i + 1/D_j)]:
//             the S tile is recomputed, turned into the G tile in registers, passed through LDS into operand
//             layout and multiplied with the P_j block that is already resident for the first product.
// Exact f32 MFMA (v_mfma_f32_32x32x2_f32) as before; fixed summation orders (column splits are summed in order
// by the finalize / reduce kernels) => bitwise reproducible.  D <= 256.
constexpr int SC_TM = 64;  // rows / columns of a tile
// LDS image of a 64-row block of P: the k index split by parity, [k & 1][row][k >> 1] with row pitch D/2 + 4 floats
// and the two parity planes 16 floats out of step -- the f32 MFMA takes one k per half-wave (lane >> 5), so a lane's
// operands for four successive k-steps are one ds_read_b128, and a column-wise read (second product of the
// backward pass) alternates planes lane by lane without bank conflicts
__device__ __host__ __forceinline__ int sc_ld2(int D) { return D / 2 + 4; }
__device__ __host__ __forceinline__ int sc_plane(int D) { return SC_TM * sc_ld2(D) + 16; }
__device__ __host__ __forceinline__ int sc_block_floats(int D) { return 2 * sc_plane(D); }

__global__ void __launch_bounds__(256)
    supcon_diag_kernel(const float* __restrict__ P, float* __restrict__ diag, int R, int D, float inv_t) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* pr = P + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s = fmaf(pr[i], pr[i], s);
  s = wave_sum(s);
  if (lane == 0) diag[row] = s * inv_t;
}

__device__ __forceinline__ void sc_stage(const float* __restrict__ P, float* sdst, int r0, int R, int D, int tid) {
  // rows r0 .. r0+63 of P -> the parity-split image (zeros beyond R)
  const int dq = D / 4, ld2 = sc_ld2(D), pl = sc_plane(D);
  for (int e = tid; e < SC_TM * dq; e += 256) {
    const int rr = e / dq, c4 = (e % dq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + rr < R) v = *reinterpret_cast<const f32x4*>(P + (size_t)(r0 + rr) * D + c4);
    float* d = sdst + rr * ld2 + (c4 >> 1);
    d[0] = v[0], d[1] = v[2];
    d[pl] = v[1], d[pl + 1] = v[3];
  }
}

// S tile (64 x 64) of row block sPi x column block sPj into the four waves' 32x32 accumulators
__device__ __forceinline__ f32x16 sc_tile(const float* sPi, const float* sPj, int D, int wm, int wn, int i, int kk) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int ld2 = sc_ld2(D), pl = sc_plane(D);
  const float* pa = sPi + kk * pl + (wm * 32 + i) * ld2;
  const float* pb = sPj + kk * pl + (wn * 32 + i) * ld2;
  for (int k4 = 0; k4 < D / 2; k4 += 4) {  // four k-steps per pair of 16-byte reads
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(pa + k4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + k4);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[q], acc, 0, 0, 0);
  }
  return acc;
}

// positives with the row side hoisted out of the column loop: folded index (i mod n) and label per row
struct ScRow {
  int a;    // row mod n (R = 2n: one conditional subtraction, no division)
  int lab;  // label of the row (labels mode)
};
__device__ __forceinline__ ScRow sc_row(const int32_t* labels, int n, int row) {
  ScRow q;
  q.a = row >= n ? row - n : row;
  q.lab = (labels && row < 2 * n) ? labels[q.a] : 0;
  return q;
}
__device__ __forceinline__ bool sc_pos(const ScRow& r, const ScRow& c, const int32_t* labels, const uint8_t* pm, int n) {
  return labels ? r.lab == c.lab : pm[(size_t)r.a * n + c.a] != 0;
}

// grid (row blocks, column splits).  part[split][R][3] = (l, ps, cnt) of the split's columns
__global__ void __launch_bounds__(256)
    supcon_fused_fwd_kernel(const float* __restrict__ P, const int32_t* __restrict__ labels,
                            const uint8_t* __restrict__ pm, const float* __restrict__ Mptr,
                            float* __restrict__ part, int n, int D, float inv_t, int nsplit) {
  extern __shared__ float sm[];
  const int R = 2 * n;
  float* sPi = sm;
  float* sPj = sm + sc_block_floats(D);
  float* sred = sPj;  // reused at the end: [2 wn][64 rows][3]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.x * SC_TM;
  const int ncb = (R + SC_TM - 1) / SC_TM;
  const int cb0 = (ncb * (int)blockIdx.y) / nsplit, cb1 = (ncb * ((int)blockIdx.y + 1)) / nsplit;
  const float M = Mptr[0];
  sc_stage(P, sPi, m0, R, D, tid);
  float l[16], ps[16], cnt[16];
  ScRow rws[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    l[q] = ps[q] = cnt[q] = 0.f;
    rws[q] = sc_row(labels, n, m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk);
  }
  for (int cb = cb0; cb < cb1; ++cb) {
    const int n0 = cb * SC_TM;
    __syncthreads();  // previous tile's reads of sPj are done
    sc_stage(P, sPj, n0, R, D, tid);
    __syncthreads();
    const f32x16 acc = sc_tile(sPi, sPj, D, wm, wn, i, kk);
    const int col = n0 + wn * 32 + i;
    const ScRow cq = sc_row(labels, n, col);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
      if (row < R && col < R && row != col) {
        const float s = acc[reg] * inv_t;
        l[reg] += expf(s - M);
        if (sc_pos(rws[reg], cq, labels, pm, n)) {
          ps[reg] += s;
          cnt[reg] += 1.f;
        }
      }
    }
  }
  // row totals: over the 32 lanes that share kk (fixed xor tree), then over the two column halves (wn)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      l[reg] += __shfl_xor(l[reg], o, 64);
      ps[reg] += __shfl_xor(ps[reg], o, 64);
      cnt[reg] += __shfl_xor(cnt[reg], o, 64);
    }
  __syncthreads();
  if (i == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
      float* d = sred + (wn * SC_TM + rr) * 3;
      d[0] = l[reg], d[1] = ps[reg], d[2] = cnt[reg];
    }
  }
  __syncthreads();
  for (int e = tid; e < SC_TM * 3; e += 256) {
    const int rr = e / 3;
    if (m0 + rr < R) part[((size_t)blockIdx.y * R + m0 + rr) * 3 + e % 3] = sred[e] + sred[SC_TM * 3 + e];
  }
}

// single block: M = max diag (grid-wide value for the other kernels), then (after the fused pass) the loss
__global__ void __launch_bounds__(256)
    supcon_max_kernel(const float* __restrict__ diag, float* __restrict__ Mout, int R) {
  __shared__ float smax[256];
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int i = tid; i < R; i += 256) m = fmaxf(m, diag[i]);
  smax[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) smax[tid] = fmaxf(smax[tid], smax[tid + o]);
    __syncthreads();
  }
  if (tid == 0) Mout[0] = smax[0];
}

__global__ void __launch_bounds__(256)
    supcon_fused_finalize_kernel(const float* __restrict__ part, const float* __restrict__ Mptr,
                                 float* __restrict__ row_stats, float* __restrict__ loss, int R, int nsplit) {
  __shared__ double ssum[256];
  const int tid = threadIdx.x;
  const float M = Mptr[0];
  double acc = 0.0;
  for (int i = tid; i < R; i += 256) {
    float li = 0.f, ps = 0.f, cnt = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const float* p = part + ((size_t)s * R + i) * 3;
      li += p[0], ps += p[1], cnt += p[2];
    }
    const float Di = li + 1e-16f;
    acc += (double)(ps / cnt - M - logf(Di));  // cnt == 0 -> NaN like the reference
    row_stats[i * 4 + 0] = Di;
    row_stats[i * 4 + 1] = cnt;
    row_stats[i * 4 + 2] = ps;
    row_stats[i * 4 + 3] = M;
  }
  ssum[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) ssum[tid] += ssum[tid + o];
    __syncthreads();
  }
  if (tid == 0) loss[0] = (float)(-ssum[0] / (double)R);
}

// grid (row blocks, column splits).  dpart[split][R][D] = sum over the split's columns of G_ij P_j
__global__ void __launch_bounds__(256)
    supcon_fused_bwd_kernel(const float* __restrict__ P, const int32_t* __restrict__ labels,
                            const uint8_t* __restrict__ pm, const float* __restrict__ rs,
                            const float* __restrict__ gscale, float* __restrict__ dpart, int n, int D,
                            float inv_t, int nsplit) {
  extern __shared__ float sm[];
  const int R = 2 * n, ldg = SC_TM + 1;
  const int ld2 = sc_ld2(D), pl = sc_plane(D);
  float* sPi = sm;
  float* sPj = sm + sc_block_floats(D);
  float* sG = sPj + sc_block_floats(D);
  float* sRow = sG + SC_TM * ldg;  // [64][2]: 1/D_i, 1/c_i of the row block
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.x * SC_TM;
  const int ncb = (R + SC_TM - 1) / SC_TM;
  const int cb0 = (ncb * (int)blockIdx.y) / nsplit, cb1 = (ncb * ((int)blockIdx.y + 1)) / nsplit;
  const float M = rs[3];
  const float gs = gscale[0] / (float)R;
  sc_stage(P, sPi, m0, R, D, tid);
  if (tid < SC_TM) {
    const int row = m0 + tid;
    sRow[tid * 2 + 0] = row < R ? 1.f / rs[row * 4 + 0] : 0.f;
    sRow[tid * 2 + 1] = row < R ? 1.f / rs[row * 4 + 1] : 0.f;
  }
  const int nct = (D + 31) / 32;          // 32-wide column tiles of dP
  f32x16 acc2[4];                          // this wave's column tiles: wn, wn + 2, wn + 4, wn + 6
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[c][q] = 0.f;
  for (int cb = cb0; cb < cb1; ++cb) {
    const int n0 = cb * SC_TM;
    __syncthreads();  // previous tile: the second product has read sG and sPj
    sc_stage(P, sPj, n0, R, D, tid);
    __syncthreads();
    const f32x16 acc = sc_tile(sPi, sPj, D, wm, wn, i, kk);
    const int col = n0 + wn * 32 + i;
    const ScRow cq = sc_row(labels, n, col);
    const float iDj = col < R ? 1.f / rs[col * 4 + 0] : 0.f;
    const float icj = col < R ? 1.f / rs[col * 4 + 1] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
      const int row = m0 + rr;
      float v = 0.f;
      if (row < R && col < R && row != col) {
        const float e = expf(acc[reg] * inv_t - M);
        const float pos = sc_pos(sc_row(labels, n, row), cq, labels, pm, n) ? 1.f : 0.f;
        v = -gs * (pos * (sRow[rr * 2 + 1] + icj) - e * (sRow[rr * 2 + 0] + iDj));
      }
      sG[rr * ldg + wn * 32 + i] = v;
    }
    __syncthreads();
    // dP rows (wm block) += G[64 rows][64 j] * P_j[64 j][D]
    const float* ga = sG + (wm * 32 + i) * ldg + kk;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ct = wn + 2 * c;
      if (ct < nct) {
        const int dcol = ct * 32 + i;  // P_j[j][dcol] sits at [dcol & 1][j][dcol >> 1]
        const float* pb = sPj + (dcol & 1) * pl + kk * ld2 + (dcol >> 1);
#pragma unroll 8
        for (int js = 0; js < SC_TM; js += 2)
          acc2[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[js], pb[js * ld2], acc2[c], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ct = wn + 2 * c;
    const int dcol = ct * 32 + i;
    if (ct < nct && dcol < D) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
        if (row < R) dpart[((size_t)blockIdx.y * R + row) * D + dcol] = acc2[c][reg];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    supcon_dpart_reduce_kernel(const float* __restrict__ dpart, float* __restrict__ dP, long total, int nsplit,
                               float inv_t) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += dpart[(size_t)k * total + e];
    dP[e] = s * inv_t;
  }
}

inline int sc_nsplit(int R) {
  const int rb = (R + SC_TM - 1) / SC_TM;
  int ns = 256 / rb;
  if (ns > rb) ns = rb;
  if (ns < 1) ns = 1;
  return ns;
}

// ---------------------------------------------------------------- avg pool
template <typename T>
__global__ void __launch_bounds__(256)
    avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ pooled, int HW, int C) {
  // grid: (N, ceil(C/256)); thread = channel, 4 independent partial sums over pixel phases so that
  // four loads are in flight (a single chain is latency-bound: 196 dependent steps at Conv5)
  const int n = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const T* xp = x + (size_t)n * HW * C + c;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 4 <= HW; p += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += to_f32<T>(xp[(size_t)(p + u) * C]);
  }
  for (; p < HW; ++p) s[0] += to_f32<T>(xp[(size_t)p * C]);
  pooled[(size_t)n * C + c] = ((s[0] + s[1]) + (s[2] + s[3])) / (float)HW;
}

template <typename T>
__global__ void __launch_bounds__(256)
    avgpool_bwd_kernel(const float* __restrict__ dpooled, T* __restrict__ dx, int N, int HW,
                       int C) {
  const long total = (long)N * HW * C;
  const float inv = 1.f / (float)HW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % C);
    const int n = (int)(e / ((long)HW * C));
    dx[e] = from_f32<T>(dpooled[(size_t)n * C + c] * inv);
  }
}

// ---------------------------------------------------------------- linear
// one wave per output element y[m][o]
__global__ void __launch_bounds__(256)
    linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                      const float* __restrict__ b, float* __restrict__ y, int M, int I, int O,
                      int act, float slope) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= (long)M * O) return;
  const int m = (int)(e / O), o = (int)(e % O);
  const float* xr = x + (size_t)m * I;
  const float* wr = w + (size_t)o * I;
  float s = 0.f;
  for (int i = lane; i < I; i += 64) s = fmaf(xr[i], wr[i], s);
  s = wave_sum(s);
  if (lane == 0) {
    if (b) s += b[o];
    if (act == 1) s = s > 0.f ? s : s * slope;
    y[e] = s;
  }
}

// dz = dy * act'(y) is formed on the fly
__device__ __forceinline__ float act_grad(float y, float dy, int act, float slope) {
  return act == 1 ? (y > 0.f ? dy : dy * slope) : dy;
}

__global__ void __launch_bounds__(256)
    linear_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ y,
                         const float* __restrict__ dy, float* __restrict__ dx, int M, int I, int O,
                         int act, float slope) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)M * I) return;
  const int m = (int)(e / I), i = (int)(e % I);
  float s[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains keep four weight loads in flight
  int o = 0;
  for (; o + 4 <= O; o += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      s[u] = fmaf(act_grad(y[(size_t)m * O + o + u], dy[(size_t)m * O + o + u], act, slope),
                  w[(size_t)(o + u) * I + i], s[u]);
  }
  for (; o < O; ++o)
    s[0] = fmaf(act_grad(y[(size_t)m * O + o], dy[(size_t)m * O + o], act, slope), w[(size_t)o * I + i],
                s[0]);
  dx[e] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ void __launch_bounds__(256)
    linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                         const float* __restrict__ dy, float* __restrict__ dw,
                         float* __restrict__ db, int M, int I, int O, int act, float slope, int accumulate) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)O * I) return;
  const int o = (int)(e / I), i = (int)(e % I);
  float s = 0.f, sb = 0.f;
  for (int m = 0; m < M; ++m) {
    const float dz = act_grad(y[(size_t)m * O + o], dy[(size_t)m * O + o], act, slope);
    s = fmaf(dz, x[(size_t)m * I + i], s);
    sb += dz;
  }
  if (dw) dw[e] = accumulate ? dw[e] + s : s;
  if (db && i == 0) db[o] = accumulate ? db[o] + sb : sb;
}

// ---------------------------------------------------------------- L2 normalise (wave per row)
__global__ void __launch_bounds__(256)
    l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ z,
                      float* __restrict__ norms, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (size_t)m * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s = fmaf(xr[i], xr[i], s);
  s = wave_sum(s);
  const float nrm = sqrtf(s);
  const float den = fmaxf(nrm, eps);
  for (int i = lane; i < D; i += 64) z[(size_t)m * D + i] = xr[i] / den;
  if (lane == 0) norms[m] = nrm;
}

__global__ void __launch_bounds__(256)
    l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norms,
                      const float* __restrict__ dz, float* __restrict__ dx, int M, int D,
                      float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (size_t)m * D;
  const float* gr = dz + (size_t)m * D;
  const float nrm = norms[m];
  if (nrm > eps) {
    float dot = 0.f;
    for (int i = lane; i < D; i += 64) dot = fmaf(xr[i], gr[i], dot);
    dot = wave_sum(dot);
    const float inv = 1.f / nrm;
    const float k = dot * inv * inv * inv;
    for (int i = lane; i < D; i += 64) dx[(size_t)m * D + i] = gr[i] * inv - xr[i] * k;
  } else {
    const float inv = 1.f / eps;
    for (int i = lane; i < D; i += 64) dx[(size_t)m * D + i] = gr[i] * inv;
  }
}

// ---------------------------------------------------------------- projection head, one launch per direction
// ProjectionHead (contrastyou/projectors/heads.py:12-22,81-96): AdaptiveAvgPool2d(1) -> Linear(C, hid) -> LeakyReLU
// -> Linear(hid, out) -> F.normalize.  The pieces above cost four launches forward and six backward for ~0.4 MFLOP
// per sample; here one workgroup runs a sample through the whole chain (matrix rows are read coalesced, lanes across
// the input index, one wave reduction per output), and the parameter gradients -- sums over the samples -- are a
// second backward kernel with one thread per weight.
constexpr int PH_MAXC = 512, PH_MAXH = 512;  // one workgroup's LDS: 16 partial vectors of <= 512 floats
constexpr int PH_NT = 1024, PH_NW = 16;

// y[o] = sum_i w[o][i] * x[i] (+ b[o]); x in LDS.  A wave takes four rows at a time (eight loads in flight for
// I = 512, four independent reductions): the first version -- four waves, one row per iteration -- spent ~1 us of
// dependent global latency per output (147 us for 512 -> 256 -> 256).
__device__ __forceinline__ void ph_matvec(const float* __restrict__ w, const float* __restrict__ b, const float* xs,
                                          float* ys, int I, int O, int wave, int lane) {
  for (int o0 = wave * 4; o0 < O; o0 += PH_NW * 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane * 4; i < I; i += 256) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + u < O ? o0 + u : O - 1;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)o * I + i);
        s[u] = fmaf(wv[0], xv[0], fmaf(wv[1], xv[1], fmaf(wv[2], xv[2], fmaf(wv[3], xv[3], s[u]))));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = wave_sum(s[u]);
    if (lane < 4 && o0 + lane < O) {
      const float v = lane == 0 ? s[0] : (lane == 1 ? s[1] : (lane == 2 ? s[2] : s[3]));
      ys[o0 + lane] = v + (b ? b[o0 + lane] : 0.f);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(PH_NT)
    proj_head_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ w1, const float* __restrict__ b1,
                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ pooled,
                         float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ z,
                         float* __restrict__ norms, int HW, int C, int hid, int out, float slope, float eps) {
  constexpr int EPC = ElemTr<T>::EPC;
  __shared__ __attribute__((aligned(16))) float spart[PH_NW * PH_MAXC];  // pooling partials [slice][C]
  __shared__ __attribute__((aligned(16))) float sp[PH_MAXC];
  __shared__ __attribute__((aligned(16))) float sh[PH_MAXH];
  __shared__ __attribute__((aligned(16))) float so[PH_MAXH];
  __shared__ float sred[PH_NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  // average pool: thread = (16-byte channel group, position slice); every load of a thread is independent
  const int G = C / EPC;
  const int nsl = PH_NT / G < PH_NW ? PH_NT / G : PH_NW;  // slices (host: G <= 1024)
  {
    const int cg = tid % G, sl = tid / G;
    if (sl < nsl) {
      float a[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[e] = 0.f;
      const T* xp = feat + (size_t)n * HW * C + cg * EPC;
      for (int p = sl; p < HW; p += nsl) {
        float f[EPC];
        Chunk<T>::unpack(ld16(xp + (size_t)p * C), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a[e] += f[e];
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) spart[sl * C + cg * EPC + e] = a[e];
    }
  }
  __syncthreads();
  const float inv = 1.f / (float)HW;
  for (int c = tid; c < C; c += PH_NT) {
    float v = 0.f;
    for (int q = 0; q < nsl; ++q) v += spart[q * C + c];
    v *= inv;
    sp[c] = v;
    pooled[(size_t)n * C + c] = v;
  }
  __syncthreads();
  ph_matvec(w1, b1, sp, sh, C, hid, wave, lane);
  __syncthreads();
  for (int j = tid; j < hid; j += PH_NT) {
    const float v = sh[j] > 0.f ? sh[j] : slope * sh[j];
    sh[j] = v;
    y1[(size_t)n * hid + j] = v;
  }
  __syncthreads();
  ph_matvec(w2, b2, sh, so, hid, out, wave, lane);
  __syncthreads();
  float ss = 0.f;
  for (int o = tid; o < out; o += PH_NT) ss = fmaf(so[o], so[o], ss);
  ss = wave_sum(ss);
  if (lane == 0) sred[wave] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < PH_NW; ++q) tot += sred[q];
  const float nrm = sqrtf(tot);
  const float den = fmaxf(nrm, eps);
  for (int o = tid; o < out; o += PH_NT) {
    y2[(size_t)n * out + o] = so[o];
    z[(size_t)n * out + o] = so[o] / den;
  }
  if (tid == 0) norms[n] = nrm;
}

// per sample: dy2 (normalise backward), dh = (W2^T dy2) * lrelu'(y1), dpooled = W1^T dh, dfeat = dpooled / HW
template <typename T>
__global__ void __launch_bounds__(PH_NT)
    proj_head_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y1, const float* __restrict__ y2,
                         const float* __restrict__ norms, const float* __restrict__ w1,
                         const float* __restrict__ w2, float* __restrict__ dy2g, float* __restrict__ dhg,
                         T* __restrict__ dfeat, int HW, int C, int hid, int out, float slope, float eps) {
  __shared__ __attribute__((aligned(16))) float sa[PH_MAXH];           // dy2, then dh
  __shared__ __attribute__((aligned(16))) float sb[PH_NW * PH_MAXC];   // per-wave partial columns
  __shared__ float sred[PH_NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  // dy2 = dz / nrm - y2 * (y2 . dz) / nrm^3   (nrm <= eps: dz / eps), as l2norm_bwd_kernel
  const float nrm = norms[n];
  float dot = 0.f;
  for (int o = tid; o < out; o += PH_NT) dot = fmaf(y2[(size_t)n * out + o], dz[(size_t)n * out + o], dot);
  dot = wave_sum(dot);
  if (lane == 0) sred[wave] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int q = 0; q < PH_NW; ++q) dot += sred[q];
  for (int o = tid; o < out; o += PH_NT) {
    float v;
    if (nrm > eps) {
      const float iv = 1.f / nrm;
      v = dz[(size_t)n * out + o] * iv - y2[(size_t)n * out + o] * (dot * iv * iv * iv);
    } else {
      v = dz[(size_t)n * out + o] / eps;
    }
    sa[o] = v;
    dy2g[(size_t)n * out + o] = v;
  }
  __syncthreads();
  // column sums: wave w takes rows w, w + 16, ... of the matrix (loads independent of each other), lanes across
  // the columns; the sixteen partial vectors meet in LDS
  auto tmatvec = [&](const float* __restrict__ w, const float* xs, int R, int Ccols) {
    for (int c0 = lane * 4; c0 < Ccols; c0 += 256) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int rr = wave; rr < R; rr += PH_NW) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)rr * Ccols + c0);
        const float xv = xs[rr];
        acc[0] = fmaf(wv[0], xv, acc[0]), acc[1] = fmaf(wv[1], xv, acc[1]);
        acc[2] = fmaf(wv[2], xv, acc[2]), acc[3] = fmaf(wv[3], xv, acc[3]);
      }
      *reinterpret_cast<f32x4*>(&sb[wave * Ccols + c0]) = acc;
    }
  };
  tmatvec(w2, sa, out, hid);
  __syncthreads();
  for (int j = tid; j < hid; j += PH_NT) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < PH_NW; ++q) g += sb[q * hid + j];
    const float v = y1[(size_t)n * hid + j] > 0.f ? g : g * slope;
    sa[j] = v;
    dhg[(size_t)n * hid + j] = v;
  }
  __syncthreads();
  tmatvec(w1, sa, hid, C);
  __syncthreads();
  const float inv = 1.f / (float)HW;
  float* sdp = sa;  // (dh is in global memory by now)  dpooled / HW, C <= PH_MAXH
  __syncthreads();
  for (int c = tid; c < C; c += PH_NT) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < PH_NW; ++q) g += sb[q * C + c];
    sdp[c] = g * inv;
  }
  __syncthreads();
  if (dfeat) {
    constexpr int EPC = ElemTr<T>::EPC;
    T* dp = dfeat + (size_t)n * HW * C;
    const int G = C / EPC;
    for (long e = tid; e < (long)HW * G; e += PH_NT) {
      const int cg = (int)(e % G);
      st16(dp + (e / G) * C + cg * EPC, Chunk<T>::pack(sdp + cg * EPC));
    }
  }
}

// parameter gradients: thread e < out*hid: dW2[o][j] = sum_n dy2[n][o] y1[n][j]; then dW1[j][i] = sum_n dh[n][j] pooled[n][i]
__global__ void __launch_bounds__(256)
    proj_head_dw_kernel(const float* __restrict__ dy2, const float* __restrict__ dh, const float* __restrict__ y1,
                        const float* __restrict__ pooled, float* __restrict__ dw1, float* __restrict__ db1,
                        float* __restrict__ dw2, float* __restrict__ db2, int B, int C, int hid, int out,
                        int accumulate) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  const long n2 = (long)out * hid, n1 = (long)hid * C;
  const float *ga, *xa;
  float *dw, *db;
  int O, I;
  long k;
  if (e < n2) {
    ga = dy2, xa = y1, dw = dw2, db = db2, O = out, I = hid, k = e;
  } else if (e < n2 + n1) {
    ga = dh, xa = pooled, dw = dw1, db = db1, O = hid, I = C, k = e - n2;
  } else {
    return;
  }
  const int o = (int)(k / I), i = (int)(k % I);
  float s = 0.f, sbias = 0.f;
  for (int n = 0; n < B; ++n) {
    const float g = ga[(size_t)n * O + o];
    s = fmaf(g, xa[(size_t)n * I + i], s);
    sbias += g;
  }
  if (dw) dw[k] = accumulate ? dw[k] + s : s;
  if (db && i == 0) db[o] = accumulate ? db[o] + sbias : sbias;
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int cy_sgemm(const float* A, const float* B, float* C, int M, int N, int K, float alpha,
             int b_trans, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return CY_ERR_ARG;
  return launch_sgemm(A, B, C, M, N, K, alpha, b_trans, (hipStream_t)stream);
}

int cy_supcon_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* S,
                  float* loss, float* row_stats, int n, int D, float t, void* stream) {
  if (!P || (!labels && !pos_mask) || !S || !loss || !row_stats) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || !(t > 0.f)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n;
  int rc = launch_sgemm(P, P, S, R, R, D, 1.f / t, 1, st);
  if (rc != CY_OK) return rc;
  hipLaunchKernelGGL(supcon_rowstats_kernel, dim3(cy_cdiv(R, 4)), dim3(256), 0, st, S, labels,
                     pos_mask, row_stats, n);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(supcon_finalize_kernel, dim3(1), dim3(256), 0, st, row_stats, loss, R);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_supcon_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* S,
                  const float* row_stats, const float* gscale, float* G, float* dP, int n, int D,
                  float t, void* stream) {
  if (!P || (!labels && !pos_mask) || !S || !row_stats || !gscale || !G || !dP) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || !(t > 0.f)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n;
  hipLaunchKernelGGL(supcon_gsym_kernel, dim3(grid_for((long)R * R)), dim3(256), 0, st, S,
                     row_stats, labels, pos_mask, gscale, G, n);
  CY_CHECK_LAUNCH();
  return launch_sgemm(G, P, dP, R, D, R, 1.f / t, 0, st);
}

int cy_supcon_excl_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* S, float* loss,
                       float* row_stats, float* tmp, int n, int D, float t, void* stream) {
  if (!P || (!labels && !pos_mask) || !S || !loss || !row_stats || !tmp) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || !(t > 0.f)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n;
  int rc = launch_sgemm(P, P, S, R, R, D, 1.f / t, 1, st);
  if (rc != CY_OK) return rc;
  hipLaunchKernelGGL(supcon_rowstats_kernel, dim3(cy_cdiv(R, 4)), dim3(256), 0, st, S, labels, pos_mask, tmp, n);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(supcon_excl_rows_kernel, dim3(cy_cdiv(R, 4)), dim3(256), 0, st, S, labels, pos_mask, (const float*)tmp,
                     row_stats, tmp + (size_t)4 * R, n);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(supcon_excl_loss_kernel, dim3(1), dim3(256), 0, st, row_stats, loss, (const float*)(tmp + (size_t)4 * R), R);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_supcon_excl_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* S,
                       const float* row_stats, const float* tmp, const float* gscale, float* G, float* dP, int n, int D,
                       float t, void* stream) {
  if (!P || (!labels && !pos_mask) || !S || !row_stats || !tmp || !gscale || !G || !dP) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || !(t > 0.f)) return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n;
  hipLaunchKernelGGL(supcon_excl_gsym_kernel, dim3(grid_for((long)R * R)), dim3(256), 0, st, S, row_stats,
                     tmp + (size_t)4 * R, labels, pos_mask, gscale, G, n);
  CY_CHECK_LAUNCH();
  return launch_sgemm(G, P, dP, R, D, R, 1.f / t, 0, st);
}

size_t cy_supcon_fused_ws_bytes(int n, int D) {
  if (n <= 0 || D <= 0) return 0;
  const size_t R = 2 * (size_t)n;
  const size_t ns = (size_t)sc_nsplit((int)R);
  // diag[R] + M[4] + forward partials [ns][R][3] | backward partials [ns][R][D]
  const size_t fwd = ns * R * 3, bwd = ns * R * (size_t)D;
  return (R + 4 + (fwd > bwd ? fwd : bwd)) * sizeof(float);
}

int cy_supcon_fused_fwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, float* loss,
                        float* row_stats, float* diag_out, void* ws, size_t ws_bytes, int n, int D, float t,
                        void* stream) {
  if (!P || (!labels && !pos_mask) || !loss || !row_stats || !ws) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || D > 256 || D % 8 || !(t > 0.f)) return CY_ERR_SHAPE;
  if (ws_bytes < cy_supcon_fused_ws_bytes(n, D)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n, ns = sc_nsplit(R), rb = cy_cdiv(R, SC_TM);
  float* diag = diag_out ? diag_out : (float*)ws;  // (written in place: no device-to-device copy afterwards)
  float* Mv = (float*)ws + R;
  float* part = Mv + 4;
  hipLaunchKernelGGL(supcon_diag_kernel, dim3(cy_cdiv(R, 4)), dim3(256), 0, st, P, diag, R, D, 1.f / t);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(supcon_max_kernel, dim3(1), dim3(256), 0, st, diag, Mv, R);
  CY_CHECK_LAUNCH();
  const size_t smem = (size_t)2 * sc_block_floats(D) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(supcon_fused_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sc_block_floats(256) * 4) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(supcon_fused_fwd_kernel, dim3(rb, ns), dim3(256), smem, st, P, labels, pos_mask, Mv, part,
                     n, D, 1.f / t, ns);
  CY_CHECK_LAUNCH();
  hipLaunchKernelGGL(supcon_fused_finalize_kernel, dim3(1), dim3(256), 0, st, part, Mv, row_stats, loss, R, ns);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_supcon_fused_bwd(const float* P, const int32_t* labels, const uint8_t* pos_mask, const float* row_stats,
                        const float* gscale, float* dP, void* ws, size_t ws_bytes, int n, int D, float t,
                        void* stream) {
  if (!P || (!labels && !pos_mask) || !row_stats || !gscale || !dP || !ws) return CY_ERR_ARG;
  if (n <= 0 || D <= 0 || D > 256 || D % 8 || !(t > 0.f)) return CY_ERR_SHAPE;
  if (ws_bytes < cy_supcon_fused_ws_bytes(n, D)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = 2 * n, ns = sc_nsplit(R), rb = cy_cdiv(R, SC_TM);
  float* dpart = (float*)ws + R + 4;
  const size_t smem = ((size_t)2 * sc_block_floats(D) + SC_TM * (SC_TM + 1) + SC_TM * 2) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(supcon_fused_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (2 * sc_block_floats(256) + SC_TM * (SC_TM + 1) + SC_TM * 2) * 4) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(supcon_fused_bwd_kernel, dim3(rb, ns), dim3(256), smem, st, P, labels, pos_mask, row_stats,
                     gscale, dpart, n, D, 1.f / t, ns);
  CY_CHECK_LAUNCH();
  const long total = (long)R * D;
  hipLaunchKernelGGL(supcon_dpart_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, st, dpart, dP, total, ns,
                     1.f / t);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_supcon_matrices(const float* S, const float* row_stats, const int32_t* labels,
                       const uint8_t* pos_mask, float* sim_logits, float* sim_exp,
                       float* pos_out, float* neg_out, int n, void* stream) {
  if (!S || !row_stats || (!labels && !pos_mask) || n <= 0) return CY_ERR_ARG;
  const int R = 2 * n;
  hipLaunchKernelGGL(supcon_matrices_kernel, dim3(grid_for((long)R * R)), dim3(256), 0,
                     (hipStream_t)stream, S, row_stats, labels, pos_mask, sim_logits, sim_exp,
                     pos_out, neg_out, n);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_avgpool_fwd(const void* x, float* pooled, int N, int HW, int C, int dtype, void* stream) {
  if (!x || !pooled || N <= 0 || HW <= 0 || C <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(N, cy_cdiv(C, 256));
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, pooled,
                       HW, C);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, pooled,
                       HW, C);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, pooled,
                       HW, C);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_avgpool_bwd(const float* dpooled, void* dx, int N, int HW, int C, int dtype, void* stream) {
  if (!dpooled || !dx || N <= 0 || HW <= 0 || C <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long)N * HW * C);
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, dpooled, (bf16*)dx,
                       N, HW, C);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<f16>, dim3(grid), dim3(256), 0, st, dpooled, (f16*)dx,
                       N, HW, C);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, dpooled,
                       (float*)dx, N, HW, C);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int I, int O,
                  int act, float slope, void* stream) {
  if (!x || !w || !y || M <= 0 || I <= 0 || O <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cy_cdiv((long)M * O, 4)), dim3(256), 0,
                     (hipStream_t)stream, x, w, b, y, M, I, O, act, slope);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int linear_bwd_impl(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                           float* db, int M, int I, int O, int act, float slope, int accumulate, void* stream) {
  if (!x || !w || !y || !dy || M <= 0 || I <= 0 || O <= 0) return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cy_cdiv((long)M * I, 256)), dim3(256), 0, st, w,
                       y, dy, dx, M, I, O, act, slope);
    CY_CHECK_LAUNCH();
  }
  if (dw || db) {
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cy_cdiv((long)O * I, 256)), dim3(256), 0, st, x,
                       y, dy, dw, db, M, I, O, act, slope, accumulate);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                  float* dw, float* db, int M, int I, int O, int act, float slope, void* stream) {
  return linear_bwd_impl(x, w, y, dy, dx, dw, db, M, I, O, act, slope, 0, stream);
}

int cy_linear_bwd_into(const float* x, const float* w, const float* y, const float* dy, float* dx,
                       float* dw, float* db, int M, int I, int O, int act, float slope, void* stream) {
  return linear_bwd_impl(x, w, y, dy, dx, dw, db, M, I, O, act, slope, 1, stream);
}

int cy_proj_head_fwd(const void* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                     float* pooled, float* y1, float* y2, float* z, float* norms, int B, int HW, int C, int hid,
                     int out, float slope, float eps, int dtype, void* stream) {
  if (!feat || !w1 || !w2 || !pooled || !y1 || !y2 || !z || !norms || B <= 0 || HW <= 0) return CY_ERR_ARG;
  if (C <= 0 || C > PH_MAXC || C % 8 || hid <= 0 || hid > PH_MAXH || hid % 4 || out <= 0 || out > PH_MAXH)
    return CY_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(proj_head_fwd_kernel<bf16>, dim3(B), dim3(PH_NT), 0, st, (const bf16*)feat, w1, b1, w2, b2,
                       pooled, y1, y2, z, norms, HW, C, hid, out, slope, eps);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(proj_head_fwd_kernel<f16>, dim3(B), dim3(PH_NT), 0, st, (const f16*)feat, w1, b1, w2, b2,
                       pooled, y1, y2, z, norms, HW, C, hid, out, slope, eps);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(proj_head_fwd_kernel<float>, dim3(B), dim3(PH_NT), 0, st, (const float*)feat, w1, b1, w2, b2,
                       pooled, y1, y2, z, norms, HW, C, hid, out, slope, eps);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

size_t cy_proj_head_bwd_ws_bytes(int B, int hid, int out) { return (size_t)B * (hid + out) * sizeof(float); }

int cy_proj_head_bwd(const float* dz, const float* pooled, const float* y1, const float* y2, const float* norms,
                     const float* w1, const float* w2, void* dfeat, float* dw1, float* db1, float* dw2, float* db2,
                     int accumulate, void* ws, size_t ws_bytes, int B, int HW, int C, int hid, int out, float slope,
                     float eps, int dtype, void* stream) {
  if (!dz || !pooled || !y1 || !y2 || !norms || !w1 || !w2 || !ws || B <= 0 || HW <= 0) return CY_ERR_ARG;
  if (C <= 0 || C > PH_MAXC || C % 8 || hid <= 0 || hid > PH_MAXH || hid % 4 || out <= 0 || out > PH_MAXH)
    return CY_ERR_SHAPE;
  if (ws_bytes < cy_proj_head_bwd_ws_bytes(B, hid, out)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* dy2 = (float*)ws;
  float* dh = dy2 + (size_t)B * out;
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(proj_head_bwd_kernel<bf16>, dim3(B), dim3(PH_NT), 0, st, dz, y1, y2, norms, w1, w2, dy2, dh,
                       (bf16*)dfeat, HW, C, hid, out, slope, eps);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(proj_head_bwd_kernel<f16>, dim3(B), dim3(PH_NT), 0, st, dz, y1, y2, norms, w1, w2, dy2, dh,
                       (f16*)dfeat, HW, C, hid, out, slope, eps);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(proj_head_bwd_kernel<float>, dim3(B), dim3(PH_NT), 0, st, dz, y1, y2, norms, w1, w2, dy2, dh,
                       (float*)dfeat, HW, C, hid, out, slope, eps);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  if (dw1 || db1 || dw2 || db2) {
    const long total = (long)out * hid + (long)hid * C;
    hipLaunchKernelGGL(proj_head_dw_kernel, dim3(cy_cdiv(total, 256)), dim3(256), 0, st, (const float*)dy2,
                       (const float*)dh, y1, pooled, dw1, db1, dw2, db2, B, C, hid, out, accumulate);
    CY_CHECK_LAUNCH();
  }
  return CY_OK;
}

int cy_l2norm_fwd(const float* x, float* z, float* norms, int M, int D, float eps, void* stream) {
  if (!x || !z || !norms || M <= 0 || D <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cy_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     z, norms, M, D, eps);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_l2norm_bwd(const float* x, const float* norms, const float* dz, float* dx, int M, int D,
                  float eps, void* stream) {
  if (!x || !norms || !dz || !dx || M <= 0 || D <= 0) return CY_ERR_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cy_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     norms, dz, dx, M, D, eps);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
