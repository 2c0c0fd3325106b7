// BatchNorm sums across workgroups WITHOUT a finalize launch (reference: nn.BatchNorm2d in training mode,
// contrastyou/arch/unet.py:22,25,40).
//
// The per-channel sums a training-mode BatchNorm needs (forward: sum y, sum y^2; backward: sum dz, sum dz*xhat) are
// produced by hundreds of workgroups.  Rounds 1-3 wrote one f32 partial row per workgroup and reduced the rows with a
// C/4-workgroup "finalize" launch per BatchNorm layer and direction: 76 launches x 5.4 us + a dependent-launch gap each,
// all of it on the step's critical path (conv a -> finalize -> conv b).  Here the producers ADD their partials into a
// small accumulator and the CONSUMER (the next conv's BN+ReLU prologue, the apply kernels, the backward apply kernel)
// derives the coefficients itself; its first workgroup also leaves them in memory for the backward pass.
//
// The accumulation has to be independent of the order in which workgroups arrive (bitwise reproducible steps are a
// tested property of this build), so it is done in INTEGERS: a partial s is split exactly into
//     s = hi * 2^-12 + lo * 2^-44,     hi = rint(s * 2^12),  lo = rint((s - hi * 2^-12) * 2^44)
// (exact for |s| >= 2^-21, otherwise rounded at 2^-44 absolute; |hi| < 2^51 for |s| < 5e11) and both limbs are added with
// 64-bit integer atomics, which commute.  One fixed scale would have to trade range against resolution; two limbs give
// 2^-44 resolution over +-2^50 whatever the layer's magnitude.
//
// Contention: a 64-byte line takes one atomic request per ~25 ns (MI355X_MICROARCH.md, global float atomics: 0.09 TB/s into
// one 2304-byte row), so the accumulator has R replicas [R][4][C] and workgroup b adds into replica b & (R - 1); the host
// picks R ~ workgroups / 32 (a tail of < 1 us), bounded by what a consumer workgroup is asked to read (R * C <= 2048:
// 64 KB from L2).  Behind the replicas sit R flag words: a partial that is not finite (or beyond the hi limb's range)
// sets its replica's flag instead of being added, and a consumer that finds a flag set reads every sum as NaN -- a
// diverged step stays visibly diverged (GradScaler's inf check, the reference's NaN behaviour).
#pragma once
#include "cy_common.h"

#include <math.h>

struct BnFold {  // consumer side, forward statistics -> relu(scale * y + shift)
  const unsigned long long* acc;  // [R][4][C]: sum hi, sum lo, sum of squares hi, lo; null: no fold (coefficients given)
  int R, C;
  const float* gamma;  // null: 1
  const float* beta;   // null: 0
  double inv_count;    // 1 / (N H W)
  double unbias;       // count / (count - 1): running_var takes the unbiased estimate (torch)
  float eps;
  float* coef;         // [5][C]: scale, shift, mean, invstd, unbiased variance -- written by the leader workgroup; may be null
};

struct BnBwdFold {  // consumer side, backward sums -> dy = scale * dz + k1 * y + k0
  const unsigned long long* acc;  // [R][4][C]: sum dz hi, lo, sum dz * xhat hi, lo
  int R, C;
  const float* coef;  // the forward pass's [5][C]
  double inv_count;
  int batch_stats;    // 0: the BatchNorm used running statistics (k1 = k0 = 0)
  int accumulate;     // leader: dgamma / dbeta += (1) or = (0)
  float* dgamma;      // may be null
  float* dbeta;
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void bn_acc_add(unsigned long long* acc, int R, int C, int rep, int q, int c, float s) {
  const double d = (double)s;
  const double h = rint(d * 4096.0);
  long long hi = (long long)h;
  long long lo = (long long)rint((d - h * (1.0 / 4096.0)) * 17592186044416.0);
  if (!(fabsf(s) < 2.0e11f)) {  // inf / nan / out of range: flag the replica (the consumer turns the sums into NaN)
    atomicOr(acc + (size_t)R * 4 * C + rep, 1ull);
    return;
  }
  unsigned long long* p = acc + ((size_t)(rep * 4 + 2 * q) * C + c);
  atomicAdd(p, (unsigned long long)hi);  // (agent scope, no return value: executed at the memory side)
  atomicAdd(p + C, (unsigned long long)lo);
}

__device__ __forceinline__ double bn_acc_read(const unsigned long long* acc, int R, int C, int q, int c) {
  long long hi = 0, lo = 0;
  const unsigned long long* p = acc + (size_t)(2 * q) * C + c;
  const size_t rs = (size_t)4 * C;
  int r = 0;
  for (; r + 4 <= R; r += 4) {  // (four replicas' loads in flight)
    const unsigned long long h0 = p[(r + 0) * rs], h1 = p[(r + 1) * rs], h2 = p[(r + 2) * rs], h3 = p[(r + 3) * rs];
    const unsigned long long l0 = p[(r + 0) * rs + C], l1 = p[(r + 1) * rs + C], l2 = p[(r + 2) * rs + C], l3 = p[(r + 3) * rs + C];
    hi += (long long)(h0 + h1 + h2 + h3);
    lo += (long long)(l0 + l1 + l2 + l3);
  }
  for (; r < R; ++r) {
    hi += (long long)p[r * rs];
    lo += (long long)p[r * rs + C];
  }
  unsigned long long bad = 0;
  for (r = 0; r < R; ++r) bad |= acc[(size_t)R * 4 * C + r];
  if (bad) return __builtin_nan("");
  return (double)hi * (1.0 / 4096.0) + (double)lo * (1.0 / 17592186044416.0);
}

// The replicas of an accumulator summed into LDS by ALL threads of a workgroup: s_sum[4 * C + 1] (64-bit words: the four
// limb rows, then the flag).  One thread per channel walking the replicas (the first version) is a chain of dependent
// round trips -- R / 4 of them, 4-8 us on the 32-channel layers with R = 32; here every thread has its R * 4 * C / nthr
// loads in flight at once and adds them with LDS atomics (integer adds: any order gives the same words).
template <int RR>
__device__ __forceinline__ void bn_acc_gather_direct(const unsigned long long* acc, int C, unsigned long long* s_sum,
                                                     int tid, int nthr) {
  // few replicas: a thread sums its word's RR replicas itself, all loads in flight at once -- no LDS atomics, one barrier
  const int row = 4 * C;
  for (int i = tid; i < row; i += nthr) {
    unsigned long long v[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) v[r] = acc[(size_t)r * row + i];
    unsigned long long t = v[0];
#pragma unroll
    for (int r = 1; r < RR; ++r) t += v[r];
    s_sum[i] = t;
  }
  if (tid == 0) {
    unsigned long long bad = 0;
    for (int r = 0; r < RR; ++r) bad |= acc[(size_t)RR * row + r];
    s_sum[row] = bad;
  }
  __syncthreads();
}

// WIDE (the elementwise kernels, which have the registers): 16 and 32 replicas the same way -- the LDS-atomic form
// below costs a workgroup R * 4 * C 64-bit LDS atomics (8192 at R = 32), 5-13 us of every such launch on cold data
// (tools/bench_ew_cold.py) with four workgroups per CU queueing at one LDS
template <bool WIDE = false>
__device__ __forceinline__ void bn_acc_gather(const unsigned long long* acc, int R, int C, unsigned long long* s_sum,
                                              int tid, int nthr) {
  if (R == 1) return bn_acc_gather_direct<1>(acc, C, s_sum, tid, nthr);
  if (R == 2) return bn_acc_gather_direct<2>(acc, C, s_sum, tid, nthr);
  if (R == 4) return bn_acc_gather_direct<4>(acc, C, s_sum, tid, nthr);
  if (R == 8) return bn_acc_gather_direct<8>(acc, C, s_sum, tid, nthr);
  if constexpr (WIDE) {
    if (R == 16) return bn_acc_gather_direct<16>(acc, C, s_sum, tid, nthr);
    if (R == 32) return bn_acc_gather_direct<32>(acc, C, s_sum, tid, nthr);
  }
  const int row = 4 * C;
  for (int i = tid; i <= row; i += nthr) s_sum[i] = 0ull;
  __syncthreads();
  const int n = R * row;
#pragma unroll 4
  for (int i = tid; i < n; i += nthr) {
    const unsigned long long v = acc[i];
    atomicAdd(&s_sum[i % row], v);
  }
  if (tid < R && acc[n + tid]) atomicOr(&s_sum[row], 1ull);
  __syncthreads();
}

// 1 / sqrt(x) in f64 without the compiler's division + square-root sequences (a microsecond per channel in a kernel
// prologue that is nothing but latency): the hardware estimate and two Newton steps (full f64 accuracy for x > 0)
__device__ __forceinline__ double bn_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double bn_sum_read(const unsigned long long* s_sum, int C, int q, int c) {
  if (s_sum[4 * C]) return __builtin_nan("");
  return (double)(long long)s_sum[(2 * q) * C + c] * (1.0 / 4096.0) +
         (double)(long long)s_sum[(2 * q + 1) * C + c] * (1.0 / 17592186044416.0);
}

// coefficients of channel c from the two sums; the leader also leaves them (and the batch moments) in f.coef.  Same
// arithmetic as the finalize kernel of rounds 1-3 (f64 throughout, variance clamped at 0).
__device__ __forceinline__ void bn_fold_finish(const BnFold& f, int c, bool leader, double s1, double s2, float& sc, float& sh);
__device__ __forceinline__ void bn_fold_channel(const BnFold& f, int c, bool leader, float& sc, float& sh) {
  bn_fold_finish(f, c, leader, bn_acc_read(f.acc, f.R, f.C, 0, c), bn_acc_read(f.acc, f.R, f.C, 1, c), sc, sh);
}
// (after bn_acc_gather(f.acc, f.R, f.C, s_sum, ...))
__device__ __forceinline__ void bn_fold_channel_lds(const BnFold& f, const unsigned long long* s_sum, int c, bool leader,
                                                    float& sc, float& sh) {
  bn_fold_finish(f, c, leader, bn_sum_read(s_sum, f.C, 0, c), bn_sum_read(s_sum, f.C, 1, c), sc, sh);
}
// (g, b: gamma[c] and beta[c], which a caller with registers to spare requests BEFORE the gather -- read here they are one
//  more dependent round trip to memory in a prologue that is nothing but such round trips)
__device__ __forceinline__ void bn_fold_finish_gb(const BnFold& f, int c, bool leader, double s1, double s2, double g, double b,
                                                  float& sc, float& sh);
__device__ __forceinline__ void bn_fold_finish(const BnFold& f, int c, bool leader, double s1, double s2, float& sc, float& sh) {
  bn_fold_finish_gb(f, c, leader, s1, s2, f.gamma ? (double)f.gamma[c] : 1.0, f.beta ? (double)f.beta[c] : 0.0, sc, sh);
}
__device__ __forceinline__ void bn_fold_finish_gb(const BnFold& f, int c, bool leader, double s1, double s2, double g, double b,
                                                  float& sc, float& sh) {
  const double mean = s1 * f.inv_count;
  double var = s2 * f.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double istd = bn_rsqrt(var + (double)f.eps);
  sc = (float)(g * istd);
  sh = (float)(b - mean * g * istd);
  if (leader && f.coef) {
    f.coef[c] = sc;
    f.coef[f.C + c] = sh;
    f.coef[2 * f.C + c] = (float)mean;
    f.coef[3 * f.C + c] = (float)istd;
    f.coef[4 * f.C + c] = (float)(var * f.unbias);
  }
}

// backward coefficients of channel c: dy = scale * dz + k1 * y + k0; the leader adds the parameter gradients
// (after bn_acc_gather(f.acc, f.R, f.C, s_sum, ...))
// (sc, mu, is: rows 0, 2, 3 of the forward coefficients at c; dg_old, db_old: the leader's current gradient words when it
//  accumulates -- all requested by the caller before the gather, see bn_fold_finish_gb)
__device__ __forceinline__ void bn_bwd_fold_finish(const BnBwdFold& f, const unsigned long long* s_sum, int c, bool leader,
                                                   float sc_, float mu_, float is_, float dg_old, float db_old, float& k1o,
                                                   float& k0o) {
  const double t1 = bn_sum_read(s_sum, f.C, 0, c), t2 = bn_sum_read(s_sum, f.C, 1, c);
  double k1 = 0.0, k0 = 0.0;
  if (f.batch_stats) {
    const double sc = (double)sc_, mu = (double)mu_, is = (double)is_;
    k1 = -sc * is * t2 * f.inv_count;
    k0 = -sc * t1 * f.inv_count - k1 * mu;
  }
  k1o = (float)k1;
  k0o = (float)k0;
  if (leader) {
    if (f.dbeta) f.dbeta[c] = f.accumulate ? db_old + (float)t1 : (float)t1;
    if (f.dgamma) f.dgamma[c] = f.accumulate ? dg_old + (float)t2 : (float)t2;
  }
}
__device__ __forceinline__ void bn_bwd_fold_channel(const BnBwdFold& f, const unsigned long long* s_sum, int c, bool leader,
                                                    float& k1o, float& k0o) {
  float sc = 0.f, mu = 0.f, is = 0.f, dg = 0.f, db = 0.f;
  if (f.batch_stats) sc = f.coef[c], mu = f.coef[2 * f.C + c], is = f.coef[3 * f.C + c];
  if (leader && f.accumulate) {
    if (f.dbeta) db = f.dbeta[c];
    if (f.dgamma) dg = f.dgamma[c];
  }
  bn_bwd_fold_finish(f, s_sum, c, leader, sc, mu, is, dg, db, k1o, k0o);
}
#endif

static inline BnFold bn_fold_from_abi(const cy_bn_fold* f) {
  BnFold b;
  b.acc = (const unsigned long long*)f->acc;
  b.R = f->R, b.C = f->C;
  b.gamma = f->gamma, b.beta = f->beta;
  b.inv_count = 1.0 / f->count;
  b.unbias = f->count > 1.0 ? f->count / (f->count - 1.0) : 1.0;
  b.eps = f->eps;
  b.coef = f->coef;
  return b;
}

static inline size_t bn_acc_elems(int C, int R) { return (size_t)R * 4 * C + R; }  // 64-bit words

// replicas for an accumulator of C channels fed by `workgroups` producers (a power of two, 1..32)
static inline int bn_acc_replicas(int C, int workgroups) {
  int r = 1;
  while (r < 32 && r * 32 < workgroups && 2 * r * C <= 2048) r *= 2;
  return r;
}
