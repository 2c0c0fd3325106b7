// Matrix-core forms of the dense projector (cy_dense.hip) for 16-bit feature maps with C = 32, 64 or 128 channels and
// 128 or 256 hidden units -- the geometry of the dense InfoNCE hook (Up_conv2 features at max_channel 512,
// config/hooks/infonce_dense.yaml; Up_conv3 / Up_conv4 = 64 / 128 channels, the former in the pre-training scripts;
// contrastyou/projectors/heads.py:31-41,99-123).  The kernels were written for C = 32; with CB = C / 32 channel
// blocks the pre-activations accumulate over the blocks (rows and the W1 image carry a block index) and the backward
// kernels' second product (dW1, dx) is done for ONE selected block per launch (`cbs`): CB launches, each recomputing
// the cheap pre-activations, instead of the VALU kernels' 9-15 ms.  (C = 256 would need 192 registers of rows in
// flight: it stays on the VALU kernels.)
//
// No workgroup-level cooperation: a WAVE owns a job (a bin, or a cell of the bin partition) and walks its
// pixels 32 at a time.  The 1x1 convolution of a block of 32 pixels is 2 x (hid/32) MFMAs whose operands need no
// LDS at all: the A fragment of pixel-row r is 16 contiguous bytes of the NHWC map (one global load per lane), the
// B fragments (W1 rounded to the storage type) live in registers for the life of the wave (forward) or in an LDS image
// (backward kernels, which need the registers for their second product).  D[pixel][hidden]
// leaves lane (o, half) with 16 pixels of hidden unit o, so the leaky-ReLU + pixel sum (forward) and the
// multiplication with the pooled gradient (backward) are per-lane VALU work -- which is what bounds these
// kernels (3-4 VALU operations per (pixel, hidden) pair against 1/16 MFMA).
//
// Rows of a partial last block read as x = 0: their pre-activation is exactly b1, so their contribution to
// the pixel sum / to db1 is a per-lane constant that is subtracted once per job; dW1 sees x = 0 and dx is not
// stored for them.  No per-element masking anywhere.
//
// Backward, all bins: adjacent adaptive-pooling bins share at most one pixel row / column, so the (2s-1)^2
// CELLS cut out by the bin boundaries partition the image and every pixel of a cell gets its gradient from
// the same <= 4 bins: g[p][o] = lrelu'(pre[p][o]) * D_cell[o], D_cell = sum of dhpool[bin]/|bin|.  One pass
// over the pixels, dx written exactly once (no read-modify-write, no colour classes), bit-reproducible.
// With a bin list the jobs are the listed bins and dx is accumulated one colour class per launch as in the
// VALU kernel.
//   dW1^T[c][o] = sum_p x[p][c] g[p][o]: g is the B operand STRAIGHT FROM THE ACCUMULATOR REGISTERS (the k index
//     of an MFMA may be any permutation as long as A and B agree: lane-half h, slot i <-> pixel row
//     16s + 4h + i (+4 for i >= 4)); A = x^T comes from a per-wave LDS copy of the 32 x 32 pixel block through
//     ds_read_b64_tr_b16, whose four row addresses per 16-lane group are free to follow that permutation.
//   dx^T[c][p] = sum_o W1[o][c] g[p][o]: contraction over the lane index of g -- g goes through a per-wave LDS
//     tile [o][p] (8-byte writes of four consecutive pixels) and comes back as the B operand by transposed reads;
//     A = W1^T fragments live in registers.  The output leaves lane p with 16 channels of ITS pixel: NHWC stores.
#pragma once
#include "cy_conv_tile.h"

namespace {

constexpr int DPM_WGS = 512;        // persistent workgroups (2 per CU), four independent waves each
constexpr int DPM_PART = 68 * 64;   // floats of one workgroup's dW1/db1 partial: [16 * 4 + 4 registers][64 lanes]

template <typename T> using DpFrag = typename Mma<T>::Frag;

template <typename T> __device__ __forceinline__ DpFrag<T> dpm_frag(const u32x4& v) {
  DpFrag<T> f;
  f.v = __builtin_bit_cast(decltype(f.v), v);
  return f;
}

template <typename T>
__device__ __forceinline__ DpFrag<T> dpm_tr_frag(const unsigned char* lo, const unsigned char* hi) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  DpFrag<T> f;
  f.v = __builtin_bit_cast(decltype(f.v), v);
  return f;
}

// wave-uniform description of a job
struct DpJob {
  int n, r0, c0, bw, npx;  // pixel rectangle (npx == 0: empty)
  int nbin;                // bins whose pooled gradient reaches these pixels (backward)
  int bin[4];              // row of dhpool
  float inv[4];            // 1 / |bin|
  int colour;
};

__device__ __forceinline__ DpJob dpm_bin_job(const int32_t* bins, int b, int sh, int sw, int H, int W) {
  const BinRect R = bin_rect(bins, b, sh, sw, H, W);
  DpJob J;
  J.n = R.n, J.r0 = R.r0, J.c0 = R.c0, J.bw = R.c1 - R.c0;
  J.npx = (R.r1 - R.r0) * J.bw;
  J.nbin = 1;
  J.bin[0] = b, J.bin[1] = J.bin[2] = J.bin[3] = 0;
  J.inv[0] = 1.f / (float)J.npx, J.inv[1] = J.inv[2] = J.inv[3] = 0.f;
  J.colour = R.colour;
  return J;
}

// segment a (0 .. 2s-2) of an axis of length L pooled into s bins: even = pixels of bin a/2 alone,
// odd = the pixel (if any) shared by bins (a-1)/2 and (a+1)/2
__device__ __forceinline__ void dpm_segment(int a, int L, int s, int& lo, int& hi, int& j0, int& nj) {
  const int j = a >> 1;
  j0 = j;
  if ((a & 1) == 0) {
    lo = (j * L + s - 1) / s, hi = ((j + 1) * L) / s, nj = 1;
  } else {
    lo = ((j + 1) * L) / s, hi = ((j + 1) * L + s - 1) / s, nj = 2;
  }
}

__device__ __forceinline__ DpJob dpm_cell_job(int idx, int sh, int sw, int H, int W) {
  const int na = 2 * sh - 1, nbb = 2 * sw - 1;
  const int b = idx % nbb, t = idx / nbb;
  const int a = t % na, n = t / na;
  int r0, r1, i0, ni, c0, c1, j0, nj;
  dpm_segment(a, H, sh, r0, r1, i0, ni);
  dpm_segment(b, W, sw, c0, c1, j0, nj);
  DpJob J;
  J.n = n, J.r0 = r0, J.c0 = c0, J.bw = c1 - c0;
  J.npx = (r1 - r0) * (c1 - c0);
  J.nbin = ni * nj;
  J.colour = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + (k >> 1), j = j0 + (k & 1);
    const bool on = (k >> 1) < ni && (k & 1) < nj;
    const int br0 = (i * H) / sh, br1 = ((i + 1) * H + sh - 1) / sh;
    const int bc0 = (j * W) / sw, bc1 = ((j + 1) * W + sw - 1) / sw;
    J.bin[k] = on ? (n * sh + i) * sw + j : 0;
    J.inv[k] = on ? 1.f / (float)((br1 - br0) * (bc1 - bc0)) : 0.f;
  }
  return J;
}

// A fragments (both k-steps) of pixel row `blk*32 + r` of the job; rows past the job's last pixel read as zero
template <typename T, int CB>
__device__ __forceinline__ void dpm_load_rows(const T* __restrict__ x, const DpJob& J, long pix0, float inv_bw,
                                              int W, int ldx, int blk, int r, int h, u32x4 (&f)[CB][2]) {
  const int q = blk * 32 + r;
  const bool ok = q < J.npx;
  const int qc = ok ? q : 0;
  const int qr = (int)(((float)qc + 0.5f) * inv_bw), qcol = qc - qr * J.bw;
  const T* p = x + (pix0 + (long)qr * W + qcol) * ldx + 8 * h;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    f[cb][0] = ld16(p + 32 * cb);
    f[cb][1] = ld16(p + 32 * cb + 16);
    if (!ok) f[cb][0] = f[cb][1] = u32x4{0u, 0u, 0u, 0u};
  }
}

// B fragments of W1 (f32 [hid][32], rounded to T): column = hidden unit o, k = channel 16*ks + 8h + i
template <typename T>
__device__ __forceinline__ void dpm_load_w(const float* __restrict__ w1, int C, int cb, int o, int h, DpFrag<T> (&wf)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float f[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(w1 + (size_t)o * C + cb * 32 + ks * 16 + 8 * h);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w1 + (size_t)o * C + cb * 32 + ks * 16 + 8 * h + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = a[i], f[4 + i] = b[i];
    wf[ks] = dpm_frag<T>(Chunk<T>::pack(f));
  }
}

// a job as the block loops see it; jobs past the end (or of another colour class) are empty.  The kernels set the
// NEXT job up -- descriptor, coefficient loads, first pixel block -- before they work through the current one: a
// job's chain of dependent global round trips (bins -> dhpool -> x) would otherwise cost 3-5 us each, as much as its
// arithmetic
struct DpRun {
  DpJob J;
  int nblk;
  float inv_bw;
  long pix0;
};

template <bool CELLS>
__device__ __forceinline__ DpRun dpm_run(const int32_t* bins, int job, int njobs, int sh, int sw, int H, int W,
                                         int colour) {
  const int jc = job < njobs ? job : njobs - 1;
  DpRun R;
  R.J = CELLS ? dpm_cell_job(jc, sh, sw, H, W) : dpm_bin_job(bins, jc, sh, sw, H, W);
  if (job >= njobs || (!CELLS && colour >= 0 && R.J.colour != colour)) R.J.npx = 0;
  R.nblk = (R.J.npx + 31) >> 5;
  R.inv_bw = 1.f / (float)(R.J.bw > 0 ? R.J.bw : 1);
  R.pix0 = ((long)R.J.n * H + R.J.r0) * W + R.J.c0;
  return R;
}

// Backward jobs come from a table written by dpm_jobs_kernel (one thread per job): the ~25 integer divisions of a
// cell descriptor cost more than the arithmetic of a one-block cell when every wave redoes them on the vector unit.
struct __attribute__((aligned(64))) DpRec {
  long pix0;      // first pixel of the rectangle
  int bw, npx, nblk, colour;
  float inv_bw;
  int pad;
  int bin[4];     // rows of dhpool
  float inv[4];   // 1 / |bin| (0: unused slot)
};
static_assert(sizeof(DpRec) == 64, "one 64-byte scalar load per job");

__global__ void __launch_bounds__(256)
    dpm_jobs_kernel(const int32_t* __restrict__ bins, int njobs, DpRec* __restrict__ recs, int cells, int sh, int sw,
                    int H, int W) {
  const int job = blockIdx.x * 256 + threadIdx.x;
  if (job >= njobs) return;
  const DpJob J = cells ? dpm_cell_job(job, sh, sw, H, W) : dpm_bin_job(bins, job, sh, sw, H, W);
  DpRec R;
  R.pix0 = ((long)J.n * H + J.r0) * W + J.c0;
  R.bw = J.bw, R.npx = J.npx, R.nblk = (J.npx + 31) >> 5, R.colour = J.colour;
  R.inv_bw = 1.f / (float)(J.bw > 0 ? J.bw : 1);
  R.pad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) R.bin[k] = J.bin[k], R.inv[k] = J.inv[k];
  recs[job] = R;
}

// the record of `job`, empty (nblk = npx = 0) past the end or for another colour class; wave-uniform
__device__ __forceinline__ DpRec dpm_rec(const DpRec* __restrict__ recs, int job, int njobs, int colour) {
  // one vector load (lane k reads dword k) + 16 readlanes instead of a scalar load through the constant cache
  const int* p = reinterpret_cast<const int*>(recs + (job < njobs ? job : njobs - 1));
  const int v = p[threadIdx.x & 15];
  int w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = __builtin_amdgcn_readlane(v, k);
  DpRec R;
  R.pix0 = (long)(((unsigned long long)(unsigned)w[1] << 32) | (unsigned)w[0]);
  R.bw = w[2], R.npx = w[3], R.nblk = w[4], R.colour = w[5];
  R.inv_bw = __int_as_float(w[6]);
  R.pad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) R.bin[k] = w[8 + k], R.inv[k] = __int_as_float(w[12 + k]);
  if (job >= njobs || (colour >= 0 && R.colour != colour)) R.npx = 0, R.nblk = 0;
  return R;
}

__device__ __forceinline__ float dpm_rec_coef(const float* __restrict__ dhpool, const DpRec& R, int hid, int o) {
  const float d0 = dhpool[(size_t)R.bin[0] * hid + o], d1 = dhpool[(size_t)R.bin[1] * hid + o];
  const float d2 = dhpool[(size_t)R.bin[2] * hid + o], d3 = dhpool[(size_t)R.bin[3] * hid + o];
  return (d0 * R.inv[0] + d1 * R.inv[1]) + (d2 * R.inv[2] + d3 * R.inv[3]);
}

template <typename T, int CB>
__device__ __forceinline__ void dpm_rec_rows(const T* __restrict__ x, const DpRec& R, int W, int ldx, int blk, int r,
                                             int h, u32x4 (&f)[CB][2]) {
  const int q = blk * 32 + r;
  const bool ok = q < R.npx;
  const int qc = ok ? q : 0;
  const int qr = (int)(((float)qc + 0.5f) * R.inv_bw), qcol = qc - qr * R.bw;
  const T* p = x + (R.pix0 + (long)qr * W + qcol) * ldx + 8 * h;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    f[cb][0] = ld16(p + 32 * cb);
    f[cb][1] = ld16(p + 32 * cb + 16);
    if (!ok) f[cb][0] = f[cb][1] = u32x4{0u, 0u, 0u, 0u};
  }
}

// W1 rounded to T as an LDS image [hid][32] (64-byte rows, dpm_xoff swizzle): the backward kernels read their B
// fragments from it instead of pinning 4 registers per (32 units, k-step)
__device__ __forceinline__ int dpm_xoff(int row, int slot);
// (CB channel blocks: one such image per block, 256 * 64 bytes apart)
template <typename T, int CB>
__device__ __forceinline__ void dpm_stage_w(const float* __restrict__ w1, int hid, unsigned char* sw1) {
  for (int idx = threadIdx.x; idx < CB * hid * 4; idx += 256) {
    const int cb = idx / (hid * 4), rem = idx - cb * hid * 4;
    const int row = rem >> 2, slot = rem & 3;
    float f[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(w1 + (size_t)row * (32 * CB) + cb * 32 + slot * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w1 + (size_t)row * (32 * CB) + cb * 32 + slot * 8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = a[i], f[4 + i] = b[i];
    st16(sw1 + cb * (256 * 64) + dpm_xoff(row, slot), Chunk<T>::pack(f));
  }
}

// ------------------------------------------------------------------------------------------------ forward
// hpool[bin][o] = mean over the bin's pixels of lrelu(W1 x + b1).  NHB 32-unit blocks of hidden units per wave.
template <typename T, int NHB, int CB>
__global__ void __launch_bounds__(256, 2)
    dense_proj_mfma_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                               const int32_t* __restrict__ bins, int nb, float* __restrict__ hpool, int H, int W,
                               int ldx, int hid, int sh, int sw, float slope) {
  using M = Mma<T>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int npass = hid / (32 * NHB);
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int o0 = (gw % npass) * NHB * 32;

  DpFrag<T> wf[NHB][CB][2];
  float bias[NHB], sbias[NHB], lbias[NHB];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) dpm_load_w<T>(w1, 32 * CB, cb, o0 + hb * 32 + r, h, wf[hb][cb]);
    bias[hb] = b1[o0 + hb * 32 + r];
    sbias[hb] = slope * bias[hb];
    lbias[hb] = fmaxf(bias[hb], sbias[hb]);
  }

  const int stride = nw / npass;
  DpRun cur = dpm_run<false>(bins, gw / npass, nb, sh, sw, H, W, -1);
  u32x4 f0[CB][2], f1[CB][2];
  dpm_load_rows<T, CB>(x, cur.J, cur.pix0, cur.inv_bw, W, ldx, 0, r, h, f0);
  for (int b = gw / npass; b < nb; b += stride) {
    const DpRun nxt = dpm_run<false>(bins, b + stride, nb, sh, sw, H, W, -1);
    u32x4 n0[CB][2];
    dpm_load_rows<T, CB>(x, nxt.J, nxt.pix0, nxt.inv_bw, W, ldx, 0, r, h, n0);
    const DpJob& J = cur.J;
    const int nblk = cur.nblk;
    float sum[NHB];
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) sum[hb] = 0.f;

    auto block = [&](const u32x4 (&f)[CB][2]) {
#pragma unroll
      for (int hq = 0; hq < NHB / 2; ++hq) {  // two blocks of hidden units at a time: 32 accumulator registers live
        f32x16 acc[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;  // (a bias splat would pin 16 registers per block for the loop)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int k = 0; k < 2; ++k) M::mma(dpm_frag<T>(f[cb][ks]), wf[2 * hq + k][cb][ks], acc[k]);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            // lrelu(t) = max(t, slope * t) for slope <= 1, with t = v + b and slope * t = fma(slope, v, slope * b)
            const float v = acc[k][i];
            sum[2 * hq + k] += fmaxf(v + bias[2 * hq + k], fmaf(slope, v, sbias[2 * hq + k]));
          }
        __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler interleaves all pairs and spills)
      }
    };

    // two register sets; every load is unconditional (a conditional one would turn the waits into vmcnt(0))
    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      dpm_load_rows<T, CB>(x, J, cur.pix0, cur.inv_bw, W, ldx, blk + 1, r, h, f1);
      block(f0);
      dpm_load_rows<T, CB>(x, J, cur.pix0, cur.inv_bw, W, ldx, min(blk + 2, nblk - 1), r, h, f0);
      block(f1);
    }
    if (nblk & 1) block(f0);

    const float npad = (float)(nblk * 32 - J.npx);
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) {
      float s = sum[hb] + __shfl_xor(sum[hb], 32, 64);
      s -= npad * lbias[hb];
      if (h == 0) hpool[(size_t)b * hid + o0 + hb * 32 + r] = s * J.inv[0];
    }
    cur = nxt;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) f0[cb][0] = n0[cb][0], f0[cb][1] = n0[cb][1];
  }
}

// ------------------------------------------------------------------------------------------------ backward
// swizzled byte offset of (row, 16-byte slot) in a per-wave tile of 64-byte rows (4 slots): rows 4 apart share banks
__device__ __forceinline__ int dpm_xoff(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }
// ... of (row, 8-byte slot) in a tile of 64-byte rows (8 slots)
__device__ __forceinline__ int dpm_goff(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 7)) << 3); }

// dW1 / db1 partials: four 32-unit blocks of hidden units per workgroup (pass = blockIdx.x % (hid/128)),
// each wave its own jobs; the workgroup adds its four waves through LDS and writes ONE partial.
template <typename T, bool CELLS, int CB>
__global__ void __launch_bounds__(256, 2)
    dense_proj_mfma_dw_kernel(const T* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                              const DpRec* __restrict__ recs, int njobs, const float* __restrict__ dhpool,
                              float* __restrict__ part, int H, int W, int ldx, int hid, int sh, int sw,
                              float slope, int cbs) {
  using M = Mma<T>;
  constexpr int HB = 4;
  __shared__ __attribute__((aligned(16))) float red[2 * DPM_PART];  // (the x tiles alias its head)
  __shared__ __attribute__((aligned(16))) unsigned char sw1[CB * 256 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int gsel = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int npass = hid / (32 * HB);
  const int pass = blockIdx.x % npass, o0 = pass * HB * 32;
  const int slot = (blockIdx.x / npass) * 4 + wave, nslot = (gridDim.x / npass) * 4;
  unsigned char* sx = reinterpret_cast<unsigned char*>(red) + wave * 2048;

  dpm_stage_w<T, CB>(w1, hid, sw1);
  __syncthreads();
  float nbias[HB];
  int woff[2];  // fragment of (hidden unit o0 + r, k-step ks); + hb * 2048 bytes (the swizzle repeats every 16 rows)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = dpm_xoff(o0 + r, 2 * ks + h);
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) nbias[hb] = -b1[o0 + hb * 32 + r];
  f32x16 dw[HB];
  float db[HB];
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) {
    db[hb] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[hb][i] = 0.f;
  }
  // transposed reads of the x tile: rows 16s + 4h + q (lo) and + 8 (hi), channels 16*gsel + 4*p4 .. +3
  int xlo[2], xhi[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int rl = 16 * s + 4 * h + q4, rh = rl + 8;
    xlo[s] = dpm_xoff(rl, 2 * gsel + (p4 >> 1)) + 8 * (p4 & 1);
    xhi[s] = dpm_xoff(rh, 2 * gsel + (p4 >> 1)) + 8 * (p4 & 1);
  }

  // (Only the first pixel block of the next job is requested ahead here, and in the dx kernel below.  With the next
  //  job's 16 coefficient loads in flight as well, THIS kernel's dW1 / db1 came out different from run to run
  //  (relative 3e-3) on geometries where a wave gets three or more jobs; full s_waitcnt's
  //  (-mllvm -amdgpu-waitcnt-forcezero) made it exact again, a full vmcnt wait before the hand-over, an lgkmcnt wait
  //  around the x tile or s_nops behind the MFMAs did not.  Round 3 re-read the ISA of both forms (carried loads:
  //  8 + 2 + 8 loads issued at the top of a job, `vmcnt(1)` / `vmcnt(0)` of the first block's row loads retire all of
  //  them in order before any MFMA of the job, `vmcnt(14) / (4) / (0)` in front of the hand-over on the path that
  //  skips the block loop): every counted wait is sufficient on every path, no load is left in flight across an
  //  MFMA block in either form -- so the difference between the forms is timing only, and the cause is not a missing
  //  wait on these loads.  Not identified; NO kernel of the library runs the carried form, and the parity test
  //  asserts run-to-run bit equality of dx, dW1 and db1 at 3-4 jobs per wave.)
  DpRec cur = dpm_rec(recs, slot, njobs, -1);
  float dpos[HB];
  u32x4 f0[CB][2], f1[CB][2];
  dpm_rec_rows<T, CB>(x, cur, W, ldx, 0, r, h, f0);
  for (int job = slot; job < njobs; job += nslot) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) dpos[hb] = dpm_rec_coef(dhpool, cur, hid, o0 + hb * 32 + r);
    const DpRec nxt = dpm_rec(recs, job + nslot, njobs, -1);
    u32x4 n0[CB][2];
    dpm_rec_rows<T, CB>(x, nxt, W, ldx, 0, r, h, n0);
    const DpRec& J = cur;
    const int nblk = cur.nblk;
    float dneg[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) dneg[hb] = slope * dpos[hb];

    auto block = [&](const u32x4 (&f)[CB][2]) {
      // (the x tile of the second product: the selected channel block's 32 channels)
      u32x4 s0 = f[0][0], s1 = f[0][1];
#pragma unroll
      for (int cb = 1; cb < CB; ++cb)
        if (cbs == cb) s0 = f[cb][0], s1 = f[cb][1];
      st16(sx + dpm_xoff(r, h), s0);
      st16(sx + dpm_xoff(r, 2 + h), s1);
      __builtin_amdgcn_wave_barrier();
      DpFrag<T> xt[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) xt[s] = dpm_tr_frag<T>(sx + xlo[s], sx + xhi[s]);
#pragma unroll
      for (int hq = 0; hq < HB / 2; ++hq) {
        f32x16 acc[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int k = 0; k < 2; ++k)
              M::mma(dpm_frag<T>(f[cb][ks]), dpm_frag<T>(ld16(sw1 + cb * (256 * 64) + woff[ks] + (2 * hq + k) * 2048)), acc[k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int hb = 2 * hq + k;
          float g[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            g[i] = acc[k][i] > nbias[hb] ? dpos[hb] : dneg[hb];  // pre = acc + b1 > 0
            db[hb] += g[i];
          }
          const DpFrag<T> g0 = dpm_frag<T>(Chunk<T>::pack(g)), g1 = dpm_frag<T>(Chunk<T>::pack(g + 8));
          M::mma(xt[0], g0, dw[hb]);
          M::mma(xt[1], g1, dw[hb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by the next block)
    };

    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      dpm_rec_rows<T, CB>(x, J, W, ldx, blk + 1, r, h, f1);
      block(f0);
      dpm_rec_rows<T, CB>(x, J, W, ldx, min(blk + 2, nblk - 1), r, h, f0);
      block(f1);
    }
    if (nblk & 1) block(f0);
    const float npad = (float)(nblk * 32 - J.npx);
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {  // (per lane: the two halves hold the padded rows between them)
      db[hb] -= 0.5f * npad * (0.f > nbias[hb] ? dpos[hb] : dneg[hb]);
    }
    cur = nxt;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) f0[cb][0] = n0[cb][0], f0[cb][1] = n0[cb][1];
  }

  // the workgroup's partial, in register order [e = hb*16 + reg | 64 + hb][lane]: (w0 + w2) + (w1 + w3)
  __syncthreads();  // every wave is done with its x tile
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) db[hb] += __shfl_xor(db[hb], 32, 64);  // both lane halves: the unit's total
  auto put = [&](float* buf) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[(hb * 16 + i) * 64 + lane] = dw[hb][i];
      buf[(64 + hb) * 64 + lane] = db[hb];
    }
  };
  auto add = [&](const float* buf) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[hb][i] += buf[(hb * 16 + i) * 64 + lane];
      db[hb] += buf[(64 + hb) * 64 + lane];
    }
  };
  if (wave >= 2) put(red + (wave & 1) * DPM_PART);
  __syncthreads();
  if (wave < 2) add(red + wave * DPM_PART);
  __syncthreads();
  if (wave == 1) put(red);
  __syncthreads();
  if (wave == 0) {
    add(red);
    put(part + (size_t)blockIdx.x * DPM_PART);
  }
}

// dW1[o][c] / db1[o] <- sum over the workgroups of pass o/128, in workgroup order (bit-reproducible)
__global__ void __launch_bounds__(256)
    dense_proj_mfma_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                  int nwg, int npass, int accumulate, int ldw, int coff) {
  __shared__ double sred[8][32];
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int E = blockIdx.x * 32 + el;  // element of [pass][68][64]
  const int pass = E / DPM_PART, e = E - pass * DPM_PART;
  double s = 0.0;
  if (pass < npass)
    for (int wg = pass + npass * grp; wg < nwg; wg += npass * 8) s += (double)part[(size_t)wg * DPM_PART + e];
  sred[grp][el] = s;
  __syncthreads();
  if (grp != 0 || pass >= npass) return;
#pragma unroll
  for (int g = 1; g < 8; ++g) s += sred[g][el];
  const int reg = e >> 6, lane = e & 63, r = lane & 31, h = lane >> 5;
  if (reg < 64) {
    const int hb = reg >> 4, i = reg & 15;
    const int o = pass * 128 + hb * 32 + r, c = (i & 3) + 8 * (i >> 2) + 4 * h;
    if (dw) dw[(size_t)o * ldw + coff + c] = accumulate ? dw[(size_t)o * ldw + coff + c] + (float)s : (float)s;
  } else if (db && h == 0) {  // (both lane halves carry the unit's total)
    const int o = pass * 128 + (reg - 64) * 32 + r;
    db[o] = accumulate ? db[o] + (float)s : (float)s;
  }
}

// dx: all hidden units in one wave (NHB = hid/32 blocks, four at a time through the accumulators)
template <typename T, int NHB, bool CELLS, int CB>
__global__ void __launch_bounds__(256, 2)
    dense_proj_mfma_dx_kernel(const T* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                              const DpRec* __restrict__ recs, int njobs, const float* __restrict__ dhpool,
                              T* __restrict__ dx, int H, int W, int ldx, int hid, int sh, int sw, float slope,
                              int colour, int cbs) {
  using M = Mma<T>;
  __shared__ __attribute__((aligned(16))) unsigned char sgall[4 * 128 * 64];  // per wave: g^T [128 units][32 pixels]
  __shared__ __attribute__((aligned(16))) unsigned char sw1[CB * 256 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int gsel = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  unsigned char* sg = sgall + wave * (128 * 64);

  dpm_stage_w<T, CB>(w1, hid, sw1);  // pre-activations: B operand from LDS
  __syncthreads();
  int woff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = dpm_xoff(r, 2 * ks + h);
  DpFrag<T> wt[NHB * 2];  // dx: A operand, row = channel r, k = hidden unit 16t + 8h + i
  float nbias[NHB];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) nbias[hb] = -b1[hb * 32 + r];
#pragma unroll
  for (int t = 0; t < NHB * 2; ++t) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = w1[(size_t)(16 * t + 8 * h + i) * (32 * CB) + cbs * 32 + r];  // (the selected block's channels)
    wt[t] = dpm_frag<T>(Chunk<T>::pack(f));
  }
  // transposed reads of the g tile: hidden units 16t + 8h + q (lo) / + 4 (hi), pixels 16*gsel + 4*p4 .. +3; the
  // swizzle term of row 16t + 8h + q is (4t + 2h) & 7: two variants (t even / odd) + t * 1024 bytes
  int glo[2], ghi[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rl = 16 * t + 8 * h + q4, rh = rl + 4;
    glo[t] = dpm_goff(rl, 4 * gsel + p4) - t * 1024;
    ghi[t] = dpm_goff(rh, 4 * gsel + p4) - t * 1024;
  }

  // (as in the dW1 kernel: only the next job's first pixel block is requested ahead -- this kernel never showed that
  //  kernel's run-to-run differences with the coefficient loads in flight too, but the cause there is not understood)
  DpRec cur = dpm_rec(recs, gw, njobs, colour);
  float dpos[NHB];
  u32x4 f0[CB][2], f1[CB][2];
  dpm_rec_rows<T, CB>(x, cur, W, ldx, 0, r, h, f0);
  for (int job = gw; job < njobs; job += nw) {
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) dpos[hb] = dpm_rec_coef(dhpool, cur, hid, hb * 32 + r);
    const DpRec nxt = dpm_rec(recs, job + nw, njobs, colour);
    u32x4 n0[CB][2];
    dpm_rec_rows<T, CB>(x, nxt, W, ldx, 0, r, h, n0);
    const DpRec& J = cur;
    const int nblk = cur.nblk;

    auto block = [&](const u32x4 (&f)[CB][2], int blk) {
      f32x16 dxa;
#pragma unroll
      for (int i = 0; i < 16; ++i) dxa[i] = 0.f;
#pragma unroll
      for (int hq = 0; hq < NHB / 4; ++hq) {  // 128 hidden units through the g tile at a time
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          f32x16 acc[2];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int k = 0; k < 2; ++k)
                M::mma(dpm_frag<T>(f[cb][ks]),
                       dpm_frag<T>(ld16(sw1 + cb * (256 * 64) + woff[ks] + (hq * 4 + hp * 2 + k) * 2048)), acc[k]);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float dp = dpos[hq * 4 + hp * 2 + k], dn = slope * dp, bs = nbias[hq * 4 + hp * 2 + k];
            float g[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = acc[k][i] > bs ? dp : dn;  // pre = acc + b1 > 0
            const u32x4 lo = Chunk<T>::pack(g), hi = Chunk<T>::pack(g + 8);
            // registers 4j .. 4j+3 = pixels 8j + 4h + 0..3 of hidden unit row: 8-byte slot 2j + h
            const int row = (hp * 2 + k) * 32 + r;
            *reinterpret_cast<u32x2*>(sg + dpm_goff(row, 0 + h)) = u32x2{lo[0], lo[1]};
            *reinterpret_cast<u32x2*>(sg + dpm_goff(row, 2 + h)) = u32x2{lo[2], lo[3]};
            *reinterpret_cast<u32x2*>(sg + dpm_goff(row, 4 + h)) = u32x2{hi[0], hi[1]};
            *reinterpret_cast<u32x2*>(sg + dpm_goff(row, 6 + h)) = u32x2{hi[2], hi[3]};
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const DpFrag<T> gt = dpm_tr_frag<T>(sg + glo[t & 1] + t * 1024, sg + ghi[t & 1] + t * 1024);
          M::mma(wt[hq * 8 + t], gt, dxa);
        }
        __builtin_amdgcn_wave_barrier();
      }
      // lane (pixel r of the block, half h): channels 8j + 4h + 0..3 in registers 4j .. 4j+3
      const int q = blk * 32 + r;
      if (q < J.npx) {
        const int qr = (int)(((float)q + 0.5f) * J.inv_bw), qcol = q - qr * J.bw;
        T* p = dx + (J.pix0 + (long)qr * W + qcol) * ldx + cbs * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          T pk[4];
          if constexpr (CELLS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = from_f32<T>(dxa[4 * j + e]);
          } else {
            const u32x2 old = *reinterpret_cast<const u32x2*>(p + 8 * j);
            float fo[8];
            Chunk<T>::unpack(u32x4{old[0], old[1], 0u, 0u}, fo);
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = from_f32<T>(fo[e] + dxa[4 * j + e]);
          }
          *reinterpret_cast<u32x2*>(p + 8 * j) = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
        }
      }
    };

    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      dpm_rec_rows<T, CB>(x, J, W, ldx, blk + 1, r, h, f1);
      block(f0, blk);
      dpm_rec_rows<T, CB>(x, J, W, ldx, min(blk + 2, nblk - 1), r, h, f0);
      block(f1, blk + 1);
    }
    if (nblk & 1) block(f0, nblk - 1);
    cur = nxt;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) f0[cb][0] = n0[cb][0], f0[cb][1] = n0[cb][1];
  }
}

inline bool dpm_enabled() {
  static const bool on = [] {
    const char* e = getenv("CY_DENSE_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
}

inline bool dpm_applicable(int dtype, int C, int hid, float slope) {
  return dpm_enabled() && (dtype == CY_BF16 || dtype == CY_F16) && (C == 32 || C == 64 || C == 128) && (hid == 128 || hid == 256) &&
         slope <= 1.f;  // (the forward kernel forms lrelu as max(t, slope * t))
}

template <typename T, int CB>
int dpm_launch_fwd_cb(const T* x, const float* w1, const float* b1, const int32_t* bins, int nb, float* hpool, int H,
                      int W, int ldx, int hid, int sh, int sw, float slope, hipStream_t st) {
  const int grid = nb < DPM_WGS * 4 ? cy_cdiv(nb, 4) : DPM_WGS;
  // (32 hidden units x CB channel blocks x 2 k-steps of W1 fragments stay in registers: eight blocks of units per wave
  //  for C = 32, four for C = 64 -- 128 registers either way; 256 units at C = 64 take two passes over the pixels)
#define CY_DPM_FWD(NHB_)                                                                                                  \
  hipLaunchKernelGGL((dense_proj_mfma_fwd_kernel<T, NHB_, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, bins, nb, hpool, H, \
                     W, ldx, hid, sh, sw, slope)
  if constexpr (CB >= 4) {  // (128 channels: two blocks of units per wave, 64 registers of W1 fragments next to 96 of rows)
    CY_DPM_FWD(2);
  } else if constexpr (CB == 2) {
    CY_DPM_FWD(4);
  } else {
    if (hid == 256) CY_DPM_FWD(8);
    else CY_DPM_FWD(4);
  }
#undef CY_DPM_FWD
  CY_CHECK_LAUNCH();
  return CY_OK;
}

template <typename T>
int dpm_launch_fwd(const T* x, const float* w1, const float* b1, const int32_t* bins, int nb, float* hpool, int H,
                   int W, int ldx, int C, int hid, int sh, int sw, float slope, hipStream_t st) {
  if (C == 128) return dpm_launch_fwd_cb<T, 4>(x, w1, b1, bins, nb, hpool, H, W, ldx, hid, sh, sw, slope, st);
  if (C == 64) return dpm_launch_fwd_cb<T, 2>(x, w1, b1, bins, nb, hpool, H, W, ldx, hid, sh, sw, slope, st);
  return dpm_launch_fwd_cb<T, 1>(x, w1, b1, bins, nb, hpool, H, W, ldx, hid, sh, sw, slope, st);
}

inline int dpm_dw_grid(int njobs, int npass) {
  const int per = cy_cdiv(njobs, 4);  // workgroups per pass that still have a job
  const int cap = DPM_WGS / npass;
  return (per < cap ? per : cap) * npass;
}

template <typename T, int CB>
int dpm_launch_bwd_cb(const T* x, const float* w1, const float* b1, const int32_t* bins, int nb, const float* dhpool,
                      T* dx, float* dw1, float* db1, int accumulate, int N, int H, int W, int ldx, int hid, int sh,
                      int sw, float slope, float* ws, hipStream_t st) {
  const bool cells = bins == nullptr;
  const int njobs = cells ? N * (2 * sh - 1) * (2 * sw - 1) : nb;
  const int npass = hid / 128;
  DpRec* recs = reinterpret_cast<DpRec*>(ws + (size_t)DPM_WGS * DPM_PART);
  hipLaunchKernelGGL(dpm_jobs_kernel, dim3(cy_cdiv(njobs, 256)), dim3(256), 0, st, bins, njobs, recs, cells ? 1 : 0, sh,
                     sw, H, W);
  CY_CHECK_LAUNCH();
  for (int cbs = 0; cbs < CB; ++cbs) {  // the second product for one block of 32 channels per launch
    if (dw1 || (db1 && cbs == 0)) {
      const int grid = dpm_dw_grid(njobs, npass);
      if (cells)
        hipLaunchKernelGGL((dense_proj_mfma_dw_kernel<T, true, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs, njobs,
                           dhpool, ws, H, W, ldx, hid, sh, sw, slope, cbs);
      else
        hipLaunchKernelGGL((dense_proj_mfma_dw_kernel<T, false, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs, njobs,
                           dhpool, ws, H, W, ldx, hid, sh, sw, slope, cbs);
      CY_CHECK_LAUNCH();
      hipLaunchKernelGGL(dense_proj_mfma_reduce_kernel, dim3(cy_cdiv((long)npass * DPM_PART, 32)), dim3(256), 0, st,
                         (const float*)ws, dw1, cbs == 0 ? db1 : (float*)nullptr, grid, npass, accumulate, 32 * CB, 32 * cbs);
      CY_CHECK_LAUNCH();
    }
    if (dx) {
      const int grid = njobs < DPM_WGS * 4 ? cy_cdiv(njobs, 4) : DPM_WGS;
      if (cells) {
        if (hid == 256)
          hipLaunchKernelGGL((dense_proj_mfma_dx_kernel<T, 8, true, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs,
                             njobs, dhpool, dx, H, W, ldx, hid, sh, sw, slope, -1, cbs);
        else
          hipLaunchKernelGGL((dense_proj_mfma_dx_kernel<T, 4, true, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs,
                             njobs, dhpool, dx, H, W, ldx, hid, sh, sw, slope, -1, cbs);
        CY_CHECK_LAUNCH();
      } else {
        for (int colour = 0; colour < 4; ++colour) {
          if (hid == 256)
            hipLaunchKernelGGL((dense_proj_mfma_dx_kernel<T, 8, false, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs,
                               njobs, dhpool, dx, H, W, ldx, hid, sh, sw, slope, colour, cbs);
          else
            hipLaunchKernelGGL((dense_proj_mfma_dx_kernel<T, 4, false, CB>), dim3(grid), dim3(256), 0, st, x, w1, b1, recs,
                               njobs, dhpool, dx, H, W, ldx, hid, sh, sw, slope, colour, cbs);
          CY_CHECK_LAUNCH();
        }
      }
    }
  }
  return CY_OK;
}

template <typename T>
int dpm_launch_bwd(const T* x, const float* w1, const float* b1, const int32_t* bins, int nb, const float* dhpool,
                   T* dx, float* dw1, float* db1, int accumulate, int N, int H, int W, int ldx, int C, int hid, int sh,
                   int sw, float slope, float* ws, hipStream_t st) {
  if (C == 128)
    return dpm_launch_bwd_cb<T, 4>(x, w1, b1, bins, nb, dhpool, dx, dw1, db1, accumulate, N, H, W, ldx, hid, sh, sw, slope,
                                   ws, st);
  if (C == 64)
    return dpm_launch_bwd_cb<T, 2>(x, w1, b1, bins, nb, dhpool, dx, dw1, db1, accumulate, N, H, W, ldx, hid, sh, sw, slope,
                                   ws, st);
  return dpm_launch_bwd_cb<T, 1>(x, w1, b1, bins, nb, dhpool, dx, dw1, db1, accumulate, N, H, W, ldx, hid, sh, sw, slope, ws,
                                 st);
}

}  // namespace
