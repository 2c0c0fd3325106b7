// 3x3 convolution (stride 1, pad 1, no bias) as an implicit GEMM on the gfx950
// matrix cores.  Replaces nn.Conv2d at contrastyou/arch/unet.py:21,24,39 and,
// through the load-side modes, nn.MaxPool2d (unet.py:67-70), nn.Upsample
// (unet.py:38), torch.cat (unet.py:142,151,160,169) and the BatchNorm+ReLU of
// the previous layer (unet.py:22-23).  The same kernel computes the data
// gradient (run on dy with the flipped/transposed weight image).
//
// GEMM view:  out[p][co] = sum_{tap,ci} in[p + tap][ci] * w[tap][co][ci]
//   M = pixels of a TH x TW spatial tile (rows are flattened n*H+h rows, so a
//       tile may span images), N = BN output channels, K = 9 taps x Cin.
// A (activations) is staged once per input-channel chunk as a (TH+2)x(TW+2)
// halo tile in LDS and re-read by all 9 taps at shifted addresses; B (weights)
// streams through a double-buffered LDS slice per (chunk, tap).  Four waves per
// workgroup, each owning M_REP x N_REP 32x32 MFMA accumulators.
// The epilogue rounds to the storage type, emits per-tile per-channel sum /
// sum-of-squares partials for the following BatchNorm (deterministic: one
// partial per tile, fixed-order reduction in cy_bn_finalize) and stores
// 16-byte NHWC chunks after a wave-private LDS transpose.
#include "cy_conv_plane.h"
#include "cy_conv_stream.h"
#include "cy_conv_flow.h"
#include "cy_conv_tile.h"

#include <cstdlib>

namespace {

template <typename T, typename TO, int TH, int TW, int BN, int WGM, int WGN, int PITCHB, bool ALLT>
struct ConvCfg {
  static constexpr int EPC = ElemTr<T>::EPC;
  static constexpr int KC = PITCHB / (int)sizeof(T);
  static constexpr int CPP = PITCHB / 16;
  static constexpr int KS = KC / 16;
  static constexpr int HW2 = TW + 2;
  static constexpr int ZSLOT = TH + 2;
  static constexpr int A_BYTES = (TH + 3) * HW2 * PITCHB;
  static constexpr int B_BYTES = BN * PITCHB;
  static constexpr int MT = TH * TW / 32;
  static constexpr int M_REP = MT / WGM;
  static constexpr int N_REP = BN / (32 * WGN);
  static constexpr int BREG = (BN * CPP + 255) / 256;
  static constexpr int TAB_BYTES = (3 * TH + 4) * 4;
  static constexpr int EPI_BYTES = 4 * 32 * 36 * 4 + WGM * 2 * BN * 4;
  static constexpr int NBUF = ALLT ? 9 : 2;  // B slices resident in LDS: all nine taps, or a 2-deep ring
  static constexpr int MAIN_BYTES = A_BYTES + NBUF * B_BYTES;
  static constexpr int SMEM = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + TAB_BYTES + 16;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert((TH * TW) % 32 == 0 && MT % WGM == 0, "tile/wave split");
  static_assert(BN % (32 * WGN) == 0, "cout/wave split");
  static_assert(256 % CPP == 0, "chunk ownership");
};

template <typename T, typename TO, int TH, int TW, int BN, int WGM, int WGN, int PITCHB, bool ALLT>
__global__ void __launch_bounds__(256, 2)
    conv3x3_igemm_kernel(const ConvArgs a) {
  using C = ConvCfg<T, TO, TH, TW, BN, WGM, WGN, PITCHB, ALLT>;
  using M = Mma<T>;
  constexpr int EPC = C::EPC, KC = C::KC, CPP = C::CPP, KS = C::KS, HW2 = C::HW2;
  constexpr int M_REP = C::M_REP, N_REP = C::N_REP, BREG = C::BREG;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAINB = (C::MAIN_BYTES > C::EPI_BYTES ? C::MAIN_BYTES : C::EPI_BYTES);
  unsigned char* sA = smem;
  unsigned char* sB = smem + C::A_BYTES;
  int* s_row1 = reinterpret_cast<int*>(smem + ((MAINB + 15) & ~15));
  int* s_row2 = s_row1 + (TH + 2);
  int* s_flag = s_row2 + (TH + 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  const int tile = blockIdx.x;
  const int ct = tile % a.tiles_w;
  const int rt = tile / a.tiles_w;
  const int R0 = rt * TH, w0 = ct * TW;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int ncc = (Cin + KC - 1) / KC;
  // split-K: this workgroup reduces the input-channel chunks [cc0, cc1)
  const int cc0 = (ncc * (int)blockIdx.z) / a.ksplit;
  const int cc1 = (ncc * ((int)blockIdx.z + 1)) / a.ksplit;

  // zero slot + row tables
  for (int idx = tid; idx < HW2 * CPP; idx += 256) {
    u32x4 z = {0u, 0u, 0u, 0u};
    st16(sA + (C::ZSLOT * HW2) * PITCHB + idx * 16, z);
  }
  conv_row_tables(a, TH, R0, tid, s_row1, s_row2, s_flag, false);
  __syncthreads();

  // per-lane pixel coordinates of the A rows this lane feeds
  // packed (ty << 12) | (tx << 4) | row flags: one register per A row-tile
  int ppk[M_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int i = (wm * M_REP + m) * 32 + r;
    const int ty = i / TW;
    ppk[m] = (ty << 12) | ((i - ty * TW) << 4) | s_flag[ty];
  }
  // per-lane B rows
  int brow[N_REP], bswz[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) {
    brow[n] = (wn * N_REP + n) * 32 + r;
    bswz[n] = lds_swz<PITCHB>(brow[n]);
  }

  f32x16 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  const T* wp = reinterpret_cast<const T*>(a.w);
  u32x4 breg[BREG];
  auto load_b = [&](int cc, int tap) {
#pragma unroll
    for (int i = 0; i < BREG; ++i) {
      const int idx = tid + i * 256;
      if (idx < BN * CPP) {
        const int row = idx / CPP, ch = idx % CPP;
        breg[i] = ld16(wp + ((size_t)(tap * a.w_co_pad + n0 + row)) * a.w_ci_pad + cc * KC + ch * EPC);
      }
    }
  };
  auto store_b = [&](unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < BREG; ++i) {
      const int idx = tid + i * 256;
      if (idx < BN * CPP) {
        const int row = idx / CPP, ch = idx % CPP;
        st16(dst + row * PITCHB + ((ch ^ lds_swz<PITCHB>(row)) << 4), breg[i]);
      }
    }
  };

  // One tap of one channel chunk: KS k-steps of M_REP x N_REP MFMAs.  The LDS fragments of k-step
  // ks+1 are requested BEFORE the MFMAs of k-step ks are issued (two register sets, order pinned by
  // sched_barrier), so LDS latency hides behind M_REP*N_REP MFMAs instead of one.
  auto mma_tap = [&](const unsigned char* sBc, const unsigned char* const* apix, const int* aswz) {
    typename M::Frag af[2][M_REP], bf[2][N_REP];
#pragma unroll
    for (int n = 0; n < N_REP; ++n) bf[0][n] = M::load(sBc + brow[n] * PITCHB, h, bswz[n]);
#pragma unroll
    for (int m = 0; m < M_REP; ++m) af[0][m] = M::load(apix[m], h, aswz[m]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KS) {
        const int fi = 2 * (ks + 1) + h;
#pragma unroll
        for (int n = 0; n < N_REP; ++n) bf[nxt][n] = M::load(sBc + brow[n] * PITCHB, fi, bswz[n]);
#pragma unroll
        for (int m = 0; m < M_REP; ++m) af[nxt][m] = M::load(apix[m], fi, aswz[m]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int n = 0; n < N_REP; ++n) M::mma(af[cur][m], bf[cur][n], acc[m][n]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (ALLT) {
    // Small-N layers (Cout <= 64): all nine weight taps of a chunk sit in LDS at once, so a
    // chunk is {stage A + B, barrier, 9 taps x KS k-steps of MFMA, barrier}: no per-tap barrier.
    constexpr int NB = 9 * BN * CPP;           // 16-byte chunks of the nine-tap weight slab
    constexpr int NBR = (NB + 255) / 256;
    for (int cc = cc0; cc < cc1; ++cc) {
      if (cc != cc0) __syncthreads();  // every wave is done with the previous chunk's LDS
      u32x4 wreg[NBR];
#pragma unroll
      for (int i = 0; i < NBR; ++i) {
        const int idx = tid + i * 256;
        if (idx < NB) {
          const int tap = idx / (BN * CPP), rem = idx % (BN * CPP);
          const int row = rem / CPP, ch = rem % CPP;
          wreg[i] = ld16(wp + ((size_t)(tap * a.w_co_pad + n0 + row)) * a.w_ci_pad + cc * KC + ch * EPC);
        }
      }
      conv_stage_halo<T, PITCHB, 1>(a, sA, s_row1, s_row2, TH, TW, HW2, w0, cc * KC, tid);
#pragma unroll
      for (int i = 0; i < NBR; ++i) {
        const int idx = tid + i * 256;
        if (idx < NB) {
          const int tap = idx / (BN * CPP), rem = idx % (BN * CPP);
          const int row = rem / CPP, ch = rem % CPP;
          st16(sB + tap * C::B_BYTES + row * PITCHB + ((ch ^ lds_swz<PITCHB>(row)) << 4), wreg[i]);
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const unsigned char* sBc = sB + tap * C::B_BYTES;
        const unsigned char* apix[M_REP];
        int aswz[M_REP];
#pragma unroll
        for (int m = 0; m < M_REP; ++m) {
          int slot = (ppk[m] >> 12) + 1 + dh;
          if ((dh < 0 && (ppk[m] & 1)) || (dh > 0 && (ppk[m] & 2))) slot = C::ZSLOT;
          const int p = slot * HW2 + ((ppk[m] >> 4) & 0xff) + 1 + dw;
          apix[m] = sA + p * PITCHB;
          aswz[m] = lds_swz<PITCHB>(p);
        }
        mma_tap(sBc, apix, aswz);
      }
    }
    __syncthreads();
  } else {
    conv_stage_halo<T, PITCHB, 1>(a, sA, s_row1, s_row2, TH, TW, HW2, w0, cc0 * KC, tid);
    load_b(cc0, 0);
    store_b(sB);
    __syncthreads();

    const int it0 = cc0 * 9, nit = cc1 * 9;
    for (int it = it0; it < nit; ++it) {
      const int cc = it / 9;
      const int tap = it - cc * 9;
      const int dh = tap / 3 - 1, dw = tap % 3 - 1;
      const bool has_next = it + 1 < nit;
      if (has_next) {
        const int it1 = it + 1;
        load_b(it1 / 9, it1 % 9);
      }
      const unsigned char* sBc = sB + ((it - it0) & 1) * C::B_BYTES;

      const unsigned char* apix[M_REP];
      int aswz[M_REP];
  #pragma unroll
      for (int m = 0; m < M_REP; ++m) {
        int slot = (ppk[m] >> 12) + 1 + dh;
        if ((dh < 0 && (ppk[m] & 1)) || (dh > 0 && (ppk[m] & 2))) slot = C::ZSLOT;
        const int p = slot * HW2 + ((ppk[m] >> 4) & 0xff) + 1 + dw;
        apix[m] = sA + p * PITCHB;
        aswz[m] = lds_swz<PITCHB>(p);
      }
      mma_tap(sBc, apix, aswz);

      if (has_next) store_b(sB + ((it + 1 - it0) & 1) * C::B_BYTES);
      if (tap == 8 && has_next) {
        __syncthreads();  // every wave is done reading the halo tile
        conv_stage_halo<T, PITCHB, 1>(a, sA, s_row1, s_row2, TH, TW, HW2, w0, (cc + 1) * KC, tid);
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  if (a.ksplit > 1) {
    // raw f32 partial sums; conv_splitk_finish_kernel adds the splits, rounds, stores, takes stats.
    // Same wave-private LDS transpose as the final epilogue below, so that a lane stores 16 bytes
    // (4 couts of one pixel) instead of 16 scattered dwords per accumulator.
    float* wsz = a.ws + (size_t)blockIdx.z * ((size_t)a.NH * a.W) * a.Cout;
    float* scr = reinterpret_cast<float*>(smem) + wave * (32 * 36);
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      const int cobase = n0 + (wn * N_REP + n) * 32;
#pragma unroll
      for (int m = 0; m < M_REP; ++m) {
        const int ibase = (wm * M_REP + m) * 32;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          scr[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 36 + r] = acc[m][n][reg];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int itx = 0; itx < 4; ++itx) {
          const int idx = lane + itx * 64;
          const int pix = idx >> 3, cch = idx & 7;
          const f32x4 v = *reinterpret_cast<const f32x4*>(scr + pix * 36 + cch * 4);
          const int i = ibase + pix;
          const int ty = i / TW, tx = i - ty * TW;
          const int R = R0 + ty, w = w0 + tx;
          const int co = cobase + cch * 4;
          if (R < a.NH && w < a.W && co < a.Cout)
            *reinterpret_cast<f32x4*>(wsz + ((size_t)R * a.W + w) * a.Cout + co) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }
  constexpr int EPO = 16 / (int)sizeof(TO);  // output elements per 16-byte chunk
  constexpr int CPO = 32 / EPO;              // chunks per 32-cout row
  float* scratch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
  float* sstat = reinterpret_cast<float*>(smem) + 4 * (32 * 36);
  const bool do_stats = a.stats != nullptr || a.sacc != nullptr;
  TO* o1 = reinterpret_cast<TO*>(a.out);
  TO* o2 = reinterpret_cast<TO*>(a.out2);

#pragma unroll
  for (int n = 0; n < N_REP; ++n) {
    float s1 = 0.f, s2 = 0.f;
    const int cobase = n0 + (wn * N_REP + n) * 32;
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int ibase = (wm * M_REP + m) * 32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const float v = acc[m][n][reg];
        scratch[row * 36 + r] = v;
        if (do_stats) {
          bool ok = true;
          if (!a.full_tiles) {
            const int i = ibase + row;
            const int ty = i / TW, tx = i - ty * TW;
            ok = (R0 + ty < a.NH) && (w0 + tx < a.W);
          }
          if (ok) {
            const float q = round_through<TO>(v);
            s1 += q;
            s2 += q * q;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int itx = 0; itx < (32 * CPO) / 64; ++itx) {
        const int idx = lane + itx * 64;
        const int pix = idx / CPO, cch = idx % CPO;
        float f[EPO];
        const f32x4* sp = reinterpret_cast<const f32x4*>(scratch + pix * 36 + cch * EPO);
#pragma unroll
        for (int q = 0; q < EPO / 4; ++q) {
          f32x4 t = sp[q];
          f[4 * q] = t[0];
          f[4 * q + 1] = t[1];
          f[4 * q + 2] = t[2];
          f[4 * q + 3] = t[3];
        }
        const int i = ibase + pix;
        const int ty = i / TW, tx = i - ty * TW;
        const int R = R0 + ty, w = w0 + tx;
        const int co = cobase + cch * EPO;
        if (R < a.NH && w < a.W && co < a.Cout) {
          const size_t gp = (size_t)R * a.W + w;
          u32x4 pk = Chunk<TO>::pack(f);
          if (a.split_c > 0 && co >= a.split_c)
            st16(o2 + gp * a.ldo2 + (co - a.split_c), pk);
          else
            st16(o1 + gp * a.ldo + co, pk);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (do_stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        const int col = (wn * N_REP + n) * 32 + r;
        sstat[(wm * 2 + 0) * BN + col] = s1;
        sstat[(wm * 2 + 1) * BN + col] = s2;
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int q = 0; q < WGM; ++q) {
        t1 += sstat[(q * 2 + 0) * BN + tid];
        t2 += sstat[(q * 2 + 1) * BN + tid];
      }
      if (a.sacc) {
        bn_acc_add(a.sacc, a.sR, a.Cout, tile & (a.sR - 1), 0, n0 + tid, t1);
        bn_acc_add(a.sacc, a.sR, a.Cout, tile & (a.sR - 1), 1, n0 + tid, t2);
      } else {
        a.stats[((size_t)tile * 2 + 0) * a.Cout + n0 + tid] = t1;
        a.stats[((size_t)tile * 2 + 1) * a.Cout + n0 + tid] = t2;
      }
    }
  }
}

// split-K finish: out = round(sum_z ws[z]) (+ split destinations), per-block BN partial sums.
// thread = (pixel lane, group of 8 couts); block = contiguous pixel range.
template <typename TO>
__global__ void __launch_bounds__(256)
    conv_splitk_finish_kernel(const float* __restrict__ ws, int Z, long npix, int Cout,
                              TO* __restrict__ out, int ldo, TO* __restrict__ out2, int ldo2,
                              int split_c, float* __restrict__ stats, unsigned long long* sacc, int sR) {
  extern __shared__ float sred[];  // [2][rows][gpp*8]
  const int G = Cout / 8;
  const int gpp = G < 256 ? G : 256;
  const int rows = 256 / gpp;
  const int tid = threadIdx.x;
  const int g = tid % gpp, prow = tid / gpp;
  const bool active = tid < rows * gpp;
  const long per = (npix + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  const size_t zstride = (size_t)npix * Cout;
  for (int page = 0; page * gpp < G; ++page) {
    const int gg = page * gpp + g;
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
    if (active && gg < G) {
      for (long p = p0 + prow; p < p1; p += rows) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
        const float* src = ws + (size_t)p * Cout + gg * 8;
        for (int z = 0; z < Z; ++z) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(src + z * zstride);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(src + z * zstride + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f[j] += lo[j];
            f[4 + j] += hi[j];
          }
        }
        const int co = gg * 8;
        TO* dst = (split_c > 0 && co >= split_c) ? out2 + (size_t)p * ldo2 + (co - split_c)
                                                 : out + (size_t)p * ldo + co;
        if constexpr (sizeof(TO) == 2) {
          st16(dst, Chunk<TO>::pack(f));
        } else {
          st16(dst, Chunk<float>::pack(f));
          st16(dst + 4, Chunk<float>::pack(f + 4));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float q = round_through<TO>(f[j]);
          a1[j] += q;
          a2[j] += q * q;
        }
      }
    }
    if (stats || sacc) {
      __syncthreads();
      if (active && gg < G) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sred[(0 * rows + prow) * (gpp * 8) + g * 8 + j] = a1[j];
          sred[(1 * rows + prow) * (gpp * 8) + g * 8 + j] = a2[j];
        }
      }
      __syncthreads();
      for (int i = tid; i < 2 * gpp * 8; i += 256) {
        const int which = i / (gpp * 8), cl = i % (gpp * 8);
        const int c = page * gpp * 8 + cl;
        if (c < Cout) {
          float s = 0.f;
          for (int q = 0; q < rows; ++q) s += sred[(which * rows + q) * (gpp * 8) + cl];
          if (sacc) bn_acc_add(sacc, sR, Cout, (int)blockIdx.x & (sR - 1), which, c, s);
          else stats[((size_t)blockIdx.x * 2 + which) * Cout + c] = s;
        }
      }
    }
  }
}

template <typename T, typename TO, int TH, int TW, int BN, int WGM, int WGN, int PITCHB, bool ALLT>
int launch_conv(ConvArgs a, hipStream_t st) {
  using C = ConvCfg<T, TO, TH, TW, BN, WGM, WGN, PITCHB, ALLT>;
  auto kern = conv3x3_igemm_kernel<T, TO, TH, TW, BN, WGM, WGN, PITCHB, ALLT>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles_h = cy_cdiv(a.NH, TH);
  a.tiles_w = cy_cdiv(a.W, TW);
  a.full_tiles = (a.NH % TH == 0) && (a.W % TW == 0);
  dim3 grid(tiles_h * a.tiles_w, cy_cdiv(a.Cout, BN), a.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM, st, a);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

struct TileChoice {
  int th, tw, bn;
};

// Tile selection: exact column tilings where the image width allows, a masked
// 16x16 tile otherwise.  7-MFMA-tile shapes (8x28, 16x14) exist for BN=128.
TileChoice choose_tile(long NH, int W, int Cout) {
  TileChoice c;
  c.bn = Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32);
  if (Cout == 128 && W % 8 == 0) {
    // 128-cout layers with few spatial tiles: two 64-cout tiles give the chip 2 workgroups per CU
    const int th = W % 32 == 0 ? 8 : (W % 16 == 0 ? 16 : 32), tw = 256 / th;
    if (cy_cdiv(NH, th) * cy_cdiv(W, tw) < 300) c.bn = 64;
  }
  if (W % 32 == 0) {
    c.th = 8, c.tw = 32;
  } else if (W % 16 == 0) {
    c.th = 16, c.tw = 16;
  } else if (W % 8 == 0) {
    c.th = 32, c.tw = 8;
  } else if (W == 28 && c.bn == 128) {
    c.th = 8, c.tw = 28;
  } else if (W == 14 && c.bn == 128) {
    c.th = 16, c.tw = 14;
  } else {
    c.th = 16, c.tw = 16;
  }
  return c;
}

// The plane kernel (cy_conv_plane.h) tiles the image in 16 x 14 outputs.  CY_CONV_PLANE=0 keeps
// every layer on conv3x3_igemm_kernel (A/B measurements).
bool use_plane_kernel(int W) {
  static const bool enabled = [] {
    const char* e = getenv("CY_CONV_PLANE");
    return !(e && e[0] == '0');
  }();
  return enabled && W % 14 == 0;
}
constexpr int kPlaneTH = 16, kPlaneTW = 14;

struct ConvPlan {
  bool plane;
  bool stream;      // plane, 16-bit storage, Cin / Cout in {32, 64}, no load transform: persistent streaming kernel (cy_conv_stream.h)
  bool one_per_cu;  // plane, 128 couts: at most one workgroup per CU, halo prefetch in registers
  bool flow;        // eight-wave LDS-DMA kernel (cy_conv_flow.h): one workgroup per CU, tile.th x tile.tw x tile.bn
  TileChoice tile;
  int ksplit;         // >1: split-K over input-channel chunks + finish kernel
  int finish_blocks;  // blocks (= stat partials) of the finish kernel
  int partials;       // number of BN partials the launch sequence writes
  size_t ws_bytes;
};

// Deep layers (14x14 / 28x28 at small batch) have too few output tiles to fill 256 CUs: split
// the reduction over input-channel chunks across blockIdx.z.
ConvPlan plan_conv(int N, int H, int W, int Cin, int Cout, int elem_bytes, bool stream_ok = false,
                   bool prologue = false, FlowChoice fc = FlowChoice{false, 0, 0, 0, false}, bool bwd_pro = false) {
  ConvPlan p;
  p.plane = use_plane_kernel(W);
  p.stream = false;
  p.flow = false;
  static const int stream_mode = [] {
    const char* e = getenv("CY_STREAM");
    return e ? atoi(e) : 1;
  }();
  // (persistent pipeline: worth it from ~4 tiles per workgroup on -- the 224x224 level at any batch size here)
  // (CY_STREAM=2: no tile-count threshold, for experiments)
  // (32 -> 64 at 112 x 112, N = 32 -- Conv2a reading the pooled tensor: 1792 tiles, 34 us against 43 plane / 38 flow)
  const long stream_min_tiles = (Cin == 32 && Cout == 64) ? 1700 : 2048;
  if (p.plane && stream_ok && stream_mode &&
      (stream_mode == 2 || cy_cdiv((long)N * H, kPlaneTH) * (W / kPlaneTW) >= stream_min_tiles)) {
    p.stream = true;
    p.one_per_cu = false;
    p.tile.th = kPlaneTH, p.tile.tw = kPlaneTW, p.tile.bn = Cout;
    p.ksplit = 1;
    p.finish_blocks = 0;
    p.partials = stream_partials(Cin, Cout, cy_cdiv((long)N * H, kPlaneTH) * (W / kPlaneTW), prologue);
    p.ws_bytes = 0;
    return p;
  }
  static const int flow_mode = [] {
    const char* e = getenv("CY_FLOW");
    return e ? atoi(e) : 1;
  }();
  if (fc.ok && flow_mode) {
    static const int flow_cfg = [] {  // experiments: 1 = always the big tile, 2 = the 16-row tile wherever it exists
      const char* e = getenv("CY_FLOW_CFG");
      return e ? atoi(e) : 0;
    }();
    const long npixf = (long)N * H * W;
    const int nccf = Cin / 16;
    // One workgroup per CU, all of them in step (load the first chunk, compute, store): the kernel pays off where
    // one or two full rounds of workgroups cover the layer and the K loop is long enough to amortise the two
    // ends.  Rules from per-layer A/B runs against the plane kernel on one box (tools/flow_sweep.sh,
    // DESIGN.md section 3 "Flow kernel"); everything else stays on the plane kernel.
    int th = fc.th, bn = fc.bn, Z = 1;
    const int tiles_big = cy_cdiv((long)N * H, th) * (W / fc.tw);
    const int blocks_big = tiles_big * (Cout / bn);
    auto fills = [](int blocks) {  // share of the CU-rounds a grid occupies
      const int rounds = (blocks + 255) / 256;
      return (double)blocks / (256.0 * rounds);
    };
    bool use = false;
    // more than a round of four-wave workgroups of 128 positions x 64 couts per wave (32 rows x 64 couts, two per CU:
    // the big tile's LDS traffic per MFMA, and each other's load / store phases covered): where a launch is several
    // rounds of work anyway this beats the one-per-CU tiles (N = 32: 64 -> 64 at 112 x 112 65 (plane) -> 57 us,
    // 128 -> 256 at 56 x 56 data gradient 66 -> 62) -- below that the one-per-CU tiles stay better (Conv3b 41 vs 45)
    const long n3264 = cy_cdiv((long)N * H, 32) * (W / fc.tw) * (Cout / 64);
    constexpr int th3264 = 512;  // (a round of them on 256 CUs at two per CU: swept with tools/flow_sweep.sh)
    // (not with the backward prologue: its extra halo buffer leaves room for one such workgroup per CU only)
    if (flow_cfg == 0 && n3264 >= th3264 && !(prologue && Cin > 256) && !bwd_pro) {
      use = true, th = 32, bn = 64;
    } else if (flow_cfg == 1) {
      use = true;
    } else if (bn == 64) {
      use = blocks_big >= 160 && (blocks_big <= 256 || (nccf >= 8 && fills(blocks_big) >= 0.75));
      // too few 64-row tiles for the chip (56 x 56, 128 -> 64: 56 of them at N = 16): four-wave workgroups on
      // 16 rows x 64 couts, two per CU (Conv3a data gradient 18 -> 14 us)
      const int n64 = cy_cdiv((long)N * H, 16) * (W / fc.tw) * (Cout / 64);
      if (!use && blocks_big < 160 && n64 >= 192 && !(prologue && Cin > 256)) use = true, th = 16;
    } else {
      if (blocks_big >= 192 && (blocks_big <= 256 || (nccf >= 8 && fills(blocks_big) >= 0.85))) {
        use = true;
      } else if (fc.small_ok && blocks_big < 192) {
        th = 16;
        const int blocks_small = cy_cdiv((long)N * H, th) * (W / fc.tw) * (Cout / bn);
        const int n64 = 2 * blocks_small;  // workgroups of the four-wave 16 x 64 tiling (two per CU)
        if (blocks_small >= 192) {
          use = blocks_small <= 256 || (nccf >= 8 && fills(blocks_small) >= 0.85);
        } else if (!(prologue && Cin > 256) && (n64 >= 192 || (n64 >= 96 && nccf < 32))) {
          // too few 16 x 128 tiles for the chip: the same wave tile (64 positions x 64 couts) in four-wave
          // workgroups of 64 couts -- twice the workgroups, two per CU, each other's load / store phases covered
          // (28 x 28 at N = 16: 256 -> 256 data gradient 28 -> 22 us, 128 -> 256 forward 20 -> 16; 14 x 14, 256 ->
          // 512 forward 28 (split-K 4) -> 21); with a long K loop and still few workgroups split-K stays better
          // (14 x 14, 512 -> 512 data gradient: 32 against 36)
          bn = 64;
          use = true;
        } else {
          // too few tiles for the chip: split-K over >= 4 chunks each only where the f32 partial slabs are shared
          // by many cout blocks and the K loop is long (28x28 at N=16, 256 -> 256: 112 workgroups without a
          // split take 33 us, 224 with one 36); otherwise the tiles there are run as they are
          if (Cout >= 256 && (blocks_small < 96 || nccf >= 32)) {
            Z = (232 + blocks_small / 2) / blocks_small;
            if (Z > nccf / 4) Z = nccf / 4;
            if (Z > 8) Z = 8;
            if (Z < 1) Z = 1;
          }
          use = blocks_small * Z >= 48;
        }
      }
    }
    if (flow_cfg == 2 && fc.small_ok) use = true, th = 16, Z = 1;
    if (flow_cfg == 3 && !(prologue && Cin > 256)) use = true, th = 16, bn = 64, Z = 1;  // (experiment: 4-wave 16 x 64 tiles)
    if (flow_cfg == 4 && !(prologue && Cin > 256) && !bwd_pro) use = true, th = 32, bn = 64, Z = 1;  // (experiment: 4-wave 32 x 64 tiles)
    if (use) {
    p.flow = true;
    p.plane = false;
    p.one_per_cu = false;
    p.tile.th = th, p.tile.tw = fc.tw, p.tile.bn = bn;
    p.ksplit = Z;
    long fb = (npixf + 15) / 16;
    if (fb > 1024) fb = 1024;
    p.finish_blocks = (int)fb;
    p.partials = Z > 1 ? p.finish_blocks : cy_cdiv((long)N * H, th) * (W / fc.tw);
    p.ws_bytes = Z > 1 ? (size_t)Z * npixf * Cout * sizeof(float) : 0;
    return p;
    }
  }
  p.tile = choose_tile((long)N * H, W, Cout);
  const long npix = (long)N * H * W;
  if (p.plane) {
    p.tile.th = kPlaneTH, p.tile.tw = kPlaneTW;
    p.tile.bn = Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32);
    // 128-cout layers with fewer tiles than CUs (56x56 at N=16: 224 tiles): two 64-cout workgroups
    // per tile instead of one (10-13 % faster there, 10-50 % slower at 448 tiles)
    constexpr int bn64_below = 300;
    if (Cout == 128 && cy_cdiv((long)N * H, kPlaneTH) * (W / kPlaneTW) < bn64_below) p.tile.bn = 64;
  }
  const int tiles = cy_cdiv((long)N * H, p.tile.th) * cy_cdiv(W, p.tile.tw);
  const int blocks = tiles * cy_cdiv(Cout, p.tile.bn);
  const int kc = (p.tile.bn <= 64 ? 64 : 128) / elem_bytes;  // must match dispatch_conv's PITCHB
  const int ncc = cy_cdiv(Cin, kc);
  int Z = 1;
  if (blocks < 192 && ncc >= 2 && Cout % 8 == 0) {
    Z = cy_cdiv(384, blocks);
    if (Z > ncc) Z = ncc;
    if (Z > 8) Z = 8;
  }
  p.one_per_cu = false;
  if (p.plane && p.tile.bn == 128) {
    if (ncc >= 2) {  // aim at 192..288 workgroups, each with >= 2 chunks to pipeline
      int z1 = blocks >= 256 ? 1 : 256 / blocks;
      if (z1 > ncc / 2) z1 = ncc / 2;
      if (z1 < 1) z1 = 1;
      if (blocks * z1 >= 192 && blocks * z1 <= 288 && Cout % 8 == 0) {
        p.one_per_cu = true;
        Z = z1;
      }
    }
  }
  p.ksplit = Z;
  long fb = (npix + 15) / 16;  // split-K layers are small (<= 12.5k pixels): many short workgroups
  if (fb > 1024) fb = 1024;
  p.finish_blocks = (int)fb;
  p.partials = Z > 1 ? p.finish_blocks : tiles;
  p.ws_bytes = Z > 1 ? (size_t)Z * npix * Cout * sizeof(float) : 0;
  return p;
}

template <typename TO>
int launch_finish(const ConvArgs& a, const ConvPlan& p, hipStream_t st) {
  const int G = a.Cout / 8;
  const int gpp = G < 256 ? G : 256;
  const int rows = 256 / gpp;
  const size_t smem = (size_t)2 * rows * gpp * 8 * sizeof(float);
  hipLaunchKernelGGL(conv_splitk_finish_kernel<TO>, dim3(p.finish_blocks), dim3(256), smem, st, a.ws,
                     p.ksplit, (long)a.NH * a.W, a.Cout, (TO*)a.out, a.ldo, (TO*)a.out2, a.ldo2,
                     a.split_c, a.stats, a.sacc, a.sR);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

template <typename T>
int dispatch_conv(const ConvArgs& a, const ConvPlan& p, hipStream_t st) {
  if (p.stream) {
    if constexpr (sizeof(T) == 2) return dispatch_conv_stream<T>(a, st);
    return CY_ERR_DTYPE;
  }
  if (p.flow) {
    if constexpr (sizeof(T) == 2) return dispatch_conv_flow<T>(a, p.tile.th, p.tile.bn, p.tile.tw, st);
    return CY_ERR_DTYPE;
  }
  if (p.plane) {
    if (p.tile.bn == 128)
      return p.one_per_cu ? launch_conv_plane<T, kPlaneTH, 128, 2, 2, 128, false, true>(a, st)
                          : launch_conv_plane<T, kPlaneTH, 128, 2, 2, 128, false, false>(a, st);
    if (p.tile.bn == 64) return launch_conv_plane<T, kPlaneTH, 64, 4, 1, 64, true>(a, st);
    return launch_conv_plane<T, kPlaneTH, 32, 4, 1, 64, true>(a, st);
  }
  const TileChoice c = p.tile;
  // BN <= 64 (small-channel, HBM-leaning layers): 64-byte channel chunks, all nine weight taps
  // resident in LDS.  BN = 128: 128-byte chunks, weights through a 2-deep per-tap ring.
#define CY_CONV_CASE(TH_, TW_, BN_, WGM_, WGN_, P_, ALLT_) \
  return launch_conv<T, T, TH_, TW_, BN_, WGM_, WGN_, P_, ALLT_>(a, st)
  if (c.th == 8 && c.tw == 32) {
    if (c.bn == 128) CY_CONV_CASE(8, 32, 128, 2, 2, 128, false);
    if (c.bn == 64) CY_CONV_CASE(8, 32, 64, 4, 1, 64, true);
    CY_CONV_CASE(8, 32, 32, 4, 1, 64, true);
  }
  if (c.th == 16 && c.tw == 16) {
    if (c.bn == 128) CY_CONV_CASE(16, 16, 128, 2, 2, 128, false);
    if (c.bn == 64) CY_CONV_CASE(16, 16, 64, 4, 1, 64, true);
    CY_CONV_CASE(16, 16, 32, 4, 1, 64, true);
  }
  if (c.th == 32 && c.tw == 8) {
    if (c.bn == 128) CY_CONV_CASE(32, 8, 128, 2, 2, 128, false);
    if (c.bn == 64) CY_CONV_CASE(32, 8, 64, 4, 1, 64, true);
    CY_CONV_CASE(32, 8, 32, 4, 1, 64, true);
  }
  if (c.th == 8 && c.tw == 28) CY_CONV_CASE(8, 28, 128, 1, 4, 128, false);
  if (c.th == 16 && c.tw == 14) CY_CONV_CASE(16, 14, 128, 1, 4, 128, false);
#undef CY_CONV_CASE
  return CY_ERR_SHAPE;
}

// ---------------------------------------------------------------------------
// weight repacking:  w[Cout][Cin][3][3] f32  ->  wf[tap][co_pad][ci_pad] (T)
//                                            ->  wd[tap][ci_pad2][co_pad2] (T), flipped taps
// The stage-contiguous images of cy_conv_flow.h follow the [tap][co][ci] images in the same buffers:
//   wf + 9*co_pad*ci_pad :  [Cout/64][Cin/16][tap][2 planes][64 rows = couts][8 channels]    (if Cout % 64 == 0, Cin % 16 == 0)
//   wd + 9*ci_pad2*co_pad2: [Cin/64][Cout/16][8-tap][2 planes][64 rows = cins][8 couts]      (if Cin % 64 == 0, Cout % 16 == 0)
__host__ __device__ inline long flow_elem_index(int row, int k, int tap, int nchunk) {
  return ((((long)(row >> 6) * nchunk + (k >> 4)) * 9 + tap) * 2 + ((k >> 3) & 1)) * 512 + (row & 63) * 8 + (k & 7);
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                    T* __restrict__ wd, int Cout, int Cin, int co_pad, int ci_pad,
                                    int ci_pad2, int co_pad2, long ff, long fd) {
  const long nf = 9L * co_pad * ci_pad;
  const long nd = wd ? 9L * ci_pad2 * co_pad2 : 0;
  if (!wd) fd = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nf + nd + ff + fd;
       i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = (int)(i % ci_pad);
      const int co = (int)((i / ci_pad) % co_pad);
      const int tap = (int)(i / ((long)ci_pad * co_pad));
      float v = 0.f;
      if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * 9 + tap];
      wf[i] = from_f32<T>(v);
    } else if (i < nf + nd) {
      const long j = i - nf;
      const int co = (int)(j % co_pad2);
      const int ci = (int)((j / co_pad2) % ci_pad2);
      const int tap = (int)(j / ((long)co_pad2 * ci_pad2));
      float v = 0.f;
      if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * 9 + (8 - tap)];
      wd[j] = from_f32<T>(v);
    } else if (i < nf + nd + ff) {
      const long j = i - nf - nd;  // (co, ci, tap) enumerated in source order
      const int tap = (int)(j % 9), ci = (int)((j / 9) % Cin), co = (int)(j / (9L * Cin));
      wf[nf + flow_elem_index(co, ci, tap, Cin / 16)] = from_f32<T>(w[j]);
    } else {
      const long j = i - nf - nd - ff;
      const int tap = (int)(j % 9), ci = (int)((j / 9) % Cin), co = (int)(j / (9L * Cin));
      wd[nd + flow_elem_index(ci, co, 8 - tap, Cout / 16)] = from_f32<T>(w[j]);
    }
  }
}

// every layer of the network in one launch: `items` (device) lists the layers; the unit of work is a
// 32 (co) x 32 (ci) tile of one layer (`first` = index of the layer's first tile), transposed
// through LDS so that the f32 source is read in contiguous runs of 288 floats and both images are
// written in contiguous runs of 32 elements
__host__ __device__ inline int pack_tiles_co(int co_pad, int co_pad2) {
  return ((co_pad > co_pad2 ? co_pad : co_pad2) + 31) / 32;
}
__host__ __device__ inline int pack_tiles_ci(int ci_pad, int ci_pad2) {
  return ((ci_pad > ci_pad2 ? ci_pad : ci_pad2) + 31) / 32;
}

template <typename T>
__global__ void __launch_bounds__(256)
    pack_weights_batched_kernel(const cy_pack_item* __restrict__ items, int n, T* __restrict__ wf_arena,
                                T* __restrict__ wd_arena) {
  __shared__ T tile[9][32][33];
  __shared__ cy_pack_item sit;
  __shared__ int s_local;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last item with first <= blockIdx.x
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].first <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sit = items[lo];
    s_local = (int)(blockIdx.x - items[lo].first);
  }
  __syncthreads();
  const cy_pack_item& it = sit;
  const int tci_n = pack_tiles_ci(it.ci_pad, it.ci_pad2);
  const int co0 = (s_local / tci_n) * 32, ci0 = (s_local % tci_n) * 32;
  const int ncol = (it.Cin - ci0 < 32 ? it.Cin - ci0 : 32) * 9;  // valid floats per source row (may be <= 0)
  for (int idx = tid; idx < 32 * 288; idx += 256) {
    const int row = idx / 288, col = idx - row * 288;
    float v = 0.f;
    if (co0 + row < it.Cout && col < ncol) v = it.w[((size_t)(co0 + row) * it.Cin + ci0) * 9 + col];
    tile[col % 9][row][col / 9] = from_f32<T>(v);
  }
  __syncthreads();
  for (int idx = tid; idx < 9 * 1024; idx += 256) {
    const int tap = idx >> 10, hi5 = (idx >> 5) & 31, lo5 = idx & 31;
    // forward image [tap][co][ci]: ci fastest
    if (co0 + hi5 < it.co_pad && ci0 + lo5 < it.ci_pad)
      wf_arena[it.off_f + ((size_t)tap * it.co_pad + co0 + hi5) * it.ci_pad + ci0 + lo5] = tile[tap][hi5][lo5];
    // data-gradient image [8-tap][ci][co]: co fastest
    if (ci0 + hi5 < it.ci_pad2 && co0 + lo5 < it.co_pad2)
      wd_arena[it.off_d + ((size_t)(8 - tap) * it.ci_pad2 + ci0 + hi5) * it.co_pad2 + co0 + lo5] =
          tile[tap][lo5][hi5];
  }
  // stage-contiguous images (cy_conv_flow.h): 16-byte items (row, group of 8 k), 9 x 32 x 4 per tile and image
  if (it.off_ff >= 0 || it.off_fd >= 0) {
    for (int idx = tid; idx < 9 * 128; idx += 256) {
      const int tap = idx >> 7, row = (idx >> 2) & 31, g = idx & 3;
      if (it.off_ff >= 0 && co0 + row < it.Cout && ci0 + 8 * g < it.Cin) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[tap][row][8 * g + j];
        T* dst = wf_arena + it.off_ff + flow_elem_index(co0 + row, ci0 + 8 * g, tap, it.Cin / 16);
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(v);
      }
      if (it.off_fd >= 0 && ci0 + row < it.Cin && co0 + 8 * g < it.Cout) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[tap][8 * g + j][row];
        T* dst = wd_arena + it.off_fd + flow_elem_index(ci0 + row, co0 + 8 * g, 8 - tap, it.Cout / 16);
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(v);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// First layer: Cin in 1..4, image is f32 NCHW.  HBM-bound on the output write;
// plain VALU.  Thread = (pixel, group of 8 output channels).
template <typename TO>
__global__ void __launch_bounds__(256)
    conv3x3_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                         TO* __restrict__ out, float* __restrict__ stats, int N, int Cin, int H,
                         int W, int Cout, unsigned long long* sacc, int sR) {
  // a block walks a contiguous pixel range, 256 / (Cout/8) pixels per iteration; the BN statistics
  // are accumulated in registers over the whole range and reduced ONCE per block (one partial per
  // block: fixed order => deterministic)
  __shared__ float sw[9 * 4 * 64];  // [tap][ci][co], Cout <= 64
  __shared__ float sred[4 * 2 * 64];  // [wave][sum | sumsq][cout]
  extern __shared__ float sx[];        // [Cin][rows of this block's pixel range + 2][W + 2], zero left / right columns
  const int tid = threadIdx.x;
  for (int i = tid; i < 9 * Cin * Cout; i += 256) {
    const int co = i % Cout;
    const int ci = (i / Cout) % Cin;
    const int tap = i / (Cout * Cin);
    sw[i] = w[((size_t)co * Cin + ci) * 9 + tap];
  }
  __syncthreads();
  const int CG = Cout / 8;   // power of two (checked by the host)
  const int ppb = 256 / CG;  // pixels per iteration
  const int cg = tid % CG;
  const long npix = (long)N * H * W;
  const long per = ((npix + gridDim.x - 1) / gridDim.x + ppb - 1) / ppb * ppb;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  // The nine taps of a pixel are nine 4-byte global loads in the first version -- 0.9 M wave-level load
  // instructions per launch, four threads per pixel repeating them: the kernel ran at 1.4 TB/s of output.  The
  // input rows of the block's pixel range (+ one row above and below) are staged in LDS once, coalesced.
  const int W2 = W + 2;
  const int g0 = p0 < npix ? (int)(p0 / W) : 0;
  const int g1 = p1 > p0 ? (int)((p1 - 1) / W) : g0;
  const int nrows = g1 - g0 + 3;
  for (int ci = 0; ci < Cin; ++ci)
    for (int r = tid / 64; r < nrows; r += 4) {  // a wave per row
      const int g = g0 - 1 + r;
      const bool rok = g >= 0 && g < N * H;
      const int n_ = rok ? g / H : 0;
      const float* xr = x + (((size_t)n_ * Cin + ci) * H + (g - n_ * H)) * W;
      float* dst = sx + ((size_t)ci * nrows + r) * W2;
      for (int c = tid & 63; c < W2; c += 64) dst[c] = (rok && c >= 1 && c <= W) ? xr[c - 1] : 0.f;
    }
  __syncthreads();
  // pixel coordinates advance incrementally.  (Also tried on top of the LDS image, both slower or equal: this
  // thread's 72 weights in registers for Cin = 1 -- 55.7 us; two pixels per thread sharing every weight read --
  // 49.7 us; as below 46.9 us per launch on average over the two batch sizes, 56.1 before the LDS image.)
  long p = p0 + tid / CG;
  int wq, hq, n;
  {
    const unsigned pu = (unsigned)(p < npix ? p : 0);
    const unsigned row = pu / (unsigned)W;
    wq = (int)(pu - row * (unsigned)W);
    n = (int)(row / (unsigned)H);
    hq = (int)(row - (unsigned)n * (unsigned)H);
  }
  const int dq = ppb / W, dr = ppb - dq * W;  // ppb = dq rows + dr pixels
  for (; p < p1; p += ppb) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int rc = n * H + hq - g0 + 1;  // LDS row of this pixel
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xc = sx + ((size_t)ci * nrows + rc) * W2 + wq + 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const int hh = hq + dh;
        const float xv = (hh >= 0 && hh < H) ? xc[dh * W2 + dw] : 0.f;  // (left / right: the zero columns)
        const float* wr = sw + (tap * Cin + ci) * Cout + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
      }
    }
    TO* op = out + (size_t)p * Cout + cg * 8;
    if constexpr (sizeof(TO) == 2) {
      st16(op, Chunk<TO>::pack(acc));
    } else {
      st16(op, Chunk<float>::pack(acc));
      st16(reinterpret_cast<float*>(op) + 4, Chunk<float>::pack(acc + 4));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float q = round_through<TO>(acc[j]);
      s1[j] += q;
      s2[j] += q * q;
    }
    wq += dr;
    hq += dq;
    if (wq >= W) wq -= W, ++hq;
    while (hq >= H) hq -= H, ++n;
  }
  if (stats || sacc) {
    // lanes of a wave that share a cout group (xor-shuffles over the pixel bits), then the four waves
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a1 = s1[j], a2 = s2[j];
      for (int o = CG; o < 64; o <<= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
      }
      if (lane < CG) {
        sred[(wave * 2 + 0) * 64 + lane * 8 + j] = a1;
        sred[(wave * 2 + 1) * 64 + lane * 8 + j] = a2;
      }
    }
    __syncthreads();
    if (tid < 2 * Cout) {
      const int which = tid / Cout, co = tid % Cout;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += sred[(q * 2 + which) * 64 + co];
      if (sacc) bn_acc_add(sacc, sR, Cout, (int)blockIdx.x & (sR - 1), which, co, t);
      else stats[((size_t)blockIdx.x * 2 + which) * Cout + co] = t;
    }
  }
}

// Cin = 1, 16-bit storage, Cout <= 32: the nine taps are one k-step of the 32x32x16 MFMA (k = 9 taps + 7 zeros).
// The VALU kernel above spends ~110 instructions and 18 LDS weight reads per (pixel, 8 couts) -- 36 / 56 us at
// N = 16 / 32 against the 13 / 25 us the output write takes.  Here a wave computes 32 pixels x 32 couts per MFMA:
// rows = couts (the weights, in registers for the whole kernel), columns = pixels (the patch, eight LDS reads per
// lane), so that a lane owns one pixel and 16 couts and stores two 16-byte chunks straight from registers (the plane
// kernel's epilogue).  Image and weights are rounded to the storage type first -- what the reference's autocast
// convolution does with its inputs.  Same LDS image, block ranges and one statistics partial per block as above.
template <typename T>
__global__ void __launch_bounds__(256)
    conv3x3_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ out,
                              float* __restrict__ stats, int N, int H, int W, int Cout, unsigned long long* sacc, int sR) {
  using M = Mma<T>;
  __shared__ float sred[4 * 2 * 32];  // [wave][sum | sumsq][cout]
  extern __shared__ float sx[];        // [rows of this block's pixel range + 2][W + 2], zero left / right columns
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // A: weights of cout r, taps 8h .. 8h+7 (tap 8 is the only one of the upper half)
  typename M::Frag fa;
  {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 8 * h + j;
      v[j] = from_f32<T>((r < Cout && tap < 9) ? w[r * 9 + tap] : 0.f);
    }
    fa.v = *reinterpret_cast<const decltype(fa.v)*>(v);
  }
  const long npix = (long)N * H * W;
  const long per = ((npix + gridDim.x - 1) / gridDim.x + 127) / 128 * 128;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  const int W2 = W + 2;
  const int g0 = p0 < npix ? (int)(p0 / W) : 0;
  const int g1 = p1 > p0 ? (int)((p1 - 1) / W) : g0;
  const int nrows = g1 - g0 + 3;
  for (int rr = tid / 64; rr < nrows; rr += 4) {  // a wave per row
    const int g = g0 - 1 + rr;
    const bool rok = g >= 0 && g < N * H;
    const float* xr = x + (size_t)(rok ? g : 0) * W;  // (Cin = 1: image rows are consecutive)
    float* dst = sx + (size_t)rr * W2;
    for (int c = lane; c < W2; c += 64) dst[c] = (rok && c >= 1 && c <= W) ? xr[c - 1] : 0.f;
  }
  __syncthreads();
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
  long p = p0 + wave * 32 + r;
  int wq, hq, n;
  {
    const unsigned pu = (unsigned)(p < npix ? p : 0);
    const unsigned row = pu / (unsigned)W;
    wq = (int)(pu - row * (unsigned)W);
    n = (int)(row / (unsigned)H);
    hq = (int)(row - (unsigned)n * (unsigned)H);
  }
  const int dq = 128 / W, dr = 128 - dq * W;  // 128 pixels = dq rows + dr pixels
  for (long pb = p0 + wave * 32; pb < p1; pb += 128, p += 128) {  // (wave-uniform trip count)
    const bool valid = p < p1;
    const float* xc = sx + (size_t)(n * H + hq - g0 + 1) * W2 + wq + 1;
    typename M::Frag fb;
    {
      T v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int dh0 = j / 3 - 1, dw0 = j % 3 - 1;  // tap j (lower half); the upper half has tap 8 = (+1, +1) at j = 0
        const int dh = h ? 1 : dh0, dw = h ? 1 : dw0;
        const bool ok = valid && (h ? j == 0 : true) && hq + dh >= 0 && hq + dh < H;
        v[j] = from_f32<T>(ok ? xc[dh * W2 + dw] : 0.f);
      }
      fb.v = *reinterpret_cast<const decltype(fb.v)*>(v);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    M::mma(fa, fb, acc);  // rows = couts: lane (r = pixel, h) holds couts (i & 3) + 8 * (i >> 2) + 4 * h
    u32x2 packed[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      T pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pk[j] = from_f32<T>(acc[4 * g + j]);
        if (valid) {
          const float qv = to_f32<T>(pk[j]);
          s1[4 * g + j] += qv;
          s2[4 * g + j] += qv * qv;
        }
      }
      packed[g] = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
    }
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      u32x2 lo = packed[g], hi = packed[g + 1];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(lo[j], hi[j], false, false);
        lo[j] = sw[0];
        hi[j] = sw[1];
      }
      const int co = 8 * g + 8 * h;
      if (valid && co < Cout) st16(out + (size_t)p * Cout + co, u32x4{lo[0], lo[1], hi[0], hi[1]});
    }
    wq += dr;
    hq += dq;
    if (wq >= W) wq -= W, ++hq;
    while (hq >= H) hq -= H, ++n;
  }
  if (stats || sacc) {
    // reduce-scatter over the 32 lanes of each half (the plane kernel's): even lanes end up with one channel each
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool up = (lane & 16) != 0;
      const float snd1 = up ? s1[i] : s1[i + 8], snd2 = up ? s2[i] : s2[i + 8];
      const float kp1 = up ? s1[i + 8] : s1[i], kp2 = up ? s2[i + 8] : s2[i];
      s1[i] = kp1 + __shfl_xor(snd1, 16, 64);
      s2[i] = kp2 + __shfl_xor(snd2, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool up = (lane & 8) != 0;
      const float snd1 = up ? s1[i] : s1[i + 4], snd2 = up ? s2[i] : s2[i + 4];
      const float kp1 = up ? s1[i + 4] : s1[i], kp2 = up ? s2[i + 4] : s2[i];
      s1[i] = kp1 + __shfl_xor(snd1, 8, 64);
      s2[i] = kp2 + __shfl_xor(snd2, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool up = (lane & 4) != 0;
      const float snd1 = up ? s1[i] : s1[i + 2], snd2 = up ? s2[i] : s2[i + 2];
      const float kp1 = up ? s1[i + 2] : s1[i], kp2 = up ? s2[i + 2] : s2[i];
      s1[i] = kp1 + __shfl_xor(snd1, 4, 64);
      s2[i] = kp2 + __shfl_xor(snd2, 4, 64);
    }
    {
      const bool up = (lane & 2) != 0;
      const float snd1 = up ? s1[0] : s1[1], snd2 = up ? s2[0] : s2[1];
      const float kp1 = up ? s1[1] : s1[0], kp2 = up ? s2[1] : s2[0];
      s1[0] = kp1 + __shfl_xor(snd1, 2, 64);
      s2[0] = kp2 + __shfl_xor(snd2, 2, 64);
    }
    s1[0] += __shfl_xor(s1[0], 1, 64);
    s2[0] += __shfl_xor(s2[0], 1, 64);
    const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int col = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if ((lane & 1) == 0) {
      sred[(wave * 2 + 0) * 32 + col] = s1[0];
      sred[(wave * 2 + 1) * 32 + col] = s2[0];
    }
    __syncthreads();
    if (tid < 2 * Cout) {
      const int which = tid / Cout, co = tid % Cout;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += sred[(q * 2 + which) * 32 + co];
      if (sacc) bn_acc_add(sacc, sR, Cout, (int)blockIdx.x & (sR - 1), which, co, t);
      else stats[((size_t)blockIdx.x * 2 + which) * Cout + co] = t;
    }
  }
}

int first_conv_blocks(long npix, int Cout, int W) {
  const int ppb = 256 / (Cout / 8);
  long b = (npix + ppb - 1) / ppb;
  if (b > 2048) b = 2048;  // 8 blocks per CU: a few hundred pixels each at the U-Net's sizes (512 ... 4096: the same time)
  // ... but never more image rows per block than its LDS image holds (96 KB, up to four input channels)
  const long rows_max = (96 * 1024 / 4) / (4 * (W + 2)) - 4;
  const long need = (npix + rows_max * W - 1) / (rows_max > 0 ? rows_max * W : 1);
  if (rows_max < 1) return -1;
  if (b < need) b = need;
  return (int)b;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int cy_abi_version(void) { return 12; }
const char* cy_build_arch(void) { return "gfx950"; }

unsigned long long cy_stream_capture_id(void* stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) return 0;
  return status == hipStreamCaptureStatusActive ? (id ? id : 1ull) : 0ull;
}

int cy_conv3x3_packed_dims(int Cout, int Cin, int* co_pad, int* ci_pad) {
  if (Cout <= 0 || Cin <= 0) return CY_ERR_ARG;
  if (co_pad) *co_pad = cy_roundup(Cout, 128);
  if (ci_pad) *ci_pad = cy_roundup(Cin, 64);
  return CY_OK;
}

long long cy_conv3x3_packed_elems(int Cout, int Cin, int dtype) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const long plane = 9L * cy_roundup(Cout, 128) * cy_roundup(Cin, 64);
  return plane + (dtype == CY_F32 ? 0 : flow_image_elems(Cout, Cin));
}

int cy_conv3x3_pack_weights(const float* w, void* wf, void* wd, int Cout, int Cin, int dtype,
                            void* stream) {
  if (!w || !wf) return CY_ERR_ARG;
  int co_pad, ci_pad, ci_pad2, co_pad2;
  cy_conv3x3_packed_dims(Cout, Cin, &co_pad, &ci_pad);
  cy_conv3x3_packed_dims(Cin, Cout, &ci_pad2, &co_pad2);  // dgrad: roles swapped
  hipStream_t st = (hipStream_t)stream;
  const long ff = dtype == CY_F32 ? 0 : flow_image_elems(Cout, Cin), fd = dtype == CY_F32 ? 0 : flow_image_elems(Cin, Cout);
  const long total = 9L * co_pad * ci_pad + ff + (wd ? 9L * ci_pad2 * co_pad2 + fd : 0);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, (bf16*)wf,
                       (bf16*)wd, Cout, Cin, co_pad, ci_pad, ci_pad2, co_pad2, ff, fd);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(pack_weights_kernel<f16>, dim3(blocks), dim3(256), 0, st, w, (f16*)wf,
                       (f16*)wd, Cout, Cin, co_pad, ci_pad, ci_pad2, co_pad2, ff, fd);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)wf,
                       (float*)wd, Cout, Cin, co_pad, ci_pad, ci_pad2, co_pad2, 0L, 0L);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_conv3x3_pack_weights_batched(const cy_pack_item* items, int n_items, long long total,
                                    void* wf_arena, void* wd_arena, int dtype, void* stream) {
  if (!items || n_items <= 0 || n_items > 64 || total <= 0 || total > 0x7fffffffLL || !wf_arena || !wd_arena)
    return CY_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CY_BF16)
    hipLaunchKernelGGL(pack_weights_batched_kernel<bf16>, dim3((unsigned)total), dim3(256), 0, st, items,
                       n_items, (bf16*)wf_arena, (bf16*)wd_arena);
  else if (dtype == CY_F16)
    hipLaunchKernelGGL(pack_weights_batched_kernel<f16>, dim3((unsigned)total), dim3(256), 0, st, items,
                       n_items, (f16*)wf_arena, (f16*)wd_arena);
  else if (dtype == CY_F32)
    hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3((unsigned)total), dim3(256), 0, st, items,
                       n_items, (float*)wf_arena, (float*)wd_arena);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

static int conv_check(const cy_conv_desc* d) {
  if (!d) return CY_ERR_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0)
    return CY_ERR_SHAPE;
  if (d->in_dtype != d->out_dtype) return CY_ERR_DTYPE;
  if (d->in_dtype != CY_F32 && d->in_dtype != CY_BF16 && d->in_dtype != CY_F16) return CY_ERR_DTYPE;
  const int epc = d->in_dtype == CY_F32 ? 4 : 8;
  if (d->C1 % epc || d->C2 % epc || d->Cout % epc) return CY_ERR_SHAPE;
  if (d->ld1 % epc || (d->C2 && d->ld2 % epc) || d->ldo % epc) return CY_ERR_SHAPE;
  if (d->ld1 < d->C1 || (d->C2 && d->ld2 < d->C2)) return CY_ERR_SHAPE;
  if (d->mode1 == CY_SRC_UP2 && ((d->H & 1) || (d->W & 1))) return CY_ERR_SHAPE;
  if (d->mode1 < 0 || d->mode1 > 2) return CY_ERR_ARG;
  if (d->prologue && d->C2) return CY_ERR_ARG;
  if (d->prologue < 0 || d->prologue > 2 || (d->prologue == 2 && d->mode1 != CY_SRC_DIRECT)) return CY_ERR_ARG;
  if (d->split_c > 0 && (d->split_c % epc || d->ldo2 % epc)) return CY_ERR_SHAPE;
  return CY_OK;
}

static ConvPlan plan_of(const cy_conv_desc* d) {
  return plan_conv(d->N, d->H, d->W, d->C1 + d->C2, d->Cout, d->in_dtype == CY_F32 ? 4 : 2, stream_applicable(d), d->prologue != 0,
                   flow_choice(d), d->prologue == 2);
}

int cy_conv3x3_num_partials(const cy_conv_desc* d) {
  if (conv_check(d) != CY_OK) return CY_ERR_ARG;
  return plan_of(d).partials;
}

int cy_conv3x3_plan(const cy_conv_desc* d, cy_conv_plan* plan) {
  const int rc = conv_check(d);
  if (rc != CY_OK) return rc;
  if (!plan) return CY_ERR_ARG;
  const ConvPlan p = plan_of(d);
  plan->kernel = p.flow ? 5 : (p.stream ? 4 : (p.plane ? 1 : 0));
  plan->th = p.tile.th, plan->tw = p.tile.tw, plan->bn = p.tile.bn;
  plan->ksplit = p.ksplit, plan->one_per_cu = p.one_per_cu ? 1 : 0, plan->partials = p.partials;
  plan->workgroups = cy_cdiv((long)d->N * d->H, p.tile.th) * cy_cdiv(d->W, p.tile.tw) *
                     cy_cdiv(d->Cout, p.tile.bn) * p.ksplit;
  return CY_OK;
}

size_t cy_conv3x3_fwd_ws_bytes(const cy_conv_desc* d) {
  if (conv_check(d) != CY_OK) return 0;
  return plan_of(d).ws_bytes;
}

// workgroups that add into one channel's sums (the streaming kernel sums its wave rows in LDS first)
static int stat_workgroups_of(const cy_conv_desc* d, const ConvPlan& p) {
  if (p.stream) return stream_grid(d->C1 + d->C2, d->Cout, cy_cdiv((long)d->N * d->H, kPlaneTH) * (d->W / kPlaneTW), d->prologue != 0);
  return p.partials;
}

int cy_conv3x3_stat_workgroups(const cy_conv_desc* d) {
  if (conv_check(d) != CY_OK) return CY_ERR_ARG;
  return stat_workgroups_of(d, plan_of(d));
}

int cy_bn_acc_replicas(int C, int workgroups) { return (C <= 0 || workgroups <= 0) ? CY_ERR_ARG : bn_acc_replicas(C, workgroups); }
size_t cy_bn_acc_bytes(int C, int R) { return (C <= 0 || R <= 0) ? 0 : bn_acc_elems(C, R) * 8; }

int cy_conv3x3_fwd(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                   const float* shift, const void* w_packed, void* out, void* out2, float* stats,
                   void* ws, size_t ws_bytes, void* stream) {
  return cy_conv3x3_fwd_bn(d, src1, src2, nullptr, scale, shift, w_packed, out, out2, stats, nullptr, ws, ws_bytes, stream);
}

static int conv_fwd_impl(const cy_conv_desc* d, const void* src1, const void* src2, const cy_bn_fold* in_fold,
                         const float* scale, const float* shift, const cy_bn_bwd_in* bwd, const void* w_packed, void* out,
                         void* out2, float* stats, const cy_bn_acc* out_acc, void* ws, size_t ws_bytes, void* stream,
                         const cy_bn_dz_out* dz = nullptr);

int cy_conv3x3_fwd_bn(const cy_conv_desc* d, const void* src1, const void* src2, const cy_bn_fold* in_fold,
                      const float* scale, const float* shift, const void* w_packed, void* out, void* out2,
                      float* stats, const cy_bn_acc* out_acc, void* ws, size_t ws_bytes, void* stream) {
  if (d && d->prologue == 2) return CY_ERR_ARG;
  return conv_fwd_impl(d, src1, src2, in_fold, scale, shift, nullptr, w_packed, out, out2, stats, out_acc, ws, ws_bytes, stream);
}

// the data gradient with the BatchNorm + ReLU backward in its load path: only the flow kernel's tilings that have room
// for the y buffer take it
static bool dgrad_bn_plan_ok(const cy_conv_desc* d, const ConvPlan& p) {
  if (!p.flow || d->in_dtype == CY_F32 || d->C1 > 512) return false;
  if (p.tile.bn == 64 && p.tile.th <= 32 && d->C1 > 256) return false;  // four-wave tilings: one coefficient set per thread
  // Every cout block of a tile (and every tile, for its halo) repeats the in-LDS pass over (dA, y): with 64-cout tiles a
  // 256-channel data gradient does it four times over, and the fused launch is slower than the two it replaces (28 x 28,
  // 256 -> 256 at N = 16: 46.8 against 12.5 + 21.3 us; tools/bench_dgrad_bn.py).  It pays on the 128-cout tilings
  // (56 x 56, 128 -> 128: 32.8 against 17.3 + 20.5).  CY_DGRAD_BN_ALL=1 lifts the rule (measurements).
  static const int all = [] { const char* e = getenv("CY_DGRAD_BN_ALL"); return e ? atoi(e) : 0; }();
  if (!all && p.tile.bn != 128) return false;
  return d->in_dtype == CY_BF16 ? flow_bwd_ok<bf16>(p.tile.th, p.tile.bn, p.tile.tw) : flow_bwd_ok<f16>(p.tile.th, p.tile.bn, p.tile.tw);
}

int cy_conv3x3_dgrad_bn_ok(const cy_conv_desc* d) {
  if (conv_check(d) != CY_OK || d->prologue != 2 || d->C2) return 0;
  return dgrad_bn_plan_ok(d, plan_of(d)) ? 1 : 0;
}

int cy_conv3x3_dgrad_bn(const cy_conv_desc* d, const void* dA, const cy_bn_bwd_in* bn, const void* w_packed, void* out,
                        void* out2, void* ws, size_t ws_bytes, void* stream) {
  if (!d || d->prologue != 2 || !bn || !bn->y || !bn->coef || !bn->acc || !bn->acc->acc || !bn->dy || bn->count <= 0) return CY_ERR_ARG;
  if (bn->acc->C != d->C1 || bn->acc->R < 1 || (bn->acc->R & (bn->acc->R - 1))) return CY_ERR_ARG;
  return conv_fwd_impl(d, dA, nullptr, nullptr, nullptr, nullptr, bn, w_packed, out, out2, nullptr, nullptr, ws, ws_bytes, stream);
}

// the epilogue can take the backward sums of the BatchNorm behind its output: flow kernel, no split-K, the channel range
// aligned with the cout blocks (all couts, or the second part of a split output)
static bool dgrad_dz_plan_ok(const cy_conv_desc* d, const ConvPlan& p, int c0, int Cc) {
  if (!p.flow || p.ksplit != 1 || d->in_dtype == CY_F32 || d->prologue) return false;
  // (16-row tiles only: with four fragments per wave and cout block the sums' working set -- 32 partial sums, 16 registers of
  //  y, 16 of coefficients beside 128 accumulators -- does not fit, the compiler spills ~900 registers in the epilogue and
  //  the launch takes 3-4 x as long: tools/bench_dgrad_dz.py)
  if (p.tile.th != 16) return false;
  if (c0 % p.tile.bn || Cc % p.tile.bn || Cc <= 0) return false;
  return c0 == 0 ? (Cc == d->Cout && d->split_c == 0) : (d->split_c == c0 && Cc == d->Cout - c0);
}

int cy_conv3x3_dgrad_dz_ok(const cy_conv_desc* d, int c0, int C) {
  if (conv_check(d) != CY_OK) return 0;
  return dgrad_dz_plan_ok(d, plan_of(d), c0, C) ? 1 : 0;
}

int cy_conv3x3_dgrad_dz(const cy_conv_desc* d, const void* dy, const void* w_packed, void* out, void* out2,
                        const cy_bn_dz_out* dz, void* ws, size_t ws_bytes, void* stream) {
  if (!dz || !dz->y || !dz->coef || !dz->acc || !dz->acc->acc || dz->acc->C != dz->C || dz->acc->R < 1 ||
      (dz->acc->R & (dz->acc->R - 1)))
    return CY_ERR_ARG;
  return conv_fwd_impl(d, dy, nullptr, nullptr, nullptr, nullptr, nullptr, w_packed, out, out2, nullptr, nullptr, ws, ws_bytes,
                       stream, dz);
}

static int conv_fwd_impl(const cy_conv_desc* d, const void* src1, const void* src2, const cy_bn_fold* in_fold,
                         const float* scale, const float* shift, const cy_bn_bwd_in* bwd, const void* w_packed, void* out,
                         void* out2, float* stats, const cy_bn_acc* out_acc, void* ws, size_t ws_bytes, void* stream,
                         const cy_bn_dz_out* dz) {
  int rc = conv_check(d);
  if (rc != CY_OK) return rc;
  if (!src1 || !w_packed || !out) return CY_ERR_ARG;
  if (d->C2 && !src2) return CY_ERR_ARG;
  if ((d->prologue == 2) != (bwd != nullptr)) return CY_ERR_ARG;
  if (d->prologue == 1 && !in_fold && (!scale || !shift)) return CY_ERR_ARG;
  if (in_fold && (!d->prologue || !in_fold->acc || !in_fold->coef || in_fold->C != d->C1 || in_fold->R < 1 ||
                  (in_fold->R & (in_fold->R - 1)) || in_fold->count <= 0))
    return CY_ERR_ARG;
  if (out_acc && (stats || !out_acc->acc || out_acc->C != d->Cout || out_acc->R < 1 || (out_acc->R & (out_acc->R - 1))))
    return CY_ERR_ARG;
  if (d->split_c > 0 && !out2) return CY_ERR_ARG;
  ConvArgs a = {};
  a.src1 = src1, a.src2 = src2, a.scale = scale, a.shift = shift, a.w = w_packed;
  a.out = out, a.out2 = out2, a.stats = stats;
  if (out_acc) a.sacc = (unsigned long long*)out_acc->acc, a.sR = out_acc->R;
  a.N = d->N, a.H = d->H, a.W = d->W, a.NH = d->N * d->H;
  a.C1 = d->C1, a.C2 = d->C2, a.Cout = d->Cout;
  a.mode1 = d->mode1, a.prologue = d->prologue;
  a.ld1 = d->ld1, a.ld2 = d->ld2, a.ldo = d->ldo, a.ldo2 = d->ldo2, a.split_c = d->split_c;
  a.tiles_w = 0, a.full_tiles = 0;
  {  // extents of the sources (the last pixel's row ends at ld * (pixels - 1) + C)
    const long eb = d->in_dtype == CY_F32 ? 4 : 2;
    const long px1 = d->mode1 == CY_SRC_POOL2 ? 4L * d->N * d->H * d->W : (d->mode1 == CY_SRC_UP2 ? (long)d->N * (d->H / 2) * (d->W / 2) : (long)d->N * d->H * d->W);
    a.bytes1 = ((px1 - 1) * d->ld1 + d->C1) * eb;
    a.bytes2 = d->C2 ? (((long)d->N * d->H * d->W - 1) * d->ld2 + d->C2) * eb : 0;
    const long opx = (long)d->N * d->H * d->W;
    const long ob = d->out_dtype == CY_F32 ? 4 : 2;
    a.bytes_o1 = ((opx - 1) * d->ldo + (d->split_c > 0 ? d->split_c : d->Cout)) * ob;
    a.bytes_o2 = d->split_c > 0 ? ((opx - 1) * d->ldo2 + (d->Cout - d->split_c)) * ob : 0;
  }
  cy_conv3x3_packed_dims(d->Cout, d->C1 + d->C2, &a.w_co_pad, &a.w_ci_pad);
  {
    const long fe = d->in_dtype == CY_F32 ? 0 : flow_image_elems(d->Cout, d->C1 + d->C2);
    a.wflow = fe ? (const unsigned char*)w_packed + 9L * a.w_co_pad * a.w_ci_pad * 2 : nullptr;
    a.bytes_w = fe * 2;
  }
  const ConvPlan p = plan_of(d);
  a.bytes_st = stats ? (long long)p.partials * 2 * d->Cout * 4 : 0;
  a.ksplit = p.ksplit;
  a.ws = (float*)ws;
  if (p.ksplit > 1 && (!ws || ws_bytes < p.ws_bytes)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (dz) {
    if (!dgrad_dz_plan_ok(d, p, dz->c0, dz->C)) return CY_ERR_SHAPE;
    a.dz_y = dz->y, a.dz_coef = dz->coef, a.dz_acc = (unsigned long long*)dz->acc->acc, a.dz_R = dz->acc->R;
    a.dz_ld = dz->C, a.dz_c0 = dz->c0, a.dz_C = dz->C;
  }
  if (bwd) {
    if (!dgrad_bn_plan_ok(d, p)) return CY_ERR_SHAPE;
    a.ysrc = bwd->y, a.bytes_y = a.bytes1;
    a.dy_out = bwd->dy, a.bytes_dy = a.bytes1;
    a.bfold.acc = (const unsigned long long*)bwd->acc->acc, a.bfold.R = bwd->acc->R, a.bfold.C = d->C1;
    a.bfold.coef = bwd->coef, a.bfold.inv_count = 1.0 / bwd->count, a.bfold.batch_stats = bwd->batch_stats;
    a.bfold.accumulate = bwd->accumulate, a.bfold.dgamma = bwd->dgamma, a.bfold.dbeta = bwd->dbeta;
  }
  if (in_fold) {
    // the flow and streaming kernels derive the coefficients in place; the register-staged kernels (plane, igemm:
    // every thread fetches its channels' pairs per chunk) take them from memory after a fold launch
    static const int fold_in_kernel = [] {  // (CY_BN_FOLD_IN_KERNEL=0: always the separate launch, A/B runs)
      const char* e = getenv("CY_BN_FOLD_IN_KERNEL");
      return e ? atoi(e) : 1;
    }();
    if ((p.flow || p.stream) && fold_in_kernel) {
      a.fold = bn_fold_from_abi(in_fold);
    } else {
      rc = cy_bn_fold_coef(in_fold, stream);
      if (rc != CY_OK) return rc;
      a.scale = in_fold->coef, a.shift = in_fold->coef + in_fold->C;
    }
  }
  rc = d->in_dtype == CY_BF16 ? dispatch_conv<bf16>(a, p, st)
       : d->in_dtype == CY_F16 ? dispatch_conv<f16>(a, p, st) : dispatch_conv<float>(a, p, st);
  if (rc != CY_OK || p.ksplit == 1) return rc;
  return d->in_dtype == CY_BF16 ? launch_finish<bf16>(a, p, st)
         : d->in_dtype == CY_F16 ? launch_finish<f16>(a, p, st) : launch_finish<float>(a, p, st);
}

// development aid: shader-clock stamps of workgroup 0 of the streaming kernel (-DCY_STREAM_STAMPS, tools/stream_stamps.py)
int cy_debug_conv_stamps(unsigned long long* dev_buf) {
  g_conv_stamp_buf = dev_buf;
  return CY_OK;
}

int cy_conv3x3_first_num_partials(int N, int H, int W, int Cout) {
  if (Cout % 8 || Cout > 64 || 256 % (Cout / 8)) return CY_ERR_SHAPE;
  const int b = first_conv_blocks((long)N * H * W, Cout, W);
  return b < 0 ? CY_ERR_SHAPE : b;
}

static int first_fwd_impl(const float* x, const float* w, void* out, float* stats, unsigned long long* sacc, int sR, int N,
                          int Cin, int H, int W, int Cout, int out_dtype, void* stream);

int cy_conv3x3_first_fwd(const float* x, const float* w, void* out, float* stats, int N, int Cin,
                         int H, int W, int Cout, int out_dtype, void* stream) {
  return first_fwd_impl(x, w, out, stats, nullptr, 0, N, Cin, H, W, Cout, out_dtype, stream);
}

int cy_conv3x3_first_fwd_acc(const float* x, const float* w, void* out, const cy_bn_acc* out_acc, int N, int Cin,
                             int H, int W, int Cout, int out_dtype, void* stream) {
  if (!out_acc || !out_acc->acc || out_acc->C != Cout || out_acc->R < 1 || (out_acc->R & (out_acc->R - 1))) return CY_ERR_ARG;
  return first_fwd_impl(x, w, out, nullptr, (unsigned long long*)out_acc->acc, out_acc->R, N, Cin, H, W, Cout, out_dtype, stream);
}

static int first_fwd_impl(const float* x, const float* w, void* out, float* stats, unsigned long long* sacc, int sR, int N,
                          int Cin, int H, int W, int Cout, int out_dtype, void* stream) {
  if (!x || !w || !out) return CY_ERR_ARG;
  if (Cin < 1 || Cin > 4) return CY_ERR_SHAPE;
  if ((long)N * H * W >= (1L << 31)) return CY_ERR_SHAPE;  // the kernel indexes pixels in 32 bits
  const int np = cy_conv3x3_first_num_partials(N, H, W, Cout);
  if (np < 0) return np;
  hipStream_t st = (hipStream_t)stream;
  // LDS image of the input rows a block touches (the kernel's own range arithmetic, upper bound)
  const long npix = (long)N * H * W;
  const int ppb = 256 / (Cout / 8);
  const long per = ((npix + np - 1) / np + ppb - 1) / ppb * ppb;
  const size_t smem = (size_t)Cin * (per / W + 4) * (W + 2) * sizeof(float);
  if (smem > 96 * 1024) return CY_ERR_SHAPE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_first_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_first_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_first_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  static const int first_mfma = [] {  // (CY_FIRST_MFMA=0: the VALU kernel, for A/B runs)
    const char* e = getenv("CY_FIRST_MFMA");
    return e ? atoi(e) : 1;
  }();
  if (first_mfma && Cin == 1 && Cout <= 32 && (out_dtype == CY_BF16 || out_dtype == CY_F16)) {
    const long per_m = ((npix + np - 1) / np + 127) / 128 * 128;
    const size_t smem_m = (size_t)(per_m / W + 4) * (W + 2) * sizeof(float);
    if (smem_m <= 96 * 1024) {
      static bool attr_m = false;
      if (!attr_m) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_first_mfma_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_first_mfma_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
          return CY_ERR_LAUNCH;
        attr_m = true;
      }
      if (out_dtype == CY_BF16)
        hipLaunchKernelGGL(conv3x3_first_mfma_kernel<bf16>, dim3(np), dim3(256), smem_m, st, x, w, (bf16*)out, stats, N, H, W, Cout, sacc, sR);
      else
        hipLaunchKernelGGL(conv3x3_first_mfma_kernel<f16>, dim3(np), dim3(256), smem_m, st, x, w, (f16*)out, stats, N, H, W, Cout, sacc, sR);
      CY_CHECK_LAUNCH();
      return CY_OK;
    }
  }
  if (out_dtype == CY_BF16)
    hipLaunchKernelGGL(conv3x3_first_kernel<bf16>, dim3(np), dim3(256), smem, st, x, w, (bf16*)out,
                       stats, N, Cin, H, W, Cout, sacc, sR);
  else if (out_dtype == CY_F16)
    hipLaunchKernelGGL(conv3x3_first_kernel<f16>, dim3(np), dim3(256), smem, st, x, w, (f16*)out,
                       stats, N, Cin, H, W, Cout, sacc, sR);
  else if (out_dtype == CY_F32)
    hipLaunchKernelGGL(conv3x3_first_kernel<float>, dim3(np), dim3(256), smem, st, x, w, (float*)out,
                       stats, N, Cin, H, W, Cout, sacc, sR);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
