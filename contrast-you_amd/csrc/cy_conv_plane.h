// "Plane" variant of the 3x3 implicit-GEMM kernel (forward / data gradient), used when the image
// width is a multiple of 14 -- which every level of the 224x224 U-Net is (224, 112, 56, 28, 14).
//
// Differences to conv3x3_igemm_kernel (cy_conv3x3.hip), all aimed at the LDS side of the tap loop,
// which is what bounds that kernel:
//  * the GEMM's M dimension is the flattened HALO grid: a tile is TH rows x 16 columns of which the
//    inner 14 are real outputs (2 of 16 accumulator columns are computed and thrown away).  A tap
//    (dh, dw) then is ONE uniform address shift (dh*16 + dw positions) for every lane: no per-lane
//    index arithmetic in the tap loop, and 16 consecutive lanes always read 16 consecutive positions;
//  * LDS holds a chunk as planes [16-byte channel group][position] instead of [position][channels]:
//    the 16 lanes a ds_read_b128 services together read 256 contiguous bytes -- conflict-free without
//    a swizzle for every tile shape (the 28- and 14-wide tilings of the other kernel spend ~45 % of
//    their LDS cycles in bank conflicts at the row jumps); fragment addresses are
//    lane base + compile-time immediate;
//  * four waves as 2x2 (128 couts) or 4x1: 0.75-1 LDS fragment reads per MFMA instead of 1.14;
//  * the next input-channel chunk's halo tile (and, for the all-taps-resident small-cout variant,
//    its weights) is requested from global memory into registers BEFORE the current chunk's MFMAs
//    and written to LDS after them: global latency hides behind the tap loop.
// Weights are read in the same packed image as the other kernel ([tap][co_pad][ci_pad]).
#pragma once
#include <cstdlib>

#include "cy_conv_tile.h"

#ifndef CY_PLANE_INTERLEAVE
#define CY_PLANE_INTERLEAVE 0
#endif

namespace {

template <int V> struct TapC {
  static constexpr int value = V;
};
#define CY_NINE_TAPS(STMT)                  \
  do {                                      \
    { constexpr int TAP_ = 0; STMT; }       \
    { constexpr int TAP_ = 1; STMT; }       \
    { constexpr int TAP_ = 2; STMT; }       \
    { constexpr int TAP_ = 3; STMT; }       \
    { constexpr int TAP_ = 4; STMT; }       \
    { constexpr int TAP_ = 5; STMT; }       \
    { constexpr int TAP_ = 6; STMT; }       \
    { constexpr int TAP_ = 7; STMT; }       \
    { constexpr int TAP_ = 8; STMT; }       \
  } while (0)

template <int I, int N, typename F> __device__ __forceinline__ void plane_static_for(F&& f) {
  if constexpr (I < N) {
    f(TapC<I>{});
    plane_static_for<I + 1, N>(f);
  }
}

template <typename T, int TH, int BN, int WGM, int WGN, int PITCHB, bool ALLT>
struct PlaneCfg {
  static constexpr int EPC = ElemTr<T>::EPC;
  static constexpr int KC = PITCHB / (int)sizeof(T);  // channels per chunk
  static constexpr int CPP = PITCHB / 16;             // planes (16-byte channel groups) per chunk
  static constexpr int KS = KC / 16;                  // MFMA k-steps per chunk
  static constexpr int NCH = Mma<T>::NCHUNK;          // planes per fragment
  static constexpr int TW = 14, HP = 16;
  static constexpr int NPOS = (TH + 2) * HP;          // halo positions of a tile
  static constexpr int ZB = NPOS + 2;                 // first position of the all-zero row (18 wide)
  static constexpr int SKEW = 16 / CPP;               // plane pitch = 16k + SKEW positions: the staging
                                                      // writes (CPP groups x 16/CPP positions) hit 16 slots
  static constexpr int APL = ((ZB + 18 - SKEW + 15) / 16) * 16 + SKEW;  // positions per A plane
  static constexpr int BPL = BN + SKEW;                                 // rows per B plane
  static constexpr int A_BYTES = CPP * APL * 16;
  static constexpr int B_BYTES = CPP * BPL * 16;
  static constexpr int MT = TH * HP / 32;
  static constexpr int M_REP = MT / WGM;
  static constexpr int N_REP = BN / (32 * WGN);
  static constexpr int NA = (NPOS * CPP + 255) / 256;       // halo 16-byte items per thread
  static constexpr int BREG = (BN * CPP + 255) / 256;       // weight items per thread, one tap
  static constexpr int NBR = (9 * BN * CPP + 255) / 256;    // ... all nine taps
  static constexpr int NBUF = ALLT ? 9 : 2;
  static constexpr int TAB_BYTES = (3 * TH + 4) * 4;
  static constexpr int EPI_BYTES = 4 * 32 * 36 * 4 + WGM * 2 * BN * 4;
  static constexpr int MAIN_BYTES = A_BYTES + NBUF * B_BYTES;
  static constexpr int MAINB = ((MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + 15) & ~15;
  static constexpr int SMEM = MAINB + TAB_BYTES + 16;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(TH % 2 == 0 && MT % WGM == 0, "tile/wave split");
  static_assert(BN % (32 * WGN) == 0, "cout/wave split");
  static_assert(256 % CPP == 0 && 16 % CPP == 0, "chunk ownership");
  static_assert(APL >= ZB + 18 && APL % 16 == SKEW % 16, "plane pitch");
};

// PREFA (ring variant only): one workgroup per CU with the 512-register budget that buys -- the next
// chunk's halo tile is requested into registers before the nine taps, as in the ALLT variants
template <typename T, int TH, int BN, int WGM, int WGN, int PITCHB, bool ALLT, bool PREFA = false>
__global__ void __launch_bounds__(256, PREFA ? 1 : 2)
    conv3x3_plane_kernel(const ConvArgs a) {
  using C = PlaneCfg<T, TH, BN, WGM, WGN, PITCHB, ALLT>;
  using M = Mma<T>;
  constexpr int EPC = C::EPC, KC = C::KC, CPP = C::CPP, KS = C::KS, NCH = C::NCH;
  constexpr int M_REP = C::M_REP, N_REP = C::N_REP, NA = C::NA;
  constexpr int APLB = C::APL * 16, BPLB = C::BPL * 16;  // plane pitches in bytes
  constexpr int TW = C::TW;
  constexpr bool INTERLEAVE = sizeof(T) == 2 && CY_PLANE_INTERLEAVE;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + C::A_BYTES;
  int* s_row1 = reinterpret_cast<int*>(smem + C::MAINB);
  int* s_row2 = s_row1 + (TH + 2);
  int* s_flag = s_row2 + (TH + 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  // Workgroups are dealt round robin over the 8 XCDs (b and b + 8 share one, MI355X_MICROARCH.md "Workgroup
  // dispatch"): with tile = b, vertically adjacent tiles -- which share two of their 18 halo rows -- sit on
  // different XCDs and the rows are fetched from HBM into two L2s (measured traffic 1.33 x algorithmic).
  // Remapped, every XCD owns a contiguous band of tiles (bijective for any tile count).
  int tile = blockIdx.x;
  if (a.xcd_remap) {
    const int nt = gridDim.x, x = tile & 7, i = tile >> 3, q = nt >> 3, rr = nt & 7;
    tile = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
  }
  const int ct = tile % a.tiles_w;
  const int rt = tile / a.tiles_w;
  const int R0 = rt * TH, w0 = ct * TW;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int ncc = (Cin + KC - 1) / KC;
  const int cc0 = (ncc * (int)blockIdx.z) / a.ksplit;  // split-K range of channel chunks
  const int cc1 = (ncc * ((int)blockIdx.z + 1)) / a.ksplit;

  for (int idx = tid; idx < CPP * 18; idx += 256)
    st16(sA + (idx / 18) * APLB + (C::ZB + idx % 18) * 16, u32x4{0u, 0u, 0u, 0u});
  conv_row_tables(a, TH, R0, tid, s_row1, s_row2, s_flag, false);
  __syncthreads();

  // per-lane fragment bases (bytes inside sA): [m][dh+1]; a row that is the first / last of its
  // image reads the all-zero row instead of its upper / lower neighbour
  int abase[M_REP][3];
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int ty = 2 * (wm * M_REP + m) + (r >> 4);
    const int hx = r & 15;
    const int flag = s_flag[ty];
    const int hoff = h * NCH * APLB;
    // (bases are for dw = -1, so the tap's column shift is a non-negative immediate)
    const int mid = (ty + 1) * 16 + hx;
    const int zer = C::ZB + hx;
    abase[m][0] = ((flag & 1) ? zer : mid - 16) * 16 + hoff;
    abase[m][1] = mid * 16 + hoff;
    abase[m][2] = ((flag & 2) ? zer : mid + 16) * 16 + hoff;
  }
  int bbase[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) bbase[n] = ((wn * N_REP + n) * 32 + r) * 16 + h * NCH * BPLB;

  f32x16 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  auto frag = [&](const unsigned char* p, int plane_bytes) {
    typename M::Frag f;
    if constexpr (NCH == 1) {
      f.v = *reinterpret_cast<const decltype(f.v)*>(p);
    } else {
      f.lo = *reinterpret_cast<const f32x4*>(p);
      f.hi = *reinterpret_cast<const f32x4*>(p + plane_bytes);
    }
    return f;
  };

  // ---- halo staging, split into "request" (global -> registers) and "commit" (-> LDS) ----------
  const int ch = tid & (CPP - 1);
  const T* s1 = reinterpret_cast<const T*>(a.src1);
  const T* s2 = reinterpret_cast<const T*>(a.src2);
  constexpr int NAW = (ALLT || PREFA) ? NA : (NA < 4 ? NA : 4);  // items in flight per thread (register window)
  u32x4 areg[NAW];
  unsigned aok = 0;  // bit i: areg[i] holds loaded data (else zero fill)
  auto a_pooled = [&](int c0) { return a.mode1 == CY_SRC_POOL2 && c0 + ch * EPC < a.C1; };
  auto a_request = [&](int c0, int i0) {
    const int cabs = c0 + ch * EPC;
    const bool in2 = cabs >= a.C1;
    const bool cvalid = cabs < Cin;
    aok = 0;
    if (a_pooled(c0)) return;  // 2x2 max on load: staged synchronously in a_commit
    const T* base = in2 ? s2 + (cabs - a.C1) : s1 + cabs;
    const int ld = in2 ? a.ld2 : a.ld1;
    const int* rtab = in2 ? s_row2 : s_row1;
    const int wsh = (!in2 && a.mode1 == CY_SRC_UP2) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NAW; ++i) {
      const int lin = (tid + (i0 + i) * 256) / CPP;
      const int hr = lin >> 4, hc = lin & 15;
      const int w = w0 - 1 + hc;
      const int rp = lin < C::NPOS ? rtab[hr] : -1;
      areg[i] = u32x4{0u, 0u, 0u, 0u};
      if (cvalid && rp >= 0 && w >= 0 && w < a.W) {
        areg[i] = ld16(base + (size_t)(rp + (w >> wsh)) * ld);
        aok |= 1u << i;
      }
    }
  };
  auto a_commit = [&](int c0, int i0) {
    const int cabs = c0 + ch * EPC;
    unsigned char* dstp = sA + ch * APLB + 16;  // position index = 1 + lin
    if (a_pooled(c0)) {
      if (i0 != 0) return;
      for (int lin = tid / CPP; lin < C::NPOS; lin += 256 / CPP) {
        const int hr = lin >> 4, hc = lin & 15;
        const int w = w0 - 1 + hc;
        u32x4 v = {0u, 0u, 0u, 0u};
        const int rp = s_row1[hr];
        if (w >= 0 && w < a.W && rp >= 0) {
          const T* p = s1 + (size_t)(rp + 2 * w) * a.ld1 + cabs;
          const size_t rowstep = (size_t)(2 * a.W) * a.ld1;
          const u32x4 v00 = ld16(p), v01 = ld16(p + a.ld1), v10 = ld16(p + rowstep),
                      v11 = ld16(p + rowstep + a.ld1);
          float f0[EPC], f1[EPC], f2[EPC], f3[EPC];
          Chunk<T>::unpack(v00, f0);
          Chunk<T>::unpack(v01, f1);
          Chunk<T>::unpack(v10, f2);
          Chunk<T>::unpack(v11, f3);
#pragma unroll
          for (int j = 0; j < EPC; ++j) f0[j] = fmaxf(fmaxf(f0[j], f1[j]), fmaxf(f2[j], f3[j]));
          v = Chunk<T>::pack(f0);
        }
        st16(dstp + lin * 16, v);
      }
      return;
    }
    const bool pro = a.prologue && cabs < a.C1;
    float sc[EPC], sh[EPC];
    if (pro) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        sc[j] = a.scale[cabs + j];
        sh[j] = a.shift[cabs + j];
      }
    }
#pragma unroll
    for (int i = 0; i < NAW; ++i) {
      const int lin = (tid + (i0 + i) * 256) / CPP;
      if (lin < C::NPOS) {
        u32x4 v = areg[i];
        if (pro && ((aok >> i) & 1u)) {
          float f[EPC];
          Chunk<T>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(sc[j], f[j], sh[j]), 0.f);
          v = Chunk<T>::pack(f);
        }
        st16(dstp + lin * 16, v);
      }
    }
  };

  // ---- weights ------------------------------------------------------------------------------------
  const T* wp = reinterpret_cast<const T*>(a.w);
  constexpr int NBREG = ALLT ? C::NBR : C::BREG;
  constexpr int NBITEMS = (ALLT ? 9 : 1) * BN * CPP;
  u32x4 breg[NBREG];
  // item (tap, cout row, plane) of thread `tid`, register i: addresses are a wave-uniform part
  // (tap, chunk, row block -> scalar registers) plus ONE per-thread 32-bit offset
  constexpr int IPT = BN * CPP;                     // 16-byte items per tap
  constexpr int TPR = IPT >= 256 ? 1 : 256 / IPT;   // taps covered by one register round (ALLT)
  static_assert(IPT % 256 == 0 || 256 % IPT == 0, "weight item split");
  const int bsub = (IPT >= 256 ? 0 : wave / (4 / TPR));  // which tap of a register round (wave-uniform)
  const int brow0 = (tid & (IPT >= 256 ? 255 : IPT - 1)) / CPP;
  const unsigned bvoff = (unsigned)((brow0 * a.w_ci_pad + ch * EPC) * (int)sizeof(T));
  const int bdst0 = ch * BPLB + brow0 * 16;
  auto b_request = [&](int cc, int tap0) {  // ALLT: all nine taps of chunk cc (tap0 ignored)
#pragma unroll
    for (int i = 0; i < NBREG; ++i) {
      const int tap = ALLT ? (IPT >= 256 ? (i * 256) / IPT : i * TPR + bsub) : tap0;
      const int rblk = IPT >= 256 ? ((i * 256) % IPT) / CPP : 0;  // first cout row of this round
      if (!ALLT || tap < 9) {
        const size_t soff = (((size_t)(tap * a.w_co_pad + n0 + rblk)) * a.w_ci_pad + cc * KC) * sizeof(T);
        breg[i] = ld16(reinterpret_cast<const unsigned char*>(wp) + soff + bvoff);
      }
    }
  };
  auto b_commit = [&](unsigned char* dst) {  // ALLT: dst = sB (nine slices), else one ring slot
#pragma unroll
    for (int i = 0; i < NBREG; ++i) {
      const int tap = ALLT ? (IPT >= 256 ? (i * 256) / IPT : i * TPR + bsub) : 0;
      const int rblk = IPT >= 256 ? ((i * 256) % IPT) / CPP : 0;
      if (!ALLT || tap < 9) st16(dst + tap * C::B_BYTES + bdst0 + rblk * 16, breg[i]);
    }
  };

  // One tap of one chunk: KS k-steps of M_REP x N_REP MFMAs; the fragments of k-step ks+1 are
  // requested before the MFMAs of k-step ks are issued (two register sets, pinned by sched_barrier).
  // (the tap is a compile-time constant: a rolled tap loop would index abase[][] through
  // s_set_gpr_idx, which makes the compiler drain every outstanding global load first)
  auto mma_tap = [&](const unsigned char* sBc, auto TAP) {
    constexpr int d = decltype(TAP)::value / 3, dw = decltype(TAP)::value % 3 - 1;
    typename M::Frag af[2][M_REP], bf[2][N_REP];
    const unsigned char* sAt = sA + (dw + 1) * 16;
#pragma unroll
    for (int n = 0; n < N_REP; ++n) bf[0][n] = frag(sBc + bbase[n], BPLB);
#pragma unroll
    for (int m = 0; m < M_REP; ++m) af[0][m] = frag(sAt + abase[m][d], APLB);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KS) {
#pragma unroll
        for (int n = 0; n < N_REP; ++n)
          bf[nxt][n] = frag(sBc + bbase[n] + (ks + 1) * 2 * NCH * BPLB, BPLB);
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
          af[nxt][m] = frag(sAt + abase[m][d] + (ks + 1) * 2 * NCH * APLB, APLB);
      }
      if constexpr (INTERLEAVE) {
        // bf16: the LDS reads of k-step ks+1 are issued BETWEEN the MFMAs of k-step ks (an MFMA holds the
        // SIMD's issue port for 8 of its 32 cycles), not as a burst in front of them during which the matrix
        // pipe idles: in-kernel stamps of the same loop in round 2's producer / consumer experiment, 48 -> 40 cycles per MFMA
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
#pragma unroll
          for (int n = 0; n < N_REP; ++n) M::mma(bf[cur][n], af[cur][m], acc[m][n]);  // rows = couts
        constexpr int NM = M_REP * N_REP;
        const int NR = (ks + 1 < KS) ? (M_REP + N_REP) * NCH : 0;
        (void)NR;
        if (ks + 1 < KS) {
          plane_static_for<0, NM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int NRC = (M_REP + N_REP) * NCH;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            constexpr int nr = ((j + 1) * NRC) / NM - (j * NRC) / NM;
            if constexpr (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
#pragma unroll
          for (int n = 0; n < N_REP; ++n) M::mma(bf[cur][n], af[cur][m], acc[m][n]);  // rows = couts
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  auto a_stage = [&](int c0) {  // synchronous staging, NAW requests in flight
    for (int i0 = 0; i0 < NA; i0 += NAW) {
      a_request(c0, i0);
      a_commit(c0, i0);
    }
  };
  b_request(cc0, 0);
  a_stage(cc0 * KC);
  b_commit(sB);
  __syncthreads();

  if constexpr (ALLT) {
    for (int cc = cc0; cc < cc1; ++cc) {
      const bool more = cc + 1 < cc1;
      if (more) {
        a_request((cc + 1) * KC, 0);
        b_request(cc + 1, 0);
      }
      CY_NINE_TAPS(mma_tap(sB + TAP_ * C::B_BYTES, TapC<TAP_>{}));
      if (more) {
        __syncthreads();  // every wave is done with this chunk's LDS
        a_commit((cc + 1) * KC, 0);
        b_commit(sB);
        __syncthreads();
      }
    }
  } else {
    const int it0 = cc0 * 9, nit = cc1 * 9;
    for (int cc = cc0; cc < cc1; ++cc) {
      const bool more = cc + 1 < cc1;
      // (two workgroups per CU: no register prefetch of the next halo tile -- 128 accumulator + 48
      // fragment registers leave no room for it; the second workgroup of the CU covers the staging)
      if (PREFA && more) a_request((cc + 1) * KC, 0);
      auto ring_tap = [&](auto TAP) {
        constexpr int tap = decltype(TAP)::value;
        const int it = cc * 9 + tap;
        const bool has_next = it + 1 < nit;
        if (has_next) b_request(tap == 8 ? cc + 1 : cc, tap == 8 ? 0 : tap + 1);
        mma_tap(sB + ((it - it0) & 1) * C::B_BYTES, TAP);
        if (has_next) b_commit(sB + ((it + 1 - it0) & 1) * C::B_BYTES);
        if (tap == 8 && more) {
          __syncthreads();  // every wave is done reading the halo tile
          if constexpr (PREFA) a_commit((cc + 1) * KC, 0); else a_stage((cc + 1) * KC);
        }
        __syncthreads();
      };
      CY_NINE_TAPS(ring_tap(TapC<TAP_>{}));
    }
  }
  if constexpr (ALLT) __syncthreads();  // the statistics scratch aliases the operand buffers

  // ---------------- epilogue: accumulators -> NHWC, straight from registers ----------------
  // The MFMAs run with the WEIGHTS as the row operand, so a lane holds ONE position (q = frag base
  // + r; q = ty*16 + hx, columns hx = 0 and 15 are the halo columns = garbage) and 16 couts in four
  // runs of four consecutive channels ((reg&3) + 8*(reg>>2) + 4*h): four 8-byte (bf16) stores per
  // fragment, no LDS transpose and no per-element index arithmetic.
  auto position = [&](int m, int& R, int& w) -> bool {
    const int q = (wm * M_REP + m) * 32 + r;
    const int hx = q & 15;
    R = R0 + (q >> 4);
    w = w0 + hx - 1;
    return hx >= 1 && hx <= TW && R < a.NH && w < a.W;
  };
  if (a.ksplit > 1) {
    float* wsz = a.ws + (size_t)blockIdx.z * ((size_t)a.NH * a.W) * a.Cout;
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      int R, w;
      if (!position(m, R, w)) continue;
      float* dst = wsz + ((size_t)R * a.W + w) * a.Cout;
#pragma unroll
      for (int n = 0; n < N_REP; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = n0 + (wn * N_REP + n) * 32 + 8 * g + 4 * h;
          if (co < a.Cout)
            *reinterpret_cast<f32x4*>(dst + co) = f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1],
                                                        acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
        }
    }
    return;
  }
  float* sstat = reinterpret_cast<float*>(smem);
  const bool do_stats = a.stats != nullptr || a.sacc != nullptr;
  T* o1 = reinterpret_cast<T*>(a.out);
  T* o2 = reinterpret_cast<T*>(a.out2);
#pragma unroll
  for (int n = 0; n < N_REP; ++n) {
    float s1[16], s2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      int R, w;
      const bool ok = position(m, R, w);
      const size_t gp = (size_t)R * a.W + w;
      u32x2 packed[4];  // bf16: the four 4-channel runs of this lane, packed
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + (wn * N_REP + n) * 32 + 8 * g + 4 * h;
        T pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[m][n][4 * g + j];
          pk[j] = from_f32<T>(v);
          if (do_stats && ok) {
            const float qv = to_f32<T>(pk[j]);
            s1[4 * g + j] += qv;
            s2[4 * g + j] += qv * qv;
          }
        }
        if constexpr (sizeof(T) == 2) {
          packed[g] = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
        } else if (ok && co < a.Cout) {
          T* dst = (a.split_c > 0 && co >= a.split_c) ? o2 + gp * a.ldo2 + (co - a.split_c)
                                                      : o1 + gp * a.ldo + co;
          *reinterpret_cast<f32x4*>(dst) = f32x4{pk[0], pk[1], pk[2], pk[3]};
        }
      }
      if constexpr (sizeof(T) == 2) {
        // lane i holds channels 8g+0..3, lane i+32 channels 8g+4..7 of the same position: one half-wave
        // swap per dword of a run pair (g, g+1) leaves the lower half with channels 8g..8g+7 and the
        // upper half with 8g+8..8g+15 -- two 16-byte stores per fragment instead of four 8-byte ones
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          u32x2 lo = packed[g], hi = packed[g + 1];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(lo[j], hi[j], false, false);
            lo[j] = sw[0];
            hi[j] = sw[1];
          }
          const int co = n0 + (wn * N_REP + n) * 32 + 8 * g + 8 * h;
          if (ok && co < a.Cout) {
            T* dst = (a.split_c > 0 && co >= a.split_c) ? o2 + gp * a.ldo2 + (co - a.split_c)
                                                        : o1 + gp * a.ldo + co;
            *reinterpret_cast<u32x4*>(dst) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        }
      }
    }
    if (do_stats) {
      // reduce-scatter over the 32 lanes of each half (same h): xor 16 / 8 / 4 / 2 halve the value
      // count while they pair the lanes, xor 1 joins the last pair; a lane ends up with the total
      // of register index ((lane>>4)&1)*8 + ((lane>>3)&1)*4 + ((lane>>2)&1)*2 + ((lane>>1)&1)
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
      if ((lane & 1) == 0) {
        const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int col = (wn * N_REP + n) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        sstat[(wm * 2 + 0) * BN + col] = s1[0];
        sstat[(wm * 2 + 1) * BN + col] = s2[0];
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int qq = 0; qq < WGM; ++qq) {
        t1 += sstat[(qq * 2 + 0) * BN + tid];
        t2 += sstat[(qq * 2 + 1) * BN + tid];
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

template <typename T, int TH, int BN, int WGM, int WGN, int PITCHB, bool ALLT, bool PREFA = false>
int launch_conv_plane(ConvArgs a, hipStream_t st) {
  using C = PlaneCfg<T, TH, BN, WGM, WGN, PITCHB, ALLT>;
  auto kern = conv3x3_plane_kernel<T, TH, BN, WGM, WGN, PITCHB, ALLT, PREFA>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  if (a.W % C::TW != 0) return CY_ERR_SHAPE;
  a.tiles_w = a.W / C::TW;
  a.full_tiles = 0;
  static const int xcd = [] {
    const char* e = getenv("CY_PLANE_XCD");
    return e ? atoi(e) : 1;
  }();
  a.xcd_remap = xcd;
  dim3 grid(cy_cdiv(a.NH, TH) * a.tiles_w, cy_cdiv(a.Cout, BN), a.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM, st, a);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // namespace
