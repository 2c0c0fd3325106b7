// Streaming form of the plane kernel for the LOW-CHANNEL layers (Cin, Cout in {32, 64}: the 224x224 and
// 112x112 levels of the U-Net and their data gradients), which are HBM-bound by arithmetic intensity
// (AI 144-288 FLOP/B < ridge 312) but run at 2.4-2.8 TB/s in the plane kernel: a workgroup there is one
// 16x14 tile -- load (a ~2 us latency chain), 36-72 MFMAs, store -- and with three of them per CU only
// ~24 KB of loads are in flight per CU, which is what 2.6 TB/s is (Little's law at ~2 us).
// Here:
//   * persistent workgroups walk a contiguous range of tiles; the halo tile of tile i+2 is requested by
//     LDS-DMA (buffer_load_dwordx4 ... lds: lane = halo position, gather offsets per lane, padding = out of
//     range = zeros; nearest-x2 upsampling and the second source of a concat are just other offsets) while
//     tile i is computed and tile i+1 is landing: three tile buffers, 40-80 KB in flight per CU, no
//     registers, no commit;
//   * the weights never touch LDS: a wave keeps the fragments of its 32 couts for all nine taps of up to 64
//     input channels in registers (72 / 144 VGPRs) for the life of the workgroup -- the MFMA loop reads one
//     LDS fragment per MFMA (the activations) and nothing else;
//   * one s_barrier per tile; vector-memory operations are issued unconditionally (invalid lanes read /
//     write out of range) so that every wait is a counted vmcnt;
//   * BN partial sums per (tile, wave row) straight from registers, no second barrier.
// Layers with a load transform (BN+ReLU prologue, 2x2 max) stay on the plane kernel.
#pragma once
#include "cy_conv_plane.h"

namespace {

static unsigned long long* g_conv_stamp_buf = nullptr;  // set by cy_debug_conv_stamps
#define P8_STAMP()                                                                              \
  do {                                                                                          \
    if (stamping && nst < 96) {                                                                 \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) s_stamp[wave * 96 + nst] = t__;                                            \
      ++nst;                                                                                    \
    }                                                                                           \
  } while (0)

// Shapes: four waves along the 256 halo positions of a tile (two 32-position blocks each) per block of 32 couts
// (NBW blocks per wave: 1); NBUF tile buffers (2 for Cin 64 -> Cout 32, so that two workgroups fit a CU's LDS).  Two
// four-wave workgroups per CU run out of phase with each other -- DMA issue, MFMAs and epilogue of a tile are
// serial within a workgroup.
// (Cout 64 as both blocks in one wave -- every activation fragment feeding two MFMAs, 144-288 weight registers,
//  one wave per SIMD at up to 512 registers -- halves the LDS reads that bound the eight-wave form (stamps: 5500
//  cycles per 72 MFMAs) but was 15-25 % slower: a wave alone on its SIMD exposes every wait)
template <int KCH, int NB> constexpr int stream_nbw() { return 1; }
template <int KCH, int NB> constexpr int stream_nbuf() { return (KCH * NB == 4) ? 3 : 2; }
// workgroups per CU: three four-wave workgroups of the 32-channel shapes (<= 168 registers, 39 KB of LDS each),
// two of Cin 64 -> Cout 32 (two waves per SIMD at 144 weight registers), one eight-wave workgroup for 64 -> 64
template <int KCH, int NB> constexpr int stream_wgs() { return KCH == 1 ? (NB == 1 ? 3 : 1) : (NB == 1 ? 2 : 1); }

template <typename T, int KCH, int NB> struct StreamCfg {
  static constexpr int TH = 16, TW = 14, HP = 16;
  static constexpr int WM = 4, M_REP = 2, NBW = stream_nbw<KCH, NB>();
  static constexpr int NWAVE = WM * NB / NBW, NTHR = 64 * NWAVE;
  static constexpr int GS = NWAVE / 4;  // waves that share a plane: position groups interleaved over them
  static constexpr int EPC = ElemTr<T>::EPC;
  static constexpr int KC = 32, CPP = 4;
  static constexpr int NPOS = (TH + 2) * HP;
  static constexpr int ZB = NPOS + 2;
  static constexpr int SKEW = 16 / CPP;
  static constexpr int APL = ((ZB + 18 - SKEW + 15) / 16) * 16 + SKEW;
  static constexpr int APLB = APL * 16;
  static constexpr int CH_BYTES = CPP * APLB;  // one 32-channel chunk of a halo tile
  static constexpr int BUF_BYTES = KCH * CH_BYTES;
  static constexpr int NBUF = stream_nbuf<KCH, NB>();
  static constexpr int NG = (NPOS + 63) / 64;  // position groups of 64 (the last one is half a wave)
  static constexpr int GPW = (NG + GS - 1) / GS;  // position groups per wave, at most
#ifdef CY_STREAM_STAMPS  // development aid (-DCY_STREAM_STAMPS): [wave][96] clock stamps of workgroup 0; costs LDS
  static constexpr int STAMP_BYTES = 8 * 96 * 8;
#else
  static constexpr int STAMP_BYTES = 0;
#endif
  static constexpr int SMEM = NBUF * BUF_BYTES + 16 + STAMP_BYTES + 2 * 2 * 20 * 4;
  static_assert(sizeof(T) == 2 && EPC == 8, "16-bit storage types only");
  static_assert(NPOS % 64 == 32, "the last group is half a wave");
};

// PRO (Cin 32 -> Cout 32 only): BN+ReLU prologue relu(scale[c] * x + shift[c]) on load -- a wave transforms, in place
// in LDS, exactly the 16-byte items its own DMA instructions brought (one plane = eight fixed channels per wave,
// so the coefficients are 16 registers for the life of the workgroup); padding positions stay zero.
template <typename T, int KCH, int NB, bool STATS, bool PRO = false>
__global__ void __launch_bounds__((256 * NB / stream_nbw<KCH, NB>()), (PRO ? 2 : stream_wgs<KCH, NB>()))
    conv3x3_stream_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // (buffer-descriptor type and builtins exist in the device pass only)
  using C = StreamCfg<T, KCH, NB>;
  using M = Mma<T>;
  constexpr int EPC = C::EPC, TH = C::TH, TW = C::TW, APLB = C::APLB, NG = C::NG, GPW = C::GPW, WM = C::WM, M_REP = C::M_REP, NBW = C::NBW;
  constexpr unsigned OOB = 0xffffff00u;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* s_stamp = reinterpret_cast<unsigned long long*>(smem + C::NBUF * C::BUF_BYTES + 16);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CY_STREAM_STAMPS
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0;
#else
  constexpr bool stamping = false;
#endif
  int nst = 0;
  const int wm = wave % WM, wn = wave / WM;  // block of 64 positions / first of this wave's NBW 32-cout blocks is wn * NBW
  const int sub = wave >> 2;                 // which of the GS waves that share plane (wave & 3)
  const int r = lane & 31, h = lane >> 5;
  const int Cin = a.C1 + a.C2;

  // ---- this workgroup's tiles: a contiguous range (vertical neighbours share halo rows in one L2) -------------
  const int ntiles = ((a.NH + TH - 1) / TH) * a.tiles_w;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  // (a workgroup without tiles still runs the protocol once: its statistics partial rows must be zeros)

  // ---- weights -> registers: fragments of this wave's 32 couts, nine taps, 2 KCH k-steps -----------------------
  typename M::Frag breg[NBW][KCH][9][2];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w);
    const size_t row = (size_t)((wn * NBW + nb) * 32 + r) * a.w_ci_pad + h * EPC;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4 v = ld16(wp + (((size_t)tap * a.w_co_pad * a.w_ci_pad) + row + kc * 32 + ks * 16) * sizeof(T));
          breg[nb][kc][tap][ks].v = __builtin_bit_cast(decltype(breg[nb][kc][tap][ks].v), v);
        }
  }

  // ---- descriptors (from wave-uniform scalars only) ---------------------------------------------------------------
  auto make_rsrc = [&](const void* p, long long bytes) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.src1, a.bytes1);
  const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(a.src2 ? a.src2 : a.src1, a.src2 ? a.bytes2 : 0);
  const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(a.out, a.bytes_o1);
  const __amdgpu_buffer_rsrc_t ro2 = make_rsrc(a.out2 ? a.out2 : a.out, a.out2 ? a.bytes_o2 : 0);
  const __amdgpu_buffer_rsrc_t rst = make_rsrc(a.stats ? (const void*)a.stats : a.out, a.stats ? a.bytes_st : 0);

  // zero rows of every chunk of every buffer (never touched by the DMA)
  for (int idx = tid; idx < C::NBUF * KCH * C::CPP * 18; idx += C::NTHR) {
    const int pl = idx / 18, z = idx % 18;
    st16(smem + pl * APLB + (C::ZB + z) * 16, u32x4{0u, 0u, 0u, 0u});
  }

  const float invH = 1.0f / (float)a.H, invTW = 1.0f / (float)a.tiles_w;
  auto tile_origin = [&](int t, int& R0, int& w0) {  // wave-uniform
    int rt = (int)(((float)t + 0.5f) * invTW);
    rt = t - rt * a.tiles_w < 0 ? rt - 1 : rt;
    R0 = rt * TH;
    w0 = (t - rt * a.tiles_w) * TW;
  };

  // ---- halo DMA of tile t into buffer `buf` (byte offset); t >= t_end: every lane out of range (counts stay static).
  // Per tile, 18 lanes of wave 0 write a row table (byte offset of (row, w = 0) in either source, or -1: the
  // nearest-x2 and image arithmetic lives there, once per row); a lane's offset for a position group then is
  // table[row] + its column term -- one ds_read_b32 and three VALU per DMA instruction.  (The first version did
  // the whole index arithmetic per lane and group: 30 quarter-rate multiplies, spilled SGPRs and 2100-2800 cycles
  // of issue phase per tile, in-kernel stamps.)
  int* s_rt = reinterpret_cast<int*>(smem + C::NBUF * C::BUF_BYTES + 16 + C::STAMP_BYTES);  // [2 parities][2 sources][20]
  const int pl = wave & 3;
  const int hc = lane & 15;
  auto write_row_table = [&](int t, int par) {  // wave 0, before the barrier of the iteration that issues tile t
    if (wave != 0 || lane >= TH + 2) return;
    int R0, w0;
    tile_origin(t < t_end ? t : t_end - 1, R0, w0);
    const int R = R0 - 1 + lane;
    int o1 = -1, o2 = -1;
    if (t < t_end && R >= 0 && R < a.NH) {
      int px = R * a.W;
      if (a.mode1 == CY_SRC_UP2) {
        int n = (int)(((float)R + 0.5f) * invH);
        n = R - n * a.H < 0 ? n - 1 : n;
        px = (n * (a.H >> 1) + ((R - n * a.H) >> 1)) * (a.W >> 1);
      }
      o1 = px * a.ld1 * (int)sizeof(T);
      o2 = R * a.W * a.ld2 * (int)sizeof(T);
    }
    s_rt[(par * 2 + 0) * 20 + lane] = o1;
    s_rt[(par * 2 + 1) * 20 + lane] = o2;
  };
  // per 32-channel chunk: which descriptor, which table / column term, which scalar offset -- fixed for the
  // kernel, so that the issue below is straight-line code (scalar branches per DMA cost more than the DMA)
  const bool cat = a.C2 > 0;  // host: then C1 = 32 = first chunk, second chunk = source 2
  const __amdgpu_buffer_rsrc_t rsk1 = (KCH == 2 && cat) ? rs2 : rs1;
  const unsigned soffk1 = (KCH == 2 && !cat) ? 32u * (unsigned)sizeof(T) : 0u;
  auto issue_tile = [&](int t, int buf, int par) {
    int R0, w0;
    tile_origin(t < t_end ? t : t_end - 1, R0, w0);
    const int w = w0 - 1 + hc;
    const bool col_ok = w >= 0 && w < a.W;
    const int wsh = a.mode1 == CY_SRC_UP2 ? 1 : 0;
    const int wt1 = ((w >> wsh) * a.ld1 + pl * EPC) * (int)sizeof(T);
    const int wt2 = (w * a.ld2 + pl * EPC) * (int)sizeof(T);
    int r1[GPW], r2[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = C::GS * gi + sub;
      const int hr = 4 * g + (lane >> 4);
      const int idx = hr < TH + 2 ? hr : 0;
      r1[gi] = s_rt[(par * 2 + 0) * 20 + idx];
      r2[gi] = s_rt[(par * 2 + 1) * 20 + idx];
    }
    unsigned o1[GPW], o2[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = C::GS * gi + sub;
      const bool in_tile = g * 64 + lane < C::NPOS;
      o1[gi] = (col_ok && in_tile && r1[gi] >= 0) ? (unsigned)(r1[gi] + wt1) : OOB;
      o2[gi] = (col_ok && in_tile && r2[gi] >= 0) ? (unsigned)(r2[gi] + wt2) : OOB;
      if (KCH == 2 && !cat) o2[gi] = o1[gi];
    }
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = C::GS * gi + sub;  // wave-uniform
      if (g >= NG) continue;
      auto* l0 = (__attribute__((address_space(3))) void*)(smem + buf + pl * APLB + 16 + g * 1024);
      if (g * 64 + 64 <= C::NPOS) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, l0, 16, o1[gi], 0, 0, 0);
        if constexpr (KCH == 2) {
          auto* l1 = (__attribute__((address_space(3))) void*)(smem + buf + C::CH_BYTES + pl * APLB + 16 + g * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk1, l1, 16, o2[gi], soffk1, 0, 0);
        }
      } else if (lane < 32) {  // the last group is half a wave
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, l0, 16, o1[gi], 0, 0, 0);
        if constexpr (KCH == 2) {
          auto* l1 = (__attribute__((address_space(3))) void*)(smem + buf + C::CH_BYTES + pl * APLB + 16 + g * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk1, l1, 16, o2[gi], soffk1, 0, 0);
        }
      }
    }
  };
  float psc[EPC], psh[EPC];
  if constexpr (PRO) {
    if (a.fold.acc) {
      // coefficients from the previous layer's sums (cy_bn_acc.h): one channel per thread, through LDS (a tile buffer:
      // the first DMA is issued behind this); workgroup 0 leaves them in memory for the backward pass
      float* scoef = reinterpret_cast<float*>(smem);
      unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(smem + 512);
      bn_acc_gather<true>(a.fold.acc, a.fold.R, a.C1, s_sum, tid, C::NTHR);
      if (tid < a.C1) {  // host: C1 == 32
        float c0, c1;
        bn_fold_channel_lds(a.fold, s_sum, tid, blockIdx.x == 0, c0, c1);
        scoef[tid] = c0;
        scoef[64 + tid] = c1;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        psc[j] = scoef[pl * EPC + j];
        psh[j] = scoef[64 + pl * EPC + j];
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        psc[j] = a.scale[pl * EPC + j];
        psh[j] = a.shift[pl * EPC + j];
      }
    }
  }
  auto transform_tile = [&](int t, int buf, int par) {  // PRO: after this wave's own vmcnt wait, before the barrier
    int R0, w0;
    tile_origin(t, R0, w0);
    const int w = w0 - 1 + hc;
    const bool col_ok = w >= 0 && w < a.W;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = C::GS * gi + sub;
      if (g >= NG) continue;
      const int hr = 4 * g + (lane >> 4);
      const bool in_tile = g * 64 + lane < C::NPOS;
      const bool valid = col_ok && in_tile && s_rt[(par * 2 + 0) * 20 + (hr < TH + 2 ? hr : 0)] >= 0;
      unsigned char* p = smem + buf + pl * APLB + 16 + (g * 64 + lane) * 16;
      if (valid) {
        float f[EPC];
        Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(p), f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(psc[j], f[j], psh[j]), 0.f);
        st16(p, Chunk<T>::pack(f));
      }
    }
  };

  // vector-memory instructions per wave: D per tile of DMA, S per tile of stores
  constexpr int D0 = ((NG + C::GS - 1) / C::GS) * KCH, D1 = (NG / C::GS) * KCH;  // sub == 0 / sub == 1
  constexpr int S = NBW * 2 * M_REP;

  auto frag = [&](const unsigned char* p) {
    typename M::Frag f;
    f.v = *reinterpret_cast<const decltype(f.v)*>(p);
    return f;
  };

  // BN statistics: per-lane running sums over all tiles of this workgroup (16 couts x this lane's positions),
  // reduced over the lanes ONCE at the end into the partial row (workgroup, wm) -- the per-tile reduce-scatter
  // (32 cross-lane exchanges) cost 2000 cycles of every tile's epilogue (stamps)
  float s1[NBW][16], s2[NBW][16];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[nb][i] = s2[nb][i] = 0.f;

  // prefetch distance PD = NBUF - 1 tiles; buffers rotate: bufs[0] = this tile's, bufs[PD] = the one being refilled
  constexpr int PD = C::NBUF - 1;
  int bufs[C::NBUF];
#pragma unroll
  for (int i = 0; i < C::NBUF; ++i) bufs[i] = i * C::BUF_BYTES;
  for (int i = 0; i < PD; ++i) {  // (serial start-up: table, barrier, issue)
    write_row_table(t_begin + i, i & 1);
    __syncthreads();
    issue_tile(t_begin + i, bufs[i], i & 1);
  }

  for (int t = t_begin; t < t_end; ++t) {
    const int it = t - t_begin;
    P8_STAMP();
    write_row_table(t + PD, (it + PD) & 1);
    // (1) tile t has landed: everything younger than its DMA may stay in flight -- per iteration a wave issues
    //     d DMA instructions (tile it + PD) and then S stores (tile it)
    {
      const int d = sub == 0 ? D0 : D1;
      const int allow = PD == 1 ? (it == 0 ? 0 : S) : (it == 0 ? d : (it == 1 ? d + S : d + 2 * S));  // wave-uniform
      // (immediates: enumerate the few values this can take)
#define CY_VMCNT_CASE(N) else if (allow == (N)) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
      if (allow == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CY_VMCNT_CASE(2); CY_VMCNT_CASE(3); CY_VMCNT_CASE(4); CY_VMCNT_CASE(5); CY_VMCNT_CASE(6); CY_VMCNT_CASE(7);
      CY_VMCNT_CASE(8); CY_VMCNT_CASE(9); CY_VMCNT_CASE(10); CY_VMCNT_CASE(11); CY_VMCNT_CASE(12); CY_VMCNT_CASE(13);
      CY_VMCNT_CASE(14); CY_VMCNT_CASE(15); CY_VMCNT_CASE(16); CY_VMCNT_CASE(17); CY_VMCNT_CASE(18); CY_VMCNT_CASE(19);
      CY_VMCNT_CASE(20); CY_VMCNT_CASE(21); CY_VMCNT_CASE(22);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CY_VMCNT_CASE
    }
    if constexpr (PRO) transform_tile(t, bufs[0], it & 1);
    P8_STAMP();
    // (2) every wave's part of tile t is in LDS, and every wave is done reading tile t-1's buffer
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    P8_STAMP();
    // (3) tile t + PD -> the buffer tile t-1 used
    issue_tile(t + PD, bufs[PD], (it + PD) & 1);

    P8_STAMP();
    // (4) this tile's geometry: fragment bases and image-boundary flags of this wave's two row pairs
    int R0, w0;
    tile_origin(t, R0, w0);
    int amid[M_REP];
    unsigned aflag = 0;
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int ty = 2 * (wm * M_REP + m) + (r >> 4);
      amid[m] = ((ty + 1) * 16 + (r & 15)) * 16 + h * APLB;
      const int R = R0 + ty;
      int n = (int)(((float)R + 0.5f) * invH);
      n = R - n * a.H < 0 ? n - 1 : n;
      const int hh = R - n * a.H;
      const unsigned f = R < a.NH ? ((hh == 0 ? 1u : 0u) | (hh == a.H - 1 ? 2u : 0u)) : 3u;
      aflag |= f << (2 * m);
    }
    const int azer = (C::ZB + (r & 15)) * 16 + h * APLB;
    auto aaddr = [&](int m, int d) {
      if (d == 1) return amid[m];
      const bool z = (aflag >> (2 * m + (d == 0 ? 0 : 1))) & 1u;
      return z ? azer : amid[m] + (d == 0 ? -256 : 256);
    };

    // (5) MFMAs: per 64-position block 18 KCH instructions, activations from LDS, weights from registers
    f32x16 acc[M_REP][NBW];
#pragma unroll
    for (int m = 0; m < M_REP; ++m)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][nb][i] = 0.f;
    {
      const unsigned char* base = smem + bufs[0];
      // flattened (m, kc, tap row d, k-step) groups of three fragments (the row's three taps), one group ahead
      constexpr int NGRP = M_REP * KCH * 3 * 2;
      typename M::Frag fa[2][3];
      auto load_group = [&](int set, int gidx) {
        const int ks = gidx % 2, d = (gidx / 2) % 3, kc = (gidx / 6) % KCH, m = gidx / (6 * KCH);
        const unsigned char* p = base + kc * C::CH_BYTES + aaddr(m, d) + ks * 2 * APLB;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) fa[set][dw] = frag(p + dw * 16);
      };
      load_group(0, 0);
      plane_static_for<0, NGRP>([&](auto G) {
        constexpr int gidx = decltype(G)::value;
        constexpr int ks = gidx % 2, d = (gidx / 2) % 3, kc = (gidx / 6) % KCH, m = gidx / (6 * KCH);
        if constexpr (gidx + 1 < NGRP) load_group((gidx + 1) & 1, gidx + 1);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
            M::mma(breg[nb][kc][d * 3 + dw][ks], fa[gidx & 1][dw], acc[m][nb]);  // rows = couts
      });
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P8_STAMP();
    // (6) epilogue: registers -> NHWC (16-byte stores, invalid lanes out of range), BN partial sums
    {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
      const int nb0 = (wn * NBW + nb) * 32;  // first cout of this block
      const bool second = a.split_c > 0 && nb0 >= a.split_c;  // host: split_c % 32 == 0
#pragma unroll
      for (int m = 0; m < M_REP; ++m) {
        const int q = (wm * M_REP + m) * 32 + r;
        const int hx = q & 15;
        const int R = R0 + (q >> 4), w = w0 + hx - 1;
        const bool ok = hx >= 1 && hx <= TW && R < a.NH && w < a.W;
        const unsigned gp = (unsigned)(R * a.W + w);
        u32x2 packed[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          T pk[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pk[j] = from_f32<T>(acc[m][nb][4 * g + j]);
            if (STATS && ok) {
              const float qv = to_f32<T>(pk[j]);
              s1[nb][4 * g + j] += qv;
              s2[nb][4 * g + j] += qv * qv;
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
          const int co = nb0 + 8 * g + 8 * h;
          const bool cok = ok && co < a.Cout;
          const u32x4 v = u32x4{lo[0], lo[1], hi[0], hi[1]};
          if (second) {
            const unsigned off = cok ? (gp * (unsigned)a.ldo2 + (unsigned)(co - a.split_c)) * (unsigned)sizeof(T) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, ro2, off, 0, 0);
          } else {
            const unsigned off = cok ? (gp * (unsigned)a.ldo + (unsigned)co) * (unsigned)sizeof(T) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, ro1, off, 0, 0);
          }
        }
      }
      }  // nb
    }
    P8_STAMP();
    const int b0 = bufs[0];
#pragma unroll
    for (int i = 0; i + 1 < C::NBUF; ++i) bufs[i] = bufs[i + 1];
    bufs[C::NBUF - 1] = b0;
  }
  if constexpr (STATS) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      float* S1 = s1[nb];
      float* S2 = s2[nb];
      const int nb0 = (wn * NBW + nb) * 32;
      {  // reduce-scatter over the 32 lanes of each half (see conv3x3_plane_kernel)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool up = (lane & 16) != 0;
          const float snd1 = up ? S1[i] : S1[i + 8], snd2 = up ? S2[i] : S2[i + 8];
          const float kp1 = up ? S1[i + 8] : S1[i], kp2 = up ? S2[i + 8] : S2[i];
          S1[i] = kp1 + __shfl_xor(snd1, 16, 64);
          S2[i] = kp2 + __shfl_xor(snd2, 16, 64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool up = (lane & 8) != 0;
          const float snd1 = up ? S1[i] : S1[i + 4], snd2 = up ? S2[i] : S2[i + 4];
          const float kp1 = up ? S1[i + 4] : S1[i], kp2 = up ? S2[i + 4] : S2[i];
          S1[i] = kp1 + __shfl_xor(snd1, 8, 64);
          S2[i] = kp2 + __shfl_xor(snd2, 8, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool up = (lane & 4) != 0;
          const float snd1 = up ? S1[i] : S1[i + 2], snd2 = up ? S2[i] : S2[i + 2];
          const float kp1 = up ? S1[i + 2] : S1[i], kp2 = up ? S2[i + 2] : S2[i];
          S1[i] = kp1 + __shfl_xor(snd1, 4, 64);
          S2[i] = kp2 + __shfl_xor(snd2, 4, 64);
        }
        {
          const bool up = (lane & 2) != 0;
          const float snd1 = up ? S1[0] : S1[1], snd2 = up ? S2[0] : S2[1];
          const float kp1 = up ? S1[1] : S1[0], kp2 = up ? S2[1] : S2[0];
          S1[0] = kp1 + __shfl_xor(snd1, 2, 64);
          S2[0] = kp2 + __shfl_xor(snd2, 2, 64);
        }
        S1[0] += __shfl_xor(S1[0], 1, 64);
        S2[0] += __shfl_xor(S2[0], 1, 64);
        // partial row (tile, wm): [2][Cout]; even lanes hold one channel each
        const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int col = nb0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const bool sok = (lane & 1) == 0 && col < a.Cout;
        if (a.sacc) {  // (accumulator form, cy_bn_acc.h: the four wave rows are summed in LDS below, one add per workgroup)
          if (nb == 0) __syncthreads();  // every wave is done with the tile buffers
          float* sred = reinterpret_cast<float*>(smem);
          if (sok) {
            sred[(wm * 2 + 0) * 64 + col] = S1[0];
            sred[(wm * 2 + 1) * 64 + col] = S2[0];
          }
          continue;
        }
        const unsigned prow = (unsigned)((int)blockIdx.x * WM + wm) * 2u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, S1[0]), rst,
                                              sok ? ((prow + 0) * (unsigned)a.Cout + (unsigned)col) * 4u : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, S2[0]), rst,
                                              sok ? ((prow + 1) * (unsigned)a.Cout + (unsigned)col) * 4u : OOB, 0, 0);
      }
    }
  }
  if constexpr (STATS) {
    if (a.sacc) {
      __syncthreads();
      const float* sred = reinterpret_cast<const float*>(smem);
      if (tid < 2 * a.Cout) {  // host: Cout <= 64
        const int q = tid / a.Cout, c = tid - q * a.Cout;
        float t = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < WM; ++w4) t += sred[(w4 * 2 + q) * 64 + c];
        bn_acc_add(a.sacc, a.sR, a.Cout, (int)blockIdx.x & (a.sR - 1), q, c, t);
      }
    }
  }
  if (stamping) {
    __syncthreads();
    for (int i = tid; i < C::NWAVE * 96; i += C::NTHR) a.stamps[(i / 96) * 128 + i % 96] = (i % 96) < nst ? s_stamp[i] : 0ull;
  }
#endif
}

template <typename T, int KCH, int NB> constexpr int stream_per_cu() {
  constexpr int by_lds = (160 * 1024) / StreamCfg<T, KCH, NB>::SMEM;
  return by_lds < stream_wgs<KCH, NB>() ? (by_lds < 1 ? 1 : by_lds) : stream_wgs<KCH, NB>();
}
// persistent grid of a launch, and the number of BN partial rows it writes (4 per workgroup)
inline int stream_grid(int Cin, int Cout, int ntiles, bool pro) {
  const int kch = Cin / 32, nb = Cout / 32;
  int per_cu = kch == 1 ? (nb == 1 ? stream_per_cu<bf16, 1, 1>() : stream_per_cu<bf16, 1, 2>())
                              : (nb == 1 ? stream_per_cu<bf16, 2, 1>() : stream_per_cu<bf16, 2, 2>());
  if (pro && per_cu > 2) per_cu = 2;  // (the prologue build is compiled for two workgroups per CU)
  const int g = 256 * per_cu;
  return g < ntiles ? g : ntiles;
}
inline int stream_partials(int Cin, int Cout, int ntiles, bool pro) { return 4 * stream_grid(Cin, Cout, ntiles, pro); }

// applicability (beyond the plane kernel's own): 16-bit storage, Cin / Cout in {32, 64}, no load transform
inline bool stream_applicable(const cy_conv_desc* d) {
  const int Cin = d->C1 + d->C2;
  if (d->in_dtype == CY_F32 || d->mode1 == CY_SRC_POOL2 || d->prologue == 2) return false;
  if (d->prologue && !(d->C1 == 32 && d->C2 == 0 && d->Cout == 32)) return false;
  if ((Cin != 32 && Cin != 64) || (d->Cout != 32 && d->Cout != 64)) return false;
  if (d->C2 != 0 && (d->C1 != 32 || d->C2 != 32)) return false;  // concat: one chunk per source
  if (d->split_c > 0 && d->split_c % 32) return false;
  if (d->W % 14) return false;
  // buffer descriptors and 32-bit byte offsets: every tensor below 2 GiB
  const long eb = 2, opx = (long)d->N * d->H * d->W;
  const long px1 = d->mode1 == CY_SRC_UP2 ? (long)d->N * (d->H / 2) * (d->W / 2) : opx;
  const long lim = (1L << 31) - 1;
  if (px1 * d->ld1 * eb > lim || (d->C2 && opx * d->ld2 * eb > lim)) return false;
  if (opx * d->ldo * eb > lim || (d->split_c > 0 && opx * d->ldo2 * eb > lim)) return false;
  return true;
}

template <typename T, int KCH, int NB>
int launch_conv_stream(ConvArgs a, hipStream_t st) {
  using C = StreamCfg<T, KCH, NB>;
  a.tiles_w = a.W / C::TW;
  const int ntiles = cy_cdiv(a.NH, C::TH) * a.tiles_w;
  const int grid = stream_grid(a.C1 + a.C2, a.Cout, ntiles, a.prologue != 0);
  auto go = [&](auto kern) {
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
        return (int)CY_ERR_LAUNCH;
      attr_done = true;
    }
    a.stamps = g_conv_stamp_buf;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::SMEM, st, a);
    return (int)CY_OK;
  };
  int rc;
  if (a.prologue) {
    if constexpr (KCH == 1 && NB == 1)
      rc = (a.stats || a.sacc) ? go(conv3x3_stream_kernel<T, 1, 1, true, true>) : go(conv3x3_stream_kernel<T, 1, 1, false, true>);
    else
      return CY_ERR_SHAPE;
  } else {
    rc = (a.stats || a.sacc) ? go(conv3x3_stream_kernel<T, KCH, NB, true>) : go(conv3x3_stream_kernel<T, KCH, NB, false>);
  }
  if (rc != CY_OK) return rc;
  CY_CHECK_LAUNCH();
  return CY_OK;
}

template <typename T>
int dispatch_conv_stream(const ConvArgs& a, hipStream_t st) {
  const int kch = (a.C1 + a.C2) / 32, nb = a.Cout / 32;
  if (kch == 1 && nb == 1) return launch_conv_stream<T, 1, 1>(a, st);
  if (kch == 2 && nb == 1) return launch_conv_stream<T, 2, 1>(a, st);
  if (kch == 1 && nb == 2) return launch_conv_stream<T, 1, 2>(a, st);
  if (kch == 2 && nb == 2) return launch_conv_stream<T, 2, 2>(a, st);
  return CY_ERR_SHAPE;
}

}  // namespace
