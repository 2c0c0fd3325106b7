// Eight-wave form of the plane kernel (cy_conv_plane.h) for the 128-cout tiles of the deep layers.
//
// Why: with 256 positions x 128 couts per workgroup the weight slice of every (32-channel chunk, tap)
// is fetched L2 -> LDS once per 1024 MFMA cycles of a wave; two such workgroups per CU pull ~16 B/clk/CU
// of weights next to the activations, which is where the loaders of these kernels top out (~20 B/clk/CU
// measured in cy_conv_pc.h), and the 128-accumulator variant has no registers left to prefetch the next
// halo tile, so every chunk boundary is a synchronous 41 KB staging that only the CU's other workgroup
// hides.  Here ONE workgroup of eight waves owns 512 halo positions (32 rows x 16 columns, 448 outputs)
// x 128 couts -- all accumulators of the CU (256 KB of its 512 KB register file):
//   * the weight stream per MFMA halves (one fetch per 512 positions) and its LDS footprint is shared;
//   * chunks are 32 channels (A: 36 KB, double buffered), so the next chunk's halo tile is requested into
//     registers one 16-byte item per tap and committed to the OTHER buffer two taps later -- never
//     synchronously -- with the BN+ReLU prologue / upsample / concat addressing of the plane kernel on the
//     way (2x2-max layers stay on the plane kernel: four loads per item);
//   * weights go through a 3-slot per-tap ring, requested two taps ahead (one 16-byte item per thread per
//     tap) and committed at the START of the tap before the one that reads them;
//   * the two waves of every SIMD run one barrier interval apart (ping-pong): a tap is a MEMORY phase
//     (fragments LDS -> registers, ring / halo commits, next requests) and a COMPUTE phase (16 MFMAs), with a
//     raw s_barrier after each; while one wave computes, the SIMD's other wave is in its memory phase.
// Same packed weight image, same epilogue (register stores, BN partial sums, split-K slabs) and the same
// flattened-row tiling (tiles span image boundaries; flagged rows read the all-zero row) as the plane kernel.
#pragma once
#include "cy_conv_plane.h"

namespace {

template <typename T> struct Plane8Cfg {
  static constexpr int TH = 32, BN = 128, WGM = 4, WGN = 2, NTHR = 512;
  static constexpr int EPC = ElemTr<T>::EPC;
  static constexpr int KC = 32, CPP = 4, KS = 2;
  static constexpr int TW = 14, HP = 16;
  static constexpr int NPOS = (TH + 2) * HP;
  static constexpr int ZB = NPOS + 2;
  static constexpr int SKEW = 16 / CPP;
  static constexpr int APL = ((ZB + 18 - SKEW + 15) / 16) * 16 + SKEW;
  static constexpr int BPL = BN + SKEW;
  static constexpr int A_BYTES = CPP * APL * 16;
  static constexpr int B_BYTES = CPP * BPL * 16;
  static constexpr int M_REP = (TH * HP / 32) / WGM, N_REP = BN / (32 * WGN);
  static constexpr int NA = (NPOS * CPP + NTHR - 1) / NTHR;
  static constexpr int COEF_MAX = 512;  // prologue channels held in LDS
  static constexpr int NSLOT = 4;  // weight ring (the register path uses three of them)
  static constexpr int MAIN_BYTES = 2 * A_BYTES + NSLOT * B_BYTES;
  static constexpr int COEF_BYTES = 2 * COEF_MAX * 4;
  static constexpr int TAB_BYTES = (3 * TH + 4) * 4;
  static constexpr int STAMP_BYTES = 8 * 96 * 8;  // development aid: [wave][96] clock stamps of workgroup 0
  static constexpr int SMEM = MAIN_BYTES + COEF_BYTES + TAB_BYTES + 16 + STAMP_BYTES;
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  static_assert(EPC * CPP == KC && BN * CPP == NTHR, "one weight item per thread per tap");
  static_assert(NA + 2 <= 9, "the halo items of a chunk are requested / committed within its nine taps");
  static_assert(WGM * 2 * BN * 4 <= MAIN_BYTES, "statistics scratch");
};

__device__ __forceinline__ void plane8_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier; global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

static unsigned long long* g_p8_stamp_buf = nullptr;  // set by cy_debug_pc_stamps
#define P8_STAMP()                                                                              \
  do {                                                                                          \
    if (stamping && nst < 96) {                                                                 \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) s_stamp[wave * 96 + nst] = t__;                                            \
      ++nst;                                                                                    \
    }                                                                                           \
  } while (0)

// BDMA: the weights come from the stage-contiguous image of cy_conv_pc.h ([blk][16-channel stage][tap][2 planes][128
// rows][8 ch]) by LDS-DMA: one global_load_lds_dwordx4 per wave per tap moves 64 rows of one plane (1 KB,
// contiguous on both sides) -- no weight registers, no commit, no address arithmetic in the memory phase.
// ADMA (with BDMA; layers without a load transform -- no BN+ReLU prologue, no 2x2 max): the halo tile too goes
// global -> LDS by DMA (buffer_load_dwordx4 ... lds): a wave-instruction fills 64 consecutive positions of one
// plane (lane = position: four halo rows x 16 columns), gather addresses per lane (nearest-x2 upsampling and
// the second source of a concat are just other offsets), padding positions read out of range = zeros.  The
// memory phase then is 12 ds_read_b128 + one or two DMA issues and no VALU.
template <typename T, bool BDMA, bool ADMA = false>
__global__ void __launch_bounds__(512, 1)
    conv3x3_plane8_kernel(const ConvArgs a) {
  using C = Plane8Cfg<T>;
  using M = Mma<T>;
  constexpr int EPC = C::EPC, KC = C::KC, CPP = C::CPP, TH = C::TH, TW = C::TW, BN = C::BN;
  constexpr int M_REP = C::M_REP, N_REP = C::N_REP, NA = C::NA, WGM = C::WGM, WGN = C::WGN;
  constexpr int APLB = C::APL * 16, BPLB = C::BPL * 16;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_coef = reinterpret_cast<float*>(smem + C::MAIN_BYTES);
  int* s_row1 = reinterpret_cast<int*>(smem + C::MAIN_BYTES + C::COEF_BYTES);
  int* s_row2 = s_row1 + (TH + 2);
  int* s_flag = s_row2 + (TH + 2);
  unsigned long long* s_stamp = reinterpret_cast<unsigned long long*>(smem + C::MAIN_BYTES + C::COEF_BYTES + C::TAB_BYTES + 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  int tile = blockIdx.x;
  if (a.xcd_remap) {  // contiguous band of tiles per XCD (see conv3x3_plane_kernel)
    const int nt = gridDim.x, x = tile & 7, i = tile >> 3, q = nt >> 3, rr = nt & 7;
    tile = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
  }
  const int ct = tile % a.tiles_w;
  const int rt = tile / a.tiles_w;
  const int R0 = rt * TH, w0 = ct * TW;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int ncc = (Cin + KC - 1) / KC;
  const int cc0 = (ncc * (int)blockIdx.z) / a.ksplit;
  const int cc1 = (ncc * ((int)blockIdx.z + 1)) / a.ksplit;

  for (int idx = tid; idx < 2 * CPP * 18; idx += C::NTHR) {
    const int buf = idx / (CPP * 18), rem = idx % (CPP * 18);
    st16(smem + buf * C::A_BYTES + (rem / 18) * APLB + (C::ZB + rem % 18) * 16, u32x4{0u, 0u, 0u, 0u});
  }
  if (a.prologue)
    for (int c = tid; c < a.C1; c += C::NTHR) {
      s_coef[c] = a.scale[c];
      s_coef[C::COEF_MAX + c] = a.shift[c];
    }
  conv_row_tables(a, TH, R0, tid, s_row1, s_row2, s_flag, false);
  __syncthreads();

  // per-lane fragment bases (bytes inside a halo buffer, for dw = -1): the centre row's address per M block,
  // two flag bits per block (first / last row of its image: the upper / lower neighbour is the all-zero row)
  int amid[M_REP];
  unsigned aflag = 0;
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int ty = 2 * (wm * M_REP + m) + (r >> 4);
    amid[m] = ((ty + 1) * 16 + (r & 15)) * 16 + h * APLB;
    aflag |= (unsigned)s_flag[ty] << (2 * m);
  }
  const int azer = (C::ZB + (r & 15)) * 16 + h * APLB;
  auto aaddr = [&](int m, int d) {  // d: tap row 0..2
    if (d == 1) return amid[m];
    const bool z = (aflag >> (2 * m + (d == 0 ? 0 : 1))) & 1u;
    return z ? azer : amid[m] + (d == 0 ? -256 : 256);
  };
  int bbase[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) bbase[n] = ((wn * N_REP + n) * 32 + r) * 16 + h * BPLB;

  f32x16 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  auto frag = [&](const unsigned char* p) {
    typename M::Frag f;
    f.v = *reinterpret_cast<const decltype(f.v)*>(p);
    return f;
  };

  // ---- halo items: request (global -> registers) and commit (-> LDS planes of the buffer at offset `dst`) ----
  // thread -> (channel group ch, column hc) fixed, rows wave + 8 i for item i
  const int ch = tid & (CPP - 1);
  const int hc = (tid / CPP) & 15;
  const int wcol = w0 - 1 + hc;
  const bool col_ok = wcol >= 0 && wcol < a.W;
  const unsigned char* s1b = reinterpret_cast<const unsigned char*>(a.src1);
  const unsigned char* s2b = reinterpret_cast<const unsigned char*>(a.src2);
  u32x4 areg[2];
  unsigned aok = 0;  // bit s: areg[s] holds loaded data (else the item is zero padding)
  // rows of this wave's items (wave-uniform): pixel index of (row, w = 0) in either source, or -1
  int arow1[NA], arow2[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int hr = wave + 8 * i;
    arow1[i] = hr < TH + 2 ? __builtin_amdgcn_readfirstlane(s_row1[hr]) : -1;
    arow2[i] = hr < TH + 2 ? __builtin_amdgcn_readfirstlane(s_row2[hr]) : -1;
  }
  // (the load is issued unconditionally -- lanes / items without a source read offset 0 and are zeroed at
  //  commit time -- so that the compiler can count outstanding loads exactly: behind a branch every later
  //  s_waitcnt becomes vmcnt(0) and the weight commit waits for the youngest request)
  auto a_request = [&](int c0, auto ITEM) {
    constexpr int i = decltype(ITEM)::value;
    constexpr int s = i & 1;
    const bool in2 = c0 >= a.C1;  // host: C1 % 32 == 0 when there is a second source
    const int rp = in2 ? arow2[i] : arow1[i];
    const bool ok = rp >= 0 && col_ok && c0 + ch * EPC < Cin;
    const unsigned char* base = in2 ? s2b : s1b;
    const int ld = in2 ? a.ld2 : a.ld1;
    const int wsh = (!in2 && a.mode1 == CY_SRC_UP2) ? 1 : 0;
    const unsigned off = ((unsigned)(rp + (wcol >> wsh)) * (unsigned)ld + (unsigned)((in2 ? c0 - a.C1 : c0) + ch * EPC)) *
                         (unsigned)sizeof(T);  // host: tensors below 4 GiB
    areg[s] = ld16(base + (ok ? off : 0u));
    aok = ok ? (aok | (1u << s)) : (aok & ~(1u << s));
  };
  auto a_commit = [&](int c0, auto ITEM, int dst) {
    constexpr int i = decltype(ITEM)::value;
    constexpr int s = i & 1;
    const int hr = wave + 8 * i;
    if (hr >= TH + 2) return;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((aok >> s) & 1u) {
      v = areg[s];
      const int cabs = c0 + ch * EPC;
      if (a.prologue && cabs < a.C1) {
        float f[EPC];
        Chunk<T>::unpack(v, f);
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_coef + cabs);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(s_coef + cabs + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[j] = fmaxf(fmaf(sc0[j], f[j], sh0[j]), 0.f);
          f[4 + j] = fmaxf(fmaf(sc1[j], f[4 + j], sh1[j]), 0.f);
        }
        v = Chunk<T>::pack(f);
      }
    }
    st16(smem + dst + ch * APLB + 16 + (hr * 16 + hc) * 16, v);  // position index = 1 + hr*16 + hc
  };

  // ---- ADMA: per-lane byte offsets of this wave's position groups (group g = (wave >> 2) + 2 i, plane = wave & 3)
  constexpr unsigned OOB = 0xffffff00u;
  unsigned poff1[NA], poff2[NA];
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-descriptor type and builtins exist in the device pass only)
  __amdgpu_buffer_rsrc_t rs1, rs2;
  if constexpr (ADMA) {
    const int pl = wave & 3;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int lin = ((wave >> 2) + 2 * i) * 64 + lane;
      const int hr = lin >> 4, w = w0 - 1 + (lin & 15);
      poff1[i] = poff2[i] = OOB;
      if (lin < C::NPOS && w >= 0 && w < a.W) {
        const int r1 = s_row1[hr], r2 = s_row2[hr];
        const int wsh = a.mode1 == CY_SRC_UP2 ? 1 : 0;
        if (r1 >= 0) poff1[i] = ((unsigned)(r1 + (w >> wsh)) * (unsigned)a.ld1 + pl * EPC) * (unsigned)sizeof(T);
        if (r2 >= 0 && a.C2 > 0) poff2[i] = ((unsigned)(r2 + w) * (unsigned)a.ld2 + pl * EPC) * (unsigned)sizeof(T);
      }
    }
    // (descriptors from wave-uniform scalars only: no waterfall loops around the loads)
    const unsigned long long b1 = (unsigned long long)a.src1, b2 = (unsigned long long)(a.src2 ? a.src2 : a.src1);
    const unsigned lo1 = __builtin_amdgcn_readfirstlane((unsigned)b1), hi1 = __builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32));
    const unsigned lo2 = __builtin_amdgcn_readfirstlane((unsigned)b2), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(b2 >> 32));
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi1 << 32) | lo1), (short)0, (int)a.bytes1, 0x00020000);
    rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi2 << 32) | lo2), (short)0, (int)a.bytes2, 0x00020000);
  }
#endif
  // item i of chunk c0 -> halo buffer at offset dst; returns whether this wave has such an item
  auto a_dma = [&](int c0, auto ITEM, int dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int i = decltype(ITEM)::value;
    const int g = (wave >> 2) + 2 * i;  // wave-uniform
    if (g * 64 >= C::NPOS) return;
    const bool in2 = c0 >= a.C1;
    const unsigned soff = (unsigned)((in2 ? c0 - a.C1 : c0) * (int)sizeof(T));
    auto* ldst = (__attribute__((address_space(3))) void*)(smem + dst + (wave & 3) * APLB + 16 + g * 1024);
    if (g * 64 + 64 <= C::NPOS || lane < C::NPOS - g * 64) {  // the last group is half a wave
      if (in2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, ldst, 16, poff2[i], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, ldst, 16, poff1[i], soff, 0, 0);
    }
#endif
  };

  // ---- weights: one 16-byte item per thread per (chunk, tap): uniform base + one 32-bit per-thread offset ----
  const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w);
  const int brow = tid / CPP;
  const unsigned bvoff = (unsigned)(((n0 + brow) * a.w_ci_pad + ch * EPC) * (int)sizeof(T));
  const size_t btap = (size_t)a.w_co_pad * a.w_ci_pad * sizeof(T);
  const int bdst = ch * BPLB + brow * 16;
  auto b_request = [&](int cc, int tap) {
    return ld16(wp + ((size_t)tap * btap + (size_t)cc * (KC * sizeof(T))) + bvoff);
  };
  // BDMA: wave -> (plane = wave % 4, rows 64 * (wave / 4) ..): 1 KB of the image -> 1 KB of the ring slot
  const int nst16 = (Cin + 15) / 16;
  const unsigned char* wdma = wp + (size_t)blockIdx.y * nst16 * (9 * 4096) + (wave & 1) * 2048 + (wave >> 2) * 1024 + lane * 16;
  const int ddst = (wave & 3) * BPLB + (wave >> 2) * 1024;
  auto b_dma = [&](int cc, int tap, int slot) {
    const unsigned char* src = wdma + ((size_t)(2 * cc + ((wave >> 1) & 1)) * 9 + tap) * 4096;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + slot + ddst), 16, 0, 0);
  };

  // ---- prologue: first chunk's halo tile and the first tap's weights synchronously ----------------------------
  // (byte offsets into smem, not pointers: rotating pointers would lose their LDS address space)
  int sA_cur = 0, sA_nxt = C::A_BYTES;
  int slot_cur = 2 * C::A_BYTES, slot_nxt = 2 * C::A_BYTES + C::B_BYTES, slot_nn = 2 * C::A_BYTES + 2 * C::B_BYTES;
  int slot_n3 = 2 * C::A_BYTES + 3 * C::B_BYTES;  // BDMA only
  const int it0 = cc0 * 9, nit = cc1 * 9;
  u32x4 bq[3];
  if constexpr (BDMA) {
    b_dma(cc0, 0, slot_cur);
    b_dma(cc0, 1, slot_nxt);
    b_dma(cc0, 2, slot_nn);
  }
  {
    u32x4 b0 = {0u, 0u, 0u, 0u};
    if constexpr (!BDMA) b0 = b_request(cc0, 0);
    if constexpr (ADMA) {
      plane_static_for<0, NA>([&](auto J) { a_dma(cc0 * KC, J, sA_cur); });
    } else {
      plane_static_for<0, (NA + 1) / 2>([&](auto P) {
        constexpr int j = decltype(P)::value * 2;
        a_request(cc0 * KC, TapC<j>{});
        if constexpr (j + 1 < NA) a_request(cc0 * KC, TapC<j + 1>{});
        a_commit(cc0 * KC, TapC<j>{}, sA_cur);
        if constexpr (j + 1 < NA) a_commit(cc0 * KC, TapC<j + 1>{}, sA_cur);
      });
    }
    if constexpr (!BDMA) st16(smem + slot_cur + bdst, b0);
  }
  // register path: the item of step j sits in bq[j % 3] = bq[tap % 3] (nine taps per chunk), requested at step
  // j-3 and committed at step j-1: static register names, no copies (a copy would wait for the load it moves)
  if constexpr (!BDMA) {
    bq[1] = b_request(cc0, 1);
    bq[2] = b_request(cc0, 2);
  }
  __syncthreads();  // (drains vmcnt: the first two DMA slices have landed)

  // Ping-pong: waves 0-3 (one per SIMD) and waves 4-7 run the same program one barrier interval apart, so
  // that on every SIMD one wave is in its MEMORY phase (fragments of its next tap LDS -> registers, this tap's
  // weight item -> ring, next requests, halo item commit) while the other is in its COMPUTE phase (the 16
  // MFMAs of a tap, nothing else): the matrix pipe never waits for a memory phase as long as that is shorter
  // than 512 cycles.  Two s_barriers per tap.  Hazards with the two sets an interval apart: the weights of
  // step j are written in the memory phase of step j-1 (last: interval 2j-1, late set) and first read in the
  // memory phase of step j (interval 2j, early set); their slot is rewritten with step j+3's weights in the
  // memory phase of step j+2 (first: interval 2j+4), after the late set's reads (interval 2j+1).
  const int late = wave >> 2;
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
  int nst = 0;
#ifdef P8_LATENCY_PROBE
  if (stamping) {  // load-to-use latency of one weight item: cold-ish, then warm, then a far tap
    P8_STAMP();
    u32x4 q = b_request(cc0, 3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q)::"memory");
    P8_STAMP();
    q = b_request(cc0, 3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q)::"memory");
    P8_STAMP();
    q = b_request(cc0 + 1, 5);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q)::"memory");
    P8_STAMP();
    q = b_request(cc0 + 1, 5);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q)::"memory");
    P8_STAMP();
    nst = 8;
  }
#endif
  typename M::Frag fa[2][M_REP], fb[2][N_REP];
  if (late) asm volatile("s_barrier" ::: "memory");

  for (int cc = cc0; cc < cc1; ++cc) {
    auto step = [&](auto TAP) {
      constexpr int t = decltype(TAP)::value;
      constexpr int d = t / 3, dw = t % 3 - 1;
      const int it = cc * 9 + t;
      // ---------------- memory phase of step it ----------------
      // Order matters (stamps): the 12 fragment reads go FIRST -- they keep the LDS busy for ~400 cycles (four
      // waves' 48 KB at 128 B/clk) while this wave's two vector-memory instructions, whose issue alone costs
      // 150-350 cycles each beside the partner's MFMA stream, queue behind them; in the opposite order the
      // two times add up (phase 1000 cycles against the partner's 650 of MFMAs).
      P8_STAMP();
      {
        const unsigned char* sAt = smem + sA_cur + (dw + 1) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int n = 0; n < N_REP; ++n) fb[ks][n] = frag(smem + slot_cur + bbase[n] + ks * 2 * BPLB);
#pragma unroll
          for (int m = 0; m < M_REP; ++m) fa[ks][m] = frag(sAt + aaddr(m, d) + ks * 2 * APLB);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // (unconditional, with clamped chunk indices past the end: see a_request)
      if constexpr (BDMA) {
        // slice of step it+3 -> the slot step it-1 read (four slots); it has landed when this wave passes the
        // barrier that ends its memory phase of step it+2 (the counted wait below leaves two phases' operations
        // in flight), one barrier before any wave reads it
        const int ccq = cc + (t + 3) / 9;
        b_dma(ccq < ncc ? ccq : ncc - 1, (t + 3) % 9, slot_n3);
      } else {
        st16(smem + slot_nxt + bdst, bq[(t + 1) % 3]);
        const int ccq = cc + (t + 3) / 9;
        bq[t % 3] = b_request(ccq < ncc ? ccq : ncc - 1, (t + 3) % 9);
      }
      {  // next chunk's halo tile: item j requested at tap j, committed at tap j + 2 (other buffer)
        const int c0n = (cc + 1 < ncc ? cc + 1 : ncc - 1) * KC;
        if constexpr (ADMA) {
          if constexpr (t < NA) a_dma(c0n, TapC<(t < NA ? t : 0)>{}, sA_nxt);
        } else {
          if constexpr (t >= 2 && t - 2 < NA) a_commit(c0n, TapC<(t >= 2 ? t - 2 : 0)>{}, sA_nxt);
          if constexpr (t < NA) a_request(c0n, TapC<(t < NA ? t : 0)>{});
        }
      }
      if constexpr (BDMA) {  // everything older than this and the previous phase's vector-memory operations has landed
        constexpr int tp = (t + 8) % 9;
        auto has_a = [&](int tt) { return ADMA ? (tt < NA && ((wave >> 2) + 2 * tt) * 64 < C::NPOS) : tt < NA; };
        const int k = 2 + (has_a(t) ? 1 : 0) + (has_a(tp) ? 1 : 0);  // wave-uniform
        if (k == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P8_STAMP();
      plane8_barrier();
      P8_STAMP();
      // ---------------- compute phase of step it ----------------
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
#pragma unroll
          for (int n = 0; n < N_REP; ++n) M::mma(fb[ks][n], fa[ks][m], acc[m][n]);  // rows = couts
      __builtin_amdgcn_sched_barrier(0);
      P8_STAMP();
      asm volatile("s_barrier" ::: "memory");
      const int sl = slot_cur;
      slot_cur = slot_nxt;
      slot_nxt = slot_nn;
      if constexpr (BDMA) {
        slot_nn = slot_n3;
        slot_n3 = sl;
      } else {
        slot_nn = sl;
      }
    };
    CY_NINE_TAPS(step(TapC<TAP_>{}));
    const int sa = sA_cur;
    sA_cur = sA_nxt;
    sA_nxt = sa;
  }
  if (!late) asm volatile("s_barrier" ::: "memory");
  __syncthreads();  // the statistics scratch aliases the operand buffers
  if (stamping)
    for (int i = tid; i < 8 * 96; i += C::NTHR) a.stamps[(i / 96) * 128 + i % 96] = (i % 96) < nst ? s_stamp[i] : 0ull;

  // ---------------- epilogue (as conv3x3_plane_kernel): accumulators -> NHWC from registers ----------------
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
  const bool do_stats = a.stats != nullptr;
  T* o1 = reinterpret_cast<T*>(a.out);
  T* o2 = reinterpret_cast<T*>(a.out2);
#pragma unroll
  for (int n = 0; n < N_REP; ++n) {
    float s1v[16], s2v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1v[i] = s2v[i] = 0.f;
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      int R, w;
      const bool ok = position(m, R, w);
      const size_t gp = (size_t)R * a.W + w;
      u32x2 packed[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pk[j] = from_f32<T>(acc[m][n][4 * g + j]);
          if (do_stats && ok) {
            const float qv = to_f32<T>(pk[j]);
            s1v[4 * g + j] += qv;
            s2v[4 * g + j] += qv * qv;
          }
        }
        packed[g] = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {  // half-wave swaps pair the 8-byte channel runs into 16-byte stores
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
    if (do_stats) {  // reduce-scatter over the 32 lanes of each half (see conv3x3_plane_kernel)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool up = (lane & 16) != 0;
        const float snd1 = up ? s1v[i] : s1v[i + 8], snd2 = up ? s2v[i] : s2v[i + 8];
        const float kp1 = up ? s1v[i + 8] : s1v[i], kp2 = up ? s2v[i + 8] : s2v[i];
        s1v[i] = kp1 + __shfl_xor(snd1, 16, 64);
        s2v[i] = kp2 + __shfl_xor(snd2, 16, 64);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool up = (lane & 8) != 0;
        const float snd1 = up ? s1v[i] : s1v[i + 4], snd2 = up ? s2v[i] : s2v[i + 4];
        const float kp1 = up ? s1v[i + 4] : s1v[i], kp2 = up ? s2v[i + 4] : s2v[i];
        s1v[i] = kp1 + __shfl_xor(snd1, 8, 64);
        s2v[i] = kp2 + __shfl_xor(snd2, 8, 64);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool up = (lane & 4) != 0;
        const float snd1 = up ? s1v[i] : s1v[i + 2], snd2 = up ? s2v[i] : s2v[i + 2];
        const float kp1 = up ? s1v[i + 2] : s1v[i], kp2 = up ? s2v[i + 2] : s2v[i];
        s1v[i] = kp1 + __shfl_xor(snd1, 4, 64);
        s2v[i] = kp2 + __shfl_xor(snd2, 4, 64);
      }
      {
        const bool up = (lane & 2) != 0;
        const float snd1 = up ? s1v[0] : s1v[1], snd2 = up ? s2v[0] : s2v[1];
        const float kp1 = up ? s1v[1] : s1v[0], kp2 = up ? s2v[1] : s2v[0];
        s1v[0] = kp1 + __shfl_xor(snd1, 2, 64);
        s2v[0] = kp2 + __shfl_xor(snd2, 2, 64);
      }
      s1v[0] += __shfl_xor(s1v[0], 1, 64);
      s2v[0] += __shfl_xor(s2v[0], 1, 64);
      if ((lane & 1) == 0) {
        const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int col = (wn * N_REP + n) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        sstat[(wm * 2 + 0) * BN + col] = s1v[0];
        sstat[(wm * 2 + 1) * BN + col] = s2v[0];
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
      a.stats[((size_t)tile * 2 + 0) * a.Cout + n0 + tid] = t1;
      a.stats[((size_t)tile * 2 + 1) * a.Cout + n0 + tid] = t2;
    }
  }
}

// applicability beyond the plane kernel's: 16-bit storage, prologue channels fit the LDS table, pool => one source
inline bool plane8_applicable(const ConvArgs& a) {
  if (a.prologue && a.C1 > Plane8Cfg<bf16>::COEF_MAX) return false;
  if (a.mode1 == CY_SRC_POOL2) return false;           // 2x2 max on load: four loads per item (plane kernel)
  if (a.C2 != 0 && a.C1 % Plane8Cfg<bf16>::KC) return false;  // a chunk reads one source
  return true;
}

static const void* g_p8_w_dma = nullptr;  // development aid (cy_debug_p8_weights): image of cy_conv3x3_pc_pack

template <typename T, bool BDMA = false, bool ADMA = false>
int launch_conv_plane8(ConvArgs a, hipStream_t st) {
  using C = Plane8Cfg<T>;
  if constexpr (!BDMA) {
    if (g_p8_w_dma && (a.C1 + a.C2) % 32 == 0) {
      a.w = g_p8_w_dma;
      static const int adma = [] {
        const char* e = getenv("CY_P8_ADMA");
        return e ? atoi(e) : 1;
      }();
      if (adma && !a.prologue && a.bytes1 > 0 && a.bytes1 < (1ll << 31) && a.bytes2 < (1ll << 31))
        return launch_conv_plane8<T, true, true>(a, st);
      return launch_conv_plane8<T, true, false>(a, st);
    }
  }
  auto kern = conv3x3_plane8_kernel<T, BDMA, ADMA>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  if (a.W % C::TW != 0 || !plane8_applicable(a)) return CY_ERR_SHAPE;
  a.tiles_w = a.W / C::TW;
  static const int dbg = [] {  // development aid: what-if switches (1: no weight requests, 2: no halo prefetch)
    const char* e = getenv("CY_P8_DEBUG");
    return e ? atoi(e) : 0;
  }();
  a.full_tiles = dbg;
  static const int xcd = [] {
    const char* e = getenv("CY_PLANE_XCD");
    return e ? atoi(e) : 1;
  }();
  a.xcd_remap = xcd;
  a.stamps = g_p8_stamp_buf;
  dim3 grid(cy_cdiv(a.NH, C::TH) * a.tiles_w, cy_cdiv(a.Cout, C::BN), a.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), C::SMEM, st, a);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // namespace
