// "Flow" form of the plane kernel (cy_conv_plane.h) for the MFMA-bound layers (Cout >= 64, 16-bit storage):
// the same GEMM view (M = flattened halo grid, a tap = one uniform address shift, LDS planes
// [16-byte channel group][position]) and the same register epilogue, but NO register staging and NO per-tap
// barrier.  Where a workgroup's time went in the plane kernel's 128-cout build: nine __syncthreads() per
// 32-channel chunk, each draining the next tap's weight request (vmcnt(0)) and followed by a ds_write commit,
// plus a synchronous 41 KB halo staging at every chunk boundary -- 40 cycles per MFMA in the tap loop and two
// workgroups per CU needed just to cover each other's commits.  Here
//   * ONE workgroup of eight waves per CU owns 512 (32 x 16) or 1024 (64 x 16) positions x 128 / 64 couts --
//     all 256 KB of accumulators a CU can hold, so the weight stream per MFMA is half the plane kernel's;
//   * a stage = one 16-channel chunk: the halo tile (2 planes) AND all nine taps of the weights (18 slices of
//     BN rows x 16 B), both brought by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write, no commit
//     phase): activations with per-lane gather offsets (padding = out of range = zeros; nearest-x2 and the
//     second source of a concat are other offsets), weights from a stage-contiguous image
//     [64-cout block][16-channel chunk][tap][plane][64 rows][8 channels] in 1 KB pieces;
//   * two stages: the DMA of chunk c+1 is issued right behind the barrier that opens chunk c, spread over
//     the taps, and has a whole chunk of MFMAs (2304 cycles per wave, two waves per SIMD) to land; ONE
//     s_barrier per chunk (per 72 MFMAs of a wave) instead of nine;
//   * the BN+ReLU prologue is an in-place LDS pass over the items the wave's own DMA brought (fixed eight
//     channels per wave), between its vmcnt wait and the barrier; 2x2-max-on-load layers stay on the plane kernel;
//   * W16: widths that are multiples of 16 (224, 112 and every level of a 256 x 256 input) use 16-wide tiles on
//     an 18-wide halo pitch -- no thrown-away accumulator columns (the 16 x 14 tiling wastes 2 of 16) -- with
//     the lanes of a 32-position block permuted so that each of ds_read_b128's 16-lane groups reads one halo row
//     (256 contiguous bytes: conflict-free on any pitch); the epilogue stores per lane, so the permutation is free.
#pragma once
#include "cy_conv_stream.h"  // (g_conv_stamp_buf)

namespace {

template <typename T, int TH_, int BN_, int WGM_, int WGN_, bool W16_> struct FlowCfg {
  static constexpr int TH = TH_, BN = BN_, WGM = WGM_, WGN = WGN_;
  static constexpr bool W16 = W16_;
  static constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  static constexpr int EPC = ElemTr<T>::EPC;
  static constexpr int KC = 16, CPP = 2;
  static constexpr int HP = W16 ? 18 : 16, TW = W16 ? 16 : 14;
  static constexpr int NPOS = (TH + 2) * HP;   // halo positions; LDS position index = 1 + hr * HP + hc
  static constexpr int ZB = NPOS + 2;          // all-zero row (18 positions + margin)
  static constexpr int APL = ((ZB + 20 + 15) / 16) * 16;
  static constexpr int APLB = APL * 16;        // bytes of one activation plane
  static constexpr int A_BYTES = CPP * APLB;
  static constexpr int BPLB = BN * 16;         // one (tap, plane) slice of the weights
  static constexpr int B_BYTES = 9 * CPP * BPLB;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int MT = TH / 2;            // 32-position blocks (two rows of 16)
  static constexpr int M_REP = MT / WGM, N_REP = BN / (32 * WGN);
  static constexpr int NGA = (NPOS + 63) / 64;                 // halo position groups of 64
  static constexpr int NAI = (NGA + NW / 2 - 1) / (NW / 2);    // ... per wave (wave -> plane wave & 1), at most
  static constexpr int NBITEMS = 9 * CPP * (BN / 64);          // 1 KB weight pieces per chunk
  static constexpr int NBI = (NBITEMS + NW - 1) / NW;
  static constexpr int COEF_MAX = NTHR < 512 ? NTHR : 512;  // prologue channels held in LDS (one set per thread)
  static constexpr int COEF_BYTES = 2 * COEF_MAX * 4;       // rows: scale, shift
  static constexpr int TAB_BYTES = ((3 * TH + 4) * 4 + 15) & ~15;
  static constexpr int SMEM = 2 * STAGE + COEF_BYTES + TAB_BYTES;
  // backward prologue (prologue == 2): one more halo tile buffer for y (single: it is consumed by the in-place pass that
  // opens the chunk, before the next chunk's request is issued), behind everything else
  static constexpr int Y_OFF = SMEM;
  static constexpr int K_OFF = SMEM + A_BYTES;  // ... and the coefficient rows k1, k0
  static constexpr int SMEM_BWD = SMEM + A_BYTES + COEF_BYTES;
  static constexpr bool BWD_OK = SMEM_BWD <= 160 * 1024 && (NW == 8 || 2 * SMEM_BWD <= 160 * 1024);
  static_assert(sizeof(T) == 2 && EPC == 8, "16-bit storage types only");
  static_assert(NW % 2 == 0 && MT % WGM == 0 && BN % (32 * WGN) == 0 && BN % 64 == 0, "wave split");
  static_assert(NAI <= 9 && NBI <= 9, "one DMA slot per tap");
  static_assert(WGM * 2 * BN * 4 <= 16384 && 16384 + 4 * BN * 4 <= STAGE, "statistics scratch, then the epilogue's coefficient rows");
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(COEF_MAX <= NTHR, "one prologue coefficient pair per thread");
  static_assert((4 * COEF_MAX + 1) * 8 <= B_BYTES, "the accumulator sums of the fold fit the second stage's weight region");
};

// elements of the stage-contiguous weight image of a (Cout, Cin) kernel (0: this geometry has none)
inline long flow_image_elems(int Cout, int Cin) {
  if (Cout % 64 || Cin % 16) return 0;
  return 9L * Cout * Cin;
}

// development aid (-DCY_FLOW_STAMPS, tools/flow_stamps.py): every workgroup records {wall clock (100 MHz) at
// start, shader clock at start / tables ready / first chunk landed / loop end / epilogue issued, wall clock at
// end, XCC id} into a.stamps[workgroup][8]
#ifdef CY_FLOW_STAMPS
#define FLOW_STAMP(K, V)                                                                    \
  do {                                                                                      \
    if (a.stamps != nullptr && tid == 0)                                                    \
      a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (K)] = (V); \
  } while (0)
#else
#define FLOW_STAMP(K, V) do { } while (0)
#endif

// BWD: the instantiation for prologue == 2 (the BatchNorm + ReLU backward in the load path); its own build so that the
// forward kernels keep their register budget
// MODE 2: the instantiation whose epilogue adds the backward sums of the BatchNorm behind its output (a.dz_y) -- its own
// build for the same reason
template <typename T, int TH, int BN, int WGM, int WGN, bool W16, int MODE = 0>
__global__ void __launch_bounds__(64 * WGM * WGN, 2)
    conv3x3_flow_kernel(const ConvArgs a) {
  constexpr bool BWD = MODE == 1, DZK = MODE == 2;
#if defined(__HIP_DEVICE_COMPILE__)  // (buffer-descriptor type and builtins exist in the device pass only)
  using C = FlowCfg<T, TH, BN, WGM, WGN, W16>;
  using M = Mma<T>;
  constexpr int EPC = C::EPC, KC = C::KC, HP = C::HP, TW = C::TW, NW = C::NW;
  constexpr int M_REP = C::M_REP, N_REP = C::N_REP, NAI = C::NAI, NBI = C::NBI;
  constexpr int APLB = C::APLB, BPLB = C::BPLB;
  constexpr unsigned OOB = 0x80000000u;  // (every tensor is below 2 GiB: out of range whatever the scalar offset adds)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_coef = reinterpret_cast<float*>(smem + 2 * C::STAGE);
  int* s_row1 = reinterpret_cast<int*>(smem + 2 * C::STAGE + C::COEF_BYTES);
  int* s_row2 = s_row1 + (TH + 2);
  int* s_flag = s_row2 + (TH + 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;
  FLOW_STAMP(0, __builtin_amdgcn_s_memrealtime());
  FLOW_STAMP(1, __builtin_amdgcn_s_memtime());
  // position of lane r inside a 32-position block: (row 0 / 1, column 0..15)
  int prow, pcol;
  if constexpr (W16) {
    // ds_read_b128 serves lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} together: each group = one row
    const int grp = (r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28;
    prow = grp;
    pcol = grp ? (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16)) : (r < 4 ? r : (r < 16 ? r - 8 : r - 12));
  } else {
    prow = r >> 4;
    pcol = r & 15;
  }

  // One grid dimension over (tile, cout block), the cout block innermost: workgroups are dealt round robin over the 8
  // XCDs, and remapped every XCD owns a contiguous band of that sequence (see conv3x3_plane_kernel) -- the cout blocks
  // of one tile are neighbours in one band, so the tile's activations come into ONE L2, once.  (With the cout block
  // on blockIdx.y its workgroups were a whole grid row apart: C5's traffic was 1.45 x algorithmic with 64-cout tiles.)
  const int nbk = a.Cout / BN;
  int id = blockIdx.x;
  if (a.xcd_remap) {
    const int nt = gridDim.x, x = id & 7, i = id >> 3, q = nt >> 3, rr = nt & 7;
    id = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
  }
  const int tile = id / nbk, nblk = id - tile * nbk;
  const int ct = tile % a.tiles_w;
  const int rt = tile / a.tiles_w;
  const int R0 = rt * TH, w0 = ct * TW;
  const int n0 = nblk * BN;
  const int Cin = a.C1 + a.C2;
  const int ncc = Cin / KC;  // host: Cin % 16 == 0
  const int cc0 = (ncc * (int)blockIdx.z) / a.ksplit;
  const int cc1 = (ncc * ((int)blockIdx.z + 1)) / a.ksplit;

  // ---- descriptors (from wave-uniform scalars only: no waterfall loops around the loads) ---------------------
  auto make_rsrc = [&](const void* p, long long bytes) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.src1, a.bytes1);
  const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(a.src2 ? a.src2 : a.src1, a.src2 ? a.bytes2 : 0);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wflow, a.bytes_w);

  // ---- weights: piece j = wave + NW i of a chunk = (tap * 2 + plane, 64-row half) ------------------------------
  const unsigned wlane = (unsigned)lane * 16u;
  const unsigned wblk = (unsigned)(nblk * (BN / 64)) * (unsigned)ncc * 18432u;
  auto b_dma = [&](int cc, auto I, int st) {
    constexpr int i = decltype(I)::value;
    const int j = wave + NW * i;  // wave-uniform
    if (j >= C::NBITEMS) return;
    const int tp = j / (BN / 64), half = j % (BN / 64);
    const unsigned soff = wblk + ((unsigned)half * (unsigned)ncc + (unsigned)cc) * 18432u + (unsigned)tp * 1024u;
    auto* dst = (__attribute__((address_space(3))) void*)(smem + st + C::A_BYTES + tp * BPLB + half * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, wlane, soff, 0, 0);
  };
  // BN+ReLU coefficients of the previous layer: requested BEFORE any DMA (vmcnt retires in order: a load behind the
  // DMAs would wait for all of them), written to LDS behind the halo requests
  constexpr bool pro2 = BWD;  // BatchNorm + ReLU BACKWARD in the load path (source 1 = dA, a.ysrc = y); host: a.prologue == 2
  float csc = 0.f, csh = 0.f;
  if (a.prologue && (pro2 || !a.fold.acc) && tid < a.C1) {  // host: C1 <= COEF_MAX <= NTHR
    csc = pro2 ? a.bfold.coef[tid] : a.scale[tid];
    csh = pro2 ? a.bfold.coef[a.C1 + tid] : a.shift[tid];
  }
  // backward sums in the epilogue (a.dz_y): this workgroup's couts' forward coefficients, requested before any DMA
  const bool dz_on = DZK && a.ksplit == 1 && n0 >= a.dz_c0 && n0 < a.dz_c0 + a.dz_C;  // wave-uniform; host: block-aligned
  float dzc[4] = {0.f, 0.f, 0.f, 0.f};
  if (dz_on && tid < BN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dzc[q] = a.dz_coef[q * a.dz_C + (n0 - a.dz_c0) + tid];
  }
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(pro2 ? a.ysrc : a.src1, pro2 ? a.bytes_y : 0);
  const __amdgpu_buffer_rsrc_t rsd = make_rsrc(pro2 ? a.dy_out : a.out, pro2 ? a.bytes_dy : 0);
  // (the first chunk's weights need nothing from the tables below: request them first)
  plane_static_for<0, NBI>([&](auto I) { b_dma(cc0, I, 0); });

  for (int idx = tid; idx < 2 * C::CPP * 20; idx += C::NTHR) {  // the zero rows of both stages (never touched by DMA)
    const int pl = idx / 20, z = idx % 20;
    st16(smem + (pl >> 1) * C::STAGE + (pl & 1) * APLB + (C::ZB - 1 + z) * 16, u32x4{0u, 0u, 0u, 0u});
  }
  conv_row_tables(a, TH, R0, tid, s_row1, s_row2, s_flag, false);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the DMA stays in flight)
  FLOW_STAMP(2, __builtin_amdgcn_s_memtime());

  // ---- halo: item i of this wave = position group g = (wave >> 1) + (NW / 2) i of plane (wave & 1) ----------
  const int apl = wave & 1;
  unsigned poff1[NAI], poff2[NAI];
#pragma unroll
  for (int i = 0; i < NAI; ++i) {
    const int lin = ((wave >> 1) + (NW / 2) * i) * 64 + lane;
    const int hr = lin / HP, hc = lin - hr * HP;
    const int w = w0 - 1 + hc;
    poff1[i] = poff2[i] = OOB;
    if (lin < C::NPOS && w >= 0 && w < a.W) {
      const int r1 = s_row1[hr], r2 = s_row2[hr];
      const int wsh = a.mode1 == CY_SRC_UP2 ? 1 : 0;
      if (r1 >= 0) poff1[i] = ((unsigned)(r1 + (w >> wsh)) * (unsigned)a.ld1 + (unsigned)(apl * EPC)) * (unsigned)sizeof(T);
      if (r2 >= 0 && a.C2 > 0) poff2[i] = ((unsigned)(r2 + w) * (unsigned)a.ld2 + (unsigned)(apl * EPC)) * (unsigned)sizeof(T);
    }
  }
  auto a_dma = [&](int cc, auto I, int st) {
    constexpr int i = decltype(I)::value;
    const int g = (wave >> 1) + (NW / 2) * i;  // wave-uniform
    if (g >= C::NGA) return;
    const int c0 = cc * KC;
    const bool in2 = c0 >= a.C1;  // host: C1 % 16 == 0
    const unsigned soff = (unsigned)((in2 ? c0 - a.C1 : c0) * (int)sizeof(T));
    auto* dst = (__attribute__((address_space(3))) void*)(smem + st + apl * APLB + 16 + g * 1024);
    if (g * 64 + 64 <= C::NPOS || lane < C::NPOS - g * 64) {  // the last group may be part of a wave
      if (in2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, dst, 16, poff2[i], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, poff1[i], soff, 0, 0);
      if constexpr (pro2) {  // the same items of y (same geometry and pitch: same offsets) into the y buffer
        auto* dsty = (__attribute__((address_space(3))) void*)(smem + C::Y_OFF + apl * APLB + 16 + g * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, dsty, 16, poff1[i], soff, 0, 0);
      }
    }
  };
  // backward prologue: which of this lane's items are OUTPUT positions of the tile (not halo): those it also writes to
  // dy_out -- every position of the image is an output position of exactly one tile; the cout blocks of a tile would
  // write the same values, so only block 0 does
  unsigned own = 0;
  if (pro2 && nblk == 0) {
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int lin = ((wave >> 1) + (NW / 2) * i) * 64 + lane;
      const int hr = lin / HP, hc = lin - hr * HP;
      if (poff1[i] != OOB && hr >= 1 && hr <= TH && hc >= 1 && hc <= TW) own |= 1u << i;
    }
  }
  // dy = scale * dA * [scale * y + shift > 0] + k1 * y + k0, in place over this wave's own items (as a_transform).  Four
  // channels at a time with the coefficients re-read from LDS: the accumulators and two fragment sets are live here.
  auto a_transform2 = [&](int cc, int st) {
    const int cabs = cc * KC + apl * EPC;
    const float* s_k = reinterpret_cast<const float*>(smem + C::K_OFF);
    const unsigned soff = (unsigned)(cc * KC * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int g = (wave >> 1) + (NW / 2) * i;
      if (g >= C::NGA) continue;
      if (poff1[i] != OOB) {
        unsigned char* p = smem + st + apl * APLB + 16 + (g * 64 + lane) * 16;
        const unsigned char* py = smem + C::Y_OFF + apl * APLB + 16 + (g * 64 + lane) * 16;
        const u32x4 pd = *reinterpret_cast<const u32x4*>(p), pv = *reinterpret_cast<const u32x4*>(py);
        u32x4 pk;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4 sf = *reinterpret_cast<const f32x4*>(s_coef + cabs + 4 * h2);
          const f32x4 bf = *reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs + 4 * h2);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(s_k + cabs + 4 * h2);
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(s_k + C::COEF_MAX + cabs + 4 * h2);
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float dv, yv;
            if constexpr (__is_same(T, bf16)) {
              const unsigned wd = pd[2 * h2 + (j >> 1)], wy = pv[2 * h2 + (j >> 1)];
              dv = __uint_as_float((j & 1) ? (wd & 0xffff0000u) : (wd << 16));
              yv = __uint_as_float((j & 1) ? (wy & 0xffff0000u) : (wy << 16));
            } else {
              const unsigned wd = pd[2 * h2 + (j >> 1)], wy = pv[2 * h2 + (j >> 1)];
              dv = (float)__builtin_bit_cast(f16, (unsigned short)((j & 1) ? (wd >> 16) : (wd & 0xffffu)));
              yv = (float)__builtin_bit_cast(f16, (unsigned short)((j & 1) ? (wy >> 16) : (wy & 0xffffu)));
            }
            const float z = fmaf(sf[j], yv, bf[j]);
            const float dz = z > 0.f ? dv : 0.f;
            o[j] = fmaf(sf[j], dz, fmaf(k1[j], yv, k0[j]));
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const T lo = from_f32<T>(o[2 * j]), hi = from_f32<T>(o[2 * j + 1]);
            pk[2 * h2 + j] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
          }
        }
        st16(p, pk);
        if ((own >> i) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pk, rsd, poff1[i], soff, 0);
      }
    }
  };

  // BN+ReLU prologue: in place over this wave's own items of chunk cc (after its vmcnt wait, before the barrier);
  // padding positions stay zero
  auto a_transform = [&](int cc, int st) {
    const int cabs = cc * KC + apl * EPC;
    if (cabs >= a.C1) return;
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_coef + cabs);
    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(s_coef + cabs + 4);
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs);
    const f32x4 sh1 = *reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs + 4);
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int g = (wave >> 1) + (NW / 2) * i;
      if (g >= C::NGA) continue;
      if (poff1[i] != OOB) {
        unsigned char* p = smem + st + apl * APLB + 16 + (g * 64 + lane) * 16;
        float f[EPC];
        Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(p), f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[j] = fmaxf(fmaf(sc0[j], f[j], sh0[j]), 0.f);
          f[4 + j] = fmaxf(fmaf(sc1[j], f[4 + j], sh1[j]), 0.f);
        }
        st16(p, Chunk<T>::pack(f));
      }
    }
  };

  // ---- per-lane fragment bases (bytes inside a stage) ------------------------------------------------------
  int amid[M_REP];
  unsigned aflag = 0;
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int ty = 2 * (wm * M_REP + m) + prow;
    amid[m] = ((ty + 1) * HP + pcol + (C::W16 ? 1 : 0)) * 16 + h * APLB;
    aflag |= (unsigned)s_flag[ty] << (2 * m);
  }
  const int azer = (C::ZB + pcol) * 16 + h * APLB;
  auto aaddr = [&](int m, int d) {  // d: tap row 0..2; first / last row of an image: the neighbour is the zero row
    if (d == 1) return amid[m];
    const bool z = (aflag >> (2 * m + (d == 0 ? 0 : 1))) & 1u;
    return z ? azer : amid[m] + (d == 0 ? -HP * 16 : HP * 16);
  };
  int bbase[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) bbase[n] = C::A_BYTES + ((wn * N_REP + n) * 32 + r) * 16 + h * BPLB;

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

  // first chunk's halo tile
  plane_static_for<0, NAI>([&](auto I) { a_dma(cc0, I, 0); });
  if (a.prologue) {  // wave-uniform
    // coefficients from the previous layer's sums (cy_bn_acc.h): behind the first chunk's DMA, whose flight time covers
    // the accumulator reads; workgroup 0 leaves them in memory for the backward pass
    float ck1 = 0.f, ck0 = 0.f;
    if constexpr (pro2) {  // (k1, k0) from the backward sums; workgroup 0 adds dgamma / dbeta (what bn_bwd_finalize_kernel did)
      unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(smem + C::STAGE + C::A_BYTES);
      bn_acc_gather<true>(a.bfold.acc, a.bfold.R, a.C1, s_sum, tid, C::NTHR);
      if (tid < a.C1) bn_bwd_fold_channel(a.bfold, s_sum, tid, blockIdx.x == 0 && blockIdx.z == 0, ck1, ck0);
    } else if (a.fold.acc) {  // (the sums of the replicas through LDS: the weight region of the second stage is idle until chunk 1)
      unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(smem + C::STAGE + C::A_BYTES);
      bn_acc_gather<true>(a.fold.acc, a.fold.R, a.C1, s_sum, tid, C::NTHR);
      if (tid < a.C1) bn_fold_channel_lds(a.fold, s_sum, tid, blockIdx.x == 0 && blockIdx.z == 0, csc, csh);
    }
    if (tid < a.C1) {
      s_coef[tid] = csc;
      s_coef[C::COEF_MAX + tid] = csh;
      if constexpr (pro2) {
        float* s_k = reinterpret_cast<float*>(smem + C::K_OFF);
        s_k[tid] = ck1;
        s_k[C::COEF_MAX + tid] = ck0;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- main loop ------------------------------------------------------------------------------------------------
  // Per chunk: nine taps of M_REP x N_REP MFMAs; the fragments of tap t+1 are requested before the MFMAs of tap t
  // (two register sets).  The pipeline runs ACROSS the chunk boundary: the fragments of a chunk's last tap are in
  // registers when the wave reaches the barrier, and behind the barrier it first requests tap 0 of the next chunk
  // and then issues the last tap's MFMAs -- the matrix pipe has work while the first fragments of the new stage
  // are on their way (stamps: 820 of 5430 cycles per chunk were boundary, barrier + exposed fragment latency).
  // Nine taps per chunk flip the parity of the register sets, so the loop body is two chunks (static set names).
  typename M::Frag fa[2][M_REP], fb[2][N_REP];
  auto load_tap = [&](int st, auto SET, auto TAP) {
    constexpr int set = decltype(SET)::value, t = decltype(TAP)::value;
    constexpr int d = t / 3, dw = t % 3 - 1;
    const unsigned char* sb = smem + st;
#pragma unroll
    for (int n = 0; n < N_REP; ++n) fb[set][n] = frag(sb + bbase[n] + t * 2 * BPLB);
#pragma unroll
    for (int m = 0; m < M_REP; ++m) fa[set][m] = frag(sb + aaddr(m, d) + (dw + 1) * 16);
  };
  auto mma_set = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < M_REP; ++m)
#pragma unroll
      for (int n = 0; n < N_REP; ++n) M::mma(fb[set][n], fa[set][m], acc[m][n]);  // rows = couts
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // the wave's part of chunk cc has landed (and is transformed); every wave's has; the other stage is free
  auto open_chunk = [&](int cc, int st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (pro2) a_transform2(cc, st);
    else if (a.prologue) a_transform(cc, st);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  int st_cur = 0, st_nxt = C::STAGE;
  open_chunk(cc0, st_cur);
  FLOW_STAMP(3, __builtin_amdgcn_s_memtime());
  load_tap(st_cur, TapC<0>{}, TapC<0>{});
  // one chunk whose tap t sits in register set (t + P) & 1; entered with tap 0 requested
  auto chunk = [&](int cc, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    const bool more = cc + 1 < cc1;
    plane_static_for<0, 8>([&](auto TAP) {
      constexpr int t = decltype(TAP)::value;
      load_tap(st_cur, TapC<(t + 1 + P) & 1>{}, TapC<t + 1>{});
      mma_set(TapC<(t + P) & 1>{});
      // the DMA of chunk cc + 1 behind the MFMAs of taps 0..: one halo item and one weight piece per tap and wave
      if (more) {
        if constexpr (t < NBI) b_dma(cc + 1, TapC<(t < NBI ? t : 0)>{}, st_nxt);
        if constexpr (t < NAI) a_dma(cc + 1, TapC<(t < NAI ? t : 0)>{}, st_nxt);
      }
    });
    if (more) {
      if constexpr (8 < NBI) b_dma(cc + 1, TapC<(8 < NBI ? 8 : 0)>{}, st_nxt);
      if constexpr (8 < NAI) a_dma(cc + 1, TapC<(8 < NAI ? 8 : 0)>{}, st_nxt);
      open_chunk(cc + 1, st_nxt);  // (the last tap's fragments are in registers: lgkmcnt(0) in front of the barrier)
      load_tap(st_nxt, TapC<(9 + P) & 1>{}, TapC<0>{});
    }
    mma_set(TapC<(8 + P) & 1>{});
    const int sw = st_cur;
    st_cur = st_nxt;
    st_nxt = sw;
  };
  {
    int cc = cc0;
    for (; cc + 1 < cc1; cc += 2) {
      chunk(cc, TapC<0>{});
      chunk(cc + 1, TapC<1>{});
    }
    if (cc < cc1) chunk(cc, TapC<0>{});
  }
  FLOW_STAMP(4, __builtin_amdgcn_s_memtime());
  __syncthreads();  // the statistics scratch aliases the operand buffers

  // ---------------- epilogue (as conv3x3_plane_kernel): accumulators -> NHWC from registers ----------------
  auto position = [&](int m, int& R, int& w) -> bool {
    R = R0 + 2 * (wm * M_REP + m) + prow;
    if constexpr (C::W16) {
      w = w0 + pcol;
      return R < a.NH && w < a.W;
    } else {
      w = w0 + pcol - 1;
      return pcol >= 1 && pcol <= TW && R < a.NH && w < a.W;
    }
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
    FLOW_STAMP(5, __builtin_amdgcn_s_memtime());
    FLOW_STAMP(6, __builtin_amdgcn_s_memrealtime());
    return;
  }
  float* sstat = reinterpret_cast<float*>(smem);
  const bool do_stats = a.stats != nullptr || a.sacc != nullptr;
  T* o1 = reinterpret_cast<T*>(a.out);
  T* o2 = reinterpret_cast<T*>(a.out2);
  // Statistics of the ROUNDED outputs.  A lane's column is fixed, so when every row of the tile is inside the
  // grid (all but the last row tile) the per-element validity select is ONE select per sum at the end, and the
  // sums run as packed f32 pairs straight off the packed 16-bit store payload (stamps: the statistics were
  // 11 of the epilogue's 17 thousand cycles -- eight waves of pure VALU work with nothing to overlap).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool rows_full = R0 + TH <= a.NH;  // wave-uniform
  bool col_ok;
  {
    int R, w;
    (void)position(0, R, w);
    col_ok = C::W16 ? (w < a.W) : (pcol >= 1 && pcol <= TW && w < a.W);
  }
  // DZ (STATS_ == 2): the output is the dA of relu(bn(y)); the sums are of dz = dA * [scale * y + shift > 0] and dz * xhat
  // (xhat = y * invstd - mean * invstd), with this lane's four channels of y per fragment and group fetched one fragment
  // ahead and the coefficient rows [scale, shift, invstd, -mean * invstd][BN] read from LDS (s_dz)
  const float* s_dz = reinterpret_cast<const float*>(smem + 16384);
  const T* dzy = reinterpret_cast<const T*>(a.dz_y);
  auto epilogue = [&](auto STATS_, auto MASKED_) {
    constexpr bool STATS = decltype(STATS_)::value != 0, MASKED = decltype(MASKED_)::value != 0;
    constexpr bool DZ = decltype(STATS_)::value == 2;
    u32x2 ycur[4], ynxt[4];
    auto yload = [&](int n, int m, u32x2 (&yv)[4]) {
      int R, w;
      // (unconditional loads from a clamped address: a load behind a per-lane branch turns every live value of the
      //  unrolled epilogue into a phi web; invalid positions are masked out of the sums below)
      const bool ok = position(m, R, w);
      const size_t gpos = ok ? (size_t)R * a.W + w : 0;
      const T* yp = dzy + gpos * a.dz_ld + (n0 - a.dz_c0) + (wn * N_REP + n) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) yv[g] = *reinterpret_cast<const u32x2*>(yp + 8 * g);
    };
    auto unpack2 = [&](unsigned wd) {
      if constexpr (__is_same(T, bf16)) {
        return f32x2{__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)};
      } else {
        return f32x2{(float)__builtin_bit_cast(f16, (unsigned short)(wd & 0xffffu)),
                     (float)__builtin_bit_cast(f16, (unsigned short)(wd >> 16))};
      }
    };
    if constexpr (DZ) yload(0, 0, ycur);
#pragma unroll
  for (int n = 0; n < N_REP; ++n) {
    // phase A: the block's accumulators -> the packed 16-bit store payload (frees 16 registers per fragment for 8:
    // everything below works from the payload, and with the DZ sums the live set would not fit otherwise)
    u32x2 packed[M_REP][4];
#pragma unroll
    for (int m = 0; m < M_REP; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = from_f32<T>(acc[m][n][4 * g + j]);
        packed[m][g] = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
        if constexpr (DZ) {  // (materialise the payload HERE: the optimiser otherwise sinks the conversions to their first
                             //  use and the 16 accumulator registers of every fragment stay live through the whole block)
          asm volatile("" : "+v"(packed[m][g][0]), "+v"(packed[m][g][1]));
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 p1[8], p2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p1[i] = p2[i] = f32x2{0.f, 0.f};
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      int R, w;
      const bool ok = position(m, R, w);
      const size_t gp = (size_t)R * a.W + w;
      if constexpr (DZ) {
        if (m + 1 < M_REP) yload(n, m + 1, ynxt);
        else if (n + 1 < N_REP) yload(n + 1, 0, ynxt);
        // (the coefficient rows read below are the same for every m: without a barrier the unrolled code reads all
        //  sixteen vectors of a cout block once, up front -- 64 registers beside the accumulators, ~900 spills)
        asm volatile("" ::: "memory");
      }
      int zofs = 0;  // an opaque zero per fragment: the coefficient reads of different fragments must not be merged
      if constexpr (DZ) asm volatile("s_mov_b32 %0, 0" : "=s"(zofs));
      if constexpr (STATS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 csf, cbf, cis, cmi;
          if constexpr (DZ) {
            asm volatile("" ::: "memory");  // (pins this group's four reads here: see above)
            const int cl = (wn * N_REP + n) * 32 + 8 * g + 4 * h + zofs;
            csf = *reinterpret_cast<const f32x4*>(s_dz + cl), cbf = *reinterpret_cast<const f32x4*>(s_dz + BN + cl);
            cis = *reinterpret_cast<const f32x4*>(s_dz + 2 * BN + cl), cmi = *reinterpret_cast<const f32x4*>(s_dz + 3 * BN + cl);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            f32x2 q = unpack2(packed[m][g][k]);
            if constexpr (MASKED) {
              if (!ok) q = f32x2{0.f, 0.f};
            }
            if constexpr (DZ) {
              const f32x2 yv = unpack2(ycur[g][k]);
              const f32x2 sf2 = f32x2{csf[2 * k], csf[2 * k + 1]}, bf2 = f32x2{cbf[2 * k], cbf[2 * k + 1]};
              const f32x2 is2 = f32x2{cis[2 * k], cis[2 * k + 1]}, mi2 = f32x2{cmi[2 * k], cmi[2 * k + 1]};
              const f32x2 z = sf2 * yv + bf2;
              const f32x2 dz = f32x2{z[0] > 0.f ? q[0] : 0.f, z[1] > 0.f ? q[1] : 0.f};
              p1[2 * g + k] += dz;
              p2[2 * g + k] += dz * (yv * is2 + mi2);
            } else {
              p1[2 * g + k] += q;
              p2[2 * g + k] += q * q;
            }
          }
          if constexpr (DZ) __builtin_amdgcn_sched_barrier(0);  // (one channel group's four coefficient vectors at a time)
        }
      }
      if constexpr (DZ) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ycur[g] = ynxt[g];
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {  // half-wave swaps pair the 8-byte channel runs into 16-byte stores
        u32x2 lo = packed[m][g], hi = packed[m][g + 1];
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
      __builtin_amdgcn_sched_barrier(0);  // (one fragment at a time)
    }
    float s1v[16], s2v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool keep = col_ok || MASKED;  // (the slow path has masked per element)
      s1v[2 * i] = keep ? p1[i][0] : 0.f;
      s1v[2 * i + 1] = keep ? p1[i][1] : 0.f;
      s2v[2 * i] = keep ? p2[i][0] : 0.f;
      s2v[2 * i + 1] = keep ? p2[i][1] : 0.f;
    }
    if constexpr (STATS) {  // reduce-scatter over the 32 lanes of each half (see conv3x3_plane_kernel)
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
  };
  // (compile-time variants: a run-time `do_stats` inside the unrolled loops turns every sum into a phi web)
  if constexpr (DZK) {
    if (dz_on) {  // (the coefficient rows of this block's couts: in LDS behind the statistics scratch)
      if (tid < BN) {
        float* w_dz = reinterpret_cast<float*>(smem + 16384);
        w_dz[tid] = dzc[0], w_dz[BN + tid] = dzc[1], w_dz[2 * BN + tid] = dzc[3], w_dz[3 * BN + tid] = -dzc[2] * dzc[3];
      }
      __syncthreads();
      if (rows_full) epilogue(TapC<2>{}, TapC<0>{});
      else epilogue(TapC<2>{}, TapC<1>{});
    } else {
      epilogue(TapC<0>{}, TapC<0>{});
    }
  } else if (!do_stats) epilogue(TapC<0>{}, TapC<0>{});
  else if (rows_full) epilogue(TapC<1>{}, TapC<0>{});
  else epilogue(TapC<1>{}, TapC<1>{});
  if (do_stats || dz_on) {
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int qq = 0; qq < WGM; ++qq) {
        t1 += sstat[(qq * 2 + 0) * BN + tid];
        t2 += sstat[(qq * 2 + 1) * BN + tid];
      }
      if (dz_on) {
        bn_acc_add(a.dz_acc, a.dz_R, a.dz_C, tile & (a.dz_R - 1), 0, n0 - a.dz_c0 + tid, t1);
        bn_acc_add(a.dz_acc, a.dz_R, a.dz_C, tile & (a.dz_R - 1), 1, n0 - a.dz_c0 + tid, t2);
      } else if (a.sacc) {
        bn_acc_add(a.sacc, a.sR, a.Cout, tile & (a.sR - 1), 0, n0 + tid, t1);
        bn_acc_add(a.sacc, a.sR, a.Cout, tile & (a.sR - 1), 1, n0 + tid, t2);
      } else {
        a.stats[((size_t)tile * 2 + 0) * a.Cout + n0 + tid] = t1;
        a.stats[((size_t)tile * 2 + 1) * a.Cout + n0 + tid] = t2;
      }
    }
  }
  FLOW_STAMP(5, __builtin_amdgcn_s_memtime());
  FLOW_STAMP(6, __builtin_amdgcn_s_memrealtime());
#ifdef CY_FLOW_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the output stores have been acknowledged)
  FLOW_STAMP(7, __builtin_amdgcn_s_memrealtime());
#endif
#endif
}

// ---- host side -----------------------------------------------------------------------------------------------
struct FlowChoice {
  bool ok;
  int th, bn, tw;  // big tile: 32 rows x 128 couts (4 x 2 waves) or 64 rows x 64 couts (8 x 1 waves); 16- or 14-wide
  bool small_ok;   // Cout % 128 == 0: also 16 rows x 128 couts (4 x 2 waves of 64 positions x 64 couts)
};

// applicability beyond the plane kernel's: 16-bit storage, no 2x2 max on load, 16-channel chunks read one source,
// 64-cout blocks, the stage-contiguous weight image exists, every tensor below 2 GiB (32-bit buffer offsets)
inline FlowChoice flow_choice(const cy_conv_desc* d) {
  FlowChoice f = {false, 0, 0, 0, false};
  const int Cin = d->C1 + d->C2;
  if (d->in_dtype == CY_F32 || d->mode1 == CY_SRC_POOL2) return f;
  if (d->C1 % 16 || d->C2 % 16 || d->Cout % 64) return f;
  if (d->prologue && d->C1 > 512) return f;
  if (d->split_c > 0 && d->split_c % 8) return f;
  if (d->W % 16 == 0) f.tw = 16;
  else if (d->W % 14 == 0) f.tw = 14;
  else return f;
  const long eb = 2, opx = (long)d->N * d->H * d->W;
  const long px1 = d->mode1 == CY_SRC_UP2 ? (long)d->N * (d->H / 2) * (d->W / 2) : opx;
  const long lim = (1L << 31) - 1;
  if (px1 * d->ld1 * eb > lim || (d->C2 && opx * d->ld2 * eb > lim)) return f;
  if (flow_image_elems(d->Cout, Cin) == 0) return f;
  if (d->Cout % 128 == 0) f.th = 32, f.bn = 128, f.small_ok = true;
  else f.th = 64, f.bn = 64;
  f.ok = true;
  return f;
}

template <typename T, int TH, int BN, int WGM, int WGN, bool W16, int MODE = 0>
int launch_conv_flow(ConvArgs a, hipStream_t st) {
  using C = FlowCfg<T, TH, BN, WGM, WGN, W16>;
  constexpr bool BWD = MODE == 1;
  if constexpr (MODE == 0) {
    if (a.prologue == 2) {
      if constexpr (C::BWD_OK) return launch_conv_flow<T, TH, BN, WGM, WGN, W16, 1>(a, st);
      else return CY_ERR_SHAPE;
    }
    if (a.dz_y != nullptr) return launch_conv_flow<T, TH, BN, WGM, WGN, W16, 2>(a, st);
  }
  auto kern = conv3x3_flow_kernel<T, TH, BN, WGM, WGN, W16, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            BWD ? C::SMEM_BWD : C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  if (a.W % C::TW != 0 || a.Cout % BN != 0) return CY_ERR_SHAPE;
  a.tiles_w = a.W / C::TW;
  a.stamps = g_conv_stamp_buf;
  static const int xcd = [] {
    const char* e = getenv("CY_PLANE_XCD");
    return e ? atoi(e) : 1;
  }();
  a.xcd_remap = xcd;
  dim3 grid(cy_cdiv(a.NH, TH) * a.tiles_w * (a.Cout / BN), 1, a.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), BWD ? C::SMEM_BWD : C::SMEM, st, a);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

// can this tiling take the backward prologue (one more halo buffer in LDS; the four-wave tilings must still fit twice)?
template <typename T> bool flow_bwd_ok(int th, int bn, int tw) {
  if (th == 16 && bn == 128) return tw == 16 ? FlowCfg<T, 16, 128, 4, 2, true>::BWD_OK : FlowCfg<T, 16, 128, 4, 2, false>::BWD_OK;
  if (th == 32 && bn == 128) return tw == 16 ? FlowCfg<T, 32, 128, 4, 2, true>::BWD_OK : FlowCfg<T, 32, 128, 4, 2, false>::BWD_OK;
  if (th == 64 && bn == 64) return tw == 16 ? FlowCfg<T, 64, 64, 8, 1, true>::BWD_OK : FlowCfg<T, 64, 64, 8, 1, false>::BWD_OK;
  if (th == 32 && bn == 64) return tw == 16 ? FlowCfg<T, 32, 64, 4, 1, true>::BWD_OK : FlowCfg<T, 32, 64, 4, 1, false>::BWD_OK;
  if (th == 16 && bn == 64) return tw == 16 ? FlowCfg<T, 16, 64, 4, 1, true>::BWD_OK : FlowCfg<T, 16, 64, 4, 1, false>::BWD_OK;
  return false;
}

template <typename T>
int dispatch_conv_flow(const ConvArgs& a, int th, int bn, int tw, hipStream_t st) {
  if (th == 16 && bn == 128) {
    // (measured and dropped: four waves of 128 positions x 64 couts on this tile -- 0.75 KB of LDS fragments per MFMA
    //  instead of 1.0, but one wave per SIMD: 15-25 % slower on every 56 x 56 / 28 x 28 layer at N = 16)
    if (tw == 16) return launch_conv_flow<T, 16, 128, 4, 2, true>(a, st);
    return launch_conv_flow<T, 16, 128, 4, 2, false>(a, st);
  }
  if (th == 32 && bn == 128) {
    if (tw == 16) return launch_conv_flow<T, 32, 128, 4, 2, true>(a, st);
    return launch_conv_flow<T, 32, 128, 4, 2, false>(a, st);
  }
  if (th == 64 && bn == 64) {
    if (tw == 16) return launch_conv_flow<T, 64, 64, 8, 1, true>(a, st);
    return launch_conv_flow<T, 64, 64, 8, 1, false>(a, st);
  }
  if (th == 32 && bn == 64) {  // four waves of 128 positions x 64 couts, two workgroups per CU
    if (a.prologue && a.C1 > 256) return CY_ERR_SHAPE;
    if (tw == 16) return launch_conv_flow<T, 32, 64, 4, 1, true>(a, st);
    return launch_conv_flow<T, 32, 64, 4, 1, false>(a, st);
  }
  if (th == 16 && bn == 64) {  // four waves of 64 positions x 64 couts, two workgroups per CU
    if (a.prologue && a.C1 > 256) return CY_ERR_SHAPE;
    if (tw == 16) return launch_conv_flow<T, 16, 64, 4, 1, true>(a, st);
    return launch_conv_flow<T, 16, 64, 4, 1, false>(a, st);
  }
  return CY_ERR_SHAPE;
}

}  // namespace
