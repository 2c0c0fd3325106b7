// Weight gradient of the 3x3 convolution (autograd of nn.Conv2d at
// contrastyou/arch/unet.py:21,24,39) on the gfx950 matrix cores.
//
//   dw[co][ci][tap] = sum_p dy[p][co] * in[p + tap][ci]
//
// GEMM view per tap: D[co][ci] += A[co][k] * B[k][ci] with k = pixel.  Both
// operands are stored pixel-major (NHWC), i.e. with k along LDS rows, so the
// bf16 path feeds the MFMA through ds_read_b64_tr_b16 (4 rows x 16 columns
// delivered column-major); the f32 path uses the 32x32x2 f32 MFMA whose
// operands are one element per lane and needs no transpose.
//
// Work split: a workgroup owns a (32*WCO) x (32*WCI) block of (co,ci) for all
// nine taps and walks a strided subset of the spatial tiles (split-K over
// pixels, `S` splits); inside the workgroup WK waves split the pixels of a tile.
// Each split writes its f32 slab ws[s][tap][co][ci]; wgrad_reduce_kernel adds
// the slabs in split order (bitwise reproducible) into the reference layout
// dw[Cout][Cin][3][3].
//
// The input tile is staged by the same halo stager as the forward kernel, so
// pool / upsample / concat / BN+ReLU-prologue addressing is identical.
#include "cy_conv_tile.h"

#include <cstdlib>

namespace {

struct WgradArgs {
  ConvArgs c;      // sources, geometry (out/out2/w/stats unused)
  const void* dy;  // [N,H,W,Cout], pitch ldy
  float* ws;       // [S][9][co_pad][ci_pad]
  int ldy;
  int TH, TW;  // spatial tile (TH divides H)
  int tiles_h, tiles_w;
  int S;
  int co_pad, ci_pad;
  // second pixel segment (twelve-wave kernel): the same layer evaluated on another batch -- tile rows
  // >= tiles_h_a belong to it.  Same geometry per image, its own tensors and BN coefficients.
  int tiles_h_a;
  const void* src1_b;
  const void* src2_b;
  const void* dy_b;
  const float* scale_b;
  const float* shift_b;
};

template <typename T> struct WFrag;

// bf16: two transposed reads give k = 8h + {0..3} and 8h + {4..7} for column (lane&31)
template <typename T16> struct WFrag16 {
  __device__ __forceinline__ static typename Mma<T16>::Frag load(const unsigned char* row_lo,
                                                                 const unsigned char* row_hi) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row_lo));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    typename Mma<T16>::Frag f;
    f.v = __builtin_bit_cast(decltype(f.v), v);
    return f;
  }
};
template <> struct WFrag<bf16> : WFrag16<bf16> {};
template <> struct WFrag<f16> : WFrag16<f16> {};

template <typename T, int WCO, int WCI, int WK>
struct WgCfg {
  static constexpr int BCO = 32 * WCO, BCI = 32 * WCI;
  static constexpr int PA = BCO * (int)sizeof(T);  // dy tile pitch (bytes)
  static constexpr int PB = BCI * (int)sizeof(T);  // halo pitch
  static constexpr int MAXPIX = 256;               // TH*TW padded to 16 <= 256
  static constexpr int MAXHALO = 360;              // (TH+2)*roundup(TW+2,4) upper bound used for sizing
  static constexpr int A_BYTES = MAXPIX * PA;
  static constexpr int B_BYTES = MAXHALO * PB;
  static constexpr int RED_BYTES = WK > 1 ? WCO * WCI * WK * 32 * 32 * 4 : 0;
  static constexpr int MAIN = A_BYTES + B_BYTES;
  static constexpr int HP = 36;                    // fixed LDS row pitch of the halo tile (pixels)
  static constexpr int TAB = 3 * 40 * 4 + 256 * 4;  // row tables + pixel->halo index table
  static constexpr int SMEM = (MAIN > RED_BYTES ? MAIN : RED_BYTES) + TAB + 16;
  static_assert(WCO * WCI * WK == 4, "4 waves");
  static_assert(PB == 64 || PB == 128, "halo pitch");
};

template <typename T, int WCO, int WCI, int WK>
__global__ void __launch_bounds__(256, 2)
    wgrad_kernel(const WgradArgs g) {
  using C = WgCfg<T, WCO, WCI, WK>;
  constexpr int PA = C::PA, PB = C::PB;
  constexpr int EPC = ElemTr<T>::EPC;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  // bf16 operands with 128-byte pixels are read by ds_read_b64_tr_b16: pair-swizzle them
  constexpr int SWA = (IS_BF16 && PA == 128) ? 2 : 0;
  constexpr int SWB = (IS_BF16 && PB == 128) ? 2 : 0;
  const ConvArgs& a = g.c;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sDy = smem;
  unsigned char* sIn = smem + C::A_BYTES;
  constexpr int MAINB = (C::MAIN > C::RED_BYTES ? C::MAIN : C::RED_BYTES);
  int* s_row1 = reinterpret_cast<int*>(smem + ((MAINB + 15) & ~15));
  int* s_row2 = s_row1 + 40;
  int* s_flag = s_row2 + 40;
  int* s_ktab = s_flag + 40;  // tile pixel k -> halo pixel index (ty+1)*HP + tx+1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave % WK;
  const int wci = (wave / WK) % WCI;
  const int wco = wave / (WK * WCI);
  const int r = lane & 31, h = lane >> 5;

  const int TH = g.TH, TW = g.TW;
  // Fixed halo row pitch: the nine tap offsets are compile-time immediates of the LDS reads and the
  // pair-swizzle term of a shifted pixel depends on dw only (pitch % 4 == 0).
  constexpr int HW2 = C::HP;
  const int npix = TH * TW;
  const int nsteps = (npix + 15) / 16;
  const int npix_pad = nsteps * 16;
  const int ci_tiles = g.ci_pad / C::BCI;
  const int co_t = blockIdx.x / ci_tiles, ci_t = blockIdx.x % ci_tiles;
  const int co0 = co_t * C::BCO, ci0 = ci_t * C::BCI;
  const int split = blockIdx.y;
  const int ntiles = g.tiles_h * g.tiles_w;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const T* dyp = reinterpret_cast<const T*>(g.dy);
  // the tile shape is the same for every tile: tabulate pixel -> halo index once
  {
    const int npix_ = g.TH * g.TW;
    const int kk = tid < npix_ ? tid : 0;
    const int ty = kk / g.TW, tx = kk - ty * g.TW;
    s_ktab[tid] = (ty + 1) * C::HP + tx + 1;
  }

  for (int tile = split; tile < ntiles; tile += g.S) {
    const int ct = tile % g.tiles_w, rt = tile / g.tiles_w;
    const int R0 = rt * TH, w0 = ct * TW;
    __syncthreads();  // previous tile fully consumed
    conv_row_tables(a, TH, R0, tid, s_row1, s_row2, s_flag, true);
    __syncthreads();  // (also publishes s_ktab on the first tile)
    conv_stage_halo<T, PB, SWB>(a, sIn, s_row1, s_row2, TH, TW, HW2, w0, ci0, tid);
    // dy tile: rows = pixels k (ty*TW+tx), BCO channels
    {
      constexpr int CPA = PA / 16;
      const int ch = tid % CPA;
      const int co = co0 + ch * EPC;
      const int nch = npix_pad * CPA;
      for (int idx0 = tid; idx0 < nch; idx0 += 1024) {  // four independent requests in flight
        u32x4 v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int k = (idx0 + b * 256) / CPA;
          v[b] = u32x4{0u, 0u, 0u, 0u};
          if (k < npix && co < a.Cout) {
            const int hp = s_ktab[k];
            const int ty = hp / HW2 - 1, tx = hp % HW2 - 1;
            const int R = R0 + ty, w = w0 + tx;
            if (R < a.NH && w < a.W) v[b] = ld16(dyp + ((size_t)R * a.W + w) * g.ldy + co);
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int idx = idx0 + b * 256;
          if (idx < nch) {
            const int k = idx / CPA;
            st16(sDy + k * PA + (halo_chunk<PA, SWA>(k, ch) << 4), v[b]);
          }
        }
      }
    }
    __syncthreads();

    for (int step = wk; step < nsteps; step += WK) {
      const int kb = step * 16;
      if constexpr (IS_BF16) {
        // this lane supplies row q of the 4x16 blocks of its 16-lane group
        const int q = (lane & 15) >> 2, p = lane & 3, gsel = (lane >> 4) & 1;
        const int k1 = kb + 8 * h + q, k2 = k1 + 4;
        // (k1 >> 1) & 1 == ((k1 + 4) >> 1) & 1: both rows share the swizzle term
        const int acol = ((wco * 32 + 16 * gsel + 4 * p) * 2) ^ (SWA == 2 ? (((k1 >> 1) & 1) << 6) : 0);
        const unsigned char* a_lo = sDy + k1 * PA + acol;
        const unsigned char* a_hi = a_lo + 4 * PA;
        const typename Mma<T>::Frag af = WFrag<T>::load(a_lo, a_hi);
        const int colb = (wci * 32 + 16 * gsel + 4 * p) * 2;
        const int p1 = s_ktab[k1], p2 = s_ktab[k2];  // (pad pixels map to pixel 0; their dy rows are zero)
        // swizzle term of pixel p + dh*HW2 + dw depends on dw only (HW2 % 4 == 0)
        int c1[3], c2[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          c1[d] = colb ^ (SWB == 2 ? ((((p1 + d - 1) >> 1) & 1) << 6) : 0);
          c2[d] = colb ^ (SWB == 2 ? ((((p2 + d - 1) >> 1) & 1) << 6) : 0);
        }
        // three bases per row (one per dw); the tap offsets below are compile-time immediates
        const unsigned char* b1[3];
        const unsigned char* b2[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          b1[d] = sIn + p1 * PB + c1[d];
          b2[d] = sIn + p2 * PB + c2[d];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          constexpr int dummy = 0;
          const int off = ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) * PB + dummy;
          const typename Mma<T>::Frag bf = WFrag<T>::load(b1[tap % 3] + off, b2[tap % 3] + off);
          Mma<T>::mma(af, bf, acc[tap]);
        }
      } else {
        // f32: element j of lane (r,h) is pixel kb + 8h + j
        Mma<float>::Frag af;
        int hidx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kb + 8 * h + j;
          const float v = *reinterpret_cast<const float*>(sDy + k * PA + (wco * 32 + r) * 4);
          if (j < 4) af.lo[j] = v; else af.hi[j - 4] = v;
          hidx[j] = s_ktab[k];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int off = (tap / 3 - 1) * HW2 + (tap % 3 - 1);
          Mma<float>::Frag bf;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v =
                *reinterpret_cast<const float*>(sIn + (hidx[j] + off) * PB + (wci * 32 + r) * 4);
            if (j < 4) bf.lo[j] = v; else bf.hi[j - 4] = v;
          }
          Mma<float>::mma(af, bf, acc[tap]);
        }
      }
    }
  }

  // ---- write the split's slab ----
  float* slab = g.ws + (size_t)split * 9 * g.co_pad * g.ci_pad;
  if constexpr (WK == 1) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        slab[((size_t)tap * g.co_pad + co0 + wco * 32 + row) * g.ci_pad + ci0 + wci * 32 + r] =
            acc[tap][reg];
      }
  } else {
    float* red = reinterpret_cast<float*>(smem);
    const int grp = wco * WCI + wci;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      __syncthreads();
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        red[((grp * WK + wk) * 32 + row) * 32 + r] = acc[tap][reg];
      }
      __syncthreads();
      // the WK waves of a group share the 1024 outputs
      for (int e = wk * 64 + lane; e < 1024; e += WK * 64) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < WK; ++q) s += red[((grp * WK + q) * 32) * 32 + e];
        const int row = e >> 5, col = e & 31;
        slab[((size_t)tap * g.co_pad + co0 + wco * 32 + row) * g.ci_pad + ci0 + wci * 32 + col] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient, twelve waves per workgroup (one workgroup per CU).
//
// wgrad_kernel above keeps all nine taps of a 32x32 (co,ci) block in one wave: 144 accumulator
// registers, which leaves no room to keep a tile's worth of global loads in flight, so its staging
// runs at ~1.8 TB/s and only overlaps with the MFMA phase through the CU's second workgroup.  Here a
// wave owns ONE kernel row (dh) of a 32x32 block -- 48 accumulator registers -- and the workgroup is
// (co blocks x ci blocks x pixel splits = 4) x 3 kernel rows = 12 waves, 3 per SIMD:
//  * the next tile's dy rows and input halo are requested into registers before this tile's MFMAs
//    (7-14 x 16 bytes per thread x 768 threads in flight) and written to the OTHER half of a
//    double-buffered LDS image after them: one barrier per tile, global latency behind the MFMAs;
//  * operand fragments are the same transposed LDS reads (ds_read_b64_tr_b16) as above.
template <int WCO, int WCI, int WK>
struct Wg12Cfg {
  static constexpr int BCO = 32 * WCO, BCI = 32 * WCI;
  static constexpr int PA = BCO * 2, PB = BCI * 2;  // dy / halo pixel pitch in bytes (bf16)
  static constexpr int NT = 768;
  static constexpr int MAXPIX = 256, HP = 36, MAXHALO = 360, MAXITEMS = 340;
  static constexpr int A_BYTES = MAXPIX * PA, B_BYTES = MAXHALO * PB, BUF = A_BYTES + B_BYTES;
  static constexpr int RED_BYTES = WK > 1 ? 12 * 1024 * 4 : 0;
  static constexpr int MAIN = 2 * BUF > RED_BYTES ? 2 * BUF : RED_BYTES;
  static constexpr int SMEM = MAIN + 256 * 4 + 16;
  static constexpr int CPA = PA / 16, CPB = PB / 16;
  static constexpr int NDY = (MAXPIX * CPA + NT - 1) / NT;     // dy items (16 B) per thread
  static constexpr int NHL = (MAXITEMS * CPB + NT - 1) / NT;   // halo items per thread
  static_assert(WCO * WCI * WK == 4, "4 x 3 waves");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

static unsigned long long* g_w12_stamp_buf = nullptr;  // cy_debug_wgrad_stamps
#ifdef CY_WGRAD_STAMPS  // development aid (-DCY_WGRAD_STAMPS): per-wave shader-clock totals of the loop phases of
                        // workgroup (0, 0), kept in scalar registers (no memory traffic inside the loop), written once:
                        // [wave][8] = {request, mfma loop, commit, barrier, tiles}
#define W12_STAMP(PH)                                                      \
  do {                                                                     \
    if (stamping) {                                                        \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();         \
      ph_sum[PH] += t__ - t_last;                                          \
      t_last = t__;                                                        \
    }                                                                      \
  } while (0)
#else
#define W12_STAMP(PH) do {} while (0)
#endif

template <int WCO, int WCI, int WK, typename T = bf16>
__global__ void __launch_bounds__(768, 1)
    wgrad12_kernel(const WgradArgs g) {
  using C = Wg12Cfg<WCO, WCI, WK>;
  constexpr int PA = C::PA, PB = C::PB, CPA = C::CPA, CPB = C::CPB, NT = C::NT;
  constexpr int EPC = 8;
  constexpr int SWA = PA == 128 ? 2 : 0, SWB = PB == 128 ? 2 : 0;  // pair swizzle of 128-byte pixels
  constexpr int HW2 = C::HP;
  const ConvArgs& a = g.c;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_ktab = reinterpret_cast<int*>(smem + C::MAIN);  // tile pixel k -> halo pixel (ty+1)*HP + tx+1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dh = wave % 3 - 1;
  const int rest = wave / 3;
  const int wk = rest % WK;
  const int wci = (rest / WK) % WCI;
  const int wco = rest / (WK * WCI);
  const int r = lane & 31, h = lane >> 5;
#ifdef CY_WGRAD_STAMPS
  const bool stamping = g.c.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long ph_sum[4] = {0ull, 0ull, 0ull, 0ull}, t_last = 0ull;
  int ntile_done = 0;
#endif

  const int TH = g.TH, TW = g.TW;
  const int npix = TH * TW;
  const int nsteps = (npix + 15) / 16;
  const int npix_pad = nsteps * 16;
  const int ci_tiles = g.ci_pad / C::BCI;
  const int co_t = blockIdx.x / ci_tiles, ci_t = blockIdx.x % ci_tiles;
  const int co0 = co_t * C::BCO, ci0 = ci_t * C::BCI;
  const int split = blockIdx.y;
  const int ntiles = g.tiles_h * g.tiles_w;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  if (tid < 256) {
    const int kk = tid < npix ? tid : 0;
    const int ty = kk / TW, tx = kk - ty * TW;
    s_ktab[tid] = (ty + 1) * HW2 + tx + 1;
  }

  // ---- staging: request (global -> registers) / commit (registers -> LDS buffer) ----------------
  // tiles never span images (TH | H): the image of a tile and its first row in it are uniform
  const int ld1v = a.ld1, ld2v = a.ld2;
  const int cha = tid % CPA, chb = tid % CPB;  // constant per thread (768 % CP* == 0)
  const int HWt = TW + 2;
  const int nhalo = (TH + 2) * HWt;
  const int cabs = ci0 + chb * EPC;
  const bool in2 = cabs >= a.C1;
  const bool cvalid = cabs < a.C1 + a.C2;
  const bool pooled = a.mode1 == CY_SRC_POOL2 && !in2;  // 2x2 max on load: staged in the commit phase
  const bool pro = a.prologue && !in2 && cvalid;
  float psc[EPC], psh[EPC];
  int coef_seg = -1;  // segment whose BN coefficients psc / psh hold
  // the tensors of the segment a tile row belongs to (wave-uniform)
  struct Seg {
    const T* dy;
    const T* s1;
    const T* s2;
    int rt;   // tile row inside the segment
    int id;
  };
  auto segment = [&](int rt) {
    Seg sg;
    const bool second = rt >= g.tiles_h_a;
    sg.id = second ? 1 : 0;
    sg.rt = second ? rt - g.tiles_h_a : rt;
    sg.dy = reinterpret_cast<const T*>(second ? g.dy_b : g.dy);
    sg.s1 = reinterpret_cast<const T*>(second ? g.src1_b : a.src1);
    sg.s2 = reinterpret_cast<const T*>(second ? g.src2_b : a.src2);
    return sg;
  };
  auto src_row = [&](int n, int hh) -> int {  // pixel index of (row hh of image n, column 0), or -1
    if (hh < 0 || hh >= a.H) return -1;
    if (in2 || a.mode1 == CY_SRC_DIRECT) return (n * a.H + hh) * a.W;
    if (a.mode1 == CY_SRC_POOL2) return (n * 2 * a.H + 2 * hh) * (2 * a.W);
    return (n * (a.H >> 1) + (hh >> 1)) * (a.W >> 1);
  };
  u32x4 dreg[C::NDY], hreg[C::NHL];
  unsigned hok = 0;
  auto request = [&](int tile) {
    const int ct = tile % g.tiles_w;
    const Seg sg = segment(tile / g.tiles_w);
    const T* dyp = sg.dy;
    const T* s1 = sg.s1;
    const T* s2 = sg.s2;
    if (pro && sg.id != coef_seg) {  // (the tiles of a workgroup switch segment at most once)
      const float* sc = sg.id ? g.scale_b : a.scale;
      const float* sh = sg.id ? g.shift_b : a.shift;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        psc[j] = sc[cabs + j];
        psh[j] = sh[cabs + j];
      }
      coef_seg = sg.id;
    }
    const int R0 = sg.rt * TH, w0 = ct * TW;
    const int n = R0 / a.H, hh0 = R0 - n * a.H;
    const int co = co0 + cha * EPC;
#pragma unroll
    for (int i = 0; i < C::NDY; ++i) {
      const int k = (tid + i * NT) / CPA;
      dreg[i] = u32x4{0u, 0u, 0u, 0u};
      if (k < npix && co < a.Cout) {
        const int ty = k / TW, tx = k - ty * TW;
        const int w = w0 + tx;
        if (w < a.W) dreg[i] = ld16(dyp + ((size_t)(R0 + ty) * a.W + w) * g.ldy + co);
      }
    }
    hok = 0;
    if (pooled) return;
    const T* base = in2 ? s2 + (cabs - a.C1) : s1 + cabs;
    // (a per-lane choice between two kernel-argument FIELDS compiles to a per-lane pointer and a vector load of the
    //  argument followed by s_waitcnt vmcnt(0) -- a full memory round trip in the middle of every request, waiting
    //  for the dy loads just issued.  Scalars first.)
    const int ld = in2 ? ld2v : ld1v;
    const int wsh = (!in2 && a.mode1 == CY_SRC_UP2) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < C::NHL; ++i) {
      const int lin = (tid + i * NT) / CPB;
      const int hr = lin / HWt, hc = lin - hr * HWt;
      const int w = w0 - 1 + hc;
      hreg[i] = u32x4{0u, 0u, 0u, 0u};
      if (lin < nhalo && cvalid && w >= 0 && w < a.W) {
        const int rp = src_row(n, hh0 - 1 + hr);
        if (rp >= 0) {
          hreg[i] = ld16(base + (size_t)(rp + (w >> wsh)) * ld);
          hok |= 1u << i;
        }
      }
    }
  };
  auto commit = [&](int tile, unsigned char* sDy, unsigned char* sIn) {
#pragma unroll
    for (int i = 0; i < C::NDY; ++i) {
      const int k = (tid + i * NT) / CPA;
      if (k < npix_pad) st16(sDy + k * PA + (halo_chunk<PA, SWA>(k, cha) << 4), dreg[i]);
    }
    if (pooled) {
      const int ct = tile % g.tiles_w;
      const Seg sg = segment(tile / g.tiles_w);
      const T* s1 = sg.s1;
      const int R0 = sg.rt * TH, w0 = ct * TW;
      const int n = R0 / a.H, hh0 = R0 - n * a.H;
      for (int lin = tid / CPB; lin < nhalo; lin += NT / CPB) {
        const int hr = lin / HWt, hc = lin - hr * HWt;
        const int pix = hr * HW2 + hc;
        const int w = w0 - 1 + hc;
        u32x4 v = {0u, 0u, 0u, 0u};
        const int rp = src_row(n, hh0 - 1 + hr);
        if (cvalid && w >= 0 && w < a.W && rp >= 0) {
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
        st16(sIn + pix * PB + (halo_chunk<PB, SWB>(pix, chb) << 4), v);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < C::NHL; ++i) {
      const int lin = (tid + i * NT) / CPB;
      if (lin < nhalo) {
        const int hr = lin / HWt, hc = lin - hr * HWt;
        const int pix = hr * HW2 + hc;
        u32x4 v = hreg[i];
        if (pro && ((hok >> i) & 1u)) {
          float f[EPC];
          Chunk<T>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(psc[j], f[j], psh[j]), 0.f);
          v = Chunk<T>::pack(f);
        }
        st16(sIn + pix * PB + (halo_chunk<PB, SWB>(pix, chb) << 4), v);
      }
    }
  };

  if (split < ntiles) {
    request(split);
    commit(split, smem, smem + C::A_BYTES);
  }
  __syncthreads();  // first tile staged, s_ktab published

  // per-lane constants of the transposed fragment reads (see wgrad_kernel)
  const int q = (lane & 15) >> 2, p4 = lane & 3, gsel = (lane >> 4) & 1;
  const int acol0 = (wco * 32 + 16 * gsel + 4 * p4) * 2;
  const int bcol0 = (wci * 32 + 16 * gsel + 4 * p4) * 2;
  const int dhoff = dh * HW2 * PB;

  int cur = 0;
  for (int tile = split; tile < ntiles; tile += g.S, cur ^= 1) {
    const int next = tile + g.S;
#ifdef CY_WGRAD_STAMPS
    if (stamping && t_last == 0ull) t_last = __builtin_amdgcn_s_memtime();
#endif
    // (staging instructions go in front of other waves' MFMA-loop instructions on the SIMD: Conv1b 94 -> 89 us,
    //  Up_conv2a 148 -> 135 us at N = 32, the rest unchanged)
    __builtin_amdgcn_s_setprio(2);
    if (next < ntiles) request(next);
    __builtin_amdgcn_s_setprio(0);
    W12_STAMP(0);
    const unsigned char* sDy = smem + cur * C::BUF;
    const unsigned char* sIn = sDy + C::A_BYTES;
    for (int step = wk; step < nsteps; step += WK) {
      const int k1 = step * 16 + 8 * h + q, k2 = k1 + 4;
      // (k1 >> 1) & 1 == ((k1 + 4) >> 1) & 1: both rows share the swizzle term
      const int acol = acol0 ^ (SWA == 2 ? (((k1 >> 1) & 1) << 6) : 0);
      const unsigned char* a_lo = sDy + k1 * PA + acol;
      const typename Mma<T>::Frag af = WFrag<T>::load(a_lo, a_lo + 4 * PA);
      const int p1 = s_ktab[k1], p2 = s_ktab[k2];  // (pad pixels map to pixel 0; their dy rows are zero)
      typename Mma<T>::Frag bfr[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // the swizzle term of pixel p + dh*HW2 + dw depends on dw only (HW2 % 4 == 0)
        const int c1 = bcol0 ^ (SWB == 2 ? ((((p1 + d - 1) >> 1) & 1) << 6) : 0);
        const int c2 = bcol0 ^ (SWB == 2 ? ((((p2 + d - 1) >> 1) & 1) << 6) : 0);
        bfr[d] = WFrag<T>::load(sIn + (p1 + d - 1) * PB + dhoff + c1, sIn + (p2 + d - 1) * PB + dhoff + c2);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int d = 0; d < 3; ++d) Mma<T>::mma(af, bfr[d], acc[d]);
      __builtin_amdgcn_s_setprio(0);
    }
    W12_STAMP(1);
    __builtin_amdgcn_s_setprio(2);
    if (next < ntiles) commit(next, smem + (cur ^ 1) * C::BUF, smem + (cur ^ 1) * C::BUF + C::A_BYTES);
    __builtin_amdgcn_s_setprio(0);
    W12_STAMP(2);
    __syncthreads();  // this tile consumed by every wave, the next one staged
    W12_STAMP(3);
#ifdef CY_WGRAD_STAMPS
    ++ntile_done;
#endif
  }

#ifdef CY_WGRAD_STAMPS
  if (stamping && lane == 0) {
    for (int q = 0; q < 4; ++q) g.c.stamps[wave * 8 + q] = ph_sum[q];
    g.c.stamps[wave * 8 + 4] = (unsigned long long)ntile_done;
  }
#endif
  // ---- write the split's slab: taps (dh, dw = -1..1) of this wave ----
  float* slab = g.ws + (size_t)split * 9 * g.co_pad * g.ci_pad;
  if constexpr (WK == 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tap = (dh + 1) * 3 + d;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        slab[((size_t)tap * g.co_pad + co0 + wco * 32 + row) * g.ci_pad + ci0 + wci * 32 + r] = acc[d][reg];
      }
    }
  } else {
    float* red = reinterpret_cast<float*>(smem);  // [kernel row][block][pixel split][32][32]
    const int grp = (dh + 1) * (WCO * WCI) + wco * WCI + wci;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tap = (dh + 1) * 3 + d;
      __syncthreads();
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        red[((grp * WK + wk) * 32 + row) * 32 + r] = acc[d][reg];
      }
      __syncthreads();
      for (int e = wk * 64 + lane; e < 1024; e += WK * 64) {  // the WK waves of a group share 1024 outputs
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WK; ++w) sum += red[((grp * WK + w) * 32) * 32 + e];
        const int row = e >> 5, col = e & 31;
        slab[((size_t)tap * g.co_pad + co0 + wco * 32 + row) * g.ci_pad + ci0 + wci * 32 + col] = sum;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int S, int SG, int Cout,
                        int Cin, int co_pad, int ci_pad, int accumulate) {
  // thread = four consecutive ci of one co (16-byte slab reads, 36 contiguous output floats) x one
  // of SG slab groups.  Group g sums slabs g, g+SG, ... in increasing order, then the SG group sums
  // are added in group order: the summation tree is a function of (S, SG) only => bitwise reproducible.
  __shared__ float sred[256 * 36];
  const int opb = 256 / SG;
  const int ol = threadIdx.x % opb, sg = threadIdx.x / opb;
  const int Q = Cin / 4;
  const long total = (long)Cout * Q;
  const long i = (long)blockIdx.x * opb + ol;
  const size_t slab = (size_t)9 * co_pad * ci_pad;
  const size_t tapstride = (size_t)co_pad * ci_pad;
  f32x4 s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int co = i < total ? (int)(i / Q) : 0;
  const int ci = i < total ? (int)(i % Q) * 4 : 0;
  if (i < total) {
    const size_t off = (size_t)co * ci_pad + ci;
    for (int q = sg; q < S; q += SG) {
      const float* p = ws + q * slab + off;
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] += *reinterpret_cast<const f32x4*>(p + t * tapstride);
    }
  }
  if (SG > 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) sred[(sg * opb + ol) * 36 + t * 4 + j] = s[t][j];
    __syncthreads();
    if (sg == 0) {
      for (int g = 1; g < SG; ++g)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) s[t][j] += sred[(g * opb + ol) * 36 + t * 4 + j];
    }
  }
  if (sg == 0 && i < total) {
    float* o = dw + ((size_t)co * Cin + ci) * 9;  // [co][ci][tap]: 4 ci x 9 taps contiguous
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (accumulate)
          o[j * 9 + t] += s[t][j];
        else
          o[j * 9 + t] = s[t][j];
      }
  }
}

struct WgPlan {
  bool twelve;  // 16-bit storage: the twelve-wave kernels (f32: wgrad_kernel)
  bool spec;    // ... in its wave-specialised form (64 x 64 blocks; CY_WGRAD_SPEC=0 keeps wgrad12_kernel)
  bool dma;     // ... with the loader waves on LDS-DMA (CY_WGRAD_DMA=0: register staging)
  bool blk_order;  // ... and 4 x 4 pixel patches as k-steps where the tile allows (CY_WGRAD_BLK=0: row-major k)
  int wco, wci, wk;
  int TH, TW, tiles_h, tiles_w, S, co_pad, ci_pad;
};

WgPlan plan_wgrad(const cy_conv_desc* d) {
  WgPlan p;
  p.twelve = d->in_dtype != CY_F32;  // (the four-wave wgrad_kernel serves f32 only; the paired launch has no four-wave form)
  const int Cin = d->C1 + d->C2;
  if (d->in_dtype == CY_F32) {
    p.wco = 1, p.wci = 1, p.wk = 4;
  } else if (d->Cout > 32 && Cin > 32) {
    p.wco = 2, p.wci = 2, p.wk = 1;
  } else if (d->Cout > 32) {
    p.wco = 2, p.wci = 1, p.wk = 2;
  } else if (Cin > 32) {
    p.wco = 1, p.wci = 2, p.wk = 2;
  } else {
    p.wco = 1, p.wci = 1, p.wk = 4;
  }
  p.co_pad = cy_roundup(d->Cout, 32 * p.wco);
  p.ci_pad = cy_roundup(Cin, 32 * p.wci);
  // spatial tile: TW <= 32 columns (halo row pitch is fixed at 36 pixels), TH <= 8 rows with
  // TH | H (tiles never span images) and TH*TW <= 256
  p.TW = d->W <= 32 ? d->W : (d->W % 32 == 0 ? 32 : (d->W % 28 == 0 ? 28 : 16));
  int best = 1;
  for (int th = 1; th <= 8 && th <= d->H && th * p.TW <= 256; ++th)
    if (d->H % th == 0) best = th;
  p.TH = best;
  p.tiles_h = (d->N * d->H) / p.TH;
  p.tiles_w = cy_cdiv(d->W, p.TW);
  static const bool spec_enabled = [] {
    const char* e = getenv("CY_WGRAD_SPEC");
    return !(e && e[0] == '0');
  }();
  static const bool dma_enabled = [] {
    const char* e = getenv("CY_WGRAD_DMA");
    return !(e && e[0] == '0');
  }();
  static const bool blk_enabled = [] {
    const char* e = getenv("CY_WGRAD_BLK");
    return !(e && e[0] == '0');
  }();
  {
    // the wave-specialised kernel (cy_wgrad_spec.h).  With register-staging loaders it pays on the 64 x 64 blocks
    // only (measured per layer at N = 32: the pooled-on-load layers -- four synchronous loads per halo item -- and the
    // 14 x 14 layers -- 98-pixel tiles -- are faster in the unspecialised kernel; on the 32-channel blocks of the
    // 224^2 level four loader waves could not issue a tile's gathers fast enough: Conv1b 89 -> 109 us).  With the
    // loaders on LDS-DMA and 4 x 4 patches as k-steps the 32-channel blocks take it too, and so do the 14 x 14
    // layers (Conv5a / Conv5b, both passes paired: 74 -> 58 and 131 -> 116 us).
    // DMA: 32-bit buffer offsets (every tensor below 2 GiB; the caller's total batch is in d->N), a ci block reads
    // one source.
    const long lim = (1L << 31) - 1, eb = 2;
    const long opx = (long)d->N * d->H * d->W;
    const long px1 = d->mode1 == CY_SRC_UP2 ? opx / 4 : opx;
    const bool small = px1 * d->ld1 * eb <= lim && (!d->C2 || opx * d->ld2 * eb <= lim) && opx * d->ldo * eb <= lim;
    constexpr int min_w = 14;
    const bool base = p.twelve && spec_enabled && d->mode1 != CY_SRC_POOL2 && d->W >= min_w;
    const bool dma_ok = dma_enabled && small && (d->C2 == 0 || d->C1 % (32 * p.wci) == 0);
    const bool patches = dma_ok && blk_enabled && p.TH % 4 == 0 && p.TW % 4 == 0;
    p.spec = base && ((p.wco == 2 && p.wci == 2) || (p.wco == 1 && patches));
    p.dma = p.spec && dma_ok;
    p.blk_order = p.dma && patches;
  }
  const int out_tiles = (p.co_pad / (32 * p.wco)) * (p.ci_pad / (32 * p.wci));
  const int ntiles = p.tiles_h * p.tiles_w;
  // enough workgroups to fill 256 CUs twice (wgrad_kernel: two per CU) or once (twelve waves: one
  // per CU), slabs bounded to ~48 MB
  int S = (p.twelve ? 256 : 512) / out_tiles;
  const long slab_bytes = 9L * p.co_pad * p.ci_pad * 4;
  const long cap = (48L << 20) / slab_bytes;
  if (S > cap) S = (int)cap;
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  p.S = S;
  return p;
}

template <typename T, int WCO, int WCI, int WK>
int launch_wgrad(const WgradArgs& g, const WgPlan& p, hipStream_t st) {
  using C = WgCfg<T, WCO, WCI, WK>;
  auto kern = wgrad_kernel<T, WCO, WCI, WK>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  dim3 grid((p.co_pad / C::BCO) * (p.ci_pad / C::BCI), p.S);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM, st, g);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

template <int WCO, int WCI, int WK, typename T = bf16>
int launch_wgrad12(const WgradArgs& g, const WgPlan& p, hipStream_t st) {
  using C = Wg12Cfg<WCO, WCI, WK>;
  auto kern = wgrad12_kernel<WCO, WCI, WK, T>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  dim3 grid((p.co_pad / C::BCO) * (p.ci_pad / C::BCI), p.S);
  WgradArgs ga = g;
  ga.c.stamps = g_w12_stamp_buf;
  hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::SMEM, st, ga);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

#include "cy_wgrad_spec.h"  // (inside the anonymous namespace)

// ---------------------------------------------------------------------------
// first layer weight gradient (Cin 1..4, f32 NCHW image): plain VALU reduction
template <typename T>
__global__ void __launch_bounds__(256)
    first_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                       float* __restrict__ ws, int N, int Cin, int H, int W, int Cout) {
  // thread = (pixel lane, group of 8 couts); accumulates 9 taps x 8 couts for one ci at a time
  __shared__ float sred[4 * 9 * 64];  // [wave][tap][cout], Cout <= 64
  extern __shared__ float sx[];        // [rows of this block's pixel range + 2][W + 2] of the current ci (see
                                       // conv3x3_first_kernel: nine 4-byte global loads per pixel otherwise)
  const int CG = Cout / 8;
  const int rows = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, prow = tid / CG;
  const long npix = (long)N * H * W;
  const long per = (npix + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  for (int ci = 0; ci < Cin; ++ci) {
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const int W2 = W + 2;
    const int g0 = p0 < npix ? (int)(p0 / W) : 0;
    const int g1 = p1 > p0 ? (int)((p1 - 1) / W) : g0;
    const int nrows = g1 - g0 + 3;
    __syncthreads();  // (the previous ci's image is no longer read)
    for (int r = tid / 64; r < nrows; r += 4) {  // a wave per row
      const int g = g0 - 1 + r;
      const bool rok = g >= 0 && g < N * H;
      const int n_ = rok ? g / H : 0;
      const float* xr = x + (((size_t)n_ * Cin + ci) * H + (g - n_ * H)) * W;
      float* dst = sx + (size_t)r * W2;
      for (int c = tid & 63; c < W2; c += 64) dst[c] = (rok && c >= 1 && c <= W) ? xr[c - 1] : 0.f;
    }
    __syncthreads();
    long p = p0 + prow;
    int wq, hq, n;
    {
      const unsigned pu = (unsigned)(p < npix ? p : 0);  // (npix < 2^31 checked by the host)
      const unsigned row = pu / (unsigned)W;
      wq = (int)(pu - row * (unsigned)W);
      n = (int)(row / (unsigned)H);
      hq = (int)(row - (unsigned)n * (unsigned)H);
    }
    const int dq = rows / W, dr = rows - dq * W;
    for (; p < p1; p += rows) {
      float d[8];
      if constexpr (sizeof(T) == 2) {
        Chunk<T>::unpack(ld16(dy + p * Cout + cg * 8), d);
      } else {
        Chunk<float>::unpack(ld16(dy + p * Cout + cg * 8), d);
        Chunk<float>::unpack(ld16(dy + p * Cout + cg * 8 + 4), d + 4);
      }
      const float* xc = sx + (size_t)(n * H + hq - g0 + 1) * W2 + wq + 1;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dh = t / 3 - 1, dw = t % 3 - 1;
        const int hh = hq + dh;
        const float xv = (hh >= 0 && hh < H) ? xc[dh * W2 + dw] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = fmaf(xv, d[j], acc[t][j]);
      }
      wq += dr;
      hq += dq;
      if (wq >= W) wq -= W, ++hq;
      while (hq >= H) hq -= H, ++n;
    }
    // block reduction over the pixel lanes: xor-shuffles across the lanes of a wave that share a
    // cout group (CG is a power of two), then the four waves through LDS
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[t][j];
        for (int o = CG; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (lane < CG) sred[((wave * 9 + t) * CG + lane) * 8 + j] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e < 9 * Cout; e += 256) {
      const int t = e / Cout, co = e - t * Cout;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) s += sred[((q * 9 + t) * CG + (co >> 3)) * 8 + (co & 7)];
      // ws[block][co][ci][tap]
      ws[(((size_t)blockIdx.x * Cout + co) * Cin + ci) * 9 + t] = s;
    }
  }
}

// The same on the matrix core (Cin = 1, Cout = 32, 16-bit storage, H % 8 == 0, W % 16 == 0): the VALU kernel above spends
// 72 FMAs, nine LDS reads and a dependent 16-byte load per (pixel, 8 couts) and runs at 2.1 TB/s of dy (93 us of a C2 step,
// 1.36 ms of a C5 step).  dW[co][tap] = sum_px dy[px][co] x[px + tap] is one 32x32x16 MFMA per 16 consecutive pixels of an
// image row: rows = couts (dy, transposed through a 1 KB per-wave LDS tile with ds_read_b64_tr_b16 like the weight
// gradient of the other layers), columns = taps (9 of 32 used), k = the 16 pixels; a tap's fragment is ONE aligned 16-byte
// LDS read from the image rows, which are staged in the storage type three times (shifted by -1 / 0 / +1 columns).
// A workgroup walks sub-blocks of eight image rows (ten staged); one partial per workgroup, the four waves summed in wave
// order: bitwise reproducible.  The image is rounded to the storage type, as the forward kernel on the matrix core does.
template <typename T>
__global__ void __launch_bounds__(256)
    first_wgrad_mfma_kernel(const float* __restrict__ x, const T* __restrict__ dy, float* __restrict__ ws, int N, int H,
                            int W, int spb) {
  using M = Mma<T>;
  constexpr int Cout = 32, SUBR = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* sdy = smem + wave * 1024;          // this wave's [16 px][64 B] tile
  T* simg = reinterpret_cast<T*>(smem + 4096);      // [3 shifts][SUBR + 2 rows][W]
  const int h = lane >> 5, tap = lane & 31;
  const int dh = tap < 9 ? tap / 3 - 1 : 0, dw = tap < 9 ? tap % 3 - 1 : 0;
  // per-lane constants of the transposed dy fragment (WFrag, wgrad12 kernels)
  const int q = (lane & 15) >> 2, p4 = lane & 3, gsel = (lane >> 4) & 1;
  const unsigned char* arow = sdy + (8 * h + q) * 64 + (16 * gsel + 4 * p4) * 2;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int NH = N * H, nsub = NH / SUBR;
  const int gpr = W / 16;  // pixel groups per image row
  const int sb0 = (int)blockIdx.x * spb, sb1 = sb0 + spb < nsub ? sb0 + spb : nsub;
  for (int sb = sb0; sb < sb1; ++sb) {
    const int r0 = sb * SUBR;  // first global row (n * H + h) of the sub-block; its rows belong to one image
    __syncthreads();           // (the previous sub-block's image is no longer read)
    for (int r = wave; r < SUBR + 2; r += 4) {
      const int g = r0 - 1 + r;
      const bool rok = g >= 0 && g < NH;
      const float* xr = x + (size_t)(rok ? g : 0) * W;
      for (int c = lane; c < W; c += 64) {
        const float v0 = (rok && c >= 1) ? xr[c - 1] : 0.f, v1 = rok ? xr[c] : 0.f, v2 = (rok && c + 1 < W) ? xr[c + 1] : 0.f;
        simg[(0 * (SUBR + 2) + r) * W + c] = from_f32<T>(v0);
        simg[(1 * (SUBR + 2) + r) * W + c] = from_f32<T>(v1);
        simg[(2 * (SUBR + 2) + r) * W + c] = from_f32<T>(v2);
      }
    }
    __syncthreads();
    const int ngr = SUBR * gpr;
    const T* dyb = dy + (size_t)r0 * W * Cout;
    u32x4 d = {0u, 0u, 0u, 0u};
    if (wave < ngr) d = ld16(dyb + (size_t)wave * 16 * Cout + lane * 8);
    for (int gi = wave; gi < ngr; gi += 4) {
      const int rr = gi / gpr, w0 = (gi - rr * gpr) * 16;
      const int hq = (r0 + rr) % H;
      st16(sdy + lane * 16, d);
      // (the next group's rows in flight; three groups ahead measured the same: a launch at C2's sizes is its fixed costs)
      if (gi + 4 < ngr) d = ld16(dyb + (size_t)(gi + 4) * 16 * Cout + lane * 8);
      __builtin_amdgcn_wave_barrier();
      const typename M::Frag af = WFrag<T>::load(arow, arow + 4 * 64);
      typename M::Frag bf;
      {
        const bool on = tap < 9 && hq + dh >= 0 && hq + dh < H;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (on) v = *reinterpret_cast<const u32x4*>(simg + ((dw + 1) * (SUBR + 2) + rr + 1 + dh) * W + w0 + 8 * h);
        bf.v = __builtin_bit_cast(decltype(bf.v), v);
      }
      M::mma(af, bf, acc);
      __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by the next group)
    }
  }
  // the workgroup's partial: waves 0..3 summed in wave order, then ws[block][co][tap]
  __syncthreads();
  float* sred = reinterpret_cast<float*>(smem);  // [4 waves][16 regs][64 lanes] = 16 KB (aliases the tiles and the image)
#pragma unroll
  for (int i = 0; i < 16; ++i) sred[(wave * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
  for (int e = tid; e < Cout * 9; e += 256) {
    const int co = e / 9, t = e - co * 9;
    const int l = t + 32 * ((co >> 2) & 1), v = 4 * (co >> 3) + (co & 3);  // D[co][t]: lane t + 32 (co/4 % 2), register 4 (co/8) + co % 4
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) sum += sred[(wv * 16 + v) * 64 + l];
    ws[((size_t)blockIdx.x * Cout + co) * 9 + t] = sum;
  }
}

// dw[i] (+)= sum over the block partials: 4 outputs per workgroup, 64 slices each, fixed order
__global__ void __launch_bounds__(256)
    first_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk,
                              int total, int accumulate) {
  __shared__ float sr[64][4];
  const int el = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int i = blockIdx.x * 4 + el;
  float s = 0.f;
  if (i < total)
    for (int q = sl; q < nblk; q += 64) s += ws[(size_t)q * total + i];
  sr[sl][el] = s;
  __syncthreads();
  if (sl == 0 && i < total) {
    float t = 0.f;
    for (int q = 0; q < 64; ++q) t += sr[q][el];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// workgroups of first_wgrad_mfma_kernel (sub-blocks of eight rows, contiguous ranges of them), or 0 where it does not apply
int first_wgrad_mfma_blocks(int N, int Cin, int H, int W, int Cout, int dtype) {
  static const bool enabled = [] {
    const char* e = getenv("CY_FIRST_WGRAD_MFMA");
    return !(e && e[0] == '0');
  }();
  if (!enabled || Cin != 1 || Cout != 32 || (dtype != CY_BF16 && dtype != CY_F16) || H % 8 || W % 16 || W > 2048) return 0;
  const long nsub = (long)N * H / 8;
  return (int)(nsub < 1024 ? nsub : 1024);
}

int first_wgrad_blocks(long npix, int W) {
  long b = (npix + 1023) / 1024;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  // never more image rows per block than its LDS image holds (96 KB, one input channel at a time)
  const long rows_max = (96 * 1024 / 4) / (W + 2) - 4;
  if (rows_max < 1) return -1;
  const long need = (npix + rows_max * W - 1) / (rows_max * W);
  if (b < need) b = need;
  return (int)b;
}

}  // namespace

extern "C" {

int cy_debug_wgrad_stamps(unsigned long long* dev_buf) {  // development aid, see W12_STAMP
  g_w12_stamp_buf = dev_buf;
  return CY_OK;
}


int cy_conv3x3_wgrad_plan(const cy_conv_desc* d, int n_b, cy_wgrad_plan* plan) {
  if (!d || !plan || d->Cout <= 0 || d->C1 <= 0 || d->H <= 0 || d->W <= 0 || d->N <= 0 || n_b < 0) return CY_ERR_ARG;
  cy_conv_desc dt = *d;
  dt.N += n_b;
  const WgPlan p = plan_wgrad(&dt);
  plan->twelve = p.spec ? 2 : (p.twelve ? 1 : 0);
  plan->wco = p.wco, plan->wci = p.wci, plan->wk = p.wk, plan->th = p.TH, plan->tw = p.TW, plan->splits = p.S;
  plan->workgroups = (p.co_pad / (32 * p.wco)) * (p.ci_pad / (32 * p.wci)) * p.S;
  return CY_OK;
}

size_t cy_conv3x3_wgrad_ws_bytes(const cy_conv_desc* d) {
  if (!d || d->Cout <= 0 || d->C1 <= 0 || d->H <= 0 || d->W <= 0 || d->N <= 0) return 0;
  const WgPlan p = plan_wgrad(d);
  return (size_t)p.S * 9 * p.co_pad * p.ci_pad * sizeof(float);
}

// one launch over the pixels of `d` (segment a) and, when n_b > 0, of the same layer on a second batch
// of n_b images (segment b: its own tensors and BN coefficients); plan from the total batch
static int wgrad_impl(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                      const float* shift, const void* dy, int n_b, const void* src1_b, const void* src2_b,
                      const float* scale_b, const float* shift_b, const void* dy_b, float* dw,
                      int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!d || !src1 || !dy || !dw || !ws) return CY_ERR_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0)
    return CY_ERR_SHAPE;
  if (d->in_dtype != d->out_dtype) return CY_ERR_DTYPE;
  if (d->in_dtype != CY_F32 && d->in_dtype != CY_BF16 && d->in_dtype != CY_F16) return CY_ERR_DTYPE;
  const int epc = d->in_dtype == CY_F32 ? 4 : 8;
  if (d->C1 % epc || d->C2 % epc || d->Cout % epc || d->ld1 % epc || d->ldo % epc)
    return CY_ERR_SHAPE;
  if (d->C2 && (!src2 || d->ld2 % epc)) return CY_ERR_ARG;
  if (d->prologue && (d->C2 || !scale || !shift)) return CY_ERR_ARG;
  if (d->mode1 == CY_SRC_UP2 && ((d->H & 1) || (d->W & 1))) return CY_ERR_SHAPE;
  cy_conv_desc dt = *d;  // the plan sees the total batch
  dt.N = d->N + (n_b > 0 ? n_b : 0);
  const WgPlan p = plan_wgrad(&dt);
  if (n_b > 0) {
    if (!p.twelve) return CY_ERR_DTYPE;  // two segments: twelve-wave (bf16) kernel only
    if (!src1_b || !dy_b || (d->C2 && !src2_b) || (d->prologue && (!scale_b || !shift_b))) return CY_ERR_ARG;
  }
  if (ws_bytes < (size_t)p.S * 9 * p.co_pad * p.ci_pad * sizeof(float)) return CY_ERR_WORKSPACE;
  WgradArgs g = {};
  ConvArgs& a = g.c;
  a.src1 = src1, a.src2 = src2, a.scale = scale, a.shift = shift, a.w = nullptr;
  a.out = nullptr, a.out2 = nullptr, a.stats = nullptr;
  a.N = dt.N, a.H = d->H, a.W = d->W, a.NH = dt.N * d->H;
  a.C1 = d->C1, a.C2 = d->C2, a.Cout = d->Cout;
  a.mode1 = d->mode1, a.prologue = d->prologue;
  a.ld1 = d->ld1, a.ld2 = d->ld2, a.ldo = 0, a.ldo2 = 0, a.split_c = 0;
  a.tiles_w = p.tiles_w, a.w_co_pad = 0, a.w_ci_pad = 0, a.full_tiles = 0;
  g.dy = dy, g.ws = (float*)ws, g.ldy = d->ldo;
  g.TH = p.TH, g.TW = p.TW, g.tiles_h = p.tiles_h, g.tiles_w = p.tiles_w, g.S = p.S;
  g.co_pad = p.co_pad, g.ci_pad = p.ci_pad;
  g.tiles_h_a = (d->N * d->H) / p.TH;  // (TH | H: segments begin on tile rows)
  g.src1_b = src1_b, g.src2_b = src2_b, g.dy_b = dy_b, g.scale_b = scale_b, g.shift_b = shift_b;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (d->in_dtype == CY_F32) {
    rc = launch_wgrad<float, 1, 1, 4>(g, p, st);
  } else if (p.spec) {
    rc = d->in_dtype == CY_F16 ? dispatch_wgrad12s<f16>(g, p, st) : dispatch_wgrad12s<bf16>(g, p, st);
  } else if (p.twelve && d->in_dtype == CY_F16) {
    if (p.wco == 2 && p.wci == 2) rc = launch_wgrad12<2, 2, 1, f16>(g, p, st);
    else if (p.wco == 2) rc = launch_wgrad12<2, 1, 2, f16>(g, p, st);
    else if (p.wci == 2) rc = launch_wgrad12<1, 2, 2, f16>(g, p, st);
    else rc = launch_wgrad12<1, 1, 4, f16>(g, p, st);
  } else if (d->in_dtype == CY_F16) {
    return CY_ERR_DTYPE;  // (f16 runs the twelve-wave kernel only)
  } else if (p.twelve) {
    if (p.wco == 2 && p.wci == 2) rc = launch_wgrad12<2, 2, 1>(g, p, st);
    else if (p.wco == 2) rc = launch_wgrad12<2, 1, 2>(g, p, st);
    else if (p.wci == 2) rc = launch_wgrad12<1, 2, 2>(g, p, st);
    else rc = launch_wgrad12<1, 1, 4>(g, p, st);
  } else if (p.wco == 2 && p.wci == 2) {
    rc = launch_wgrad<bf16, 2, 2, 1>(g, p, st);
  } else if (p.wco == 2) {
    rc = launch_wgrad<bf16, 2, 1, 2>(g, p, st);
  } else if (p.wci == 2) {
    rc = launch_wgrad<bf16, 1, 2, 2>(g, p, st);
  } else {
    rc = launch_wgrad<bf16, 1, 1, 4>(g, p, st);
  }
  if (rc != CY_OK) return rc;
  const int Cin = d->C1 + d->C2;
  if (Cin % 4) return CY_ERR_SHAPE;
  const long total = (long)d->Cout * (Cin / 4);  // one thread per four consecutive ci
  int SG = 1;  // slab groups per output: more when there are few outputs and many slabs
  while (SG < 32 && SG * 2 <= p.S && (total * SG) / 256 < 1024) SG *= 2;
  const int opb = 256 / SG;
  const int blocks = (int)((total + opb - 1) / opb);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw,
                     p.S, SG, d->Cout, Cin, p.co_pad, p.ci_pad, accumulate);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

int cy_conv3x3_wgrad(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                     const float* shift, const void* dy, float* dw, int accumulate, void* ws,
                     size_t ws_bytes, void* stream) {
  return wgrad_impl(d, src1, src2, scale, shift, dy, 0, nullptr, nullptr, nullptr, nullptr, nullptr, dw,
                    accumulate, ws, ws_bytes, stream);
}

size_t cy_conv3x3_wgrad_pair_ws_bytes(const cy_conv_desc* d, int n_b) {
  if (!d || n_b <= 0) return 0;
  cy_conv_desc dt = *d;
  dt.N += n_b;
  return cy_conv3x3_wgrad_ws_bytes(&dt);
}

int cy_conv3x3_wgrad_pair(const cy_conv_desc* d, const void* src1, const void* src2, const float* scale,
                          const float* shift, const void* dy, int n_b, const void* src1_b,
                          const void* src2_b, const float* scale_b, const float* shift_b,
                          const void* dy_b, float* dw, int accumulate, void* ws, size_t ws_bytes,
                          void* stream) {
  if (n_b <= 0) return CY_ERR_ARG;
  return wgrad_impl(d, src1, src2, scale, shift, dy, n_b, src1_b, src2_b, scale_b, shift_b, dy_b, dw,
                    accumulate, ws, ws_bytes, stream);
}

size_t cy_conv3x3_first_wgrad_ws_bytes(int N, int Cin, int H, int W, int Cout) {
  long nb = first_wgrad_blocks((long)N * H * W, W);
  const long nm = first_wgrad_mfma_blocks(N, Cin, H, W, Cout, CY_BF16);  // (either kernel's partials fit)
  if (nm > nb) nb = nm;
  return (size_t)nb * Cout * Cin * 9 * sizeof(float);
}

int cy_conv3x3_first_wgrad(const float* x, const void* dy, float* dw, int accumulate, int N, int Cin,
                           int H, int W, int Cout, int dy_dtype, void* ws, size_t ws_bytes,
                           void* stream) {
  if (!x || !dy || !dw || !ws) return CY_ERR_ARG;
  if (Cin < 1 || Cin > 4 || Cout % 8 || Cout > 64 || 256 % (Cout / 8)) return CY_ERR_SHAPE;
  if ((long)N * H * W >= (1L << 31)) return CY_ERR_SHAPE;
  if (ws_bytes < cy_conv3x3_first_wgrad_ws_bytes(N, Cin, H, W, Cout)) return CY_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (const int nm = first_wgrad_mfma_blocks(N, Cin, H, W, Cout, dy_dtype)) {
    const int nsub = N * H / 8, spb = (nsub + nm - 1) / nm;
    size_t smem = 4096 + (size_t)3 * 10 * W * 2;
    if (smem < 16384) smem = 16384;  // (the four waves' accumulators pass through it at the end)
    if (smem > 48 * 1024) {          // (wide images: W > 730)
      static bool big_done = false;
      if (!big_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(first_wgrad_mfma_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(first_wgrad_mfma_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return CY_ERR_LAUNCH;
        big_done = true;
      }
    }
    if (dy_dtype == CY_BF16)
      hipLaunchKernelGGL(first_wgrad_mfma_kernel<bf16>, dim3(nm), dim3(256), smem, st, x, (const bf16*)dy, (float*)ws, N, H, W, spb);
    else
      hipLaunchKernelGGL(first_wgrad_mfma_kernel<f16>, dim3(nm), dim3(256), smem, st, x, (const f16*)dy, (float*)ws, N, H, W, spb);
    CY_CHECK_LAUNCH();
    const int total = Cout * 9;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3(cy_cdiv(total, 4)), dim3(256), 0, st, (const float*)ws, dw, nm, total,
                       accumulate);
    CY_CHECK_LAUNCH();
    return CY_OK;
  }
  const int nblk = first_wgrad_blocks((long)N * H * W, W);
  const long per = ((long)N * H * W + nblk - 1) / nblk;
  const size_t smem = (size_t)(per / W + 4) * (W + 2) * sizeof(float);
  if (nblk < 0 || smem > 96 * 1024) return CY_ERR_SHAPE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(first_wgrad_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(first_wgrad_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(first_wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  if (dy_dtype == CY_BF16)
    hipLaunchKernelGGL(first_wgrad_kernel<bf16>, dim3(nblk), dim3(256), smem, st, x, (const bf16*)dy,
                       (float*)ws, N, Cin, H, W, Cout);
  else if (dy_dtype == CY_F16)
    hipLaunchKernelGGL(first_wgrad_kernel<f16>, dim3(nblk), dim3(256), smem, st, x, (const f16*)dy,
                       (float*)ws, N, Cin, H, W, Cout);
  else if (dy_dtype == CY_F32)
    hipLaunchKernelGGL(first_wgrad_kernel<float>, dim3(nblk), dim3(256), smem, st, x,
                       (const float*)dy, (float*)ws, N, Cin, H, W, Cout);
  else
    return CY_ERR_DTYPE;
  CY_CHECK_LAUNCH();
  const int total = Cout * Cin * 9;
  hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3(cy_cdiv(total, 4)), dim3(256), 0, st,
                     (const float*)ws, dw, nblk, total, accumulate);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // extern "C"
