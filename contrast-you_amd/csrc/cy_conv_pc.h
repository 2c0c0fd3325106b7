// Persistent producer / consumer form of the plane kernel (cy_conv_plane.h): forward and data gradient of
// the 3x3 convolutions whose width is a multiple of 14 (every level of the 224 x 224 U-Net), bf16 / f16.
//
// One 512-thread workgroup per CU walks a list of work units (output tile x cout block x K split):
//   waves 0-3  CONSUMERS: LDS fragments -> MFMA -> epilogue.  They never touch global memory on the load
//              side and never wait for anything but the stage barrier.
//   waves 4-7  PRODUCERS: keep the LDS ring full.  Activations (halo tile of a 64-channel chunk) go
//              global -> registers -> (BN+ReLU prologue / 2x2 max / nearest x2 / concat addressing) -> LDS,
//              requested a whole chunk (>= 2304 MFMA cycles) before they are committed; weights go by
//              LDS-DMA (global_load_lds_dwordx4) from an image that is packed stage-contiguously
//              (cy_conv3x3_pc_pack), 1 KiB per wave instruction, no registers, no ds_write.
// A "stage" is 36,864 bytes of weights: 9 taps x BN couts x KCB channels (KCB = 2048 / BN = 16 / 32 / 64
// for BN = 128 / 64 / 32), i.e. 2304 MFMA cycles per SIMD whatever the variant.  LDS: two activation
// buffers (64 channels each) + two weight slots = 156 KB.  One s_barrier per stage hands slot g+1 to the
// consumers and slot g-1 back to the producers; the pipeline runs across unit boundaries (the producers
// are already loading the next tile while the consumers store the current one), which is what a
// one-workgroup-per-tile launch cannot do: there, first-tile latency and the epilogue are ~40 % of a
// workgroup's life (in-kernel stamps of the plane kernel), and the dispatcher needs 16-24 us
// just to start the 7-10k workgroups of a 224 x 224 layer.
// BN statistics: the consumers keep per-channel sums of their units in registers and write ONE partial row
// per (workgroup, wave row) at the end: partials = workgroups x WGM instead of tiles.
#pragma once
#include "cy_conv_plane.h"
#include "cy_conv_tile.h"

namespace {

struct PcArgs {
  ConvArgs c;
  const void* wpc;   // packed [cout block][stage][tap][plane][BN][8 channels]
  int tiles;         // output tiles (16 x 14)
  int nblk;          // cout blocks
  int nst;           // stages over all input channels (ceil(Cin / KCB))
  int units;         // tiles * nblk * ksplit
  float inv_h;       // 1 / H
  float inv_ks, inv_tiles, inv_tiles_w;  // reciprocals for the division-free unit decode
  int xcd_chunked;   // 1: units are dealt to the 8 XCDs in contiguous ranges
  unsigned long long* stamps;  // development aid: shader-clock stamps of workgroup 0, [8 waves][128], or null
  int debug;         // what-if timing switches (CY_PC_DEBUG): 2 no activation loads, 4 no stores, 8 no weight DMA
};

template <int BN, int NCW = 4> struct PcCfg {
  static constexpr int NT = (NCW + 4) * 64;                             // consumer waves + four producer waves
  static constexpr int TH = 16, TW = 14, HP = 16;
  static constexpr int NPOS = (TH + 2) * HP;                            // 288 halo positions
  static constexpr int ZB = NPOS + 2;                                   // all-zero row
  // activation chunk: 64 channels with four consumer waves; 32 channels with eight (BN = 128), which buys a
  // THIRD weight slot: the LDS-DMA of a 36 KB stage needs ~2000 cycles to issue and ~1000 to land (in-kernel
  // stamps) -- with two slots it must start and finish inside one stage interval and the consumers wait for it
  static constexpr int KCA = NCW == 8 ? 32 : 64;
  static constexpr int CPA = KCA / 8;                                   // planes per activation chunk
  static constexpr int CPAL = CPA == 8 ? 3 : 2;                         // log2(CPA)
  static constexpr int NBS = NCW == 8 ? 3 : 2;                          // weight slots
  static constexpr int SKEW = 16 / CPA;
  static constexpr int APL = ((ZB + 18 - SKEW + 15) / 16) * 16 + SKEW;  // positions per plane (322 / 308)
  static constexpr int APLB = APL * 16;
  static constexpr int A_BYTES = CPA * APLB;                            // 41,216 / 19,712
  static constexpr int KCB = 2048 / BN < KCA ? 2048 / BN : KCA;         // channels per weight stage
  static constexpr int PPB = KCB / 8;                                   // planes per stage
  static constexpr int KSB = KCB / 16;                                  // MFMA k-steps per stage and tap
  static constexpr int SPA = KCA / KCB;                                 // stages per activation chunk
  static constexpr int BPLB = BN * 16;                                  // bytes per weight plane (one tap)
  static constexpr int TAPB = PPB * BPLB;                               // bytes per tap
  static constexpr int B_BYTES = 9 * TAPB;                              // 36,864
  static constexpr int NPW = B_BYTES / 1024 / 4;                        // LDS-DMA instructions per producer wave and stage
  static constexpr int WGN = BN == 128 ? 2 : 1, WGM = NCW / WGN;
  static constexpr int M_REP = 8 / WGM, N_REP = BN / (32 * WGN);
  static_assert(WGM * WGN == NCW && M_REP >= 1 && N_REP >= 1, "consumer wave grid");
  static constexpr int RSTEP = 256 / (CPA * 16);                        // halo rows between a producer thread's items
  static constexpr int NA = (TH + 2 + RSTEP - 1) / RSTEP;               // items per producer thread (9 / 5)
  static constexpr int DEPTH = BN == 128 ? 2 : 3;                        // fragment register sets of the consumers
  static constexpr int COEF_MAX = 512;                                  // channels with a BN+ReLU prologue
  static constexpr int SMEM = 2 * A_BYTES + NBS * B_BYTES + 2 * COEF_MAX * 4;
  static_assert(B_BYTES == 36864 && B_BYTES % 4096 == 0 && SMEM <= 160 * 1024, "stage geometry");
};

// workgroup barrier that leaves the `VM` youngest vector-memory operations of this wave in flight
// (producers: the activation loads of the next chunk; LDS-DMA and loads count together, in issue order)
template <int VM> __device__ __forceinline__ void pc_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}
// consumers: their LDS reads are done; their global STORES (epilogue of the previous unit) may stay in flight
__device__ __forceinline__ void pc_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// keeps a value alive without cost (what-if timing builds must not let the compiler delete its producers)
template <typename V> __device__ __forceinline__ void pc_keep(const V& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"v"(v));
#endif
}

template <int I, int N, typename F> __device__ __forceinline__ void pc_static_for(F&& f) {
  if constexpr (I < N) {
    f(TapC<I>{});
    pc_static_for<I + 1, N>(f);
  }
}

// the unit list of workgroup `wg`: `it` = 0, 1, ... -> unit index, or -1 when exhausted
struct PcUnits {
  int first, stride, end;
  __device__ __forceinline__ int at(int it) const {
    const int u = first + it * stride;
    return u < end ? u : -1;
  }
};
__device__ __forceinline__ PcUnits pc_units(const PcArgs& a, int wg, int nwg) {
  PcUnits q;
  if (a.xcd_chunked) {  // XCD x (= wg % 8, dispatch round robin) owns a contiguous range of units
    const int per = (a.units + 7) / 8;
    const int x = wg & 7;
    q.first = x * per + (wg >> 3);
    q.stride = nwg >> 3;
    q.end = (x + 1) * per < a.units ? (x + 1) * per : a.units;
  } else {
    q.first = wg, q.stride = nwg, q.end = a.units;
  }
  return q;
}

struct PcUnit {
  int tile, blk, s0, s1;  // stage range [s0, s1) of the input channels (split-K)
  int z;
};
// x / d for 0 <= x < 2^21 with inv = 1.0f / d (exact: the quotient's fractional part is at least 0.5 / d away
// from an integer, the float error is ~1e-7 * x / d)
__device__ __forceinline__ int pc_fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

__device__ __forceinline__ PcUnit pc_decode(const PcArgs& a, int u) {
  // unit order: cout block major, then tile, then K split (neighbouring tiles and the splits of a tile run
  // at the same time on the same XCD and share halo rows / weights in its L2)
  PcUnit q;
  const int ks = a.c.ksplit;
  const int t = ks == 1 ? u : pc_fdiv(u, a.inv_ks);
  q.z = u - t * ks;
  q.blk = a.nblk == 1 ? 0 : pc_fdiv(t, a.inv_tiles);
  q.tile = t - q.blk * a.tiles;
  q.s0 = (a.nst * q.z) / ks;
  q.s1 = (a.nst * (q.z + 1)) / ks;
  return q;
}

// in-kernel stamps of workgroup 0 (debug bit 16): [wave][slot] shader-clock values, read back by
// cy_debug_pc_stamps
static unsigned long long* g_pc_stamp_buf = nullptr;  // device buffer [8][128], set by cy_debug_pc_stamps
#define PC_STAMP(slot_)                                                                              \
  do {                                                                                               \
    const int sl__ = (slot_);                                                                        \
    if (a.stamps && blockIdx.x == 0 && lane == 0 && sl__ < 128) a.stamps[wave * 128 + sl__] = __builtin_amdgcn_s_memtime(); \
  } while (0)

template <typename T, int BN, int NCW>
__global__ void __launch_bounds__((NCW + 4) * 64, NCW == 8 ? 3 : 2) conv3x3_pc_kernel(const PcArgs a) {
  using C = PcCfg<BN, NCW>;
  using M = Mma<T>;
  constexpr int EPC = 8;
  constexpr int M_REP = C::M_REP, N_REP = C::N_REP, WGN = C::WGN;
  constexpr int APLB = C::APLB, BPLB = C::BPLB, KCB = C::KCB, PPB = C::PPB, KSB = C::KSB, SPA = C::SPA;
  constexpr int TW = C::TW, TH = C::TH;
  constexpr int DEPTH = C::DEPTH;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA0 = smem;
  unsigned char* const sB0 = smem + 2 * C::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = a.c.C1 + a.c.C2;
  const PcUnits ul = pc_units(a, blockIdx.x, gridDim.x);

  // BN+ReLU prologue coefficients of all input channels: LDS-resident (the producers' register budget is
  // the consumers': two activation chunks in flight leave no room for 16 more floats per set)
  float* const s_coef = reinterpret_cast<float*>(smem + 2 * C::A_BYTES + C::NBS * C::B_BYTES);
  if (a.c.prologue) {
    for (int c = tid; c < a.c.C1 && c < C::COEF_MAX; c += C::NT) {
      s_coef[c] = a.c.scale[c];
      s_coef[C::COEF_MAX + c] = a.c.shift[c];
    }
  }
  // zero rows of both activation buffers (never overwritten: halo positions end at index NPOS)
  for (int idx = tid; idx < 2 * C::CPA * 18; idx += C::NT) {
    const int b = idx / (C::CPA * 18), rem = idx % (C::CPA * 18);
    st16(sA0 + b * C::A_BYTES + (rem / 18) * APLB + (C::ZB + rem % 18) * 16, u32x4{0u, 0u, 0u, 0u});
  }

  if (wave >= NCW) {
    // ======================================= PRODUCERS =======================================
    // They share each SIMD's issue port with one consumer wave that runs its MFMA bursts at priority 1;
    // at equal or lower priority their address / transform arithmetic gets the leftover slots only and a
    // chunk commit takes 4-8k cycles (in-kernel stamps).  They are memory-bound and mostly asleep: let
    // them win the port whenever they are awake.
    __builtin_amdgcn_s_setprio(3);
    const int pt = tid - NCW * 64;
    const int pw = wave - NCW;
    constexpr int RSTEP = C::RSTEP;
    const int ch = pt & (C::CPA - 1);           // plane (8 channels) of this thread's items
    const int hc = (pt >> C::CPAL) & 15;        // halo column
    const int hr0 = pt >> (C::CPAL + 4);        // halo rows hr0, hr0 + RSTEP, ... (NA of them; rows >= 18 do not exist)
    const T* s1p = reinterpret_cast<const T*>(a.c.src1);
    const T* s2p = reinterpret_cast<const T*>(a.c.src2);
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.wpc);

    // ---- iterator over this workgroup's (unit, activation chunk) sequence ----
    int p_it = 0;            // position in the unit list
    PcUnit pu;               // current unit
    int p_unit = ul.at(0);
    if (p_unit >= 0) pu = pc_decode(a, p_unit);
    int p_chunk = 0;         // chunk inside the unit (64 channels = SPA stages)
    int p_R0 = 0, p_w0 = 0;
    auto unit_geom = [&]() {
      const int rt = pc_fdiv(pu.tile, a.inv_tiles_w);
      p_R0 = rt * TH;
      p_w0 = (pu.tile - rt * a.c.tiles_w) * TW;
    };
    if (p_unit >= 0) unit_geom();
    auto chunks_of = [&](const PcUnit& q) { return (q.s1 - q.s0 + SPA - 1) / SPA; };
    auto advance_chunk = [&]() {
      if (p_unit < 0) return;
      if (++p_chunk >= chunks_of(pu)) {
        p_chunk = 0;
        p_unit = ul.at(++p_it);
        if (p_unit >= 0) {
          pu = pc_decode(a, p_unit);
          unit_geom();
        }
      }
    };

    // two register sets: the chunk that is committed next and the one after it are both in flight
    // (set k feeds activation buffer k); what a set holds is described by its RegChunk
    struct RegChunk {
      u32x4 v[C::NA];
      unsigned aok;
      int unit, c0, R0, w0, len;  // len: stages the chunk covers
    };
    RegChunk rc0, rc1;
    rc0.unit = rc1.unit = -1;
    rc0.len = rc1.len = 1 << 28;
    rc0.aok = rc1.aok = 0;
    // the nine halo rows of this thread, R0 - 1 + hr0 + 2 i: (image, row in image) of the first one by ONE
    // reciprocal multiplication, the others by stepping; row base (pixel index of column 0) in source 1 / 2
    struct RowIt {
      int R, n, hh;
    };
    auto row_begin = [&](int R0) -> RowIt {
      RowIt it;
      const int Rb = R0 - 1 + hr0 + RSTEP;  // (= first row + RSTEP >= 1: the first row itself may be -1)
      const int nb = pc_fdiv(Rb, a.inv_h);
      it.R = Rb - RSTEP;
      it.n = nb;
      it.hh = Rb - nb * a.c.H - RSTEP;
      if (it.hh < 0) it.hh += a.c.H, --it.n;
      return it;
    };
    auto row_next = [&](RowIt& it) {
      it.R += RSTEP;
      it.hh += RSTEP;
      if (it.hh >= a.c.H) it.hh -= a.c.H, ++it.n;
    };
    auto row_base = [&](const RowIt& it, int i, bool second) -> int {
      if (hr0 + RSTEP * i >= TH + 2) return -1;  // (item beyond the halo tile)
      if (it.R < 0 || it.R >= a.c.NH) return -1;
      if (second || a.c.mode1 == CY_SRC_DIRECT) return it.R * a.c.W;
      if (a.c.mode1 == CY_SRC_POOL2) return (it.n * 2 * a.c.H + 2 * it.hh) * (2 * a.c.W);
      return (it.n * (a.c.H >> 1) + (it.hh >> 1)) * (a.c.W >> 1);
    };
    // current (unit, chunk) -> registers, then advance.  Returns the number of load instructions issued:
    // NA or 0, the same for every lane (the barrier's vmcnt count relies on it: loads are unconditional,
    // lanes without a source pixel read the tensor's first chunk and are zeroed at commit)
    auto a_request = [&](RegChunk& rc) -> int {
      rc.unit = p_unit;
      rc.aok = 0;
      rc.len = 1 << 28;
      if (p_unit < 0) return 0;
      const int sbeg = pu.s0 + p_chunk * SPA;  // absolute stage of the chunk's first channel
      rc.len = pu.s1 - sbeg < SPA ? pu.s1 - sbeg : SPA;
      rc.c0 = sbeg * KCB;
      rc.R0 = p_R0, rc.w0 = p_w0;
      int issued = 0;
      if (a.c.mode1 != CY_SRC_POOL2 && !(a.debug & 2)) {  // (2x2 max, C2 == 0 checked by the host: loaded in a_commit)
        const int cabs = rc.c0 + ch * EPC;
        const bool cvalid = cabs < Cin;
        const bool in2 = cvalid && cabs >= a.c.C1;
        const T* base = !cvalid ? s1p : (in2 ? s2p + (cabs - a.c.C1) : s1p + cabs);
        const int ld = in2 ? a.c.ld2 : a.c.ld1;
        const int wsh = (!in2 && a.c.mode1 == CY_SRC_UP2) ? 1 : 0;
        const int w = rc.w0 - 1 + hc;
        const bool wok = cvalid && w >= 0 && w < a.c.W;
        RowIt rit = row_begin(rc.R0);
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
          const int rp = row_base(rit, i, in2);
          row_next(rit);
          const bool ok = wok && rp >= 0;
          rc.v[i] = ld16(base + (ok ? (size_t)(rp + (w >> wsh)) * ld : (size_t)0));
          rc.aok |= (ok ? 1u : 0u) << i;
        }
        issued = C::NA;
      }
      advance_chunk();
      return issued;
    };
    auto a_commit = [&](const RegChunk& rc, unsigned char* sA) {  // registers -> LDS (with the load-side transforms)
      if (rc.unit < 0) return;
      const int cabs = rc.c0 + ch * EPC;
      unsigned char* dstp = sA + ch * APLB + 16 + (hc + 16 * hr0) * 16;  // position index = 1 + lin
      if (cabs < a.c.C1 && a.c.mode1 == CY_SRC_POOL2) {
        const int w = rc.w0 - 1 + hc;
        const bool wok = w >= 0 && w < a.c.W;
        const size_t rowstep = (size_t)(2 * a.c.W) * a.c.ld1;
        RowIt rit = row_begin(rc.R0);
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {  // (one item at a time: four loads in flight; the register budget)
          const int rp = row_base(rit, i, false);
          row_next(rit);
          if (hr0 + RSTEP * i >= TH + 2) continue;
          u32x4 o = {0u, 0u, 0u, 0u};
          if (wok && rp >= 0) {
            const T* p = s1p + (size_t)(rp + 2 * w) * a.c.ld1 + cabs;
            const u32x4 v0 = ld16(p), v1 = ld16(p + a.c.ld1), v2 = ld16(p + rowstep), v3 = ld16(p + rowstep + a.c.ld1);
            float f0[EPC], f1[EPC];
            Chunk<T>::unpack(v0, f0);
            Chunk<T>::unpack(v1, f1);
#pragma unroll
            for (int q = 0; q < EPC; ++q) f0[q] = fmaxf(f0[q], f1[q]);
            Chunk<T>::unpack(v2, f1);
#pragma unroll
            for (int q = 0; q < EPC; ++q) f0[q] = fmaxf(f0[q], f1[q]);
            Chunk<T>::unpack(v3, f1);
#pragma unroll
            for (int q = 0; q < EPC; ++q) f0[q] = fmaxf(f0[q], f1[q]);
            o = Chunk<T>::pack(f0);
          }
          st16(dstp + i * RSTEP * 16 * 16, o);
        }
        return;
      }
      const bool pro = a.c.prologue && cabs < a.c.C1;
      float sc[EPC], sh[EPC];
      if (pro) {
        const f32x4* cs = reinterpret_cast<const f32x4*>(s_coef + cabs);
        const f32x4* ch_ = reinterpret_cast<const f32x4*>(s_coef + C::COEF_MAX + cabs);
        const f32x4 a0 = cs[0], a1 = cs[1], b0 = ch_[0], b1 = ch_[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = a0[j], sc[4 + j] = a1[j], sh[j] = b0[j], sh[4 + j] = b1[j];
      }
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        u32x4 v = ((rc.aok >> i) & 1u) ? rc.v[i] : u32x4{0u, 0u, 0u, 0u};
        if (pro && ((rc.aok >> i) & 1u)) {
          float f[EPC];
          Chunk<T>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(sc[j], f[j], sh[j]), 0.f);
          v = Chunk<T>::pack(f);
        }
        if (hr0 + RSTEP * i < TH + 2) st16(dstp + i * RSTEP * 16 * 16, v);  // rows hr0 + RSTEP i
      }
    };

    // ---- iterator over the (unit, stage) sequence for the weights ----
    int b_it = 0, b_stage = 0;
    PcUnit bu;
    int b_unit = ul.at(0);
    if (b_unit >= 0) bu = pc_decode(a, b_unit);
    // weights resident: one cout block, no K split, and the unit's stages map onto the slots the same way in
    // every unit (one stage: every slot holds it; NBS stages: slot = stage) -> loaded once
    const bool resident = a.nblk == 1 && a.c.ksplit == 1 && (a.nst == 1 || a.nst == C::NBS);
    int b_slot = 0;  // slot of the next stage to issue (= global stage % NBS)
    auto b_issue = [&](int g) -> bool {  // weights of global stage g -> slot g % NBS (LDS-DMA), then advance
      if (b_unit < 0) return false;
      const bool dma = !(resident && g >= C::NBS) && !(a.debug & 8);
      const int slot = b_slot;
      b_slot = b_slot + 1 == C::NBS ? 0 : b_slot + 1;
      if (dma) {
        const unsigned char* src =
            wbase + ((size_t)bu.blk * a.nst + (bu.s0 + b_stage)) * C::B_BYTES + (pw * C::NPW) * 1024 + lane * 16;
        unsigned char* dst = sB0 + slot * C::B_BYTES + (pw * C::NPW) * 1024;
#pragma unroll
        for (int j = 0; j < C::NPW; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                           (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
      }
      if (++b_stage >= bu.s1 - bu.s0) {
        b_stage = 0;
        b_unit = ul.at(++b_it);
        if (b_unit >= 0) bu = pc_decode(a, b_unit);
      }
      return dma;
    };

    // total stages of this workgroup (= barriers)
    int gtot = 0;
    for (int it = 0;; ++it) {
      const int u = ul.at(it);
      if (u < 0) break;
      const PcUnit q = pc_decode(a, u);
      gtot += q.s1 - q.s0;
    }
    __syncthreads();  // zero rows written (all 512 threads)

    if (gtot == 0) return;  // (a workgroup without units: its consumers execute no barrier either)
    // Barrier with the right vmcnt.  The weights of the NEXT stage must have landed; they were issued LEAD - 1
    // intervals ago, so the LDS-DMA of this interval (NPW instructions, stage g + LEAD) and the activation
    // loads issued after it (n_new = 0 or NA) may stay in flight.
    constexpr int LEAD = C::NBS - 1;
    auto sync = [&](bool dma, int n_new) {
      if (LEAD == 1) {
        if (!dma) pc_barrier_lds();
        else if (n_new) pc_barrier<C::NA>();
        else pc_barrier<0>();
      } else {
        if (dma && n_new) pc_barrier<C::NPW + C::NA>();
        else if (dma) pc_barrier<C::NPW>();
        else if (n_new) pc_barrier<C::NA>();
        else pc_barrier<0>();
      }
    };
    // prologue: weights of the first LEAD stages in flight, chunk 0 committed, chunks 1 and 2 in registers
    bool dma = false;
    for (int l = 0; l < LEAD; ++l) dma = b_issue(l);
    asm volatile("" ::: "memory");  // (pins the issue order the vmcnt counts assume)
    a_request(rc0);
    a_request(rc1);
    a_commit(rc0, sA0);
    int ga = 1;                       // global index of the chunk that is committed next (set / buffer ga & 1)
    int next_chunk_stage = rc0.len;   // global stage at which chunk `ga` is first read
    asm volatile("" ::: "memory");
    a_request(rc0);
    pc_barrier<0>();
    int pst = 0;
    PC_STAMP(pst++);
    for (int g = 0; g + 1 < gtot; ++g) {
      // consumers compute stage g; slot (g+LEAD) % NBS and -- at a chunk boundary -- buffer ga&1 are free.
      // Order inside a chunk-boundary interval: commit FIRST (the compiler's wait for the set's registers is
      // a vmcnt(0): it must not include the LDS-DMA of this interval), then the DMA, then the next request
      int n_new = 0;
      const bool boundary = g + 1 == next_chunk_stage;
      if (boundary) {
        if (ga & 1) a_commit(rc1, sA0 + C::A_BYTES); else a_commit(rc0, sA0);
        asm volatile("" ::: "memory");
      }
      PC_STAMP(pst++);
      dma = b_issue(g + LEAD);
      if (boundary) {
        asm volatile("" ::: "memory");  // the next request's loads are YOUNGER than this stage's LDS-DMA
        if (ga & 1) {
          next_chunk_stage += rc1.len;
          n_new = a_request(rc1);
        } else {
          next_chunk_stage += rc0.len;
          n_new = a_request(rc0);
        }
        ++ga;
      }
      PC_STAMP(pst++);
      sync(dma, n_new);
      PC_STAMP(pst++);
    }
    return;
  }

  // ========================================= CONSUMERS =========================================
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;
  int bbase[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) bbase[n] = ((wn * N_REP + n) * 32 + r) * 16 + h * BPLB;

  auto frag = [&](const unsigned char* p) {
    typename M::Frag f;
    f.v = *reinterpret_cast<const decltype(f.v)*>(p);
    return f;
  };

  // BN statistics of this wave's units.  BN = 128 (128 accumulator registers): reduced per unit over the
  // lanes (reduce-scatter, 62 shuffles per 32 couts) into two registers per 32 couts.  BN <= 64: the raw
  // per-lane sums stay in registers across units and are reduced ONCE, when the partial row is written --
  // the 224 x 224 / 112 x 112 layers have thousands of short units, where the shuffles cost as much as the
  // unit's MFMAs.
  constexpr bool LAZY = BN == 32;
  constexpr int NLZ = LAZY ? N_REP : 1;
  float lz1[NLZ][16], lz2[NLZ][16];
#pragma unroll
  for (int n = 0; n < NLZ; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) lz1[n][i] = lz2[n][i] = 0.f;
  float tot1[N_REP], tot2[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) tot1[n] = tot2[n] = 0.f;
  // reduce-scatter over the 32 lanes of each half (cy_conv_plane.h): afterwards lane `l` holds in s[0] the
  // total of register index ((l>>4)&1)*8 + ((l>>3)&1)*4 + ((l>>2)&1)*2 + ((l>>1)&1)
  auto scatter = [&](float (&s1)[16], float (&s2)[16]) {
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
  };
  int stat_blk = -1;
  unsigned visited = 0;
  const bool do_stats = a.c.stats != nullptr && a.c.ksplit == 1;
  float* const stat_row = a.c.stats + (size_t)(blockIdx.x * C::WGM + wm) * 2 * a.c.Cout;
  auto flush_stats = [&]() {
    if (stat_blk < 0) return;
    if constexpr (LAZY) {
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
        scatter(lz1[n], lz2[n]);
        tot1[n] = lz1[n][0], tot2[n] = lz2[n][0];
#pragma unroll
        for (int i = 0; i < 16; ++i) lz1[n][i] = lz2[n][i] = 0.f;
      }
    }
    if ((lane & 1) == 0) {
      const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
        const int co = stat_blk * BN + (wn * N_REP + n) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (co < a.c.Cout) {
          stat_row[co] = tot1[n];
          stat_row[a.c.Cout + co] = tot2[n];
        }
      }
    }
#pragma unroll
    for (int n = 0; n < N_REP; ++n) tot1[n] = tot2[n] = 0.f;
  };

  __syncthreads();  // zero rows written

  // Deferred output stores (eight consumer waves): a unit's packed bf16 outputs stay in registers and are
  // stored two per stage during the NEXT unit's first stages, between its MFMAs.  With every workgroup in
  // step, immediate stores make the whole chip write its tiles at the same moment (in-kernel stamps: 19-25k
  // cycles per epilogue, matrix pipes idle) and then compute with HBM idle.
  constexpr bool DEFER = NCW == 8;
  constexpr int NPEND = DEFER ? M_REP * N_REP * 2 : 1;
  u32x4 pend[NPEND];
  int pend_from = NPEND;  // first entry of `pend` not stored yet (NPEND: nothing pending)
  int pend_R0 = 0, pend_w0 = 0, pend_n0 = 0;
  auto store_pending = [&](auto K) {  // entry k = ((n * M_REP + m) * 2 + half)
    constexpr int k = decltype(K)::value;
    constexpr int n = k / (M_REP * 2), m = (k / 2) % M_REP, gq = (k % 2) * 2;
    const int qq = (wm * M_REP + m) * 32 + r;
    const int hx = qq & 15;
    const int R = pend_R0 + (qq >> 4), w = pend_w0 + hx - 1;
    const int co = pend_n0 + (wn * N_REP + n) * 32 + 8 * gq + 8 * h;
    if (hx >= 1 && hx <= TW && R < a.c.NH && w < a.c.W && co < a.c.Cout) {
      const size_t gp = (size_t)R * a.c.W + w;
      T* o1 = reinterpret_cast<T*>(a.c.out);
      T* o2 = reinterpret_cast<T*>(a.c.out2);
      T* dst = (a.c.split_c > 0 && co >= a.c.split_c) ? o2 + gp * a.c.ldo2 + (co - a.c.split_c) : o1 + gp * a.c.ldo + co;
      *reinterpret_cast<u32x4*>(dst) = pend[k];
    }
  };
  auto drain_pending = [&](int count) {  // store the next `count` pending entries
    if constexpr (DEFER) {
      const int to = pend_from + count < NPEND ? pend_from + count : NPEND;
      pc_static_for<0, NPEND>([&](auto K) {
        if (decltype(K)::value >= pend_from && decltype(K)::value < to) store_pending(K);
      });
      pend_from = to;
    }
  };
  int cst = 0;
  int c_slot = 0;  // weight slot of the next stage (= global stage % NBS)
  int g = 0;   // global stage counter
  int ga = 0;  // global activation chunk counter (buffer = ga & 1)
  for (int it = 0;; ++it) {
    const int unit = ul.at(it);
    if (unit < 0) break;
    const PcUnit q = pc_decode(a, unit);
    const int rt_ = pc_fdiv(q.tile, a.inv_tiles_w);
    const int R0 = rt_ * TH, w0 = (q.tile - rt_ * a.c.tiles_w) * TW;
    const int n0 = q.blk * BN;

    // per-lane fragment bases: [m][dh+1]; a row that is the first / last of its image reads the zero row
    int abase[M_REP][3];
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int ty = 2 * (wm * M_REP + m) + (r >> 4);
      const int hx = r & 15;
      const int R = R0 + ty;
      int flag = 3;
      if (R < a.c.NH) {
        const int n = (int)(((float)R + 0.5f) * a.inv_h);
        const int hh = R - n * a.c.H;
        flag = (hh == 0 ? 1 : 0) | (hh == a.c.H - 1 ? 2 : 0);
      }
      const int hoff = h * APLB;
      const int mid = (ty + 1) * 16 + hx;
      const int zer = C::ZB + hx;
      abase[m][0] = ((flag & 1) ? zer : mid - 16) * 16 + hoff;
      abase[m][1] = mid * 16 + hoff;
      abase[m][2] = ((flag & 2) ? zer : mid + 16) * 16 + hoff;
    }

    f32x16 acc[M_REP][N_REP];
#pragma unroll
    for (int m = 0; m < M_REP; ++m)
#pragma unroll
      for (int n = 0; n < N_REP; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int ns = q.s1 - q.s0;
    for (int s = 0; s < ns; ++s, ++g) {
      PC_STAMP(cst++);
      pc_barrier_lds();
      PC_STAMP(cst++);
      const int sub = s % SPA;  // stage inside the activation chunk
      const unsigned char* sA = sA0 + ((ga + s / SPA) & 1) * C::A_BYTES + sub * PPB * APLB;
      const unsigned char* sB = sB0 + c_slot * C::B_BYTES;
      c_slot = c_slot + 1 == C::NBS ? 0 : c_slot + 1;
      // the 9 x KS (tap, k-step) steps of the stage as ONE software pipeline, DEPTH register sets deep:
      // the fragments of step i + DEPTH - 1 are requested before the MFMAs of step i are issued (order
      // pinned by sched_barrier), across tap boundaries too.  The consumer wave is alone on its SIMD's
      // matrix pipe, so an LDS round trip (~200-250 cycles with the producers' stores in the queue) that
      // is not covered by MFMAs of earlier steps is dead time: 8 MFMAs (256 cycles) per step barely cover
      // one round trip, 2-4 MFMAs per step (the small-cout variants) need three steps of cover.
      // KS < KSB: a stage whose upper channels lie beyond Cin (32 input channels in a 64-channel stage).
      auto run_stage = [&](auto KSC) {
        constexpr int KS = decltype(KSC)::value;
        constexpr int NSTEP = 9 * KS;
        typename M::Frag af[DEPTH][M_REP], bf[DEPTH][N_REP];
        auto load_step = [&](auto IDX) {
          constexpr int idx = decltype(IDX)::value, set = idx % DEPTH;
          constexpr int tap = idx / KS, ks = idx % KS;
          constexpr int d = tap / 3, dw = tap % 3 - 1;
#pragma unroll
          for (int n = 0; n < N_REP; ++n) bf[set][n] = frag(sB + tap * C::TAPB + ks * 2 * BPLB + bbase[n]);
#pragma unroll
          for (int m = 0; m < M_REP; ++m) af[set][m] = frag(sA + (dw + 1) * 16 + ks * 2 * APLB + abase[m][d]);
        };
        pc_static_for<0, DEPTH - 1>([&](auto I) { load_step(I); });
        pc_static_for<0, NSTEP>([&](auto IDX) {
          constexpr int idx = decltype(IDX)::value;
          constexpr int cur = idx % DEPTH;
          // the LDS reads of step idx + DEPTH - 1 are issued BETWEEN the MFMAs of step idx (one read per
          // MFMA gap: an MFMA occupies the SIMD's issue port for 8 of its 32 cycles), not as a burst in
          // front of them -- the matrix pipe would idle for the length of the burst
          if constexpr (idx + DEPTH - 1 < NSTEP) load_step(TapC<idx + DEPTH - 1>{});
#pragma unroll
          for (int m = 0; m < M_REP; ++m)
#pragma unroll
            for (int n = 0; n < N_REP; ++n) M::mma(bf[cur][n], af[cur][m], acc[m][n]);  // rows = couts
          constexpr int NM = M_REP * N_REP, NR = (idx + DEPTH - 1 < NSTEP) ? M_REP + N_REP : 0;
          pc_static_for<0, NM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            // reads spread over the gaps: gap j gets reads [j*NR/NM, (j+1)*NR/NM)
            constexpr int nr = ((j + 1) * NR) / NM - (j * NR) / NM;
            if constexpr (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);  // DS reads
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      if constexpr (KSB >= 2) {
        if ((q.s0 + s) * KCB + KCB / 2 >= Cin) run_stage(TapC<KSB / 2>{});
        else run_stage(TapC<KSB>{});
      } else {
        run_stage(TapC<KSB>{});
      }
      if (DEFER && pend_from < NPEND) drain_pending(2);
    }
    ga += (ns + SPA - 1) / SPA;
    if (DEFER && pend_from < NPEND) drain_pending(NPEND);  // (a unit shorter than the drain)

    PC_STAMP(cst++);
    // ---------------- epilogue: accumulators -> NHWC, straight from registers ----------------
    auto position = [&](int m, int& R, int& w) -> bool {
      const int qq = (wm * M_REP + m) * 32 + r;
      const int hx = qq & 15;
      R = R0 + (qq >> 4);
      w = w0 + hx - 1;
      return hx >= 1 && hx <= TW && R < a.c.NH && w < a.c.W;
    };
    if (a.debug & 4) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int n = 0; n < N_REP; ++n) pc_keep(acc[m][n]);
      continue;
    }
    if (a.c.ksplit > 1) {
      float* wsz = a.c.ws + (size_t)q.z * ((size_t)a.c.NH * a.c.W) * a.c.Cout;
#pragma unroll
      for (int m = 0; m < M_REP; ++m) {
        int R, w;
        if (!position(m, R, w)) continue;
        float* dst = wsz + ((size_t)R * a.c.W + w) * a.c.Cout;
#pragma unroll
        for (int n = 0; n < N_REP; ++n)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int co = n0 + (wn * N_REP + n) * 32 + 8 * gq + 4 * h;
            if (co < a.c.Cout)
              *reinterpret_cast<f32x4*>(dst + co) = f32x4{acc[m][n][4 * gq], acc[m][n][4 * gq + 1],
                                                          acc[m][n][4 * gq + 2], acc[m][n][4 * gq + 3]};
          }
      }
      continue;
    }
    if (do_stats && stat_blk != q.blk) {
      flush_stats();
      stat_blk = q.blk;
      visited |= 1u << q.blk;
    }
    T* o1 = reinterpret_cast<T*>(a.c.out);
    T* o2 = reinterpret_cast<T*>(a.c.out2);
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      float s1[16], s2[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
      for (int m = 0; m < M_REP; ++m) {
        int R, w;
        const bool ok = position(m, R, w);
        const size_t gp = (size_t)R * a.c.W + w;
        u32x2 packed[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          T pk[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[m][n][4 * gq + j];
            pk[j] = from_f32<T>(v);
            if (do_stats && ok) {
              const float qv = to_f32<T>(pk[j]);
              s1[4 * gq + j] += qv;
              s2[4 * gq + j] += qv * qv;
            }
          }
          packed[gq] = __builtin_bit_cast(u32x2, *reinterpret_cast<const s16x4*>(pk));
        }
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
          u32x2 lo = packed[gq], hi = packed[gq + 1];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(lo[j], hi[j], false, false);
            lo[j] = sw[0];
            hi[j] = sw[1];
          }
          if constexpr (DEFER) {
            pend[(n * M_REP + m) * 2 + gq / 2] = u32x4{lo[0], lo[1], hi[0], hi[1]};
          } else {
            const int co = n0 + (wn * N_REP + n) * 32 + 8 * gq + 8 * h;
            if (ok && co < a.c.Cout) {
              T* dst = (a.c.split_c > 0 && co >= a.c.split_c) ? o2 + gp * a.c.ldo2 + (co - a.c.split_c)
                                                              : o1 + gp * a.c.ldo + co;
              *reinterpret_cast<u32x4*>(dst) = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
          }
        }
      }
      if (do_stats) {
        if constexpr (LAZY) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            lz1[n][i] += s1[i];
            lz2[n][i] += s2[i];
          }
        } else {
          scatter(s1, s2);
          tot1[n] += s1[0];
          tot2[n] += s2[0];
        }
      }
    }
    if constexpr (DEFER) pend_from = 0, pend_R0 = R0, pend_w0 = w0, pend_n0 = n0;
  }
  if (DEFER && pend_from < NPEND) drain_pending(NPEND);
  if (do_stats) {
    flush_stats();
    // cout blocks this workgroup never produced: their columns of its partial rows are zero
    if ((lane & 1) == 0) {
      const int reg = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
      for (int b = 0; b < a.nblk; ++b) {
        if ((visited >> b) & 1u) continue;
#pragma unroll
        for (int n = 0; n < N_REP; ++n) {
          const int co = b * BN + (wn * N_REP + n) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (co < a.c.Cout) {
            stat_row[co] = 0.f;
            stat_row[a.c.Cout + co] = 0.f;
          }
        }
      }
    }
  }
}

// ---- weights: reference layout -> stage-contiguous image --------------------------------------------
// wpc[blk][stage][tap][plane][BN rows][8 channels]; dgrad: roles of co / ci swapped, taps flipped
template <typename T>
__global__ void __launch_bounds__(256)
    pack_weights_pc_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int bn, int dgrad) {
  const int kcb = 2048 / bn, ppb = kcb / 8;
  const int Co = dgrad ? Cin : Cout, Ci = dgrad ? Cout : Cin;  // geometry of the GEMM this image feeds
  const int nblk = (Co + bn - 1) / bn, nst = (Ci + kcb - 1) / kcb;
  const long items = (long)nblk * nst * 9 * ppb * bn;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < items; i += (long)gridDim.x * 256L) {
    const int row = (int)(i % bn);
    long t = i / bn;
    const int plane = (int)(t % ppb);
    t /= ppb;
    const int tap = (int)(t % 9);
    t /= 9;
    const int st = (int)(t % nst);
    const int blk = (int)(t / nst);
    const int co = blk * bn + row;
    const int ci0 = st * kcb + plane * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = ci0 + j;
      float v = 0.f;
      if (co < Co && ci < Ci) v = dgrad ? w[((size_t)ci * Cin + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap];
      f[j] = v;
    }
    st16(out + i * 8, Chunk<T>::pack(f));
  }
}

inline int pc_bn_for(int Cout) { return Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32); }
inline long pc_packed_elems(int Cout, int Cin) {  // elements of the image that feeds a Cout x Cin GEMM
  const int bn = pc_bn_for(Cout), kcb = 2048 / bn;
  return (long)cy_cdiv(Cout, bn) * cy_cdiv(Cin, kcb) * 9 * kcb * bn;
}

struct PcPlan {
  int bn, ncw, tiles, nblk, nst, ksplit, units, grid, partials;
  size_t ws_bytes;
  int finish_blocks;
};

inline PcPlan plan_pc(int N, int H, int W, int Cin, int Cout) {
  PcPlan p;
  p.bn = pc_bn_for(Cout);
  static const int ncw128 = [] {
    const char* e = getenv("CY_PC_NCW");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  p.ncw = p.bn == 128 ? ncw128 : 4;
  const int kcb = 2048 / p.bn;
  p.tiles = cy_cdiv((long)N * H, 16) * (W / 14);
  p.nblk = cy_cdiv(Cout, p.bn);
  p.nst = cy_cdiv(Cin, kcb);
  const int base = p.tiles * p.nblk;
  // K split: the launch takes ceil(units / 256) rounds of (stages per unit + fixed per-unit cost); a split
  // shortens the units but adds f32 partial traffic and the finish kernel.  Costs in stage times (~1 us).
  int Z = 1;
  if (base < 256 && p.nst >= 2 && Cout % 8 == 0) {
    double best = 1e30;
    for (int z = 1; z <= 16 && z <= p.nst; ++z) {
      const int rounds = cy_cdiv((long)base * z, 256);
      const double cost = rounds * (cy_cdiv(p.nst, z) + 3.0) + (z > 1 ? 5.0 + 0.5 * z : 0.0);
      if (cost < best - 1e-9) best = cost, Z = z;
    }
  }
  if (const char* ov = getenv("CY_KSPLIT")) {
    const int z = atoi(ov);
    if (z >= 1 && Cout % 8 == 0) Z = z > p.nst ? p.nst : z;
  }
  p.ksplit = Z;
  p.units = base * Z;
  p.grid = p.units < 256 ? p.units : 256;
  const long npix = (long)N * H * W;
  long fb = (npix + 15) / 16;
  if (fb > 1024) fb = 1024;
  p.finish_blocks = (int)fb;
  p.partials = Z > 1 ? p.finish_blocks : p.grid * (p.bn == 128 ? p.ncw / 2 : 4);
  p.ws_bytes = Z > 1 ? (size_t)Z * npix * Cout * sizeof(float) : 0;
  return p;
}

template <typename T, int BN, int NCW = 4> int launch_conv_pc(PcArgs a, const PcPlan& p, hipStream_t st) {
  using C = PcCfg<BN, NCW>;
  auto kern = conv3x3_pc_kernel<T, BN, NCW>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  if (a.c.W % C::TW != 0) return CY_ERR_SHAPE;
  a.c.tiles_w = a.c.W / C::TW;
  a.xcd_chunked = (p.grid % 8 == 0 && p.units >= 8 * (p.grid / 8)) ? 1 : 0;
  static const int dbg = [] {
    const char* e = getenv("CY_PC_DEBUG");
    return e ? atoi(e) : 0;
  }();
  a.debug = dbg;
  a.stamps = g_pc_stamp_buf;
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(C::NT), C::SMEM, st, a);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

}  // namespace
