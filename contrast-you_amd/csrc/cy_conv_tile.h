// Shared pieces of the 3x3 implicit-GEMM kernels (forward / data-gradient in
// cy_conv3x3.hip, weight-gradient in cy_wgrad.hip): the kernel argument block,
// the MFMA fragment abstraction and the LDS halo-tile stager.
//
// Data layout in LDS (one input-channel chunk of KC = PITCHB/sizeof(T)
// channels at a time):
//   halo tile  sA[slot][col][PITCHB bytes]   slot = 0..TH+1 are the TH+2 image
//              rows R0-1 .. R0+TH (flattened row index R = n*H + h), slot TH+2
//              is an all-zero row used for "the neighbour row is outside this
//              image"; col = 0..TW+1 are image columns w0-1 .. w0+TW with
//              out-of-image columns written as zeros.
//   inside a pixel the 16-byte chunks are XOR-swizzled with the LDS pixel
//   index so that the 16 lanes a ds_read_b128 services together hit 16
//   different 16-byte slots of the 256-byte bank row.
#pragma once
#include "cy_bn_acc.h"

struct ConvArgs {
  const void* src1;
  const void* src2;
  const float* scale;
  const float* shift;
  const void* w;  // packed [9][w_co_pad][w_ci_pad]
  const void* wflow;  // stage-contiguous image behind it (cy_conv_flow.h), or null
  long long bytes_w;  // ... its size in bytes
  void* out;
  void* out2;
  float* stats;  // [tiles][2][Cout] or null
  int N, H, W, NH;
  int C1, C2, Cout;
  int mode1, prologue;
  int ld1, ld2, ldo, ldo2, split_c;
  int tiles_w;  // number of column tiles
  int w_co_pad, w_ci_pad;
  int full_tiles;  // 1: every tile is entirely inside the image grid
  float* ws;       // split-K: f32 partial outputs [ksplit][N*H*W][Cout]
  int ksplit;      // number of K splits over input-channel chunks (blockIdx.z)
  int xcd_remap;   // plane kernel: workgroup b -> tile such that the 8 XCDs own contiguous bands of tiles
  long long bytes1, bytes2;    // sizes of the source tensors in bytes (buffer descriptors of the LDS-DMA kernels), or 0
  long long bytes_o1, bytes_o2, bytes_st;  // ... of the outputs and the statistics partials
  unsigned long long* stamps;  // development aid (cy_debug_conv_stamps): shader-clock stamps of workgroup 0, or null
  // BatchNorm without finalize launches (cy_bn_acc.h):
  BnFold fold;               // fold.acc != null: the prologue coefficients come from the previous layer's sums
  unsigned long long* sacc;  // != null: the statistics are ADDED into this accumulator [sR][4][Cout] (no partial rows)
  int sR;
  // prologue == 2 (data gradient with the BatchNorm + ReLU backward in its load path, flow kernel): source 1 is dA, the
  // gradient w.r.t. relu(bn(y)); dy = scale * dA * [scale * y + shift > 0] + k1 * y + k0 is formed in LDS from (dA, y) with
  // (k1, k0) derived from the backward sums' accumulator, convolved, and also written to dy_out for the weight gradient
  const void* ysrc;    // y: the forward layer's raw conv output, same geometry and pitch as source 1
  long long bytes_y;
  BnBwdFold bfold;     // bfold.coef = the forward pass's [5][C1]
  void* dy_out;        // [N,H,W,C1], pitch ld1
  long long bytes_dy;
  // data gradient whose output is the dA of a BatchNorm + ReLU (flow kernel epilogue, cy_conv3x3_dgrad_dz): the sums of
  // dz = dA * [scale * y + shift > 0] and dz * xhat of the output channels [dz_c0, dz_c0 + dz_C) are added into dz_acc
  // from the values in registers -- the separate reduce pass over (dA, y) goes
  const void* dz_y;         // y of that BatchNorm [N,H,W,dz_C], pitch dz_ld; null: off
  const float* dz_coef;     // its forward [5][dz_C]
  unsigned long long* dz_acc;
  int dz_ld, dz_c0, dz_C, dz_R;
};

// ---- MFMA fragment abstraction: one "k-step" is 16 input channels ----------
template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int NCHUNK = 1;  // 16-byte chunks per lane per k-step
  struct Frag {
    bf16x8 v;
  };
  __device__ __forceinline__ static Frag load(const unsigned char* pix, int fi, int swz) {
    Frag f;
    f.v = *reinterpret_cast<const bf16x8*>(pix + ((fi ^ swz) << 4));
    return f;
  }
  __device__ __forceinline__ static void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
  }
};
template <> struct Mma<f16> {
  static constexpr int NCHUNK = 1;
  struct Frag {
    f16x8 v;
  };
  __device__ __forceinline__ static Frag load(const unsigned char* pix, int fi, int swz) {
    Frag f;
    f.v = *reinterpret_cast<const f16x8*>(pix + ((fi ^ swz) << 4));
    return f;
  }
  __device__ __forceinline__ static void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int NCHUNK = 2;
  struct Frag {
    f32x4 lo, hi;
  };
  __device__ __forceinline__ static Frag load(const unsigned char* pix, int fi, int swz) {
    Frag f;
    f.lo = *reinterpret_cast<const f32x4*>(pix + (((2 * fi) ^ swz) << 4));
    f.hi = *reinterpret_cast<const f32x4*>(pix + (((2 * fi + 1) ^ swz) << 4));
    return f;
  }
  // lane (r,h) holds k = 8h+j (j=0..7); instruction j sums k=j (h=0) and k=8+j (h=1):
  // any pairing works as long as A and B use the same one.
  __device__ __forceinline__ static void mma(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], c, 0, 0, 0);
  }
};

// swizzle of the 16-byte chunk index inside an LDS "pixel" (row of PITCHB bytes)
template <int PITCHB> __device__ __forceinline__ int lds_swz(int p) {
  constexpr int CPP = PITCHB / 16;
  constexpr int PPR = 16 / CPP;  // pixels per 256-byte bank row
  return (p / PPR) & (CPP - 1);
}

// per-block row tables: pixel index of (row, w=0) in source 1 / 2, or -1
// restrict_img: rows that belong to another image than row R0 are invalid (used
// by tilings that never span images, where they stand for zero padding).
__device__ __forceinline__ void conv_row_tables(const ConvArgs& a, int TH, int R0, int tid,
                                                int* s_row1, int* s_row2, int* s_flag,
                                                bool restrict_img) {
  if (tid < TH + 2) {
    const int R = R0 - 1 + tid;
    int r1 = -1, r2 = -1;
    bool ok = R >= 0 && R < a.NH;
    if (ok && restrict_img) ok = (R / a.H) == (R0 / a.H);
    if (ok) {
      const int n = R / a.H;
      const int hh = R - n * a.H;
      if (a.mode1 == CY_SRC_DIRECT)
        r1 = R * a.W;
      else if (a.mode1 == CY_SRC_POOL2)
        r1 = (n * 2 * a.H + 2 * hh) * (2 * a.W);
      else
        r1 = (n * (a.H >> 1) + (hh >> 1)) * (a.W >> 1);
      r2 = R * a.W;
    }
    s_row1[tid] = r1;
    s_row2[tid] = r2;
  }
  if (tid < TH) {
    const int R = R0 + tid;
    int f = 3;
    if (R < a.NH) {
      const int hh = R % a.H;
      f = (hh == 0 ? 1 : 0) | (hh == a.H - 1 ? 2 : 0);
    }
    s_flag[tid] = f;
  }
}

// Stage the (TH+2)x(TW+2) halo tile of input channels [c0, c0+KC) into sA.
// Loads are issued in batches of four independent 16-byte requests per thread before any of
// them is consumed (a one-load-per-iteration loop serialises on L2/HBM latency).
// SWZ: 0 = linear, 1 = forward-kernel chunk swizzle (lds_swz), 2 = "pair" swizzle for the
// transposed reads of the weight-gradient kernel (128-byte pixels: the two 64-byte halves of
// every second pixel pair are exchanged, so 4 consecutive pixels x 64 B cover all 64 banks).
// HP = LDS row pitch of the halo tile in pixels (>= TW+2).
template <int PITCHB, int SWZ> __device__ __forceinline__ int halo_chunk(int pix, int ch) {
  if (SWZ == 1) return ch ^ lds_swz<PITCHB>(pix);
  if (SWZ == 2) return ch ^ (((pix >> 1) & 1) << 2);
  return ch;
}

template <typename T, int PITCHB, int SWZ>
__device__ __forceinline__ void conv_stage_halo(const ConvArgs& a, unsigned char* sA,
                                                const int* s_row1, const int* s_row2, int TH,
                                                int TW, int HP, int w0, int c0, int tid) {
  constexpr int EPC = ElemTr<T>::EPC;
  constexpr int CPP = PITCHB / 16;
  constexpr int NB = 4;  // loads in flight per thread
  const int HW2 = TW + 2;
  const int NCH = (TH + 2) * HW2 * CPP;
  const int ch = tid & (CPP - 1);  // constant per thread (256 % CPP == 0)
  const int cabs = c0 + ch * EPC;
  const bool in2 = cabs >= a.C1;
  const bool cvalid = cabs < a.C1 + a.C2;
  const T* s1 = reinterpret_cast<const T*>(a.src1);
  const T* s2 = reinterpret_cast<const T*>(a.src2);

  if (a.mode1 == CY_SRC_POOL2 && !in2) {
    // 2x2 max on load: four requests per chunk are already independent
    for (int idx = tid; idx < NCH; idx += 256) {
      const int lin = idx / CPP;
      const int hr = lin / HW2;
      const int hc = lin - hr * HW2;
      const int pix = hr * HP + hc;
      const int w = w0 - 1 + hc;
      u32x4 v = {0u, 0u, 0u, 0u};
      const int rp = s_row1[hr];
      if (cvalid && w >= 0 && w < a.W && rp >= 0) {
        const T* p = s1 + (size_t)(rp + 2 * w) * a.ld1 + cabs;
        const size_t rowstep = (size_t)(2 * a.W) * a.ld1;
        u32x4 v00 = ld16(p), v01 = ld16(p + a.ld1), v10 = ld16(p + rowstep),
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
      st16(sA + pix * PITCHB + (halo_chunk<PITCHB, SWZ>(pix, ch) << 4), v);
    }
    return;
  }

  // direct / nearest-upsample / second (concat) source: one request per chunk
  const T* base = in2 ? s2 + (cabs - a.C1) : s1 + cabs;
  const int ld = in2 ? a.ld2 : a.ld1;
  const int* rtab = in2 ? s_row2 : s_row1;
  const int wsh = (!in2 && a.mode1 == CY_SRC_UP2) ? 1 : 0;
  const bool pro = a.prologue && !in2 && cvalid;
  float sc[EPC], sh[EPC];
  if (pro) {
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      sc[j] = a.scale[cabs + j];
      sh[j] = a.shift[cabs + j];
    }
  }
  for (int idx0 = tid; idx0 < NCH; idx0 += 256 * NB) {
    u32x4 v[NB];
    int dst[NB];
    bool ok[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int idx = idx0 + b * 256;
      const int lin = idx / CPP;
      const int hr = lin / HW2;
      const int hc = lin - hr * HW2;
      const int pix = hr * HP + hc;
      const int w = w0 - 1 + hc;
      dst[b] = idx < NCH ? pix * PITCHB + (halo_chunk<PITCHB, SWZ>(pix, ch) << 4) : -1;
      const int rp = idx < NCH ? rtab[hr] : -1;
      ok[b] = cvalid && rp >= 0 && w >= 0 && w < a.W;
      v[b] = u32x4{0u, 0u, 0u, 0u};
      if (ok[b]) v[b] = ld16(base + (size_t)(rp + (w >> wsh)) * ld);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (dst[b] >= 0) {
        if (pro && ok[b]) {
          float f[EPC];
          Chunk<T>::unpack(v[b], f);
#pragma unroll
          for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(sc[j], f[j], sh[j]), 0.f);
          v[b] = Chunk<T>::pack(f);
        }
        st16(sA + dst[b], v[b]);
      }
    }
  }
}
