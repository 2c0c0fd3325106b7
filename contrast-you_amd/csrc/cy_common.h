// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the
// Contrast-You hot path.  Everything here is wave64 / MFMA specific; there is
// no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/contrastyou_hip.h"

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CY_WAVE 64

#define CY_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return CY_ERR_LAUNCH;            \
  } while (0)

// ---- element traits -------------------------------------------------------
template <typename T> struct ElemTr;
template <> struct ElemTr<bf16> {
  static constexpr int EPC = 8;  // elements per 16-byte chunk
};
template <> struct ElemTr<f16> {
  static constexpr int EPC = 8;
};
template <> struct ElemTr<float> {
  static constexpr int EPC = 4;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) {
  return __uint_as_float(lo16 << 16);
}

// unpack a 16-byte chunk of T into floats (8 for bf16, 4 for f32)
template <typename T> struct Chunk;
template <> struct Chunk<bf16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16 lo = (bf16)f[2 * i];
      bf16 hi = (bf16)f[2 * i + 1];
      uint16_t l16 = __builtin_bit_cast(uint16_t, lo);
      uint16_t h16 = __builtin_bit_cast(uint16_t, hi);
      v[i] = (uint32_t)l16 | ((uint32_t)h16 << 16);
    }
    return v;
  }
};
template <> struct Chunk<f16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const u32x4& v, float* f) {
    const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (f16)f[i];
    return __builtin_bit_cast(u32x4, h);
  }
};
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void unpack(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
    return v;
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }

// round a float through the storage type (identity for f32)
template <typename T> __device__ __forceinline__ float round_through(float v) {
  return to_f32<T>(from_f32<T>(v));
}

__device__ __forceinline__ u32x4 ld16(const void* p) {
  return *reinterpret_cast<const u32x4*>(p);
}
// streaming read: the line is not wanted in the caches after this use (the last reader of a large tensor)
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ void st16(void* p, const u32x4& v) {
  *reinterpret_cast<u32x4*>(p) = v;
}

// wave64 sum over all lanes (result in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cy_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int cy_roundup(int a, int b) { return ((a + b - 1) / b) * b; }
