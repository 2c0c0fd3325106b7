// HBM rate of a "two reads + one write" elementwise pass (the shape of the BatchNorm backward apply) under launch
// shapes and cache policies:   hipcc --offload-arch=gfx950 -O3 -o ew_bw ew_bw.hip && ./ew_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNR, bool NT> __global__ void __launch_bounds__(256) k_pass(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, long n) {
  const long stride = (long)gridDim.x * 256L;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
    u32x4 x[UNR], y[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if constexpr (NT) { x[u] = __builtin_nontemporal_load(a + i + u * stride); y[u] = __builtin_nontemporal_load(b + i + u * stride); }
      else { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      u32x4 r = x[u] ^ y[u];
      r.x += 0x10001u;
      if constexpr (NT) __builtin_nontemporal_store(r, o + i + u * stride); else o[i + u * stride] = r;
    }
  }
  for (; i < n; i += stride) o[i] = a[i] ^ b[i];
}
// contiguous per-workgroup ranges instead of grid-stride
template <int UNR, bool NT> __global__ void __launch_bounds__(256) k_range(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, long n) {
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per, end = beg + per < n ? beg + per : n;
  long i = beg + threadIdx.x;
  for (; i + (UNR - 1) * 256 < end; i += UNR * 256) {
    u32x4 x[UNR], y[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if constexpr (NT) { x[u] = __builtin_nontemporal_load(a + i + u * 256); y[u] = __builtin_nontemporal_load(b + i + u * 256); }
      else { x[u] = a[i + u * 256]; y[u] = b[i + u * 256]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      u32x4 r = x[u] ^ y[u];
      r.x += 0x10001u;
      if constexpr (NT) __builtin_nontemporal_store(r, o + i + u * 256); else o[i + u * 256] = r;
    }
  }
  for (; i < end; i += 256) o[i] = a[i] ^ b[i];
}
template <typename F> float timeit(F f, int it = 20) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(s);
  for (int i = 0; i < it; ++i) f();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms / it * 1e3f;
}
int main() {
  const long sizes[] = {102760448L, 51380224L, 25690112L, 12845056L, 6422528L};  // bytes of one tensor
  for (long bytes : sizes) {
    const long n = bytes / 16;
    // rotate over 4 buffer sets so that nothing is served from the 256 MB infinity cache by repetition
    const int NS = 4;
    std::vector<u32x4*> A(NS), B(NS), O(NS);
    for (int s = 0; s < NS; ++s) { hipMalloc(&A[s], bytes); hipMalloc(&B[s], bytes); hipMalloc(&O[s], bytes); hipMemset(A[s], 1, bytes); hipMemset(B[s], 2, bytes); }
    int rot = 0;
    printf("%6.1f MB/tensor:", bytes / 1e6);
#define RUN(KERN, G, LABEL) { float us = timeit([&] { hipLaunchKernelGGL(KERN, dim3(G), dim3(256), 0, 0, A[rot], B[rot], O[rot], n); rot = (rot + 1) % NS; }); printf("  %s %5.1f us %4.2f TB/s |", LABEL, us, 3.0 * bytes / us / 1e6); }
    const int full = (int)((n + 255) / 256);
    RUN((k_pass<1, false>), full > 8192 ? 8192 : full, "gs8192 u1");
    RUN((k_pass<1, false>), full > 1024 ? 1024 : full, "gs1024 u1");
    RUN((k_pass<2, false>), full > 1024 ? 1024 : full, "gs1024 u2");
    RUN((k_pass<4, false>), full > 1024 ? 1024 : full, "gs1024 u4");
    RUN((k_pass<2, false>), full > 2048 ? 2048 : full, "gs2048 u2");
    RUN((k_pass<4, true>), full > 1024 ? 1024 : full, "gs1024 u4 nt");
    RUN((k_pass<2, true>), full > 2048 ? 2048 : full, "gs2048 u2 nt");
    RUN((k_range<4, false>), 1024, "rng1024 u4");
    RUN((k_range<4, true>), 1024, "rng1024 u4 nt");
    RUN((k_range<4, true>), 2048, "rng2048 u4 nt");
    RUN((k_pass<1, false>), full, "full u1");
    printf("\n");
    for (int s = 0; s < NS; ++s) { hipFree(A[s]); hipFree(B[s]); hipFree(O[s]); }
  }
  return 0;
}
