// What would summing the weight gradient's slabs INSIDE the weight-gradient kernel buy?  (lead (b))
// 256 workgroups (one per CU, 768 threads, like wgrad12s) each write a 64 x 64 x 9 f32 slab (147 KB, 37.7 MB in all);
//   A: a second kernel sums the S slabs of every (co, ci) block                      (what the library does)
//   B: the workgroups of a block meet at a barrier (agent-scope release / acquire, bounded spin) and each sums ITS
//      1/S slice of the block over the S slabs; a second kernel only copies the nblocks x 147 KB result
// hipcc --offload-arch=gfx950 -O3 -o slab_reduce slab_reduce.hip && ./slab_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int SLAB = 64 * 64 * 9;  // floats
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void write_slab(float* slab, int wg) {
  for (int i = threadIdx.x; i < SLAB / 4; i += blockDim.x) {
    const float v = (float)((wg * 131 + i) & 1023) * (1.f / 1024.f);
    reinterpret_cast<f32x4*>(slab)[i] = f32x4{v, v + 1.f, v + 2.f, v + 3.f};
  }
}
__global__ void __launch_bounds__(768) k_write(float* ws) { write_slab(ws + (size_t)blockIdx.x * SLAB, blockIdx.x); }

// A: thread = one 16-byte chunk of one block's result, sums the S slabs (wgid = s * nblocks + blk) in order
__global__ void __launch_bounds__(256) k_reduce(const float* ws, float* out, int nblocks, int S) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)nblocks * (SLAB / 4)) return;
  const int blk = (int)(e / (SLAB / 4)), i = (int)(e % (SLAB / 4));
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int q = 0; q < S; ++q) s += reinterpret_cast<const f32x4*>(ws + ((size_t)q * nblocks + blk) * SLAB)[i];
  reinterpret_cast<f32x4*>(out + (size_t)blk * SLAB)[i] = s;
}

// B
template <int VARIANT>
__global__ void __launch_bounds__(768) k_write_group_reduce(float* ws, float* part, unsigned* counter, unsigned* timeouts, int nblocks, int S) {
  const int wg = blockIdx.x, blk = wg % nblocks, sp = wg / nblocks;
  write_slab(ws + (size_t)wg * SLAB, wg);
  if (VARIANT == 0) __threadfence();  // release by every thread: this workgroup's slab is visible to the agent
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    // (VARIANT 1: one release for the workgroup -- cumulative over the barrier -- and a slower poll)
    __hip_atomic_fetch_add(counter + blk, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int good = 0;
    for (int it = 0; it < 4000000; ++it) {  // bounded: a wave that never sees its group must still drain
      if (__hip_atomic_load(counter + blk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)S) { good = 1; break; }
      if (VARIANT == 0) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(64);
    }
    if (!good) atomicAdd(timeouts, 1u);
    ok = good;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (!ok) return;
  // slice sp of the block: chunks [sp * per, (sp + 1) * per)
  const int per = (SLAB / 4 + S - 1) / S;
  const int c0 = sp * per, c1 = c0 + per < SLAB / 4 ? c0 + per : SLAB / 4;
  for (int i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int q = 0; q < S; ++q) s += reinterpret_cast<const f32x4*>(ws + ((size_t)q * nblocks + blk) * SLAB)[i];
    reinterpret_cast<f32x4*>(part + (size_t)blk * SLAB)[i] = s;
  }
}
__global__ void __launch_bounds__(256) k_copy(const float* part, float* out, long n4) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e < n4) reinterpret_cast<f32x4*>(out)[e] = reinterpret_cast<const f32x4*>(part)[e];
}

template <typename F> float timeit(F f, int it = 20) {
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(s);
  for (int i = 0; i < it; ++i) f();
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / it * 1e3f;
}
int main() {
  float *ws, *part, *outA, *outB; unsigned *counter, *timeouts;
  (void)hipMalloc(&ws, 256UL * SLAB * 4); (void)hipMalloc(&part, 64UL * SLAB * 4); (void)hipMalloc(&outA, 64UL * SLAB * 4); (void)hipMalloc(&outB, 64UL * SLAB * 4);
  (void)hipMalloc(&counter, 64 * 4); (void)hipMalloc(&timeouts, 4); (void)hipMemset(timeouts, 0, 4);
  const float tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(768), 0, 0, ws); });
  printf("slab write alone: %.1f us\n", tw);
  for (int nblocks : {1, 4, 16, 64}) {
    const int S = 256 / nblocks;
    const long n4 = (long)nblocks * (SLAB / 4);
    const float ta = timeit([&] {
      hipLaunchKernelGGL(k_write, dim3(256), dim3(768), 0, 0, ws);
      hipLaunchKernelGGL(k_reduce, dim3((int)((n4 + 255) / 256)), dim3(256), 0, 0, ws, outA, nblocks, S);
    });
    const float tb = timeit([&] {
      (void)hipMemsetAsync(counter, 0, 64 * 4, 0);
      hipLaunchKernelGGL(k_write_group_reduce<0>, dim3(256), dim3(768), 0, 0, ws, part, counter, timeouts, nblocks, S);
      hipLaunchKernelGGL(k_copy, dim3((int)((n4 + 255) / 256)), dim3(256), 0, 0, part, outB, n4);
    });
    const float tc = timeit([&] {
      (void)hipMemsetAsync(counter, 0, 64 * 4, 0);
      hipLaunchKernelGGL(k_write_group_reduce<1>, dim3(256), dim3(768), 0, 0, ws, part, counter, timeouts, nblocks, S);
      hipLaunchKernelGGL(k_copy, dim3((int)((n4 + 255) / 256)), dim3(256), 0, 0, part, outB, n4);
    });
    (void)hipDeviceSynchronize();
    std::vector<float> a(n4 * 4), b(n4 * 4);
    (void)hipMemcpy(a.data(), outA, n4 * 16, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), outB, n4 * 16, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < n4 * 4; ++i) bad += a[i] != b[i];
    unsigned to = 0; (void)hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
    printf("blocks %2d x splits %3d: A write + reduce %6.1f us | B write + group reduce + copy %6.1f us (one release per workgroup, slow poll: %6.1f) | mismatching floats %ld, timeouts %u\n", nblocks, S, ta, tb, tc, bad, to);
  }
  return 0;
}
