// copy_ceiling.hip — what a stream copy can reach on this box (the practical ceiling bench.py reports beside the
// 8 TB/s spec peak).  Variants: vectors in flight per lane (U), nontemporal loads / stores, grid size.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_ceiling tools/ubench/copy_ceiling.hip ; run: /tmp/copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  const size_t tile = (size_t)256 * U;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = base + (size_t)k * 256 + threadIdx.x;
      const size_t j = i < n ? i : n - 1;
      v[k] = NTL ? __builtin_nontemporal_load(src + j) : src[j];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = base + (size_t)k * 256 + threadIdx.x;
      if (i < n) { if (NTS) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
    }
  }
}

template <int U, bool NTL, bool NTS>
float run(const void* a, void* b, size_t bytes, int grid, int iters) {
  const size_t n = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_k<U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((copy_k<U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (float)(2.0 * bytes * iters / (ms * 1e-3) / 1e12);
}

int main() {
  for (size_t bytes : {(size_t)1 << 30, (size_t)1 << 28, (size_t)13 << 20}) {
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    const int iters = bytes >= ((size_t)1 << 28) ? 20 : 400;
    printf("== %zu MiB per direction (TB/s, read + write)\n", bytes >> 20);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
      printf("grid %5d:  U1 %.2f  U2 %.2f  U4 %.2f  U8 %.2f | nt-store U4 %.2f U8 %.2f | nt-both U4 %.2f U8 %.2f | nt-load U4 %.2f\n", grid,
             run<1, false, false>(a, b, bytes, grid, iters), run<2, false, false>(a, b, bytes, grid, iters),
             run<4, false, false>(a, b, bytes, grid, iters), run<8, false, false>(a, b, bytes, grid, iters),
             run<4, false, true>(a, b, bytes, grid, iters), run<8, false, true>(a, b, bytes, grid, iters),
             run<4, true, true>(a, b, bytes, grid, iters), run<8, true, true>(a, b, bytes, grid, iters),
             run<4, true, false>(a, b, bytes, grid, iters));
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
