// copy_ceiling2.hip — follow-up to copy_ceiling.hip at 1 GiB: larger grids, one tile per workgroup (no grid-stride
// loop), contiguous regions per workgroup.   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/cc2 tools/ubench/copy_ceiling2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool CONTIG>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  const size_t tile = (size_t)256 * U;
  const size_t ntiles = (n + tile - 1) / tile;
  const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const size_t t0 = CONTIG ? (size_t)blockIdx.x * per : blockIdx.x, t1 = CONTIG ? (t0 + per < ntiles ? t0 + per : ntiles) : ntiles;
  for (size_t t = t0; t < t1; t += CONTIG ? 1 : gridDim.x) {
    const size_t base = t * tile;
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; v[k] = __builtin_nontemporal_load(src + (i < n ? i : n - 1)); }
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n) __builtin_nontemporal_store(v[k], dst + i); }
  }
}
template <int U, bool CONTIG>
float run(const void* a, void* b, size_t bytes, int grid, int iters) {
  const size_t n = bytes / 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_k<U, CONTIG>), dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((copy_k<U, CONTIG>), dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return (float)(2.0 * bytes * iters / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  for (int grid : {4096, 16384, 32768, 65536, 131072, 262144})
    printf("grid %6d: stride U2 %.2f U4 %.2f U8 %.2f | contiguous U2 %.2f U4 %.2f U8 %.2f\n", grid,
           run<2, false>(a, b, bytes, grid, 20), run<4, false>(a, b, bytes, grid, 20), run<8, false>(a, b, bytes, grid, 20),
           run<2, true>(a, b, bytes, grid, 20), run<4, true>(a, b, bytes, grid, 20), run<8, true>(a, b, bytes, grid, 20));
  return 0;
}
