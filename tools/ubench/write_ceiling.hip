// write_ceiling.hip — pure store bandwidth (no reads): what the write-dominated observation surfaces (DICT 62 MB, RGB 160 MB
// per transition at 65 536 envs) can reach at best.  16-byte and 12-byte stores per lane, plain / nontemporal.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/write_ceiling tools/ubench/write_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(4))) u32x3_a4;

template <bool NT>
__global__ __launch_bounds__(256) void fill16(u32x4* dst, size_t n, uint32_t v) {
  const u32x4 x = u32x4{v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, dst + i); else dst[i] = x;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill12(uint8_t* dst, size_t n, uint32_t v) {  // n = number of 12-byte pieces
  const u32x3 x = u32x3{v, v + 1, v + 2};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x3_a4*>(dst + 12 * i)); else *reinterpret_cast<u32x3_a4*>(dst + 12 * i) = x;
  }
}
template <typename F>
float timed(F launch, size_t bytes, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return (float)((double)bytes * iters / (ms * 1e-3) / 1e12);
}
int main() {
  for (size_t mb : {13, 62, 160, 512, 2048}) {
    const size_t bytes = mb << 20;
    void* a; hipMalloc(&a, bytes);
    const int iters = mb <= 160 ? 200 : 20;
    for (int grid : {2048, 8192}) {
      printf("%5zu MiB grid %5d:  16B plain %.2f  16B nt %.2f  12B plain %.2f  12B nt %.2f TB/s\n", mb, grid,
             timed([&] { hipLaunchKernelGGL(fill16<false>, dim3(grid), dim3(256), 0, 0, (u32x4*)a, bytes / 16, 7u); }, bytes, iters),
             timed([&] { hipLaunchKernelGGL(fill16<true>, dim3(grid), dim3(256), 0, 0, (u32x4*)a, bytes / 16, 7u); }, bytes, iters),
             timed([&] { hipLaunchKernelGGL(fill12<false>, dim3(grid), dim3(256), 0, 0, (uint8_t*)a, bytes / 12, 7u); }, bytes / 12 * 12, iters),
             timed([&] { hipLaunchKernelGGL(fill12<true>, dim3(grid), dim3(256), 0, 0, (uint8_t*)a, bytes / 12, 7u); }, bytes / 12 * 12, iters));
    }
    hipFree(a);
  }
  return 0;
}
