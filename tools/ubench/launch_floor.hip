// Micro-benchmark (runs ON THE GPU BOX): back-to-back dependent launches of a kernel that only touches one record per
// thread, for different grid shapes / dynamic LDS sizes — what part of a step launch is the launch floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_touch(uint32_t* rec, int n) {
  extern __shared__ uint32_t sh[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) sh[0] = 1;
  if (i < n) rec[i] += 1;
}
int main() {
  const int n = 65536;
  uint32_t* rec; CK(hipMalloc(&rec, n * 4 * 2)); CK(hipMemset(rec, 0, n * 4 * 2));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct { int grid, block, lds; } cfg[] = {{1024, 64, 0}, {1024, 64, 18000}, {1536, 128, 18000}, {1024, 128, 18000}, {1024, 128, 0}, {768, 256, 36000},
                                            {512, 256, 36000}, {384, 512, 64000}, {256, 512, 64000}, {2048, 64, 9000}, {1536, 64, 0}};
  for (int rep = 0; rep < 3; ++rep)
  for (auto c : cfg) {
    CK(hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_touch, dim3(c.grid), dim3(c.block), c.lds, 0, rec, n);
    CK(hipEventRecord(a));
    const int K = 500;
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_touch, dim3(c.grid), dim3(c.block), c.lds, 0, rec, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("grid %5d x block %4d, lds %6d B: %.2f us/launch\n", c.grid, c.block, c.lds, ms / K * 1e3);
  }
  return 0;
}
