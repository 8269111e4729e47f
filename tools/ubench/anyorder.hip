// Micro-benchmark (runs ON THE GPU BOX): is hipExtAnyOrderLaunch (AQL packet without the barrier bit) honoured on
// gfx950?  Back-to-back launches of a ~5 us kernel with and without the flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(uint32_t* rec, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = i;
  for (int k = 0; k < iters; ++k) v = v * 1664525u + 1013904223u;
  if (i < n) rec[i] += v;
}
int main() {
  const int n = 65536;
  uint32_t* rec; CK(hipMalloc(&rec, n * 4)); CK(hipMemset(rec, 0, n * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int iters : {0, 2000}) for (int flag : {0, 1, 0, 1}) {
    for (int i = 0; i < 50; ++i) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, flag, rec, n, iters);
    CK(hipEventRecord(a, st));
    const int K = 500;
    for (int i = 0; i < K; ++i) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, flag, rec, n, iters);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("iters %5d anyorder %d: %.2f us/launch\n", iters, flag, ms / K * 1e3);
  }
  return 0;
}
