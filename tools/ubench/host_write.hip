// host_write.hip — how fast can a step's results reach page-locked host memory?  (round 4: the gymnasium-contract path of
// TetrisVectorEnv.step moves 13.1 MB (board observation) / 62 MB (dict) per step at 65 536 envs.)
//   a) hipMemcpyAsync device → pinned host, one stream                        (what tg_fetch does)
//   b) the same bytes as two halves on two streams                            (two SDMA engines?)
//   c) a copy KERNEL whose destination is the pinned host buffer (device-mapped): stores cross PCIe directly
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/host_write tools/ubench/host_write.hip ; run: /tmp/host_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    if (NT) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t sizes[] = {13107200, 14155776, 61865984};
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (size_t bytes : sizes) {
    void *d, *h, *hw;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 7, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&hw, bytes, hipHostMallocNonCoherent));
    memset(h, 0, bytes); memset(hw, 0, bytes);
    auto timeit = [&](const char* name, auto fn) {
      for (int i = 0; i < 3; ++i) fn();
      double best = 1e30, sum = 0;
      const int R = 20;
      for (int i = 0; i < R; ++i) { const double t0 = now_us(); fn(); const double t = now_us() - t0; sum += t; if (t < best) best = t; }
      printf("%9zu B  %-46s  mean %8.1f us  best %8.1f us  = %5.1f GB/s (best)\n", bytes, name, sum / R, best, bytes / best / 1e3);
    };
    timeit("hipMemcpyAsync D2H pinned, 1 stream + sync", [&] { CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); });
    timeit("hipMemcpyAsync D2H non-coherent pinned", [&] { CK(hipMemcpyAsync(hw, d, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); });
    timeit("two halves on two streams + 2 syncs", [&] {
      const size_t half = (bytes / 2) & ~(size_t)4095;
      CK(hipMemcpyAsync(h, d, half, hipMemcpyDeviceToHost, s0));
      CK(hipMemcpyAsync((char*)h + half, (char*)d + half, bytes - half, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    });
    for (int grid : {64, 256, 1024, 4096}) {
      char nm[96];
      snprintf(nm, sizeof nm, "copy kernel -> host-mapped (coherent), grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(grid), dim3(256), 0, s0, (const u32x4*)d, (u32x4*)h, bytes / 16); CK(hipStreamSynchronize(s0)); });
      snprintf(nm, sizeof nm, "copy kernel -> host-mapped nontemporal, grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(grid), dim3(256), 0, s0, (const u32x4*)d, (u32x4*)h, bytes / 16); CK(hipStreamSynchronize(s0)); });
    }
    timeit("copy kernel -> non-coherent host, grid 1024", [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(1024), dim3(256), 0, s0, (const u32x4*)d, (u32x4*)hw, bytes / 16); CK(hipStreamSynchronize(s0)); });
    if (((unsigned char*)h)[bytes - 1] != 7 || ((unsigned char*)hw)[bytes - 1] != 7) printf("DATA MISMATCH\n");
    CK(hipFree(d)); CK(hipHostFree(h)); CK(hipHostFree(hw));
  }
  return 0;
}
