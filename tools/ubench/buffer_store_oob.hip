// buffer_store_oob.hip — a raw-buffer store whose offset lies outside the descriptor's num_records is dropped by the hardware
// (gfx950, descriptor word 3 = 0x00020000): the predicated store without an exec-mask region.  Checks it, and times 13 stores
// per lane predicated (a) by exec-mask regions (`if (c) global store`), (b) by an out-of-range offset.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bso tools/ubench/buffer_store_oob.hip && /tmp/bso
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned char* dst, const u32x4* src, unsigned long long* cyc, unsigned mask) {
  unsigned char* base = dst + (size_t)blockIdx.x * 13312;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 13312, 0x00020000);
  u32x4 v[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) v[i] = src[threadIdx.x + 64 * i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    const bool c = ((v[i].x ^ mask) & 7u) == 0;  // 1 lane in 8 stores
    if (MODE == 0) {
      if (c) __builtin_nontemporal_store(v[i], reinterpret_cast<u32x4*>(base) + threadIdx.x + 64 * i);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(v[i], r, c ? (threadIdx.x + 64 * i) * 16u : 0x80000000u, 0, 2);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int blocks = 1024;
  unsigned char* d; u32x4* s; unsigned long long* c;
  hipMalloc(&d, (size_t)blocks * 13312 + (1 << 20)); hipMalloc(&s, 64 * 13 * 16); hipMalloc(&c, blocks * 8);
  std::vector<unsigned> hs(64 * 13 * 4);
  for (size_t i = 0; i < hs.size(); ++i) hs[i] = (unsigned)(i * 2654435761u >> 7);
  hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned char> out[2];
  for (int m = 0; m < 2; ++m) {
    hipMemset(d, 0xEE, (size_t)blocks * 13312 + (1 << 20));
    for (int rep = 0; rep < 3; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, s, c, 0u);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, s, c, 0u);
      if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: kernel failed\n", m); return 1; }
    }
    std::vector<unsigned long long> hc(blocks);
    hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : hc) sum += (double)x;
    out[m].resize((size_t)blocks * 13312 + (1 << 20));
    hipMemcpy(out[m].data(), d, out[m].size(), hipMemcpyDeviceToHost);
    printf("%-48s: %7.1f cycles for 13 predicated 16-byte stores per lane (issue side)\n", m == 0 ? "exec-mask regions (if (c) global_store)" : "raw buffer store, out-of-range offset when !c", sum / blocks);
  }
  size_t stored = 0; for (auto b : out[1]) stored += b != 0xEE;
  printf("results identical: %s   (%zu bytes written of %zu)\n", out[0] == out[1] ? "yes" : "NO", stored, out[1].size());
  return out[0] == out[1] ? 0 : 2;
}
