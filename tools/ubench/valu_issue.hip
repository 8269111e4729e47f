// valu_issue.hip — how a wave64 integer VALU stream is paced on gfx950: cycles per instruction for a DEPENDENT chain and for
// four INDEPENDENT chains, with 1, 2 and 4 waves per SIMD (s_memtime around 4096 instructions of lane 0's wave).
// What it decides: whether the step kernel's logic wave is bound by dependency latency (interleaving independent work would
// then be free) or by instruction issue (only fewer instructions help), and what a co-resident wave costs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/ubench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ILP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, unsigned seed, int waves_per_block) {
  unsigned x[4] = {seed + threadIdx.x, seed * 3u + threadIdx.x, seed * 5u + 1u, seed * 7u + 2u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
#pragma unroll
    for (int j = 0; j < 16 / ILP; ++j) {
#pragma unroll
      for (int c = 0; c < ILP; ++c) x[c] = __builtin_amdgcn_alignbit(x[c], x[c], 13) ^ (x[c] + 0x9E3779B9u);  // 3 VALU: alignbit, add, xor
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned r = x[0] ^ x[1] ^ x[2] ^ x[3];
  if (r == 0x12345u) out[1000000] = r;  // (keep the chains)
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 << 20);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("CUs %d, clock %.0f MHz; 256 iterations x 16 x 3 = 12288 VALU instructions per wave\n", cus, pr.clockRate / 1e3);
  for (int wps : {1, 2, 4}) {            // waves per SIMD: blocks of 256 threads (4 waves = one per SIMD), wps blocks per CU
    for (int ilp : {1, 2, 4}) {
      const int blocks = cus * wps;
      std::vector<unsigned long long> h(blocks * 4);
      for (int rep = 0; rep < 2; ++rep) {
        if (ilp == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep, 4);
        if (ilp == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep, 4);
        if (ilp == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep, 4);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0; unsigned long long mx = 0;
      for (auto v : h) { s += (double)v; if (v > mx) mx = v; }
      printf("waves/SIMD %d  independent chains %d: %.2f s_memtime ticks per VALU instruction (mean), %.2f (slowest wave)\n", wps, ilp,
             s / h.size() / 12288.0, (double)mx / 12288.0);
    }
  }
  return 0;
}
