// salu_issue.hip — what scalar instructions and predication cost a wave64 on gfx950 (s_memtime around the loop, one wave per
// SIMD): (a) VALU only, (b) SALU only, (c) one SALU between every two VALU, (d) the exec-mask idiom of an `if (lane cond)`
// region — s_and_saveexec / s_cbranch_execz / body / s_or exec — against the same body as v_cndmask selects.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/salu_issue tools/ubench/salu_issue.hip && /tmp/salu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, unsigned seed) {
  unsigned x = seed + threadIdx.x, y = seed * 3u + 1u;
  unsigned s = seed, t = seed * 5u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) { asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0" : "+v"(x), "+v"(y)); }
      if (MODE == 1) { asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0" : "+s"(s), "+s"(t) : : "scc"); }
      if (MODE == 2) { asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 %2, %2, %3\n\tv_xor_b32 %1, %1, %0\n\ts_xor_b32 %3, %3, %2" : "+v"(x), "+v"(y), "+s"(s), "+s"(t) : : "scc"); }
      if (MODE == 3) {  // predicated region: 2 VALU under an exec mask
        asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n1:\n\ts_or_b64 exec, exec, s[20:21]"
                     : "+v"(x), "+v"(y) : : "vcc", "scc", "s20", "s21");
      }
      if (MODE == 5) {  // the region without the skip branch
        asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n\ts_or_b64 exec, exec, s[20:21]"
                     : "+v"(x), "+v"(y) : : "vcc", "scc", "s20", "s21");
      }
      if (MODE == 6) {  // 2 VALU + a scalar compare and a branch that is never taken
        asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n\ts_cmp_eq_u32 %2, %2\n\ts_cbranch_scc0 1f\n1:" : "+v"(x), "+v"(y), "+s"(s) : : "scc");
      }
      if (MODE == 7) {  // 2 VALU + an unconditional branch to the next instruction
        asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n\ts_branch 1f\n1:" : "+v"(x), "+v"(y));
      }
      if (MODE == 8) {  // v_cmp -> vcc -> s_cbranch_vccz (never taken: x > y is forced by the seed) + 2 VALU
        asm volatile("v_cmp_ne_u32 vcc, %0, %0\n\ts_cbranch_vccnz 1f\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n1:" : "+v"(x), "+v"(y) : : "vcc");
      }
      if (MODE == 4) {  // the same with selects
        asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_add_u32 %2, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_xor_b32 %2, %1, %0\n\tv_cndmask_b32 %1, %1, %2, vcc"
                     : "+v"(x), "+v"(y), "+v"(s) : : "vcc");
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((x ^ y ^ s ^ t) == 0x12345u) out[1000000] = x;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 16 << 20);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const char* names[] = {"2 VALU", "2 SALU", "2 VALU + 2 SALU interleaved", "cmp + saveexec + branch + 2 VALU + restore (6 instr)", "cmp + 2 VALU + 2 cndmask (5 instr)",
                         "cmp + saveexec + 2 VALU + restore, no skip branch (5 instr)", "2 VALU + s_cmp + s_cbranch_scc0 not taken (4 instr)",
                         "2 VALU + s_branch to next (3 instr)", "v_cmp + s_cbranch_vccnz not taken + 2 VALU (4 instr)"};
  for (int wps : {1, 2}) for (int m = 0; m < 9; ++m) {
    const int blocks = cus * wps;
    std::vector<unsigned long long> h(blocks * 4);
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      if (m == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, 7u + rep);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("waves/SIMD %d  %-55s: %7.2f cycles per group of the pattern\n", wps, names[m], s / h.size() / 4096.0);
  }
  return 0;
}
