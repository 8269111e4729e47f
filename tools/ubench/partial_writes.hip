// Micro-benchmark (runs ON THE GPU BOX): cost of scattered byte stores into per-env 200-byte slabs vs streaming the
// whole slab, at 65 536 envs.  Informs the incremental-observation design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_bytes(uint8_t* obs, const uint32_t* rec, int B, int nbytes, int step) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  const uint32_t r = rec[e * 8];
  const int x = (r + step) % 7, y = ((r >> 8) + step) % 17;  // a 4-cell piece inside the 10x20 slab: rows y..y+3
  uint8_t* s = obs + (size_t)e * 200;
  for (int i = 0; i < nbytes; ++i) s[(y + (i & 3)) * 10 + x + (i >> 2)] = (uint8_t)(step + i);
}
__global__ void k_full(uint8_t* obs, const uint8_t* board, int B) {  // coalesced 16-byte stream copy of the slabs
  const size_t n = (size_t)B * 200 / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(obs)[i] = reinterpret_cast<const uint4*>(board)[i];
}
int main() {
  const int B = 65536;
  uint8_t *obs, *board; uint32_t* rec;
  CK(hipMalloc(&obs, (size_t)B * 200)); CK(hipMalloc(&board, (size_t)B * 200)); CK(hipMalloc(&rec, (size_t)B * 32));
  std::vector<uint32_t> h(B * 8); for (int i = 0; i < B * 8; ++i) h[i] = i * 2654435761u;
  CK(hipMemcpy(rec, h.data(), B * 32, hipMemcpyHostToDevice)); CK(hipMemset(obs, 0, (size_t)B * 200)); CK(hipMemset(board, 1, (size_t)B * 200));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int K = 200;
  for (int nb : {0, 4, 8, 12}) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_bytes, dim3(B / 64), dim3(64), 0, 0, obs, rec, B, nb, i);
    CK(hipEventRecord(a));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_bytes, dim3(B / 64), dim3(64), 0, 0, obs, rec, B, nb, i);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("scattered byte stores: %2d bytes/env  %.2f us/launch\n", nb, ms / K * 1e3);
  }
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_full, dim3(1024), dim3(256), 0, 0, obs, board, B);
  CK(hipEventRecord(a));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_full, dim3(1024), dim3(256), 0, 0, obs, board, B);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("full slab stream copy (13 MB in, 13 MB out): %.2f us/launch\n", ms / K * 1e3);
  return 0;
}
