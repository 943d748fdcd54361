// Micro-benchmark: the chip's plain read ceiling.  Every thread streams uint4 loads (16 B per
// lane, fully coalesced, contiguous per wave) over a 16 GB buffer and XORs them together:
// no LDS, no tiling, nothing else.  The number the scan kernels' GB/s should be read against.
// Build: hipcc --offload-arch=gfx950 -O3 read_stream.hip -o read_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ src, size_t n16, uint32_t* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  for (; i < n16; i += stride) { const u32x4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

template <bool NT, int UNROLL>
static void run(const u32x4* d, size_t bytes, int blocks, uint32_t* dout) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, bytes / 16, dout);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, bytes / 16, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("read %s unroll %d blocks %6d: %.3f ms  %.0f GB/s\n", NT ? "nt     " : "default", UNROLL, blocks, ms, (double)bytes / ms / 1e6);
}

int main() {
  const size_t bytes = (size_t)16 << 30;
  u32x4* d; uint32_t* dout;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&dout, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 65, bytes);
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    run<true, 4>(d, bytes, blocks, dout);
    run<true, 8>(d, bytes, blocks, dout);
    run<false, 8>(d, bytes, blocks, dout);
  }
  return 0;
}
