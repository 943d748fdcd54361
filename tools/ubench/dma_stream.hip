// Micro-benchmark: how fast can the scan's access pattern be streamed at all?
// Each single-wave workgroup stages N rows x SEG bytes tiles into LDS with
// global_load_lds_dwordx4 (nt), reads them back once (AND/OR like tier 1) and moves on.
// Upper bound for scan_kernel at a given tile geometry.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int SEG, int AUX>
__global__ void __launch_bounds__(64) k(const uint8_t* __restrict__ bases, uint32_t n_rows, uint32_t n_sites,
                                        uint32_t stride, uint32_t tiles_per_chunk, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t LPR = SEG / 16, RPI = 64 / LPR;
  const uint32_t lane = threadIdx.x;
  const uint8_t* mat = bases + (uint64_t)blockIdx.x * n_rows * stride;
  const uint32_t n_tiles = (n_sites + SEG - 1) / SEG;
  const uint32_t t0 = blockIdx.y * tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = min(t0 + tiles_per_chunk, n_tiles);
  const uint32_t n_inst = (n_rows + RPI - 1) / RPI;
  uint32_t acc = 0;
  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t c = lane % LPR, rr = lane / LPR;
    const uint32_t col = min(t * SEG + 16u * c, stride - 16u);
    for (uint32_t g = 0; g < n_inst; ++g) {
      const uint32_t r = g * RPI + rr;
      if (r < n_rows)
        __builtin_amdgcn_global_load_lds((gptr_t)(mat + (uint64_t)r * stride + col), (lptr_t)(lds + g * 1024u), 16, 0, AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t a = 0xFFFFFFFFu, o = 0;
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(lds);
    for (uint32_t i = lane; i < n_rows * (SEG / 4); i += 64u) { const uint32_t x = t32[i]; a &= x; o |= x; }
    acc += a ^ o;
    __syncthreads();
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int SEG, int AUX>
static void run(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t* dout) {
  const uint32_t n_tiles = (L + SEG - 1) / SEG, chunks = (n_tiles + tpc - 1) / tpc;
  const size_t lds = (size_t)N * SEG;
  hipFuncSetAttribute((const void*)k<SEG, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SEG, AUX>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, dout);
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SEG, AUX>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("SEG %4d aux %d tpc %2u lds/wave %6zu: %.4f ms  %.0f GB/s (algorithmic N*L)\n", SEG, AUX, tpc, lds, ms, (double)B * N * L / ms / 1e6);
}

int main(int argc, char** argv) {
  const uint32_t B = 8192, N = argc > 1 ? atoi(argv[1]) : 50, L = 10000;
  const uint32_t stride = argc > 2 ? atoi(argv[2]) : 10112;
  uint8_t* d; uint32_t* dout;
  hipMalloc(&d, (size_t)B * N * stride); hipMalloc(&dout, 64);
  hipMemset(d, 65, (size_t)B * N * stride);
  for (uint32_t tpc : {4u, 10u, 40u}) {
    run<256, 2>(d, B, N, L, stride, tpc, dout);
    run<256, 0>(d, B, N, L, stride, tpc, dout);
  }
  for (uint32_t tpc : {2u, 5u, 20u}) run<512, 2>(d, B, N, L, stride, tpc, dout);
  for (uint32_t tpc : {1u, 3u, 10u}) run<1024, 2>(d, B, N, L, stride, tpc, dout);
  for (uint32_t tpc : {8u, 20u}) run<128, 2>(d, B, N, L, stride, tpc, dout);
  return 0;
}
