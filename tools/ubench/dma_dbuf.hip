// Micro-benchmark 2: single-buffered vs double-buffered (prefetch next tile during compute)
// LDS-DMA tiles with an adjustable amount of dummy per-tile VALU work, to size what
// prefetching could buy scan_kernel.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ inline void stage(const uint8_t* mat, uint8_t* buf, uint32_t t, uint32_t n_rows, uint32_t stride, uint32_t lane) {
  const uint32_t c = lane & 15u, rr = lane >> 4;
  const uint32_t col = min(t * 256u + 16u * c, stride - 16u);
  const uint32_t n_inst = (n_rows + 3u) / 4u;
  for (uint32_t g = 0; g < n_inst; ++g) {
    const uint32_t r = g * 4u + rr;
    if (r < n_rows)
      __builtin_amdgcn_global_load_lds((gptr_t)(mat + (uint64_t)r * stride + col), (lptr_t)(buf + g * 1024u), 16, 0, 2);
  }
}

template <bool DBUF>
__global__ void __launch_bounds__(64) k(const uint8_t* __restrict__ bases, uint32_t n_rows, uint32_t n_sites,
                                        uint32_t stride, uint32_t tpc, uint32_t work, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint8_t* mat = bases + (uint64_t)blockIdx.x * n_rows * stride;
  const uint32_t n_tiles = (n_sites + 255u) / 256u;
  const uint32_t t0 = blockIdx.y * tpc;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = min(t0 + tpc, n_tiles);
  const uint32_t tile_bytes = n_rows * 256u;
  uint32_t acc = 0, cur = 0;
  if (DBUF) stage(mat, lds, t0, n_rows, stride, lane);
  for (uint32_t t = t0; t < t1; ++t) {
    uint8_t* buf = lds + (DBUF ? cur * tile_bytes : 0u);
    if (!DBUF) stage(mat, buf, t, n_rows, stride, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (DBUF && t + 1 < t1) stage(mat, lds + (cur ^ 1u) * tile_bytes, t + 1, n_rows, stride, lane);
    uint32_t a = 0xFFFFFFFFu, o = 0;
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(buf);
    for (uint32_t r = 0; r < n_rows; ++r) { const uint32_t x = t32[r * 64u + lane]; a &= x; o |= x; }
    uint32_t v = a ^ o;
    for (uint32_t i = 0; i < work; ++i) v = v * 1664525u + 1013904223u;   // dependent dummy VALU chain (2 ops/iter)
    acc += v;
    __syncthreads();
    cur ^= 1u;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <bool DBUF>
static void run(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t work, uint32_t* dout) {
  const uint32_t n_tiles = (L + 255) / 256, chunks = (n_tiles + tpc - 1) / tpc;
  const size_t lds = (size_t)N * 256 * (DBUF ? 2 : 1);
  hipFuncSetAttribute((const void*)k<DBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<DBUF>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<DBUF>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%s tpc %2u work %4u lds/wave %6zu: %.4f ms  %.0f GB/s\n", DBUF ? "dbuf  " : "single", tpc, work, lds, ms, (double)B * N * L / ms / 1e6);
}

int main(int argc, char** argv) {
  const uint32_t B = 8192, N = argc > 1 ? atoi(argv[1]) : 50, L = 10000, stride = 10112;
  uint8_t* d; uint32_t* dout;
  hipMalloc(&d, (size_t)B * N * stride); hipMalloc(&dout, 64);
  hipMemset(d, 65, (size_t)B * N * stride);
  for (uint32_t work : {0u, 100u, 200u, 400u, 800u}) {
    run<false>(d, B, N, L, stride, 10, work, dout);
    run<true>(d, B, N, L, stride, 10, work, dout);
    run<true>(d, B, N, L, stride, 20, work, dout);
  }
  return 0;
}
