// Micro-benchmark 5 (round 4): does a SECOND tile per wavefront held in REGISTERS lift the 128-site-tile kernels?
// scan_kernel keeps one tile per wavefront in LDS (LDS-DMA); while the wavefront processes it (tier 1, compaction, the column
// walk: dependent LDS round trips) that buffer neither loads nor is free, and LDS (160 KB / CU) caps the bytes in flight.
// The VGPR file is 512 KB / CU and the 128-site kernels use a quarter of it.  Variants, same tiles, same dummy processing
// (a tier-1-like pass over the tile + `work` dependent LDS round trips):
//   dma     as scan_kernel: LDS-DMA into the one LDS tile, next tile requested when the current one has been processed
//   regs    the next tile travels global -> VGPRs (global_load_dwordx4 nt) WHILE the current one is processed out of LDS,
//           then VGPRs -> LDS (ds_write_b128): one tile per wavefront in flight all the time, LDS per wavefront unchanged
// Build: hipcc --offload-arch=gfx950 -O3 -o reg_prefetch reg_prefetch.hip ; run: ./reg_prefetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// WK: what the dummy processing is made of after the tier-1-like pass: 0 = `work` dependent LDS round trips; 1 = scan_kernel's
// tier 2 on `work` fake variable sites (8 lanes per site walk the column byte by byte, DPP sums, an LDS histogram atomic);
// 2 = `work` x 8 dependent VALU instructions (no LDS)
template <int TW, int NI, int MODE, int WK>
__global__ void __launch_bounds__(64) k(const uint8_t* __restrict__ bases, uint32_t n_rows, uint32_t n_sites, uint32_t stride, uint32_t tpc,
                                        uint32_t work, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t LPR = TW / 16u, RPI = 64u / LPR, GROUPS = 256u / TW;
  const uint32_t lane = threadIdx.x;
  const uint8_t* mat = bases + (uint64_t)blockIdx.x * n_rows * stride;
  const uint32_t n_tiles = (n_sites + TW - 1u) / TW;
  const uint32_t t0 = blockIdx.y * tpc;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = min(t0 + tpc, n_tiles);
  const uint32_t c = lane % LPR, rr = lane / LPR;
  uint32_t acc = 0;
  v4u regs[NI];
  auto src_of = [&](uint32_t t, uint32_t g) { return mat + (uint64_t)(g * RPI + rr) * stride + min(t * TW + 16u * c, stride - 16u); };
  auto stage_dma = [&](uint32_t t) {
#pragma unroll
    for (uint32_t g = 0; g < NI; ++g)
      if (g * RPI + rr < n_rows) __builtin_amdgcn_global_load_lds((gptr_t)src_of(t, g), (lptr_t)(lds + g * 1024u), 16, 0, 2);
  };
  auto load_regs = [&](uint32_t t) {
#pragma unroll
    for (uint32_t g = 0; g < NI; ++g)
      if (g * RPI + rr < n_rows) regs[g] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src_of(t, g)));
  };
  auto regs_to_lds = [&]() {
#pragma unroll
    for (uint32_t g = 0; g < NI; ++g)
      if (g * RPI + rr < n_rows) *reinterpret_cast<v4u*>(lds + g * 1024u + lane * 16u) = regs[g];
  };
  if (MODE == 0) stage_dma(t0); else load_regs(t0);
  for (uint32_t t = t0; t < t1; ++t) {
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      regs_to_lds();                       // (waits for the loads of tile t)
      if (t + 1 < t1) load_regs(t + 1);    // tile t + 1 travels while tile t is processed
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t a = 0xFFFFFFFFu, o = 0;
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(lds);
    const uint32_t steps = n_rows / GROUPS;
#pragma unroll 8
    for (uint32_t r = 0; r < steps; ++r) { const uint32_t x = t32[r * 64u + lane]; a &= x; o |= x; }
    uint32_t v = (a ^ o) & 1023u;
    if (WK == 0) {
      for (uint32_t i = 0; i < work; ++i) v = t32[(v + lane) & 2047u] & 1023u;   // dependent LDS round trips (the column walk's chain)
    } else if (WK == 1) {
      uint32_t* hist = reinterpret_cast<uint32_t*>(lds + NI * 1024u);
      for (uint32_t p0 = 0; p0 < work; p0 += 8u) {
        const uint32_t slot = lane >> 3, ph = lane & 7u;
        const bool act = p0 + slot < work;
        const uint8_t* colp = lds + ((17u * (p0 + slot) + 3u + v) & (TW - 1u));
        const uint32_t x = colp[0], y = x ^ 6u;
        uint32_t c1 = 0, c2 = 0;
#pragma unroll 4
        for (uint32_t r = ph; r < n_rows; r += 8u) { const uint32_t b = colp[r * TW]; c1 += (b == x); c2 += (b == y); }
        c1 += __shfl_xor(c1, 1); c1 += __shfl_xor(c1, 2); c1 += __shfl_xor(c1, 4);
        c2 += __shfl_xor(c2, 1); c2 += __shfl_xor(c2, 2); c2 += __shfl_xor(c2, 4);
        if (act && ph == 0u) { acc += n_rows * n_rows - c1 * c1 - c2 * c2; atomicAdd(&hist[(c2 + slot) & 63u], 1u); }
      }
    } else {
      for (uint32_t i = 0; i < work * 8u; ++i) v = v * 1664525u + 1013904223u;
    }
    acc += v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 0 && t + 1 < t1) stage_dma(t + 1);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int TW, int NI, int MODE, int WK>
static void run(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t work, uint32_t* dout) {
  const uint32_t n_tiles = (L + TW - 1) / TW, chunks = (n_tiles + tpc - 1) / tpc;
  const size_t lds = (size_t)NI * 1024 + 2048;   // (+ the fake histogram)
  auto kern = k<TW, NI, MODE, WK>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)kern);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("N %3u TW %3d %-4s %s work %3u  lds/wave %6zu  vgprs %3d  %.4f ms  %5.0f GB/s  (%.3f of 8 TB/s)\n", N, TW, MODE ? "regs" : "dma", WK == 0 ? "lds-chain" : WK == 1 ? "tier2-walk" : "valu-chain", work, lds, fa.numRegs, ms,
         (double)B * N * L / ms / 1e6, (double)B * N * L / ms / 1e6 / 8000.0);
}

template <int TW, int NI>
static void both(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t* dout) {
  for (uint32_t work : {0u, 32u}) {
    run<TW, NI, 0, 0>(d, B, N, L, stride, tpc, work, dout);
    run<TW, NI, 1, 0>(d, B, N, L, stride, tpc, work, dout);
  }
  for (uint32_t work : {8u, 16u, 32u}) {
    run<TW, NI, 0, 1>(d, B, N, L, stride, tpc, work, dout);
    run<TW, NI, 1, 1>(d, B, N, L, stride, tpc, work, dout);
  }
  for (uint32_t work : {32u, 128u}) {
    run<TW, NI, 0, 2>(d, B, N, L, stride, tpc, work, dout);
    run<TW, NI, 1, 2>(d, B, N, L, stride, tpc, work, dout);
  }
}

int main() {
  const uint32_t L = 100000, stride = 100096;
  uint8_t* d; uint32_t* dout;
  const size_t bytes = (size_t)128 * 200 * stride;    // 2.56 GB
  hipMalloc(&d, bytes); hipMalloc(&dout, 64);
  hipMemset(d, 65, bytes);
  // discarded first measurement (a box's first seconds run at higher clocks)
  run<128, 13, 0, 0>(d, 256, 100, L, stride, 2, 8, dout);
  printf("---- N = 100, 128-site tiles (12.8 KB per tile: 12 wavefronts / CU)\n");
  both<128, 13>(d, 256, 100, L, stride, 2, dout);
  printf("---- N = 200, 128-site tiles (25.6 KB per tile: 6 wavefronts / CU)\n");
  both<128, 25>(d, 128, 200, L, stride, 2, dout);
  printf("---- N = 50, 256-site tiles (12.8 KB per tile: 12 wavefronts / CU)\n");
  both<256, 13>(d, 512, 50, L, stride, 2, dout);
  printf("---- N = 100 again (order effects)\n");
  both<128, 13>(d, 256, 100, L, stride, 2, dout);
  return 0;
}
