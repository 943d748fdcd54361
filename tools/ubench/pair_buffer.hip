// Micro-benchmark 6 (round 4): TWO wavefronts per tile buffer.  The 128-site-tile kernels at 100-250 rows are bound by LDS: a
// 200-row tile is 25.6 KB, six fit a CU, and while ONE wavefront scans a buffer (tier 1, the column walk) that buffer neither
// loads nor is free (profiles/r04_lds_pad_wavefronts_per_cu_sweep.txt: one wavefront fewer per CU costs coding N = 200 13 %).
// Would a second wavefront on the SAME buffer — a 128-thread workgroup, half of the DMA instructions, half of the tier-1 steps and
// half of the walk each, a barrier per phase — shorten the time a buffer is held enough to pay?  Same tiles, same dummy processing
// as reg_prefetch.hip (a tier-1-like pass + scan_kernel's tier 2 on `work` fake variable sites, or `work` x 8 dependent VALU
// instructions):
//   one     as scan_kernel: one wavefront per workgroup and buffer
//   pair    two wavefronts per workgroup and buffer
// Build: hipcc --offload-arch=gfx950 -O3 -o pair_buffer pair_buffer.hip ; run: ./pair_buffer
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int TW, int NI, int WAVES, int WK>
__global__ void __launch_bounds__(64 * WAVES) k(const uint8_t* __restrict__ bases, uint32_t n_rows, uint32_t n_sites, uint32_t stride, uint32_t tpc,
                                                uint32_t work, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t LPR = TW / 16u, RPI = 64u / LPR, GROUPS = 256u / TW;
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint8_t* mat = bases + (uint64_t)blockIdx.x * n_rows * stride;
  const uint32_t n_tiles = (n_sites + TW - 1u) / TW;
  const uint32_t t0 = blockIdx.y * tpc;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = min(t0 + tpc, n_tiles);
  const uint32_t c = lane % LPR, rr = lane / LPR;
  uint32_t acc = 0;
  uint32_t* const xchg = reinterpret_cast<uint32_t*>(lds + NI * 1024u);          // [2 waves][2 words][64 lanes]
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + NI * 1024u + 1024u);
  auto sync = [&]() {
    if (WAVES > 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto stage = [&](uint32_t t) {
#pragma unroll
    for (uint32_t g = 0; g < NI; ++g)
      if (g % WAVES == w && g * RPI + rr < n_rows)
        __builtin_amdgcn_global_load_lds((gptr_t)(mat + (uint64_t)(g * RPI + rr) * stride + min(t * TW + 16u * c, stride - 16u)), (lptr_t)(lds + g * 1024u), 16, 0, 2);
  };
  stage(t0);
  for (uint32_t t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sync();                                          // the whole tile has landed (both wavefronts' halves)
    uint32_t a = 0xFFFFFFFFu, o = 0;
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(lds);
    const uint32_t steps = n_rows / GROUPS;
#pragma unroll 8
    for (uint32_t r = w; r < steps; r += WAVES) { const uint32_t x = t32[r * 64u + lane]; a &= x; o |= x; }
    if (WAVES > 1) {                                 // the other wavefront's half of the rows
      xchg[(w * 2u) * 64u + lane] = a;
      xchg[(w * 2u + 1u) * 64u + lane] = o;
      __syncthreads();
      a &= xchg[((w ^ 1u) * 2u) * 64u + lane];
      o |= xchg[((w ^ 1u) * 2u + 1u) * 64u + lane];
    }
    uint32_t v = (a ^ o) & 1023u;
    if (WK == 1) {
      // sites dealt out over the wavefronts: wavefront w takes sites w, w + WAVES, ...
      for (uint32_t p0 = w * 8u; p0 < work; p0 += 8u * WAVES) {
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
      for (uint32_t i = 0; i < work * 8u / WAVES; ++i) v = v * 1664525u + 1013904223u;
    }
    acc += v;
    sync();                                          // nobody reads the tile any more
    if (t + 1 < t1) stage(t + 1);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int TW, int NI, int WAVES, int WK>
static void run(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t work, uint32_t* dout) {
  const uint32_t n_tiles = (L + TW - 1) / TW, chunks = (n_tiles + tpc - 1) / tpc;
  const size_t lds = (size_t)NI * 1024 + 1024 + 256;
  auto kern = k<TW, NI, WAVES, WK>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)kern);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(B, chunks), dim3(64 * WAVES), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(B, chunks), dim3(64 * WAVES), lds, 0, d, N, L, stride, tpc, work, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("N %3u TW %3d %-4s %s work %3u  lds/buffer %6zu  vgprs %3d  %.4f ms  %5.0f GB/s  (%.3f of 8 TB/s)\n", N, TW, WAVES == 1 ? "one" : "pair", WK == 1 ? "tier2-walk" : "valu-chain", work, lds,
         fa.numRegs, ms, (double)B * N * L / ms / 1e6, (double)B * N * L / ms / 1e6 / 8000.0);
}

template <int TW, int NI>
static void both(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t tpc, uint32_t* dout) {
  for (int round = 0; round < 2; ++round) {
    for (uint32_t work : {8u, 16u, 32u}) {
      run<TW, NI, 1, 1>(d, B, N, L, stride, tpc, work, dout);
      run<TW, NI, 2, 1>(d, B, N, L, stride, tpc, work, dout);
    }
    for (uint32_t work : {32u, 128u}) {
      run<TW, NI, 1, 2>(d, B, N, L, stride, tpc, work, dout);
      run<TW, NI, 2, 2>(d, B, N, L, stride, tpc, work, dout);
    }
  }
}

int main() {
  const uint32_t L = 100000, stride = 100096;
  uint8_t* d; uint32_t* dout;
  const size_t bytes = (size_t)512 * 200 * stride;    // 10 GB: launches of 2.5 ms
  hipMalloc(&d, bytes); hipMalloc(&dout, 64);
  hipMemset(d, 65, bytes);
  run<128, 25, 1, 1>(d, 512, 200, L, stride, 4, 8, dout);      // discarded (a box's first seconds run at higher clocks)
  printf("---- N = 200, 128-site tiles (25.6 KB per buffer: 6 buffers / CU)\n");
  both<128, 25>(d, 512, 200, L, stride, 4, dout);
  printf("---- N = 100, 128-site tiles (12.8 KB per buffer: 12 buffers / CU)\n");
  both<128, 13>(d, 1024, 100, L, stride, 4, dout);
  printf("---- N = 50, 256-site tiles (12.8 KB per buffer: 12 buffers / CU)\n");
  both<256, 13>(d, 2048, 50, L, stride, 4, dout);
  return 0;
}
