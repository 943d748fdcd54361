// Micro-benchmark: the ceiling of scan_tall_kernel's access pattern.  A single-wave workgroup streams the rows of a SEG-byte-wide
// column of sites in chunks of R rows (LDS-DMA, nt), ANDs / ORs every chunk once (tier 1 only) and moves on: what the pattern
// allows when nothing but tier 1 is done.  Arguments: N L units [pad_bytes]; LDS per workgroup = R * SEG + pad (waves per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o tall_stream tall_stream.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// T2: what follows tier 1 in a chunk, as LDS traffic only.  0 nothing; 1 the shipped pattern: two passes of twelve byte reads per
// lane (four lanes per site, rows ph + 4 j: every row of a site sits in the same bank) and two codon slots of three byte reads with
// one row per lane (64 lanes in one bank); 2 the same number of reads with the addresses skewed as a layout padded by 16 bytes per
// 1 KiB DMA block would have them.
template <int SEG, int T2>
__global__ void __launch_bounds__(64) k(const uint8_t* __restrict__ bases, uint32_t N, uint32_t L, uint32_t stride, uint32_t R,
                                        uint32_t tiles_per_chunk, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t LPR = SEG / 16, RPI = 64 / LPR;
  const uint32_t lane = threadIdx.x, c = lane % LPR, rr = lane / LPR;
  const uint8_t* mat = bases + (uint64_t)blockIdx.x * N * stride;
  const uint32_t n_tiles = (L + SEG - 1) / SEG;
  const uint32_t t0 = blockIdx.y * tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = min(t0 + tiles_per_chunk, n_tiles);
  uint32_t acc = 0;
  auto stage = [&](uint32_t t, uint32_t r0) {
    const uint32_t Rc = min(R, N - r0);
    const uint32_t col = min(t * SEG + 16u * c, stride - 16u);
    const uint8_t* src = mat + (uint64_t)(r0 + rr) * stride + col;
    const uint32_t n_inst = (Rc + RPI - 1) / RPI;
    for (uint32_t g = 0; g < n_inst; ++g) {
      if (g * RPI + rr < Rc) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + g * 1024u), 16, 0, 2);
      src += (uint64_t)stride * RPI;
    }
  };
  stage(t0, 0);
  for (uint32_t t = t0; t < t1; ++t) {
    uint32_t a = 0xFFFFFFFFu, o = 0;
    for (uint32_t r0 = 0; r0 < N; r0 += R) {
      const uint32_t Rc = min(R, N - r0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(lds);
      const uint32_t steps = Rc * (SEG / 4) / 64;
#pragma unroll 8
      for (uint32_t i = 0; i < steps; ++i) { const uint32_t x = t32[i * 64 + lane]; a &= x; o |= x; }
      if constexpr (T2 >= 10) {   // T2 = 10 + v: v x 120 dependent VALU operations per chunk (10 per KiB each), no LDS traffic
#pragma unroll 1
        for (uint32_t i = 0; i < (uint32_t)(T2 - 10) * 30u; ++i) {
          a = a * 0x9E3779B1u + o;
          o = (o ^ (a >> 7)) + lane;
          a = __builtin_amdgcn_perm(a, o, 0x02010003u);
          o += a & 0xFFu;
        }
      } else if constexpr (T2 != 0) {
        const uint32_t ph = lane & 3u, slot = lane >> 2;
        uint32_t sum = 0;
        for (uint32_t pass = 0; pass < 2u; ++pass) {
          const uint32_t site = ((slot + 16u * pass) * 37u + t * 11u) & 255u;
#pragma unroll
          for (uint32_t j = 0; j < 12u; ++j) {
            const uint32_t row = T2 == 1 ? ph + 4u * j : 12u * ph + j;
            const uint32_t off = T2 == 1 ? row * 256u : (row >> 2) * 1040u + (row & 3u) * 256u;
            sum += lds[(off + site) % (R * SEG)];
          }
        }
        for (uint32_t cs = 0; cs < 2u; ++cs) {
          const uint32_t first = (cs * 101u + t * 7u) & 0xFCu;
          const uint32_t row = lane < Rc ? lane : 0u;
          const uint32_t off = T2 == 1 ? row * 256u : ((row >> 2) * 1040u + (row & 3u) * 256u) % (R * SEG);
          sum += lds[off + first] + lds[off + first + 1u] + lds[off + first + 2u];
        }
        a ^= sum;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (r0 + R < N) stage(t, r0 + R);
      else if (t + 1 < t1) stage(t + 1, 0);
    }
    acc += a ^ o;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// alignment-like contents: every column one of A C G T, a row differs from it in 1 % of the sites ('r'), or random letters ('R')
__global__ void fill(uint8_t* d, size_t n, uint32_t stride, int mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t col = (uint32_t)(i % stride);
    uint64_t h = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31; h *= 0x94D049BB133111EBull; h ^= h >> 29;
    const uint32_t cons = (col * 2654435761u) >> 30;
    const uint32_t c = mode == 2 ? (uint32_t)(h & 3u) : ((h >> 8) % 100u == 0u ? (uint32_t)(h & 3u) : cons);
    d[i] = "ACGT"[c];
  }
}

template <int SEG, int T2 = 0>
static void run(const uint8_t* d, uint32_t B, uint32_t N, uint32_t L, uint32_t stride, uint32_t R, uint32_t tpc, uint32_t pad, uint32_t* dout) {
  const uint32_t n_tiles = (L + SEG - 1) / SEG, chunks = (n_tiles + tpc - 1) / tpc;
  const size_t lds = (size_t)R * SEG + pad;
  hipFuncSetAttribute((const void*)k<SEG, T2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<SEG, T2>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, R, tpc, dout);
  hipEventRecord(e0);
  const int reps = 8;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SEG, T2>), dim3(B, chunks), dim3(64), lds, 0, d, N, L, stride, R, tpc, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const uint32_t gran = (uint32_t)((lds + 511) / 512);
  printf("T2 %d N %5u SEG %4d R %3u tpc %2u lds %6zu (%u waves/CU): %.3f ms  %.0f GB/s = %.3f of 8 TB/s\n", T2, N, SEG, R, tpc, lds, 320u / gran, ms,
         (double)B * N * L / ms / 1e6, (double)B * N * L / ms / 1e6 / 8000.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t N = argc > 1 ? atoi(argv[1]) : 5000, L = argc > 2 ? atoi(argv[2]) : 9000, B = argc > 3 ? atoi(argv[3]) : 1024;
  const uint32_t stride = (L + 127u) / 128u * 128u;
  uint8_t* d; uint32_t* dout;
  if (hipMalloc(&d, (size_t)B * N * stride) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&dout, 64);
  hipMemset(d, 65, (size_t)B * N * stride);
  if (argc > 4 && (argv[4][0] == 'r' || argv[4][0] == 'R')) {   // contents: constant against alignment-like against random letters
    for (int mode = 0; mode < 3; ++mode) {
      if (mode) { hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, d, (size_t)B * N * stride, stride, mode); hipDeviceSynchronize(); }
      printf("contents: %s\n", mode == 0 ? "every byte 'A'" : mode == 1 ? "alignment-like (1 %% of the bytes differ from the column's letter)" : "random letters");
      for (int rep = 0; rep < 2; ++rep) {
        run<256, 0>(d, B, N, L, stride, 48, 1, 4096, dout);
        run<256, 13>(d, B, N, L, stride, 48, 1, 4096, dout);
      }
    }
    return 0;
  }
  if (argc > 4 && argv[4][0] == 'c') {   // what the bank conflicts of tier 2 and of the codon slots cost
    for (int rep = 0; rep < 2; ++rep) {
      run<256, 0>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 1>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 2>(d, B, N, L, stride, 48, 1, 4096, dout);
    }
    return 0;
  }
  if (argc > 4 && argv[4][0] == 'v') {   // what instructions between the wait and the next request cost
    for (int rep = 0; rep < 2; ++rep) {
      run<256, 0>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 11>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 12>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 13>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 14>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 16>(d, B, N, L, stride, 48, 1, 4096, dout);
      run<256, 13>(d, B, N, L, stride, 48, 1, 0, dout);
      run<256, 13>(d, B, N, L, stride, 48, 1, 14848, dout);
    }
    return 0;
  }
  if (argc > 4) {   // the knee: waves per CU from 13 down to 3
    for (uint32_t pad : {0u, 5120u, 10752u, 14848u, 20480u, 28672u, 41984u}) run<256>(d, B, N, L, stride, 48, 1, pad, dout);
    return 0;
  }
  for (uint32_t pad : {0u, 2048u, 4096u, 5120u}) run<256>(d, B, N, L, stride, 48, 1, pad, dout);
  run<256>(d, B, N, L, stride, 48, 2, 4096, dout);
  run<256>(d, B, N, L, stride, 48, 4, 4096, dout);
  run<256>(d, B, N, L, stride, 32, 1, 4096, dout);
  run<256>(d, B, N, L, stride, 64, 1, 4096, dout);
  run<512>(d, B, N, L, stride, 24, 1, 4096, dout);
  run<512>(d, B, N, L, stride, 32, 1, 4096, dout);
  run<1024>(d, B, N, L, stride, 12, 1, 4096, dout);
  run<128>(d, B, N, L, stride, 96, 1, 4096, dout);
  return 0;
}
