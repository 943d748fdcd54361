// pfa_scan_wg.hip — scan_wg_kernel: non-coding units of about 65 to 208 rows, W = 2 or 4
// wavefronts per workgroup sharing one 256-site tile.
//
// With one wavefront per tile, LDS per wave grows with N: at N = 200 a CU holds 6 wavefronts of
// 128-site tiles (or 3 of 256-site tiles), each with a tier-1 walk of 100-200 steps and column walks
// of 200 rows — too few, too long-running wavefronts to keep the memory system busy.  Here the W
// wavefronts of a workgroup split the ROWS of a 256-site tile (full 256-byte row segments, like
// the 50-row case): every wave stages and walks its own N / W rows, the per-wave AND / OR words meet
// in LDS, and the variable sites' columns are counted by 4 W lanes per site, the sites dealt over
// the waves.  Per wave this is the work of a 50-row tile; a CU again holds 12 wavefronts.
// Three workgroup barriers per tile; partial sums leave through atomics as everywhere else.
// Reference: getvarsites PolyFastA.py:252-261, nucleotide_diversity :485-492, getsfs :274-282,
// population subsets :122-134.
#include "pfa_kernels.h"
#include "pfa_launch.h"

namespace pfa {

template <int W, bool MAPPED>
__global__ void __launch_bounds__(64 * W) scan_wg_kernel(const ScanArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t TW = 256u, LPS = 4u * W, SPI = 64u / LPS;   // lanes per site, sites per wave and pass
  constexpr uint32_t lps_log = W == 2 ? 3u : 4u;
  static_assert(W == 2 || W == 4, "wavefronts per workgroup");

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t unit = A.unit_ids ? A.unit_ids[blockIdx.x] : blockIdx.x;
  const pfa_desc D = A.descs[unit];
  const uint32_t N = D.n_rows, L = D.n_sites;
  if (N == 0u || L == 0u) return;
  const uint32_t n_tiles = (L + TW - 1u) / TW;
  const uint32_t t0 = blockIdx.y * A.tiles_per_chunk;
  if (t0 >= n_tiles) return;                       // the same for every wave of the workgroup
  const uint32_t t1 = umin(t0 + A.tiles_per_chunk, n_tiles);

  const WgLayout lo = wg_layout(W, A.n_max, MAPPED);
  uint8_t* const tile8 = lds + lo.tile;
  uint2* const xch = reinterpret_cast<uint2*>(lds + lo.xch);   // per wave and lane: AND word, OR word
  uint8_t* const list = lds + lo.list;                         // the tile's variable sites (tile-relative)
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + lo.hist);
  uint32_t* const rowmap = reinterpret_cast<uint32_t*>(lds + lo.rowmap);

  // rows of this wave: [r_lo, r_hi), a multiple of 4 per wave (one DMA instruction moves 4 rows)
  const uint32_t per = ((N + W - 1u) / W + 3u) & ~3u;
  const uint32_t r_lo = umin(wave * per, N), r_hi = umin(r_lo + per, N), n_mine = r_hi - r_lo;

  const uint32_t nbins = N / 2u;
  const bool mapped = MAPPED && D.row_idx_off != PFA_NO_ROW_IDX;
  if (mapped)
    for (uint32_t i = threadIdx.x; i < N; i += 64u * W) rowmap[i] = A.row_idx[D.row_idx_off + i] * (D.row_stride >> 4);
  for (uint32_t i = threadIdx.x; i < nbins; i += 64u * W) hist[i] = 0u;
  __syncthreads();

  uint32_t acc_S = 0u, acc_bad = 0u;
  unsigned long long acc_H = 0ull;
  const uint8_t* const mat = A.bases + D.base_off;
  const uint32_t NN = N * N;

  // ---------------- stage this wave's rows of a tile (asynchronous)
  auto stage = [&](uint32_t tile) {
    const uint32_t c = lane % 16u, rr = lane / 16u;
    const uint32_t col = umin(tile * TW + 16u * c, D.row_stride - 16u);
    const uint32_t n_full = n_mine / 4u, rem = n_mine % 4u;
    uint8_t* const dst = tile8 + r_lo * TW;
    if (!mapped) {
      const uint8_t* src = mat + (uint64_t)(r_lo + rr) * D.row_stride + col;
      const uint64_t step = 4ull * D.row_stride;
      uint32_t g = 0;
      for (; g < n_full; ++g, src += step)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + g * 1024u), 16, 0, 2);
      if (rr < rem)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + g * 1024u), 16, 0, 2);
    } else {
      const uint8_t* const base = mat + col;
      const uint32_t n_inst = n_full + (rem ? 1u : 0u);
      for (uint32_t g0 = 0; g0 < n_inst; g0 += 8u) {     // offsets out of LDS eight instructions at a time
        uint32_t off16[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
          const uint32_t r = r_lo + (g0 + k) * 4u + rr;
          off16[k] = rowmap[r < N ? r : 0u];
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
          const uint32_t g = g0 + k;
          if (g < n_full || (g == n_full && rr < rem))
            __builtin_amdgcn_global_load_lds((gptr_t)(base + ((uint64_t)off16[k] << 4)), (lptr_t)(dst + g * 1024u), 16, 0, 2);
        }
      }
    }
  };
  stage(t0);

  for (uint32_t tile = t0; tile < t1; ++tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t lim = umin(TW, L - site0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows have landed
    wave_lds_sync();

    // ---------------- tier 1 on this wave's rows
    uint32_t pa = 0xFFFFFFFFu, po = 0u;
    {
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile8) + r_lo * 64u + lane;
#pragma unroll 8
      for (uint32_t r = 0; r < n_mine; ++r) {
        const uint32_t x = t32[r * 64u];
        pa &= x;
        po |= x;
      }
    }
    xch[wave * 64u + lane] = make_uint2(pa, po);
    __syncthreads();   // every wave's words are in LDS — and with them every wave's rows of the tile
    uint32_t va = 0xFFFFFFFFu, vo = 0u;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)W; ++w) {
      const uint2 v = xch[w * 64u + lane];
      va &= v.x;
      vo |= v.y;
    }
    const uint32_t vd = va ^ vo;

    // ---------------- compact the variable sites (every wave computes the positions, wave 0 writes the list)
    uint32_t P = 0u;
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t off = 4u * lane + b;
      const bool pb = ((vd >> (8u * b)) & 0xFFu) != 0u && off < lim;
      const uint64_t m = __ballot(pb);
      if (pb && wave == 0u) list[P + mbcnt64(m)] = uint8_t(off);
      P += (uint32_t)__popcll(m);
    }
    P = __builtin_amdgcn_readfirstlane(P);
    acc_S += (threadIdx.x == 0u) ? P : 0u;
    __syncthreads();   // the list is visible

    // ---------------- tier 2: LPS lanes per site walk its column; the sites are dealt over the waves
    const uint32_t slot = lane / LPS, ph = lane % LPS;
    for (uint32_t p0 = wave * SPI; p0 < P; p0 += SPI * W) {
      const uint32_t idx = p0 + slot;
      const bool act = idx < P;
      const uint32_t s = list[act ? idx : 0u];
      // the site's XOR byte sits in the tier-1 words of the lane that owns it (every lane is active here)
      const uint32_t dword = (uint32_t)__shfl((int)vd, (int)(s >> 2));
      const uint8_t* colp = tile8 + s;
      const uint32_t x = colp[0], y = x ^ ((dword >> (8u * (s & 3u))) & 0xFFu);
      uint32_t c1 = 0, c2 = 0;
#pragma unroll 4
      for (uint32_t r = ph; r < N; r += LPS) {
        const uint32_t v = colp[r * TW];
        c1 += (v == x);
        c2 += (v == y);
      }
      c1 = group_sum(c1, lps_log);
      c2 = group_sum(c2, lps_log);
      const bool xc = base_clean(x), yc = base_clean(y);
      uint32_t sumsq = c1 * c1 + c2 * c2;
      uint32_t second = (xc && yc) ? umin(c1, c2) : 0u;   // second-largest A/C/G/T count (PolyFastA.py:279-281)
      const bool multi = act && c1 + c2 != N;             // the same in all lanes of the site's group
      if (__any(multi)) {
        if (multi) {
          uint32_t cA = 0, cC = 0, cG = 0, cT = 0, cGap = 0, cNn = 0;
#pragma unroll 2
          for (uint32_t r = ph; r < N; r += LPS) {
            const uint32_t v = colp[r * TW];
            cA += (v == 'A');
            cC += (v == 'C');
            cG += (v == 'G');
            cT += (v == 'T');
            cGap += (v == '-');
            cNn += (v == 'N');
          }
          cA = group_sum(cA, lps_log);
          cC = group_sum(cC, lps_log);
          cG = group_sum(cG, lps_log);
          cT = group_sum(cT, lps_log);
          cGap = group_sum(cGap, lps_log);
          cNn = group_sum(cNn, lps_log);
          sumsq = cA * cA + cC * cC + cG * cG + cT * cT + cGap * cGap + cNn * cNn;
          const uint32_t n_other = N - (cA + cC + cG + cT + cGap + cNn);
          if (n_other != 0u) {
            // exact path for any other byte value: sum over such rows i of #{j : x_j == x_i}
            uint32_t extra = 0u;
            for (uint32_t i = ph; i < N; i += LPS) {
              const uint32_t xi = colp[i * TW];
              const bool six = xi == 'A' || xi == 'C' || xi == 'G' || xi == 'T' || xi == '-' || xi == 'N';
              if (!six) {
                uint32_t cnt = 0u;
                for (uint32_t j = 0; j < N; ++j) cnt += (colp[j * TW] == xi);
                extra += cnt;
              }
            }
            sumsq += group_sum(extra, lps_log);
          }
          const uint32_t hi1 = umax(cA, cC), lo1 = umin(cA, cC), hi2 = umax(cG, cT), lo2 = umin(cG, cT);
          second = umax(umin(hi1, hi2), umax(lo1, lo2));
        }
      }
      if (act && ph == 0u) {
        acc_H += NN - sumsq;
        if (second == 0u) acc_bad += 1u;
        else atomicAdd(&hist[second - 1u], 1u);
      }
    }
    __syncthreads();   // every wave has read the tile and the list for the last time
    if (tile + 1u < t1) stage(tile + 1u);
  }

  // ---------------- flush the chunk's partial sums (one set of atomics per wave)
  acc_S = wave_sum(acc_S);
  acc_bad = wave_sum(acc_bad);
  acc_H = wave_sum(acc_H);
  const kernarg_ptr_t K = kernargs_again();
  const uint32_t rep_off = (blockIdx.y & K->rep_mask) * K->rep_stride;   // this chunk's output replica
  if (lane == 0u) {
    pfa_result* const R = replica(K->results, rep_off) + unit;
    if (acc_S | acc_bad) atomicAdd(reinterpret_cast<unsigned long long*>(&R->seg_sites), (unsigned long long)acc_S | (unsigned long long)acc_bad << 32);
    if (acc_H) atomicAdd(reinterpret_cast<unsigned long long*>(&R->H), acc_H);
  }
  uint32_t* const sfs = replica(K->sfs, rep_off) + D.sfs_off;   // (the last barrier of the loop made every LDS add visible)
  for (uint32_t i = threadIdx.x; i < nbins; i += 64u * W) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&sfs[i], v);
  }
}

template <int W, bool MAPPED>
static int launch_one(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  auto kern = scan_wg_kernel<W, MAPPED>;
  static uint32_t lds_limit_set = 0;  // per instantiation; one process drives one device
  if (c.lds_bytes > lds_limit_set) {
    PFA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)c.lds_bytes));
    lds_limit_set = c.lds_bytes;
  }
  hipLaunchKernelGGL(kern, dim3(c.n_units, c.n_chunks, 1), dim3(64 * W, 1, 1), c.lds_bytes, st, a);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}

int launch_wg(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  if (c.cds || c.bitmaps || (c.waves != 2u && c.waves != 4u)) {
    set_error("internal: the multi-wave kernel serves non-coding launches without bitmaps, 2 or 4 waves");
    return PFA_ERR_ARG;
  }
  if (c.waves == 2u) return c.mapped ? launch_one<2, true>(c, a, st) : launch_one<2, false>(c, a, st);
  return c.mapped ? launch_one<4, true>(c, a, st) : launch_one<4, false>(c, a, st);
}

}  // namespace pfa
