// pfa_scan_classic.hip — scan_kernel: units whose rows all fit LDS while the CU stays full.
//
// Per tile (N rows x TW sites):
//   stage   N rows x TW sites -> LDS with LDS-DMA (global_load_lds_dwordx4, one KiB per
//           instruction, no VGPR round trip, non-temporal: the bases are read once)
//   tier 1  lane = 4 adjacent sites; walk the rows with one conflict-free ds_read per step, keep
//           running AND / OR of the bytes: a site is monomorphic iff AND == OR (2 VALU ops per
//           row per lane).  Narrow tiles (TW = 128, 64): the lane groups take every 2nd / 4th row
//           and merge their words at the end (Geo<TW>)
//   compact ballot + mbcnt turn the per-lane flags into a dense list of the tile's variable sites
//           (typically ~4 % of the sites)
//   tier 2  4 lanes per variable site walk the column out of LDS: two-allele fast path
//           (x = row 0's byte, y = x ^ XOR byte), else six counters A C G T - N, else an exact
//           generic path (sum_i #{j : x_j = x_i}); H, SFS bin (LDS histogram), optional bitmaps
//   coding  register window over the tier-1 words (CodonPass, pfa_kernels.h); 256- and 128-site tiles
// Reference: getvarsites PolyFastA.py:252-261, nucleotide_diversity :485-492, getsfs :274-282,
// getvarCDSsites :284-315, get_syn_nonsyn_cod_sites :319-434.
#include "pfa_kernels.h"
#include "pfa_launch.h"

namespace pfa {

// Tier 2, the rare columns (three or more alleles, ~1 % of the variable sites; any byte outside A C G T - N): the column of a
// tile in LDS (`col_off`: byte offset of its first row, rows `tw` bytes apart) walked again with six counters by the 2^lps_log
// lanes of the site's group (`ph` = this lane's place in it), and any other byte value by the exact generic path
// (sum over such rows i of #{j : x_j == x_i}).  Out of line (round 5): its twelve counters and three loops no longer sit in
// the register allocation of the walk every variable site takes.  Lanes with `mine` false come along (the DPP sums want their
// groups whole) and get zeros.
struct OtherAlleles {
  uint32_t sumsq, second, pres;
};
__device__ __attribute__((noinline)) OtherAlleles walk_other_alleles(uint32_t col_off, uint32_t tw, uint32_t N, uint32_t ph,
                                                                     uint32_t lps_log, bool mine) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint8_t* const colp = lds + col_off;
  const uint32_t LPS = 1u << lps_log;
  uint32_t cA = 0, cC = 0, cG = 0, cT = 0, cGap = 0, cNn = 0;
  if (mine) {
#pragma unroll 2
    for (uint32_t r = ph; r < N; r += LPS) {
      const uint32_t v = colp[r * tw];
      cA += (v == 'A');
      cC += (v == 'C');
      cG += (v == 'G');
      cT += (v == 'T');
      cGap += (v == '-');
      cNn += (v == 'N');
    }
  }
  cA = group_sum(cA, lps_log);
  cC = group_sum(cC, lps_log);
  cG = group_sum(cG, lps_log);
  cT = group_sum(cT, lps_log);
  cGap = group_sum(cGap, lps_log);
  cNn = group_sum(cNn, lps_log);
  uint32_t sumsq = cA * cA + cC * cC + cG * cG + cT * cT + cGap * cGap + cNn * cNn;
  const uint32_t n_other = mine ? N - (cA + cC + cG + cT + cGap + cNn) : 0u;
  uint32_t extra = 0u;
  if (n_other != 0u) {
    for (uint32_t i = ph; i < N; i += LPS) {
      const uint32_t xi = colp[i * tw];
      const bool six = xi == 'A' || xi == 'C' || xi == 'G' || xi == 'T' || xi == '-' || xi == 'N';
      if (!six) {
        uint32_t cnt = 0u;
        for (uint32_t j = 0; j < N; ++j) cnt += (colp[j * tw] == xi);
        extra += cnt;
      }
    }
  }
  if (__any(n_other != 0u)) sumsq += group_sum(extra, lps_log);
  const uint32_t hi1 = umax(cA, cC), lo1 = umin(cA, cC), hi2 = umax(cG, cT), lo2 = umin(cG, cT);
  return {sumsq, umax(umin(hi1, hi2), umax(lo1, lo2)), (cA ? 1u : 0u) | (cC ? 2u : 0u) | (cT ? 4u : 0u) | (cG ? 8u : 0u)};
}

// STAGE: cache policy of the tile loads, 2 = non-temporal (default), 0 = default policy (A/B).
// MAPPED: the bucket's units read their rows through a row-index list (--pops).  BITMAPS: the plan
// wants per-site bitmaps.  Both compile their branches — and the scalar registers they pin — out
// of the batch path of the CLI and the bench.
template <int STAGE, bool HAS_CDS, bool MAPPED, bool BITMAPS, int TWP>
__global__ void __launch_bounds__(64) scan_kernel(const ScanArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  using G = Geo<TWP>;
  static_assert(!HAS_CDS || TWP >= 128, "the coding pass works on 256- and 128-site tiles");
  static_assert(!BITMAPS || TWP == 256, "bitmaps are laid out per 256-site tile");
  constexpr uint32_t TW = G::TW;

  const uint32_t lane = threadIdx.x;
  const uint32_t unit = A.unit_ids ? A.unit_ids[PFA_UNIT_IDX] : PFA_UNIT_IDX;
  const pfa_desc D = A.descs[unit];
  const bool cds = HAS_CDS && (D.flags & PFA_FLAG_CDS) != 0u;
  const uint32_t N = D.n_rows, L = D.n_sites;
  if (N == 0u || L == 0u) return;
  const uint32_t n_tiles = (L + TW - 1u) / TW;
  const uint32_t t0 = PFA_CHUNK_IDX * A.tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = umin(t0 + A.tiles_per_chunk, n_tiles);

  const LdsLayout lo = lds_layout(TW, A.n_max, HAS_CDS, MAPPED);
  uint8_t* const tile8 = lds + lo.tile;
  // the tile's variable sites; wide entries when a coding launch holds units above 64 rows (pfa_layout.h)
  uint16_t* const list16 = reinterpret_cast<uint16_t*>(lds + lo.list);
  uint32_t* const list32 = reinterpret_cast<uint32_t*>(lds + lo.list);
  const bool wide_list = HAS_CDS && A.n_max > 64u;
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + lo.hist);
  uint32_t* const rowmap = reinterpret_cast<uint32_t*>(lds + lo.rowmap);
  uint32_t* const synhist = reinterpret_cast<uint32_t*>(lds + lo.synhist);
  // entry i of the list once tier 2 has replaced it by the site's H and allele mask; -2, -1: the last
  // two sites of the previous tile
  auto list_hmask = [&](int i) -> HMask<uint32_t> {
    if (!wide_list) { const uint32_t v = list16[i]; return {v & 0xFFFu, v >> 12}; }
    const uint32_t v = list32[i];
    return {v & 0xFFFFFFu, v >> 24};
  };

  const uint32_t nbins = N / 2u;
  const bool mapped = MAPPED && D.row_idx_off != PFA_NO_ROW_IDX;   // population subset: rows through an index list
  if (mapped)
    // the row's byte offset in the matrix, in 16-byte units (plans keep matrices below 64 GB)
    for (uint32_t i = lane; i < N; i += 64u) rowmap[i] = A.row_idx[D.row_idx_off + i] * (D.row_stride >> 4);
  for (uint32_t i = lane; i < nbins; i += 64u) hist[i] = 0u;
  if (cds && lane < kSynLdsBins) synhist[lane] = 0u;
  wave_lds_sync();

  // per-lane partial sums of this chunk
  uint32_t acc_S = 0u, acc_bad = 0u;
  unsigned long long acc_H = 0ull;
  CodonPass<BITMAPS, TWP == 128> cp;   // 128-site tiles: the two lane groups share the codon columns out
  if constexpr (HAS_CDS) cp.init(lane, unit, synhist, A, D, G::LPG - 1u);

  const uint8_t* const mat = A.bases + D.base_off;
  const uint32_t NN = N * N;
  const uint32_t jl = lane % G::LPG, rg = lane / G::LPG;   // word of the tile row, row group

  // ---------------- stage a column tile into LDS (asynchronous)
  auto stage = [&](uint32_t tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t c = lane % G::LPR, rr = lane / G::LPR;
    const uint32_t col = umin(site0 + 16u * c, D.row_stride - 16u);
    const uint32_t n_full = N / G::RPI;
    if (!mapped) {
      // a running per-lane pointer; only the last instruction can be partial (EXEC-masked lanes
      // move nothing and leave their LDS slot untouched)
      const uint8_t* src = mat + (uint64_t)rr * D.row_stride + col;
      const uint64_t step = (uint64_t)G::RPI * D.row_stride;
      uint32_t g = 0;
      for (; g < n_full; ++g, src += step)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
      if (rr < N % G::RPI)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
    } else {
      // The row offsets come out of LDS eight instructions at a time: the compiler keeps every LDS
      // read behind the LDS-DMA issued before it (both touch LDS), so reading them one by one would
      // put an LDS round trip in front of every DMA instruction.
      const uint8_t* const base = mat + col;
      const uint32_t n_inst = n_full + (N % G::RPI ? 1u : 0u);
      for (uint32_t g0 = 0; g0 < n_inst; g0 += 8u) {
        uint32_t off16[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
          const uint32_t r = (g0 + k) * G::RPI + rr;
          off16[k] = rowmap[r < N ? r : 0u];
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
          const uint32_t g = g0 + k;
          if (g < n_full || (g == n_full && rr < N % G::RPI))
            __builtin_amdgcn_global_load_lds((gptr_t)(base + ((uint64_t)off16[k] << 4)), (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
        }
      }
    }
  };
  // The tile of iteration t+1 is requested as soon as iteration t has read the last byte it needs
  // from the tile buffer; what is left of iteration t by then (register / table work of the coding
  // pass) runs under the loads.
  stage(t0);

  const uint32_t steps = N / G::GROUPS, tail = N % G::GROUPS;
  for (uint32_t tile = t0; tile < t1; ++tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t lim = umin(TW, L - site0);  // valid sites of this tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_sync();

    // ---------------- tier 1: running AND / OR down the rows
    uint32_t va = 0xFFFFFFFFu, vo = 0u;
    {
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile8);
#pragma unroll PFA_UNROLL_T1
      for (uint32_t r = 0; r < steps; ++r) {
        const uint32_t x = t32[r * 64u + lane];
        va &= x;
        vo |= x;
      }
      if constexpr (G::GROUPS > 1u) {
        if (rg < tail) {   // the last N % GROUPS rows
          const uint32_t x = t32[steps * 64u + lane];
          va &= x;
          vo |= x;
        }
        // merge the lane groups: afterwards every group holds the words of all N rows
        va &= (uint32_t)__shfl_xor((int)va, 32);
        vo |= (uint32_t)__shfl_xor((int)vo, 32);
        if constexpr (G::GROUPS > 2u) {
          va &= (uint32_t)__shfl_xor((int)va, 16);
          vo |= (uint32_t)__shfl_xor((int)vo, 16);
        }
      }
    }
    const uint32_t vd = va ^ vo;

    // ---------------- compact the variable sites of the tile
    uint32_t P = 0u;
    uint32_t idx4 = 0u;   // list positions of this lane's variable sites, one byte per site (coding pass)
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t off = 4u * jl + b;
      const bool varies = ((vd >> (8u * b)) & 0xFFu) != 0u && off < lim;   // the same in every lane group
      const bool pb = varies && rg == 0u;
      const uint64_t m = __ballot(pb);
      if (varies) {
        // position in the list: variable sites of lower lanes of group 0 (the twin lanes of the other
        // groups note it too: the coding pass of 128-site tiles runs on both)
        const uint32_t i = P + (uint32_t)__popcll(m & ((1ull << jl) - 1ull));
        if (pb) {
          const uint32_t ent = off | ((vd >> (8u * b)) & 0xFFu) << 8;
          if (wide_list) list32[i] = ent;
          else list16[i] = uint16_t(ent);
        }
        if constexpr (HAS_CDS) idx4 |= i << (8u * b);
      }
      if constexpr (BITMAPS)
        if (A.var_bm != nullptr && lane == 0u) A.var_bm[(uint64_t)(D.tile_off + tile) * 4u + b] = m;
      P += (uint32_t)__popcll(m);
    }
    P = __builtin_amdgcn_readfirstlane(P);
    acc_S += (lane == 0u) ? P : 0u;
    wave_lds_sync();

    // ---------------- tier 2: exact allele counts of the variable sites
    // LPS lanes per site walk its column: 4, or 8 in 128-site tiles with at most 8 variable sites (below).  Every row of a column
    // sits in the same LDS bank (rows are 128 or 256 bytes apart), but those conflicts are NOT what bounds the walk: with every
    // intra-group conflict removed the shapes gain 0-2.5 % (profiles/r03_lds_bank_conflict_ablation.txt), and no single stage of
    // tier 2 carries its cost (profiles/r04_classic_stage_ablation.txt).  Fast path: x = the byte of row 0, y = x ^ (XOR byte of the
    // site); when every row holds x or y the site has exactly these two alleles (2 compares per
    // byte).  Otherwise (3+ alleles, ~1 % of variable sites) the column is walked again with six
    // counters, and any byte outside A C G T - N takes the exact generic path.
    auto tier2 = [&](auto lps_tag) {
      constexpr uint32_t LPS = decltype(lps_tag)::value, SPI = 64u / LPS;   // lanes per site, sites per pass
      constexpr uint32_t lps_log = LPS == 4u ? 2u : LPS == 8u ? 3u : 4u;
      const uint32_t slot = lane / LPS, ph = lane % LPS;
      for (uint32_t p0 = 0; p0 < P; p0 += SPI) {
        const uint32_t idx = p0 + slot;
        const bool act = idx < P;
        const uint32_t ent = wide_list ? list32[act ? idx : 0u] : (uint32_t)list16[act ? idx : 0u];
        const uint32_t s = ent & 0xFFu;
        const uint8_t* colp = tile8 + s;
        const uint32_t x = colp[0], y = x ^ ((ent >> 8) & 0xFFu);
        uint32_t c1 = 0, c2 = 0;
#pragma unroll PFA_UNROLL_T2
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
        uint32_t pres = (xc ? 1u << base_code(x) : 0u) | (yc ? 1u << base_code(y) : 0u);   // which of A, C, T, G occur
        const bool multi = act && c1 + c2 != N;   // the same in all lanes of the site's group
        if (__any(multi)) {      // (~1 % of the variable sites: out of line, walk_other_alleles)
          const OtherAlleles o = walk_other_alleles(lo.tile + s, TW, N, ph, lps_log, multi);
          if (multi) {
            sumsq = o.sumsq;
            second = o.second;
            pres = o.pres;
          }
        }
        if (act && ph == 0u) {
          const uint32_t hs = NN - sumsq;
          acc_H += hs;
          if (cds) {   // the list entry becomes the site's H + which of A, C, T, G occur there (codon base order)
            if (wide_list) list32[idx] = hs | pres << 24;
            else list16[idx] = uint16_t(hs | pres << 12);
          }
          if (second == 0u) acc_bad += 1u;
          else lds_add_u32(&hist[second - 1u], 1u);
        }
      }
    };
    // 128-site tiles (units of 65 to about 200 rows) with few variable sites: 8 lanes per site halve the rows a lane walks and
    // with them the dependent LDS round trips of the walk (N = 200 coding 0.722 -> 0.740, N = 200 0.776 -> 0.786; the 256-site
    // tiles of units up to 64 rows have 16 rows per lane at most and more than 8 variable sites as a rule: nothing to gain).
    // (One body with the lanes per site as a wave-uniform run-time value was tried in round 5: N = 200 0.78 -> 0.68, coding
    // 0.75 -> 0.62 — the walk wants its stride as a constant.)
    bool wide_walk = false;
    if constexpr (TWP == 128u) wide_walk = P <= 8u;
    if (wide_walk) {
      if constexpr (TWP == 128u) tier2(std::integral_constant<uint32_t, 8u>{});
    } else {
      tier2(std::integral_constant<uint32_t, 4u>{});
    }

    // ---------------- coding sequences: the codon columns that END in this tile (CodonPass)
    if (cds) {
      wave_lds_sync();   // tier 2's H | mask entries are visible
      cp.window(va, vd, site0, lim);
      // H and allele mask of a variable site (tile-relative, -2 / -1: previous tile) through the list
      // position its owner lane noted: of any site for a wave-uniform `rel` ...
      auto hget_uniform = [&](int rel) -> HMask<uint32_t> {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)idx4, (rel < 0 ? 0 : rel) >> 2);
        return list_hmask(rel < 0 ? rel : (int)((w >> (8u * ((uint32_t)rel & 3u))) & 0xFFu));
      };
      // ... and, per lane, of the sites of this lane and the one before it (every lane active)
      const unsigned long long idx8 = (unsigned long long)wave_shr1(idx4) | (unsigned long long)idx4 << 32;
      auto hget_lane = [&](int rel) -> HMask<uint32_t> {
        const uint32_t k = (uint32_t)(rel - (4 * (int)jl - 4)) & 7u;
        return list_hmask(rel < 0 ? rel : (int)((uint32_t)(idx8 >> (8u * k)) & 0xFFu));
      };
      // columns with two or three variable positions: one row per lane out of the LDS tile; the one or
      // two sites a codon reaches back into the previous tile are read from memory (a chunk starts on a
      // codon boundary, so they belong to the unit; rare)
      auto cell = [&](uint32_t r, int rel) -> uint32_t {
        if (rel >= 0) return tile8[r * TW + (uint32_t)rel];
        const uint64_t row = mapped ? (uint64_t)kernargs_again()->row_idx[D.row_idx_off + r] : (uint64_t)r;
        return mat[row * D.row_stride + site0 - (uint32_t)(-rel)];
      };
      cp.joint_columns(
          [&](int rel) {
            unsigned long long Mv = 0ull;
            for (uint32_t r = lane; r < N; r += 64u) Mv |= codon_bit(cell(r, rel), cell(r, rel + 1), cell(r, rel + 2));
            return Mv;
          },
          hget_uniform);
      // ---- nothing below reads the tile buffer: request the next tile now
      wave_lds_sync();
      if (tile + 1u < t1) stage(tile + 1u);
      cp.mono_and_single_columns(hget_lane);
      if (tile + 1u == n_tiles) cp.last_tile(L);
      // H of the last two sites, for the codons that straddle into the next tile
      if (lane == G::LPG - 1u) {
        const uint32_t i2 = (idx4 >> 16) & 0xFFu, i3 = idx4 >> 24;
        if (wide_list) { list32[-2] = list32[i2]; list32[-1] = list32[i3]; }
        else { list16[-2] = list16[i2]; list16[-1] = list16[i3]; }
      }
    } else {
      wave_lds_sync();   // every read of the tile buffer is done: request the next tile
      if (tile + 1u < t1) stage(tile + 1u);
    }
  }

  // ---------------- flush the chunk's partial sums (output pointers re-read from the kernarg segment)
  acc_S = wave_sum(acc_S);
  acc_bad = wave_sum(acc_bad);
  acc_H = wave_sum(acc_H);
  const kernarg_ptr_t K = kernargs_again();
  const uint32_t rep_off = (PFA_CHUNK_IDX & K->rep_mask) * K->rep_stride;   // this chunk's output replica
  if (lane == 0u) {
    pfa_result* const R = replica(K->results, rep_off) + unit;
    // seg_sites and sfs_bad are neighbours in the record: one 64-bit add (neither can carry: < 2^32 sites)
    if (acc_S | acc_bad) atomicAdd(reinterpret_cast<unsigned long long*>(&R->seg_sites), (unsigned long long)acc_S | (unsigned long long)acc_bad << 32);
    if (acc_H) atomicAdd(reinterpret_cast<unsigned long long*>(&R->H), acc_H);
  }
  wave_lds_sync();
  uint32_t* const sfs = replica(K->sfs, rep_off) + D.sfs_off;
  for (uint32_t i = lane; i < nbins; i += 64u) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&sfs[i], v);
  }
  if (cds) cp.flush(unit);
}

// The device classifier on a list of codon sets (test entry point: pfa_classify_codon_masks_dev)
__global__ void __launch_bounds__(64) classify_masks_kernel(const unsigned long long* __restrict__ masks, uint32_t n,
                                                            uint32_t* __restrict__ labels) {
  const uint32_t lane = threadIdx.x;
  const uint32_t my_info = d_tables.annot_info[lane];
  const unsigned long long my_same = d_tables.same_annot[lane];
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const unsigned long long m = masks[i];
    const unsigned long long mu = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32;
    const uint32_t lab = classify_uniform(mu & ~kHostTables.stop_mask, lane, my_info, my_same);
    if (lane == 0u) labels[i] = lab;
  }
}

// ---------------------------------------------------------------- launchers
template <int STAGE, bool CDS, bool MAPPED, bool BITMAPS, int TW>
static int launch_one(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  auto kern = scan_kernel<STAGE, CDS, MAPPED, BITMAPS, TW>;
  static uint32_t lds_limit_set = 0;  // per instantiation; one process drives one device
  if (c.lds_bytes > lds_limit_set) {
    PFA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)c.lds_bytes));
    lds_limit_set = c.lds_bytes;
  }
  hipLaunchKernelGGL(kern, PFA_GRID(c.n_units, c.n_chunks), dim3(64, 1, 1), c.lds_bytes, st, a);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}

template <int STAGE, bool CDS, int TW>
static int launch_features(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  if constexpr (TW == 256) {
    if (c.bitmaps) return launch_one<STAGE, CDS, true, true, TW>(c, a, st);   // the general instantiation
  }
  if (c.mapped) return launch_one<STAGE, CDS, true, false, TW>(c, a, st);
  return launch_one<STAGE, CDS, false, false, TW>(c, a, st);
}

int launch_classic(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  if (c.tw != 256u && (c.bitmaps || c.stage != 2 || (c.cds && c.tw != 128u))) {
    set_error("internal: narrow tiles serve launches without bitmaps only (coding: 128 sites)");
    return PFA_ERR_ARG;
  }
  if (c.stage == 0) return c.cds ? launch_features<0, true, 256>(c, a, st) : launch_features<0, false, 256>(c, a, st);
  if (c.cds) return c.tw == 256u ? launch_features<2, true, 256>(c, a, st) : launch_features<2, true, 128>(c, a, st);
  if (c.tw == 256u) return launch_features<2, false, 256>(c, a, st);
  if (c.tw == 128u) return launch_features<2, false, 128>(c, a, st);
  return launch_features<2, false, 64>(c, a, st);
}

void warm_classic() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(scan_kernel<2, false, false, false, 128>));
  (void)hipGetLastError();
}

int launch_classify_masks(const unsigned long long* d_masks, uint32_t n, uint32_t* d_labels, hipStream_t st) {
  hipLaunchKernelGGL(classify_masks_kernel, dim3(n < 4096u ? n : 4096u), dim3(64), 0, st, d_masks, n, d_labels);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}

}  // namespace pfa
