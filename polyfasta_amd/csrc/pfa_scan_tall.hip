// pfa_scan_tall.hip — scan_tall_kernel: units with many rows, streamed in row chunks
// (DESIGN.md §3.2).  LDS per wavefront, and with it the occupancy, do not depend on N; N itself
// is bounded by the 32-bit allele counters only (coding units above 2 048 rows keep the per-site
// H in 56 bits: WIDE).
// Reference: getvarsites PolyFastA.py:252-261, nucleotide_diversity :485-492, getsfs :274-282,
// getvarCDSsites :284-315 (no row limit there).
#include "pfa_kernels.h"
#include "pfa_launch.h"

namespace pfa {

// Row-chunked coding scan, rare path (a column turns joint in a chunk after the first): the distinct
// clean codons of the rows BEFORE the chunk, from the per-position alleles seen so far — the
// consensus byte where the position did not vary yet, else the site's count slot as it stood before
// the chunk (`list` / `cnt` at the given LDS byte offsets).  Returns ~0 when a position has no
// usable counts (no slot, or a slot waiting for the column path).  Kept out of line so that its
// scalar state does not weigh on the register allocation of the chunk loop.
__device__ __attribute__((noinline)) unsigned long long codons_before_chunk(uint32_t list_off, uint32_t cnt_off, uint32_t P_prev,
                                                                            uint32_t va_pre, uint32_t vd_pre, uint32_t rel,
                                                                            uint32_t lane) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t* const list = reinterpret_cast<const uint32_t*>(lds + list_off);
  const uint32_t* const cnt = reinterpret_cast<const uint32_t*>(lds + cnt_off);
  uint32_t pm[3] = {0u, 0u, 0u};             // presence masks over A, C, T, G (codon base order)
  bool usable = true;
#pragma unroll
  for (uint32_t q = 0; q < 3u; ++q) {
    const uint32_t sq = rel + q;
    const uint32_t dbyte = ((uint32_t)__builtin_amdgcn_readlane((int)vd_pre, sq >> 2) >> (8u * (sq & 3u))) & 0xFFu;
    const uint32_t abyte = ((uint32_t)__builtin_amdgcn_readlane((int)va_pre, sq >> 2) >> (8u * (sq & 3u))) & 0xFFu;
    if (dbyte == 0u) {                       // every earlier row holds the consensus byte
      pm[q] = base_clean(abyte) ? 1u << base_code(abyte) : 0u;
    } else {                                 // the site's count slot, as it stood before this chunk
      uint64_t hit = 0ull;
      uint32_t base = 0u;
      for (; base < P_prev && hit == 0ull; base += 64u)
        hit = __ballot(base + lane < P_prev && (list[base + lane] >> 24) == sq);
      if (hit == 0ull) { usable = false; continue; }
      const uint32_t j = base - 64u + (uint32_t)__builtin_ctzll(hit);
      const uint32_t ent = list[j];
      const uint32_t mode = (ent >> 16) & 0xFFu;
      const uint32_t* c6 = cnt + j * 6u;
      if (mode == 0u) {
        const uint32_t x = ent & 0xFFu, y = x ^ ((ent >> 8) & 0xFFu);
        pm[q] = ((c6[0] != 0u && base_clean(x)) ? 1u << base_code(x) : 0u) |
                ((c6[1] != 0u && base_clean(y)) ? 1u << base_code(y) : 0u);
      } else if (mode == 1u) {               // counters A C G T - N
        pm[q] = (c6[0] ? 1u : 0u) | (c6[1] ? 2u : 0u) | (c6[3] ? 4u : 0u) | (c6[2] ? 8u : 0u);
      } else {
        usable = false;
      }
    }
  }
  if (!usable) return ~0ull;
  unsigned long long S = 0ull;
#pragma unroll
  for (uint32_t e0 = 0; e0 < 4u; ++e0)
#pragma unroll
    for (uint32_t e1 = 0; e1 < 4u; ++e1)
#pragma unroll
      for (uint32_t e2 = 0; e2 < 4u; ++e2)
        if (((pm[0] >> e0) & (pm[1] >> e1) & (pm[2] >> e2) & 1u) != 0u) S |= 1ull << (16u * e0 + 4u * e1 + e2);
  return S;
}

// ---------------------------------------------------------------- the tall kernel
// Units with many rows (non-coding).  The classic kernel keeps all N rows of a 256-site tile in
// LDS, so its occupancy falls as N grows (6 waves per CU at N = 100, 1 at N = 600).  Here a
// wavefront streams the rows of its tile in chunks of R <= 64 rows; LDS per wave, and with it
// the occupancy, no longer depend on N, and N itself is unbounded.
//   per chunk   stage R rows (LDS-DMA) -> tier 1 (AND / OR of the chunk folded into the running
//               words) -> sites that turn variable in this chunk get a count slot, initialised
//               with "all earlier rows held the consensus byte x" -> tier 2 counts the chunk's
//               rows of every slotted site (4 lanes per site): in two-allele mode the rows
//               equal to x and to y = x ^ (XOR byte when slotted); the first chunk whose rows
//               do not all hold x or y promotes the slot to six counters (A C G T - N), seeded
//               with the exact x / y counts of the earlier chunks
//   per tile    H and the SFS bin of every slot; sites holding other bytes than A C G T - N
//               in six-counter mode, and sites that found no free slot, are recounted exactly
//               from their global column through a 256-bin LDS histogram (whole wave per
//               site; rare)
__device__ inline uint32_t sym6(uint32_t x) {
  return x == 'A' ? 0u : x == 'C' ? 1u : x == 'G' ? 2u : x == 'T' ? 3u : x == '-' ? 4u : x == 'N' ? 5u : 6u;
}

template <int STAGE, bool HAS_CDS, bool MAPPED, bool BITMAPS, bool WIDE>
__global__ void __launch_bounds__(64) scan_tall_kernel(const ScanArgs A) {
  static_assert(HAS_CDS || !WIDE, "only coding units keep a per-site H");
  using HT = typename std::conditional<WIDE, unsigned long long, uint32_t>::type;
  constexpr uint32_t MSH = WIDE ? 56u : 24u;   // hsite entry: H | allele mask << MSH
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t TW = 256u;
  constexpr uint32_t RF = kTallRowsFast;   // chunks of exactly this many rows run straight-line code (the plan's default chunk)
  const uint32_t lane = threadIdx.x;
  const uint32_t unit = A.unit_ids ? A.unit_ids[PFA_UNIT_IDX] : PFA_UNIT_IDX;
  const pfa_desc D = A.descs[unit];
  const uint32_t N = D.n_rows, L = D.n_sites;
  if (N == 0u || L == 0u) return;
  const uint32_t n_tiles = (L + TW - 1u) / TW;
  const uint32_t t0 = PFA_CHUNK_IDX * A.tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = umin(t0 + A.tiles_per_chunk, n_tiles);

  const uint32_t R = A.rows_per_chunk, MAXS = A.max_slots;
  const TallLayout lo = tall_layout(R, MAXS, A.hist_bins, HAS_CDS, WIDE);
  uint8_t* const tile8 = lds + lo.tile;
  uint32_t* const list = reinterpret_cast<uint32_t*>(lds + lo.list);
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds + lo.cnt);
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + lo.hist);
  uint8_t* const xlist = lds + lo.list;                                      // end of tile: sites for the column path
  uint32_t* const colhist = reinterpret_cast<uint32_t*>(lds + lo.list + 256u);   // and its 256 bins

  HT* const hsite = reinterpret_cast<HT*>(lds + lo.hsite);
  uint32_t* const synhist = reinterpret_cast<uint32_t*>(lds + lo.synhist);
  const bool cds = HAS_CDS && (D.flags & PFA_FLAG_CDS) != 0u;

  const uint32_t nbins = N / 2u, HB = umin(nbins, A.hist_bins);
  const bool mapped = MAPPED && D.row_idx_off != PFA_NO_ROW_IDX;   // population subset: rows through an index list
  const uint32_t* const ridx = A.row_idx + (mapped ? D.row_idx_off : 0u);
  const bool multi = N > R;
  for (uint32_t i = lane; i < HB; i += 64u) hist[i] = 0u;
  if (cds && lane < kSynLdsBins) synhist[lane] = 0u;
  // lane r holds the matrix row of the chunk's row r; the next chunk's are fetched while this one is scanned
  uint32_t idx_cur = 0u;
  const uint32_t stride16 = D.row_stride >> 4;
  if (mapped && lane < umin(R, N)) idx_cur = ridx[lane] * stride16;
  wave_lds_sync();

  uint32_t acc_S = 0u, acc_bad = 0u;
  unsigned long long acc_H = 0ull;
  const uint8_t* const mat = A.bases + D.base_off;
  const unsigned long long NN = (unsigned long long)N * N;
  CodonPass<BITMAPS> cp;
  if constexpr (HAS_CDS) cp.init(lane, unit, synhist, A, D);
  auto hsite_get = [&](int rel) -> HMask<HT> {
    const HT v = hsite[rel];
    return {v & (((HT)1 << MSH) - 1u), (uint32_t)(v >> MSH)};
  };
  // H and which of A, C, T, G occur (codon base order) of a variable site, for the coding pass
  auto hsite_put = [&](uint32_t s, unsigned long long hs, uint32_t cA, uint32_t cC, uint32_t cG, uint32_t cT) {
    hsite[s] = (HT)hs | (HT)((cA ? 1u : 0u) | (cC ? 2u : 0u) | (cT ? 4u : 0u) | (cG ? 8u : 0u)) << MSH;
  };

  // second-largest A/C/G/T count (PolyFastA.py:279-281) -> LDS histogram, high bins to global
  auto sfs_add = [&](uint32_t cA, uint32_t cC, uint32_t cG, uint32_t cT) {
    const uint32_t hi1 = umax(cA, cC), lo1 = umin(cA, cC), hi2 = umax(cG, cT), lo2 = umin(cG, cT);
    const uint32_t second = umax(umin(hi1, hi2), umax(lo1, lo2));
    if (second == 0u) acc_bad += 1u;
    else if (second - 1u < HB) atomicAdd(&hist[second - 1u], 1u);   // (lds_add_u32, pfa_kernels.h, measured here: equal within the spread)
    else {
      const kernarg_ptr_t K = kernargs_again();
      atomicAdd(replica(K->sfs, (PFA_CHUNK_IDX & K->rep_mask) * K->rep_stride) + D.sfs_off + second - 1u, 1u);
    }
  };

  // word offset of a codon in its row (read-back of columns that no codon slot tracked): the last
  // codon of a row that fills its stride is read one byte earlier, so that no load leaves the row
  auto codon_word_at = [&](uint32_t first_site, uint32_t& off, uint32_t& sh) {
    const bool at_end = first_site + 4u > D.row_stride;
    off = at_end ? first_site - 1u : first_site;
    sh = at_end ? 8u : 0u;
  };

  // Codon slots (coding units).  The rows of a column with two or three variable positions are gone
  // from LDS by the time the column is recognised — unless its codon set is accumulated on the
  // way: from the chunk in which the column's SECOND position starts to vary, every chunk adds the
  // codons of its rows (one row per lane, OR-reduced).  The rows before that chunk held the
  // consensus byte everywhere except at the one position that already varied, so their codons are
  // the consensus codon with that position's alleles substituted — known from its count slot.
  // Columns that cannot be tracked (they reach back into the previous tile, the slot table is full,
  // a position without usable counts) are read back from memory at the end of the tile.
  constexpr uint32_t CS = 8u;
  int* const cs_end = reinterpret_cast<int*>(lds + lo.cslots);                                   // tile-relative end site
  unsigned long long* const cs_set = reinterpret_cast<unsigned long long*>(lds + lo.cslots + 32u);
  int endoT[2] = {-1, -1};         // this lane's candidate codon columns of the tile (as in CodonPass::window)
  uint32_t shT[2] = {0u, 0u};
  auto joint_ballots = [&](uint32_t vdrun, uint64_t (&jb)[2]) {
    uint32_t pd = wave_shr1(vdrun);
    if (lane == 0u) pd = cp.carry_vd;
    const unsigned long long Wd = (unsigned long long)pd | (unsigned long long)vdrun << 32;
#pragma unroll
    for (uint32_t it = 0; it < 2u; ++it) {
      const uint32_t dv = (uint32_t)(Wd >> shT[it]) & 0xFFFFFFu;
      const uint32_t nvar = (uint32_t)((dv & 0xFFu) != 0u) + (uint32_t)((dv & 0xFF00u) != 0u) + (uint32_t)((dv & 0xFF0000u) != 0u);
      jb[it] = __ballot(endoT[it] >= 0 && nvar >= 2u);
    }
  };

  // ---------------- stage rows [r0, r0 + Rc) x 256 sites of a tile (asynchronous for LDS-DMA)
  const uint32_t c16 = lane % 16u, rr = lane / 16u;
  typedef __attribute__((address_space(3))) uint8_t* lds8_t;
  const lds8_t tile3 = (lds8_t)tile8;   // LDS address arithmetic in its own address space (no null checks per instruction)
  auto stage_chunk = [&](uint32_t tile, uint32_t r0, uint32_t idx_rows) {
    const uint32_t Rc = umin(R, N - r0);
    const uint32_t col = umin(tile * TW + 16u * c16, D.row_stride - 16u);
    if (!mapped) {
      const uint64_t step = (uint64_t)D.row_stride * 4u;
      if (Rc == RF) {
        // a whole chunk, written out: twelve loads, the address a 64-bit add and the LDS offset (M0) a scalar add apart;
        // the M0 write and the next load are separated by the address update (one wait state is required).
        // (Left to the compiler, the twelve addresses sit in 24 VGPRs and the twelve LDS offsets in spilled SGPRs.)
        static_assert(RF == 48u, "twelve LDS-DMA instructions below");
        const uint8_t* src = mat + (uint64_t)(r0 + rr) * D.row_stride + col;
        const uint32_t lds0 = (uint32_t)(uintptr_t)tile3;
#define PFA_LD(pol) "global_load_lds_dwordx4 %0, off" pol "\n\ts_add_u32 m0, m0, 0x400\n\tv_lshl_add_u64 %0, %0, 0, %2\n\t"
#define PFA_LD12(pol) "s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) \
                      PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) PFA_LD(pol) "global_load_lds_dwordx4 %0, off" pol "\n\ts_mov_b32 m0, %1"
        uint32_t m0_saved;   // M0 is the compiler's: handed back as found
        if constexpr (STAGE == 2) asm volatile(PFA_LD12(" nt") : "+v"(src), "=&s"(m0_saved) : "s"(step), "s"(lds0) : "memory", "scc");
        else asm volatile(PFA_LD12("") : "+v"(src), "=&s"(m0_saved) : "s"(step), "s"(lds0) : "memory", "scc");
#undef PFA_LD12
#undef PFA_LD
        return;
      }
      // one address per lane and chunk, then a 64-bit step per instruction; whole groups of 4 rows need no predicate
      const uint8_t* src = mat + (uint64_t)(r0 + rr) * D.row_stride + col;
      const uint32_t n_full = Rc >> 2;
      uint32_t g = 0;
#pragma unroll 4
      for (; g < n_full; ++g) {
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile3 + g * 1024u), 16, 0, STAGE);
        src += step;
      }
      if (rr < (Rc & 3u)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile3 + g * 1024u), 16, 0, STAGE);
      return;
    }
    const uint32_t n_inst = (Rc + 3u) / 4u;
    for (uint32_t g = 0; g < n_inst; ++g) {
      const uint32_t r = 4u * g + rr;
      // the lanes hold byte offsets of the chunk's rows in 16-byte units
      const uint32_t held = (uint32_t)__shfl((int)idx_rows, (int)(r & 63u));
      if (r < Rc) __builtin_amdgcn_global_load_lds((gptr_t)(mat + ((uint64_t)held << 4) + col), (lptr_t)(tile3 + g * 1024u), 16, 0, STAGE);
    }
  };
  // A chunk is requested as soon as the previous one has been read for the last time: the next
  // chunk of the tile, or the first chunk of the next tile — whose loads then run under the
  // end-of-tile work (H, SFS, column path), which touches only the slot tables.
  stage_chunk(t0, 0u, idx_cur);

  for (uint32_t tile = t0; tile < t1; ++tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t lim = umin(TW, L - site0);
    uint32_t va = 0xFFFFFFFFu, vo = 0u, vd = 0u;   // running AND / OR / XOR words of the rows seen so far
    uint32_t P = 0u, ovf = 0u;                     // variable sites so far; own sites that found no slot
    bool defer_request = false;
    uint32_t n_joint = 0u;
    uint32_t ncs = 0u;                       // codon slots in use (wave-uniform)
    uint64_t jb_prev[2] = {0ull, 0ull};      // columns that were joint after the previous chunk
    uint64_t tracked[2] = {0ull, 0ull};      // joint columns of the finished tile whose set sits in a slot
    if (cds) {
      const uint32_t o1 = cp.first_codon_end(site0);
#pragma unroll
      for (uint32_t it = 0; it < 2u; ++it) {
        const uint32_t o = o1 + 3u * it;
        endoT[it] = (o < 4u && 4u * lane + o < lim) ? (int)(4u * lane + o) : -1;
        shT[it] = 8u * (2u + (o & 3u));
      }
    }

    for (uint32_t r0 = 0; r0 < N; r0 += R) {
      const uint32_t Rc = umin(R, N - r0);
      uint32_t idx_nxt = idx_cur;
      if (mapped && multi) {
        const uint32_t r0n = r0 + R < N ? r0 + R : 0u;
        if (lane < umin(R, N - r0n)) idx_nxt = ridx[r0n + lane] * stride16;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wave_lds_sync();
      if (cds && r0 == 0u && tile > t0) {
        // H of the previous tile's last two sites, for the codons that straddle into this tile
        if (lane < 2u) hsite[(int)lane - 2] = hsite[TW - 2u + lane];
        wave_lds_sync();
      }

      // ---------------- tier 1 on the chunk, folded into the running words
      uint32_t ca = 0xFFFFFFFFu, co = 0u;
      const uint32_t x0w = reinterpret_cast<const uint32_t*>(tile8)[lane];   // first row of the chunk
      {
        // two rows per step, so that AND and OR take their three-input forms (one instruction per row and word);
        // 16 rows per trip of the main loop: eight paired LDS reads in flight
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile8) + lane;
        const uint32_t n16 = Rc >> 4;
#pragma clang loop unroll(disable) vectorize(disable)
        for (uint32_t i = 0; i < n16; ++i) {
          // all eight paired reads in flight, then one wait (written out: under register pressure the compiler waits for
          // every read before it issues the next, eight LDS round trips per 16 rows instead of one)
          unsigned long long p0, p1, p2, p3, p4, p5, p6, p7;
          const uint32_t at = (uint32_t)(uintptr_t)(tile8 + i * 4096u) + lane * 4u;
          asm volatile(
              "ds_read2st64_b32 %0, %8 offset1:1\n\t"
              "ds_read2st64_b32 %1, %8 offset0:2 offset1:3\n\t"
              "ds_read2st64_b32 %2, %8 offset0:4 offset1:5\n\t"
              "ds_read2st64_b32 %3, %8 offset0:6 offset1:7\n\t"
              "ds_read2st64_b32 %4, %8 offset0:8 offset1:9\n\t"
              "ds_read2st64_b32 %5, %8 offset0:10 offset1:11\n\t"
              "ds_read2st64_b32 %6, %8 offset0:12 offset1:13\n\t"
              "ds_read2st64_b32 %7, %8 offset0:14 offset1:15\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
              : "v"(at)
              : "memory");
          const unsigned long long ps[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
#pragma unroll
          for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t x0 = (uint32_t)ps[k], x1 = (uint32_t)(ps[k] >> 32);
            ca = ca & x0 & x1;
            co = co | x0 | x1;
          }
        }
#pragma clang loop unroll(disable) vectorize(disable)
        for (uint32_t r = n16 << 4; r < Rc; ++r) {
          const uint32_t x = t32[r * 64u];
          ca &= x;
          co |= x;
        }
      }
      const uint32_t nva = va & ca, nvo = vo | co, nvd = nva ^ nvo;

      // ---------------- sites that turn variable in this chunk get a count slot (most chunks after the first have none)
      const bool any_new = __any(byte_nonzero_bits(nvd) != byte_nonzero_bits(vd));
      if (any_new) {
#pragma unroll
      for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t off = 4u * lane + b;
        const bool now = ((nvd >> (8u * b)) & 0xFFu) != 0u && off < lim;
        const bool was = ((vd >> (8u * b)) & 0xFFu) != 0u;
        const bool newly = now && !was;
        const uint64_t m = __ballot(newly);
        if (m != 0ull) {
          if (newly) {
            const uint32_t slot = P + mbcnt64(m);
            if (slot < MAXS) {
              // x: the byte every earlier row held (the site's AND byte), or that of the very first row
              const uint32_t xb = ((r0 ? va : x0w) >> (8u * b)) & 0xFFu;
              list[slot] = xb | ((nvd >> (8u * b)) & 0xFFu) << 8 | off << 24;   // mode 0: two alleles
              cnt[slot * 6u] = r0;
              cnt[slot * 6u + 1u] = 0u;
            } else {
              ovf |= 1u << b;
            }
          }
          P += (uint32_t)__popcll(m);
        }
      }
      }
      const uint32_t va_pre = va, vd_pre = vd, P_prev = umin(P, MAXS);   // state before this chunk
      va = nva;
      vo = nvo;
      vd = nvd;
      P = __builtin_amdgcn_readfirstlane(P);
      const uint32_t Pc = umin(P, MAXS);
      wave_lds_sync();

      // ---------------- coding units: columns that turn joint in this chunk get a codon slot
      if (cds && any_new) {   // a column can only turn joint in a chunk in which one of its sites turns variable
        uint64_t jb_now[2];
        joint_ballots(nvd, jb_now);
#pragma unroll
        for (uint32_t it = 0; it < 2u; ++it) {
          for (uint64_t m = jb_now[it] & ~jb_prev[it]; m != 0ull; m &= m - 1ull) {
            const int end = __builtin_amdgcn_readlane(endoT[it], (uint32_t)__builtin_ctzll(m));
            const int rel = end - 2;
            if (rel < 0 || ncs >= CS) continue;          // read back at the end of the tile
            // the distinct clean codons of the rows before this chunk
            const unsigned long long S = r0 == 0u ? 0ull : wave_uniform(codons_before_chunk(lo.list, lo.cnt, P_prev, va_pre, vd_pre, (uint32_t)rel, lane));
            if (S == ~0ull) continue;                    // a position without usable counts: read back
            if (lane == 0u) { cs_end[ncs] = end; cs_set[ncs] = S; }
            ++ncs;
          }
          jb_prev[it] = jb_now[it];
        }
        wave_lds_sync();
      }

      // ---------------- tier 2: the chunk's rows of every slotted site
      const uint32_t slot16 = lane >> 2, ph = lane & 3u;
      const uint32_t rows_after = r0 + Rc;
      for (uint32_t p0 = 0; p0 < Pc; p0 += 16u) {
        const uint32_t idx = p0 + slot16;
        const bool act = idx < Pc;
        const uint32_t ent = list[act ? idx : 0u];
        const uint32_t x = ent & 0xFFu, y = x ^ ((ent >> 8) & 0xFFu), mode = (ent >> 16) & 0xFFu;
        const uint8_t* colp = tile8 + (ent >> 24);
        uint32_t* c6 = cnt + (act ? idx : 0u) * 6u;
        bool need6 = act && mode == 1u;
        if (act && mode == 0u) {   // the same in all 4 lanes of the quad
          uint32_t c1 = 0, c2 = 0;
          {
            // lane ph of the quad: rows ph, ph + 4, ...; the trip count is the same in every lane (scalar loop)
            const uint8_t* q = colp + ph * TW;
            if (Rc == RF) {
              static_assert(RF == 48u, "twelve reads per lane below");
              // The twelve bytes of this lane: all reads in flight and one wait (see tier 1), packed four to a word and compared
              // with x and with y byte-parallel: a byte of w ^ xxxx is zero where the row holds x (12 compares + 12 adds per
              // value otherwise).  (d16 loads, which deliver two bytes per register, cost more than the packing they save:
              // 0.79 -> 0.69 at N = 1 000.)
              uint32_t v[12];
              const uint32_t at = (uint32_t)(uintptr_t)q;
              asm volatile(
                  "ds_read_u8 %0, %12\n\tds_read_u8 %1, %12 offset:1024\n\tds_read_u8 %2, %12 offset:2048\n\t"
                  "ds_read_u8 %3, %12 offset:3072\n\tds_read_u8 %4, %12 offset:4096\n\tds_read_u8 %5, %12 offset:5120\n\t"
                  "ds_read_u8 %6, %12 offset:6144\n\tds_read_u8 %7, %12 offset:7168\n\tds_read_u8 %8, %12 offset:8192\n\t"
                  "ds_read_u8 %9, %12 offset:9216\n\tds_read_u8 %10, %12 offset:10240\n\tds_read_u8 %11, %12 offset:11264\n\t"
                  "s_waitcnt lgkmcnt(0)"
                  : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
                    "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                  : "v"(at)
                  : "memory");
              const uint32_t x4 = __builtin_amdgcn_perm(0u, x, 0u), y4 = __builtin_amdgcn_perm(0u, y, 0u);   // the byte in all four
              uint32_t nx = 0u, ny = 0u;   // rows that do not hold x / y
#pragma unroll
              for (uint32_t j = 0; j < 3u; ++j) {
                const uint32_t w = (v[4u * j] | v[4u * j + 1u] << 16) | (v[4u * j + 2u] | v[4u * j + 3u] << 16) << 8;
                nx += (uint32_t)__builtin_popcount(byte_nonzero_bits(w ^ x4));
                ny += (uint32_t)__builtin_popcount(byte_nonzero_bits(w ^ y4));
              }
              c1 = 12u - nx;
              c2 = 12u - ny;
            } else {
              const uint32_t n4 = Rc >> 2;
#pragma unroll 4
              for (uint32_t i = 0; i < n4; ++i) {
                const uint32_t v = q[i * (4u * TW)];
                c1 += (v == x);
                c2 += (v == y);
              }
              if (ph < (Rc & 3u)) {
                const uint32_t v = q[n4 * (4u * TW)];
                c1 += (v == x);
                c2 += (v == y);
              }
            }
          }
          c1 = quad_sum(c1);
          c2 = quad_sum(c2);
          const uint32_t o1 = c6[0], o2 = c6[1];
          if (o1 + o2 + c1 + c2 == rows_after) {
            if (ph == 0u) { c6[0] = o1 + c1; c6[1] = o2 + c2; }
          } else {
            // a third allele in this chunk: the earlier rows held exactly o1 x's and o2 y's
            const uint32_t sx = sym6(x), sy = sym6(y);
            const bool odd = (o1 != 0u && sx == 6u) || (o2 != 0u && sy == 6u);
            if (ph == 0u) {
#pragma unroll
              for (uint32_t k = 0; k < 6u; ++k) c6[k] = (sx == k ? o1 : 0u) + (sy == k ? o2 : 0u);
              list[idx] = (ent & 0xFF00FFFFu) | (odd ? 2u : 1u) << 16;   // mode 2: column path at the end of the tile
            }
            need6 = !odd;
          }
        }
        if (__any(need6)) {
          if (need6) {
            uint32_t cA = 0, cC = 0, cG = 0, cT = 0, cGap = 0, cNn = 0;
            const uint32_t n4 = (Rc + 3u) >> 2;
#pragma unroll 4
            for (uint32_t i = 0; i < n4; ++i) {
              const uint32_t r = 4u * i + ph;
              const uint32_t v = r < Rc ? colp[r * TW] : 0u;
              cA += (v == 'A');
              cC += (v == 'C');
              cG += (v == 'G');
              cT += (v == 'T');
              cGap += (v == '-');
              cNn += (v == 'N');
            }
            cA = quad_sum(cA);
            cC = quad_sum(cC);
            cG = quad_sum(cG);
            cT = quad_sum(cT);
            cGap = quad_sum(cGap);
            cNn = quad_sum(cNn);
            if (ph == 0u) {
              c6[0] += cA;
              c6[1] += cC;
              c6[2] += cG;
              c6[3] += cT;
              c6[4] += cGap;
              c6[5] += cNn;
            }
          }
        }
      }
      // ---------------- coding units: the chunk's rows of every codon slot (one row per lane)
      for (uint32_t k = 0; k < ncs; ++k) {
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(cs_end[k] - 2);
        const uint8_t* q = tile8 + lane * TW + first;
        const uint32_t w = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16;
        // three bytes at once: 2-bit codes, the letters those codes stand for, and whether they are the bytes read
        const uint32_t codes = (w >> 1) & 0x030303u;
        const bool ok = __builtin_amdgcn_perm(0u, 0x47544341u, codes | 0x0C000000u) == w && lane < Rc;
        const uint32_t idx = ((codes << 4) | (codes >> 6) | (codes >> 16)) & 63u;
        // (the set as a ballot of the lanes that a ds_permute of 1s to lane `idx` reaches: no faster than the DPP OR)
        const unsigned long long M = wave_or_uniform(ok ? 1ull << idx : 0ull);
        if (lane == 0u) cs_set[k] |= M;
      }
      wave_lds_sync();   // the tile buffer has been read for the last time
      idx_cur = idx_nxt;
      if (r0 + R < N) {
        stage_chunk(tile, r0 + R, idx_cur);
      } else {
        // Coding units: the register window over the final tier-1 words gives the tile's codon
        // columns with two or three variable positions.
        if (cds) {
          cp.window(va, vd, site0, lim);
          // joint columns whose codon set sits in a slot need no read-back
#pragma unroll
          for (uint32_t it = 0; it < 2u; ++it) {
            for (uint64_t m = cp.mjoint[it]; m != 0ull; m &= m - 1ull) {
              const uint32_t owner = (uint32_t)__builtin_ctzll(m);
              const int end = __builtin_amdgcn_readlane(cp.endo[it], owner);
              bool found = false;
              for (uint32_t k = 0; k < ncs; ++k) found = found || cs_end[k] == end;
              if (found) tracked[it] |= 1ull << owner;
            }
            cp.mjoint[it] &= ~tracked[it];
          }
          n_joint = (uint32_t)__popcll(cp.mjoint[0]) + (uint32_t)__popcll(cp.mjoint[1]);
          // the others (rare) are read back from global memory; those reads would queue behind the
          // next tile's loads, so in that case the request waits for them
          defer_request = n_joint != 0u;
        }
        if (!defer_request && tile + 1u < t1) stage_chunk(tile + 1u, 0u, idx_cur);
      }
    }

    // ---------------- end of tile: S, bitmap, H and SFS of the slotted sites
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const bool pb = ((vd >> (8u * b)) & 0xFFu) != 0u && 4u * lane + b < lim;
      const uint64_t m = __ballot(pb);
      if (BITMAPS && A.var_bm != nullptr && lane == 0u) A.var_bm[(uint64_t)(D.tile_off + tile) * 4u + b] = m;
      acc_S += lane == 0u ? (uint32_t)__popcll(m) : 0u;
    }
    const uint32_t Pc = umin(P, MAXS);
    uint32_t X = 0u;   // sites for the column path
    for (uint32_t base = 0; base < Pc; base += 64u) {
      const uint32_t idx = base + lane;
      const bool act = idx < Pc;
      bool ex = false;
      uint32_t ent = 0u;
      if (act) {
        ent = list[idx];
        const uint32_t mode = (ent >> 16) & 0xFFu;
        const uint32_t* c6 = cnt + idx * 6u;
        if (mode == 0u) {           // exactly two alleles, x and y
          const uint32_t x = ent & 0xFFu, y = x ^ ((ent >> 8) & 0xFFu);
          const uint32_t c1 = c6[0], c2 = c6[1];
          const unsigned long long hs = NN - ((unsigned long long)c1 * c1 + (unsigned long long)c2 * c2);
          acc_H += hs;
          const uint32_t second = (base_clean(x) && base_clean(y)) ? umin(c1, c2) : 0u;
          sfs_add(second, 0u, 0u, second);
          if (cds) {
            const uint32_t px = base_clean(x) ? 1u << base_code(x) : 0u, py = base_clean(y) ? 1u << base_code(y) : 0u;
            hsite[ent >> 24] = (HT)hs | (HT)(px | py) << MSH;
          }
        } else if (mode == 1u) {
          const uint32_t cA = c6[0], cC = c6[1], cG = c6[2], cT = c6[3], cGap = c6[4], cNn = c6[5];
          ex = cA + cC + cG + cT + cGap + cNn != N;   // some row holds another byte
          if (!ex) {
            const unsigned long long sq = (unsigned long long)cA * cA + (unsigned long long)cC * cC +
                                          (unsigned long long)cG * cG + (unsigned long long)cT * cT +
                                          (unsigned long long)cGap * cGap + (unsigned long long)cNn * cNn;
            acc_H += NN - sq;
            sfs_add(cA, cC, cG, cT);
            if (cds) hsite_put(ent >> 24, NN - sq, cA, cC, cG, cT);
          }
        } else {
          ex = true;
        }
      }
      const uint64_t m = __ballot(ex);
      if (ex) xlist[X + mbcnt64(m)] = uint8_t(ent >> 24);
      X += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const bool o = (ovf >> b) & 1u;
      const uint64_t m = __ballot(o);
      if (o) xlist[X + mbcnt64(m)] = uint8_t(4u * lane + b);
      X += (uint32_t)__popcll(m);
    }
    X = __builtin_amdgcn_readfirstlane(X);
    wave_lds_sync();

    // ---------------- column path: exact allele counts of one site from global memory
    for (uint32_t i = 0; i < X; ++i) {
      const uint32_t s = xlist[i];
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) colhist[lane + 64u * k] = 0u;
      wave_lds_sync();
      const uint8_t* gcol = mat + site0 + s;
      for (uint32_t r = lane; r < N; r += 64u) {
        const uint32_t row = mapped ? ridx[r] : r;
        atomicAdd(&colhist[gcol[(uint64_t)row * D.row_stride]], 1u);
      }
      wave_lds_sync();
      unsigned long long sq = 0ull;
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const unsigned long long v = colhist[lane + 64u * k];
        sq += v * v;
      }
      sq = wave_sum(sq);
      if (lane == 0u) {
        acc_H += NN - sq;
        sfs_add(colhist['A'], colhist['C'], colhist['G'], colhist['T']);
        if (cds) hsite_put(s, NN - sq, colhist['A'], colhist['C'], colhist['G'], colhist['T']);
      }
      wave_lds_sync();
    }

    // ---------------- coding units: the codon columns that END in this tile (CodonPass)
    if (cds) {
      wave_lds_sync();   // every hsite entry of the tile is visible
      // columns with two or three variable positions: their rows come from global memory (one row
      // per lane); a chunk starts on a codon boundary, so site0 + rel never falls before the unit
#pragma unroll
      for (uint32_t it = 0; it < 2u; ++it) {
        for (uint64_t m = tracked[it]; m != 0ull; m &= m - 1ull) {
          const int end = __builtin_amdgcn_readlane(cp.endo[it], (uint32_t)__builtin_ctzll(m));
          unsigned long long M = 0ull;
          for (uint32_t k = 0; k < ncs; ++k)
            if (cs_end[k] == end) M = cs_set[k];
          cp.joint_finish(M, end - 2, hsite_get);
        }
      }
      if (n_joint) {
        cp.joint_columns_batched(
            [&](const int (&rels)[4], uint32_t n, unsigned long long (&Mv)[4]) {
              uint32_t off[4], sh[4];
#pragma unroll
              for (uint32_t c = 0; c < 4u; ++c) codon_word_at(site0 + (uint32_t)rels[c < n ? c : 0u], off[c], sh[c]);
              for (uint32_t r = lane; r < N; r += 64u) {
                const uint8_t* q = mat + (uint64_t)(mapped ? ridx[r] : r) * D.row_stride;
                uint32_t w[4];
#pragma unroll
                for (uint32_t c = 0; c < 4u; ++c) {
                  uint32_t v = 0u;
                  if (c < n) __builtin_memcpy(&v, q + off[c], 4);
                  w[c] = v >> sh[c];
                }
#pragma unroll
                for (uint32_t c = 0; c < 4u; ++c)
                  if (c < n) Mv[c] |= codon_bit(w[c] & 0xFFu, (w[c] >> 8) & 0xFFu, (w[c] >> 16) & 0xFFu);
              }
            },
            hsite_get);
      }
      if (defer_request && tile + 1u < t1) stage_chunk(tile + 1u, 0u, idx_cur);
      cp.mono_and_single_columns(hsite_get);
      if (tile + 1u == n_tiles) cp.last_tile(L);
      wave_lds_sync();
    }
  }

  // ---------------- flush the chunk's partial sums (output pointers re-read from the kernarg segment)
  acc_S = wave_sum(acc_S);
  acc_bad = wave_sum(acc_bad);
  acc_H = wave_sum(acc_H);
  const kernarg_ptr_t K = kernargs_again();
  const uint32_t rep_off = (PFA_CHUNK_IDX & K->rep_mask) * K->rep_stride;   // this chunk's output replica
  if (lane == 0u) {
    pfa_result* const Rz = replica(K->results, rep_off) + unit;
    if (acc_S | acc_bad) atomicAdd(reinterpret_cast<unsigned long long*>(&Rz->seg_sites), (unsigned long long)acc_S | (unsigned long long)acc_bad << 32);
    if (acc_H) atomicAdd(reinterpret_cast<unsigned long long*>(&Rz->H), acc_H);
  }
  wave_lds_sync();
  uint32_t* const sfs = replica(K->sfs, rep_off) + D.sfs_off;
  for (uint32_t i = lane; i < HB; i += 64u) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&sfs[i], v);
  }
  if (cds) cp.flush(unit);
}

// ---------------------------------------------------------------- launchers
template <int STAGE, bool CDS, bool MAPPED, bool BITMAPS, bool WIDE>
static int launch_one(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  auto kern = scan_tall_kernel<STAGE, CDS, MAPPED, BITMAPS, WIDE>;
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

template <int STAGE, bool CDS, bool WIDE>
static int launch_features(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  if (c.bitmaps || c.mapped) return launch_one<STAGE, CDS, true, true, WIDE>(c, a, st);   // the general instantiation
  return launch_one<STAGE, CDS, false, false, WIDE>(c, a, st);
}

int launch_tall(const LaunchCfg& c, const ScanArgs& a, hipStream_t st) {
  if (c.wide && !c.cds) { set_error("internal: only coding launches are wide"); return PFA_ERR_ARG; }
  if (c.stage == 0 && !c.wide) return c.cds ? launch_features<0, true, false>(c, a, st) : launch_features<0, false, false>(c, a, st);
  if (c.wide) return launch_features<2, true, true>(c, a, st);
  return c.cds ? launch_features<2, true, false>(c, a, st) : launch_features<2, false, false>(c, a, st);
}

}  // namespace pfa
