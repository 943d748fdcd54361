// pfa_scan.hip — the polymorphism-scan kernels of libpfa_hip.so (gfx950 only)
// and the plan / launch entry points of include/pfa.h.
//
// What is computed (all integers), per scan unit of N rows:
//   S      = number of site columns with more than one distinct byte   PolyFastA.py:252-261
//   H      = sum over those columns of (N^2 - sum_j c_j^2)              PolyFastA.py:485-492
//   SFS    = histogram of the second-largest A/C/G/T allele count        PolyFastA.py:274-282
//   coding = per codon column the set of distinct clean codons -> nstops,
//            missing columns, synonymous-site numerators, and the S / N label
//            of each variable codon position with the H of its site       PolyFastA.py:284-315,319-434
//
// Kernel shape.  One 64-lane wavefront per workgroup; no
// inter-wave synchronisation anywhere.  A wavefront owns a chunk of
// consecutive column tiles of one unit.  Per tile (scan_kernel):
//   stage   N rows x 256 sites -> LDS with LDS-DMA (global_load_lds_dwordx4,
//           one KiB per instruction, no VGPR round trip); each instruction
//           covers 4 rows so every lane moves 16 contiguous bytes
//   tier 1  lane = 4 adjacent sites; walk the rows with one conflict-free
//           ds_read per row, keep running AND / OR of the bytes: a site is
//           monomorphic iff AND == OR (2 VALU ops per row per lane)
//   compact ballot + mbcnt turn the per-lane flags into a dense list of the
//           tile's variable sites (typically ~4 % of the sites)
//   tier 2  4 lanes per variable site walk the column out of LDS: two-allele
//           fast path (x = row 0's byte, y = x ^ XOR byte), else six counters
//           A C G T - N, else an exact generic path (sum_i #{j : x_j = x_i});
//           H, SFS bin (LDS histogram), optional bitmaps
//   coding  register window over the tier-1 words: monomorphic codon columns
//           from a LUT, columns with one variable position from a compile-time
//           table on the owning lane, columns with two or three (rare) read
//           row by row, OR-reduced with DPP and classified wave-uniformly
// Plans whose units have too many rows to keep the CU full that way run
// scan_tall_kernel, which streams the rows of a tile in chunks (LDS per wave
// independent of N) and hands the same per-site results to the coding pass.
// Per-chunk partial sums leave through integer atomics (order-independent).
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "pfa_codon.h"
#include "pfa_common.h"

namespace pfa {

Context& ctx() {
  static Context c;
  return c;
}

static thread_local char g_err[768] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static constexpr ct::Tables kHostTables = ct::make_tables();   // also the source of literal constants in device code

}  // namespace pfa

__constant__ pfa::ct::Tables d_tables = pfa::ct::make_tables();
__constant__ pfa::SingleLut d_single_lut = pfa::make_single_lut();

namespace pfa {

// ---------------------------------------------------------------- geometry
// Every kernel works on 256-site tiles: a lane owns LB = 4 adjacent sites (one 32-bit word of a
// tile row), an LDS-DMA instruction moves 4 rows x 256 bytes.
static constexpr int LB = 4;
struct G {
  static constexpr uint32_t WW = 64u * LB;            // sites (= bytes) per LDS tile row
  static constexpr uint32_t LPR = WW / 16u;           // DMA lanes per row
  static constexpr uint32_t RPI = 64u / LPR;          // rows per DMA instruction
};

struct LdsLayout {
  uint32_t tile, list, hist, rowmap, hsite, carry, synhist, total;
};

PFA_HD inline uint32_t round16(uint32_t x) { return (x + 15u) & ~15u; }

PFA_HD inline LdsLayout lds_layout(uint32_t lb, uint32_t n_max, bool cds, bool rowmap) {
  const uint32_t ww = 64u * lb, rpi = 1024u / ww;
  const uint32_t n = n_max ? n_max : 1u;
  LdsLayout o;
  o.tile = 0;
  (void)rpi;
  o.list = round16(n * ww);  // exactly N rows: lanes of the last DMA instruction beyond row N-1 are masked off
  o.hist = o.list + 2u * ww;   // list: the tile's variable sites, ww u16 entries (site | XOR byte << 8)
  o.rowmap = o.hist + round16((n / 2u) * 4u + 4u);
  // hsite[s]: H and allele mask of variable site s (written by tier 2; entries of other sites are stale
  // and never used): 16 bits (H < 4096 | mask << 12) when no unit has more than 64 rows, else 32
  // (H | mask << 24); hsite[-2], hsite[-1]: the last two sites of the previous tile
  o.hsite = o.rowmap + (rowmap ? round16(n * 4u) : 0u) + (cds ? 16u : 0u);
  o.carry = o.hsite + (cds ? ww * (n <= 64u ? 2u : 4u) : 0u);
  o.synhist = o.carry + (cds ? round16(2u * n) : 0u);   // carry: last two columns of the previous tile, per row
  o.total = o.synhist + (cds ? round16(PFA_SYN_BINS * 4u) : 0u);
  return o;
}

struct ScanArgs {
  const uint8_t* bases;
  const pfa_desc* descs;
  const uint32_t* row_idx;
  pfa_result* results;
  uint32_t* sfs;
  uint32_t* syn_num;
  unsigned long long* var_bm;
  unsigned long long* syn_bm;
  unsigned long long* non_bm;
  uint32_t n_desc;
  uint32_t tiles_per_chunk;
  uint32_t n_max;
  uint32_t use_rowmap; // 1 when some unit carries a row-index list
  uint32_t rows_per_chunk;  // tall kernel: rows staged per pass (multiple of 4, 8..64)
  uint32_t max_slots;       // tall kernel: count-table slots per tile (variable sites beyond take the column path)
  uint32_t hist_bins;       // tall kernel: SFS bins kept in LDS (higher bins go straight to global atomics)
};

struct TallLayout {
  uint32_t tile, list, cnt, hist, hsite, synhist, cslots, total;
};

PFA_HD inline TallLayout tall_layout(uint32_t rows_per_chunk, uint32_t max_slots, uint32_t hist_bins, bool cds) {
  TallLayout o;
  o.tile = 0;                                   // R x 256 bytes
  o.list = round16(rows_per_chunk * 256u);      // list: per slot  x | d << 8 | mode << 16 | site << 24
  o.cnt = o.list + max_slots * 4u;              // cnt: six u32 counters per slot (two-allele mode uses two)
  // Once the slots of a tile have been turned into H and SFS, list + cnt are reused for the
  // column path: its site list (256 B at `list`) and its 256-bin histogram (1 KiB behind it).
  const uint32_t slots_bytes = max_slots * 28u;
  o.hist = o.list + (slots_bytes > 1280u ? slots_bytes : 1280u);
  // coding plans: hsite[s] = H | allele mask << 24 of variable site s, with two entries in front for
  // the last two sites of the previous tile, and the bins of synonymous-site numerators
  o.hsite = o.hist + round16(hist_bins * 4u + 4u) + (cds ? 16u : 0u);
  o.synhist = o.hsite + (cds ? 256u * 4u : 0u);
  // codon slots: the codon sets of up to 8 joint columns of the tile, tracked chunk by chunk
  o.cslots = o.synhist + (cds ? round16(PFA_SYN_BINS * 4u) : 0u);
  o.total = o.cslots + (cds ? 32u + 64u : 0u);
  return o;
}

// ---------------------------------------------------------------- device helpers
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ inline uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}
__device__ inline uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over the 4 lanes of a quad with DPP quad_perm moves (VALU only; __shfl_xor would go
// through ds_bpermute and the LDS queue)
__device__ inline uint32_t quad_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  return v;
}
// Workgroups are ONE wavefront: "synchronising" it only means ordering its LDS traffic (lanes
// exchange data through LDS).  __syncthreads() would also wait for every global load in flight
// (s_waitcnt vmcnt(0)), i.e. for the tile that was just requested ahead of time.
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// value of the previous lane (lane 0: 0): one DPP move instead of a ds_bpermute
__device__ inline uint32_t wave_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);   // wave_shr:1
}
// OR over the 64 lanes with DPP moves only; the result is wave-uniform (an SGPR pair)
__device__ inline uint32_t wave_or_uniform(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline unsigned long long wave_or_uniform(unsigned long long v) {
  return (unsigned long long)wave_or_uniform((uint32_t)v) | (unsigned long long)wave_or_uniform((uint32_t)(v >> 32)) << 32;
}
__device__ inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// Wave-uniform form of classify_sense_mask (pfa_codon.h; PolyFastA.py:331-434) for a set `ms`
// held in scalar registers: lane c stands for codon c (`my_info` = annot_info[c], `my_same` =
// same_annot[c]), so the multiplicity rule for three or more codons needs no loop — a codon's
// label multiplicity is one popcount, and "some label occurs at least t times" is one ballot.
__device__ inline uint32_t classify_uniform(unsigned long long ms, uint32_t lane, uint32_t my_info,
                                            unsigned long long my_same) {
  const int k = __popcll(ms);
  if (k < 2) return 0u;
  if (k == 2) {
    const int c1 = __builtin_ctzll(ms);
    const int c2 = __builtin_ctzll(ms & (ms - 1ull));
    const uint32_t x = uint32_t(c1 ^ c2);
    const bool d0 = (x & 0x30u) != 0u, d1 = (x & 0x0Cu) != 0u, d2 = (x & 0x03u) != 0u;
    if (int(d0) + int(d1) + int(d2) == 1) {                       // PolyFastA.py:349-358
      const uint32_t i1 = (uint32_t)__builtin_amdgcn_readlane((int)my_info, c1);
      const uint32_t i2 = (uint32_t)__builtin_amdgcn_readlane((int)my_info, c2);
      const bool syn = i1 == i2 || (d0 && (i1 & i2 & 0x60u) != 0u);
      const uint32_t bit = d0 ? 1u : d1 ? 2u : 4u;
      return syn ? bit : bit << 3;
    }
    return d_tables.pair[c1 * 64 + c2];                           // PolyFastA.py:360-414
  }
  // three or more distinct sense codons                          PolyFastA.py:415-432
  uint32_t nb0 = 0, nb1 = 0, nb2 = 0;
#pragma unroll
  for (uint32_t e = 0; e < 4u; ++e) {
    nb0 += (ms & (0xFFFFull << (16u * e))) != 0ull;
    nb1 += (ms & (0x000F000F000F000Full << (4u * e))) != 0ull;
    nb2 += (ms & (0x1111111111111111ull << e)) != 0ull;
  }
  const uint32_t var = (nb0 > 1u ? 1u : 0u) | (nb1 > 1u ? 2u : 0u) | (nb2 > 1u ? 4u : 0u);
  const bool present = (ms >> lane) & 1ull;
  const uint32_t mult = present ? (uint32_t)__popcll(ms & my_same) : 0u;
  if (__ballot(mult >= 2u) == 0ull) return var << 3;              // all labels distinct: PolyFastA.py:430-432
  const uint32_t last = (var & 4u) ? 2u : (var & 2u) ? 1u : 0u;
  const uint32_t nb_last = last == 2u ? nb2 : last == 1u ? nb1 : nb0;
  const uint32_t s = __ballot(mult >= nb_last) != 0ull ? (1u << last) : 0u;   // PolyFastA.py:426-427
  const uint32_t n = var & ((1u << last) - 1u);                   // PolyFastA.py:428-429
  return s | n << 3;
}


// ---------------------------------------------------------------- the coding pass
// Everything the syn / nonsyn split does once the allele scan of a tile is complete
// (PolyFastA.py:284-315, 319-434), shared by scan_kernel and scan_tall_kernel.  Tiles are 256
// sites wide and line-aligned; 256 = 1 (mod 3), so the first codon of tile t starts
// t mod 3 sites before it: the tier-1 words of the previous tile's last lane are carried in
// registers, and chunks of coding plans hold a multiple of 3 tiles, so a chunk always starts
// on a codon boundary.  What a kernel supplies: the tile's tier-1 words, a reader of per-site
// H | allele mask << 24 (variable sites only), and a reader of the codon set of one column.
template <bool LEAN>
struct CodonPass {
  static constexpr uint32_t WW = 64u * LB, NIT = 2u;
  // wave constants: lane c stands for codon c in the tables below
  uint32_t lane;
  uint32_t my_info;               // annot_info[c] (classify_uniform)
  unsigned long long my_same;     // same_annot[c]
  uint32_t my_cinfo;              // synonymous-neighbour count | stop << 7, read with ds_bpermute
  uint32_t my_slut[3];            // the three 256-byte planes of SingleLut, one word per lane each
  uint32_t* synhist;              // LDS: PFA_SYN_BINS bins of synonymous-site numerators by codon count
  unsigned long long* syn_bm;     // optional per-site S / N bitmaps
  unsigned long long* non_bm;
  uint32_t tile_off;
  // per-lane partial sums of the chunk
  uint32_t acc_segS = 0u, acc_segN = 0u, acc_stops = 0u, acc_missing = 0u, acc_syn1 = 0u;
  unsigned long long acc_HS = 0ull, acc_HN = 0ull;
  uint32_t carry_va = 0u, carry_vd = 0u;   // tier-1 AND / XOR words of the previous tile's last lane
  // per tile: the codon columns that END at one of this lane's sites
  uint32_t xw[2], dw[2];          // 3 consensus bytes, 3 XOR bytes (0 = monomorphic) per candidate column
  int endo[2];                    // tile-relative site where it ends, or -1
  uint64_t mjoint[2];             // ballots: columns with two or three variable positions
  bool single[2], mono[2];
  uint32_t pv[2];                 // the variable position of a `single` column
  uint32_t site0;

  __device__ void init(uint32_t lane_, uint32_t* synhist_, const ScanArgs& A, const pfa_desc& D) {
    lane = lane_;
    synhist = synhist_;
    syn_bm = A.syn_bm;
    non_bm = A.non_bm;
    tile_off = D.tile_off;
    my_info = d_tables.annot_info[lane];
    my_same = d_tables.same_annot[lane];
    my_cinfo = (uint32_t)d_tables.syn3[lane] | (uint32_t)((d_tables.stop_mask >> lane) & 1ull) << 7;
#pragma unroll
    for (uint32_t pl = 0; pl < 3u; ++pl) my_slut[pl] = reinterpret_cast<const uint32_t*>(d_single_lut.v)[pl * 64u + lane];
  }

  // table entry of another lane (the LDS crossbar, no LDS memory); every lane must be enabled
  __device__ static uint32_t lane_lookup(uint32_t from_lane, uint32_t mine) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)mine);
  }

  __device__ void mark_site(bool is_s, int rel_site) const {
    const uint32_t site = site0 + (uint32_t)rel_site;   // rel_site may be -2..-1: previous tile
    const uint64_t w = (uint64_t)(tile_off + site / WW) * LB + (site % WW) % LB;
    atomicOr(&(is_s ? syn_bm : non_bm)[w], 1ull << ((site % WW) / LB));
  }

  // Register window: every lane looks at the codon columns that END at one of its own sites.
  // The three bytes (tier-1 AND word = the consensus byte of a monomorphic site) and the three
  // "varies" flags come from this lane's and the previous lane's tier-1 words (lane 0: the words
  // carried over from the previous tile): no memory traffic for the 87 % of columns that do not vary.
  __device__ void window(uint32_t va, uint32_t vd, uint32_t site0_, uint32_t lim) {
    site0 = site0_;
    uint32_t pa = wave_shr1(va), pd = wave_shr1(vd);
    if (lane == 0u) { pa = carry_va; pd = carry_vd; }
    carry_va = __builtin_amdgcn_readlane(va, 63);
    carry_vd = __builtin_amdgcn_readlane(vd, 63);
    const unsigned long long Wa = (unsigned long long)pa | (unsigned long long)va << 32;
    const unsigned long long Wd = (unsigned long long)pd | (unsigned long long)vd << 32;
    const uint32_t o1 = (5u - (site0 + 4u * lane) % 3u) % 3u;   // first own site that ends a codon
#pragma unroll
    for (uint32_t it = 0; it < 2u; ++it) {
      const uint32_t o = o1 + 3u * it;
      const bool has = o < 4u && 4u * lane + o < lim;
      xw[it] = (uint32_t)(Wa >> (8u * (2u + (o & 3u)))) & 0xFFFFFFu;
      dw[it] = (uint32_t)(Wd >> (8u * (2u + (o & 3u)))) & 0xFFFFFFu;
      endo[it] = has ? (int)(4u * lane + o) : -1;
    }
    mjoint[0] = mjoint[1] = 0ull;
    single[0] = single[1] = mono[0] = mono[1] = false;
    pv[0] = pv[1] = 0u;
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      bool joint = false;
      if (endo[it] >= 0) {
        const uint32_t dv = dw[it];
        const bool v0 = (dv & 0xFFu) != 0u, v1 = (dv & 0xFF00u) != 0u, v2 = (dv & 0xFF0000u) != 0u;
        const uint32_t nvar = (uint32_t)v0 + (uint32_t)v1 + (uint32_t)v2;
        joint = nvar >= 2u;                                // rare: the codons must be read row by row
        single[it] = nvar == 1u;                           // one variable position: resolved from its allele mask
        mono[it] = nvar == 0u;
        pv[it] = v0 ? 0u : v1 ? 1u : 2u;
      }
      mjoint[it] = __ballot(joint);
    }
  }

  // One column with two or three variable positions, given the wave-uniform set `M` of its
  // distinct clean codons: stop / synonymous-site bookkeeping and the classifier
  // (PolyFastA.py:301-315,319-434).  `hget(rel)`: H | allele mask << 24 of a variable site.
  template <class HGet>
  __device__ void joint_finish(unsigned long long M, int rel, HGet hget) {
    const uint32_t k = (uint32_t)__popcll(M);
    if (k == 0u) { acc_missing += lane == 0u ? 1u : 0u; return; }
    constexpr unsigned long long kStop = kHostTables.stop_mask;
    const uint32_t t3 = (uint32_t)__popcll(M & kHostTables.t_mask[1]) + 2u * (uint32_t)__popcll(M & kHostTables.t_mask[2]) +
                        3u * (uint32_t)__popcll(M & kHostTables.t_mask[3]) + 4u * (uint32_t)__popcll(M & kHostTables.t_mask[4]);
    const bool stop = (M & kStop) != 0ull;
    const uint32_t labels = classify_uniform(M & ~kStop, lane, my_info, my_same);
    if (lane == 0u) {
      if (k == 1u) acc_syn1 += t3;
      else atomicAdd(&synhist[k], t3);
      acc_stops += stop ? 1u : 0u;
#pragma unroll
      for (uint32_t p = 0; p < 3u; ++p) {
        const bool is_s = (labels >> p) & 1u, is_n = (labels >> (3u + p)) & 1u;
        if (is_s || is_n) {
          const uint32_t hs = hget(rel + (int)p) & 0xFFFFFFu;
          if (is_s) { acc_segS += 1u; acc_HS += hs; }
          else { acc_segN += 1u; acc_HN += hs; }
          if (!LEAN && syn_bm != nullptr) mark_site(is_s, rel + (int)p);
        }
      }
    }
  }

  // Columns with two or three variable positions (~0.6 %), one at a time (rows in LDS):
  // `read_set(rel)` returns this lane's share of the column's codon-presence set (the kernel reads
  // one row per lane); the sets are OR-reduced with DPP moves.
  template <class ReadSet, class HGet>
  __device__ void joint_columns(ReadSet read_set, HGet hget) {
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      for (uint64_t m = mjoint[it]; m != 0ull; m &= m - 1ull) {
        const uint32_t owner = (uint32_t)__builtin_ctzll(m);
        const int rel = __builtin_amdgcn_readlane(endo[it], owner) - 2;
        joint_finish(wave_or_uniform(read_set(rel)), rel, hget);
      }
    }
  }

  // The same, up to four columns per call of `read_sets(rels, n, Mv)` (rows in global memory: one
  // round trip for all of them).
  template <class ReadSets, class HGet>
  __device__ void joint_columns_batched(ReadSets read_sets, HGet hget) {
    int rels[4] = {0, 0, 0, 0};
    uint32_t nj = 0u;   // wave-uniform: the slots are filled by scalar code
    auto flush_batch = [&]() {
      unsigned long long Mv[4] = {0ull, 0ull, 0ull, 0ull};
      read_sets(rels, nj, Mv);
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c)
        if (c < nj) joint_finish(wave_or_uniform(Mv[c]), rels[c], hget);
      nj = 0u;
    };
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      for (uint64_t m = mjoint[it]; m != 0ull; m &= m - 1ull) {
        const uint32_t owner = (uint32_t)__builtin_ctzll(m);
        const int rel = __builtin_amdgcn_readlane(endo[it], owner) - 2;
        if (nj == 0u) rels[0] = rel; else if (nj == 1u) rels[1] = rel; else if (nj == 2u) rels[2] = rel; else rels[3] = rel;
        if (++nj == 4u) flush_batch();
      }
    }
    if (nj) flush_batch();
  }

  // Monomorphic columns and columns with ONE variable position: registers and tables only, so a
  // kernel calls this after it has requested its next tile.
  template <class HGet>
  __device__ void mono_and_single_columns(HGet hget) {
    // monomorphic: exactly-A/C/G/T test of the three consensus bytes at once (the canonical byte
    // of the 2-bit code (x >> 1) & 3 is 0x41 + 2 code + 0x0F [code == 2]; a byte is clean iff it
    // equals it), then one table read: synonymous weight | stop << 7.  Every lane computes and
    // looks up, so that the ds_bpermute runs with all lanes enabled (a disabled lane would not
    // supply its table entry).
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      const uint32_t w = xw[it];
      const uint32_t t2 = w & 0x060606u;
      const uint32_t e2 = ((w >> 2) & 0x010101u) & ~(w >> 1);
      const uint32_t bad = (0x414141u + t2 + (e2 << 4) - e2) ^ w;
      const uint32_t tt = t2 >> 1;
      const uint32_t ci = lane_lookup(((tt & 3u) << 4) | ((tt >> 6) & 0xCu) | ((tt >> 16) & 3u), my_cinfo);
      if (mono[it]) {
        if (bad == 0u) {
          acc_syn1 += ci & 7u;
          acc_stops += ci >> 7;
        } else {
          acc_missing += 1u;
        }
      }
    }
    // One variable position, on the lane that owns the column: the codon set is the consensus
    // codon with every A/C/G/T allele of that position substituted (rows with another symbol
    // there hold an unclean codon and drop out, PolyFastA.py:301).  Only that position can be
    // labelled.  The two fixed bytes are the lane's consensus bytes; the allele mask and H of
    // the variable site come from the allele scan.  All lanes run it, `valid` marks the lanes
    // that own such a column (the table reads are ds_bpermute's and need every lane enabled).
    auto single_col = [&](bool valid, uint32_t w, uint32_t p, int end_rel) {
      const int rel = valid ? end_rel - 2 : 0;
      const uint32_t hv = hget(rel + (int)p);
      const uint32_t am = valid ? hv >> 24 : 0u;
      // exactly-A/C/G/T test of the two fixed bytes (as above; the byte of position p is meaningless)
      const uint32_t t2 = w & 0x060606u;
      const uint32_t e2 = ((w >> 2) & 0x010101u) & ~(w >> 1);
      const uint32_t bad = ((0x414141u + t2 + (e2 << 4) - e2) ^ w) & ~(0xFFu << (8u * p));
      const uint32_t k = bad ? 0u : (uint32_t)__popc(am);   // distinct clean codons
      const uint32_t tt = t2 >> 1;
      const uint32_t c0 = tt & 3u, c1 = (tt >> 8) & 3u, c2 = (tt >> 16) & 3u;
      const uint32_t b4 = p == 0u ? (c1 << 2 | c2) : p == 1u ? (c0 << 2 | c2) : (c0 << 2 | c1);
      const uint32_t j = b4 * 16u + (am & 15u);             // entry of plane p
      const uint32_t w0 = lane_lookup(j >> 2, my_slut[0]), w1 = lane_lookup(j >> 2, my_slut[1]), w2 = lane_lookup(j >> 2, my_slut[2]);
      const uint32_t v = ((p == 0u ? w0 : p == 1u ? w1 : w2) >> (8u * (j & 3u))) & 0xFFu;
      if (!valid) return;
      if (k == 0u) { acc_missing += 1u; return; }
      const uint32_t t3 = v & 0x1Fu;
      if (k == 1u) acc_syn1 += t3;
      else atomicAdd(&synhist[k], t3);
      acc_stops += (v >> 5) & 1u;
      const bool is_s = (v & 0x40u) != 0u, is_n = (v & 0x80u) != 0u;
      const uint32_t hs = hv & 0xFFFFFFu;
      if (is_s) { acc_segS += 1u; acc_HS += hs; }
      if (is_n) { acc_segN += 1u; acc_HN += hs; }
      if (!LEAN && (is_s || is_n) && syn_bm != nullptr) mark_site(is_s, rel + (int)p);
    };
    const bool any1 = single[0] || single[1];
    if (__any(any1)) {
      const uint32_t sel = single[0] ? 0u : 1u;
      single_col(any1, sel ? xw[1] : xw[0], (sel ? pv[1] : pv[0]) % 3u, sel ? endo[1] : endo[0]);
    }
    const bool both = single[0] && single[1];            // a lane owning two such columns: rare
    if (__any(both)) single_col(both, xw[1], pv[1] % 3u, endo[1]);
  }

  // the trailing 1-2 sites of an alignment whose length is not a multiple of 3 never form a
  // clean codon (PolyFastA.py:301-305)
  __device__ void last_tile(uint32_t L) {
    if ((L % 3u) != 0u && lane == 0u) acc_missing += 1u;
  }

  __device__ void flush(pfa_result* R, uint32_t* syn_num_unit) {
    acc_segS = wave_sum(acc_segS);
    acc_segN = wave_sum(acc_segN);
    acc_stops = wave_sum(acc_stops);
    acc_missing = wave_sum(acc_missing);
    acc_syn1 = wave_sum(acc_syn1);
    acc_HS = wave_sum(acc_HS);
    acc_HN = wave_sum(acc_HN);
    if (lane == 0u) {
      if (acc_segS) atomicAdd(&R->seg_S, acc_segS);
      if (acc_segN) atomicAdd(&R->seg_N, acc_segN);
      if (acc_stops) atomicAdd(&R->nstops, acc_stops);
      if (acc_missing) atomicAdd(&R->missing_cols, acc_missing);
      if (acc_HS) atomicAdd(reinterpret_cast<unsigned long long*>(&R->H_S), acc_HS);
      if (acc_HN) atomicAdd(reinterpret_cast<unsigned long long*>(&R->H_N), acc_HN);
      if (acc_syn1) atomicAdd(&syn_num_unit[1], acc_syn1);
    }
    wave_lds_sync();
    for (uint32_t i = lane; i < PFA_SYN_BINS; i += 64u) {
      const uint32_t v = synhist[i];
      if (v) atomicAdd(&syn_num_unit[i], v);
    }
  }
};

// codon of three bytes packed in a word (byte 0 = first base): its index as a one-hot 64-bit set,
// or 0 when any byte is not exactly A/C/G/T (PolyFastA.py:301)
__device__ inline unsigned long long codon_bit(uint32_t x0, uint32_t x1, uint32_t x2) {
  const bool ok = base_clean(x0) && base_clean(x1) && base_clean(x2);
  return ok ? 1ull << (base_code(x0) * 16u + base_code(x1) * 4u + base_code(x2)) : 0ull;
}

// ---------------------------------------------------------------- the kernel
// LEAN: the plan has no row-index list and wants no per-site bitmap (the batch path of the CLI
// and the bench): those branches, and the scalar registers they pin, are compiled out.
template <int STAGE, bool HAS_CDS, bool LEAN>
__global__ void __launch_bounds__(64) scan_kernel(const ScanArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const uint32_t lane = threadIdx.x;
  const pfa_desc D = A.descs[blockIdx.x];
  const bool cds = HAS_CDS && (D.flags & PFA_FLAG_CDS) != 0u;
  constexpr uint32_t TW = G::WW;
  const uint32_t N = D.n_rows, L = D.n_sites;
  if (N == 0u || L == 0u) return;
  const uint32_t n_tiles = (L + TW - 1u) / TW;
  const uint32_t t0 = blockIdx.y * A.tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = umin(t0 + A.tiles_per_chunk, n_tiles);

  const LdsLayout lo = lds_layout(LB, A.n_max, HAS_CDS, A.use_rowmap != 0u);
  uint8_t* const tile8 = lds + lo.tile;
  uint16_t* const list = reinterpret_cast<uint16_t*>(lds + lo.list);
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + lo.hist);
  uint32_t* const rowmap = reinterpret_cast<uint32_t*>(lds + lo.rowmap);
  uint32_t* const hsite32 = reinterpret_cast<uint32_t*>(lds + lo.hsite);
  uint16_t* const hsite16 = reinterpret_cast<uint16_t*>(lds + lo.hsite);
  const bool h16 = A.n_max <= 64u;
  // H and allele mask of a variable site (tile-relative index, -2 and -1: previous tile), widened to H | mask << 24
  auto hsite_get = [&](int rel) -> uint32_t {
    if (h16) { const uint32_t v = hsite16[rel]; return (v & 0xFFFu) | (v >> 12) << 24; }
    return hsite32[rel];
  };
  uint8_t* const carry = lds + lo.carry;
  uint32_t* const synhist = reinterpret_cast<uint32_t*>(lds + lo.synhist);

  const uint32_t nbins = N / 2u;
  const bool mapped = !LEAN && D.row_idx_off != PFA_NO_ROW_IDX;   // population subset: rows through an index list
  if (mapped)
    // the row's byte offset in the matrix, in 16-byte units (plans keep matrices below 64 GB)
    for (uint32_t i = lane; i < N; i += 64u) rowmap[i] = A.row_idx[D.row_idx_off + i] * (D.row_stride >> 4);
  for (uint32_t i = lane; i < nbins; i += 64u) hist[i] = 0u;
  if (cds)
  {
    for (uint32_t i = lane; i < PFA_SYN_BINS; i += 64u) synhist[i] = 0u;
  }
  wave_lds_sync();

  // per-lane partial sums of this chunk
  uint32_t acc_S = 0u, acc_bad = 0u;
  unsigned long long acc_H = 0ull;
  CodonPass<LEAN> cp;
  if constexpr (HAS_CDS) cp.init(lane, synhist, A, D);

  const uint8_t* const mat = A.bases + D.base_off;
  const uint32_t n_inst = (N + G::RPI - 1u) / G::RPI;
  const uint32_t NN = N * N;

  // ---------------- stage a column tile into LDS (asynchronous for the LDS-DMA variants)
  auto stage = [&](uint32_t tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t c = lane % G::LPR, rr = lane / G::LPR;
    const uint32_t col = umin(site0 + 16u * c, D.row_stride - 16u);
    if constexpr (STAGE != 1) {
      // STAGE 0: default cache policy; STAGE 2: non-temporal (aux = 2), the bases are read once
      if (!mapped) {
        // a running per-lane pointer; only the last instruction can be partial (EXEC-masked lanes
        // move nothing and leave their LDS slot untouched)
        const uint8_t* src = mat + (uint64_t)rr * D.row_stride + col;
        const uint64_t step = (uint64_t)G::RPI * D.row_stride;
        const uint32_t n_full = N / G::RPI;
        uint32_t g = 0;
        for (; g < n_full; ++g, src += step)
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
        if (rr < N % G::RPI)
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
      } else {
        const uint8_t* const base = mat + col;
        const uint32_t n_full = N / G::RPI;
        uint32_t g = 0;
        for (; g < n_full; ++g)
          __builtin_amdgcn_global_load_lds((gptr_t)(base + ((uint64_t)rowmap[g * G::RPI + rr] << 4)),
                                           (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
        if (rr < N % G::RPI)
          __builtin_amdgcn_global_load_lds((gptr_t)(base + ((uint64_t)rowmap[g * G::RPI + rr] << 4)),
                                           (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
      }
    } else {
#pragma unroll 4
      for (uint32_t g = 0; g < n_inst; ++g) {
        const uint32_t r = g * G::RPI + rr;
        if (r < N) {
          const uint4 v = *reinterpret_cast<const uint4*>(mat + (mapped ? (uint64_t)rowmap[r] << 4 : (uint64_t)r * D.row_stride) + col);
          *reinterpret_cast<uint4*>(tile8 + g * 1024u + lane * 16u) = v;
        }
      }
    }
  };
  // The tile of iteration t+1 is requested as soon as iteration t has read the last byte it needs
  // from the tile buffer; what is left of iteration t by then (register / table work of the coding
  // pass) runs under the loads.
  stage(t0);

  for (uint32_t tile = t0; tile < t1; ++tile) {
    const uint32_t site0 = tile * TW;
    const uint32_t lim = umin(TW, L - site0);  // valid sites of this tile
    if constexpr (STAGE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_sync();

    // ---------------- tier 1: running AND / OR down the rows
    uint32_t va = 0xFFFFFFFFu, vo = 0u;
    {
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile8);
#pragma unroll 8
      for (uint32_t r = 0; r < N; ++r) {
        const uint32_t x = t32[r * 64u + lane];
        va &= x;
        vo |= x;
      }
    }
    const uint32_t vd = va ^ vo;

    // ---------------- compact the variable sites of the tile
    uint32_t P = 0u;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)LB; ++b) {
      const uint32_t off = LB * lane + b;
      const bool pb = ((vd >> (8u * b)) & 0xFFu) != 0u && off < lim;
      const uint64_t m = __ballot(pb);
      if (pb) list[P + mbcnt64(m)] = uint16_t(off | ((vd >> (8u * b)) & 0xFFu) << 8);
      if (!LEAN && A.var_bm != nullptr && lane == 0u)
        A.var_bm[(uint64_t)(D.tile_off + tile) * LB + b] = m;
      P += (uint32_t)__popcll(m);
    }
    P = __builtin_amdgcn_readfirstlane(P);
    acc_S += (lane == 0u) ? P : 0u;
    wave_lds_sync();

    // ---------------- tier 2: exact allele counts of the variable sites
    // 4 lanes per site walk its column.  Fast path: x = the byte of row 0, y = x ^ (XOR byte of
    // the site); when every row holds x or y the site has exactly these two alleles (2 compares
    // per byte).  Otherwise (3+ alleles, ~1 % of variable sites) the column is walked again
    // with six counters, and any byte outside A C G T - N takes the exact generic path.
    const uint32_t slot = lane >> 2, ph = lane & 3u;
    for (uint32_t p0 = 0; p0 < P; p0 += 16u) {
      const uint32_t idx = p0 + slot;
      const bool act = idx < P;
      const uint32_t ent = list[act ? idx : 0u];
      const uint32_t s = ent & 0xFFu;
      const uint8_t* colp = tile8 + s;
      const uint32_t x = colp[0], y = x ^ (ent >> 8);
      uint32_t c1 = 0, c2 = 0;
#pragma unroll 4
      for (uint32_t r = ph; r < N; r += 4u) {
        const uint32_t v = colp[r * G::WW];
        c1 += (v == x);
        c2 += (v == y);
      }
      c1 = quad_sum(c1);
      c2 = quad_sum(c2);
      const bool xc = base_clean(x), yc = base_clean(y);
      uint32_t sumsq = c1 * c1 + c2 * c2;
      uint32_t second = (xc && yc) ? umin(c1, c2) : 0u;   // second-largest A/C/G/T count (PolyFastA.py:279-281)
      uint32_t pres = (xc ? 1u << base_code(x) : 0u) | (yc ? 1u << base_code(y) : 0u);   // which of A, C, T, G occur
      const bool multi = act && c1 + c2 != N;   // the same in all 4 lanes of the quad
      if (__any(multi)) {
        if (multi) {
          uint32_t cA = 0, cC = 0, cG = 0, cT = 0, cGap = 0, cNn = 0;
#pragma unroll 4
          for (uint32_t r = ph; r < N; r += 4u) {
            const uint32_t v = colp[r * G::WW];
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
          sumsq = cA * cA + cC * cC + cG * cG + cT * cT + cGap * cGap + cNn * cNn;
          const uint32_t n_other = N - (cA + cC + cG + cT + cGap + cNn);
          if (n_other != 0u) {
            // exact path for any other byte value: sum over such rows i of #{j : x_j == x_i}
            uint32_t extra = 0u;
            for (uint32_t i = ph; i < N; i += 4u) {
              const uint32_t xi = colp[i * G::WW];
              const bool six = xi == 'A' || xi == 'C' || xi == 'G' || xi == 'T' || xi == '-' || xi == 'N';
              if (!six) {
                uint32_t cnt = 0u;
                for (uint32_t j = 0; j < N; ++j) cnt += (colp[j * G::WW] == xi);
                extra += cnt;
              }
            }
            sumsq += quad_sum(extra);
          }
          const uint32_t hi1 = umax(cA, cC), lo1 = umin(cA, cC), hi2 = umax(cG, cT), lo2 = umin(cG, cT);
          second = umax(umin(hi1, hi2), umax(lo1, lo2));
          pres = (cA ? 1u : 0u) | (cC ? 2u : 0u) | (cT ? 4u : 0u) | (cG ? 8u : 0u);
        }
      }
      if (act && ph == 0u) {
        const uint32_t hs = NN - sumsq;
        acc_H += hs;
        if (cds) {   // H of the site + which of A, C, T, G occur there (codon base order)
          if (h16) hsite16[s] = uint16_t(hs | pres << 12);
          else hsite32[s] = hs | pres << 24;
        }
        if (second == 0u) acc_bad += 1u;
        else atomicAdd(&hist[second - 1u], 1u);
      }
    }

    // ---------------- coding sequences: the codon columns that END in this tile (CodonPass)
    if (cds) {
      wave_lds_sync();   // tier 2's hsite entries are visible
      cp.window(va, vd, site0, lim);
      // columns with two or three variable positions: one row per lane out of the LDS tile; the
      // one or two sites a codon reaches back into the previous tile come from `carry`
      auto cell = [&](uint32_t r, int rel) -> uint32_t {
        const uint8_t* q = rel < 0 ? carry + 2u * r + (uint32_t)(2 + rel) : tile8 + r * G::WW + (uint32_t)rel;
        return *q;
      };
      cp.joint_columns(
          [&](int rel) {
            unsigned long long Mv = 0ull;
            for (uint32_t r = lane; r < N; r += 64u) Mv |= codon_bit(cell(r, rel), cell(r, rel + 1), cell(r, rel + 2));
            return Mv;
          },
          hsite_get);
      // carry the last two columns (bytes of every row) into the next tile
      for (uint32_t i = lane; i < N; i += 64u) {
        carry[2u * i] = tile8[i * G::WW + G::WW - 2u];
        carry[2u * i + 1u] = tile8[i * G::WW + G::WW - 1u];
      }
      // ---- nothing below reads the tile buffer: request the next tile now
      wave_lds_sync();
      if (tile + 1u < t1) stage(tile + 1u);
      cp.mono_and_single_columns(hsite_get);
      if (tile + 1u == n_tiles) cp.last_tile(L);
      // H of the last two sites, for the codons that straddle into the next tile
      if (lane < 2u) {
        if (h16) hsite16[(int)lane - 2] = hsite16[G::WW - 2u + lane];
        else hsite32[(int)lane - 2] = hsite32[G::WW - 2u + lane];
      }
    } else {
      wave_lds_sync();   // every read of the tile buffer is done: request the next tile
      if (tile + 1u < t1) stage(tile + 1u);
    }
  }

  // ---------------- flush the chunk's partial sums
  pfa_result* R = A.results + blockIdx.x;
  acc_S = wave_sum(acc_S);
  acc_bad = wave_sum(acc_bad);
  acc_H = wave_sum(acc_H);
  if (lane == 0u) {
    if (acc_S) atomicAdd(&R->seg_sites, acc_S);
    if (acc_bad) atomicAdd(&R->sfs_bad, acc_bad);
    if (acc_H) atomicAdd(reinterpret_cast<unsigned long long*>(&R->H), acc_H);
  }
  wave_lds_sync();
  for (uint32_t i = lane; i < nbins; i += 64u) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&A.sfs[D.sfs_off + i], v);
  }
  if (cds) cp.flush(R, A.syn_num + (uint64_t)blockIdx.x * PFA_SYN_BINS);
}


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

template <int STAGE, bool HAS_CDS, bool LEAN>
__global__ void __launch_bounds__(64) scan_tall_kernel(const ScanArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr uint32_t TW = 256u;
  const uint32_t lane = threadIdx.x;
  const pfa_desc D = A.descs[blockIdx.x];
  const uint32_t N = D.n_rows, L = D.n_sites;
  if (N == 0u || L == 0u) return;
  const uint32_t n_tiles = (L + TW - 1u) / TW;
  const uint32_t t0 = blockIdx.y * A.tiles_per_chunk;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = umin(t0 + A.tiles_per_chunk, n_tiles);

  const uint32_t R = A.rows_per_chunk, MAXS = A.max_slots;
  const TallLayout lo = tall_layout(R, MAXS, A.hist_bins, HAS_CDS);
  uint8_t* const tile8 = lds + lo.tile;
  uint32_t* const list = reinterpret_cast<uint32_t*>(lds + lo.list);
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds + lo.cnt);
  uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + lo.hist);
  uint8_t* const xlist = lds + lo.list;                                      // end of tile: sites for the column path
  uint32_t* const colhist = reinterpret_cast<uint32_t*>(lds + lo.list + 256u);   // and its 256 bins

  uint32_t* const hsite = reinterpret_cast<uint32_t*>(lds + lo.hsite);
  uint32_t* const synhist = reinterpret_cast<uint32_t*>(lds + lo.synhist);
  const bool cds = HAS_CDS && (D.flags & PFA_FLAG_CDS) != 0u;

  const uint32_t nbins = N / 2u, HB = umin(nbins, A.hist_bins);
  const bool mapped = !LEAN && D.row_idx_off != PFA_NO_ROW_IDX;   // population subset: rows through an index list
  const uint32_t* const ridx = A.row_idx + (mapped ? D.row_idx_off : 0u);
  const bool multi = N > R;
  for (uint32_t i = lane; i < HB; i += 64u) hist[i] = 0u;
  if (cds)
    for (uint32_t i = lane; i < PFA_SYN_BINS; i += 64u) synhist[i] = 0u;
  // lane r holds the matrix row of the chunk's row r; the next chunk's are fetched while this one is scanned
  uint32_t idx_cur = 0u;
  const uint32_t stride16 = D.row_stride >> 4;
  if (mapped && lane < umin(R, N)) idx_cur = ridx[lane] * stride16;
  wave_lds_sync();

  uint32_t acc_S = 0u, acc_bad = 0u;
  unsigned long long acc_H = 0ull;
  const uint8_t* const mat = A.bases + D.base_off;
  const unsigned long long NN = (unsigned long long)N * N;
  CodonPass<LEAN> cp;
  if constexpr (HAS_CDS) cp.init(lane, synhist, A, D);
  auto hsite_get = [&](int rel) -> uint32_t { return hsite[rel]; };
  // H and which of A, C, T, G occur (codon base order) of a variable site, for the coding pass
  auto hsite_put = [&](uint32_t s, unsigned long long hs, uint32_t cA, uint32_t cC, uint32_t cG, uint32_t cT) {
    hsite[s] = (uint32_t)hs | ((cA ? 1u : 0u) | (cC ? 2u : 0u) | (cT ? 4u : 0u) | (cG ? 8u : 0u)) << 24;
  };

  // second-largest A/C/G/T count (PolyFastA.py:279-281) -> LDS histogram, high bins to global
  auto sfs_add = [&](uint32_t cA, uint32_t cC, uint32_t cG, uint32_t cT) {
    const uint32_t hi1 = umax(cA, cC), lo1 = umin(cA, cC), hi2 = umax(cG, cT), lo2 = umin(cG, cT);
    const uint32_t second = umax(umin(hi1, hi2), umax(lo1, lo2));
    if (second == 0u) acc_bad += 1u;
    else if (second - 1u < HB) atomicAdd(&hist[second - 1u], 1u);
    else atomicAdd(&A.sfs[D.sfs_off + second - 1u], 1u);
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
  auto stage_chunk = [&](uint32_t tile, uint32_t r0, uint32_t idx_rows) {
    const uint32_t Rc = umin(R, N - r0);
    const uint32_t col = umin(tile * TW + 16u * c16, D.row_stride - 16u);
    const uint32_t n_inst = (Rc + 3u) / 4u;
    for (uint32_t g = 0; g < n_inst; ++g) {
      const uint32_t r = 4u * g + rr;
      // mapped: the lanes hold byte offsets of the chunk's rows in 16-byte units
      const uint32_t held = mapped ? (uint32_t)__shfl((int)idx_rows, (int)(r & 63u)) : 0u;
      if (r < Rc) {
        const uint8_t* src = mat + (mapped ? (uint64_t)held << 4 : (uint64_t)(r0 + r) * D.row_stride) + col;
        if constexpr (STAGE == 1)
          *reinterpret_cast<uint4*>(tile8 + g * 1024u + lane * 16u) = *reinterpret_cast<const uint4*>(src);
        else
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile8 + g * 1024u), 16, 0, STAGE);
      }
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
      const uint32_t o1 = (5u - (site0 + 4u * lane) % 3u) % 3u;
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
      if constexpr (STAGE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile8);
#pragma unroll 8
        for (uint32_t r = 0; r < Rc; ++r) {
          const uint32_t x = t32[r * 64u + lane];
          ca &= x;
          co |= x;
        }
      }
      const uint32_t nva = va & ca, nvo = vo | co, nvd = nva ^ nvo;

      // ---------------- sites that turn variable in this chunk get a count slot
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
      const uint32_t va_pre = va, vd_pre = vd, P_prev = umin(P, MAXS);   // state before this chunk
      va = nva;
      vo = nvo;
      vd = nvd;
      P = __builtin_amdgcn_readfirstlane(P);
      const uint32_t Pc = umin(P, MAXS);
      wave_lds_sync();

      // ---------------- coding units: columns that turn joint in this chunk get a codon slot
      if (cds) {
        uint64_t jb_now[2];
        joint_ballots(nvd, jb_now);
#pragma unroll
        for (uint32_t it = 0; it < 2u; ++it) {
          for (uint64_t m = jb_now[it] & ~jb_prev[it]; m != 0ull; m &= m - 1ull) {
            const int end = __builtin_amdgcn_readlane(endoT[it], (uint32_t)__builtin_ctzll(m));
            const int rel = end - 2;
            if (rel < 0 || ncs >= CS) continue;          // read back at the end of the tile
            // the distinct clean codons of the rows before this chunk
            const unsigned long long S = r0 == 0u ? 0ull : codons_before_chunk(lo.list, lo.cnt, P_prev, va_pre, vd_pre, (uint32_t)rel, lane);
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
#pragma unroll 4
          for (uint32_t r = ph; r < Rc; r += 4u) {
            const uint32_t v = colp[r * TW];
            c1 += (v == x);
            c2 += (v == y);
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
#pragma unroll 4
            for (uint32_t r = ph; r < Rc; r += 4u) {
              const uint32_t v = colp[r * TW];
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
        const uint32_t first = (uint32_t)(cs_end[k] - 2);
        const uint8_t* q = tile8 + lane * TW + first;
        const unsigned long long M = wave_or_uniform(lane < Rc ? codon_bit(q[0], q[1], q[2]) : 0ull);
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
      if (!LEAN && A.var_bm != nullptr && lane == 0u) A.var_bm[(uint64_t)(D.tile_off + tile) * 4u + b] = m;
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
            hsite[ent >> 24] = (uint32_t)hs | (px | py) << 24;
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

  // ---------------- flush the chunk's partial sums
  pfa_result* Rz = A.results + blockIdx.x;
  acc_S = wave_sum(acc_S);
  acc_bad = wave_sum(acc_bad);
  acc_H = wave_sum(acc_H);
  if (lane == 0u) {
    if (acc_S) atomicAdd(&Rz->seg_sites, acc_S);
    if (acc_bad) atomicAdd(&Rz->sfs_bad, acc_bad);
    if (acc_H) atomicAdd(reinterpret_cast<unsigned long long*>(&Rz->H), acc_H);
  }
  wave_lds_sync();
  for (uint32_t i = lane; i < HB; i += 64u) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&A.sfs[D.sfs_off + i], v);
  }
  if (cds) cp.flush(Rz, A.syn_num + (uint64_t)blockIdx.x * PFA_SYN_BINS);
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

// Result slab -> pinned host memory, written by the compute units.  (A hipMemcpyAsync here goes
// through the SDMA engines, whose queues the runtime creates lazily: single copies were seen to
// take ~8 ms, longer than a whole batch scan.)
__global__ void __launch_bounds__(256) slab_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// ---------------------------------------------------------------- plans
static constexpr uint32_t kLdsPerCu = 160u * 1024u;
static constexpr uint32_t kTallRows = 48u;      // its rows per chunk (upper bound)
static constexpr uint32_t kTallSlots = 64u;     // its count-table slots per tile (measured: 64 6.03, 96 5.68 TB/s at N = 100)
static constexpr int kEventRing = 64;

}  // namespace pfa

struct pfa_plan {
  uint32_t n_desc = 0;
  uint32_t lb = 4;
  uint32_t n_max = 0;
  uint32_t max_tiles = 0;
  uint32_t n_tiles_total = 0;
  uint32_t tiles_per_chunk = 1;
  uint32_t n_chunks = 1;
  uint32_t want = 0;
  bool any_cds = false;
  bool use_rowmap = false;
  bool tall = false;              // row-chunked kernel (non-coding plans with many-row units)
  uint32_t rows_per_chunk = 0, max_slots = 0, hist_bins = 0;
  uint64_t sfs_words = 0;
  uint64_t n_bases_bytes = 0;
  uint32_t lds_bytes = 0;
  pfa_desc* d_descs = nullptr;
  uint32_t* d_row_idx = nullptr;
  // results | sfs | syn_num live in ONE device slab (one memset per run) mirrored
  // by one pinned host slab (one D2H per fetch)
  uint8_t* d_slab = nullptr;
  uint8_t* h_slab = nullptr;
  uint8_t* h_slab_dev = nullptr;   // the device's address of h_slab
  size_t slab_bytes = 0, off_sfs = 0, off_syn = 0;
  pfa_result* d_results = nullptr;
  uint32_t* d_sfs = nullptr;
  uint32_t* d_syn = nullptr;
  unsigned long long* d_var_bm = nullptr;
  unsigned long long* d_syn_bm = nullptr;
  unsigned long long* d_non_bm = nullptr;
  hipEvent_t ev0[pfa::kEventRing] = {};
  hipEvent_t ev1[pfa::kEventRing] = {};
  uint64_t n_runs = 0;
};

namespace pfa {

static uint32_t tile_sites_for(uint32_t lb, bool /*cds*/) { return 64u * lb; }
// every plan uses 256-site tiles (lane = 4 sites): units with many rows stream their rows in
// chunks (scan_tall_kernel) instead of narrowing the tile
static uint32_t lb_for(uint32_t /*n_max*/, bool /*any_cds*/) { return 4u; }
static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* e = getenv(name);
  return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : dflt;
}

template <int STAGE, bool CDS, bool LEAN>
static int launch_tall(const pfa_plan* p, const ScanArgs& a) {
  auto kern = scan_tall_kernel<STAGE, CDS, LEAN>;
  static uint32_t lds_limit_set = 0;
  if (p->lds_bytes > lds_limit_set) {
    PFA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    lds_limit_set = p->lds_bytes;
  }
  dim3 grid(p->n_desc, p->n_chunks, 1), block(64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, p->lds_bytes, ctx().stream, a);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}

template <int STAGE, bool CDS, bool LEAN>
static int launch(const pfa_plan* p, const ScanArgs& a) {
  auto kern = scan_kernel<STAGE, CDS, LEAN>;
  static uint32_t lds_limit_set = 0;  // per instantiation; one process drives one device
  if (p->lds_bytes > lds_limit_set) {
    PFA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    lds_limit_set = p->lds_bytes;
  }
  dim3 grid(p->n_desc, p->n_chunks, 1), block(64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, p->lds_bytes, ctx().stream, a);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}

}  // namespace pfa

extern "C" {

int pfa_abi_version(void) { return PFA_ABI_VERSION; }

const char* pfa_last_error(void) { return pfa::g_err; }

int pfa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pfa_init(int device) {
  pfa::Context& c = pfa::ctx();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    pfa::set_error("no HIP device visible: libpfa_hip has no CPU fallback");
    return PFA_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    pfa::set_error("device %d out of range (0..%d)", device, n - 1);
    return PFA_ERR_ARG;
  }
  if (c.ready && c.device == device) return PFA_OK;
  if (c.ready) pfa_shutdown();
  PFA_HIP(hipSetDevice(device));
  // host waits spin instead of sleeping on an interrupt: a blocked wait was seen to return
  // up to 8 ms late on the GPU box, longer than a whole batch scan (hint only; ignored if the
  // runtime already fixed its flags)
  (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
  (void)hipGetLastError();
  hipDeviceProp_t prop;
  PFA_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    pfa::set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return PFA_ERR_NO_DEVICE;
  }
  c.n_cu = prop.multiProcessorCount;
  c.lds_per_block = prop.sharedMemPerBlock;
  PFA_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  c.device = device;
  c.ready = true;
  return PFA_OK;
}

void pfa_shutdown(void) {
  pfa::Context& c = pfa::ctx();
  if (!c.ready) return;
  (void)hipStreamSynchronize(c.stream);
  (void)hipStreamDestroy(c.stream);
  c = pfa::Context();
}

int pfa_sync(void) {
  PFA_NEED_INIT();
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

int pfa_dev_alloc(size_t bytes, void** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_dev_alloc: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  PFA_HIP(hipMalloc(out, bytes ? bytes : 16));
  return PFA_OK;
}

int pfa_dev_free(void* p) {
  PFA_NEED_INIT();
  if (p) PFA_HIP(hipFree(p));
  return PFA_OK;
}

int pfa_host_alloc(size_t bytes, void** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_host_alloc: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  PFA_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return PFA_OK;
}

int pfa_host_free(void* p) {
  PFA_NEED_INIT();
  if (p) PFA_HIP(hipHostFree(p));
  return PFA_OK;
}

int pfa_upload(void* dst_dev, const void* src_host, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pfa::ctx().stream));
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

int pfa_upload_async(void* dst_dev, const void* src_host, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pfa::ctx().stream));
  return PFA_OK;
}

int pfa_download(void* dst_host, const void* src_dev, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, pfa::ctx().stream));
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

uint32_t pfa_tile_sites(uint32_t max_rows, uint32_t cds) {
  return pfa::tile_sites_for(pfa::lb_for(max_rows, cds != 0), cds != 0);
}

uint32_t pfa_var_words_per_tile(uint32_t max_rows, uint32_t cds) { return pfa::lb_for(max_rows, cds != 0); }

void pfa_plan_destroy(pfa_plan* p) {
  if (!p) return;
  if (pfa::ctx().ready) (void)hipStreamSynchronize(pfa::ctx().stream);
  (void)hipFree(p->d_descs);
  (void)hipFree(p->d_row_idx);
  (void)hipFree(p->d_slab);
  if (p->h_slab) (void)hipHostFree(p->h_slab);
  (void)hipFree(p->d_var_bm);
  (void)hipFree(p->d_syn_bm);
  (void)hipFree(p->d_non_bm);
  for (int i = 0; i < pfa::kEventRing; ++i) {
    if (p->ev0[i]) (void)hipEventDestroy(p->ev0[i]);
    if (p->ev1[i]) (void)hipEventDestroy(p->ev1[i]);
  }
  delete p;
}

int pfa_plan_create(const pfa_desc* descs, uint32_t n_desc, const uint32_t* row_idx,
                    size_t n_row_idx, uint64_t n_bases_bytes, uint32_t want, pfa_plan** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_plan_create: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  if (n_desc && !descs) { pfa::set_error("pfa_plan_create: descs is NULL"); return PFA_ERR_ARG; }
  if (n_desc == 0) { pfa::set_error("pfa_plan_create: empty descriptor table"); return PFA_ERR_ARG; }

  uint32_t n_max = 0;
  bool any_cds = false;
  for (uint32_t i = 0; i < n_desc; ++i) {
    n_max = std::max(n_max, descs[i].n_rows);
    any_cds = any_cds || (descs[i].flags & PFA_FLAG_CDS);
  }
  if (any_cds && n_max > PFA_MAX_ROWS_CDS) {
    pfa::set_error("a coding scan unit has %u rows; at most %u fit one LDS column tile", n_max, PFA_MAX_ROWS_CDS);
    return PFA_ERR_TOO_MANY_ROWS;
  }
  if (n_max > PFA_MAX_ROWS) {
    pfa::set_error("a scan unit has %u rows; the limit is %u", n_max, PFA_MAX_ROWS);
    return PFA_ERR_TOO_MANY_ROWS;
  }
  const uint32_t lb = pfa::lb_for(n_max, any_cds);
  uint64_t sfs_words = 0, tiles_total = 0;
  uint32_t max_tiles = 0;
  for (uint32_t i = 0; i < n_desc; ++i) {
    const pfa_desc& d = descs[i];
    if (d.flags & ~PFA_FLAG_CDS) {
      pfa::set_error("desc %u: unknown flag bits 0x%x", i, d.flags);
      return PFA_ERR_ARG;
    }
    const uint32_t tw = pfa::tile_sites_for(lb, d.flags & PFA_FLAG_CDS);
    const uint32_t nt = (d.n_rows && d.n_sites) ? (uint32_t)(((uint64_t)d.n_sites + tw - 1) / tw) : 0u;
    sfs_words = std::max<uint64_t>(sfs_words, (uint64_t)d.sfs_off + d.n_rows / 2u);
    tiles_total = std::max<uint64_t>(tiles_total, (uint64_t)d.tile_off + nt);
    max_tiles = std::max(max_tiles, nt);
    if (nt == 0) continue;
    if (d.row_stride % 16u || d.base_off % 16u) {
      pfa::set_error("desc %u: base_off and row_stride must be multiples of 16", i);
      return PFA_ERR_ARG;
    }
    if ((uint64_t)d.row_stride < (((uint64_t)d.n_sites + 15u) & ~15ull)) {
      pfa::set_error("desc %u: row_stride %u < round_up(n_sites=%u, 16)", i, d.row_stride, d.n_sites);
      return PFA_ERR_ARG;
    }
    if (d.mat_rows == 0 || (uint64_t)d.base_off + (uint64_t)d.mat_rows * d.row_stride > n_bases_bytes) {
      pfa::set_error("desc %u: matrix [%llu, +%u x %u) exceeds the %llu-byte base buffer", i,
                     (unsigned long long)d.base_off, d.mat_rows, d.row_stride,
                     (unsigned long long)n_bases_bytes);
      return PFA_ERR_ARG;
    }
    if (d.row_idx_off == PFA_NO_ROW_IDX) {
      if (d.n_rows > d.mat_rows) {
        pfa::set_error("desc %u: n_rows %u > mat_rows %u", i, d.n_rows, d.mat_rows);
        return PFA_ERR_ARG;
      }
    } else {
      if (!row_idx || (uint64_t)d.row_idx_off + d.n_rows > n_row_idx) {
        pfa::set_error("desc %u: row_idx range out of bounds", i);
        return PFA_ERR_ARG;
      }
      if ((uint64_t)d.mat_rows * d.row_stride > (1ull << 36)) {
        pfa::set_error("desc %u: a matrix scanned through a row-index list must be smaller than 64 GB", i);
        return PFA_ERR_ARG;
      }
      for (uint32_t r = 0; r < d.n_rows; ++r)
        if (row_idx[d.row_idx_off + r] >= d.mat_rows) {
          pfa::set_error("desc %u: row index %u >= mat_rows %u", i, row_idx[d.row_idx_off + r], d.mat_rows);
          return PFA_ERR_ARG;
        }
    }
  }
  if (tiles_total > 0xFFFFFFFFull || sfs_words > 0xFFFFFFFFull) {
    pfa::set_error("plan too large (tiles or SFS words overflow 32 bits)");
    return PFA_ERR_ARG;
  }

  pfa_plan* p = new (std::nothrow) pfa_plan();
  if (!p) { pfa::set_error("out of host memory"); return PFA_ERR_NOMEM; }
  p->n_desc = n_desc;
  p->lb = lb;
  p->n_max = n_max;
  p->any_cds = any_cds;
  p->want = want;
  p->sfs_words = sfs_words;
  p->n_tiles_total = (uint32_t)tiles_total;
  p->max_tiles = max_tiles;
  p->n_bases_bytes = n_bases_bytes;
  p->use_rowmap = n_row_idx != 0;
  p->lds_bytes = pfa::lds_layout(lb, n_max, any_cds, p->use_rowmap).total;
  // Plans with many-row units run the row-chunked kernel.  Keeping all rows of a tile in LDS is
  // the faster shape while it still fills the CU: at least 5 waves whose tiles add up to 125 KB of
  // loads in flight (measured, non-coding: N = 100, 6 waves, 6.45 TB/s against 6.03 row-chunked;
  // N = 128, 4 waves, 5.84 against 5.88; N = 160, 3 waves, 5.04 against 6.06; coding: N = 100,
  // 5 waves, 5.13 against 4.78; N = 200, 3 waves, 2.82 against 4.80).
  // Equal chunks of at most kTallRows rows (a multiple of 4: one LDS-DMA instruction moves 4 rows).
  // PFA_TALL_* environment overrides exist for timing experiments.
  bool want_tall = false;
  if (n_max > 64u) {
    const uint32_t waves = pfa::kLdsPerCu / std::max(p->lds_bytes, 1u);
    want_tall = waves < 5u || (uint64_t)waves * n_max * 256u < 125000ull;
  }
  if (getenv("PFA_TALL_MIN")) want_tall = n_max > pfa::env_u32("PFA_TALL_MIN", 64u) || p->lds_bytes > pfa::kLdsPerCu;
  if (want_tall) {
    const uint32_t rcap = std::min(64u, std::max(8u, pfa::env_u32("PFA_TALL_ROWS", pfa::kTallRows)));
    const uint32_t nc = (n_max + rcap - 1u) / rcap;
    p->tall = true;
    p->rows_per_chunk = std::min(64u, std::max(8u, ((n_max + nc - 1u) / nc + 3u) & ~3u));
    p->max_slots = std::min(256u, std::max(16u, pfa::env_u32("PFA_TALL_SLOTS", pfa::kTallSlots)));
    p->hist_bins = std::min(n_max / 2u, 256u);
    p->lds_bytes = pfa::tall_layout(p->rows_per_chunk, p->max_slots, p->hist_bins, any_cds).total;
  }

  // chunking: enough single-wave workgroups to fill 256 CUs several times over,
  // but whole alignments per wave once the batch is large (fewer atomics)
  uint64_t sum_tiles = 0;
  for (uint32_t i = 0; i < n_desc; ++i) {
    const pfa_desc& d = descs[i];
    const uint32_t tw = pfa::tile_sites_for(lb, d.flags & PFA_FLAG_CDS);
    if (d.n_rows && d.n_sites) sum_tiles += ((uint64_t)d.n_sites + tw - 1) / tw;
  }
  // 96 single-wave workgroups per CU measured best for the bench workload (10 tiles per chunk:
  // 6.48 TB/s; 20 tiles 6.36, 4 tiles 6.36, 1 tile 5.43)
  const uint64_t target_blocks = (uint64_t)std::max(pfa::ctx().n_cu, 1) * 96ull;
  uint64_t tpc = std::max<uint64_t>(1, sum_tiles / target_blocks);
  if (getenv("PFA_TILES_PER_CHUNK")) tpc = std::max(1u, pfa::env_u32("PFA_TILES_PER_CHUNK", 10u));   // timing experiments
  tpc = std::min<uint64_t>(tpc, std::max<uint32_t>(max_tiles, 1));
  tpc = std::max<uint64_t>(tpc, ((uint64_t)max_tiles + 65534ull) / 65535ull);
  p->n_chunks = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)max_tiles + tpc - 1) / tpc));
  // balance: the same number of chunks, equal shares (40 tiles in 4 chunks = 10 each, not 13+13+13+1)
  p->tiles_per_chunk = std::max<uint32_t>(1u, (max_tiles + p->n_chunks - 1) / p->n_chunks);
  if (any_cds) {  // a coding chunk must start on a codon boundary: 3 tiles = 768 (192) sites
    p->tiles_per_chunk = (p->tiles_per_chunk + 2u) / 3u * 3u;
    p->n_chunks = std::max<uint32_t>(1u, (max_tiles + p->tiles_per_chunk - 1) / p->tiles_per_chunk);
  }

#define PLAN_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      pfa::set_error("%s failed: %s", #call, hipGetErrorString(e__));                         \
      pfa_plan_destroy(p);                                                                    \
      return PFA_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

  hipStream_t st = pfa::ctx().stream;
  PLAN_HIP(hipMalloc((void**)&p->d_descs, (size_t)n_desc * sizeof(pfa_desc)));
  PLAN_HIP(hipMemcpyAsync(p->d_descs, descs, (size_t)n_desc * sizeof(pfa_desc), hipMemcpyHostToDevice, st));
  if (n_row_idx) {
    PLAN_HIP(hipMalloc((void**)&p->d_row_idx, n_row_idx * sizeof(uint32_t)));
    PLAN_HIP(hipMemcpyAsync(p->d_row_idx, row_idx, n_row_idx * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  }
  {
    auto up = [](size_t x) { return (x + 255u) & ~size_t(255); };
    p->off_sfs = up((size_t)n_desc * sizeof(pfa_result));
    p->off_syn = p->off_sfs + up(std::max<size_t>(16, sfs_words * sizeof(uint32_t)));
    p->slab_bytes = p->off_syn + (any_cds ? up((size_t)n_desc * PFA_SYN_BINS * sizeof(uint32_t)) : 0);
    PLAN_HIP(hipMalloc((void**)&p->d_slab, p->slab_bytes));
    PLAN_HIP(hipHostMalloc((void**)&p->h_slab, p->slab_bytes, hipHostMallocDefault));
    PLAN_HIP(hipHostGetDevicePointer((void**)&p->h_slab_dev, p->h_slab, 0));
    p->d_results = reinterpret_cast<pfa_result*>(p->d_slab);
    p->d_sfs = reinterpret_cast<uint32_t*>(p->d_slab + p->off_sfs);
    p->d_syn = any_cds ? reinterpret_cast<uint32_t*>(p->d_slab + p->off_syn) : nullptr;
  }
  const size_t nt = std::max<size_t>(1, p->n_tiles_total);
  if (want & PFA_WANT_VAR_BITMAP)
    PLAN_HIP(hipMalloc((void**)&p->d_var_bm, nt * lb * sizeof(uint64_t)));
  if ((want & PFA_WANT_SN_BITMAP) && any_cds) {
    PLAN_HIP(hipMalloc((void**)&p->d_syn_bm, nt * lb * sizeof(uint64_t)));
    PLAN_HIP(hipMalloc((void**)&p->d_non_bm, nt * lb * sizeof(uint64_t)));
  }
  PLAN_HIP(hipStreamSynchronize(st));   // (the timing events of the ring are created on first use)
#undef PLAN_HIP
  *out = p;
  return PFA_OK;
}

int pfa_plan_run(pfa_plan* p, const uint8_t* d_bases, int variant) {
  PFA_NEED_INIT();
  if (!p || !d_bases) { pfa::set_error("pfa_plan_run: NULL argument"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  PFA_HIP(hipMemsetAsync(p->d_slab, 0, p->slab_bytes, st));
  if (p->d_syn_bm) {  // S / N bitmaps are built with atomicOr: clear them every run
    const size_t nb = std::max<size_t>(1, p->n_tiles_total) * p->lb * sizeof(uint64_t);
    PFA_HIP(hipMemsetAsync(p->d_syn_bm, 0, nb, st));
    PFA_HIP(hipMemsetAsync(p->d_non_bm, 0, nb, st));
  }

  pfa::ScanArgs a;
  a.bases = d_bases;
  a.descs = p->d_descs;
  a.row_idx = p->d_row_idx;
  a.results = p->d_results;
  a.sfs = p->d_sfs;
  a.syn_num = p->d_syn;
  a.var_bm = p->d_var_bm;
  a.syn_bm = p->d_syn_bm;
  a.non_bm = p->d_non_bm;
  a.n_desc = p->n_desc;
  a.tiles_per_chunk = p->tiles_per_chunk;
  a.n_max = p->n_max;
  a.use_rowmap = p->use_rowmap ? 1u : 0u;
  a.rows_per_chunk = p->rows_per_chunk;
  a.max_slots = p->max_slots;
  a.hist_bins = p->hist_bins;

  const int slot = (int)(p->n_runs % pfa::kEventRing);
  if (!p->ev0[slot]) PFA_HIP(hipEventCreate(&p->ev0[slot]));
  if (!p->ev1[slot]) PFA_HIP(hipEventCreate(&p->ev1[slot]));
  PFA_HIP(hipEventRecord(p->ev0[slot], st));
  int rc;
  // kernel STAGE: 2 = LDS-DMA non-temporal (default), 1 = register staging, 0 = LDS-DMA default cache policy
  const int stage = variant == 1 ? 1 : variant == 2 ? 0 : 2;
#define PFA_LAUNCH(CDS, LEAN)                                          \
  (stage == 0 ? pfa::launch<0, CDS, LEAN>(p, a)                        \
              : stage == 1 ? pfa::launch<1, CDS, LEAN>(p, a) : pfa::launch<2, CDS, LEAN>(p, a))
  const bool lean = !p->use_rowmap && !p->d_var_bm && !p->d_syn_bm;
#define PFA_LAUNCH_TALL(CDS, LEAN)                                     \
  (stage == 0 ? pfa::launch_tall<0, CDS, LEAN>(p, a)                   \
              : stage == 1 ? pfa::launch_tall<1, CDS, LEAN>(p, a) : pfa::launch_tall<2, CDS, LEAN>(p, a))
  if (p->tall && lean) rc = p->any_cds ? PFA_LAUNCH_TALL(true, true) : PFA_LAUNCH_TALL(false, true);
  else if (p->tall)    rc = p->any_cds ? PFA_LAUNCH_TALL(true, false) : PFA_LAUNCH_TALL(false, false);
  else if (lean) rc = p->any_cds ? PFA_LAUNCH(true, true) : PFA_LAUNCH(false, true);
  else           rc = p->any_cds ? PFA_LAUNCH(true, false) : PFA_LAUNCH(false, false);
#undef PFA_LAUNCH_TALL
#undef PFA_LAUNCH
  if (rc != PFA_OK) return rc;
  PFA_HIP(hipEventRecord(p->ev1[slot], st));
  p->n_runs += 1;
  return PFA_OK;
}

int pfa_plan_kernel_ms(pfa_plan* p, float* ms_out, int n) {
  PFA_NEED_INIT();
  if (!p || !ms_out || n < 0) { pfa::set_error("pfa_plan_kernel_ms: bad argument"); return PFA_ERR_ARG; }
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  const uint64_t have = std::min<uint64_t>(p->n_runs, pfa::kEventRing);
  const int cnt = (int)std::min<uint64_t>(have, (uint64_t)n);
  for (int i = 0; i < cnt; ++i) {
    const uint64_t run = p->n_runs - cnt + i;
    const int slot = (int)(run % pfa::kEventRing);
    PFA_HIP(hipEventElapsedTime(&ms_out[i], p->ev0[slot], p->ev1[slot]));
  }
  return cnt;
}

int pfa_plan_fetch(pfa_plan* p, pfa_result* results, uint32_t* sfs, uint32_t* syn_num,
                   uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap) {
  PFA_NEED_INIT();
  if (!p) { pfa::set_error("pfa_plan_fetch: plan is NULL"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  const size_t nt = p->n_tiles_total;
  if (syn_num && !p->d_syn) { pfa::set_error("pfa_plan_fetch: plan has no coding unit"); return PFA_ERR_ARG; }
  if (results || sfs || syn_num) {
    // one device-side copy into pinned memory, then plain memcpy into the caller's (pageable) arrays
    const size_t n16 = p->slab_bytes / 16u;   // slab sections are padded to 256 bytes
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255u) / 256u, 1024u);
    hipLaunchKernelGGL(pfa::slab_to_host_kernel, dim3(blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(p->d_slab), reinterpret_cast<uint4*>(p->h_slab_dev), n16);
    PFA_HIP(hipGetLastError());
    PFA_HIP(hipStreamSynchronize(st));
    if (results) memcpy(results, p->h_slab, (size_t)p->n_desc * sizeof(pfa_result));
    if (sfs && p->sfs_words) memcpy(sfs, p->h_slab + p->off_sfs, p->sfs_words * sizeof(uint32_t));
    if (syn_num) memcpy(syn_num, p->h_slab + p->off_syn, (size_t)p->n_desc * PFA_SYN_BINS * sizeof(uint32_t));
  }
  if (var_bitmap) {
    if (!p->d_var_bm) { pfa::set_error("pfa_plan_fetch: plan was created without PFA_WANT_VAR_BITMAP"); return PFA_ERR_ARG; }
    if (nt) PFA_HIP(hipMemcpyAsync(var_bitmap, p->d_var_bm, nt * p->lb * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  }
  if (syn_bitmap || non_bitmap) {
    if (!p->d_syn_bm) { pfa::set_error("pfa_plan_fetch: plan has no S/N bitmaps"); return PFA_ERR_ARG; }
    const size_t w = p->lb;
    if (syn_bitmap && nt) PFA_HIP(hipMemcpyAsync(syn_bitmap, p->d_syn_bm, nt * w * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (non_bitmap && nt) PFA_HIP(hipMemcpyAsync(non_bitmap, p->d_non_bm, nt * w * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  }
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_plan_results_dev(pfa_plan* p, void** d_results, size_t* bytes) {
  PFA_NEED_INIT();
  if (!p || !d_results || !bytes) { pfa::set_error("pfa_plan_results_dev: NULL argument"); return PFA_ERR_ARG; }
  *d_results = p->d_results;
  *bytes = (size_t)p->n_desc * sizeof(pfa_result);
  return PFA_OK;
}

uint32_t pfa_plan_n_tiles(const pfa_plan* p) { return p ? p->n_tiles_total : 0u; }
uint64_t pfa_plan_sfs_words(const pfa_plan* p) { return p ? p->sfs_words : 0ull; }
uint32_t pfa_plan_tiles_per_chunk(const pfa_plan* p) { return p ? p->tiles_per_chunk : 0u; }
uint32_t pfa_plan_lds_bytes(const pfa_plan* p) { return p ? p->lds_bytes : 0u; }
uint32_t pfa_plan_rows_per_chunk(const pfa_plan* p) { return p ? p->rows_per_chunk : 0u; }

int pfa_scan_batch(const uint8_t* bases, uint64_t n_bases_bytes, const pfa_desc* descs,
                   uint32_t n_desc, const uint32_t* row_idx, size_t n_row_idx,
                   pfa_result* results, uint32_t* sfs, uint32_t* syn_num, uint64_t* var_bitmap,
                   uint64_t* syn_bitmap, uint64_t* non_bitmap) {
  PFA_NEED_INIT();
  if (!bases && n_bases_bytes) { pfa::set_error("pfa_scan_batch: bases is NULL"); return PFA_ERR_ARG; }
  pfa_plan* plan = nullptr;
  const uint32_t want = (var_bitmap ? PFA_WANT_VAR_BITMAP : 0u) | ((syn_bitmap || non_bitmap) ? PFA_WANT_SN_BITMAP : 0u);
  int rc = pfa_plan_create(descs, n_desc, row_idx, n_row_idx, n_bases_bytes, want, &plan);
  if (rc != PFA_OK) return rc;
  void* d_bases = nullptr;
  rc = pfa_dev_alloc((size_t)n_bases_bytes, &d_bases);
  if (rc == PFA_OK) rc = pfa_upload(d_bases, bases, (size_t)n_bases_bytes);
  if (rc == PFA_OK) rc = pfa_plan_run(plan, (const uint8_t*)d_bases, 0);
  if (rc == PFA_OK)
    rc = pfa_plan_fetch(plan, results, sfs, plan->d_syn ? syn_num : nullptr, var_bitmap,
                        plan->d_syn_bm ? syn_bitmap : nullptr, plan->d_syn_bm ? non_bitmap : nullptr);
  if (d_bases) (void)hipFree(d_bases);
  pfa_plan_destroy(plan);
  return rc;
}

int64_t pfa_bitmap_positions(const uint64_t* words, uint32_t n_tiles, uint32_t words_per_tile,
                             uint32_t tile_sites, uint32_t n_sites, uint32_t* pos_out, uint64_t cap) {
  if (!words || (!pos_out && cap)) { pfa::set_error("pfa_bitmap_positions: NULL argument"); return PFA_ERR_ARG; }
  // bit l of word b of a tile <-> tile-local site  words_per_tile * l + b
  const uint32_t per_lane = words_per_tile;
  uint64_t n = 0;
  for (uint32_t t = 0; t < n_tiles; ++t) {
    const uint64_t* w = words + (uint64_t)t * words_per_tile;
    // gather in ascending site order: iterate lanes then sub-position
    const uint32_t groups = words_per_tile / per_lane;
    for (uint32_t g = 0; g < groups; ++g) {
      bool any = false;
      for (uint32_t b = 0; b < per_lane; ++b) any = any || w[g * per_lane + b];
      if (!any) continue;
      for (uint32_t l = 0; l < 64u; ++l)
        for (uint32_t b = 0; b < per_lane; ++b)
          if ((w[g * per_lane + b] >> l) & 1ull) {
            const uint64_t site = (uint64_t)t * tile_sites + (uint64_t)per_lane * (64u * g + l) + b;
            if (site >= n_sites) continue;
            if (n >= cap) { pfa::set_error("pfa_bitmap_positions: output capacity %llu too small", (unsigned long long)cap); return PFA_ERR_ARG; }
            pos_out[n++] = (uint32_t)site;
          }
    }
  }
  return (int64_t)n;
}

uint32_t pfa_classify_codon_mask(uint64_t mask) {
  return pfa::classify_sense_mask(mask & ~pfa::kHostTables.stop_mask, pfa::kHostTables, pfa::kHostTables.annot_info);
}

int pfa_classify_codon_masks_dev(const uint64_t* masks, uint32_t n, uint32_t* labels_out) {
  PFA_NEED_INIT();
  if (n == 0) return PFA_OK;
  if (!masks || !labels_out) { pfa::set_error("pfa_classify_codon_masks_dev: NULL argument"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  unsigned long long* d_m = nullptr;
  uint32_t* d_l = nullptr;
  PFA_HIP(hipMalloc((void**)&d_m, (size_t)n * 8u));
  if (hipMalloc((void**)&d_l, (size_t)n * 4u) != hipSuccess) { (void)hipFree(d_m); pfa::set_error("hipMalloc failed"); return PFA_ERR_HIP; }
  hipError_t e = hipMemcpyAsync(d_m, masks, (size_t)n * 8u, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pfa::classify_masks_kernel, dim3(std::min<uint32_t>(n, 4096u)), dim3(64), 0, st, d_m, n, d_l);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(labels_out, d_l, (size_t)n * 4u, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_m);
  (void)hipFree(d_l);
  if (e != hipSuccess) { pfa::set_error("pfa_classify_codon_masks_dev: %s", hipGetErrorString(e)); return PFA_ERR_HIP; }
  return PFA_OK;
}

uint32_t pfa_codon_syn_thirds(uint32_t codon) { return codon < 64u ? pfa::kHostTables.syn3[codon] : 0u; }

}  // extern "C"
