// pfa_kernels.h — device code shared by the scan kernels of libpfa_hip.so (gfx950 only):
// LDS layouts, wavefront helpers, the wave-uniform codon-set classifier and the coding pass.
//
// What the kernels compute (all integers), per scan unit of N rows:
//   S      = number of site columns with more than one distinct byte   PolyFastA.py:252-261
//   H      = sum over those columns of (N^2 - sum_j c_j^2)              PolyFastA.py:485-492
//   SFS    = histogram of the second-largest A/C/G/T allele count        PolyFastA.py:274-282
//   coding = per codon column the set of distinct clean codons -> nstops,
//            missing columns, synonymous-site numerators, and the S / N label
//            of each variable codon position with the H of its site       PolyFastA.py:284-315,319-434
//
// Kernel shape.  One 64-lane wavefront per workgroup; no inter-wave
// synchronisation anywhere.  A wavefront owns a chunk of consecutive column tiles of one
// unit; partial sums leave through integer atomics (order-independent).
//   pfa_scan_classic.hip  scan_kernel: all N rows of a tile in LDS (tiles of 256, 128 or 64 sites)
//   pfa_scan_tall.hip     scan_tall_kernel: units with many rows, streamed in row chunks
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pfa_codon.h"
#include "pfa_common.h"
#include "pfa_layout.h"

// unroll depths of the row loops of tier 1 (LDS words) and tier 2 (column bytes): how many LDS
// reads a wavefront keeps in flight per round trip (timing experiments may override them)
#ifndef PFA_UNROLL_T1
#define PFA_UNROLL_T1 8
#endif
#ifndef PFA_UNROLL_T2
#define PFA_UNROLL_T2 4
#endif

namespace pfa {
static constexpr ct::Tables kHostTables = ct::make_tables();   // also the source of literal constants in device code
}

// one copy per translation unit (the library is built without relocatable device code)
static __constant__ pfa::ct::Tables d_tables = pfa::ct::make_tables();
static __constant__ pfa::SingleLut d_single_lut = pfa::make_single_lut();

namespace pfa {

// The kernel arguments once more, straight from the kernarg segment.  Pointers that are only
// needed after the tile loop (results, sfs, syn_num) are read through this at their point of use:
// an s_load there is cheaper than keeping them in scalar registers across the loop (the scan
// kernels are short of SGPRs), and the empty asm keeps the compiler from merging these loads
// with the prologue's.
typedef const __attribute__((address_space(4))) ScanArgs* kernarg_ptr_t;
__device__ __forceinline__ kernarg_ptr_t kernargs_again() {
  kernarg_ptr_t p = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// the output replica `off` bytes behind the first (ScanArgs::rep_stride)
template <class T>
__device__ __forceinline__ T* replica(T* p, uint32_t off) {
  return reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(p) + off);
}

// H and allele-presence mask (which of A, C, T, G occur: codon base order) of a variable site,
// as the coding pass reads them; HT = uint32_t, or unsigned long long for units above 2 048 rows
template <class HT>
struct HMask {
  HT h;
  uint32_t m;
};

// ---------------------------------------------------------------- device helpers
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Grid: x = the bucket's units, y = chunks of consecutive tiles, so consecutive workgroups scan the same columns of different
// units (the transposed grid measured equal or slower, docs/TUNING.md).
#define PFA_UNIT_IDX blockIdx.x
#define PFA_CHUNK_IDX blockIdx.y
#define PFA_GRID(units, chunks) dim3((units), (chunks), 1)

__device__ inline uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}
// Sum over the 64 lanes, the same in every lane afterwards: DPP moves only (six VALU adds and a readlane).  __shfl_xor goes through
// ds_bpermute, i.e. six dependent round trips through the LDS queue per sum — about a microsecond for the three sums at the end of
// a chunk and three for a coding chunk's ten, 4-12 % of the life of a workgroup that scans four tiles (round 5).
__device__ inline uint32_t wave_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror: the other quad of eight lanes
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror: the other half of the row of sixteen
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit: four 32-bit sums of 16-bit limbs (none can overflow), step by step next to each other so that one chain's add fills
// the others' DPP wait states
__device__ inline unsigned long long wave_sum(unsigned long long v) {
  uint32_t a = (uint32_t)v & 0xFFFFu, b = (uint32_t)v >> 16, c = (uint32_t)(v >> 32) & 0xFFFFu, d = (uint32_t)(v >> 48);
#define PFA_SUM_STEP(ctrl, rows)                                                         \
  a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, ctrl, rows, 0xF, false); \
  b += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, ctrl, rows, 0xF, false); \
  c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, ctrl, rows, 0xF, false); \
  d += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, ctrl, rows, 0xF, false);
  PFA_SUM_STEP(0xB1, 0xF)
  PFA_SUM_STEP(0x4E, 0xF)
  PFA_SUM_STEP(0x141, 0xF)
  PFA_SUM_STEP(0x140, 0xF)
  PFA_SUM_STEP(0x142, 0xA)
  PFA_SUM_STEP(0x143, 0xC)
#undef PFA_SUM_STEP
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)a, 63) +
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b, 63) << 16) +
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c, 63) << 32) +
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)d, 63) << 48);
}
// sum over the 4 lanes of a quad with DPP quad_perm moves (VALU only; __shfl_xor would go
// through ds_bpermute and the LDS queue)
__device__ inline uint32_t quad_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  return v;
}
// sum over aligned groups of 4, 8 or 16 lanes (`lps_log` = 2, 3, 4: wave-uniform), every lane of the
// group gets it (DPP moves only)
__device__ inline uint32_t group_sum(uint32_t v, uint32_t lps_log) {
  v = quad_sum(v);
  if (lps_log >= 3u) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror: the other quad
  if (lps_log >= 4u) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror: the other half row
  return v;
}
// Workgroups are ONE wavefront: "synchronising" it only means ordering its LDS traffic (lanes
// exchange data through LDS).  __syncthreads() would also wait for every global load in flight
// (s_waitcnt vmcnt(0)), i.e. for the tile that was just requested ahead of time.
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LDS atomic add (no return value) that the compiler does not order against the LDS-DMA in flight.  It cannot tell that these
// bins and the tile buffer the DMA writes never overlap, so in front of every ds_add it emits it puts `s_waitcnt vmcnt(0)` — a
// wait for the WHOLE tile a wavefront has just requested ahead of time: the coding pass's table work after the request, meant
// to run under those loads, then ran behind them (round 4, read off the ISA).  Completion is ordered by the kernels' own
// wave_lds_sync() (LDS operations of a wavefront complete in order).  PFA_LDS_ATOMIC_WAITS=1 (compile time): the builtin again.
// REQUIREMENT on every call site: a wave_lds_sync() stands between the last lds_add_u32() on a bin and any read of it (hist and
// synhist are read at the end of a chunk only, behind one).  Nothing the compiler checks: tests/test_gpu_parity.py builds the
// library both ways and compares the coding cases of G5 / G10 bit for bit.
__device__ inline void lds_add_u32(uint32_t* p, uint32_t v) {
#if defined(PFA_LDS_ATOMIC_WAITS) && PFA_LDS_ATOMIC_WAITS
  atomicAdd(p, v);
#else
  asm volatile("ds_add_u32 %0, %1" ::"v"((uint32_t)(uintptr_t)(lptr_t)p), "v"(v) : "memory");
#endif
}

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
// both halves step by step, so that one chain's instruction fills the other's DPP wait states
__device__ inline unsigned long long wave_or_uniform(unsigned long long v) {
  uint32_t a = (uint32_t)v, b = (uint32_t)(v >> 32);
#define PFA_OR_STEP(ctrl, rows)                                                          \
  a |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, ctrl, rows, 0xF, false); \
  b |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, ctrl, rows, 0xF, false);
  PFA_OR_STEP(0xB1, 0xF)
  PFA_OR_STEP(0x4E, 0xF)
  PFA_OR_STEP(0x141, 0xF)
  PFA_OR_STEP(0x140, 0xF)
  PFA_OR_STEP(0x142, 0xA)
  PFA_OR_STEP(0x143, 0xC)
#undef PFA_OR_STEP
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)a, 63) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b, 63) << 32;
}
// a value the program knows to be the same in every lane, made a scalar for the compiler
__device__ inline unsigned long long wave_uniform(unsigned long long v) {
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
         (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}
// bit 7 of every byte of `v` that is not zero
__device__ inline uint32_t byte_nonzero_bits(uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; }
__device__ inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// Wave-uniform form of classify_sense_mask (pfa_codon.h; PolyFastA.py:331-434) for a set `ms`
// held in scalar registers: lane c stands for codon c (`my_info` = annot_info[c], `my_same` =
// same_annot[c]), so the multiplicity rule for three or more codons needs no loop — a codon's
// label multiplicity is one popcount, and "some label occurs at least t times" is one ballot.
// Out of line (round 5): a codon column with two or three variable positions comes up about once in two tiles, and the pair
// table's address arithmetic and the multiplicity rule's scalar state no longer weigh on the registers of the tile loop.
__device__ __attribute__((noinline)) uint32_t classify_uniform(unsigned long long ms_, uint32_t lane, uint32_t my_info,
                                                               unsigned long long my_same) {
  const unsigned long long ms = wave_uniform(ms_);   // (arguments arrive in vector registers)
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
// sites wide and line-aligned (scan_kernel also runs it on 128-site tiles: the lanes of the second
// row group then own no site); 256 = 1 (mod 3), so the first codon of tile t starts
// t mod 3 sites before it (128 = 2 (mod 3): 2t mod 3): the tier-1 words of the previous tile's last lane are carried in
// registers, and chunks of coding plans hold a multiple of 3 tiles, so a chunk always starts
// on a codon boundary.  What a kernel supplies: the tile's tier-1 words, a reader of the per-site
// HMask (variable sites only), and a reader of the codon set of one column.
// BITMAPS: the plan may want the per-site S / N bitmaps (compiled out otherwise).
// SPLIT (128-site tiles): both lane groups hold the tile's tier-1 words after the merge, so the two
// candidate columns of a group-0 lane are shared out — lane j takes the first, its twin j + 32 the
// second — and every per-column loop below runs once instead of twice.
template <bool BITMAPS, bool SPLIT = false>
struct CodonPass {
  static constexpr uint32_t LB = 4u, WW = 64u * LB, NIT = SPLIT ? 1u : 2u;
  // wave constants: lane c stands for codon c in the tables below
  uint32_t lane, lane3;           // lane, (lane within its group) mod 3
  uint32_t jlane, it0;            // SPLIT: lane within its group of 32, which of the two candidate columns; else lane, 0
  uint32_t last_lane;             // lane that owns the tile's last four sites (63; 31 for 128-site tiles)
  uint32_t my_info;               // annot_info[c] (classify_uniform)
  unsigned long long my_same;     // same_annot[c]
  uint32_t my_cinfo;              // synonymous-neighbour count | stop << 7, read with ds_bpermute
  uint32_t my_slut[3];            // the three 256-byte planes of SingleLut, one word per lane each
  uint32_t* synhist;              // LDS: kSynLdsBins bins of synonymous-site numerators, codon counts 2 .. kSynLdsBins + 1
  uint32_t unit;
  uint32_t rep_off;               // byte offset of this chunk's output replica (ScanArgs::rep_stride)
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

  __device__ void init(uint32_t lane_, uint32_t unit_, uint32_t* synhist_, const ScanArgs& A, const pfa_desc& D, uint32_t last_lane_ = 63u) {
    lane = lane_;
    last_lane = last_lane_;
    jlane = SPLIT ? lane_ & 31u : lane_;
    it0 = SPLIT ? lane_ >> 5 : 0u;
    lane3 = jlane % 3u;
    unit = unit_;
    rep_off = (PFA_CHUNK_IDX & A.rep_mask) * A.rep_stride;
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

  // synonymous-site numerator t3 of a column with k distinct clean codons (PolyFastA.py:307): columns
  // with one codon are summed in a register, 2 .. kSynLdsBins + 1 in LDS, more (hardly ever) in memory
  __device__ void syn_add(uint32_t k, uint32_t t3) {
    if (k == 1u) acc_syn1 += t3;
    else if (k - 2u < kSynLdsBins) lds_add_u32(&synhist[k - 2u], t3);
    else atomicAdd(replica(kernargs_again()->syn_num, rep_off) + (uint64_t)unit * PFA_SYN_BINS + k, t3);
  }

  // first of this lane's four sites that ends a codon: (5 - (site0 + 4 lane) mod 3) mod 3, without a
  // division per tile: 4 = 1 (mod 3), so it is 2 - ((site0 mod 3 + lane mod 3) mod 3)
  __device__ uint32_t first_codon_end(uint32_t site0_) const {
    uint32_t m = site0_ % 3u + lane3;     // site0 is wave-uniform: scalar arithmetic
    m -= m >= 3u ? 3u : 0u;
    return 2u - m;
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
    if (jlane == 0u) { pa = carry_va; pd = carry_vd; }
    carry_va = __builtin_amdgcn_readlane(va, last_lane);
    carry_vd = __builtin_amdgcn_readlane(vd, last_lane);
    const unsigned long long Wa = (unsigned long long)pa | (unsigned long long)va << 32;
    const unsigned long long Wd = (unsigned long long)pd | (unsigned long long)vd << 32;
    const uint32_t o1 = first_codon_end(site0);
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      const uint32_t o = o1 + 3u * (it0 + it);
      const bool has = o < 4u && 4u * jlane + o < lim;
      xw[it] = (uint32_t)(Wa >> (8u * (2u + (o & 3u)))) & 0xFFFFFFu;
      dw[it] = (uint32_t)(Wd >> (8u * (2u + (o & 3u)))) & 0xFFFFFFu;
      endo[it] = has ? (int)(4u * jlane + o) : -1;
      mjoint[it] = 0ull;
      single[it] = mono[it] = false;
      pv[it] = 0u;
    }
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      // (branch-free: predicated selects instead of an exec-masked block per candidate column)
      const bool has = endo[it] >= 0;
      const uint32_t dv = dw[it];
      const bool v0 = (dv & 0xFFu) != 0u, v1 = (dv & 0xFF00u) != 0u, v2 = (dv & 0xFF0000u) != 0u;
      const uint32_t nvar = (uint32_t)v0 + (uint32_t)v1 + (uint32_t)v2;
      single[it] = has && nvar == 1u;                      // one variable position: resolved from its allele mask
      mono[it] = has && nvar == 0u;
      pv[it] = v0 ? 0u : v1 ? 1u : 2u;
      mjoint[it] = __ballot(has && nvar >= 2u);            // rare: the codons must be read row by row
    }
  }

  // One column with two or three variable positions, given the wave-uniform set `M` of its
  // distinct clean codons: stop / synonymous-site bookkeeping and the classifier
  // (PolyFastA.py:301-315,319-434).  `hget(rel)`: HMask of a variable site.
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
      syn_add(k, t3);
      acc_stops += stop ? 1u : 0u;
#pragma unroll
      for (uint32_t p = 0; p < 3u; ++p) {
        const bool is_s = (labels >> p) & 1u, is_n = (labels >> (3u + p)) & 1u;
        if (is_s || is_n) {
          const auto hs = hget(rel + (int)p).h;
          if (is_s) { acc_segS += 1u; acc_HS += hs; }
          else { acc_segN += 1u; acc_HN += hs; }
          if (BITMAPS && syn_bm != nullptr) mark_site(is_s, rel + (int)p);
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
    uint32_t bad_[NIT], cidx_[NIT];   // per candidate column: unclean-byte flags (one byte each), codon index of the consensus bytes
#pragma unroll
    for (uint32_t it = 0; it < NIT; ++it) {
      const uint32_t w = xw[it];
      const uint32_t t2 = w & 0x060606u;
      const uint32_t e2 = ((w >> 2) & 0x010101u) & ~(w >> 1);
      bad_[it] = (0x414141u + t2 + __umul24(e2, 15u)) ^ w;
      // 2-bit codes at bits 1, 9, 17 of t2 -> 16 c0 + 4 c1 + c2: one 24-bit multiply lines them up at bits 17..22
      cidx_[it] = (__umul24(t2, 0x100401u) >> 17) & 0x3Fu;
      const uint32_t ci = lane_lookup(cidx_[it], my_cinfo);
      const bool clean = mono[it] && bad_[it] == 0u;       // (predicated adds: no exec-masked blocks)
      acc_syn1 += clean ? (ci & 7u) : 0u;
      acc_stops += clean ? (ci >> 7) : 0u;
      acc_missing += (mono[it] && bad_[it] != 0u) ? 1u : 0u;
    }
    // One variable position, on the lane that owns the column: the codon set is the consensus
    // codon with every A/C/G/T allele of that position substituted (rows with another symbol
    // there hold an unclean codon and drop out, PolyFastA.py:301).  Only that position can be
    // labelled.  The two fixed bytes are the lane's consensus bytes; the allele mask and H of
    // the variable site come from the allele scan.  All lanes run it, `valid` marks the lanes
    // that own such a column (the table reads are ds_bpermute's and need every lane enabled).
    auto single_col = [&](bool valid, uint32_t bad3, uint32_t cidx, uint32_t p, int end_rel) {
      const int rel = valid ? end_rel - 2 : 0;
      const auto hv = hget(rel + (int)p);
      const uint32_t am = valid ? hv.m : 0u;
      // exactly-A/C/G/T test of the two fixed bytes (the byte of position p is meaningless)
      const uint32_t bad = bad3 & ~(0xFFu << (8u * p));
      const uint32_t k = bad ? 0u : (uint32_t)__popc(am);   // distinct clean codons
      // the two fixed bases out of the codon index 16 c0 + 4 c1 + c2
      const uint32_t b4 = p == 0u ? (cidx & 15u) : p == 1u ? ((cidx >> 2) & 12u) | (cidx & 3u) : cidx >> 2;
      const uint32_t j = b4 * 16u + (am & 15u);             // entry of plane p
      const uint32_t w0 = lane_lookup(j >> 2, my_slut[0]), w1 = lane_lookup(j >> 2, my_slut[1]), w2 = lane_lookup(j >> 2, my_slut[2]);
      const uint32_t v = ((p == 0u ? w0 : p == 1u ? w1 : w2) >> (8u * (j & 3u))) & 0xFFu;
      const bool live = valid && k != 0u;
      acc_missing += (valid && k == 0u) ? 1u : 0u;
      if (live) syn_add(k, v & 0x1Fu);
      acc_stops += live ? (v >> 5) & 1u : 0u;
      const bool is_s = live && (v & 0x40u) != 0u, is_n = live && (v & 0x80u) != 0u;
      const auto hs = hv.h;
      acc_segS += is_s ? 1u : 0u;
      acc_HS += is_s ? hs : 0;
      acc_segN += is_n ? 1u : 0u;
      acc_HN += is_n ? hs : 0;
      if (BITMAPS && (is_s || is_n) && syn_bm != nullptr) mark_site(is_s, rel + (int)p);
    };
    if constexpr (SPLIT) {
      if (__any(single[0])) single_col(single[0], bad_[0], cidx_[0], pv[0], endo[0]);
    } else {
      const bool any1 = single[0] || single[NIT - 1u];
      if (__any(any1)) {
        const bool sel = !single[0];
        single_col(any1, sel ? bad_[NIT - 1u] : bad_[0], sel ? cidx_[NIT - 1u] : cidx_[0], sel ? pv[NIT - 1u] : pv[0], sel ? endo[NIT - 1u] : endo[0]);
      }
      const bool both = single[0] && single[NIT - 1u];     // a lane owning two such columns: rare
      if (__any(both)) single_col(both, bad_[NIT - 1u], cidx_[NIT - 1u], pv[NIT - 1u], endo[NIT - 1u]);
    }
  }

  // the trailing 1-2 sites of an alignment whose length is not a multiple of 3 never form a
  // clean codon (PolyFastA.py:301-305)
  __device__ void last_tile(uint32_t L) {
    if ((L % 3u) != 0u && lane == 0u) acc_missing += 1u;
  }

  // the chunk's partial sums -> the unit's record and its bins of synonymous-site numerators
  __device__ void flush(uint32_t unit) {
    acc_segS = wave_sum(acc_segS);
    acc_segN = wave_sum(acc_segN);
    acc_stops = wave_sum(acc_stops);
    acc_missing = wave_sum(acc_missing);
    acc_syn1 = wave_sum(acc_syn1);
    acc_HS = wave_sum(acc_HS);
    acc_HN = wave_sum(acc_HN);
    const kernarg_ptr_t K = kernargs_again();
    pfa_result* const R = replica(K->results, rep_off) + unit;
    uint32_t* const syn_num_unit = replica(K->syn_num, rep_off) + (uint64_t)unit * PFA_SYN_BINS;
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
    if (lane < kSynLdsBins) {
      const uint32_t v = synhist[lane];
      if (v) atomicAdd(&syn_num_unit[lane + 2u], v);
    }
  }
};

// codon of three bytes packed in a word (byte 0 = first base): its index as a one-hot 64-bit set,
// or 0 when any byte is not exactly A/C/G/T (PolyFastA.py:301)
__device__ inline unsigned long long codon_bit(uint32_t x0, uint32_t x1, uint32_t x2) {
  const bool ok = base_clean(x0) && base_clean(x1) && base_clean(x2);
  return ok ? 1ull << (base_code(x0) * 16u + base_code(x1) * 4u + base_code(x2)) : 0ull;
}

}  // namespace pfa
