// pfa_layout.h — tile geometry, LDS layouts and kernel arguments of the scan kernels; shared by
// the kernel translation units and the plan code (host and device).
#pragma once
#include <stdint.h>

#include "pfa_codon.h"   // PFA_HD
#include "../../include/pfa.h"

namespace pfa {

// ---------------------------------------------------------------- geometry
// A lane owns 4 adjacent sites (one 32-bit word of a tile row); an LDS-DMA instruction moves
// 1 KiB.  Tiles are TW = 256, 128 or 64 sites wide.  A 256-site tile gives every lane its own
// word of a row; narrower tiles split the ROWS over 256 / TW lane groups instead (group g reads
// rows g, g + G, g + 2G, ...): the LDS image of a narrow tile, read as 64 words per step, is
// exactly G consecutive rows, so tier 1 keeps one conflict-free ds_read_b32 per lane per step and
// takes N / G steps.  Narrow tiles serve units whose rows would not let enough wavefronts share
// the CU with 256-site tiles (LDS per wave = N x TW), and tiny plans (more waves per byte).
template <int TW_>
struct Geo {
  static_assert(TW_ == 256 || TW_ == 128 || TW_ == 64, "tile width");
  static constexpr uint32_t TW = TW_;                 // sites (= bytes) per LDS tile row
  static constexpr uint32_t GROUPS = 256u / TW;       // lane groups sharing the rows of a tile
  static constexpr uint32_t LPG = 64u / GROUPS;       // lanes per group
  static constexpr uint32_t LPR = TW / 16u;           // DMA lanes per row
  static constexpr uint32_t RPI = 64u / LPR;          // rows per DMA instruction
};

struct LdsLayout {
  uint32_t tile, list, hist, rowmap, synhist, total;
};

PFA_HD inline uint32_t round16(uint32_t x) { return (x + 15u) & ~15u; }

// bins of synonymous-site numerators kept in LDS: codon columns with 2 .. kSynLdsBins + 1 distinct
// clean codons (one codon: a register; more: global atomics, hardly ever)
static constexpr uint32_t kSynLdsBins = 32u;

// scan_kernel: `n_max` = most rows of any unit of the launch, `tw` = its tile width; `mapped`: the
// instantiation serves row-index lists.
//   tile   exactly N rows (lanes of the last DMA instruction beyond row N-1 are masked off; narrow
//          tiles are read G rows at a time, the last step by the lane groups that still have a row)
//   list   the tile's variable sites, tw entries: `site | XOR byte << 8` from the compaction, turned
//          by tier 2 (coding launches) into the site's `H | allele mask`: 16 bits (H < 4096 | mask <<
//          12) when no unit has more than 64 rows, else 32 (H | mask << 24).  Coding launches keep two
//          more entries in FRONT of it: the last two sites of the previous tile.
PFA_HD inline LdsLayout lds_layout(uint32_t tw, uint32_t n_max, bool cds, bool mapped) {
  const uint32_t n = n_max ? n_max : 1u;
  const uint32_t entry = (cds && n > 64u) ? 4u : 2u;
  LdsLayout o;
  o.tile = 0;
  o.list = round16(n * tw) + (cds ? 16u : 0u);
  o.hist = o.list + entry * tw;
  o.rowmap = o.hist + round16((n / 2u) * 4u + 4u);
  o.synhist = o.rowmap + (mapped ? round16(n * 4u) : 0u);   // rowmap: byte offsets of the unit's rows (--pops)
  o.total = o.synhist + (cds ? kSynLdsBins * 4u : 0u);
  return o;
}

// One launch = one bucket of a plan: units that share a kernel instantiation and an LDS size.
struct ScanArgs {
  const uint8_t* bases;
  const pfa_desc* descs;
  const uint32_t* unit_ids;   // the bucket's units (indices into descs / results), or null: unit = blockIdx.x
  const uint32_t* row_idx;
  pfa_result* results;
  uint32_t* sfs;
  uint32_t* syn_num;
  unsigned long long* var_bm;
  unsigned long long* syn_bm;
  unsigned long long* non_bm;
  uint32_t tiles_per_chunk;
  uint32_t n_max;           // most rows of any unit of the bucket (sizes the LDS layout)
  uint32_t rows_per_chunk;  // tall kernel: rows staged per pass (multiple of 4, 8..64)
  uint32_t max_slots;       // tall kernel: count-table slots per tile (variable sites beyond take the column path)
  uint32_t hist_bins;       // tall kernel: SFS bins kept in LDS (higher bins go straight to global atomics)
  // Plans with few units but many chunks per unit keep `rep_mask + 1` replicas of their outputs
  // (records | sfs | syn_num, `rep_stride` bytes apart): chunk c adds into replica c & rep_mask, the
  // fetch sums them.  Otherwise thousands of wavefronts would queue on the atomics of a handful of
  // cache lines (measured: BASELINE config 3 as one launch, 24 -> 66 us when 4x more wavefronts shared it).
  uint32_t rep_mask;
  uint32_t rep_stride;
};

// Rows per chunk of the row-chunked kernel: chunks of exactly this many rows run straight-line code (staging, tier 1, tier 2);
// the plan cuts every unit into such chunks and a shorter last one.  (Rows per chunk swept 28..64 at N = 1 000 / 5 000: flat.)
static constexpr uint32_t kTallRowsFast = 48u;

struct TallLayout {
  uint32_t tile, list, cnt, hist, hsite, synhist, cslots, total;
};

// scan_tall_kernel; `wide`: coding units above 2 048 rows keep H in 56 bits (H | mask << 56)
PFA_HD inline TallLayout tall_layout(uint32_t rows_per_chunk, uint32_t max_slots, uint32_t hist_bins, bool cds, bool wide) {
  TallLayout o;
  o.tile = 0;                                   // R x 256 bytes
  o.list = round16(rows_per_chunk * 256u);      // list: per slot  x | d << 8 | mode << 16 | site << 24
  o.cnt = o.list + max_slots * 4u;              // cnt: six u32 counters per slot (two-allele mode uses two)
  // Once the slots of a tile have been turned into H and SFS, list + cnt are reused for the
  // column path: its site list (256 B at `list`) and its 256-bin histogram (1 KiB behind it).
  const uint32_t slots_bytes = max_slots * 28u;
  o.hist = o.list + (slots_bytes > 1280u ? slots_bytes : 1280u);
  // coding plans: hsite[s] = H | allele mask << 24 (wide: << 56) of variable site s, with two entries in
  // front for the last two sites of the previous tile, and the bins of synonymous-site numerators
  const uint32_t hbytes = wide ? 8u : 4u;
  o.hsite = o.hist + round16(hist_bins * 4u + 4u) + (cds ? 16u : 0u);
  o.synhist = o.hsite + (cds ? 256u * hbytes : 0u);
  // codon slots: the codon sets of up to 8 joint columns of the tile, tracked chunk by chunk
  o.cslots = o.synhist + (cds ? kSynLdsBins * 4u : 0u);
  o.total = o.cslots + (cds ? 32u + 64u : 0u);
  return o;
}

}  // namespace pfa
