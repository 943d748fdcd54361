// pfa_codon.h — genetic-code tables and the codon-set classifier, shared by the
// device kernels and the host entry points of libpfa_hip.so.
//
// Everything is derived at compile time from the standard genetic code
// (NCBI translation table 1); no table is transcribed from the reference.
// The reference's behaviour that these functions reproduce:
//   - per-codon annotation "amino acid / third-position family size / IUPAC
//     class of the family"                                  PolyFastA.py:324-329
//   - synonymous-site weight per codon (x3, integer)         PolyFastA.py:536-557
//   - decision tree on the SET of distinct sense codons of a codon column
//                                                            PolyFastA.py:319-434
// Base encoding throughout: e = (ascii >> 1) & 3, i.e. A=0, C=1, T=2, G=3;
// codon index = 16*e0 + 4*e1 + e2; a codon SET is a 64-bit mask of indices.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PFA_HD __host__ __device__
#else
#define PFA_HD
#endif

namespace pfa {

// ---- compile-time derivation ------------------------------------------------
namespace ct {

// amino acids in TCAG x TCAG x TCAG order (translation table 1)
constexpr char kAaTcag[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
// our encoding e (A=0,C=1,T=2,G=3) -> position in "TCAG"
constexpr int kTcagOfE[4] = {2, 1, 0, 3};
constexpr char kAsciiOfE[4] = {'A', 'C', 'T', 'G'};

constexpr char aa_of(int code) {
  return kAaTcag[16 * kTcagOfE[(code >> 4) & 3] + 4 * kTcagOfE[(code >> 2) & 3] + kTcagOfE[code & 3]];
}
constexpr bool is_stop(int code) { return aa_of(code) == '*'; }

// bit e2 set for every third base giving the same amino acid (same first two bases)
constexpr int family_mask(int code) {
  int m = 0;
  for (int e = 0; e < 4; ++e)
    if (aa_of((code & ~3) | e) == aa_of(code)) m |= 1 << e;
  return m;
}
constexpr int popc4(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }

// annotation packed as aa | fold << 8 | class << 16 (three characters of the
// reference's labels, e.g. 'L','4','N')
constexpr uint32_t annot(int code) {
  const int fam = family_mask(code);
  const int n = popc4(fam);
  const char fold = n > 1 ? char('0' + n) : '0';
  // e-masks: A=1, C=2, T=4, G=8
  const char cls = fam == 0x9 ? 'R' : fam == 0x6 ? 'Y' : fam == 0xF ? 'N' : fam == 0x7 ? 'H' : 'G';
  return uint32_t(uint8_t(aa_of(code))) | uint32_t(uint8_t(fold)) << 8 | uint32_t(uint8_t(cls)) << 16;
}
constexpr uint32_t lab(const char (&s)[4]) {
  return uint32_t(uint8_t(s[0])) | uint32_t(uint8_t(s[1])) << 8 | uint32_t(uint8_t(s[2])) << 16;
}
constexpr uint32_t suffix(uint32_t a) { return a >> 8; }
constexpr uint32_t suf(const char (&s)[3]) { return uint32_t(uint8_t(s[0])) | uint32_t(uint8_t(s[1])) << 8; }

constexpr int code_of(const char (&s)[4]) {
  int c = 0;
  for (int i = 0; i < 3; ++i) c = c * 4 + ((s[i] >> 1) & 3);
  return c;
}

// number of single-nucleotide neighbours with the same amino acid (stops: 0)
constexpr int syn_thirds(int code) {
  if (is_stop(code)) return 0;
  int n = 0;
  for (int p = 0; p < 3; ++p) {
    const int sh = 4 - 2 * p;
    for (int e = 0; e < 4; ++e) {
      const int nb = (code & ~(3 << sh)) | (e << sh);
      if (nb != code && aa_of(nb) == aa_of(code)) ++n;
    }
  }
  return n;
}

// ---- two-codon rule (PolyFastA.py:347-414) -----------------------------------
constexpr bool both_in(uint32_t a, uint32_t b, uint32_t x, uint32_t y) {
  return (a == x || a == y) && (b == x || b == y);
}
constexpr bool any_is(uint32_t a, uint32_t b, uint32_t x) { return a == x || b == x; }

// third position varies together with another one (PolyFastA.py:363-396)
constexpr bool third_pos_syn(int c1, int c2) {
  const uint32_t a = annot(c1), b = annot(c2);
  const uint32_t sa = suffix(a), sb = suffix(b);
  if (sa == suf("4N") || sb == suf("4N")) return true;
  if (sa == suf("2Y") && sb == suf("2Y")) return true;
  if (sa == suf("2R") && sb == suf("2R")) return true;
  if (any_is(a, b, lab("I3H"))) {
    if (sa == suf("2Y") || sb == suf("2Y")) return true;
    if (sa == suf("2R") || sb == suf("2R")) {
      constexpr int excl[6] = {code_of("AAG"), code_of("AGG"), code_of("GAG"),
                               code_of("TTG"), code_of("ATT"), code_of("ATC")};
      bool in1 = false, in2 = false;
      for (int i = 0; i < 6; ++i) { in1 = in1 || c1 == excl[i]; in2 = in2 || c2 == excl[i]; }
      if (!(in1 && in2)) return true;
    }
  }
  if (any_is(a, b, lab("L2R")) && any_is(a, b, lab("W0G"))) return true;
  if (any_is(a, b, lab("M0G")) &&
      (any_is(a, b, lab("R2R")) || any_is(a, b, lab("K2R")) || any_is(a, b, lab("L2R"))))
    return true;
  return both_in(a, b, lab("R2R"), lab("H2Y")) || both_in(a, b, lab("L2R"), lab("H2Y")) ||
         both_in(a, b, lab("D2Y"), lab("R2R")) || both_in(a, b, lab("F2Y"), lab("Q2R")) ||
         both_in(a, b, lab("C2Y"), lab("Q2R")) || both_in(a, b, lab("E2R"), lab("S2Y")) ||
         both_in(a, b, lab("S2Y"), lab("Q2R"));
}

// first position varies together with another one (PolyFastA.py:398-411)
constexpr bool first_pos_syn(int c1, int c2) {
  const uint32_t a = annot(c1), b = annot(c2);
  const uint32_t band[5] = {lab("P4N"), lab("L4N"), lab("R4N"), lab("H2Y"), lab("Q2R")};
  bool in_band = false;
  for (int i = 0; i < 5; ++i) in_band = in_band || a == band[i] || b == band[i];
  if ((any_is(a, b, lab("L2R")) || any_is(a, b, lab("R2R"))) && in_band) return true;
  if (any_is(a, b, lab("L4N")) &&
      (c1 == code_of("TCA") || c1 == code_of("TCG") || c2 == code_of("TCA") || c2 == code_of("TCG")))
    return true;  // (the reference's "T0G" test can never match: no such label)
  if (any_is(a, b, lab("R4N"))) {
    constexpr int part[6] = {code_of("ATG"), code_of("ATA"), code_of("ACA"),
                             code_of("ACG"), code_of("AAA"), code_of("AAG")};
    for (int i = 0; i < 6; ++i)
      if (c1 == part[i] || c2 == part[i]) return true;
  }
  return false;
}

// bits 0-2: synonymous positions, bits 3-5: non-synonymous; 0 unless both sense and distinct
constexpr uint8_t pair_rule(int c1, int c2) {
  if (c1 == c2 || is_stop(c1) || is_stop(c2)) return 0;
  bool v[3] = {((c1 ^ c2) >> 4 & 3) != 0, ((c1 ^ c2) >> 2 & 3) != 0, ((c1 ^ c2) & 3) != 0};
  const int nv = int(v[0]) + int(v[1]) + int(v[2]);
  const uint32_t a = annot(c1), b = annot(c2);
  unsigned s = 0, n = 0;
  if (nv == 1) {                                                   // PolyFastA.py:349-358
    const int p = v[0] ? 0 : v[1] ? 1 : 2;
    const bool syn = a == b ||
                     (p == 0 && (both_in(a, b, lab("L4N"), lab("L2R")) || both_in(a, b, lab("R4N"), lab("R2R"))));
    (syn ? s : n) |= 1u << p;
  } else {                                                         // PolyFastA.py:360-414
    if (v[2]) (third_pos_syn(c1, c2) ? s : n) |= 4u;
    if (v[0]) (first_pos_syn(c1, c2) ? s : n) |= 1u;
    if (v[1]) n |= 2u;
  }
  return uint8_t(s | n << 3);
}

struct Tables {
  uint8_t pair[64 * 64];  // pair_rule(c1, c2)
  uint8_t annot_id[64];   // dense id (< 32) of the annotation label (equal labels <=> equal ids); 255 for stops
  uint8_t annot_info[64]; // id | 0x20 if the label is L4N or L2R | 0x40 if R4N or R2R; 255 for stops
  uint8_t syn3[64];       // syn_thirds
  uint64_t stop_mask;
  uint64_t t_mask[5];     // t_mask[v]: codons whose syn_thirds == v (v = 1..4)
  uint64_t same_annot[64]; // sense codons carrying the same annotation label as codon c (0 for stops)
};

constexpr Tables make_tables() {
  Tables t{};
  for (int a = 0; a < 64; ++a)
    for (int b = 0; b < 64; ++b) t.pair[a * 64 + b] = pair_rule(a, b);
  int next = 0;
  for (int c = 0; c < 64; ++c) {
    t.syn3[c] = uint8_t(syn_thirds(c));
    if (is_stop(c)) {
      t.annot_id[c] = 255;
      t.stop_mask |= 1ull << c;
      continue;
    }
    t.t_mask[syn_thirds(c)] |= 1ull << c;
    int id = -1;
    for (int d = 0; d < c; ++d)
      if (!is_stop(d) && annot(d) == annot(c)) { id = t.annot_id[d]; break; }
    t.annot_id[c] = uint8_t(id >= 0 ? id : next++);
  }
  for (int c = 0; c < 64; ++c)
    for (int d = 0; d < 64; ++d)
      if (!is_stop(c) && !is_stop(d) && t.annot_id[c] == t.annot_id[d]) t.same_annot[c] |= 1ull << d;
  for (int c = 0; c < 64; ++c) {
    if (is_stop(c)) { t.annot_info[c] = 255; continue; }
    const uint32_t a = annot(c);
    const bool leu = a == lab("L4N") || a == lab("L2R"), arg = a == lab("R4N") || a == lab("R2R");
    t.annot_info[c] = uint8_t(t.annot_id[c] | (leu ? 0x20 : 0) | (arg ? 0x40 : 0));
  }
  return t;
}

}  // namespace ct

// ---- run-time helpers ---------------------------------------------------------
PFA_HD constexpr inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
PFA_HD constexpr inline int ctz64(uint64_t x) { return __builtin_ctzll(x); }  // x != 0

// 2-bit code of an ASCII base and whether the byte is exactly A, C, G or T
PFA_HD inline uint32_t base_code(uint32_t x) { return (x >> 1) & 3u; }
PFA_HD inline bool base_clean(uint32_t x) {
  // bytes of 0x47544341 are 'A','C','T','G' in code order
  return ((0x47544341u >> (8u * base_code(x))) & 0xFFu) == x;
}

// Classifier on the set of distinct SENSE codons of one codon column
// (PolyFastA.py:331-434).  `T` is the table block (host copy or device copy);
// `info` = T.annot_info or a faster copy of it (the kernels keep one in LDS, so
// that the common cases touch no global memory).
PFA_HD constexpr inline uint32_t classify_sense_mask(uint64_t ms, const ct::Tables& T, const uint8_t* info) {
  const int k = popc64(ms);
  if (k < 2) return 0;                                            // PolyFastA.py:333
  if (k == 2) {
    const int c1 = ctz64(ms);
    const int c2 = ctz64(ms & (ms - 1));
    const uint32_t x = uint32_t(c1 ^ c2);
    const bool d0 = (x & 0x30u) != 0u, d1 = (x & 0x0Cu) != 0u, d2 = (x & 0x03u) != 0u;
    if (int(d0) + int(d1) + int(d2) == 1) {                       // one position differs: PolyFastA.py:349-358
      const uint32_t i1 = info[c1], i2 = info[c2];
      const bool syn = i1 == i2 || (d0 && (i1 & i2 & 0x60u) != 0u);
      const uint32_t bit = d0 ? 1u : d1 ? 2u : 4u;
      return syn ? bit : bit << 3;
    }
    return T.pair[c1 * 64 + c2];                                  // PolyFastA.py:360-414
  }
  // three or more distinct sense codons                          PolyFastA.py:415-432
  uint32_t bm0 = 0, bm1 = 0, bm2 = 0;
  uint64_t lo = 0, hi = 0;  // 4-bit multiplicity counters per annotation id (a label has at most 6 codons)
  uint32_t maxmult = 0;
  for (uint64_t m = ms; m; m &= m - 1) {
    const int c = ctz64(m);
    bm0 |= 1u << (c >> 4);
    bm1 |= 1u << ((c >> 2) & 3);
    bm2 |= 1u << (c & 3);
    const uint32_t id = info[c] & 0x1Fu;
    const uint32_t sh = (id & 15u) * 4u;
    uint32_t v = 0;
    if (id < 16u) { lo += 1ull << sh; v = uint32_t(lo >> sh) & 15u; }
    else          { hi += 1ull << sh; v = uint32_t(hi >> sh) & 15u; }
    maxmult = v > maxmult ? v : maxmult;
  }
  const uint32_t nb0 = popc64(bm0), nb1 = popc64(bm1), nb2 = popc64(bm2);
  const uint32_t var = (nb0 > 1 ? 1u : 0u) | (nb1 > 1 ? 2u : 0u) | (nb2 > 1 ? 4u : 0u);
  if (maxmult < 2) return var << 3;                               // all labels distinct: PolyFastA.py:430-432
  const uint32_t last = (var & 4u) ? 2u : (var & 2u) ? 1u : 0u;   // vcp[-1]
  const uint32_t nb_last = last == 2 ? nb2 : last == 1 ? nb1 : nb0;
  const uint32_t s = maxmult >= nb_last ? (1u << last) : 0u;      // PolyFastA.py:426-427
  const uint32_t n = var & ((1u << last) - 1u);                   // PolyFastA.py:428-429
  return s | n << 3;
}

// sum over the codons of a set of their synonymous-neighbour counts
PFA_HD constexpr inline uint32_t syn_thirds_of_mask(uint64_t m, const ct::Tables& T) {
  return popc64(m & T.t_mask[1]) + 2 * popc64(m & T.t_mask[2]) + 3 * popc64(m & T.t_mask[3]) +
         4 * popc64(m & T.t_mask[4]);
}


// ---- codon columns with ONE variable position ----------------------------------
// Their codon set is the consensus codon with each A/C/G/T allele of that position
// substituted, so everything the scan needs is a function of (position p, the two fixed
// bases, the 4-bit allele mask): entry [(p * 16 + b4) * 16 + am], b4 = the two fixed 2-bit
// codes in codon order, holds  synonymous-neighbour sum (bits 0-4) | stop << 5 |
// position called synonymous << 6 | non-synonymous << 7.  Generated by running the general
// classifier above on every such set (PolyFastA.py:301-315,319-434).
struct SingleLut {
  uint8_t v[3 * 16 * 16];
};
constexpr SingleLut make_single_lut() {
  const ct::Tables T = ct::make_tables();
  SingleLut o{};
  for (uint32_t p = 0; p < 3; ++p)
    for (uint32_t b4 = 0; b4 < 16; ++b4)
      for (uint32_t am = 0; am < 16; ++am) {
        const uint32_t sh = 4u - 2u * p;
        const uint32_t hi = b4 >> 2, lo = b4 & 3u;
        const uint32_t base = p == 0 ? (hi << 2 | lo) : p == 1 ? (hi << 4 | lo) : (hi << 4 | lo << 2);
        uint64_t m = 0;
        for (uint32_t e = 0; e < 4; ++e)
          if ((am >> e) & 1u) m |= 1ull << (base | (e << sh));
        const uint64_t ms = m & ~T.stop_mask;
        const uint32_t labels = popc64(ms) >= 2 ? classify_sense_mask(ms, T, T.annot_info) : 0u;
        o.v[(p * 16u + b4) * 16u + am] =
            uint8_t(syn_thirds_of_mask(m, T) | ((m & T.stop_mask) ? 0x20u : 0u) |
                    (((labels >> p) & 1u) << 6) | (((labels >> (3u + p)) & 1u) << 7));
      }
  return o;
}

}  // namespace pfa
