/* pfa_oracle.c — CPU oracle, C restatement of the PolyFastA scan for sizes the
 * pure-Python oracle cannot finish in seconds, and the timed `cpu_baseline`
 * ("port") of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under polyfasta_amd/ links or loads this.
 * Scalar, single-threaded, one pass per site column with a 256-bin symbol
 * histogram — the reference's per-site loop (PolyFastA.py:255-260) in C.
 * Pinned by tests/test_oracle_golden.py against the fixtures frozen from the
 * reference (tests/golden/), through oracle/c_oracle.py.
 *
 * The genetic-code tables (pair classification, annotation ids, synonymous
 * weights, stop set) are NOT built here: oracle/c_oracle.py derives them from
 * the pinned Python restatement (oracle/polyfasta_oracle.py) and passes them in,
 * so this file and the product's pfa_codon.h share no table code.
 * Codon index = 16*e0 + 4*e1 + e2 with e: A=0, C=1, G=2, T=3.
 */
#include <stdint.h>
#include <string.h>

typedef struct {
  uint64_t H, H_S, H_N;
  uint32_t S, sfs_bad, seg_S, seg_N, nstops, missing_cols;
  double count_syn; /* PolyFastA.py:307, accumulated in column order */
} oracle_out;

typedef struct {
  const uint8_t* pair;     /* [64*64]: bits 0-2 syn positions, bits 3-5 nonsyn (PolyFastA.py:347-414) */
  const uint8_t* annot_id; /* [64]: label id, 255 for stops                                          */
  const uint8_t* syn3;     /* [64]: 3 x syncodfreq (PolyFastA.py:536-557)                             */
} oracle_tables;

static int base_index(uint8_t c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

/* PolyFastA.py:415-432 — three or more distinct sense codons */
static unsigned classify_many(const int* cod, int k, const oracle_tables* T) {
  int mult[256];
  unsigned bases[3] = {0, 0, 0};
  int maxmult = 0;
  memset(mult, 0, sizeof(mult));
  for (int i = 0; i < k; ++i) {
    int m = ++mult[T->annot_id[cod[i]]];
    if (m > maxmult) maxmult = m;
    bases[0] |= 1u << (cod[i] >> 4);
    bases[1] |= 1u << ((cod[i] >> 2) & 3);
    bases[2] |= 1u << (cod[i] & 3);
  }
  int nb[3], var[3], last = -1;
  for (int p = 0; p < 3; ++p) {
    nb[p] = __builtin_popcount(bases[p]);
    var[p] = nb[p] > 1;
    if (var[p]) last = p;
  }
  unsigned s = 0, n = 0;
  if (maxmult >= 2) {
    if (maxmult >= nb[last]) s |= 1u << last;
    for (int p = 0; p < last; ++p)
      if (var[p]) n |= 1u << p;
  } else {
    for (int p = 0; p < 3; ++p)
      if (var[p]) n |= 1u << p;
  }
  return s | n << 3;
}

/* One scan unit.  rows == NULL means rows 0..n_rows-1 of the matrix.
 * sfs: n_rows/2 bins (zeroed here).  h_site_out (optional, n_sites entries)
 * receives N^2 - sum c^2 per site (0 for monomorphic sites).
 * labels_out (optional, n_sites bytes): 1 = synonymous, 2 = non-synonymous. */
int pfa_oracle_scan(const uint8_t* mat, uint32_t n_rows, uint32_t n_sites, uint64_t row_stride,
                    const uint32_t* rows, int cds, const oracle_tables* T, oracle_out* out,
                    uint32_t* sfs, uint32_t* h_site_out, uint8_t* labels_out) {
  memset(out, 0, sizeof(*out));
  const uint32_t nbins = n_rows / 2;
  if (sfs) memset(sfs, 0, (size_t)nbins * sizeof(uint32_t));
  if (labels_out) memset(labels_out, 0, n_sites);
  const uint64_t nn = (uint64_t)n_rows * n_rows;
  uint64_t hcod[3] = {0, 0, 0};     /* per-site H leaves 32 bits above ~92 000 rows */

  for (uint32_t p = 0; p < n_sites; ++p) {
    /* getvarsites PolyFastA.py:255-260 + nucleotide_diversity :485-492 + getsfs :274-282 */
    uint32_t cnt[256];
    memset(cnt, 0, sizeof(cnt));
    for (uint32_t r = 0; r < n_rows; ++r) {
      const uint64_t row = rows ? rows[r] : r;
      cnt[mat[row * row_stride + p]]++;
    }
    uint64_t sumsq = 0;
    int distinct = 0;
    for (int c = 0; c < 256; ++c)
      if (cnt[c]) { ++distinct; sumsq += (uint64_t)cnt[c] * cnt[c]; }
    uint64_t hs = 0;
    if (distinct > 1) {
      hs = nn - sumsq;
      out->S += 1;
      out->H += hs;
      uint32_t a[4] = {cnt['A'], cnt['C'], cnt['G'], cnt['T']};
      for (int i = 0; i < 3; ++i)       /* sort descending */
        for (int j = i + 1; j < 4; ++j)
          if (a[j] > a[i]) { uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
      if (a[1] == 0) out->sfs_bad += 1; /* fewer than two A/C/G/T alleles: IndexError in the reference */
      else if (sfs) sfs[a[1] - 1] += 1;
    }
    if (h_site_out) h_site_out[p] = hs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hs;   /* (saturates: callers use it as a variable-site mask below 65 536 rows) */
    if (!cds) continue;

    /* getvarCDSsites PolyFastA.py:284-315 on the codon column ending here */
    hcod[p % 3] = hs;
    const int last_of_codon = (p % 3 == 2);
    const int last_site = (p + 1 == n_sites);
    if (!last_of_codon && !last_site) continue;
    if (!last_of_codon) {            /* partial last codon: never matches the regex (:301) */
      out->missing_cols += 1;
      continue;
    }
    const uint32_t cp = p - 2;
    uint64_t present = 0;            /* distinct clean codons, bit = codon index */
    for (uint32_t r = 0; r < n_rows; ++r) {
      const uint64_t row = rows ? rows[r] : r;
      const uint8_t* q = mat + row * row_stride + cp;
      const int e0 = base_index(q[0]), e1 = base_index(q[1]), e2 = base_index(q[2]);
      if (e0 >= 0 && e1 >= 0 && e2 >= 0) present |= 1ull << (16 * e0 + 4 * e1 + e2);
    }
    int cod[64], k = 0, sense[64], ks = 0, has_stop = 0;
    for (int c = 0; c < 64; ++c)
      if ((present >> c) & 1) {
        cod[k++] = c;
        if (T->annot_id[c] == 255) has_stop = 1; else sense[ks++] = c;
      }
    if (k == 0) { out->missing_cols += 1; continue; }
    out->nstops += has_stop;         /* :293 (stops are clean codons, so they are in `present`) */
    double s3 = 0.0;
    for (int i = 0; i < k; ++i) s3 += T->syn3[cod[i]] / 3.0;
    out->count_syn += s3 / k;        /* :307 */
    if (k < 2 || ks < 2) continue;   /* :309, :333 */
    const unsigned lab = ks == 2 ? T->pair[sense[0] * 64 + sense[1]] : classify_many(sense, ks, T);
    for (int q = 0; q < 3; ++q) {
      if ((lab >> q) & 1) { out->seg_S += 1; out->H_S += hcod[q]; if (labels_out) labels_out[cp + q] = 1; }
      if ((lab >> (3 + q)) & 1) { out->seg_N += 1; out->H_N += hcod[q]; if (labels_out) labels_out[cp + q] = 2; }
    }
  }
  return 0;
}

/* timing helper for bench.py: scan `n_mat` consecutive non-coding matrices */
uint64_t pfa_oracle_scan_many(const uint8_t* bases, uint32_t n_mat, uint32_t n_rows, uint32_t n_sites,
                              uint64_t row_stride, uint32_t* s_out, uint64_t* h_out) {
  uint64_t total = 0;
  oracle_out o;
  for (uint32_t m = 0; m < n_mat; ++m) {
    pfa_oracle_scan(bases + (uint64_t)m * n_rows * row_stride, n_rows, n_sites, row_stride, 0, 0, 0, &o, 0, 0, 0);
    if (s_out) s_out[m] = o.S;
    if (h_out) h_out[m] = o.H;
    total += o.S;
  }
  return total;
}
