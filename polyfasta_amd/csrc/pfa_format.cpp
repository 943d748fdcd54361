// pfa_format.cpp — CSV rows of the non-coding scan, formatted natively.
//
// print_result.no_header (PolyFastA.py:182-198) prints one row per scan unit with
// Python's repr() of three doubles; at 10^5 rows per batch that formatting, not the
// scan, is what the host spends its time on.  This file reproduces, bit for bit:
//   * polymorphism()           PolyFastA.py:502-520  (same expressions, same operand order:
//                              every intermediate is an IEEE double, as in CPython)
//   * repr(float)              the shortest digit string that round-trips
//                              (std::to_chars), laid out by CPython's rule: fixed
//                              notation for 1e-4 <= |x| < 1e16, else d.ddde+XX with at
//                              least two exponent digits; ".0" appended to integers
// Rows the reference would answer with a traceback (math domain errors) are not
// formatted here: the call reports the first such row and the caller takes the Python path.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/pfa.h"

namespace pfa {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// repr(x) of CPython 3 (float_repr_style == "short"); returns the length written (<= 32)
static int repr_double(double x, char* out) {
  if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
  if (std::isinf(x)) { const char* s = x < 0 ? "-inf" : "inf"; const int n = (int)strlen(s); memcpy(out, s, n); return n; }
  char* o = out;
  if (std::signbit(x)) { *o++ = '-'; x = -x; }
  if (x == 0.0) { memcpy(o, "0.0", 3); return int(o - out) + 3; }
  char sci[40];
  const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
  // split into digits and decimal exponent
  char digits[24];
  int nd = 0;
  const char* p = sci;
  for (; p < r.ptr && *p != 'e'; ++p)
    if (*p != '.') digits[nd++] = *p;
  ++p;   // 'e'
  const bool eneg = *p == '-';
  ++p;
  int e10 = 0;
  for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
  if (eneg) e10 = -e10;
  const int decpt = e10 + 1;   // position of the decimal point relative to the digit string
  if (decpt > -4 && decpt <= 16) {   // fixed notation
    if (decpt <= 0) {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -decpt; ++i) *o++ = '0';
      memcpy(o, digits, nd);
      o += nd;
    } else if (decpt >= nd) {
      memcpy(o, digits, nd);
      o += nd;
      for (int i = nd; i < decpt; ++i) *o++ = '0';
      *o++ = '.';
      *o++ = '0';
    } else {
      memcpy(o, digits, decpt);
      o += decpt;
      *o++ = '.';
      memcpy(o, digits + decpt, nd - decpt);
      o += nd - decpt;
    }
  } else {   // exponent notation
    *o++ = digits[0];
    if (nd > 1) {
      *o++ = '.';
      memcpy(o, digits + 1, nd - 1);
      o += nd - 1;
    }
    *o++ = 'e';
    int e = decpt - 1;
    *o++ = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    char eb[8];
    int ne = 0;
    do { eb[ne++] = char('0' + e % 10); e /= 10; } while (e);
    if (ne < 2) eb[ne++] = '0';
    while (ne) *o++ = eb[--ne];
  }
  return int(o - out);
}

static int put_u64(unsigned long long v, char* out) {
  const auto r = std::to_chars(out, out + 24, v);
  return int(r.ptr - out);
}

}  // namespace pfa

extern "C" {

int pfa_format_double(double x, char* out32) {
  if (!out32) { pfa::set_error("pfa_format_double: out is NULL"); return PFA_ERR_ARG; }
  return pfa::repr_double(x, out32);
}

// "S,pi,theta,D" of polymorphism() (PolyFastA.py:502-520) appended at out; returns the bytes
// written, or -1 when the reference would raise (math domain error in Dvar, or seqlen == 0)
static int put_polymorphism(unsigned long long S, unsigned long long H, uint32_t n_rows, double seqlen,
                            double a1, double e1, double e2, int jc, char* out) {
  if (S == 0) {                                            // PolyFastA.py:503-504, 436-437
    memcpy(out, "0,0,0,NA", 8);
    return 8;
  }
  if (n_rows < 2u || seqlen == 0.0) return -1;
  const double ss = (double)S;
  const double th_pi = (double)H / (double)((unsigned long long)n_rows * (n_rows - 1u));
  const double th_wa = ss / a1;
  const double rad = (e1 * ss) + (e2 * ss * (double)(S - 1ull));   // PolyFastA.py:534
  if (!(rad >= 0.0)) return -1;                            // math.sqrt would raise
  const double sd = std::sqrt(rad);
  double pi_site = th_pi / seqlen;
  if (jc) {                                                // PolyFastA.py:512-516 (bare except -> uncorrected value)
    const double arg = 1.0 - (4.0 / 3.0) * pi_site;
    if (arg > 0.0) pi_site = -0.75 * std::log(arg);
  }
  int w = pfa::put_u64(S, out);
  out[w++] = ',';
  w += pfa::repr_double(pi_site, out + w);
  out[w++] = ',';
  w += pfa::repr_double(th_wa / seqlen, out + w);
  out[w++] = ',';
  if (sd == 0.0) {                                         // ZeroDivisionError -> "NA" (PolyFastA.py:508-511)
    out[w++] = 'N';
    out[w++] = 'A';
  } else {
    w += pfa::repr_double((th_pi - th_wa) / sd, out + w);
  }
  return w;
}

// round(x, 2) of CPython: the correctly rounded two-decimal string, read back as a double
static double py_round2(double x) {
  char buf[400];
  snprintf(buf, sizeof(buf), "%.2f", x);
  return strtod(buf, nullptr);
}

int64_t pfa_format_plain_rows(const char* prefix_blob, const uint64_t* prefix_off, uint32_t n,
                              const uint32_t* seg_sites, const uint64_t* h_sum, const uint32_t* n_rows, const double* seqlen,
                              const double* a1, const double* e1, const double* e2, int jc,
                              char* out, uint64_t cap, int64_t* bad_row) {
  if (bad_row) *bad_row = -1;
  if (n && (!prefix_blob || !prefix_off || !seg_sites || !h_sum || !n_rows || !seqlen || !a1 || !e1 || !e2 || !out)) {
    pfa::set_error("pfa_format_plain_rows: NULL argument");
    return PFA_ERR_ARG;
  }
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t plen = prefix_off[i + 1] - prefix_off[i];
    if (w + plen + 128u > cap) { pfa::set_error("pfa_format_plain_rows: output capacity %llu too small", (unsigned long long)cap); return PFA_ERR_ARG; }
    memcpy(out + w, prefix_blob + prefix_off[i], plen);   // "file,seqlen,pop,N,"
    w += plen;
    const int k = put_polymorphism(seg_sites[i], h_sum[i], n_rows[i], seqlen[i], a1[i], e1[i], e2[i], jc, out + w);
    if (k < 0) {   // the caller formats this batch in Python (and gets the reference's exception)
      if (bad_row) *bad_row = i;
      pfa::set_error("pfa_format_plain_rows: row %u has no finite result", i);
      return PFA_ERR_ARG;
    }
    w += (uint64_t)k;
    out[w++] = '\n';
  }
  return (int64_t)w;
}

// synonymous sites of a row: sum over k of bin[k] / (3 k), ascending k (PolyFastA.py:307) — or the caller's replay of the
// reference's own accumulation (a row on a two-decimal rounding tie, pfa_plan_codon_sets)
static double syn_sites(const uint32_t* bins, double given) {
  if (!std::isnan(given)) return given;
  double s = 0.0;
  for (uint32_t k = 1; k < PFA_SYN_BINS; ++k)
    if (bins[k]) s += (double)bins[k] / (3.0 * (double)k);
  return s;
}

int64_t pfa_format_cds_rows(const char* name_blob, const uint64_t* name_off, const char* mid_blob, const uint64_t* mid_off,
                            uint32_t n, const pfa_result* res, const uint32_t* syn_num, const double* sites_s, const uint32_t* n_rows,
                            const uint32_t* seqlen, const double* a1, const double* e1, const double* e2, int jc,
                            char* out, uint64_t cap, int64_t* bad_row) {
  if (bad_row) *bad_row = -1;
  if (n && (!name_blob || !name_off || !mid_blob || !mid_off || !res || !syn_num || !n_rows || !seqlen || !a1 || !e1 || !e2 || !out)) {
    pfa::set_error("pfa_format_cds_rows: NULL argument");
    return PFA_ERR_ARG;
  }
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t nlen = name_off[i + 1] - name_off[i], mlen = mid_off[i + 1] - mid_off[i];
    if (w + nlen + mlen + 320u > cap) { pfa::set_error("pfa_format_cds_rows: output capacity %llu too small", (unsigned long long)cap); return PFA_ERR_ARG; }
    // the synonymous sites; the rest are non-synonymous (PolyFastA.py:154)
    const double ssites = syn_sites(syn_num + (uint64_t)i * PFA_SYN_BINS, sites_s ? sites_s[i] : NAN);
    const double nsites = (double)((long long)seqlen[i] - 3ll * (long long)res[i].missing_cols) - ssites;
    memcpy(out + w, name_blob + name_off[i], nlen);       // "file,"
    w += nlen;
    w += pfa::repr_double(py_round2(ssites), out + w);
    out[w++] = ',';
    w += pfa::repr_double(py_round2(nsites), out + w);
    memcpy(out + w, mid_blob + mid_off[i], mlen);         // ",pop,N,"
    w += mlen;
    if (res[i].seg_sites == 0) {                           // PolyFastA.py:160-162 (the mis-ordered no-variation row)
      memcpy(out + w, "0,0,0,NA,0,0,0,NA,0\n", 20);
      w += 20;
      continue;
    }
    char ps[128], pn[128];
    const int ks = put_polymorphism(res[i].seg_S, res[i].H_S, n_rows[i], ssites, a1[i], e1[i], e2[i], jc, ps);
    const int kn = put_polymorphism(res[i].seg_N, res[i].H_N, n_rows[i], nsites, a1[i], e1[i], e2[i], jc, pn);
    if (ks < 0 || kn < 0) {
      if (bad_row) *bad_row = i;
      pfa::set_error("pfa_format_cds_rows: row %u has no finite result", i);
      return PFA_ERR_ARG;
    }
    ps[ks] = pn[kn] = ',';   // sentinels: the fields of both tuples are interleaved (PolyFastA.py:176)
    const char *a = ps, *b = pn;
    for (int f = 0; f < 4; ++f) {
      while (*a != ',') out[w++] = *a++;
      ++a;
      out[w++] = ',';
      while (*b != ',') out[w++] = *b++;
      ++b;
      out[w++] = ',';
    }
    w += (uint64_t)pfa::put_u64(res[i].nstops, out + w);
    out[w++] = '\n';
  }
  return (int64_t)w;
}

int64_t pfa_format_rows_indexed(const char* names_blob, const uint64_t* names_off, const char* keys_blob, const uint64_t* keys_off,
                                uint32_t n, const uint32_t* file_idx, const uint32_t* key, const uint32_t* n_rows, const uint32_t* n_sites,
                                const pfa_result* res, const uint32_t* syn_num, const double* sites_s, const double* a1, const double* e1,
                                const double* e2, int jc, int coding, char* out, uint64_t cap, int64_t* bad_row) {
  if (bad_row) *bad_row = -1;
  if (n && (!names_blob || !names_off || !file_idx || !key || !n_rows || !n_sites || !res || !a1 || !e1 || !e2 || !out || (coding && !syn_num))) {
    pfa::set_error("pfa_format_rows_indexed: NULL argument");
    return PFA_ERR_ARG;
  }
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t nb = names_off[file_idx[i]], nlen = names_off[file_idx[i] + 1] - nb;
    const char* pop = "NA";
    uint64_t plen = 2;
    if (key[i] != PFA_NO_KEY) {
      if (!keys_blob || !keys_off) { pfa::set_error("pfa_format_rows_indexed: row %u names a key but there are none", i); return PFA_ERR_ARG; }
      pop = keys_blob + keys_off[key[i]];
      plen = keys_off[key[i] + 1] - keys_off[key[i]];
    }
    if (w + nlen + plen + 400u > cap) { pfa::set_error("pfa_format_rows_indexed: output capacity %llu too small", (unsigned long long)cap); return PFA_ERR_ARG; }
    memcpy(out + w, names_blob + nb, nlen);
    w += nlen;
    out[w++] = ',';
    if (!coding) {                                          // file,seqlen,pop,N,S,pi,theta,D  (PolyFastA.py:198)
      w += (uint64_t)pfa::put_u64(n_sites[i], out + w);
      out[w++] = ',';
      memcpy(out + w, pop, plen);
      w += plen;
      out[w++] = ',';
      w += (uint64_t)pfa::put_u64(n_rows[i], out + w);
      out[w++] = ',';
      const int k = put_polymorphism(res[i].seg_sites, res[i].H, n_rows[i], (double)n_sites[i], a1[i], e1[i], e2[i], jc, out + w);
      if (k < 0) {
        if (bad_row) *bad_row = i;
        pfa::set_error("pfa_format_rows_indexed: row %u has no finite result", i);
        return PFA_ERR_ARG;
      }
      w += (uint64_t)k;
      out[w++] = '\n';
      continue;
    }
    // file,sites_S,sites_N,pop,N,... (PolyFastA.py:151-180; the same arithmetic as pfa_format_cds_rows)
    const double ssites = syn_sites(syn_num + (uint64_t)i * PFA_SYN_BINS, sites_s ? sites_s[i] : NAN);
    const double nsites = (double)((long long)n_sites[i] - 3ll * (long long)res[i].missing_cols) - ssites;
    w += pfa::repr_double(py_round2(ssites), out + w);
    out[w++] = ',';
    w += pfa::repr_double(py_round2(nsites), out + w);
    out[w++] = ',';
    memcpy(out + w, pop, plen);
    w += plen;
    out[w++] = ',';
    w += (uint64_t)pfa::put_u64(n_rows[i], out + w);
    out[w++] = ',';
    if (res[i].seg_sites == 0) {                           // PolyFastA.py:160-162
      memcpy(out + w, "0,0,0,NA,0,0,0,NA,0\n", 20);
      w += 20;
      continue;
    }
    char ps[128], pn[128];
    const int ks = put_polymorphism(res[i].seg_S, res[i].H_S, n_rows[i], ssites, a1[i], e1[i], e2[i], jc, ps);
    const int kn = put_polymorphism(res[i].seg_N, res[i].H_N, n_rows[i], nsites, a1[i], e1[i], e2[i], jc, pn);
    if (ks < 0 || kn < 0) {
      if (bad_row) *bad_row = i;
      pfa::set_error("pfa_format_rows_indexed: row %u has no finite result", i);
      return PFA_ERR_ARG;
    }
    ps[ks] = pn[kn] = ',';
    const char *a = ps, *b = pn;
    for (int f = 0; f < 4; ++f) {
      while (*a != ',') out[w++] = *a++;
      ++a;
      out[w++] = ',';
      while (*b != ',') out[w++] = *b++;
      ++b;
      out[w++] = ',';
    }
    w += (uint64_t)pfa::put_u64(res[i].nstops, out + w);
    out[w++] = '\n';
  }
  return (int64_t)w;
}

}  // extern "C"
