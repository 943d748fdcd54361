// host_fuzz.cpp — sanitizer harness for the HOST code of libpfa_hip.so (test infrastructure).
// Built by tests/test_host_sanitizers.py with plain g++ -fsanitize=address,undefined from
// csrc/pfa_fasta.cpp + csrc/pfa_format.cpp (no HIP needed) and run on random FASTA-like
// byte strings, truncated inputs, files on disk and random row tables: the packer reads
// arbitrary user files, so every out-of-bounds access there is a bug whatever the return code.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <cmath>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../include/pfa.h"

// the two symbols the host translation units expect from pfa_scan.hip
static thread_local char g_err[512];
namespace pfa {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pfa
extern "C" const char* pfa_last_error(void) { return g_err; }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t below(uint32_t n) { return n ? uint32_t(rnd() % n) : 0u; }

static std::string random_fasta() {
  static const char* eols[] = {"\n", "\r\n", "\r", "\n\n", " \n", "\t\r\n"};
  static const char alphabet[] = "ACGTacgtNn-RYKM?*. \t";
  std::string s;
  const uint32_t style = below(8);
  if (style == 0) s += "junk before\n";
  if (style == 1) return std::string(below(40), char(below(256)));
  const uint32_t n_rec = below(7);
  for (uint32_t r = 0; r < n_rec; ++r) {
    s += '>';
    if (below(6)) s += "seq" + std::to_string(below(4));          // duplicates on purpose
    if (below(9) == 0) s += char(128 + below(128));
    s += eols[below(6)];
    const uint32_t n_lines = below(4);
    for (uint32_t l = 0; l < n_lines; ++l) {
      const uint32_t len = below(4) ? below(70) : below(3);
      for (uint32_t k = 0; k < len; ++k) s += (below(300) == 0) ? char(128 + below(128)) : alphabet[below(sizeof(alphabet) - 1)];
      if (r + 1 < n_rec || l + 1 < n_lines || below(2)) s += eols[below(6)];
    }
  }
  if (style == 2 && !s.empty()) s.resize(below((uint32_t)s.size()));   // truncated anywhere
  return s;
}

// pfa::valid_utf8 (pfa_fasta.cpp: which files the loader hands back because the reference would die decoding them, PolyFastA.py:239-240)
// against a validator written the other way round — decode the code point, then judge it: shortest form, no surrogates, at most
// U+10FFFF (what CPython's strict UTF-8 decoder accepts).
namespace pfa { bool valid_utf8(const uint8_t* p, uint64_t n); }
static bool utf8_by_decoding(const uint8_t* p, size_t n) {
  size_t i = 0;
  while (i < n) {
    const uint32_t c = p[i];
    uint32_t len, cp;
    if (c < 0x80) { ++i; continue; }
    else if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; }
    else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; }
    else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; }
    else return false;
    if (i + len > n) return false;
    for (uint32_t k = 1; k < len; ++k) {
      if ((p[i + k] & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (p[i + k] & 0x3F);
    }
    static const uint32_t least[5] = {0, 0, 0x80, 0x800, 0x10000};
    if (cp < least[len] || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
    i += len;
  }
  return true;
}
static void fuzz_utf8() {
  uint8_t buf[24];
  const uint32_t n = below(sizeof buf + 1);
  // bytes drawn so that multi-byte sequences, their edges (E0 A0, ED 9F / A0, F0 90, F4 8F / 90) and truncations come up often
  static const uint8_t pool[] = {0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xEC, 0xED, 0xEE, 0xEF,
                                 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xFF};
  for (uint32_t i = 0; i < n; ++i) buf[i] = below(3) ? pool[below(sizeof pool)] : (uint8_t)below(256);
  uint8_t* exact = (uint8_t*)malloc(n ? n : 1);     // exact-size copy: ASan sees a read past the end of a truncated sequence
  memcpy(exact, buf, n);
  const bool a = pfa::valid_utf8(exact, n), b = utf8_by_decoding(exact, n);
  if (a != b) {
    fprintf(stderr, "valid_utf8 = %d but decoding says %d for", (int)a, (int)b);
    for (uint32_t i = 0; i < n; ++i) fprintf(stderr, " %02X", buf[i]);
    fprintf(stderr, "\n");
    abort();
  }
  free(exact);
}

static void fuzz_text(const std::string& s) {
  // exact-size heap copy: ASan sees any read past the end
  uint8_t* text = (uint8_t*)malloc(s.size() ? s.size() : 1);
  memcpy(text, s.data(), s.size());
  {   // the in-place indexer of the streaming ingest, on its own exact-size copy
    uint8_t* t2 = (uint8_t*)malloc(s.size() ? s.size() : 1);
    memcpy(t2, s.data(), s.size());
    std::vector<pfa_fasta_row> rows(s.size() / 2 + 2);
    const int64_t k = pfa_fasta_index_compact(s.size() ? t2 : nullptr, s.size(), rows.data(), rows.size());
    for (int64_t i = 0; i < k; ++i) {
      const pfa_fasta_row& r = rows[(size_t)i];
      if (r.name_off + r.name_len > s.size() || (r.seq_len && r.seq_off + r.seq_len > s.size())) { fprintf(stderr, "index_compact: record %lld out of bounds\n", (long long)i); abort(); }
      volatile uint8_t sink = 0;
      for (uint32_t b = 0; b < r.seq_len; ++b) sink ^= t2[r.seq_off + b];
    }
    free(t2);
  }
  const int64_t n = pfa_fasta_index(s.size() ? text : nullptr, s.size(), nullptr, 0);
  if (n > 0) {
    std::vector<pfa_fasta_rec> recs((size_t)n);
    const int64_t n2 = pfa_fasta_index(text, s.size(), recs.data(), (uint64_t)n);
    if (n2 != n) { fprintf(stderr, "index count changed %lld -> %lld\n", (long long)n, (long long)n2); abort(); }
    uint32_t maxlen = 0;
    for (auto& r : recs) maxlen = r.seq_len > maxlen ? r.seq_len : maxlen;
    const uint32_t stride = ((maxlen + 15u) & ~15u) + 16u * below(3);
    if (stride) {
      uint8_t* out = (uint8_t*)malloc((size_t)n * stride);
      (void)pfa_fasta_pack(text, s.size(), recs.data(), (uint64_t)n, out, stride);
      if (n > 1) (void)pfa_fasta_pack(text, s.size(), recs.data(), 1, out, stride);
      free(out);
    }
    // a too-small stride and a too-small record table must be refused, not overrun
    if (maxlen > 16) {
      uint8_t* out = (uint8_t*)malloc((size_t)n * 16);
      (void)pfa_fasta_pack(text, s.size(), recs.data(), (uint64_t)n, out, 16);
      free(out);
    }
    if (n > 1) (void)pfa_fasta_index(text, s.size(), recs.data(), (uint64_t)n - 1);
  }
  free(text);
}

static void fuzz_files(const char* dir) {
  const uint32_t nf = 1 + below(12);
  std::vector<std::string> paths, made_dirs;
  for (uint32_t i = 0; i < nf; ++i) {
    std::string p = std::string(dir) + "/f" + std::to_string(i) + ".fa";
    if (below(10) == 0) p += ".missing";
    else if (below(12) == 0) {      // a sub-directory among the files: not a regular file -> PFA_ERR_IO, no abort
      p += ".dir";
      mkdir(p.c_str(), 0700);
      made_dirs.push_back(p);
    } else {
      FILE* f = fopen(p.c_str(), "wb");
      std::string s;
      if (below(3) == 0) s = random_fasta();
      else {   // a well-formed alignment, sometimes ragged
        const uint32_t rows = 1 + below(6), len = below(300);
        for (uint32_t r = 0; r < rows; ++r) {
          s += ">r" + std::to_string(r) + "\n";
          s += std::string(len + (below(12) == 0 ? 1 : 0), "ACGT"[below(4)]) + "\n";
        }
      }
      fwrite(s.data(), 1, s.size(), f);
      fclose(f);
    }
    paths.push_back(p);
  }
  std::vector<const char*> cp;
  for (auto& p : paths) cp.push_back(p.c_str());
  const uint64_t cap = 256u * below(40);                       // often too small: files get deferred
  uint8_t* bases = (uint8_t*)malloc(cap ? cap : 1);
  std::vector<pfa_file_info> info(nf);
  const uint64_t names_cap = below(3) ? 4096 : below(16);
  char* names = (char*)malloc(names_cap ? names_cap : 1);
  uint64_t used = 0, names_used = 0;
  (void)pfa_fasta_load_files(cp.data(), nf, 16u << below(5), 1 + below(4), bases, cap, info.data(),
                             below(2) ? names : nullptr, names_cap, &used, &names_used);
  if (used > cap) { fprintf(stderr, "load_files used %llu > cap %llu\n", (unsigned long long)used, (unsigned long long)cap); abort(); }
  free(names);
  free(bases);
  for (uint32_t i = 0; i < nf; ++i)
    if (paths[i].size() > 4 && paths[i].compare(paths[i].size() - 4, 4, ".dir") == 0 && info[i].status != PFA_ERR_IO) {
      fprintf(stderr, "a directory was not reported as PFA_ERR_IO (status %d)\n", info[i].status);
      abort();
    }
  for (auto& p : paths) unlink(p.c_str());
  for (auto& p : made_dirs) rmdir(p.c_str());
}

static void fuzz_rows() {
  const uint32_t n = below(50);
  std::string blob;
  std::vector<uint64_t> off(n + 1, 0);
  std::vector<uint32_t> seg(n), nrows(n);
  std::vector<uint64_t> h(n);
  std::vector<double> seqlen(n), a1(n), e1(n), e2(n);
  for (uint32_t i = 0; i < n; ++i) {
    blob += "f" + std::to_string(below(100000)) + "," + std::to_string(below(100000)) + ",NA," + std::to_string(below(300)) + ",";
    off[i + 1] = blob.size();
    seg[i] = below(4) ? below(100000) : 0;
    nrows[i] = below(20) ? 2 + below(400) : below(3);
    h[i] = rnd() >> below(64);
    seqlen[i] = below(20) ? 1 + below(100000) : 0.0;
    a1[i] = below(20) ? 0.5 + below(1000) / 100.0 : 0.0;
    e1[i] = (int32_t(below(2001)) - 1000) / 5000.0;
    e2[i] = (int32_t(below(2001)) - 1000) / 500000.0;
  }
  const uint64_t cap = below(4) ? blob.size() + 128ull * n + 16 : below(200);
  char* out = (char*)malloc(cap ? cap : 1);
  int64_t bad = -1;
  const int64_t got = pfa_format_plain_rows(blob.data(), off.data(), n, seg.data(), h.data(), nrows.data(), seqlen.data(),
                                            a1.data(), e1.data(), e2.data(), (int)below(2), out, cap, &bad);
  if (got > (int64_t)cap) { fprintf(stderr, "format wrote %lld > cap %llu\n", (long long)got, (unsigned long long)cap); abort(); }
  free(out);
  // coding rows on the same random table
  {
    std::vector<pfa_result> res(n);
    std::vector<uint32_t> syn((size_t)n * PFA_SYN_BINS), sl(n);
    std::vector<double> ssx(n);   // optional per-row synonymous-site override: NaN (use the bins) or a value
    for (uint32_t i = 0; i < n; ++i) ssx[i] = below(3) ? NAN : below(100000) / 8.0;
    std::vector<uint64_t> noff(n + 1, 0), moff(n + 1, 0);
    std::string names, mids;
    for (uint32_t i = 0; i < n; ++i) {
      memset(&res[i], 0, sizeof(pfa_result));
      res[i].seg_sites = seg[i];
      res[i].seg_S = below(seg[i] + 1);
      res[i].seg_N = below(seg[i] + 1);
      res[i].H_S = h[i] >> 1;
      res[i].H_N = h[i] >> 2;
      res[i].nstops = below(50);
      res[i].missing_cols = below(1000);
      for (uint32_t k = 0; k < PFA_SYN_BINS; ++k) syn[(size_t)i * PFA_SYN_BINS + k] = below(4) ? 0 : below(100000);
      sl[i] = below(20) ? below(100000) : 0;
      names += "g" + std::to_string(i) + ",";
      noff[i + 1] = names.size();
      mids += ",NA," + std::to_string(nrows[i]) + ",";
      moff[i + 1] = mids.size();
    }
    const uint64_t cap2 = below(4) ? names.size() + mids.size() + 320ull * n + 16 : below(400);
    char* out2 = (char*)malloc(cap2 ? cap2 : 1);
    int64_t bad2 = -1;
    const int64_t got2 = pfa_format_cds_rows(names.data(), noff.data(), mids.data(), moff.data(), n, res.data(), syn.data(),
                                             below(2) ? ssx.data() : nullptr, nrows.data(), sl.data(), a1.data(), e1.data(), e2.data(), (int)below(2), out2, cap2, &bad2);
    if (got2 > (int64_t)cap2) { fprintf(stderr, "cds format wrote %lld > cap %llu\n", (long long)got2, (unsigned long long)cap2); abort(); }
    free(out2);
  }
  char buf[32];
  union { uint64_t u; double d; } x;
  x.u = rnd();
  const int len = pfa_format_double(x.d, buf);
  if (len <= 0 || len > 32) { fprintf(stderr, "repr length %d\n", len); abort(); }
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  const char* dir = argc > 2 ? argv[2] : "/tmp";
  for (long i = 0; i < iters; ++i) {
    fuzz_text(random_fasta());
    for (int k = 0; k < 8; ++k) fuzz_utf8();
    if (i % 16 == 0) fuzz_files(dir);
    fuzz_rows();
  }
  printf("host fuzz: %ld iterations clean\n", iters);
  return 0;
}
