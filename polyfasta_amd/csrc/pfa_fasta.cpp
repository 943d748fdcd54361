// pfa_fasta.cpp — host-side FASTA packer of libpfa_hip.so: text -> row-major
// upper-case byte matrix (one byte per base), with the exact record semantics
// of the reference reader (PolyFastA.py:227-250):
//   * lines end at "\n", "\r\n" or a lone "\r" (Python text mode, universal newlines)
//   * a line whose first character is '>' starts a record; the name is the rest
//     of the line with trailing whitespace removed                 (:241-243)
//   * other lines are appended, rstrip()ed and upper-cased, to the current
//     record; they are ignored while no named record is open       (:244-245)
//   * a repeated name resets its sequence but keeps its first position (dict)
//   * "not FASTA" when the last header seen is empty / absent      (:246-248)
// Trailing-whitespace set = the ASCII characters for which Python's
// str.isspace() is true.  Header bytes >= 0x80 (UTF-8 sample names) are kept as
// they are; in SEQUENCE lines they are refused (PFA_ERR_UNSUPPORTED, per file
// PFA_FILE_UNSUPPORTED): the reference would decode them as UTF-8 and apply
// Unicode case rules, outside this path's one-byte-per-base contract.
// No exception and no abort crosses the ABI: every entry point catches.
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <stdio.h>
#include <sys/stat.h>

#include <new>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pfa.h"
#include "pfa_host.h"
static std::string pfa_last_error_local() { return std::string(pfa_last_error()); }

namespace {

inline bool py_space(uint8_t c) {
  return c == ' ' || (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x1F);
}

struct Line {
  uint64_t begin, end, next;  // [begin, end) content without terminator; next = start of the following line
};

inline Line next_line(const uint8_t* t, uint64_t n, uint64_t pos) {
  // memchr for '\n', then look for an earlier '\r' inside that span (CRLF or a lone CR)
  const uint8_t* nl = static_cast<const uint8_t*>(memchr(t + pos, '\n', n - pos));
  uint64_t e = nl ? uint64_t(nl - t) : n;
  const uint8_t* cr = static_cast<const uint8_t*>(memchr(t + pos, '\r', e - pos));
  if (cr) e = uint64_t(cr - t);
  uint64_t nx = e;
  if (nx < n) nx += (t[nx] == '\r' && nx + 1 < n && t[nx + 1] == '\n') ? 2 : 1;
  return Line{pos, e, nx};
}

// upper-case copy; returns the OR of all bytes (bit 7 set <=> a non-ASCII byte was seen)
inline uint8_t copy_upper(uint8_t* dst, const uint8_t* src, uint64_t n) {
  uint8_t acc = 0;
  for (uint64_t k = 0; k < n; ++k) {
    const uint8_t c = src[k];
    acc |= c;
    dst[k] = uint8_t(c - ((uint8_t(c - 'a') < 26u) << 5));
  }
  return acc;
}

inline uint64_t rstrip_end(const uint8_t* t, uint64_t b, uint64_t e) {
  while (e > b && py_space(t[e - 1])) --e;
  return e;
}

}  // namespace

// ---- single-pass index for the streaming ingest (pfa_ingest.hip) ---------------
// One pass over the text of a file that sits in the pinned staging ring: the records, their
// names, and for every record ONE contiguous run of raw sequence bytes.  Records written on
// several lines are compacted in place (the stripped lines are moved together: the destination
// never passes the source); single-line records — what the reference's examples and most
// pipelines write — are not touched at all.  Upper-casing and the byte >= 0x80 test are left to
// the card (repack_kernel), which reads every byte anyway.  Same record rules as
// fasta_index_impl above (PolyFastA.py:227-250).
namespace pfa {

bool valid_utf8(const uint8_t* p, uint64_t n) {
  uint64_t i = 0;
  while (i < n) {
    const uint8_t c = p[i];
    if (c < 0x80u) { ++i; continue; }
    uint32_t need;
    uint8_t lo = 0x80u, hi = 0xBFu;      // range of the first continuation byte
    if (c >= 0xC2u && c <= 0xDFu) need = 1;
    else if (c == 0xE0u) { need = 2; lo = 0xA0u; }
    else if (c >= 0xE1u && c <= 0xECu) need = 2;
    else if (c == 0xEDu) { need = 2; hi = 0x9Fu; }
    else if (c == 0xEEu || c == 0xEFu) need = 2;
    else if (c == 0xF0u) { need = 3; lo = 0x90u; }
    else if (c >= 0xF1u && c <= 0xF3u) need = 3;
    else if (c == 0xF4u) { need = 3; hi = 0x8Fu; }
    else return false;
    if (i + need >= n) return false;      // truncated
    if (p[i + 1] < lo || p[i + 1] > hi) return false;
    for (uint32_t k = 2; k <= need; ++k)
      if ((p[i + k] & 0xC0u) != 0x80u) return false;
    i += need + 1u;
  }
  return true;
}

// a line outside the kept sequence lines: any byte >= 0x80 sends it through the validator (rare)
static inline bool line_text_ok(const uint8_t* p, uint64_t n) {
  uint8_t acc = 0;
  for (uint64_t i = 0; i < n; ++i) acc |= p[i];
  return acc < 0x80u || valid_utf8(p, n);
}

int64_t index_compact(uint8_t* text, uint64_t n_text, std::vector<pfa_fasta_row>& rows, IndexScratch& sc) {
  rows.clear();
  sc.by_name.clear();
  sc.spans.clear();
  sc.bad_text = false;
  if (!text && n_text) { set_error("index_compact: text is NULL"); return PFA_ERR_ARG; }
  // a file without any '\r' (the usual case) needs no second look at its lines
  const bool has_cr = n_text && memchr(text, '\r', n_text) != nullptr;
  bool have_head = false, head_named = false, multi = false;
  size_t cur = 0;
  uint64_t pos = 0;
  while (pos < n_text) {
    uint64_t e, nx;
    if (!has_cr) {
      const uint8_t* nl = static_cast<const uint8_t*>(memchr(text + pos, '\n', n_text - pos));
      e = nl ? uint64_t(nl - text) : n_text;
      nx = nl ? e + 1 : n_text;
    } else {
      const Line ln = next_line(text, n_text, pos);
      e = ln.end;
      nx = ln.next;
    }
    if (text[pos] == '>') {
      if (have_head) sc.spans[cur].end = pos;
      if (!line_text_ok(text + pos, e - pos)) sc.bad_text = true;
      const uint64_t nb = pos + 1, ne = rstrip_end(text, nb > e ? e : nb, e);
      std::string_view name(reinterpret_cast<const char*>(text + nb), ne > nb ? ne - nb : 0);
      auto it = sc.by_name.find(name);
      if (it == sc.by_name.end()) {
        rows.push_back(pfa_fasta_row{nb, (uint32_t)name.size(), 0u, nx});
        sc.spans.push_back(IndexScratch::Span{nx, nx, 0u});
        cur = rows.size() - 1;
        sc.by_name.emplace(name, (uint32_t)cur);
      } else {   // a repeated name starts over but keeps its first position (dict)
        cur = it->second;
        // (the lines of the earlier occurrence are dropped here, but the reference has decoded them like every other line)
        if (sc.spans[cur].end > sc.spans[cur].first_line &&
            !line_text_ok(text + sc.spans[cur].first_line, sc.spans[cur].end - sc.spans[cur].first_line)) sc.bad_text = true;
        rows[cur].seq_len = 0;
        rows[cur].seq_off = nx;
        sc.spans[cur] = IndexScratch::Span{nx, nx, 0u};
      }
      have_head = true;
      head_named = !name.empty();
    } else if (head_named) {
      const uint64_t e2 = rstrip_end(text, pos, e);
      const uint64_t add = e2 - pos;
      if (add) {
        if ((uint64_t)rows[cur].seq_len + add > 0xFFFFFFFFull) { set_error("sequence longer than 2^32-1"); return PFA_ERR_FORMAT; }
        if (sc.spans[cur].n_lines == 0) rows[cur].seq_off = pos;
        else multi = true;
        rows[cur].seq_len += (uint32_t)add;
        sc.spans[cur].n_lines += 1;
      }
    } else if (e > pos && !line_text_ok(text + pos, e - pos)) {
      sc.bad_text = true;      // a line nobody keeps (before the first header, behind an empty one): the reference decodes it all the same
    }
    pos = nx;
  }
  if (have_head) sc.spans[cur].end = n_text;
  if (!have_head || !head_named) {
    set_error("not FASTA: no header line, or the last header is empty");
    return PFA_ERR_FORMAT;
  }
  if (multi) {
    for (size_t r = 0; r < rows.size(); ++r) {
      if (sc.spans[r].n_lines < 2) continue;
      uint64_t w = rows[r].seq_off, p = rows[r].seq_off;
      const uint64_t end = sc.spans[r].end;
      while (p < end) {
        const Line ln = next_line(text, end, p);
        const uint64_t e2 = rstrip_end(text, ln.begin, ln.end);
        if (e2 > ln.begin) {
          if (w != ln.begin) memmove(text + w, text + ln.begin, e2 - ln.begin);
          w += e2 - ln.begin;
        }
        p = ln.next;
      }
    }
  }
  return (int64_t)rows.size();
}

}  // namespace pfa

// ---- 2-bit transport rows (pfa_ingest.hip) --------------------------------------
// A row that holds nothing but A/C/G/T (either case) crosses PCIe as two bits per base — the card
// expands it to the upper-case bytes readfasta would have produced (repack_kernel) — so the link
// carries a quarter of the text.  Code = (byte >> 1) & 3 after case folding: A 0, C 1, T 2, G 3 (the
// kernels' codon base order); base i of a group of four sits in bits 2i+1..2i of its byte.  Any other
// byte makes the row travel as it is (every distinct byte is an allele, PolyFastA.py:258).
namespace pfa {

static bool pack2_scalar(const uint8_t* src, uint64_t n, uint8_t* dst) {
  static const uint8_t canon[4] = {'A', 'C', 'T', 'G'};
  uint32_t bad = 0;
  uint64_t i = 0;
  for (; i + 4 <= n; i += 4) {
    uint32_t b = 0;
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t u = src[i + k] & 0xDFu, c = (u >> 1) & 3u;
      bad |= canon[c] ^ u;
      b |= c << (2 * k);
    }
    dst[i >> 2] = (uint8_t)b;
  }
  if (i < n) {
    uint32_t b = 0;
    for (uint32_t k = 0; i + k < n; ++k) {
      const uint32_t u = src[i + k] & 0xDFu, c = (u >> 1) & 3u;
      bad |= canon[c] ^ u;
      b |= c << (2 * k);
    }
    dst[i >> 2] = (uint8_t)b;
  }
  return bad == 0;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool pack2_avx2(const uint8_t* src, uint64_t n, uint8_t* dst) {
  const __m256i fold = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3);
  const __m256i canon = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i w8 = _mm256_set1_epi16(0x0401), w16 = _mm256_set1_epi32(0x00100001);
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  __m256i bad = _mm256_setzero_si256();
  uint64_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i u = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i)), fold);
    const __m256i c = _mm256_and_si256(_mm256_srli_epi16(u, 1), three);
    bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(canon, c), u));
    const __m256i t = _mm256_madd_epi16(_mm256_maddubs_epi16(c, w8), w16);        // one packed byte per 32-bit lane
    const __m256i q = _mm256_shuffle_epi8(t, pick);                               // 4 bytes at the bottom of each 128-bit half
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(q), hi = (uint32_t)_mm256_extract_epi32(q, 4);
    memcpy(dst + (i >> 2), &lo, 4);
    memcpy(dst + (i >> 2) + 4, &hi, 4);
  }
  const bool ok_tail = i == n || pack2_scalar(src + i, n - i, dst + (i >> 2));
  return ok_tail && _mm256_testz_si256(bad, bad);
}
#endif

#if defined(__x86_64__)
// 64 bases per step: the same arithmetic on 512-bit registers, and the sixteen packed bytes leave through one
// dword -> byte narrowing store instead of two extracts (pack2_row is three quarters of a loader thread's time after read())
__attribute__((target("avx512f,avx512bw"))) static bool pack2_avx512(const uint8_t* src, uint64_t n, uint8_t* dst) {
  const __m512i fold = _mm512_set1_epi8((char)0xDF), three = _mm512_set1_epi8(3);
  const __m512i canon = _mm512_broadcast_i32x4(_mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i w8 = _mm512_set1_epi16(0x0401), w16 = _mm512_set1_epi32(0x00100001);
  __m512i bad = _mm512_setzero_si512();
  uint64_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i u = _mm512_and_si512(_mm512_loadu_si512(src + i), fold);
    const __m512i c = _mm512_and_si512(_mm512_srli_epi16(u, 1), three);
    bad = _mm512_or_si512(bad, _mm512_xor_si512(_mm512_shuffle_epi8(canon, c), u));
    const __m512i t = _mm512_madd_epi16(_mm512_maddubs_epi16(c, w8), w16);        // one packed byte per 32-bit lane
    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + (i >> 2)), _mm512_cvtepi32_epi8(t));
  }
  const bool ok_tail = i == n || pack2_avx2(src + i, n - i, dst + (i >> 2));
  return ok_tail && _mm512_test_epi8_mask(bad, bad) == 0;
}
#endif

bool pack2_row(const uint8_t* src, uint64_t n, uint8_t* dst) {
#if defined(__x86_64__)
  static const int level = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f") && !getenv("PFA_NO_AVX512") ? 2
                           : __builtin_cpu_supports("avx2") ? 1 : 0;
  if (level == 2) return pack2_avx512(src, n, dst);
  if (level == 1) return pack2_avx2(src, n, dst);
#endif
  return pack2_scalar(src, n, dst);
}

}  // namespace pfa

namespace pfa {

static double read_quota_v2(const std::string& file) {
  FILE* f = fopen(file.c_str(), "r");
  if (!f) return 1e9;
  char a[64] = "", b[64] = "";
  const int n = fscanf(f, "%63s %63s", a, b);
  fclose(f);
  if (n < 2 || strcmp(a, "max") == 0) return 1e9;
  const double q = atof(a), p = atof(b);
  return q > 0 && p > 0 ? q / p : 1e9;
}

static double read_number(const std::string& file) {
  FILE* f = fopen(file.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

double cgroup_cpu_quota(const char* cgroup_root, const char* proc_cgroup) {
  double best = 1e9;
  std::string v2_path = "/", v1_path = "/";
  if (FILE* f = fopen(proc_cgroup, "r")) {
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
      // "0::/path" (v2) or "4:cpu,cpuacct:/path" (v1)
      char* c1 = strchr(line, ':');
      char* c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
      if (!c2) continue;
      std::string ctl(c1 + 1, c2), path(c2 + 1);
      while (!path.empty() && (path.back() == '\n' || path.back() == '\r')) path.pop_back();
      if (ctl.empty()) v2_path = path;
      else if ((',' + ctl + ',').find(",cpu,") != std::string::npos) v1_path = path;
    }
    fclose(f);
  }
  const std::string root(cgroup_root);
  // v2: the group and every ancestor (a limit anywhere on the way up holds)
  for (std::string p = v2_path;;) {
    best = std::min(best, read_quota_v2(root + (p == "/" ? "" : p) + "/cpu.max"));
    if (p == "/" || p.empty()) break;
    const size_t sl = p.rfind('/');
    p = sl == 0 || sl == std::string::npos ? "/" : p.substr(0, sl);
  }
  // v1: cpu controller mounted at <root>/cpu (or cpu,cpuacct)
  for (const char* mnt : {"/cpu", "/cpu,cpuacct"}) {
    for (std::string p = v1_path;;) {
      const std::string dir = root + mnt + (p == "/" ? "" : p);
      const double q = read_number(dir + "/cpu.cfs_quota_us"), per = read_number(dir + "/cpu.cfs_period_us");
      if (q > 0 && per > 0) best = std::min(best, q / per);
      if (p == "/" || p.empty()) break;
      const size_t sl = p.rfind('/');
      p = sl == 0 || sl == std::string::npos ? "/" : p.substr(0, sl);
    }
  }
  return best;
}

// the CPUs of the process's affinity mask grouped by `key(cpu)` (a number read from sysfs; < 0: unreadable), in mask order
static std::vector<std::vector<int>> cpu_groups(const char* sys_root, long (*key)(const std::string& cpu_dir)) {
  std::vector<std::vector<int>> out;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) != 0) return out;
  std::vector<long> keys;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &set)) continue;
    const long k = key(std::string(sys_root) + "/cpu" + std::to_string(c) + "/");
    if (k < 0) return {};
    const size_t i = (size_t)(std::find(keys.begin(), keys.end(), k) - keys.begin());
    if (i == keys.size()) { keys.push_back(k); out.emplace_back(); }
    out[i].push_back(c);
  }
  return out;
}

static bool pin_to_group(const std::vector<std::vector<int>>& groups, uint32_t k) {
  if (groups.empty()) return false;
  cpu_set_t one;
  CPU_ZERO(&one);
  for (int c : groups[k % groups.size()]) CPU_SET(c, &one);
  return pthread_setaffinity_np(pthread_self(), sizeof(one), &one) == 0;
}

bool pin_to_core(uint32_t k) {
  static const std::vector<std::vector<int>> cores = cpu_groups("/sys/devices/system/cpu", [](const std::string& d) -> long {
    const double pkg = read_number(d + "topology/physical_package_id"), core = read_number(d + "topology/core_id");
    return pkg < 0 || core < 0 ? -1 : ((long)pkg << 20) | (long)core;
  });
  return pin_to_group(cores, k);
}

std::vector<std::vector<int>> l3_domains(const char* sys_root) {
  return cpu_groups(sys_root, [](const std::string& d) -> long { return (long)read_number(d + "cache/index3/id"); });
}

bool pin_to_l3(uint32_t k) {
  // (the mask as it is when the first loader of the process starts: a rank has bound itself to its CPUs by then).  A mask
  // that spans sockets (nothing bound the process: no GPU topology, PFA_NO_AFFINITY) is left to the scheduler: dealing
  // loaders out over both sockets puts half of them across the socket link from the pinned ring.
  static const std::vector<std::vector<int>> domains = [] {
    std::vector<std::vector<int>> d = l3_domains("/sys/devices/system/cpu");
    long pkg = -1;
    for (const std::vector<int>& g : d) {
      const long p = (long)read_number("/sys/devices/system/cpu/cpu" + std::to_string(g.front()) + "/topology/physical_package_id");
      if (pkg >= 0 && p != pkg) return std::vector<std::vector<int>>{};
      pkg = p;
    }
    return d;
  }();
  return domains.size() > 1 && pin_to_group(domains, k);
}

double cpu_share() {
  if (const char* e = getenv("PFA_CPUS")) { const double v = atof(e); if (v > 0) return v; }
  cpu_set_t set;
  double cpus = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = (double)std::max(1, CPU_COUNT(&set));
  double ranks = 1;
  for (const char* var : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"})
    if (const char* e = getenv(var)) { const long v = strtol(e, nullptr, 10); if (v > 0) { ranks = (double)v; break; } }
  const double by_affinity = getenv("PFA_CPUS_ARE_MINE") ? cpus : cpus / ranks;
  return std::max(1.0, std::min(by_affinity, cgroup_cpu_quota() / ranks));
}

}  // namespace pfa

extern "C" {

int pfa_host_l3_domains(const char* cpu_sys_root, int* domain_of, int cap) {
  const std::vector<std::vector<int>> dom = pfa::l3_domains(cpu_sys_root ? cpu_sys_root : "/sys/devices/system/cpu");
  if (domain_of) {
    // position of every CPU in the affinity mask (ascending CPU number)
    std::vector<std::pair<int, int>> cpus;
    for (size_t d = 0; d < dom.size(); ++d)
      for (int c : dom[d]) cpus.push_back({c, (int)d});
    std::sort(cpus.begin(), cpus.end());
    for (size_t i = 0; i < cpus.size() && (int)i < cap; ++i) domain_of[i] = cpus[i].second;
  }
  return (int)dom.size();
}

double pfa_host_cpu_share(const char* cgroup_root, const char* proc_cgroup) {
  if (!cgroup_root && !proc_cgroup) return pfa::cpu_share();
  return pfa::cgroup_cpu_quota(cgroup_root ? cgroup_root : "/sys/fs/cgroup", proc_cgroup ? proc_cgroup : "/proc/self/cgroup");
}

static int64_t fasta_index_impl(const uint8_t* text, uint64_t n_text, pfa_fasta_rec* recs, uint64_t cap) {
  if (!text && n_text) { pfa::set_error("pfa_fasta_index: text is NULL"); return PFA_ERR_ARG; }
  std::vector<pfa_fasta_rec> out;
  std::unordered_map<std::string_view, size_t> by_name;
  bool have_head = false;     // a header line has been seen
  bool head_named = false;    // ... and the current one has a non-empty name
  size_t cur = 0;
  uint64_t pos = 0;
  while (pos < n_text) {
    const Line ln = next_line(text, n_text, pos);
    if (text[pos] == '>') {
      if (have_head) out[cur].end_off = pos;
      const uint64_t nb = pos + 1, ne = rstrip_end(text, pos + 1 > ln.end ? ln.end : pos + 1, ln.end);
      std::string_view name(reinterpret_cast<const char*>(text + nb), ne > nb ? ne - nb : 0);
      auto it = by_name.find(name);
      if (it == by_name.end()) {
        pfa_fasta_rec r;
        r.name_off = nb;
        r.name_len = (uint32_t)name.size();
        r.seq_len = 0;
        r.first_line_off = ln.next;
        r.end_off = ln.next;
        out.push_back(r);
        cur = out.size() - 1;
        by_name.emplace(name, cur);
      } else {
        cur = it->second;
        out[cur].seq_len = 0;
        out[cur].first_line_off = ln.next;
        out[cur].end_off = ln.next;
      }
      have_head = true;
      head_named = !name.empty();
    } else if (head_named) {
      const uint64_t e = rstrip_end(text, ln.begin, ln.end);   // (bytes >= 0x80 are rejected by the pack pass)
      const uint64_t add = e - ln.begin;
      if ((uint64_t)out[cur].seq_len + add > 0xFFFFFFFFull) { pfa::set_error("sequence longer than 2^32-1"); return PFA_ERR_FORMAT; }
      out[cur].seq_len += (uint32_t)add;
    }
    pos = ln.next;
  }
  if (have_head) out[cur].end_off = n_text;
  if (!have_head || !head_named) {
    pfa::set_error("not FASTA: no header line, or the last header is empty");
    return PFA_ERR_FORMAT;
  }
  // an empty-named record never receives sequence lines: its span is empty
  for (auto& r : out)
    if (r.name_len == 0) { r.seq_len = 0; r.end_off = r.first_line_off; }
  if (recs) {
    if (out.size() > cap) { pfa::set_error("pfa_fasta_index: %zu records, capacity %llu", out.size(), (unsigned long long)cap); return PFA_ERR_ARG; }
    if (!out.empty()) memcpy(recs, out.data(), out.size() * sizeof(pfa_fasta_rec));
  }
  return (int64_t)out.size();
}

int64_t pfa_fasta_index(const uint8_t* text, uint64_t n_text, pfa_fasta_rec* recs, uint64_t cap) {
  try {
    return fasta_index_impl(text, n_text, recs, cap);
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_fasta_index: out of host memory");
    return PFA_ERR_NOMEM;
  } catch (...) {
    pfa::set_error("pfa_fasta_index: unexpected exception");
    return PFA_ERR_ARG;
  }
}

int64_t pfa_fasta_index_compact(uint8_t* text, uint64_t n_text, pfa_fasta_row* rows, uint64_t cap) {
  try {
    std::vector<pfa_fasta_row> v;
    pfa::IndexScratch sc;
    const int64_t n = pfa::index_compact(text, n_text, v, sc);
    if (n < 0) return n;
    if ((uint64_t)n > cap) { pfa::set_error("pfa_fasta_index_compact: %lld records, capacity %llu", (long long)n, (unsigned long long)cap); return PFA_ERR_ARG; }
    if (n && !rows) { pfa::set_error("pfa_fasta_index_compact: rows is NULL"); return PFA_ERR_ARG; }
    if (n) memcpy(rows, v.data(), (size_t)n * sizeof(pfa_fasta_row));
    return n;
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_fasta_index_compact: out of host memory");
    return PFA_ERR_NOMEM;
  } catch (...) {
    pfa::set_error("pfa_fasta_index_compact: unexpected exception");
    return PFA_ERR_ARG;
  }
}

int pfa_fasta_pack2(const uint8_t* src, uint64_t n, uint8_t* dst) {
  if ((!src || !dst) && n) { pfa::set_error("pfa_fasta_pack2: NULL argument"); return PFA_ERR_ARG; }
  return pfa::pack2_row(src, n, dst) ? 1 : 0;
}

int pfa_fasta_pack(const uint8_t* text, uint64_t n_text, const pfa_fasta_rec* recs, uint64_t n_recs,
                   uint8_t* bases_out, uint32_t row_stride) {
  if (!text || !recs || !bases_out) { pfa::set_error("pfa_fasta_pack: NULL argument"); return PFA_ERR_ARG; }
  for (uint64_t i = 0; i < n_recs; ++i) {
    const pfa_fasta_rec& r = recs[i];
    if (r.seq_len > row_stride || r.end_off > n_text || r.first_line_off > r.end_off) {
      pfa::set_error("pfa_fasta_pack: record %llu does not fit (len %u, stride %u)", (unsigned long long)i, r.seq_len, row_stride);
      return PFA_ERR_ARG;
    }
    uint8_t* row = bases_out + i * (uint64_t)row_stride;
    uint64_t w = 0, pos = r.first_line_off;
    while (pos < r.end_off) {
      const Line ln = next_line(text, r.end_off, pos);
      const uint64_t e = rstrip_end(text, ln.begin, ln.end);
      if (w + (e - ln.begin) > r.seq_len) { pfa::set_error("pfa_fasta_pack: record %llu longer than indexed", (unsigned long long)i); return PFA_ERR_ARG; }
      if (copy_upper(row + w, text + ln.begin, e - ln.begin) & 0x80u) {
        pfa::set_error("non-ASCII byte in a sequence line of record %llu (one byte per base only)", (unsigned long long)i);
        return PFA_ERR_UNSUPPORTED;
      }
      w += e - ln.begin;
      pos = ln.next;
    }
    if (w != r.seq_len) { pfa::set_error("pfa_fasta_pack: record %llu shorter than indexed", (unsigned long long)i); return PFA_ERR_ARG; }
    if (row_stride > w) memset(row + w, 0, row_stride - w);
  }
  return PFA_OK;
}


// ---- multi-threaded batch loader ---------------------------------------------
// Reads, indexes and packs many FASTA files with a pool of host threads: what
// limits alignments/s end to end is this ingest, not the scan.
// One pass per file (read -> index -> reserve -> pack) with thread-local,
// reused text buffers; matrices are bump-allocated in `bases_out`, so their
// order in the buffer is arbitrary (descriptors carry the offsets).
int pfa_fasta_load_files(const char* const* paths, uint32_t n_files, uint32_t stride_align, uint32_t n_threads,
                         uint8_t* bases_out, uint64_t cap, pfa_file_info* info, char* names_out,
                         uint64_t names_cap, uint64_t* bytes_used, uint64_t* names_used) {
  if (!paths || !bases_out || !info || !bytes_used) { pfa::set_error("pfa_fasta_load_files: NULL argument"); return PFA_ERR_ARG; }
  if (stride_align < 16 || (stride_align & (stride_align - 1))) { pfa::set_error("stride_align must be a power of two >= 16"); return PFA_ERR_ARG; }
  const uint32_t nt = std::max(1u, std::min<uint32_t>(n_threads ? n_threads : 16u, std::max(1u, n_files)));
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> used{0}, nused{0};
  std::atomic<int> hard_rc{0};
  std::string hard_msg;
  std::mutex msg_mu;

  auto reserve = [](std::atomic<uint64_t>& ctr, uint64_t need, uint64_t limit, uint64_t* off) {
    uint64_t cur = ctr.load();
    do {
      if (cur + need > limit) return false;
    } while (!ctr.compare_exchange_weak(cur, cur + need));
    *off = cur;
    return true;
  };

  // files above this size are refused (PFA_ERR_IO) instead of attempted
  constexpr uint64_t kMaxFileBytes = 1ull << 40;
  auto worker_body = [&]() {
    std::vector<uint8_t> text;
    std::vector<pfa_fasta_rec> recs;
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= n_files) break;
      pfa_file_info& fi = info[i];
      memset(&fi, 0, sizeof(fi));
      FILE* fh = fopen(paths[i], "rb");
      if (!fh) { fi.status = PFA_ERR_IO; continue; }
      // only regular files have a size to trust (a directory opens fine and ftell()s LONG_MAX)
      struct stat sb;
      if (fstat(fileno(fh), &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 0 || (uint64_t)sb.st_size > kMaxFileBytes) {
        fclose(fh);
        fi.status = PFA_ERR_IO;
        continue;
      }
      const long sz = (long)sb.st_size;
      if (text.size() < (size_t)sz) text.resize((size_t)sz);
      const size_t got = sz > 0 ? fread(text.data(), 1, (size_t)sz, fh) : 0;
      fclose(fh);
      if (got != (size_t)sz) { fi.status = PFA_ERR_IO; continue; }
      int64_t n = pfa_fasta_index(text.data(), (uint64_t)sz, nullptr, 0);
      if (n == PFA_ERR_FORMAT) { fi.status = PFA_FILE_NOT_FASTA; continue; }
      if (n >= 0) {
        recs.resize((size_t)n);
        n = pfa_fasta_index(text.data(), (uint64_t)sz, recs.data(), (uint64_t)n);
      }
      if (n < 0) {
        std::lock_guard<std::mutex> g(msg_mu);
        hard_rc = (int)n;
        hard_msg = std::string(paths[i]) + ": " + pfa_last_error_local();
        continue;
      }
      fi.n_rows = (uint32_t)n;
      fi.n_sites = n ? recs[0].seq_len : 0;
      for (auto& r : recs)
        if (r.seq_len != fi.n_sites) { fi.status = PFA_FILE_UNEQUAL; break; }
      uint64_t st = 0, need = 0, nneed = 0;
      if (fi.status == 0) {
        st = std::max<uint64_t>(16, ((uint64_t)fi.n_sites + stride_align - 1) & ~uint64_t(stride_align - 1));
        if (st > 0xFFFFFFF0ull) { fi.status = PFA_FILE_UNEQUAL; continue; }
        fi.row_stride = (uint32_t)st;
        need = (((uint64_t)std::max(fi.n_rows, 1u) * st) + 255u) & ~255ull;
      }
      if (names_out)
        for (auto& r : recs) nneed += r.name_len + 1u;
      uint64_t off = 0, noff = 0;
      if (need && !reserve(used, need, cap, &off)) { fi.status = PFA_FILE_DEFERRED; fi.row_stride = (uint32_t)std::min<uint64_t>(need >> 8, 0xFFFFFFFFull); continue; }
      if (nneed && !reserve(nused, nneed, names_cap, &noff)) { fi.status = PFA_FILE_DEFERRED; continue; }
      fi.base_off = off;
      fi.names_off = noff;
      if (names_out) {
        char* q = names_out + noff;
        for (auto& r : recs) { memcpy(q, text.data() + r.name_off, r.name_len); q[r.name_len] = 0; q += r.name_len + 1u; }
      }
      if (fi.status == 0 && fi.n_rows) {
        const int rc = pfa_fasta_pack(text.data(), (uint64_t)sz, recs.data(), recs.size(), bases_out + off, fi.row_stride);
        if (rc == PFA_ERR_UNSUPPORTED) {
          fi.status = PFA_FILE_UNSUPPORTED;     // this file only; its buffer space stays unused
        } else if (rc != PFA_OK) {
          std::lock_guard<std::mutex> g(msg_mu);
          hard_rc = rc;
          hard_msg = std::string(paths[i]) + ": " + pfa_last_error_local();
        }
      }
    }
  };
  // nothing may leave a thread: an exception there would end the process (std::terminate)
  auto worker = [&]() {
    try {
      worker_body();
    } catch (const std::bad_alloc&) {
      std::lock_guard<std::mutex> g(msg_mu);
      hard_rc = PFA_ERR_NOMEM;
      hard_msg = "pfa_fasta_load_files: out of host memory";
      next.store(n_files);
    } catch (...) {
      std::lock_guard<std::mutex> g(msg_mu);
      hard_rc = PFA_ERR_ARG;
      hard_msg = "pfa_fasta_load_files: unexpected exception";
      next.store(n_files);
    }
  };
  try {
    std::vector<std::thread> th;
    struct Joiner {
      std::vector<std::thread>& t;
      ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); }
    } joiner{th};
    for (uint32_t t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
  } catch (...) {      // thread creation failed: the threads that did start have been joined
    pfa::set_error("pfa_fasta_load_files: cannot start the loader threads");
    return PFA_ERR_NOMEM;
  }
  if (hard_rc != 0) { pfa::set_error("%s", hard_msg.c_str()); return hard_rc; }
  *bytes_used = used.load();
  if (names_used) *names_used = nused.load();
  return PFA_OK;
}

}  // extern "C"
