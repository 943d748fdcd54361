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
// str.isspace() is true.  Bytes >= 0x80 are rejected (PFA_ERR_FORMAT): the
// reference would decode them as UTF-8 and apply Unicode case rules, which is
// outside this path's one-byte-per-base contract (out of scope).
#include <stdint.h>
#include <string.h>

#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/pfa.h"

namespace pfa {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}

namespace {

inline bool py_space(uint8_t c) {
  return c == ' ' || (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x1F);
}

struct Line {
  uint64_t begin, end, next;  // [begin, end) content without terminator; next = start of the following line
};

inline Line next_line(const uint8_t* t, uint64_t n, uint64_t pos) {
  uint64_t e = pos;
  while (e < n && t[e] != '\n' && t[e] != '\r') ++e;
  uint64_t nx = e;
  if (nx < n) nx += (t[nx] == '\r' && nx + 1 < n && t[nx + 1] == '\n') ? 2 : 1;
  return Line{pos, e, nx};
}

inline uint64_t rstrip_end(const uint8_t* t, uint64_t b, uint64_t e) {
  while (e > b && py_space(t[e - 1])) --e;
  return e;
}

}  // namespace

extern "C" {

int64_t pfa_fasta_index(const uint8_t* text, uint64_t n_text, pfa_fasta_rec* recs, uint64_t cap) {
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
      for (uint64_t i = nb; i < ne; ++i)
        if (text[i] >= 0x80) { pfa::set_error("non-ASCII byte 0x%02x in a header at offset %llu", text[i], (unsigned long long)i); return PFA_ERR_FORMAT; }
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
      const uint64_t e = rstrip_end(text, ln.begin, ln.end);
      for (uint64_t i = ln.begin; i < e; ++i)
        if (text[i] >= 0x80) { pfa::set_error("non-ASCII byte 0x%02x in a sequence line at offset %llu", text[i], (unsigned long long)i); return PFA_ERR_FORMAT; }
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
      for (uint64_t k = ln.begin; k < e; ++k) {
        const uint8_t c = text[k];
        row[w++] = (c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c;
      }
      pos = ln.next;
    }
    if (w != r.seq_len) { pfa::set_error("pfa_fasta_pack: record %llu shorter than indexed", (unsigned long long)i); return PFA_ERR_ARG; }
    if (row_stride > w) memset(row + w, 0, row_stride - w);
  }
  return PFA_OK;
}

}  // extern "C"
