// ingest_tsan.cpp — CPU sanitizer harness for the streaming ingest (csrc/pfa_ingest.hip).  TEST INFRASTRUCTURE.
// Built by tests/test_host_sanitizers.py with g++ -fsanitize=thread (and again with address,undefined) from
// csrc/pfa_ingest.hip (-DPFA_INGEST_HOST_TEST: no device code) + csrc/pfa_fasta.cpp + this file, against
// tests/hip_mock.h: streams are threads that run their copies LATER than the call that enqueued them.  The harness
// supplies doubles of what the ingest calls in the rest of the library (context, plans, the repack launch — a host
// restatement of repack_kernel's contract), feeds directories of awkward files through the real pipeline with many
// loader threads, tiny slots (ring wrap-around, oversized files in buffers of their own), several feeds, population
// keys, and checks every matrix byte the "card" ends up with against an independent reading of the files.
#include <stdarg.h>
#include <stdio.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../polyfasta_amd/csrc/pfa_common.h"
#include "../polyfasta_amd/csrc/pfa_host.h"

// ---------------------------------------------------------------- doubles of the library around the ingest
static thread_local char g_err[768];
namespace pfa {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
Context& ctx() {
  static Context c;
  return c;
}
// host restatement of repack_kernel: packed runs -> 'A' 'C' 'T' 'G' by code, text runs upper-cased, the rest of the row
// zeroed, bytes >= 0x80 flagged per file; enqueued on the stream like a kernel launch
int launch_repack(const uint8_t* text, const Seg* segs, uint32_t n_segs, uint8_t* packed, uint32_t stride_align, uint32_t* bad, hipStream_t st) {
  st->push([=]() {
    for (uint32_t s = 0; s < n_segs; ++s) {
      const Seg g = segs[s];
      const uint8_t* src = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(text) + g.src);   // (text may be null: absolute addresses)
      uint8_t* dst = packed + g.dst;
      uint32_t stride = (g.len + stride_align - 1u) & ~(stride_align - 1u);
      if (stride < 16u) stride = 16u;
      for (uint32_t i = 0; i < stride; ++i) {
        uint8_t c = 0;
        if (i < g.len) {
          if (g.file & kSegPacked) {
            c = (uint8_t)"ACTG"[(src[i >> 2] >> (2u * (i & 3u))) & 3u];
          } else {
            c = src[i];
            if (c & 0x80u) bad[g.file] = 1u;
            if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
          }
        }
        dst[i] = c;
      }
    }
  });
  return PFA_OK;
}
}  // namespace pfa
extern "C" const char* pfa_last_error(void) { return g_err; }

struct pfa_plan {
  std::vector<pfa_desc> descs;
  std::vector<uint32_t> ridx;
  const uint8_t* bases = nullptr;
  uint64_t n_bytes = 0;
  std::vector<uint8_t> snap;   // the matrices as the "scan" saw them: copied in stream order, as a kernel would read them
};
extern "C" int pfa_plan_create(const pfa_desc* descs, uint32_t n, const uint32_t* row_idx, size_t n_row_idx, uint64_t n_bytes, uint32_t, pfa_plan** out) {
  pfa_plan* p = new pfa_plan();
  p->descs.assign(descs, descs + n);
  if (n_row_idx) p->ridx.assign(row_idx, row_idx + n_row_idx);
  p->n_bytes = n_bytes;
  *out = p;
  return PFA_OK;
}
extern "C" int pfa_plan_run(pfa_plan* p, const uint8_t* d_bases, int) {
  p->bases = d_bases;
  // (another ingest's next batch may overwrite the shared matrix buffer behind this point of the stream, as on the card)
  pfa::ctx().stream->push([p, d_bases]() { p->snap.assign(d_bases, d_bases + p->n_bytes); });
  return PFA_OK;
}
extern "C" void pfa_plan_destroy(pfa_plan* p) {
  if (!p) return;
  (void)hipStreamSynchronize(pfa::ctx().stream);
  delete p;
}

// ---------------------------------------------------------------- corpus
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t below(uint32_t n) { return n ? uint32_t(rnd() % n) : 0u; }

struct Expect {
  int status = 0;                       // 0, PFA_FILE_NOT_FASTA, PFA_FILE_UNEQUAL, PFA_ERR_IO
  std::vector<std::string> names, rows; // rows upper-cased as readfasta returns them
  bool bad = false;                     // a byte >= 0x80 in a sequence line
};

static Expect write_file(const std::string& path, uint32_t kind) {
  Expect e;
  std::string text;
  if (kind == 0) {            // not FASTA
    text = "no header here\nACGT\n";
    e.status = PFA_FILE_NOT_FASTA;
  } else {
    const char* eol = below(4) == 0 ? "\r\n" : "\n";
    const uint32_t n = 1 + below(kind == 3 ? 6 : 40);
    const uint32_t L = kind == 3 ? 150000 + below(100000) : (below(8) == 0 ? below(4) : below(3000));
    const bool unequal = kind == 2 && n > 1;
    const bool dirty_file = below(3) == 0;
    if (below(6) == 0) text += "junk before the first header\n";
    for (uint32_t r = 0; r < n; ++r) {
      const uint32_t len = unequal && r == n - 1 ? L + 1 : L;
      std::string seq;
      const bool dirty = dirty_file && below(3) == 0;
      for (uint32_t k = 0; k < len; ++k) {
        char c = "ACGTacgt"[below(8)];
        if (dirty && below(50) == 0) c = "N-nRY?"[below(6)];
        if (kind == 4 && r == 0 && k == len / 2) c = (char)0xC3;
        seq += c;
      }
      std::string name = (r % 2 ? "pop2_ind" : "pop1_ind") + std::to_string(r);
      text += ">" + name + (below(5) == 0 ? "  " : "") + eol;
      const uint32_t w = below(3) == 0 ? 60 + below(40) : 1u << 30;
      for (size_t k = 0; k < seq.size(); k += w) text += seq.substr(k, w) + (below(7) == 0 ? " " : "") + eol;
      if (below(9) == 0) text += eol;
      std::string up = seq;
      for (char& c : up)
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
      e.names.push_back(name);
      e.rows.push_back(up);
      if (kind == 4 && r == 0 && len) e.bad = true;
    }
    if (unequal) e.status = PFA_FILE_UNEQUAL;
  }
  FILE* fh = fopen(path.c_str(), "wb");
  fwrite(text.data(), 1, text.size(), fh);
  fclose(fh);
  return e;
}

#define CHECK(cond, ...)                                              \
  do {                                                                \
    if (!(cond)) {                                                    \
      fprintf(stderr, "ingest_tsan: CHECK failed: %s: ", #cond);      \
      fprintf(stderr, __VA_ARGS__);                                   \
      fprintf(stderr, "\n");                                          \
      abort();                                                        \
    }                                                                 \
  } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 12;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  pfa::ctx().ready = true;
  pfa::ctx().device = 0;
  (void)hipStreamCreateWithFlags(&pfa::ctx().stream, hipStreamNonBlocking);
  uint64_t files_checked = 0, bytes_checked = 0;
  // every file of a batch against the independent reading
  auto check_batch = [&](const pfa_ingest_batch& b, const std::vector<Expect>& want, const std::vector<std::string>& paths, const pfa_ingest_opts& o,
                         const char* const* keys, std::vector<int>& seen, int round) {
      uint32_t u = 0;
      for (uint32_t f = 0; f < b.n_files; ++f) {
        const uint32_t gi = b.file_idx[f];
        const Expect& e = want[gi];
        seen[gi] += 1;
        CHECK(b.status[f] == e.status, "round %d file %u (%s): status %d, expected %d", round, gi, paths[gi].c_str(), b.status[f], e.status);
        if (e.status != 0) continue;
        CHECK(b.n_rows[f] == e.rows.size() && b.n_sites[f] == e.rows[0].size(), "round %d file %u: %u x %u, expected %zu x %zu", round, gi, b.n_rows[f],
              b.n_sites[f], e.rows.size(), e.rows[0].size());
        CHECK((b.bad_bytes[f] != 0) == e.bad, "round %d file %u: bad flag %u, expected %d", round, gi, b.bad_bytes[f], (int)e.bad);
        std::vector<std::vector<uint32_t>> sel;               // expected units of this file: row lists
        if (o.n_keys == 0) {
          std::vector<uint32_t> all(e.rows.size());
          for (uint32_t r = 0; r < all.size(); ++r) all[r] = r;
          sel.push_back(all);
        } else {
          for (uint32_t k = 0; k < o.n_keys; ++k) {
            std::vector<uint32_t> rows;
            for (uint32_t r = 0; r < e.names.size(); ++r)
              if (e.names[r].find(keys[k]) != std::string::npos) rows.push_back(r);
            CHECK(b.key_rows[(size_t)f * o.n_keys + k] == rows.size(), "round %d file %u key %u: %u rows, expected %zu", round, gi, k,
                  b.key_rows[(size_t)f * o.n_keys + k], rows.size());
            if (!rows.empty()) sel.push_back(rows);
          }
        }
        for (const auto& rows : sel) {
          CHECK(u < b.n_units && b.unit_file[u] == f && b.unit_rows[u] == rows.size(), "round %d file %u: unit table", round, gi);
          const pfa_desc& d = b.plan->descs[u];
          // a unit is a matrix of its own: all records of the file, or — population split — the records its key names, in record order
          CHECK(d.n_rows == rows.size() && d.n_sites == e.rows[0].size() && d.mat_rows == rows.size() && d.row_idx_off == PFA_NO_ROW_IDX &&
                    d.row_stride % 16 == 0 && d.base_off % 256 == 0 && d.base_off + (uint64_t)d.mat_rows * d.row_stride <= b.plan->n_bytes,
                "round %d file %u: descriptor", round, gi);
          for (uint32_t k = 0; k < rows.size(); ++k) {
            const uint32_t r = rows[k];
            const uint8_t* got = b.plan->snap.data() + d.base_off + (uint64_t)k * d.row_stride;
            CHECK(memcmp(got, e.rows[r].data(), d.n_sites) == 0, "round %d file %u row %u: bytes differ (threads %u, slot %llu)", round, gi, r,
                  o.n_threads, (unsigned long long)o.slot_bytes);
            for (uint32_t c = d.n_sites; c < d.row_stride && d.n_sites; ++c) CHECK(got[c] == 0, "round %d file %u row %u: padding", round, gi, r);
            bytes_checked += d.n_sites;
          }
          ++u;
        }
        ++files_checked;
      }
      CHECK(u == b.n_units, "round %d: %u units walked, batch holds %u", round, u, b.n_units);
  };
  for (int round = 0; round < rounds; ++round) {
    // ---- a directory of awkward files
    const uint32_t n_files = round % 5 == 4 ? 1 + below(4) : 20 + below(60);      // (a handful of files: the ingest's direct mode)
    std::vector<std::string> paths;
    std::vector<Expect> want;
    for (uint32_t i = 0; i < n_files; ++i) {
      const uint32_t pick = n_files < 5 ? below(9) : below(40);     // (small jobs: every kind of file comes up often)
      const uint32_t kind = pick == 0 ? 0 : pick == 1 ? 2 : pick == 2 ? 3 : pick == 3 ? 4 : pick == 4 ? 5 : 1;
      std::string p = dir + "/r" + std::to_string(round) + "_f" + std::to_string(i) + ".fa";
      if (kind == 5) {          // a path that cannot be read (a missing file, or a directory)
        if (below(2)) { mkdir(p.c_str(), 0700); }
        Expect e;
        e.status = PFA_ERR_IO;
        want.push_back(e);
      } else {
        want.push_back(write_file(p, kind));
      }
      paths.push_back(p);
    }
    // ---- the pipeline under test
    std::vector<const char*> cpaths;
    for (auto& p : paths) cpaths.push_back(p.c_str());
    const char* keys[2] = {"pop1", "nopop"};
    pfa_ingest_opts o{};
    o.n_threads = 1 + below(16);
    o.stride_align = below(2) ? 128 : 16;
    o.n_keys = round % 3 == 0 ? 2 : 0;
    o.keys = keys;
    o.slot_bytes = (16u << 10) << below(7);          // 16 KiB .. 1 MiB: ring wrap-around, oversized files
    o.n_slots = 2 + below(4);
    pfa_ingest* ing = nullptr;
    CHECK(pfa_ingest_open(cpaths.data(), n_files, &o, &ing) == PFA_OK, "%s", g_err);
    // feed in one piece, or in grains with batches taken in between (a rank of a multi-GPU job)
    std::vector<uint32_t> order(n_files);
    for (uint32_t i = 0; i < n_files; ++i) order[i] = i;
    if (round % 2) std::reverse(order.begin(), order.end());
    const uint32_t grain = round % 4 == 3 ? 7 : n_files;
    uint32_t fed = 0;
    std::vector<int> seen(n_files, 0);
    for (;;) {
      while (fed < n_files && pfa_ingest_backlog(ing) < 2 * grain) {
        const uint32_t k = std::min(grain, n_files - fed);
        CHECK(pfa_ingest_feed(ing, order.data() + fed, k, fed + k == n_files) == PFA_OK, "%s", g_err);
        fed += k;
      }
      pfa_ingest_batch b;
      const int rc = pfa_ingest_next(ing, &b);
      CHECK(rc >= 0, "next: %s", g_err);
      if (rc == 0) break;
      (void)hipStreamSynchronize(pfa::ctx().stream);          // the "fetch": repack and scan are done
      check_batch(b, want, paths, o, keys, seen, round);
    }
    for (uint32_t i = 0; i < n_files; ++i) CHECK(seen[i] == 1, "round %d: file %u handed out %d times", round, i, seen[i]);
    double st[8];
    CHECK(pfa_ingest_stats(ing, st) == PFA_OK, "stats");
    pfa_ingest_close(ing);
    for (auto& p : paths) { unlink(p.c_str()); rmdir(p.c_str()); }
  }
  // ---- two ingests open at once, their batches taken alternately (the API allows it): B's next() runs between A's next()
  // and A's "fetch" — A's flags for bytes >= 0x80 and A's matrices must be what A's files say (each ingest stages its
  // segments and flags in a buffer of its own; the matrix buffer is shared in stream order, as on the card)
  for (int round = 0; round < std::max(2, rounds / 3); ++round) {
    std::vector<std::string> paths[2];
    std::vector<Expect> want[2];
    for (int w = 0; w < 2; ++w)
      for (uint32_t i = 0; i < 24; ++i) {
        std::string p = dir + "/x" + std::to_string(round) + "_" + std::to_string(w) + "_f" + std::to_string(i) + ".fa";
        want[w].push_back(write_file(p, (w == 0 && i % 5 == 2) ? 4 : (i == 7 ? 3 : 1)));      // A: every fifth file holds a byte >= 0x80
        paths[w].push_back(p);
      }
    pfa_ingest_opts o{};
    o.n_threads = 1 + below(8);
    o.stride_align = 128;
    o.slot_bytes = 64u << 10;
    o.n_slots = 3;
    const char* keys[1] = {""};
    pfa_ingest* ing[2] = {nullptr, nullptr};
    std::vector<int> seen[2] = {std::vector<int>(24, 0), std::vector<int>(24, 0)};
    for (int w = 0; w < 2; ++w) {
      std::vector<const char*> cp;
      for (auto& p : paths[w]) cp.push_back(p.c_str());
      CHECK(pfa_ingest_open(cp.data(), 24, &o, &ing[w]) == PFA_OK, "%s", g_err);
      CHECK(pfa_ingest_feed(ing[w], nullptr, 24, 1) == PFA_OK, "%s", g_err);
    }
    bool live[2] = {true, true};
    while (live[0] || live[1]) {
      pfa_ingest_batch b[2];
      int rc[2] = {0, 0};
      for (int w = 0; w < 2; ++w)
        if (live[w]) { rc[w] = pfa_ingest_next(ing[w], &b[w]); CHECK(rc[w] >= 0, "interleaved next: %s", g_err); live[w] = rc[w] == 1; }
      (void)hipStreamSynchronize(pfa::ctx().stream);
      for (int w = 0; w < 2; ++w)
        if (rc[w] == 1) check_batch(b[w], want[w], paths[w], o, keys, seen[w], 1000 + round);
    }
    for (int w = 0; w < 2; ++w) {
      for (uint32_t i = 0; i < 24; ++i) CHECK(seen[w][i] == 1, "interleaved round %d: file %u of ingest %d handed out %d times", round, i, w, seen[w][i]);
      pfa_ingest_close(ing[w]);
      for (auto& p : paths[w]) unlink(p.c_str());
    }
  }
  pfa::ingest_release_cache();
  (void)hipStreamDestroy(pfa::ctx().stream);
  printf("ingest_tsan: %d rounds clean, %llu files, %llu matrix bytes checked\n", rounds, (unsigned long long)files_checked, (unsigned long long)bytes_checked);
  return 0;
}
