// pfa_host.h — host-side helpers shared by the host translation units of libpfa_hip.so
// (pfa_fasta.cpp, pfa_ingest.hip); no HIP types.
#pragma once
#include <stdint.h>

#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/pfa.h"

namespace pfa {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// one record's sequence run: `len` raw bytes at text + src -> packed + dst, upper-cased, the rest
// of the row (up to its stride) zeroed
struct Seg {
  unsigned long long src;
  unsigned long long dst;
  uint32_t len;
  uint32_t file;   // index into the batch's `bad` words | kSegPacked: the run is 2 bits per base
};
static constexpr uint32_t kSegPacked = 0x80000000u;

// per-thread scratch of index_compact(), reused from file to file
struct IndexScratch {
  struct Span { uint64_t first_line, end; uint32_t n_lines; };
  std::unordered_map<std::string_view, uint32_t> by_name;
  std::vector<Span> spans;
  // out: some line that is NOT a kept sequence line (a header, text before the first header or behind an empty one) is not
  // valid UTF-8.  The reference decodes every line of a file (PolyFastA.py:239-240) and dies with UnicodeDecodeError on such
  // a file; the caller hands it back to the host language, which reads it the reference's way.  (Sequence lines are looked
  // at where their bytes are touched anyway: by the card, or by the loader for files it does not upload.)
  bool bad_text = false;
};

// strict UTF-8 as Python decodes it (no overlong forms, no surrogates, nothing above U+10FFFF)
bool valid_utf8(const uint8_t* p, uint64_t n);

// The records of one FASTA text (readfasta's rules, PolyFastA.py:227-250) in one pass; records
// written on several lines are compacted in place so that every record's sequence is one run
// of raw bytes.  Returns the record count, PFA_ERR_FORMAT for "not FASTA".
int64_t index_compact(uint8_t* text, uint64_t n_text, std::vector<pfa_fasta_row>& rows, IndexScratch& sc);

// `n` bases -> ceil(n / 4) bytes, two bits per base (A 0, C 1, T 2, G 3; case folded); false (and `dst`
// undefined) when a byte is not A/C/G/T/a/c/g/t
bool pack2_row(const uint8_t* src, uint64_t n, uint8_t* dst);

// CPUs this PROCESS may keep busy: the CPUs it may run on (sched_getaffinity) and the CPU-time quota of its control
// group (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`: the tightest limit from the process's group up to the root), both
// shared out over the `LOCAL_WORLD_SIZE` ranks of the node (the affinity share not when the rank has already bound itself to
// its own CPUs, PFA_CPUS_ARE_MINE).  A container that may run on 256 CPUs but has a quota of 16 CPU-seconds per second is
// frozen for the rest of every 100 ms period in which its threads used the quota up: loader pools and spinning waits
// are sized by this number, not by the CPU count.  PFA_CPUS=x overrides it.
double cpu_share();
// the quota alone, for the whole control group (CPUs; a very large number when there is none)
double cgroup_cpu_quota(const char* cgroup_root = "/sys/fs/cgroup", const char* proc_cgroup = "/proc/self/cgroup");
// (timing experiment, PFA_PIN_LOADERS=<core stride>) pin the calling thread to the hardware threads of ONE physical core of the process's
// affinity mask: the `k`-th distinct core, counted round the mask.  False when the topology cannot be read.
bool pin_to_core(uint32_t k);
// The default for the ingest's loader threads: the calling thread may run on the CPUs of the mask that share the `k`-th
// last-level cache (a CCD of an EPYC: 8 cores and their sibling threads), counted round the mask — spread over the domains,
// free inside one.  False (and nothing done) when the mask holds one such domain only or sysfs does not say.
bool pin_to_l3(uint32_t k);
std::vector<std::vector<int>> l3_domains(const char* sys_root);

}  // namespace pfa
