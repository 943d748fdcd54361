// pfa_ingest.hip — streaming ingest of libpfa_hip.so: FASTA files -> device-resident matrices ->
// launched scan plans (include/pfa.h: pfa_ingest_*).  The feeder of the reference's file loop
// (PolyFastA.py:104-140; readfasta :227-250; population split :122-134).
//
// What limits files -> CSV is the PCIe link (57 GB/s measured on this pool), provided the host
// touches every byte once and the card never waits for a whole batch:
//   loader threads  claim the files in feed order, read() each one ONCE into a buffer that stays in
//                   their core's L2, index it in place (index_compact: one memchr pass) and match the
//                   --pops keys against its record names; then every row is turned into its TRANSPORT
//                   form — two bits per base when it holds nothing but A/C/G/T (pack2_row, AVX2), the
//                   bytes as they are otherwise — and the file's rows are copied, in feed order, into a
//                   ring of pinned slots.  The link then carries a quarter of the text: 57 GB/s of PCIe
//                   (measured) stop being the bound of files -> CSV; the host's read() rate is.
//                   (16 threads read 110 GB/s from tmpfs; more only fight over the memory controllers:
//                   tools/ubench/ingest_probe.hip.  PFA_INGEST_TRANSPORT=raw: no packing, for A/B runs.)
//   uploader thread copies a slot to the card as soon as its files have landed, on its own stream,
//                   two copies in flight at most: the link never idles while there is text.
//   pfa_ingest_next (the caller's thread) takes whatever has been uploaded since its last call,
//                   lets the card turn the raw text into row-major, stride-aligned, upper-case
//                   matrices (repack_kernel below: 1 MB of HBM traffic per 500 KB alignment,
//                   nothing at 6 TB/s), and creates and launches the scan plan of those files.
// The rows of a record reach the card as (offset, length, form) triples; expansion of packed rows,
// upper-casing of the others and the test for bytes >= 0x80 (outside the one-byte-per-base contract)
// happen in repack_kernel, which writes every byte of the matrices anyway.  Nothing here scans on the CPU.
#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pfa_common.h"
#include "pfa_host.h"

namespace pfa {

#ifdef PFA_INGEST_HOST_TEST
// CPU sanitizer build of the pipeline (tests/ingest_tsan.cpp, with tests/hip_mock.h in place of the HIP runtime):
// the harness supplies a host double of the repack launch
int launch_repack(const uint8_t* text, const Seg* segs, uint32_t n_segs, uint8_t* packed, uint32_t stride_align, uint32_t* bad, hipStream_t st);
#else
__device__ inline uint32_t swar_upper(uint32_t w) {
  const uint32_t h = w & 0x7F7F7F7Fu;
  const uint32_t ge_a = h + 0x1F1F1F1Fu;   // bit 7 of a byte: its low seven bits >= 'a'
  const uint32_t gt_z = h + 0x05050505u;   // ... > 'z'
  const uint32_t lower = ge_a & ~gt_z & ~w & 0x80808080u;
  return w ^ (lower >> 2);
}

// One workgroup per record.  The source run starts at any byte: every thread reads the five
// aligned words that cover its 16 output bytes and shifts them into place (v_alignbyte); the
// destination rows are 16-byte aligned by construction (matrices on 256 bytes, strides multiples
// of 16).  `bad` lives in pinned host memory (written only when a byte >= 0x80 turns up).
__global__ void __launch_bounds__(256) repack_kernel(const uint8_t* __restrict__ text, const Seg* __restrict__ segs, uint32_t n_segs,
                                                     uint8_t* __restrict__ packed, uint32_t stride_align, uint32_t* bad) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(4))) V4 { u32x4 v; };
  for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
    const Seg g = segs[s];
    const uint8_t* const src = text + g.src;
    uint8_t* const dst = packed + g.dst;
    const uint32_t len = g.len;
    uint32_t stride = (len + stride_align - 1u) & ~(stride_align - 1u);
    if (stride < 16u) stride = 16u;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
    uint32_t seen = 0u;
    if (g.file & kSegPacked) {
      // two bits per base (pack2_row: A 0, C 1, T 2, G 3): one dword in, sixteen upper-case bytes out.
      // The four codes of a byte are spread over the four selector bytes of a v_perm_b32 that picks
      // 'A' 'C' 'T' 'G' out of one constant word: nibbles to 16-bit spacing (b * 0x1001: the two copies do
      // not overlap, so the product is their OR), then codes to 8-bit spacing (t * 0x41, likewise).
      const uint32_t* const in = reinterpret_cast<const uint32_t*>(src);   // rows start on 16 bytes
      for (uint32_t i = threadIdx.x * 16u; i < stride; i += 256u * 16u) {
        u32x4 o = {0u, 0u, 0u, 0u};
        if (i < len) {
          const uint32_t w = in[i >> 4];
          uint32_t q[4];
#pragma unroll
          for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t t = (((w >> (8u * k)) & 0xFFu) * 0x1001u) & 0x000F000Fu;
            const uint32_t sel = (t * 0x41u) & 0x03030303u;
            const uint32_t have = len - i > 4u * k ? len - i - 4u * k : 0u;
            q[k] = __builtin_amdgcn_perm(0u, 0x47544341u, sel) & (have >= 4u ? 0xFFFFFFFFu : (1u << (8u * have)) - 1u);
          }
          o = u32x4{q[0], q[1], q[2], q[3]};
        }
        *reinterpret_cast<u32x4*>(dst + i) = o;
      }
      continue;
    }
    for (uint32_t i = threadIdx.x * 16u; i < stride; i += 256u * 16u) {
      u32x4 o = {0u, 0u, 0u, 0u};
      if (i < len) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(src + i) & ~uintptr_t(3));
        const u32x4 a = reinterpret_cast<const V4*>(p)->v;
        const uint32_t a4 = sh ? p[4] : 0u;
        uint32_t w[4];
        w[0] = __builtin_amdgcn_alignbyte(a.y, a.x, sh);
        w[1] = __builtin_amdgcn_alignbyte(a.z, a.y, sh);
        w[2] = __builtin_amdgcn_alignbyte(a.w, a.z, sh);
        w[3] = __builtin_amdgcn_alignbyte(a4, a.w, sh);
        const uint32_t left = len - i;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
          const uint32_t have = left > 4u * k ? left - 4u * k : 0u;   // valid bytes of word k
          const uint32_t m = have >= 4u ? 0xFFFFFFFFu : (1u << (8u * have)) - 1u;
          w[k] &= m;
          seen |= w[k];
          w[k] = swar_upper(w[k]);
        }
        o = u32x4{w[0], w[1], w[2], w[3]};
      }
      *reinterpret_cast<u32x4*>(dst + i) = o;
    }
    if (seen & 0x80808080u) bad[g.file] = 1u;   // (kSegPacked is clear here)
  }
}

void warm_ingest() {   // pfa_prewarm: load this translation unit's code object
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(repack_kernel));
  (void)hipGetLastError();
}

static int launch_repack(const uint8_t* text, const Seg* segs, uint32_t n_segs, uint8_t* packed, uint32_t stride_align, uint32_t* bad, hipStream_t st) {
  hipLaunchKernelGGL(repack_kernel, dim3(n_segs < (1u << 20) ? n_segs : (1u << 20)), dim3(256), 0, st, text, segs, n_segs, packed, stride_align, bad);
  PFA_HIP(hipGetLastError());
  return PFA_OK;
}
#endif

namespace {

// a slot = one H2D copy of TRANSPORT bytes (a quarter of the text for clean A/C/G/T rows): 16 MiB is
// 128 alignments of 50 x 10 kb and copies at 55 of the link's 57.5 GB/s
constexpr size_t kDefaultSlot = size_t(16) << 20;
constexpr size_t kMinSlot = size_t(256) << 10;
constexpr size_t kRampSlot = size_t(2) << 20;
constexpr uint32_t kDefaultSlots = 8;
constexpr size_t kMaxBatchBytes = size_t(128) << 20;
constexpr unsigned long long kDirectMaxText = 256ull << 20;   // direct mode: most text (bytes of the files) of a job read in place by the card
constexpr uint64_t kRowPacked = 1ull << 63;   // FileRec::row_off: the row travels two bits per base
inline size_t up16(size_t x) { return (x + 15u) & ~size_t(15); }
constexpr int kBatchEvents = 64;

inline size_t up256(size_t x) { return (x + 255u) & ~size_t(255); }
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pinned { uint8_t* h = nullptr; size_t cap = 0; };

// Buffers outlive an ingest: hipHostMalloc costs 130-200 ms per GB on this pool, and a process
// (the CLI per rank, the bench, the tests) opens one ingest after the other.
struct Cache {
  std::mutex mu;
  std::vector<Pinned> pinned;
  uint8_t* d_text = nullptr; size_t d_text_cap = 0;
  uint8_t* d_packed = nullptr; size_t d_packed_cap = 0;
  uint8_t* d_segs = nullptr; size_t d_segs_cap = 0;
  Pinned stage;   // segs + bad words of one batch: parked here between ingests; an open ingest holds its own (two ingests open
                  // at once must not share it: the second one's next() would clear the first one's flags for bytes >= 0x80
                  // before its caller has fetched them, and rewrite segments whose H2D copy is still in flight)
  void release() {
    std::lock_guard<std::mutex> g(mu);
    for (Pinned& p : pinned) (void)hipHostFree(p.h);
    pinned.clear();
    if (stage.h) (void)hipHostFree(stage.h);
    stage = Pinned();
    (void)hipFree(d_text);
    (void)hipFree(d_packed);
    (void)hipFree(d_segs);
    d_text = d_packed = d_segs = nullptr;
    d_text_cap = d_packed_cap = d_segs_cap = 0;
  }
};
Cache g_cache;

struct FileRec {
  int32_t status = 0;
  uint32_t n_rows = 0, n_sites = 0;
  uint32_t use = 0;
  std::vector<uint64_t> row_off;    // per record: offset of its transport run in the use's buffer | kRowPacked (status 0)
  std::vector<uint32_t> key_cnt;    // per key: matching records
  std::vector<uint32_t> key_rows;   // their indices, key after key
};

// one filling of a pinned slot = one H2D copy
struct Use {
  uint8_t* h = nullptr;
  uint8_t* d = nullptr;
  size_t cap = 0, used = 0;
  std::vector<uint32_t> pos;   // feed positions of its files, in the order they were placed
  uint32_t pending = 0;   // files placed here that are still being read / indexed
  int host_slot = -1, dev_slot = -1;
  bool big = false, sealed = false, ready = false, enqueued = false, consumed = false;
  hipEvent_t ev = nullptr;        // behind the copy
  hipEvent_t repack_ev = nullptr; // behind the repack of the batch that consumed it
};

}  // namespace

void ingest_release_cache() { g_cache.release(); }

}  // namespace pfa

struct pfa_ingest {
  // ---- immutable after open
  std::vector<std::string> paths;
  std::string names_blob;
  std::vector<uint64_t> names_off;
  std::vector<std::string> keys;
  uint32_t stride_align = 128, unit_flags = 0, want_threads = 0, n_slots = pfa::kDefaultSlots, n_dev_slots = 2 * pfa::kDefaultSlots;
  bool pack_rows = true;
  // a handful of files (the one-file runs of BASELINE configs[1..3]): no copy stream (an HSA queue: 20 ms), no copy engine
  // (9 ms of SDMA set-up), no device text ring — repack_kernel reads the rows out of the pinned slots over PCIe
  bool direct = false;
  size_t max_slot = pfa::kDefaultSlot;
  int device = 0;

  // ---- pipeline state (mu)
  std::mutex mu;
  std::condition_variable cv_work, cv_slot, cv_open, cv_up, cv_batch, cv_consumed;
  bool opening = false;                // a loader is opening the next use
  std::vector<uint32_t> feed;          // feed order: indices into paths
  std::deque<pfa::FileRec> files;      // by feed position
  bool feed_closed = false, stop = false;
  uint32_t next_claim = 0, n_placed = 0, n_handed = 0;   // files claimed by a loader / placed in the ring / handed out in batches
  std::deque<pfa::Use> uses;
  int open_use = -1;
  size_t slot_bytes = 0;               // decided by the first reservation
  std::vector<pfa::Pinned> slots;      // the ring of pinned slots
  std::vector<int> slot_user, dev_user;   // use that holds the slot, -1: free
  uint64_t ring_seq = 0;
  uint8_t* d_text = nullptr; size_t d_text_cap = 0;
  uint32_t uploaded = 0, consumed = 0;    // uses whose copy is enqueued / that a batch has taken
  std::string hard_error;
  int hard_rc = 0;
  std::vector<hipEvent_t> ev_pool;

  std::vector<std::thread> workers;
  std::thread uploader;
  bool threads_started = false;
  hipStream_t copy_stream = nullptr;
  hipStream_t copy_streams[4] = {};   // copies alternate over `n_streams` of them (copy_stream = the first)
  uint32_t n_streams = 1, max_inflight = 2;
  hipEvent_t batch_ev[pfa::kBatchEvents] = {};
  uint32_t n_batches = 0;

  // ---- the current batch (caller's thread)
  pfa_plan* plan = nullptr;
  std::vector<uint32_t> b_file_idx, b_n_rows, b_n_sites, b_key_rows, b_unit_file, b_unit_key, b_unit_rows, b_ridx;
  std::vector<int32_t> b_status;
  std::vector<pfa_desc> b_descs;
  std::vector<pfa::Pinned> big_pinned_done;   // pinned buffers of oversized files whose copy has completed
  pfa::Pinned stage;                          // segments + per-file flags of the current batch (pinned; from / back to the cache)
  std::vector<uint8_t*> big_dev;              // device buffers of oversized files of the current batch
  uint32_t* bad_words = nullptr;
  bool launched = false;                      // something of the current batch sits on the context stream

  // ---- statistics
  std::atomic<uint64_t> ns_read{0}, ns_index{0}, bytes_read{0}, ns_copy{0}, bytes_copied{0}, ns_main{0}, ns_wait{0};
};

namespace pfa {
namespace {

void fail(pfa_ingest* g, int rc, const std::string& msg) {   // mu held
  if (g->hard_rc == 0) { g->hard_rc = rc; g->hard_error = msg; }
  g->stop = true;
  g->cv_work.notify_all(); g->cv_slot.notify_all(); g->cv_open.notify_all();
  g->cv_up.notify_all(); g->cv_batch.notify_all(); g->cv_consumed.notify_all();
}

hipEvent_t take_event(pfa_ingest* g) {   // mu held
  if (!g->ev_pool.empty()) { hipEvent_t e = g->ev_pool.back(); g->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

bool all_reserved(const pfa_ingest* g) { return g->n_placed == g->feed.size(); }

void seal(pfa_ingest* g, int u) {   // mu held
  if (u < 0) return;
  Use& s = g->uses[(size_t)u];
  if (s.sealed) return;
  s.sealed = true;
  if (s.pending == 0) { s.ready = true; g->cv_up.notify_all(); }
}

Pinned take_pinned(size_t want) {
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    size_t best = g_cache.pinned.size();
    for (size_t i = 0; i < g_cache.pinned.size(); ++i)
      if (g_cache.pinned[i].cap >= want && g_cache.pinned[i].cap <= 4 * want && (best == g_cache.pinned.size() || g_cache.pinned[i].cap < g_cache.pinned[best].cap)) best = i;
    if (best != g_cache.pinned.size()) {
      Pinned p = g_cache.pinned[best];
      g_cache.pinned.erase(g_cache.pinned.begin() + (long)best);
      return p;
    }
  }
  Pinned p;
  if (hipHostMalloc((void**)&p.h, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p.h = nullptr; return p; }
  p.cap = want;
  return p;
}

void give_pinned(Pinned p) {
  if (!p.h) return;
  std::lock_guard<std::mutex> lk(g_cache.mu);
  size_t total = p.cap;
  for (const Pinned& q : g_cache.pinned) total += q.cap;
  if (g_cache.pinned.size() >= 16 || total > (size_t(1) << 30)) { (void)hipHostFree(p.h); return; }
  g_cache.pinned.push_back(p);
}

// a new use for a file of `need` bytes; mu held through `lk` (released while it waits or allocates)
int new_use(pfa_ingest* g, std::unique_lock<std::mutex>& lk, size_t need, uint32_t pos) {
  if (g->slot_bytes == 0) {
    // the first file sizes the ring: small inputs get small slots (a one-file run pins 256 KiB, not 64 MiB)
    const size_t remaining = g->feed.size() - pos;
    size_t est = need * std::max<size_t>(remaining, 1);
    est = (est + kMinSlot - 1) / kMinSlot * kMinSlot;
    // (the first file to arrive may be the smallest of a directory — an empty or rejected one even: unless the job is a
    // handful of files, slots are not sized below 4 MiB; they are allocated one by one as they come into use)
    if (remaining > 8) est = std::max(est, size_t(4) << 20);
    g->slot_bytes = std::min(std::max(est, kMinSlot), g->max_slot);
    const size_t want = g->direct ? 0 : (size_t)g->n_dev_slots * g->slot_bytes + 256u;
    if (want) {
      std::lock_guard<std::mutex> ck(g_cache.mu);
      if (g_cache.d_text_cap >= want) { g->d_text = g_cache.d_text; g->d_text_cap = g_cache.d_text_cap; g_cache.d_text = nullptr; g_cache.d_text_cap = 0; }
    }
    if (want && !g->d_text) {
      if (hipMalloc((void**)&g->d_text, want) != hipSuccess) { (void)hipGetLastError(); fail(g, PFA_ERR_NOMEM, "ingest: cannot allocate the device text ring"); return -1; }
      g->d_text_cap = want;
    }
  }
  Use u;
  if (need > g->slot_bytes) {   // a file larger than a slot travels alone, in buffers of its own
    u.big = true;
    lk.unlock();
    Pinned p = take_pinned(need);
    uint8_t* d = nullptr;
    const bool ok = p.h && (g->direct ? hipHostGetDevicePointer((void**)&d, p.h, 0) == hipSuccess : hipMalloc((void**)&d, need + 256u) == hipSuccess);
    lk.lock();
    if (!ok) { (void)hipGetLastError(); give_pinned(p); fail(g, PFA_ERR_NOMEM, "ingest: cannot allocate buffers for a " + std::to_string(need) + "-byte file"); return -1; }
    u.h = p.h; u.cap = p.cap; u.d = d;
  } else {
    // direct mode (small jobs: the card reads the pinned buffers itself, nothing is recycled): the "ring" grows by one buffer per
    // use, each as large as the use may get
    if (g->direct && g->ring_seq >= g->slots.size()) {
      g->slots.resize((size_t)g->ring_seq + 1u);
      g->slot_user.resize((size_t)g->ring_seq + 1u, -1);
    }
    const int hs = g->direct ? (int)g->ring_seq : (int)(g->ring_seq % g->n_slots);
    const int ds = g->direct ? 0 : (int)(g->ring_seq % g->n_dev_slots);
    g->ring_seq += 1;
    while (!g->stop && g->slot_user[(size_t)hs] >= 0) g->cv_slot.wait(lk);   // its last copy is still in flight
    if (g->stop) return -1;
    // the first slots are closed early (2, 4, 8 ... MiB): the link starts on the first few files, not on the first 64 MiB
    u.cap = std::min(g->slot_bytes, std::max(need, kRampSlot << std::min<uint64_t>(g->ring_seq - 1, 8)));
    if (!g->slots[(size_t)hs].h) {
      const size_t want = g->direct ? u.cap : g->slot_bytes;
      lk.unlock();
      Pinned p = take_pinned(want);
      lk.lock();
      if (!p.h) { fail(g, PFA_ERR_NOMEM, "ingest: cannot allocate a pinned slot"); return -1; }
      g->slots[(size_t)hs] = p;
    }
    u.h = g->slots[(size_t)hs].h;
    u.host_slot = hs;
    u.dev_slot = ds;
    if (g->direct) {
      if (hipHostGetDevicePointer((void**)&u.d, u.h, 0) != hipSuccess) { fail(g, PFA_ERR_HIP, "ingest: hipHostGetDevicePointer failed"); return -1; }
    } else {
      u.d = g->d_text + (size_t)ds * g->slot_bytes;
    }
  }
  u.ev = take_event(g);
  if (!u.ev) { fail(g, PFA_ERR_HIP, "ingest: hipEventCreate failed"); return -1; }
  g->uses.push_back(u);
  const int idx = (int)g->uses.size() - 1;
  if (u.host_slot >= 0) g->slot_user[(size_t)u.host_slot] = idx;
  return idx;
}

// The file of feed position `pos` gets `size` bytes in the pinned ring, in the order the loaders finish
// (NOT in feed order: handing a turn from thread to sleeping thread costs ~10 us, i.e. caps the ingest
// at ~100 k files/s, and every loader would run at the pace of the slowest file before it).  A batch is
// therefore a SET of feed positions; whoever prints rows in file order holds the stragglers' successors
// back for a batch (cli.run_files).  Returns false when the ingest is stopping.
bool reserve(pfa_ingest* g, uint32_t pos, size_t size, uint32_t* use_out, size_t* off_out, uint8_t** text_out) {
  const size_t need = up256(std::max<size_t>(size, 1));
  std::unique_lock<std::mutex> lk(g->mu);
  int u;
  for (;;) {
    if (g->stop) return false;
    if (g->opening) { g->cv_open.wait(lk); continue; }     // somebody is opening the next use (it may wait for a free slot)
    u = g->open_use;
    if (u >= 0 && need <= g->slot_bytes && g->uses[(size_t)u].used + need <= g->uses[(size_t)u].cap) break;
    // this file starts a new use (or, larger than a slot, travels in buffers of its own); one thread at a time
    g->opening = true;
    if (u < 0 || need <= g->slot_bytes) { seal(g, g->open_use); g->open_use = -1; }
    u = new_use(g, lk, need, pos);
    g->opening = false;
    g->cv_open.notify_all();
    if (u < 0) return false;
    if (!g->uses[(size_t)u].big) g->open_use = u;
    break;
  }
  Use& s = g->uses[(size_t)u];
  *use_out = (uint32_t)u;
  *off_out = s.used;
  *text_out = s.h + s.used;
  s.used += need;
  s.pending += 1;
  s.pos.push_back(pos);
  g->files[pos].use = (uint32_t)u;
  if (s.big) seal(g, u);
  g->n_placed += 1;
  // nothing further to place for now: let what has been read travel (a rank between two claims
  // of the shared queue must not sit on a half-filled slot)
  if (all_reserved(g)) { seal(g, g->open_use); g->open_use = -1; }
  return true;
}

void finish(pfa_ingest* g, uint32_t use) {
  std::lock_guard<std::mutex> lk(g->mu);
  Use& s = g->uses[use];
  s.pending -= 1;
  if (s.sealed && s.pending == 0) { s.ready = true; g->cv_up.notify_all(); }
}

void worker_body(pfa_ingest* g) {
  (void)hipSetDevice(g->device);
  IndexScratch sc;
  std::vector<pfa_fasta_row> rows;
  std::vector<uint8_t> text, image;    // the file as read; its rows in transport form (reused from file to file)
  for (;;) {
    uint32_t pos;
    FileRec* fp;
    uint32_t path_idx;
    {
      // (the containers grow under the lock: element addresses are taken under it as well)
      std::unique_lock<std::mutex> lk(g->mu);
      while (!g->stop && g->next_claim >= g->feed.size() && !g->feed_closed) g->cv_work.wait(lk);
      if (g->stop || g->next_claim >= g->feed.size()) return;
      pos = g->next_claim++;
      fp = &g->files[pos];
      path_idx = g->feed[pos];
    }
    FileRec& f = *fp;
    const std::string& path = g->paths[path_idx];
    int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    size_t size = 0;
    if (fd >= 0) {
      // only regular files have a size to trust (a directory opens fine)
      struct stat sb;
      if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 0 || (uint64_t)sb.st_size > (1ull << 40)) { close(fd); fd = -1; }
      else size = (size_t)sb.st_size;
    }
    if (fd < 0) f.status = PFA_ERR_IO;
    size_t image_bytes = 0;
    if (fd >= 0) {
      const double t0 = now_s();
      if (text.size() < size) text.resize(size);
      size_t got = 0;
      while (got < size) {
        const ssize_t r = read(fd, text.data() + got, size - got);
        if (r <= 0) break;
        got += (size_t)r;
      }
      close(fd);
      if (got != size) f.status = PFA_ERR_IO;
      const double t1 = now_s();
      g->ns_read += (uint64_t)((t1 - t0) * 1e9);
      g->bytes_read += got;
      if (f.status == 0) {
        const int64_t n = index_compact(text.data(), size, rows, sc);
        if (n == PFA_ERR_FORMAT && strncmp(pfa_last_error(), "not FASTA", 9) == 0) {
          // (the reference decodes the whole file before it says so: bytes that are not text make it die first, :239-240)
          f.status = valid_utf8(text.data(), size) ? PFA_FILE_NOT_FASTA : PFA_FILE_UNSUPPORTED;
        } else if (n < 0) {
          std::lock_guard<std::mutex> lk(g->mu);
          fail(g, (int)n, path + ": " + pfa_last_error());
        } else {
          f.n_rows = (uint32_t)n;
          f.n_sites = n ? rows[0].seq_len : 0u;
          for (const pfa_fasta_row& r : rows)
            if (r.seq_len != f.n_sites) { f.status = PFA_FILE_UNEQUAL; break; }   // PolyFastA.py:111-112
          if (f.status == PFA_FILE_UNEQUAL) {
            // byte lengths say nothing about a file whose sequences hold non-ASCII characters (the reference counts
            // characters): hand it back as such, the caller reads it as text
            for (const pfa_fasta_row& r : rows) {
              const uint8_t* q = text.data() + r.seq_off;
              bool high = false;
              for (size_t i = 0; i < r.seq_len && !high; ++i) high = q[i] >= 0x80u;
              if (high) { f.status = PFA_FILE_UNSUPPORTED; break; }
            }
          }
          // a header (or a line nobody keeps) that is not valid UTF-8: the reference dies decoding it (PolyFastA.py:239-240);
          // the caller reads the file the reference's way and does the same
          if (sc.bad_text) f.status = PFA_FILE_UNSUPPORTED;
          if (f.status == 0) {
            // the rows in transport form, each on a 16-byte boundary: two bits per base when the row is
            // clean A/C/G/T, its bytes otherwise
            const size_t L = f.n_sites, raw16 = up16(L), packed16 = up16((L + 3u) / 4u);
            if (image.size() < rows.size() * raw16 + 64u) image.resize(rows.size() * raw16 + 64u);
            f.row_off.resize(rows.size());
            for (size_t i = 0; i < rows.size(); ++i) {
              const uint8_t* src = text.data() + rows[i].seq_off;
              uint8_t* dst = image.data() + image_bytes;
              if (L && g->pack_rows && pack2_row(src, L, dst)) {
                f.row_off[i] = image_bytes | kRowPacked;
                image_bytes += packed16;
              } else {
                if (L) memcpy(dst, src, L);
                f.row_off[i] = image_bytes;
                image_bytes += raw16;
              }
            }
            if (!g->keys.empty()) {   // PolyFastA.py:125: `pop in header`, a substring match, in record order
              f.key_cnt.assign(g->keys.size(), 0u);
              for (size_t k = 0; k < g->keys.size(); ++k) {
                const std::string& key = g->keys[k];
                for (size_t i = 0; i < rows.size(); ++i) {
                  const char* nm = reinterpret_cast<const char*>(text.data() + rows[i].name_off);
                  if (key.empty() || memmem(nm, rows[i].name_len, key.data(), key.size()) != nullptr) {
                    f.key_rows.push_back((uint32_t)i);
                    f.key_cnt[k] += 1;
                  }
                }
              }
            }
          }
        }
        g->ns_index += (uint64_t)((now_s() - t1) * 1e9);
      }
    }
    // its place in the pinned ring, in feed order (batches are consecutive runs of the feed order)
    uint32_t use = 0;
    size_t off = 0;
    uint8_t* dst = nullptr;
    const double tw = now_s();
    if (!reserve(g, pos, image_bytes, &use, &off, &dst)) return;
    const double tc = now_s();
    g->ns_wait += (uint64_t)((tc - tw) * 1e9);
    if (image_bytes) memcpy(dst, image.data(), image_bytes);
    for (uint64_t& o : f.row_off) o += off;       // (the flag bit stays where it is)
    g->ns_index += (uint64_t)((now_s() - tc) * 1e9);
    finish(g, use);
  }
}

void worker(pfa_ingest* g, uint32_t t) {
  try {
    // A single process's loaders are spread over the last-level-cache domains of its CPUs (pfa_host.h): two boxes of three gained
    // 10-40 % in the median run, the third lost 1-2 % (profiles/r04_loader_l3_spread_ab.txt).  The ranks of a multi-GPU job
    // (4 loaders each inside a 16-CPU quota) measured 4-7 % slower with it, inside a wide spread: left to the scheduler.
    // PFA_PIN_LOADERS=0: never, =l3: always, =<n>: timing experiment, one physical core each, n cores apart.
    const char* const pin = getenv("PFA_PIN_LOADERS");
    const char* const ws = getenv("WORLD_SIZE");
    // (a pool of one to three loaders — a run on a few files — has nothing to spread, and reading the topology costs a millisecond)
    if (pin ? strcmp(pin, "l3") == 0 : !(ws && atoi(ws) > 1) && g->want_threads >= 4u) (void)pin_to_l3(t);
    else if (pin && atoi(pin) > 0) (void)pin_to_core(t * (uint32_t)atoi(pin));
    worker_body(g);
  } catch (const std::bad_alloc&) {
    std::lock_guard<std::mutex> lk(g->mu);
    fail(g, PFA_ERR_NOMEM, "ingest: out of host memory");
  } catch (...) {
    std::lock_guard<std::mutex> lk(g->mu);
    fail(g, PFA_ERR_ARG, "ingest: unexpected exception in a loader thread");
  }
}

void uploader_body(pfa_ingest* g) {
  (void)hipSetDevice(g->device);
  uint32_t next_up = 0;
  std::deque<uint32_t> inflight;
  double t_busy0 = 0.0;
  std::unique_lock<std::mutex> lk(g->mu);
  auto reap = [&]() {   // wait for the oldest copy; its pinned slot is free again
    const uint32_t u = inflight.front();
    hipEvent_t ev = g->uses[u].ev;
    lk.unlock();
    const hipError_t e = hipEventSynchronize(ev);
    lk.lock();
    if (e != hipSuccess) { fail(g, PFA_ERR_HIP, std::string("ingest: H2D copy failed: ") + hipGetErrorString(e)); return; }
    Use& s = g->uses[u];
    if (s.host_slot >= 0) { g->slot_user[(size_t)s.host_slot] = -1; g->cv_slot.notify_all(); }
    else { g->big_pinned_done.push_back(Pinned{s.h, s.cap}); s.h = nullptr; }   // (handed on exactly once: close() only gives what is left)
    inflight.pop_front();
    if (inflight.empty()) g->ns_copy += (uint64_t)((now_s() - t_busy0) * 1e9);
  };
  for (;;) {
    if (g->stop) break;
    const bool have = next_up < g->uses.size() && g->uses[next_up].ready;
    if (have && g->direct) {       // nothing to copy: the card reads the pinned slot itself
      g->uses[next_up].enqueued = true;
      next_up += 1;
      g->uploaded = next_up;
      g->cv_batch.notify_all();
      continue;
    }
    if (have && inflight.size() < g->max_inflight) {
      if (!g->copy_stream) {
        lk.unlock();
        hipError_t e = hipSuccess;
        for (uint32_t k = 0; k < g->n_streams && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&g->copy_streams[k], hipStreamNonBlocking);
        lk.lock();
        if (e != hipSuccess) { fail(g, PFA_ERR_HIP, "ingest: cannot create the copy stream"); break; }
        g->copy_stream = g->copy_streams[0];
      }
      Use& s = g->uses[next_up];
      hipEvent_t wait_ev = nullptr;
      if (!s.big) {
        // the device slot's previous text must have been repacked (its batch taken by the caller)
        const int prev = g->dev_user[(size_t)s.dev_slot];
        if (prev >= 0 && !g->uses[(size_t)prev].consumed) {
          if (!inflight.empty()) { reap(); continue; }
          g->cv_consumed.wait(lk);
          continue;
        }
        if (prev >= 0) wait_ev = g->uses[(size_t)prev].repack_ev;
        g->dev_user[(size_t)s.dev_slot] = (int)next_up;
      }
      uint8_t* const d = s.d;
      uint8_t* const h = s.h;
      const size_t n = s.used;
      hipEvent_t ev = s.ev;
      hipStream_t cs = g->copy_streams[next_up % g->n_streams];
      lk.unlock();
      hipError_t e = hipSuccess;
      if (wait_ev) e = hipStreamWaitEvent(cs, wait_ev, 0);
      if (e == hipSuccess && n) e = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, cs);
      if (e == hipSuccess) e = hipEventRecord(ev, cs);
      lk.lock();
      if (e != hipSuccess) { fail(g, PFA_ERR_HIP, std::string("ingest: cannot enqueue an H2D copy: ") + hipGetErrorString(e)); break; }
      if (inflight.empty()) t_busy0 = now_s();
      g->bytes_copied += n;
      g->uses[next_up].enqueued = true;
      inflight.push_back(next_up);
      next_up += 1;
      g->uploaded = next_up;
      g->cv_batch.notify_all();
      continue;
    }
    if (!inflight.empty()) { reap(); continue; }
    if (g->feed_closed && all_reserved(g) && next_up == g->uses.size()) break;   // everything is on the card
    g->cv_up.wait(lk);
  }
  while (!inflight.empty() && g->hard_rc == 0) reap();
}

void uploader(pfa_ingest* g) {
  try {
    uploader_body(g);
  } catch (...) {
    std::lock_guard<std::mutex> lk(g->mu);
    fail(g, PFA_ERR_ARG, "ingest: unexpected exception in the uploader thread");
  }
}

uint32_t auto_threads() {
  if (const char* e = getenv("PFA_THREADS")) { const long v = strtol(e, nullptr, 10); if (v > 0) return (uint32_t)std::min<long>(v, 256); }
  // The CPUs this rank may keep busy (pfa_fasta.cpp: cpu_share — affinity and the control group's CPU quota, shared out over
  // the ranks of the node).  Round 4: the GPU boxes of this pool may run on 256 CPUs but have a quota of 16 CPU-seconds per
  // second; 4 ranks x 16 loaders on such a box spent most of every 100 ms period frozen by the scheduler
  // (profiles/r04_rank_rehearsals_one_gpu.txt).  One process inside that quota: 16 loaders use 14.8 CPUs and are never
  // throttled — 218 / 213 k alignments/s against 205 / 198 k with 14 and 187 k with 12 (profiles/r04_e2e_cpu_budget.txt) — so
  // nothing is held back for the uploader and the caller's thread (PFA_CPU_RESERVE=x does).
  // At most 16 (profiles/r03_e2e_16_vs_20_vs_24_threads.txt, six alternating rounds): 16 loaders 219-230 k alignments/s best /
  // 193-217 k median; 20: 217-249 k / 187-205 k; 24: 180-239 k / 170-190 k; 28 and more collapse (r03_e2e_thread_sweep.txt): the
  // host's read() is capped in aggregate, more loaders only wait for ring slots
  double reserve = 0.0;
  if (const char* e = getenv("PFA_CPU_RESERVE")) reserve = atof(e);
  const double room = cpu_share() - reserve;
  return (uint32_t)std::max(1.0, std::min(16.0, room));
}

int ensure_dev(uint8_t** p, size_t* cap, size_t need) {
  if (*cap >= need) return PFA_OK;
  if (*p) { PFA_HIP(hipStreamSynchronize(ctx().stream)); PFA_HIP(hipFree(*p)); }
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 4 + 4096;
  PFA_HIP(hipMalloc((void**)p, want));
  *cap = want;
  return PFA_OK;
}

}  // namespace
}  // namespace pfa

extern "C" {

static int ingest_open_impl(std::vector<std::string>&& paths, const pfa_ingest_opts* opts, pfa_ingest** out) {
  pfa_ingest_opts o{};
  if (opts) o = *opts;
  if (o.stride_align == 0) o.stride_align = 128;
  if (o.stride_align < 16 || (o.stride_align & (o.stride_align - 1))) { pfa::set_error("pfa_ingest_open: stride_align must be a power of two >= 16"); return PFA_ERR_ARG; }
  if (o.unit_flags & ~PFA_FLAG_CDS) { pfa::set_error("pfa_ingest_open: unknown unit flags 0x%x", o.unit_flags); return PFA_ERR_ARG; }
  if (o.n_keys && !o.keys) { pfa::set_error("pfa_ingest_open: keys is NULL"); return PFA_ERR_ARG; }
  pfa_ingest* g = nullptr;
  try {
    g = new pfa_ingest();
    g->paths = std::move(paths);
    g->names_off.reserve(g->paths.size() + 1);
    g->names_off.push_back(0);
    for (const std::string& p : g->paths) {
      const size_t sl = p.rfind('/');                        // display name = text after the last '/' (PolyFastA.py:110)
      g->names_blob.append(p, sl == std::string::npos ? 0 : sl + 1, std::string::npos);
      g->names_off.push_back(g->names_blob.size());
    }
    for (uint32_t k = 0; k < o.n_keys; ++k) g->keys.emplace_back(o.keys[k] ? o.keys[k] : "");
    g->stride_align = o.stride_align;
    g->unit_flags = o.unit_flags;
    g->want_threads = o.n_threads ? std::min(o.n_threads, 256u) : pfa::auto_threads();
    g->n_slots = o.n_slots ? std::min(std::max(o.n_slots, 2u), 64u) : pfa::kDefaultSlots;
    g->n_dev_slots = 2 * g->n_slots;
    g->max_slot = o.slot_bytes ? std::max<size_t>(pfa::up256((size_t)o.slot_bytes), 4096) : pfa::kDefaultSlot;
    if (const char* e = getenv("PFA_INGEST_TRANSPORT")) g->pack_rows = strcmp(e, "raw") != 0;
    if (const char* e = getenv("PFA_INGEST_SLOTS")) { const long v = strtol(e, nullptr, 10); if (v >= 2 && v <= 64) { g->n_slots = (uint32_t)v; g->n_dev_slots = 2 * g->n_slots; } }
    if (const char* e = getenv("PFA_INGEST_SLOT_BYTES")) { const long long v = strtoll(e, nullptr, 10); if (v > 0) g->max_slot = std::max<size_t>(pfa::up256((size_t)v), 4096); }
    g->device = pfa::ctx().device;
    g->slots.assign(g->n_slots, pfa::Pinned());
    g->slot_user.assign(g->n_slots, -1);
    g->dev_user.assign(g->n_dev_slots, -1);
  } catch (const std::bad_alloc&) {
    delete g;
    pfa::set_error("pfa_ingest_open: out of host memory");
    return PFA_ERR_NOMEM;
  }
  if (const char* e = getenv("PFA_INGEST_INFLIGHT")) { const long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 16) g->max_inflight = (uint32_t)v; }
  if (const char* e = getenv("PFA_INGEST_STREAMS")) { const long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 4) g->n_streams = (uint32_t)v; }
  *out = g;      // (the copy streams are made by the uploader thread when the first copy is due: a small job never needs one)
  return PFA_OK;
}

int pfa_ingest_open(const char* const* paths, uint32_t n_files, const pfa_ingest_opts* opts, pfa_ingest** out) {
  PFA_NEED_INIT();
  if (!out || (n_files && !paths)) { pfa::set_error("pfa_ingest_open: NULL argument"); return PFA_ERR_ARG; }
  *out = nullptr;
  try {
    std::vector<std::string> v;
    v.reserve(n_files);
    for (uint32_t i = 0; i < n_files; ++i) {
      if (!paths[i]) { pfa::set_error("pfa_ingest_open: path %u is NULL", i); return PFA_ERR_ARG; }
      v.emplace_back(paths[i]);
    }
    return ingest_open_impl(std::move(v), opts, out);
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_ingest_open: out of host memory");
    return PFA_ERR_NOMEM;
  }
}

// well-formed UTF-8 (no overlong forms, no surrogates, <= U+10FFFF): for such names byte order is code-point order
static bool valid_utf8(const char* s) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
  while (*p) {
    if (*p < 0x80) { ++p; continue; }
    uint32_t n, cp, lo;
    if ((*p & 0xE0) == 0xC0) { n = 1; cp = *p & 0x1Fu; lo = 0x80; }
    else if ((*p & 0xF0) == 0xE0) { n = 2; cp = *p & 0x0Fu; lo = 0x800; }
    else if ((*p & 0xF8) == 0xF0) { n = 3; cp = *p & 0x07u; lo = 0x10000; }
    else return false;
    ++p;
    for (uint32_t k = 0; k < n; ++k, ++p) {
      if ((*p & 0xC0) != 0x80) return false;
      cp = cp << 6 | (*p & 0x3Fu);
    }
    if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
  }
  return true;
}

int pfa_ingest_open_dir(const char* dir, const pfa_ingest_opts* opts, pfa_ingest** out, uint32_t* n_files) {
  PFA_NEED_INIT();
  if (!dir || !out || !n_files) { pfa::set_error("pfa_ingest_open_dir: NULL argument"); return PFA_ERR_ARG; }
  *out = nullptr;
  *n_files = 0;
  try {
    DIR* dp = opendir(dir);
    if (!dp) { pfa::set_error("pfa_ingest_open_dir: cannot open %s", dir); return PFA_ERR_IO; }
    std::vector<std::string> names;
    bool utf8 = valid_utf8(dir);
    while (dirent* e = readdir(dp)) {
      const char* n = e->d_name;
      if (n[0] == '.' && (n[1] == 0 || (n[1] == '.' && n[2] == 0))) continue;     // os.listdir leaves "." and ".." out, nothing else
      utf8 = utf8 && valid_utf8(n);
      names.emplace_back(n);
    }
    closedir(dp);
    if (!utf8) {   // Python would sort the escaped bytes by code point (U+DC80..): leave such directories to the caller
      pfa::set_error("pfa_ingest_open_dir: a name in %s is not UTF-8", dir);
      return PFA_ERR_UNSUPPORTED;
    }
    if (names.size() > 0xFFFFFFF0ull) { pfa::set_error("pfa_ingest_open_dir: too many entries"); return PFA_ERR_ARG; }
    std::sort(names.begin(), names.end());                   // sorted(dir + "/" + name): the common prefix drops out (PolyFastA.py:94, :104)
    const std::string prefix = std::string(dir) + "/";
    for (std::string& n : names) n.insert(0, prefix);
    const uint32_t n = (uint32_t)names.size();
    const int rc = ingest_open_impl(std::move(names), opts, out);
    if (rc == PFA_OK) *n_files = n;
    return rc;
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_ingest_open_dir: out of host memory");
    return PFA_ERR_NOMEM;
  }
}

int pfa_ingest_open_names(const char* dir, const char* names_blob, const uint64_t* names_off, uint32_t n_files,
                          const pfa_ingest_opts* opts, pfa_ingest** out) {
  PFA_NEED_INIT();
  if (!dir || !out || (n_files && (!names_blob || !names_off))) { pfa::set_error("pfa_ingest_open_names: NULL argument"); return PFA_ERR_ARG; }
  *out = nullptr;
  try {
    const std::string prefix = std::string(dir) + "/";
    std::vector<std::string> v;
    v.reserve(n_files);
    for (uint32_t i = 0; i < n_files; ++i) {
      if (names_off[i + 1] < names_off[i]) { pfa::set_error("pfa_ingest_open_names: offsets of name %u run backwards", i); return PFA_ERR_ARG; }
      v.emplace_back(prefix);
      v.back().append(names_blob + names_off[i], (size_t)(names_off[i + 1] - names_off[i]));
    }
    return ingest_open_impl(std::move(v), opts, out);
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_ingest_open_names: out of host memory");
    return PFA_ERR_NOMEM;
  }
}

const char* pfa_ingest_path(const pfa_ingest* g, uint32_t i) { return g && i < g->paths.size() ? g->paths[i].c_str() : nullptr; }

int pfa_ingest_feed(pfa_ingest* g, const uint32_t* file_idx, uint32_t n, int last) {
  PFA_NEED_INIT();
  if (!g) { pfa::set_error("pfa_ingest_feed: ingest is NULL"); return PFA_ERR_ARG; }
  try {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->feed_closed) { pfa::set_error("pfa_ingest_feed: the feed was closed by an earlier call"); return PFA_ERR_ARG; }
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t f = file_idx ? file_idx[i] : i;
      if (f >= g->paths.size()) { pfa::set_error("pfa_ingest_feed: file index %u >= %zu", f, g->paths.size()); return PFA_ERR_ARG; }
      g->feed.push_back(f);
      g->files.emplace_back();
    }
    if (last) g->feed_closed = true;
    if (!g->threads_started && (!g->feed.empty() || g->feed_closed)) {
      g->threads_started = true;
      // Direct mode: a job that is small — a handful of files, or (round 4) up to 4 096 files holding at most kDirectMaxText bytes
      // (stat() of each, PFA_INGEST_DIRECT_MAX overrides) — and fed in one piece pays neither the copy stream's queue (20 ms) nor the
      // copy engine's set-up (9 ms) nor the device text ring: repack_kernel reads the pinned buffers over PCIe (a fresh process,
      // 8-64 files: files -> rows 21-30 ms with the copy pipeline).  Nothing is recycled in this mode: the pinned "ring" grows by
      // one buffer per use (new_use), bounded by the text the job holds.
      auto small_job = [&]() -> bool {
        if (g->feed.size() <= std::min<size_t>(4, g->n_slots)) return true;
        unsigned long long cap = pfa::kDirectMaxText;
        if (const char* e = getenv("PFA_INGEST_DIRECT_MAX")) cap = strtoull(e, nullptr, 10);
        if (cap == 0 || g->feed.size() > 4096) return false;
        unsigned long long total = 0;
        for (const uint32_t f : g->feed) {
          struct stat sb;
          if (stat(g->paths[f].c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) total += (unsigned long long)sb.st_size;
          if (total > cap) return false;
        }
        return true;
      };
      g->direct = g->feed_closed && !getenv("PFA_INGEST_NO_DIRECT") && small_job();
      const uint32_t nt = std::max(1u, std::min<uint32_t>(g->want_threads, (uint32_t)std::max<size_t>(g->feed.size(), 1)));
      g->want_threads = nt;
      for (uint32_t t = 0; t < nt; ++t) g->workers.emplace_back(pfa::worker, g, t);
      g->uploader = std::thread(pfa::uploader, g);
    }
    g->cv_work.notify_all();
    g->cv_up.notify_all();
    g->cv_batch.notify_all();
  } catch (...) {
    pfa::set_error("pfa_ingest_feed: cannot start the loader threads");
    return PFA_ERR_NOMEM;
  }
  return PFA_OK;
}

uint32_t pfa_ingest_backlog(const pfa_ingest* g) {
  if (!g) return 0;
  pfa_ingest* m = const_cast<pfa_ingest*>(g);
  std::lock_guard<std::mutex> lk(m->mu);
  return (uint32_t)g->feed.size() - g->n_handed;
}

static void release_batch(pfa_ingest* g) {
  if (g->plan) { pfa_plan_destroy(g->plan); g->plan = nullptr; }   // synchronises the stream: the last repack is done
  else if (g->launched && pfa::ctx().ready) (void)hipStreamSynchronize(pfa::ctx().stream);
  g->launched = false;
  for (uint8_t* d : g->big_dev) (void)hipFree(d);
  g->big_dev.clear();
  std::vector<pfa::Pinned> done;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    done.swap(g->big_pinned_done);
  }
  for (pfa::Pinned& p : done) pfa::give_pinned(p);
}

int pfa_ingest_next(pfa_ingest* g, pfa_ingest_batch* out) {
  PFA_NEED_INIT();
  if (!g || !out) { pfa::set_error("pfa_ingest_next: NULL argument"); return PFA_ERR_ARG; }
  memset(out, 0, sizeof(*out));
  release_batch(g);
  const double t_in = pfa::now_s();
  uint32_t lo, hi;
  struct View { uint8_t* d; bool big; hipEvent_t ev; };
  std::vector<View> uv;
  std::vector<uint32_t> bpos;   // the batch's feed positions
  {
    std::unique_lock<std::mutex> lk(g->mu);
    for (;;) {
      if (g->hard_rc) { pfa::set_error("%s", g->hard_error.c_str()); return g->hard_rc; }
      if (g->consumed < g->uploaded) break;
      if (g->feed_closed && pfa::all_reserved(g) && g->consumed == g->uses.size()) return 0;
      if (!g->threads_started) { pfa::set_error("pfa_ingest_next: nothing has been fed"); return PFA_ERR_ARG; }
      if (!g->feed_closed && pfa::all_reserved(g) && g->consumed == g->uses.size()) {
        pfa::set_error("pfa_ingest_next: every fed file has been handed out and the feed is still open");
        return PFA_ERR_ARG;
      }
      g->cv_batch.wait(lk);
    }
    lo = g->consumed;
    hi = lo + 1;
    if (g->direct) {
      hi = g->uploaded;
    } else if (!g->uses[lo].big) {
      size_t bytes = g->uses[lo].used;
      while (hi < g->uploaded && !g->uses[hi].big && bytes + g->uses[hi].used <= pfa::kMaxBatchBytes) { bytes += g->uses[hi].used; ++hi; }
    }
    for (uint32_t u = lo; u < hi; ++u) {
      uv.push_back(View{g->uses[u].d, g->uses[u].big, g->uses[u].ev});
      bpos.insert(bpos.end(), g->uses[u].pos.begin(), g->uses[u].pos.end());
    }
  }
  if (getenv("PFA_INGEST_DEBUG")) {
    std::lock_guard<std::mutex> lk(g->mu);
    for (uint32_t u = lo; u < hi; ++u) {
      const pfa::Use& s = g->uses[u];
      fprintf(stderr, "ingest batch %u: use %u big=%d host_slot=%d dev_slot=%d d=+%zu used=%zu cap=%zu files:", g->n_batches, u, (int)s.big, s.host_slot, s.dev_slot,
              s.big ? (size_t)0 : (size_t)(s.d - g->d_text), s.used, s.cap);
      for (uint32_t p : s.pos) fprintf(stderr, " %u", p);
      fprintf(stderr, "\n");
    }
  }
  try {
    std::sort(bpos.begin(), bpos.end());     // within a batch: feed order
    const uint32_t nf = (uint32_t)bpos.size(), nk = (uint32_t)g->keys.size();
    const uint32_t p0 = nf ? bpos.front() : 0u;
    g->b_file_idx.resize(nf); g->b_status.resize(nf); g->b_n_rows.resize(nf); g->b_n_sites.resize(nf);
    g->b_key_rows.assign((size_t)nf * nk, 0u);
    g->b_unit_file.clear(); g->b_unit_key.clear(); g->b_unit_rows.clear(); g->b_ridx.clear(); g->b_descs.clear();
    bool regular = true;
    uint64_t packed = 0, n_segs = 0;
    for (uint32_t b = 0; b < nf; ++b) {
      const pfa::FileRec& f = g->files[bpos[b]];
      g->b_file_idx[b] = g->feed[bpos[b]];
      g->b_status[b] = f.status;
      g->b_n_rows[b] = f.n_rows;
      g->b_n_sites[b] = f.n_sites;
      if (f.status != 0) { regular = false; continue; }
      if ((g->unit_flags & PFA_FLAG_CDS) && f.n_sites % 3u != 0u) regular = false;
      // rows the card will write: every row once — or, with --pops, once per key that names it
      n_segs += !f.n_sites ? 0u : nk == 0 ? f.n_rows : (uint64_t)f.key_rows.size();
    }
    // staging: the segments and the per-file `bad` words of this batch, in pinned memory
    const size_t seg_bytes = (size_t)n_segs * sizeof(pfa::Seg), bad_off = pfa::up256(seg_bytes), stage_need = bad_off + pfa::up256((size_t)nf * 4u + 4u);
    if (g->stage.cap < stage_need) {
      // (the previous batch of THIS ingest is done with it: release_batch() above has waited for its repack)
      if (!g->stage.h) {
        std::lock_guard<std::mutex> ck(pfa::g_cache.mu);
        if (pfa::g_cache.stage.cap >= stage_need) { g->stage = pfa::g_cache.stage; pfa::g_cache.stage = pfa::Pinned(); }
      }
      if (g->stage.cap < stage_need) {
        if (g->stage.h) (void)hipHostFree(g->stage.h);
        g->stage = pfa::Pinned();
        const size_t want = stage_need + stage_need / 2 + 65536;
        if (hipHostMalloc((void**)&g->stage.h, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pfa::set_error("pfa_ingest_next: cannot allocate the staging buffer"); return PFA_ERR_NOMEM; }
        g->stage.cap = want;
      }
    }
    pfa::Seg* const h_segs = reinterpret_cast<pfa::Seg*>(g->stage.h);
    g->bad_words = reinterpret_cast<uint32_t*>(g->stage.h + bad_off);
    memset(g->bad_words, 0, (size_t)nf * 4u + 4u);
    // (direct mode: every use is an allocation of its own and the segments carry absolute addresses)
    const uint8_t* const text_base = g->direct ? nullptr : uv.front().big ? uv.front().d : g->d_text;
    uint64_t si = 0;
    for (uint32_t b = 0; b < nf; ++b) {
      const pfa::FileRec& f = g->files[bpos[b]];
      if (f.status != 0) continue;
      const uint64_t stride = std::max<uint64_t>(16, ((uint64_t)f.n_sites + g->stride_align - 1) & ~uint64_t(g->stride_align - 1));
      if (stride > 0xFFFFFFF0ull) { pfa::set_error("pfa_ingest_next: %s: sequences too long", g->paths[g->b_file_idx[b]].c_str()); return PFA_ERR_ARG; }
      const uint64_t use_off = (uint64_t)(reinterpret_cast<uintptr_t>(uv[f.use - lo].d) - reinterpret_cast<uintptr_t>(text_base));
      auto seg_of = [&](uint32_t r, uint64_t dst) {
        return pfa::Seg{use_off + (f.row_off[r] & ~pfa::kRowPacked), dst, f.n_sites, b | ((f.row_off[r] & pfa::kRowPacked) ? pfa::kSegPacked : 0u)};
      };
      pfa_desc d{};
      d.n_sites = f.n_sites;
      d.row_stride = (uint32_t)stride;
      d.flags = g->unit_flags;
      d.row_idx_off = PFA_NO_ROW_IDX;
      if (nk == 0) {
        d.base_off = packed;
        d.n_rows = d.mat_rows = f.n_rows;
        if (f.n_sites)
          for (uint32_t r = 0; r < f.n_rows; ++r) h_segs[si++] = seg_of(r, packed + (uint64_t)r * stride);
        packed += pfa::up256((size_t)(std::max<uint64_t>(f.n_rows, 1) * stride));
        g->b_descs.push_back(d);
        g->b_unit_file.push_back(b); g->b_unit_key.push_back(PFA_NO_KEY); g->b_unit_rows.push_back(f.n_rows);
      } else {
        // The population split (PolyFastA.py:122-134) happens while the card writes the matrices: every key gets a matrix of its
        // own, holding the records whose name contains it in record order (a record named by two keys is written twice, one
        // named by none not at all).  The units then need no row-index lists: they run the lean kernel instantiations
        // (the row-list form of the same 100-row units reads 0.78 of the HBM peak, the contiguous one 0.81).
        size_t at = 0;
        for (uint32_t k = 0; k < nk; ++k) {
          const uint32_t c = f.key_cnt[k];
          g->b_key_rows[(size_t)b * nk + k] = c;
          if (c == 0) { regular = false; continue; }               // "# Pop X string was not found" (PolyFastA.py:126-131)
          d.base_off = packed;
          d.n_rows = d.mat_rows = c;
          if (f.n_sites)
            for (uint32_t i = 0; i < c; ++i) h_segs[si++] = seg_of(f.key_rows[at + i], packed + (uint64_t)i * stride);
          packed += pfa::up256((size_t)((uint64_t)c * stride));
          at += c;
          g->b_descs.push_back(d);
          g->b_unit_file.push_back(b); g->b_unit_key.push_back(k); g->b_unit_rows.push_back(c);
        }
      }
    }
    // sfs_off / tile_off: prefix sums in unit order (the plan's bitmap / SFS geometry)
    {
      uint64_t sfs = 0, tiles = 0;
      for (pfa_desc& d : g->b_descs) {
        if (sfs > 0xFFFFFFFFull || tiles > 0xFFFFFFFFull) { pfa::set_error("pfa_ingest_next: batch too large for 32-bit SFS / tile offsets"); return PFA_ERR_ARG; }
        d.sfs_off = (uint32_t)sfs;
        d.tile_off = (uint32_t)tiles;
        sfs += d.n_rows / 2u;
        tiles += (d.n_rows && d.n_sites) ? ((uint64_t)d.n_sites + PFA_TILE_SITES - 1u) / PFA_TILE_SITES : 0u;
      }
    }
    const uint32_t nu = (uint32_t)g->b_descs.size();
    hipStream_t st = pfa::ctx().stream;
    int rc = PFA_OK;
    if (nu) {
      rc = pfa_plan_create(g->b_descs.data(), nu, g->b_ridx.empty() ? nullptr : g->b_ridx.data(), g->b_ridx.size(), packed, 0u, &g->plan);
      if (rc != PFA_OK) return rc;
    }
    {
      std::lock_guard<std::mutex> ck(pfa::g_cache.mu);
      rc = pfa::ensure_dev(&pfa::g_cache.d_packed, &pfa::g_cache.d_packed_cap, (size_t)packed + 256u);
      if (rc == PFA_OK) rc = pfa::ensure_dev(&pfa::g_cache.d_segs, &pfa::g_cache.d_segs_cap, seg_bytes + 256u);
    }
    if (rc != PFA_OK) return rc;
    // everything below is enqueued behind the copies of this batch's slots
    // (copies of one stream complete in order; with several copy streams every slot's event is waited for)
    if (!g->direct)
      for (size_t k = g->n_streams > 1 ? 0 : uv.size() - 1; k < uv.size(); ++k) PFA_HIP(hipStreamWaitEvent(st, uv[k].ev, 0));
    g->launched = true;
    if (n_segs) {
      uint32_t* d_bad = nullptr;
      PFA_HIP(hipHostGetDevicePointer((void**)&d_bad, g->bad_words, 0));
      PFA_HIP(hipMemcpyAsync(pfa::g_cache.d_segs, h_segs, seg_bytes, hipMemcpyHostToDevice, st));
      rc = pfa::launch_repack(text_base, reinterpret_cast<const pfa::Seg*>(pfa::g_cache.d_segs), (uint32_t)std::min<uint64_t>(n_segs, 0xFFFFFFFFull),
                              pfa::g_cache.d_packed, g->stride_align, d_bad, st);
      if (rc != PFA_OK) return rc;
    }
    hipEvent_t& bev = g->batch_ev[g->n_batches % pfa::kBatchEvents];
    if (!bev) PFA_HIP(hipEventCreateWithFlags(&bev, hipEventDisableTiming));
    PFA_HIP(hipEventRecord(bev, st));
    if (g->plan) {
      rc = pfa_plan_run(g->plan, pfa::g_cache.d_packed, 0);
      if (rc != PFA_OK) return rc;
    }
    {
      std::lock_guard<std::mutex> lk(g->mu);
      for (uint32_t u = lo; u < hi; ++u) {
        g->uses[u].consumed = true;
        g->uses[u].repack_ev = bev;
        if (g->uses[u].big && !g->direct) g->big_dev.push_back(g->uses[u].d);
        // the rows of these files are on their way: their tables can go
        for (uint32_t p : g->uses[u].pos) {
          pfa::FileRec& f = g->files[p];
          std::vector<uint64_t>().swap(f.row_off);
          std::vector<uint32_t>().swap(f.key_rows);
        }
      }
      g->consumed = hi;
      g->n_handed += nf;
      g->cv_consumed.notify_all();
    }
    g->n_batches += 1;
    out->n_files = nf;
    out->n_units = nu;
    out->first_pos = p0;
    out->flags = regular ? PFA_BATCH_REGULAR : 0u;
    out->file_idx = g->b_file_idx.data();
    out->status = g->b_status.data();
    out->n_rows = g->b_n_rows.data();
    out->n_sites = g->b_n_sites.data();
    out->key_rows = g->b_key_rows.data();
    out->unit_file = g->b_unit_file.data();
    out->unit_key = g->b_unit_key.data();
    out->unit_rows = g->b_unit_rows.data();
    out->bad_bytes = g->bad_words;
    out->plan = g->plan;
  } catch (const std::bad_alloc&) {
    pfa::set_error("pfa_ingest_next: out of host memory");
    return PFA_ERR_NOMEM;
  }
  g->ns_main += (uint64_t)((pfa::now_s() - t_in) * 1e9);
  return 1;
}

int pfa_ingest_names(const pfa_ingest* g, const char** blob, const uint64_t** off) {
  if (!g || !blob || !off) { pfa::set_error("pfa_ingest_names: NULL argument"); return PFA_ERR_ARG; }
  *blob = g->names_blob.data();
  *off = g->names_off.data();
  return PFA_OK;
}

int pfa_ingest_stats(const pfa_ingest* g, double* out8) {
  if (!g || !out8) { pfa::set_error("pfa_ingest_stats: NULL argument"); return PFA_ERR_ARG; }
  out8[0] = (double)g->ns_read.load() * 1e-9;
  out8[1] = (double)g->ns_index.load() * 1e-9;
  out8[2] = (double)g->bytes_read.load();
  out8[3] = (double)g->ns_copy.load() * 1e-9;
  out8[4] = (double)g->bytes_copied.load();
  out8[5] = (double)g->want_threads;
  out8[6] = (double)g->ns_wait.load() * 1e-9;
  out8[7] = (double)g->n_batches;
  return PFA_OK;
}

void pfa_ingest_close(pfa_ingest* g) {
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
    g->cv_work.notify_all(); g->cv_slot.notify_all(); g->cv_open.notify_all();
    g->cv_up.notify_all(); g->cv_batch.notify_all(); g->cv_consumed.notify_all();
  }
  for (std::thread& t : g->workers) if (t.joinable()) t.join();
  if (g->uploader.joinable()) g->uploader.join();
  const bool live = pfa::ctx().ready;
  if (live) {
    for (uint32_t k = 0; k < g->n_streams; ++k) if (g->copy_streams[k]) (void)hipStreamSynchronize(g->copy_streams[k]);
    release_batch(g);
    (void)hipStreamSynchronize(pfa::ctx().stream);
  }
  for (pfa::Use& u : g->uses) {
    if (u.ev) (void)hipEventDestroy(u.ev);
    if (u.big) {   // oversized files that never reached a batch, or whose copy was never waited for
      if (!u.consumed && !g->direct) (void)hipFree(u.d);
      if (u.h) { g->big_pinned_done.push_back(pfa::Pinned{u.h, u.cap}); u.h = nullptr; }
    }
  }
  for (hipEvent_t e : g->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : g->batch_ev) if (e) (void)hipEventDestroy(e);
  for (pfa::Pinned& p : g->big_pinned_done) pfa::give_pinned(p);
  for (pfa::Pinned& p : g->slots) pfa::give_pinned(p);
  if (g->stage.h) {
    std::lock_guard<std::mutex> ck(pfa::g_cache.mu);
    if (live && pfa::g_cache.stage.cap < g->stage.cap) {
      if (pfa::g_cache.stage.h) (void)hipHostFree(pfa::g_cache.stage.h);
      pfa::g_cache.stage = g->stage;
    } else {
      (void)hipHostFree(g->stage.h);
    }
    g->stage = pfa::Pinned();
  }
  if (g->d_text) {
    std::lock_guard<std::mutex> ck(pfa::g_cache.mu);
    if (live && pfa::g_cache.d_text_cap < g->d_text_cap) {
      if (pfa::g_cache.d_text) (void)hipFree(pfa::g_cache.d_text);
      pfa::g_cache.d_text = g->d_text;
      pfa::g_cache.d_text_cap = g->d_text_cap;
    } else {
      (void)hipFree(g->d_text);
    }
  }
  for (uint32_t k = 0; k < g->n_streams; ++k) if (g->copy_streams[k]) (void)hipStreamDestroy(g->copy_streams[k]);
  delete g;
}

}  // extern "C"
