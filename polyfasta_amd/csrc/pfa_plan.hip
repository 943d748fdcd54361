// pfa_plan.hip — context, device memory, scan plans and the remaining entry points of
// include/pfa.h (the kernels live in pfa_scan_classic.hip / pfa_scan_tall.hip).
//
// A plan validates a descriptor table, sorts its units into BUCKETS — units that share a kernel
// instantiation and an LDS size — and launches one kernel per bucket: a few many-row alignments
// in a batch of 50-row ones run the row-chunked kernel on their own, the rest keeps the fast
// shape and its occupancy.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "pfa_common.h"
#include "pfa_host.h"
#include "pfa_launch.h"
#include "pfa_layout.h"

namespace pfa {

Context& ctx() {
  static Context c;
  return c;
}

static thread_local char g_err[768] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static constexpr ct::Tables kHostTables = ct::make_tables();

// Result slab -> pinned host memory, written by the compute units, and the slab cleared for the
// next run in the same pass.  (A hipMemcpyAsync here goes through the SDMA engines, whose queues
// the runtime creates lazily: single copies were seen to take ~8 ms, longer than a whole batch scan.)
// `flag` (single-workgroup copies of small slabs only): a word in pinned memory that receives `ticket`
// once every byte of the copy is visible to the host — the host polls it instead of paying a stream
// synchronisation (2-3 us of an 18 us call).
__device__ inline void signal_host(volatile uint32_t* flag, uint32_t ticket) {
  if (flag == nullptr) return;      // uniform
  __threadfence_system();           // this thread's stores to host memory are out
  __syncthreads();                  // ... and everybody else's
  if (threadIdx.x == 0) {
    *flag = ticket;
    __threadfence_system();
  }
}

__global__ void __launch_bounds__(256) slab_to_host_kernel(uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                           volatile uint32_t* flag, uint32_t ticket) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    dst[i] = src[i];
    src[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  signal_host(flag, ticket);
}

// The same for a plan with output replicas (ScanArgs::rep_mask): the replicas are summed on the way
// — 64-bit words of the records section that hold H, H_S, H_N (words 0-2 of the 7 of a record) as
// 64-bit integers, everything else as pairs of 32-bit counters — and cleared.  dst == src sums in place
// (pfa_plan_results_dev): neither pointer is `__restrict__`, so the store of the sum stays behind the
// clearing store to the same word.
__global__ void __launch_bounds__(256) sum_replicas_kernel(unsigned long long* src, unsigned long long* dst, size_t n64,
                                                           uint32_t replicas, size_t stride64, size_t n_rec_words,
                                                           volatile uint32_t* flag, uint32_t ticket) {
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n64; w += (size_t)gridDim.x * blockDim.x) {
    const bool wide = w < n_rec_words && (w % 7u) < 3u;
    unsigned long long acc = 0ull;
    for (uint32_t r0 = 0; r0 < replicas; r0 += 8u) {   // eight independent loads in flight, then their zeros
      unsigned long long v[8];
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) v[k] = r0 + k < replicas ? src[w + (r0 + k) * stride64] : 0ull;
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        if (r0 + k < replicas) src[w + (r0 + k) * stride64] = 0ull;
        acc = wide ? acc + v[k]
                   : (unsigned long long)((uint32_t)acc + (uint32_t)v[k]) | (unsigned long long)((uint32_t)(acc >> 32) + (uint32_t)(v[k] >> 32)) << 32;
      }
    }
    dst[w] = acc;
  }
  signal_host(flag, ticket);
}

// The distinct clean codons of every codon column of ONE unit (pfa_plan_codon_sets; PolyFastA.py:291, :301): lane = codon
// column, so the 64 lanes of a wavefront read 192 consecutive bytes of a row; blockIdx.y = a block of `rows_per_block` rows
// (units with many rows: the blocks OR their sets together).  A re-scan of the rare unit whose synonymous-site count lands on
// a rounding tie — a few units in a thousand — not a bandwidth kernel.
__global__ void __launch_bounds__(64) codon_sets_kernel(const uint8_t* __restrict__ bases, const pfa_desc* __restrict__ descs,
                                                        const uint32_t* __restrict__ row_idx, uint32_t unit, uint32_t rows_per_block,
                                                        unsigned long long* __restrict__ sets) {
  const pfa_desc D = descs[unit];
  const uint32_t col = blockIdx.x * 64u + threadIdx.x;
  if ((uint64_t)col * 3u + 2u >= D.n_sites) return;     // beyond the last whole codon (a partial one holds no clean codon, :301)
  const uint32_t r0 = blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < D.n_rows ? r0 + rows_per_block : D.n_rows;
  const uint8_t* const mat = bases + D.base_off + (uint64_t)col * 3u;
  const uint32_t* const rows = D.row_idx_off != PFA_NO_ROW_IDX ? row_idx + D.row_idx_off : nullptr;
  unsigned long long m = 0ull;
#pragma unroll 4
  for (uint32_t r = r0; r < r1; ++r) {
    const uint8_t* const p = mat + (uint64_t)(rows ? rows[r] : r) * D.row_stride;
    const uint32_t x0 = p[0], x1 = p[1], x2 = p[2];
    if (base_clean(x0) && base_clean(x1) && base_clean(x2)) m |= 1ull << (base_code(x0) * 16u + base_code(x1) * 4u + base_code(x2));
  }
  if (gridDim.y == 1u) sets[col] = m;
  else if (m) atomicOr(&sets[col], m);
}

static constexpr uint32_t kLdsPerCu = 160u * 1024u;
static constexpr uint32_t kTallSlots = 64u;     // its count-table slots per tile (measured: 64 6.03, 96 5.68 TB/s at N = 100)
static constexpr int kEventRing = 64;

static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* e = getenv(name);
  return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : dflt;
}

// Device / pinned buffers of destroyed plans are kept for the next plan: the CLI builds one plan
// per ingest batch, and hipMalloc + hipHostMalloc cost more than a small batch's scan.
struct BufPool {
  struct Entry { void* p; size_t bytes; };
  std::vector<Entry> dev, host;
  static constexpr size_t kMaxEntries = 24, kMaxBytes = size_t(1) << 30;

  static size_t size_class(size_t need) {
    size_t c = 4096;
    while (c < need) c += c / 2 < 4096 ? 4096 : c / 2;   // steps of 1.5x
    return c;
  }
  hipError_t take(bool pinned, size_t need, void** out, size_t* got) {
    std::vector<Entry>& v = pinned ? host : dev;
    const size_t want = size_class(need);
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i].bytes >= need && v[i].bytes <= 4 * want && (best == v.size() || v[i].bytes < v[best].bytes)) best = i;
    if (best != v.size()) {
      *out = v[best].p;
      *got = v[best].bytes;
      v.erase(v.begin() + best);
      return hipSuccess;
    }
    *got = want;
    return pinned ? hipHostMalloc(out, want, hipHostMallocDefault) : hipMalloc(out, want);
  }
  void give(bool pinned, void* p, size_t bytes) {
    if (!p) return;
    std::vector<Entry>& v = pinned ? host : dev;
    size_t total = bytes;
    for (const Entry& e : v) total += e.bytes;
    if (v.size() >= kMaxEntries || total > kMaxBytes) {
      if (pinned) (void)hipHostFree(p); else (void)hipFree(p);
      return;
    }
    v.push_back({p, bytes});
  }
  void clear() {
    for (const Entry& e : dev) (void)hipFree(e.p);
    for (const Entry& e : host) (void)hipHostFree(e.p);
    dev.clear();
    host.clear();
  }
};
static BufPool g_pool;

// One kernel launch of a plan.
struct Bucket {
  LaunchCfg cfg;
  bool tall = false;
  uint32_t n_max = 0;             // most rows of any of its units
  uint32_t max_tiles = 0;
  uint64_t sum_tiles = 0;
  uint64_t bytes = 0;             // bases its units read
  uint32_t tiles_per_chunk = 1;
  uint32_t rows_per_chunk = 0, max_slots = 0, hist_bins = 0;
  uint32_t first = 0;             // offset of its units in d_unit_ids
  std::vector<uint32_t> units;
};

// classic kernel: LDS-resident tiles keep the CU full when at least 6 wavefronts fit whose tiles
// add up to 125 KB of loads in flight (measured round 1: N = 100 x 256 sites, 6 waves, 6.45 TB/s
// against 6.03 row-chunked; N = 128, 4 waves, 5.84 against 5.88; N = 160, 3 waves, 5.04 against 6.06;
// round 3, 128-site tiles, classic / row-chunked (tools/rows_sweep.sh): 6 waves — N = 192 0.771 / 0.765, 200 0.766 / 0.762,
// coding 0.736 / 0.664 and 0.720 / 0.709; 5 waves — N = 208 0.691 / 0.726, 224 0.719 / 0.726, 250 0.612 / 0.715, coding 208
// 0.618 / 0.664, 224 0.652 / 0.659)
// Single-wavefront workgroups a CU holds by their LDS: the 160 KB are handed out in 512-byte granules (round 5, wave trace: the
// headline's 13 424 B fit 11 times, not the 12 that 160 KB / 13 424 and hipOccupancyMaxActiveBlocksPerMultiprocessor say).
static uint32_t waves_by_lds(uint32_t lds_bytes) {
  return kLdsPerCu / std::max((lds_bytes + 511u) & ~511u, 512u);
}

static bool classic_fills_cu(uint32_t lds_bytes, uint32_t n, uint32_t tw) {
  const uint32_t waves = waves_by_lds(lds_bytes);
  return waves >= 6u && (uint64_t)waves * n * tw >= 125000ull;
}

}  // namespace pfa

struct pfa_plan {
  uint32_t n_desc = 0;
  uint32_t n_tiles_total = 0;     // 256-site tiles (bitmap geometry)
  uint32_t want = 0;
  bool any_cds = false;
  bool timing = false;
  uint64_t sfs_words = 0;
  uint64_t n_bases_bytes = 0;
  std::vector<pfa::Bucket> buckets;   // largest first
  std::vector<pfa_desc> h_descs;      // the caller's table (pfa_plan_codon_sets sizes its re-scan of one unit by it)
  const uint8_t* last_bases = nullptr; // the matrices of the last run
  pfa_desc* d_descs = nullptr;
  uint32_t* d_row_idx = nullptr;
  uint32_t* d_unit_ids = nullptr;
  // results | sfs | syn_num live in ONE device slab mirrored by one pinned host slab.  The slab is
  // clear before every run: cleared at creation and by every fetch (copy-and-clear kernel), by a
  // memset only when two runs follow each other without a fetch.
  uint8_t* d_slab = nullptr;
  uint8_t* h_slab = nullptr;
  uint8_t* h_slab_dev = nullptr;   // the device's address of h_slab
  bool slab_clean = false;         // the device slab holds zeros
  bool host_valid = false;         // h_slab holds the outputs of the last run
  size_t slab_bytes = 0, off_sfs = 0, off_syn = 0;   // one replica
  uint32_t replicas = 1;           // copies of the slab that the chunks of a unit spread their atomics over
  bool summed = true;              // replica 0 holds the sum, the others zeros
  uint32_t fetch_ticket = 0;       // copies signalled through the flag word behind h_slab
  pfa_result* d_results = nullptr;
  uint32_t* d_sfs = nullptr;
  uint32_t* d_syn = nullptr;
  unsigned long long* d_var_bm = nullptr;
  unsigned long long* d_syn_bm = nullptr;
  unsigned long long* d_non_bm = nullptr;
  hipEvent_t ev0[pfa::kEventRing] = {};
  hipEvent_t ev1[pfa::kEventRing] = {};
  uint64_t n_runs = 0, n_timed = 0;
  std::string description;
  // capacities of the pooled buffers (pfa::BufPool)
  size_t cap_descs = 0, cap_row_idx = 0, cap_unit_ids = 0, cap_slab = 0, cap_hslab = 0, cap_var = 0, cap_syn = 0, cap_non = 0;
};

extern "C" {

int pfa_abi_version(void) { return PFA_ABI_VERSION; }

// "PFA_BUILD_ID=<16 hex digits>": the translation unit polyfasta_amd/_build.py generates from the hash of the sources (the
// marker lets the host side read the id out of the file without loading it)
extern "C" const char pfa_build_id_string[];
const char* pfa_build_id(void) { return pfa_build_id_string + 13; }

const char* pfa_last_error(void) { return pfa::g_err; }

int pfa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int init_impl(int device) {
  pfa::Context& c = pfa::ctx();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    pfa::set_error("no HIP device visible: libpfa_hip has no CPU fallback");
    return PFA_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    pfa::set_error("device %d out of range (0..%d)", device, n - 1);
    return PFA_ERR_ARG;
  }
  if (c.ready && c.device == device) { PFA_HIP(hipSetDevice(device)); return PFA_OK; }   // (the device is a per-thread setting)
  if (c.ready) pfa_shutdown();
  PFA_HIP(hipSetDevice(device));
  // host waits spin instead of sleeping on an interrupt: a blocked wait was seen to return
  // up to 8 ms late on the GPU box, longer than a whole batch scan (hint only; ignored if the
  // runtime already fixed its flags) — unless this process is short of CPUs (several ranks inside one
  // control group's CPU quota: every spinning wait is quota the loader threads do not get, and a group that
  // uses its quota up is frozen until the 100 ms period ends): then waits sleep.  PFA_WAIT=spin|block overrides.
  c.cpu_share = pfa::cpu_share();
  c.spin_waits = c.cpu_share >= 12.0;
  if (const char* e = getenv("PFA_WAIT")) c.spin_waits = strcmp(e, "block") != 0;
  (void)hipSetDeviceFlags(c.spin_waits ? hipDeviceScheduleSpin : hipDeviceScheduleBlockingSync);
  (void)hipGetLastError();
  hipDeviceProp_t prop;
  PFA_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    pfa::set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return PFA_ERR_NO_DEVICE;
  }
  c.n_cu = prop.multiProcessorCount;
  c.lds_per_block = prop.sharedMemPerBlock;
  PFA_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  c.device = device;
  c.ready = true;
  return PFA_OK;
}

// The whole start-up of a one-file run on a thread of the library's own, started by the command line before it imports
// its modules (tools/ubench/init_probe.hip: hipInit 50-230 ms, a stream = an HSA queue 20 ms, the first H2D copy 9 ms of SDMA
// set-up, a translation unit's code object on its first launch): context, stream, the code objects of the scan / plan / ingest
// kernels and the copy engine are up by the time Python has imported numpy.  pfa_init() waits for it.
// (The thread object lives on the heap: a joinable std::thread in static storage would call std::terminate from the
// library's static destructors when the interpreter exits before pfa_init() — an uncaught exception ahead of the first scan.)
static std::thread* g_prewarm = nullptr;
static std::mutex g_prewarm_mu;

static void join_prewarm() {
  std::thread* t = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_prewarm_mu);
    t = g_prewarm;
    g_prewarm = nullptr;
  }
  if (t) {
    if (t->joinable()) t->join();
    delete t;
  }
}

void pfa_prewarm(int device) {
  std::lock_guard<std::mutex> lk(g_prewarm_mu);
  if (pfa::ctx().ready || g_prewarm) return;
  try {
    g_prewarm = new std::thread([device]() {
      if (init_impl(device) != PFA_OK) return;       // (pfa_init() repeats it on the caller's thread and reports the error there)
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(pfa::slab_to_host_kernel));
      pfa::warm_classic();
      pfa::warm_ingest();
      void *h = nullptr, *d = nullptr;
      if (hipHostMalloc(&h, 4096, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, 4096) == hipSuccess) {
        memset(h, 0, 4096);
        (void)hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, pfa::ctx().stream);
        (void)hipStreamSynchronize(pfa::ctx().stream);
      }
      if (d) (void)hipFree(d);
      if (h) (void)hipHostFree(h);
      (void)hipGetLastError();
    });
  } catch (...) {
    g_prewarm = nullptr;
  }
}

void pfa_prewarm_join(void) { join_prewarm(); }

int pfa_init(int device) {
  join_prewarm();
  return init_impl(device);
}

void pfa_shutdown(void) {
  join_prewarm();
  pfa::Context& c = pfa::ctx();
  if (!c.ready) return;
  (void)hipStreamSynchronize(c.stream);
  pfa::g_pool.clear();
  pfa::ingest_release_cache();
  (void)hipStreamDestroy(c.stream);
  c = pfa::Context();
}

int pfa_device_index(void) { return pfa::ctx().ready ? pfa::ctx().device : -1; }

int pfa_sync(void) {
  PFA_NEED_INIT();
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

int pfa_dev_alloc(size_t bytes, void** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_dev_alloc: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  PFA_HIP(hipMalloc(out, bytes ? bytes : 16));
  return PFA_OK;
}

int pfa_dev_free(void* p) {
  PFA_NEED_INIT();
  if (p) PFA_HIP(hipFree(p));
  return PFA_OK;
}

int pfa_host_alloc(size_t bytes, void** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_host_alloc: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  PFA_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return PFA_OK;
}

int pfa_host_free(void* p) {
  PFA_NEED_INIT();
  if (p) PFA_HIP(hipHostFree(p));
  return PFA_OK;
}

int pfa_upload(void* dst_dev, const void* src_host, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pfa::ctx().stream));
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

int pfa_upload_async(void* dst_dev, const void* src_host, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pfa::ctx().stream));
  return PFA_OK;
}

int pfa_download(void* dst_host, const void* src_dev, size_t bytes) {
  PFA_NEED_INIT();
  if (!bytes) return PFA_OK;
  PFA_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, pfa::ctx().stream));
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  return PFA_OK;
}

// bitmaps (and tile_off) are laid out per 256-site tile whatever tile width a kernel works on
uint32_t pfa_tile_sites(uint32_t /*max_rows*/, uint32_t /*cds*/) { return PFA_TILE_SITES; }
uint32_t pfa_var_words_per_tile(uint32_t /*max_rows*/, uint32_t /*cds*/) { return 4u; }

void pfa_plan_destroy(pfa_plan* p) {
  if (!p) return;
  if (pfa::ctx().ready) {
    (void)hipStreamSynchronize(pfa::ctx().stream);
    pfa::BufPool& pool = pfa::g_pool;
    pool.give(false, p->d_descs, p->cap_descs);
    pool.give(false, p->d_row_idx, p->cap_row_idx);
    pool.give(false, p->d_unit_ids, p->cap_unit_ids);
    pool.give(false, p->d_slab, p->cap_slab);
    pool.give(true, p->h_slab, p->cap_hslab);
    pool.give(false, p->d_var_bm, p->cap_var);
    pool.give(false, p->d_syn_bm, p->cap_syn);
    pool.give(false, p->d_non_bm, p->cap_non);
  } else {   // the context is gone (pfa_shutdown() before the plan): nothing to pool, just release
    (void)hipFree(p->d_descs);
    (void)hipFree(p->d_row_idx);
    (void)hipFree(p->d_unit_ids);
    (void)hipFree(p->d_slab);
    if (p->h_slab) (void)hipHostFree(p->h_slab);
    (void)hipFree(p->d_var_bm);
    (void)hipFree(p->d_syn_bm);
    (void)hipFree(p->d_non_bm);
  }
  for (int i = 0; i < pfa::kEventRing; ++i) {
    if (p->ev0[i]) (void)hipEventDestroy(p->ev0[i]);
    if (p->ev1[i]) (void)hipEventDestroy(p->ev1[i]);
  }
  delete p;
}

}  // extern "C"

namespace pfa {

// Which kernel serves a unit: the bucket key.
//   coding units, and every unit of a plan that wants bitmaps: 256-site tiles; LDS-resident rows
//     while they fill the CU, else the row-chunked kernel (wide above 2 048 coding rows)
//   other units: 256-site tiles up to 64 rows, 128-site tiles while those fill the CU (about 250
//     rows), else the row-chunked kernel; tiny plans take 128-site tiles whatever their rows: more wavefronts
struct BucketKey {
  uint32_t tall, cds, mapped, wide, tw, waves;   // waves: occupancy class
  bool operator<(const BucketKey& o) const {
    return std::tie(tall, cds, mapped, wide, tw, waves) < std::tie(o.tall, o.cds, o.mapped, o.wide, o.tw, o.waves);
  }
};

static BucketKey bucket_of(const pfa_desc& d, bool bitmaps, uint32_t narrow_tw) {
  BucketKey k{};
  const uint32_t n = d.n_rows;
  k.cds = (d.flags & PFA_FLAG_CDS) ? 1u : 0u;
  k.mapped = d.row_idx_off != PFA_NO_ROW_IDX ? 1u : 0u;
  k.tw = 256u;
  const bool rowmap = k.mapped || bitmaps;   // the instantiation with bitmaps also carries the row map
  if (k.cds || bitmaps) {
    // coding units above 64 rows: 128-site tiles while those fill the CU (bitmaps are laid out per 256 sites)
    const uint32_t forced = env_u32("PFA_FORCE_TW", 0u);
    if (!bitmaps && (forced == 128u || (!forced && n > env_u32("PFA_TW256_MAX_ROWS", 64u)))) k.tw = 128u;
    uint32_t lds = lds_layout(k.tw, n, k.cds, rowmap).total;
    if (k.tw == 128u && n > 64u && !classic_fills_cu(lds, n, 128u) && !forced) {
      k.tw = 256u;
      lds = lds_layout(k.tw, n, k.cds, rowmap).total;
    }
    k.tall = n > 64u && !classic_fills_cu(lds, n, k.tw);
    if (getenv("PFA_TALL_MIN")) k.tall = n > env_u32("PFA_TALL_MIN", 64u) || lds > kLdsPerCu;
    if (k.tall) k.tw = 256u;
  } else {
    const uint32_t tw256_max = env_u32("PFA_TW256_MAX_ROWS", 64u);
    const uint32_t forced = env_u32("PFA_FORCE_TW", 0u);
    uint32_t tw = forced ? forced : narrow_tw ? narrow_tw : (n <= tw256_max ? 256u : 128u);
    if (tw != 256u && tw != 128u && tw != 64u) tw = 256u;
    uint32_t lds = lds_layout(tw, n, false, rowmap).total;
    if (n > 64u && !classic_fills_cu(lds, n, tw) && tw == 256u) {   // too many rows for 256-site tiles
      tw = 128u;
      lds = lds_layout(tw, n, false, rowmap).total;
    }
    k.tw = tw;
    k.tall = n > 64u && !classic_fills_cu(lds, n, tw);
    if (getenv("PFA_TALL_MIN")) k.tall = n > env_u32("PFA_TALL_MIN", 64u) || lds > kLdsPerCu;
    if (k.tall) k.tw = 256u;
  }
  if (k.tall) {
    k.wide = k.cds && n > 2048u;
    k.waves = 0u;
  } else {
    const uint32_t lds = lds_layout(k.tw, n, k.cds, rowmap).total;
    // units of one bucket share an LDS size: group them by the wavefronts per CU their own size
    // allows, so that nobody loses occupancy to a larger neighbour; 16-bit per-site H up to 64 rows
    k.waves = std::min(waves_by_lds(lds), 32u) | (k.cds && n <= 64u ? 0x100u : 0u);
  }
  return k;
}

static uint32_t tiles_of(const pfa_desc& d, uint32_t tw) {
  return (d.n_rows && d.n_sites) ? (uint32_t)(((uint64_t)d.n_sites + tw - 1u) / tw) : 0u;
}

static void finish_bucket(Bucket& b, const BucketKey& k, bool bitmaps, const pfa_desc* descs) {
  b.tall = k.tall != 0u;
  b.cfg.cds = k.cds != 0u;
  b.cfg.mapped = k.mapped != 0u;
  b.cfg.bitmaps = bitmaps;
  b.cfg.wide = k.wide != 0u;
  b.cfg.tw = k.tw;
  b.cfg.n_units = (uint32_t)b.units.size();
  for (uint32_t u : b.units) {
    const pfa_desc& d = descs[u];
    const uint32_t nt = tiles_of(d, k.tw);
    b.n_max = std::max(b.n_max, d.n_rows);
    b.max_tiles = std::max(b.max_tiles, nt);
    b.sum_tiles += nt;
    b.bytes += (uint64_t)d.n_rows * d.n_sites;
  }
  if (b.tall) {
    // chunks of kTallRowsFast rows and a shorter last one (a multiple of 4: one LDS-DMA instruction moves 4 rows);
    // PFA_TALL_ROWS (timing experiments, tests of the general path): equal chunks of at most that many rows
    if (getenv("PFA_TALL_ROWS")) {
      const uint32_t rcap = std::min(64u, std::max(8u, env_u32("PFA_TALL_ROWS", kTallRowsFast)));
      const uint32_t nc = (b.n_max + rcap - 1u) / rcap;
      b.rows_per_chunk = std::min(64u, std::max(8u, ((b.n_max + nc - 1u) / nc + 3u) & ~3u));
    } else {
      b.rows_per_chunk = std::min(kTallRowsFast, std::max(8u, (b.n_max + 3u) & ~3u));
    }
    b.max_slots = std::min(256u, std::max(16u, env_u32("PFA_TALL_SLOTS", kTallSlots)));
    b.hist_bins = std::min(b.n_max / 2u, 256u);
    b.cfg.lds_bytes = tall_layout(b.rows_per_chunk, b.max_slots, b.hist_bins, b.cfg.cds, b.cfg.wide).total;
  } else {
    b.cfg.lds_bytes = lds_layout(k.tw, b.n_max, b.cfg.cds, b.cfg.mapped || bitmaps).total;
  }
  // chunking: enough single-wave workgroups to fill 256 CUs several times over, but whole
  // alignments per wave once the batch is large (fewer atomics).  96 single-wave workgroups per
  // CU measured best for the bench workload (10 tiles per chunk: 6.48 TB/s; 20 tiles 6.36,
  // 4 tiles 6.36, 1 tile 5.43)
  const uint64_t target_blocks = (uint64_t)std::max(ctx().n_cu, 1) * 96ull;
  uint64_t tpc = std::max<uint64_t>(1, b.sum_tiles / target_blocks);
  if (getenv("PFA_TILES_PER_CHUNK")) tpc = std::max(1u, env_u32("PFA_TILES_PER_CHUNK", 10u));   // timing experiments
  tpc = std::min<uint64_t>(tpc, std::max<uint32_t>(b.max_tiles, 1));
  tpc = std::max<uint64_t>(tpc, ((uint64_t)b.max_tiles + 65534ull) / 65535ull);
  uint32_t n_chunks = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)b.max_tiles + tpc - 1) / tpc));
  // balance: the same number of chunks, equal shares (40 tiles in 4 chunks = 10 each, not 13+13+13+1)
  b.tiles_per_chunk = std::max<uint32_t>(1u, (b.max_tiles + n_chunks - 1) / n_chunks);
  if (b.cfg.cds) {  // a coding chunk must start on a codon boundary: 3 tiles = 768 sites
    b.tiles_per_chunk = (b.tiles_per_chunk + 2u) / 3u * 3u;
    n_chunks = std::max<uint32_t>(1u, (b.max_tiles + b.tiles_per_chunk - 1) / b.tiles_per_chunk);
  }
  b.cfg.n_chunks = n_chunks;
}

static std::string describe(const pfa_plan* p) {
  std::string s;
  char buf[256];
  for (const Bucket& b : p->buckets) {
    if (!s.empty()) s += " | ";
    if (b.tall)
      snprintf(buf, sizeof buf, "pfa::scan_tall_kernel<cds=%d,mapped=%d,bitmaps=%d,wide=%d> units=%u n_max=%u rows_per_chunk=%u lds=%u tiles_per_chunk=%u",
               (int)b.cfg.cds, (int)(b.cfg.mapped || b.cfg.bitmaps), (int)(b.cfg.mapped || b.cfg.bitmaps), (int)b.cfg.wide, b.cfg.n_units, b.n_max, b.rows_per_chunk,
               b.cfg.lds_bytes, b.tiles_per_chunk);
    else
      snprintf(buf, sizeof buf, "pfa::scan_kernel<cds=%d,mapped=%d,bitmaps=%d,tw=%u> units=%u n_max=%u lds=%u tiles_per_chunk=%u",
               (int)b.cfg.cds, (int)(b.cfg.mapped || b.cfg.bitmaps), (int)b.cfg.bitmaps, b.cfg.tw, b.cfg.n_units, b.n_max, b.cfg.lds_bytes, b.tiles_per_chunk);
    s += buf;
  }
  if (p->replicas > 1u) {
    snprintf(buf, sizeof buf, " | %u output replicas", p->replicas);
    s += buf;
  }
  return s;
}

}  // namespace pfa

extern "C" {

int pfa_plan_create(const pfa_desc* descs, uint32_t n_desc, const uint32_t* row_idx,
                    size_t n_row_idx, uint64_t n_bases_bytes, uint32_t want, pfa_plan** out) {
  PFA_NEED_INIT();
  if (!out) { pfa::set_error("pfa_plan_create: out is NULL"); return PFA_ERR_ARG; }
  *out = nullptr;
  if (n_desc && !descs) { pfa::set_error("pfa_plan_create: descs is NULL"); return PFA_ERR_ARG; }
  if (n_desc == 0) { pfa::set_error("pfa_plan_create: empty descriptor table"); return PFA_ERR_ARG; }
  if (want & ~(PFA_WANT_VAR_BITMAP | PFA_WANT_SN_BITMAP)) { pfa::set_error("pfa_plan_create: unknown `want` bits 0x%x", want); return PFA_ERR_ARG; }

  bool any_cds = false;
  uint64_t sfs_words = 0, tiles_total = 0, tiles256_live = 0;
  for (uint32_t i = 0; i < n_desc; ++i) {
    const pfa_desc& d = descs[i];
    if (d.flags & ~PFA_FLAG_CDS) {
      pfa::set_error("desc %u: unknown flag bits 0x%x", i, d.flags);
      return PFA_ERR_ARG;
    }
    if (d.n_rows > PFA_MAX_ROWS) {
      pfa::set_error("desc %u: %u rows; the limit is %u", i, d.n_rows, PFA_MAX_ROWS);
      return PFA_ERR_TOO_MANY_ROWS;
    }
    any_cds = any_cds || (d.flags & PFA_FLAG_CDS);
    const uint32_t nt = pfa::tiles_of(d, PFA_TILE_SITES);
    sfs_words = std::max<uint64_t>(sfs_words, (uint64_t)d.sfs_off + d.n_rows / 2u);
    tiles_total = std::max<uint64_t>(tiles_total, (uint64_t)d.tile_off + nt);
    tiles256_live += nt;
    if (nt == 0) continue;
    if (d.row_stride % 16u || d.base_off % 16u) {
      pfa::set_error("desc %u: base_off and row_stride must be multiples of 16", i);
      return PFA_ERR_ARG;
    }
    if ((uint64_t)d.row_stride < (((uint64_t)d.n_sites + 15u) & ~15ull)) {
      pfa::set_error("desc %u: row_stride %u < round_up(n_sites=%u, 16)", i, d.row_stride, d.n_sites);
      return PFA_ERR_ARG;
    }
    if (d.mat_rows == 0 || (uint64_t)d.base_off + (uint64_t)d.mat_rows * d.row_stride > n_bases_bytes) {
      pfa::set_error("desc %u: matrix [%llu, +%u x %u) exceeds the %llu-byte base buffer", i,
                     (unsigned long long)d.base_off, d.mat_rows, d.row_stride,
                     (unsigned long long)n_bases_bytes);
      return PFA_ERR_ARG;
    }
    if (d.row_idx_off == PFA_NO_ROW_IDX) {
      if (d.n_rows > d.mat_rows) {
        pfa::set_error("desc %u: n_rows %u > mat_rows %u", i, d.n_rows, d.mat_rows);
        return PFA_ERR_ARG;
      }
    } else {
      if (!row_idx || (uint64_t)d.row_idx_off + d.n_rows > n_row_idx) {
        pfa::set_error("desc %u: row_idx range out of bounds", i);
        return PFA_ERR_ARG;
      }
      if ((uint64_t)d.mat_rows * d.row_stride > (1ull << 36)) {
        pfa::set_error("desc %u: a matrix scanned through a row-index list must be smaller than 64 GB", i);
        return PFA_ERR_ARG;
      }
      for (uint32_t r = 0; r < d.n_rows; ++r)
        if (row_idx[d.row_idx_off + r] >= d.mat_rows) {
          pfa::set_error("desc %u: row index %u >= mat_rows %u", i, row_idx[d.row_idx_off + r], d.mat_rows);
          return PFA_ERR_ARG;
        }
    }
  }
  if (tiles_total > 0xFFFFFFFFull || sfs_words > 0xFFFFFFFFull) {
    pfa::set_error("plan too large (tiles or SFS words overflow 32 bits)");
    return PFA_ERR_ARG;
  }

  pfa_plan* p = nullptr;
  try {
    p = new pfa_plan();
    p->n_desc = n_desc;
    p->any_cds = any_cds;
    p->want = want;
    p->sfs_words = sfs_words;
    p->n_tiles_total = (uint32_t)tiles_total;
    p->n_bases_bytes = n_bases_bytes;
    p->h_descs.assign(descs, descs + n_desc);
    p->timing = getenv("PFA_PLAN_TIMING") != nullptr;

    // ---- buckets.  A plan too small to give every CU a few wavefronts takes narrower tiles
    // (more, smaller wavefront jobs: latency, not bandwidth, decides there).
    const bool bitmaps = want != 0u;
    const uint64_t n_cu = (uint64_t)std::max(pfa::ctx().n_cu, 1);
    // (64-site tiles never beat 128-site ones — one alignment per launch 7.6 against 7.7 us, config 3 as one
    // launch 13.7 against 10.8 — so they stay a PFA_FORCE_TW measurement variant)
    const uint32_t narrow_tw = tiles256_live < n_cu * 4ull ? 128u : 0u;
    std::map<pfa::BucketKey, pfa::Bucket> bmap;
    for (uint32_t i = 0; i < n_desc; ++i) {
      if (pfa::tiles_of(descs[i], PFA_TILE_SITES) == 0u) continue;   // empty units: their records stay zero
      bmap[pfa::bucket_of(descs[i], bitmaps, narrow_tw)].units.push_back(i);
    }
    for (auto& kv : bmap) {
      pfa::finish_bucket(kv.second, kv.first, bitmaps, descs);
      p->buckets.push_back(std::move(kv.second));
    }
    std::stable_sort(p->buckets.begin(), p->buckets.end(),
                     [](const pfa::Bucket& a, const pfa::Bucket& b) { return a.bytes > b.bytes; });
    // few units, many chunks per unit: spread each unit's atomics over replicas of the slab (about
    // 24 wavefronts per replica and unit)
    uint32_t max_chunks = 1;
    for (const pfa::Bucket& b : p->buckets) max_chunks = std::max(max_chunks, b.cfg.n_chunks);
    while (p->replicas < 32u && max_chunks / p->replicas > 24u && (uint64_t)n_desc * p->replicas * 2u <= 4096u) p->replicas *= 2u;
    if (getenv("PFA_REPLICAS")) p->replicas = std::min(64u, std::max(1u, pfa::env_u32("PFA_REPLICAS", 1u)));   // timing experiments (a power of two)
    if (p->replicas & (p->replicas - 1u)) p->replicas = 1u;
    p->description = pfa::describe(p);
  } catch (const std::bad_alloc&) {
    delete p;
    pfa::set_error("out of host memory");
    return PFA_ERR_NOMEM;
  }

#define PLAN_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      pfa::set_error("%s failed: %s", #call, hipGetErrorString(e__));                         \
      pfa_plan_destroy(p);                                                                    \
      return PFA_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

  hipStream_t st = pfa::ctx().stream;
  PLAN_HIP(pfa::g_pool.take(false, (size_t)n_desc * sizeof(pfa_desc), (void**)&p->d_descs, &p->cap_descs));
  PLAN_HIP(hipMemcpyAsync(p->d_descs, descs, (size_t)n_desc * sizeof(pfa_desc), hipMemcpyHostToDevice, st));
  if (n_row_idx) {
    PLAN_HIP(pfa::g_pool.take(false, n_row_idx * sizeof(uint32_t), (void**)&p->d_row_idx, &p->cap_row_idx));
    PLAN_HIP(hipMemcpyAsync(p->d_row_idx, row_idx, n_row_idx * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  }
  // the units of every bucket, back to back; a single bucket holding every unit needs no list
  std::vector<uint32_t> ids;
  if (!(p->buckets.size() == 1 && p->buckets[0].units.size() == n_desc)) {
    for (pfa::Bucket& b : p->buckets) {
      b.first = (uint32_t)ids.size();
      ids.insert(ids.end(), b.units.begin(), b.units.end());
    }
    if (!ids.empty()) {
      PLAN_HIP(pfa::g_pool.take(false, ids.size() * sizeof(uint32_t), (void**)&p->d_unit_ids, &p->cap_unit_ids));
      PLAN_HIP(hipMemcpyAsync(p->d_unit_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
  }
  {
    auto up = [](size_t x) { return (x + 255u) & ~size_t(255); };
    p->off_sfs = up((size_t)n_desc * sizeof(pfa_result));
    p->off_syn = p->off_sfs + up(std::max<size_t>(16, sfs_words * sizeof(uint32_t)));
    p->slab_bytes = p->off_syn + (any_cds ? up((size_t)n_desc * PFA_SYN_BINS * sizeof(uint32_t)) : 0);
    PLAN_HIP(pfa::g_pool.take(false, p->slab_bytes * p->replicas, (void**)&p->d_slab, &p->cap_slab));
    PLAN_HIP(hipMemsetAsync(p->d_slab, 0, p->slab_bytes * p->replicas, st));
    p->slab_clean = true;
    PLAN_HIP(pfa::g_pool.take(true, p->slab_bytes + 64u, (void**)&p->h_slab, &p->cap_hslab));   // + the flag word of small fetches
    *reinterpret_cast<volatile uint32_t*>(p->h_slab + p->slab_bytes) = 0u;
    PLAN_HIP(hipHostGetDevicePointer((void**)&p->h_slab_dev, p->h_slab, 0));
    p->d_results = reinterpret_cast<pfa_result*>(p->d_slab);
    p->d_sfs = reinterpret_cast<uint32_t*>(p->d_slab + p->off_sfs);
    p->d_syn = any_cds ? reinterpret_cast<uint32_t*>(p->d_slab + p->off_syn) : nullptr;
  }
  const size_t nt = std::max<size_t>(1, p->n_tiles_total);
  if (want & PFA_WANT_VAR_BITMAP) {
    PLAN_HIP(pfa::g_pool.take(false, nt * 4u * sizeof(uint64_t), (void**)&p->d_var_bm, &p->cap_var));
    PLAN_HIP(hipMemsetAsync(p->d_var_bm, 0, nt * 4u * sizeof(uint64_t), st));   // tiles of empty units
  }
  if ((want & PFA_WANT_SN_BITMAP) && any_cds) {
    PLAN_HIP(pfa::g_pool.take(false, nt * 4u * sizeof(uint64_t), (void**)&p->d_syn_bm, &p->cap_syn));
    PLAN_HIP(pfa::g_pool.take(false, nt * 4u * sizeof(uint64_t), (void**)&p->d_non_bm, &p->cap_non));
  }
  PLAN_HIP(hipStreamSynchronize(st));   // `ids` and the caller's tables may go (the timing events are created on first use)
#undef PLAN_HIP
  *out = p;
  return PFA_OK;
}

int pfa_plan_set_timing(pfa_plan* p, int on) {
  if (!p) { pfa::set_error("pfa_plan_set_timing: plan is NULL"); return PFA_ERR_ARG; }
  p->timing = on != 0;
  return PFA_OK;
}

int pfa_plan_run(pfa_plan* p, const uint8_t* d_bases, int variant) {
  PFA_NEED_INIT();
  if (!p || !d_bases) { pfa::set_error("pfa_plan_run: NULL argument"); return PFA_ERR_ARG; }
  if (variant != 0 && variant != 2) { pfa::set_error("pfa_plan_run: variant %d (0 = default, 2 = default cache policy)", variant); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  if (!p->slab_clean) PFA_HIP(hipMemsetAsync(p->d_slab, 0, p->slab_bytes * p->replicas, st));
  p->slab_clean = false;
  p->host_valid = false;
  p->summed = p->replicas == 1u;
  p->last_bases = d_bases;
  if (p->d_syn_bm) {  // S / N bitmaps are built with atomicOr: clear them every run
    const size_t nb = std::max<size_t>(1, p->n_tiles_total) * 4u * sizeof(uint64_t);
    PFA_HIP(hipMemsetAsync(p->d_syn_bm, 0, nb, st));
    PFA_HIP(hipMemsetAsync(p->d_non_bm, 0, nb, st));
  }
  int slot = -1;
  if (p->timing) {
    slot = (int)(p->n_timed % pfa::kEventRing);
    if (!p->ev0[slot]) PFA_HIP(hipEventCreate(&p->ev0[slot]));
    if (!p->ev1[slot]) PFA_HIP(hipEventCreate(&p->ev1[slot]));
    PFA_HIP(hipEventRecord(p->ev0[slot], st));
  }
  for (const pfa::Bucket& b : p->buckets) {
    pfa::ScanArgs a;
    a.bases = d_bases;
    a.descs = p->d_descs;
    a.unit_ids = p->d_unit_ids ? p->d_unit_ids + b.first : nullptr;
    a.row_idx = p->d_row_idx;
    a.results = p->d_results;
    a.sfs = p->d_sfs;
    a.syn_num = p->d_syn;
    a.var_bm = p->d_var_bm;
    a.syn_bm = p->d_syn_bm;
    a.non_bm = p->d_non_bm;
    a.tiles_per_chunk = b.tiles_per_chunk;
    a.n_max = b.n_max;
    a.rows_per_chunk = b.rows_per_chunk;
    a.max_slots = b.max_slots;
    a.hist_bins = b.hist_bins;
    a.rep_mask = p->replicas - 1u;
    a.rep_stride = (uint32_t)p->slab_bytes;
    pfa::LaunchCfg c = b.cfg;
    // the default-cache-policy variant exists for the 256-site lean / general instantiations (A/B)
    c.stage = (variant == 2 && c.tw == 256u && !c.wide) ? 0 : 2;
    const int rc = b.tall ? pfa::launch_tall(c, a, st) : pfa::launch_classic(c, a, st);
    if (rc != PFA_OK) return rc;
  }
  if (slot >= 0) {
    PFA_HIP(hipEventRecord(p->ev1[slot], st));
    p->n_timed += 1;
  }
  p->n_runs += 1;
  return PFA_OK;
}

int pfa_plan_kernel_ms(pfa_plan* p, float* ms_out, int n) {
  PFA_NEED_INIT();
  if (!p || !ms_out || n < 0) { pfa::set_error("pfa_plan_kernel_ms: bad argument"); return PFA_ERR_ARG; }
  PFA_HIP(hipStreamSynchronize(pfa::ctx().stream));
  const uint64_t have = std::min<uint64_t>(p->n_timed, pfa::kEventRing);
  const int cnt = (int)std::min<uint64_t>(have, (uint64_t)n);
  for (int i = 0; i < cnt; ++i) {
    const uint64_t run = p->n_timed - cnt + i;
    const int slot = (int)(run % pfa::kEventRing);
    PFA_HIP(hipEventElapsedTime(&ms_out[i], p->ev0[slot], p->ev1[slot]));
  }
  return cnt;
}

int pfa_plan_fetch(pfa_plan* p, pfa_result* results, uint32_t* sfs, uint32_t* syn_num,
                   uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap) {
  PFA_NEED_INIT();
  if (!p) { pfa::set_error("pfa_plan_fetch: plan is NULL"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  const size_t nt = p->n_tiles_total;
  if (syn_num && !p->d_syn) { pfa::set_error("pfa_plan_fetch: plan has no coding unit"); return PFA_ERR_ARG; }
  if (results || sfs || syn_num) {
    if (!p->host_valid) {
      // one device-side copy into pinned memory (which also clears the slab for the next run),
      // then plain memcpy into the caller's (pageable) arrays
      // small slabs (one workgroup copies them): the kernel signals through a flag word in pinned memory
      volatile uint32_t* const h_flag = reinterpret_cast<volatile uint32_t*>(p->h_slab + p->slab_bytes);
      volatile uint32_t* d_flag = nullptr;
      unsigned blocks;
      if (p->replicas == 1u) {
        const size_t n16 = p->slab_bytes / 16u;   // slab sections are padded to 256 bytes
        blocks = (unsigned)std::min<size_t>((n16 + 255u) / 256u, 1024u);
        if (blocks == 1u) d_flag = reinterpret_cast<volatile uint32_t*>(p->h_slab_dev + p->slab_bytes);
        hipLaunchKernelGGL(pfa::slab_to_host_kernel, dim3(blocks), dim3(256), 0, st,
                           reinterpret_cast<uint4*>(p->d_slab), reinterpret_cast<uint4*>(p->h_slab_dev), n16, d_flag, p->fetch_ticket + 1u);
      } else {
        const size_t n64 = p->slab_bytes / 8u;
        blocks = (unsigned)std::min<size_t>((n64 + 255u) / 256u, 1024u);
        if (blocks == 1u) d_flag = reinterpret_cast<volatile uint32_t*>(p->h_slab_dev + p->slab_bytes);
        hipLaunchKernelGGL(pfa::sum_replicas_kernel, dim3(blocks), dim3(256), 0, st,
                           reinterpret_cast<unsigned long long*>(p->d_slab), reinterpret_cast<unsigned long long*>(p->h_slab_dev), n64,
                           p->replicas, n64, (size_t)p->n_desc * 7u, d_flag, p->fetch_ticket + 1u);
      }
      PFA_HIP(hipGetLastError());
      bool seen = false;
      if (d_flag != nullptr && !pfa::ctx().spin_waits) {
        p->fetch_ticket += 1u;      // (short of CPUs: no polling, the stream wait below sleeps)
      } else if (d_flag != nullptr) {
        // poll the flag word (acquire: the slab reads below stay behind it) for at most 2 ms, then fall back
        p->fetch_ticket += 1u;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        for (uint32_t spin = 0; !seen; ++spin) {
          seen = __atomic_load_n(const_cast<const uint32_t*>(h_flag), __ATOMIC_ACQUIRE) == p->fetch_ticket;
          if (!seen && (spin & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) break;
        }
      }
      if (!seen) PFA_HIP(hipStreamSynchronize(st));
      p->host_valid = true;
      p->slab_clean = true;
      p->summed = true;
    }
    if (results) memcpy(results, p->h_slab, (size_t)p->n_desc * sizeof(pfa_result));
    if (sfs && p->sfs_words) memcpy(sfs, p->h_slab + p->off_sfs, p->sfs_words * sizeof(uint32_t));
    if (syn_num) memcpy(syn_num, p->h_slab + p->off_syn, (size_t)p->n_desc * PFA_SYN_BINS * sizeof(uint32_t));
  }
  bool copies = false;
  if (var_bitmap) {
    if (!p->d_var_bm) { pfa::set_error("pfa_plan_fetch: plan was created without PFA_WANT_VAR_BITMAP"); return PFA_ERR_ARG; }
    if (nt) { PFA_HIP(hipMemcpyAsync(var_bitmap, p->d_var_bm, nt * 4u * sizeof(uint64_t), hipMemcpyDeviceToHost, st)); copies = true; }
  }
  if (syn_bitmap || non_bitmap) {
    if (!p->d_syn_bm) { pfa::set_error("pfa_plan_fetch: plan has no S/N bitmaps"); return PFA_ERR_ARG; }
    if (syn_bitmap && nt) { PFA_HIP(hipMemcpyAsync(syn_bitmap, p->d_syn_bm, nt * 4u * sizeof(uint64_t), hipMemcpyDeviceToHost, st)); copies = true; }
    if (non_bitmap && nt) { PFA_HIP(hipMemcpyAsync(non_bitmap, p->d_non_bm, nt * 4u * sizeof(uint64_t), hipMemcpyDeviceToHost, st)); copies = true; }
  }
  if (copies) PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_plan_run_fetch(pfa_plan* p, const uint8_t* d_bases, int variant, pfa_result* results, uint32_t* sfs,
                       uint32_t* syn_num, uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap) {
  const int rc = pfa_plan_run(p, d_bases, variant);
  return rc != PFA_OK ? rc : pfa_plan_fetch(p, results, sfs, syn_num, var_bitmap, syn_bitmap, non_bitmap);
}

int pfa_plan_results_dev(pfa_plan* p, void** d_results, size_t* bytes) {
  PFA_NEED_INIT();
  if (!p || !d_results || !bytes) { pfa::set_error("pfa_plan_results_dev: NULL argument"); return PFA_ERR_ARG; }
  if (p->host_valid) { pfa::set_error("pfa_plan_results_dev: the records have been fetched (the device slab is cleared by a fetch)"); return PFA_ERR_ARG; }
  if (!p->summed) {   // plans with output replicas: sum them into the first, in place, behind the scans
    const size_t n64 = p->slab_bytes / 8u;
    const unsigned blocks = (unsigned)std::min<size_t>((n64 + 255u) / 256u, 1024u);
    hipLaunchKernelGGL(pfa::sum_replicas_kernel, dim3(blocks), dim3(256), 0, pfa::ctx().stream,
                       reinterpret_cast<unsigned long long*>(p->d_slab), reinterpret_cast<unsigned long long*>(p->d_slab), n64,
                       p->replicas, n64, (size_t)p->n_desc * 7u, (volatile uint32_t*)nullptr, 0u);
    PFA_HIP(hipGetLastError());
    p->summed = true;
  }
  *d_results = p->d_results;
  *bytes = (size_t)p->n_desc * sizeof(pfa_result);
  return PFA_OK;
}

int pfa_plan_codon_sets(pfa_plan* p, const uint8_t* d_bases, uint32_t unit, uint64_t* sets_out, uint64_t cap) {
  PFA_NEED_INIT();
  if (!p) { pfa::set_error("pfa_plan_codon_sets: plan is NULL"); return PFA_ERR_ARG; }
  if (unit >= p->n_desc) { pfa::set_error("pfa_plan_codon_sets: unit %u of %u", unit, p->n_desc); return PFA_ERR_ARG; }
  if (!d_bases) d_bases = p->last_bases;
  if (!d_bases) { pfa::set_error("pfa_plan_codon_sets: no matrices (the plan has not been run and d_bases is NULL)"); return PFA_ERR_ARG; }
  const pfa_desc& d = p->h_descs[unit];
  const uint64_t n_cols = ((uint64_t)d.n_sites + 2u) / 3u;
  if (n_cols == 0) return PFA_OK;
  if (!sets_out || cap < n_cols) { pfa::set_error("pfa_plan_codon_sets: %llu words needed, %llu given", (unsigned long long)n_cols, (unsigned long long)cap); return PFA_ERR_ARG; }
  memset(sets_out, 0, n_cols * sizeof(uint64_t));
  if (d.n_rows == 0 || d.n_sites < 3u) return PFA_OK;
  hipStream_t st = pfa::ctx().stream;
  unsigned long long* d_sets = nullptr;
  size_t cap_sets = 0;
  PFA_HIP(pfa::g_pool.take(false, n_cols * sizeof(uint64_t), (void**)&d_sets, &cap_sets));
  // rows shared out over blockIdx.y once a unit has more than a few thousand (the blocks OR into a cleared array)
  const uint32_t rows_per_block = 4096u;
  const uint32_t ny = (d.n_rows + rows_per_block - 1u) / rows_per_block;
  int rc = PFA_OK;
  hipError_t e = hipMemsetAsync(d_sets, 0, n_cols * sizeof(uint64_t), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pfa::codon_sets_kernel, dim3((unsigned)((d.n_sites / 3u + 63u) / 64u), ny, 1), dim3(64), 0, st,
                       d_bases, p->d_descs, p->d_row_idx, unit, rows_per_block, d_sets);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(sets_out, d_sets, n_cols * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    pfa::set_error("pfa_plan_codon_sets: %s", hipGetErrorString(e));
    rc = PFA_ERR_HIP;
    (void)hipStreamSynchronize(st);
  }
  pfa::g_pool.give(false, d_sets, cap_sets);
  return rc;
}

uint32_t pfa_plan_n_tiles(const pfa_plan* p) { return p ? p->n_tiles_total : 0u; }
uint64_t pfa_plan_sfs_words(const pfa_plan* p) { return p ? p->sfs_words : 0ull; }
// launch geometry of the plan's largest bucket (by bases read), for reports
uint32_t pfa_plan_tiles_per_chunk(const pfa_plan* p) { return p && !p->buckets.empty() ? p->buckets[0].tiles_per_chunk : 0u; }
uint32_t pfa_plan_lds_bytes(const pfa_plan* p) { return p && !p->buckets.empty() ? p->buckets[0].cfg.lds_bytes : 0u; }
uint32_t pfa_plan_rows_per_chunk(const pfa_plan* p) { return p && !p->buckets.empty() ? p->buckets[0].rows_per_chunk : 0u; }
uint32_t pfa_plan_tile_width(const pfa_plan* p) { return p && !p->buckets.empty() ? p->buckets[0].cfg.tw : 0u; }
uint32_t pfa_plan_n_launches(const pfa_plan* p) { return p ? (uint32_t)p->buckets.size() : 0u; }
const char* pfa_plan_describe(const pfa_plan* p) { return p ? p->description.c_str() : ""; }

int pfa_scan_batch(const uint8_t* bases, uint64_t n_bases_bytes, const pfa_desc* descs,
                   uint32_t n_desc, const uint32_t* row_idx, size_t n_row_idx,
                   pfa_result* results, uint32_t* sfs, uint32_t* syn_num, uint64_t* var_bitmap,
                   uint64_t* syn_bitmap, uint64_t* non_bitmap) {
  PFA_NEED_INIT();
  if (!bases && n_bases_bytes) { pfa::set_error("pfa_scan_batch: bases is NULL"); return PFA_ERR_ARG; }
  pfa_plan* plan = nullptr;
  const uint32_t want = (var_bitmap ? PFA_WANT_VAR_BITMAP : 0u) | ((syn_bitmap || non_bitmap) ? PFA_WANT_SN_BITMAP : 0u);
  int rc = pfa_plan_create(descs, n_desc, row_idx, n_row_idx, n_bases_bytes, want, &plan);
  if (rc != PFA_OK) return rc;
  void* d_bases = nullptr;
  rc = pfa_dev_alloc((size_t)n_bases_bytes, &d_bases);
  if (rc == PFA_OK) rc = pfa_upload(d_bases, bases, (size_t)n_bases_bytes);
  if (rc == PFA_OK) rc = pfa_plan_run(plan, (const uint8_t*)d_bases, 0);
  if (rc == PFA_OK)
    rc = pfa_plan_fetch(plan, results, sfs, plan->d_syn ? syn_num : nullptr, var_bitmap,
                        plan->d_syn_bm ? syn_bitmap : nullptr, plan->d_syn_bm ? non_bitmap : nullptr);
  if (d_bases) (void)hipFree(d_bases);
  pfa_plan_destroy(plan);
  return rc;
}

int64_t pfa_bitmap_positions(const uint64_t* words, uint32_t n_tiles, uint32_t words_per_tile,
                             uint32_t tile_sites, uint32_t n_sites, uint32_t* pos_out, uint64_t cap) {
  if (!words || (!pos_out && cap)) { pfa::set_error("pfa_bitmap_positions: NULL argument"); return PFA_ERR_ARG; }
  // bit l of word b of a tile <-> tile-local site  words_per_tile * l + b
  uint64_t n = 0;
  for (uint32_t t = 0; t < n_tiles; ++t) {
    const uint64_t* w = words + (uint64_t)t * words_per_tile;
    bool any = false;
    for (uint32_t b = 0; b < words_per_tile; ++b) any = any || w[b];
    if (!any) continue;
    for (uint32_t l = 0; l < 64u; ++l)
      for (uint32_t b = 0; b < words_per_tile; ++b)
        if ((w[b] >> l) & 1ull) {
          const uint64_t site = (uint64_t)t * tile_sites + (uint64_t)words_per_tile * l + b;
          if (site >= n_sites) continue;
          if (n >= cap) { pfa::set_error("pfa_bitmap_positions: output capacity %llu too small", (unsigned long long)cap); return PFA_ERR_ARG; }
          pos_out[n++] = (uint32_t)site;
        }
  }
  return (int64_t)n;
}

uint32_t pfa_classify_codon_mask(uint64_t mask) {
  return pfa::classify_sense_mask(mask & ~pfa::kHostTables.stop_mask, pfa::kHostTables, pfa::kHostTables.annot_info);
}

int pfa_classify_codon_masks_dev(const uint64_t* masks, uint32_t n, uint32_t* labels_out) {
  PFA_NEED_INIT();
  if (n == 0) return PFA_OK;
  if (!masks || !labels_out) { pfa::set_error("pfa_classify_codon_masks_dev: NULL argument"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  unsigned long long* d_m = nullptr;
  uint32_t* d_l = nullptr;
  PFA_HIP(hipMalloc((void**)&d_m, (size_t)n * 8u));
  if (hipMalloc((void**)&d_l, (size_t)n * 4u) != hipSuccess) { (void)hipFree(d_m); pfa::set_error("hipMalloc failed"); return PFA_ERR_HIP; }
  hipError_t e = hipMemcpyAsync(d_m, masks, (size_t)n * 8u, hipMemcpyHostToDevice, st);
  int rc = PFA_OK;
  if (e == hipSuccess) rc = pfa::launch_classify_masks(d_m, n, d_l, st);
  if (e == hipSuccess && rc == PFA_OK) e = hipMemcpyAsync(labels_out, d_l, (size_t)n * 4u, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_m);
  (void)hipFree(d_l);
  if (rc != PFA_OK) return rc;
  if (e != hipSuccess) { pfa::set_error("pfa_classify_codon_masks_dev: %s", hipGetErrorString(e)); return PFA_ERR_HIP; }
  return PFA_OK;
}

uint32_t pfa_codon_syn_thirds(uint32_t codon) { return codon < 64u ? pfa::kHostTables.syn3[codon] : 0u; }

}  // extern "C"
