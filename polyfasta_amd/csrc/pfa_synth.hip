// pfa_synth.hip — deterministic synthetic alignments, device and host twins of
// the numpy specification in polyfasta_amd/synth.py.
// Bench-sized batches (BASELINE config 5 is 50 GB of bases) are generated
// straight into HBM; the host twin serves tests and the CPU-baseline sample.
// Integer arithmetic only, so all three implementations agree bit for bit.
#include <algorithm>
#include <thread>
#include <vector>

#include "pfa_common.h"

namespace pfa {

struct SiteParams {  // everything a site needs, decoded once and reused for all rows
  uint32_t anc;      // 0..3 index into "ACGT"
  uint32_t seg;      // 0/1
  uint32_t k;        // carriers
  uint32_t k2;       // carriers of the second derived allele (0 = none)
  uint32_t unit;     // rank(r) = (unit * r + shift) mod N
  uint32_t shift;
  uint32_t der1, der2;
};

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__host__ __device__ inline uint64_t alignment_key(uint64_t seed, uint64_t aln) {
  return splitmix64(seed ^ (aln * 0xD6E8FEB86659FD93ull));
}

// the 61 sense codons in A=0,C=1,G=2,T=3 lexicographic order (stops TAA, TAG, TGA removed)
__host__ __device__ inline void sense_codon(uint32_t idx, uint32_t out[3]) {
  // positions of the stops in the 64-codon order: TAA=48, TAG=50, TGA=56
  uint32_t c = idx;
  if (c >= 48u) c += 1u;   // skip 48
  if (c >= 50u) c += 1u;   // skip 50
  if (c >= 56u) c += 1u;   // skip 56
  out[0] = c >> 4;
  out[1] = (c >> 2) & 3u;
  out[2] = c & 3u;
}

__host__ __device__ inline SiteParams decode_site(const pfa_synth_params& P, uint64_t akey, uint64_t site,
                                                  const uint32_t* cdf, const uint32_t* units) {
  SiteParams s;
  const uint64_t h1 = splitmix64(akey + site);
  const uint64_t h2 = splitmix64(h1);
  if (P.coding) {
    const uint64_t hc = splitmix64((akey ^ 0xC0D0C0D0C0D0C0D1ull) + site / 3u);
    uint32_t cod[3];
    sense_codon((uint32_t)(hc % 61u), cod);
    s.anc = cod[site % 3u];
  } else {
    s.anc = (uint32_t)(h1 & 3u);
  }
  const uint32_t n = P.n_rows;
  s.seg = (n >= 2u && ((h1 >> 2) & 0xFFFFFFu) < P.seg_thresh) ? 1u : 0u;
  const uint32_t uk = (uint32_t)((h1 >> 26) & 0xFFFFFFu);
  const uint32_t d1 = (uint32_t)(((h1 >> 50) & 0xFFu) % 3u);
  s.unit = units[(uint32_t)((h2 & 0xFFFFu) % P.n_units)];
  s.shift = (uint32_t)(((h2 >> 16) & 0xFFFFFFu) % n);
  const bool tri = ((h2 >> 40) & 0x3FFu) < P.tri_thresh;
  uint32_t k = 0;
  if (s.seg) {
    // k = 1 + #{i : cdf[i] <= uk}   (numpy searchsorted side='right'), capped at N-1
    uint32_t lo = 0, hi = n - 1u;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cdf[mid] <= uk) lo = mid + 1u; else hi = mid;
    }
    k = 1u + lo;
    if (k > n - 1u) k = n - 1u;
  }
  s.k = k;
  s.k2 = (tri && k >= 2u) ? (k >> 1) : 0u;
  s.der1 = (s.anc + 1u + d1) & 3u;
  s.der2 = (s.anc + 1u + (d1 + 1u) % 3u) & 3u;
  return s;
}

__host__ __device__ inline uint8_t ascii_of(uint32_t code) { return (uint8_t)((0x54474341u >> (8u * code)) & 0xFFu); }

// one thread = 4 adjacent sites of one alignment, walked down all rows
__global__ void __launch_bounds__(256) synth_kernel(uint8_t* __restrict__ out, const pfa_synth_params P,
                                                    const uint32_t* __restrict__ cdf,
                                                    const uint32_t* __restrict__ units) {
  const uint32_t groups = (P.n_sites + 3u) / 4u;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (uint64_t)groups * P.n_aln) return;
  const uint32_t m = (uint32_t)(gid / groups);
  const uint32_t s0 = (uint32_t)(gid % groups) * 4u;
  const uint64_t akey = alignment_key(P.seed, P.first_aln + m);
  SiteParams sp[4];
  uint32_t rank[4];
  uint32_t ancw = 0, any_seg = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t site = s0 + j < P.n_sites ? s0 + j : P.n_sites - 1u;
    sp[j] = decode_site(P, akey, site, cdf, units);
    rank[j] = sp[j].shift;
    ancw |= (uint32_t)ascii_of(sp[j].anc) << (8 * j);
    any_seg |= sp[j].seg;
  }
  uint8_t* base = out + (uint64_t)m * P.n_rows * P.row_stride + s0;
  const bool whole = s0 + 4u <= P.n_sites;
  for (uint32_t r = 0; r < P.n_rows; ++r) {
    uint32_t w = ancw;
    if (any_seg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (sp[j].seg) {
          const uint32_t rk = rank[j];
          if (rk < sp[j].k) {
            const uint32_t code = rk < sp[j].k2 ? sp[j].der2 : sp[j].der1;
            w = (w & ~(0xFFu << (8 * j))) | ((uint32_t)ascii_of(code) << (8 * j));
          }
          uint32_t nx = rk + sp[j].unit;
          rank[j] = nx >= P.n_rows ? nx - P.n_rows : nx;
        }
      }
    }
    uint8_t* p = base + (uint64_t)r * P.row_stride;
    if (whole) *reinterpret_cast<uint32_t*>(p) = w;
    else
      for (uint32_t j = 0; s0 + j < P.n_sites; ++j) p[j] = (uint8_t)(w >> (8 * j));
  }
}

static int check_params(const pfa_synth_params* p, const uint32_t* cdf, const uint32_t* units) {
  if (!p || !units || (p->n_rows >= 2 && !cdf)) { set_error("pfa_synth: NULL argument"); return PFA_ERR_ARG; }
  if (p->n_rows == 0 || p->n_sites == 0 || p->n_units == 0 || p->row_stride < p->n_sites || p->row_stride % 4u) {
    set_error("pfa_synth: bad shape (rows %u, sites %u, stride %u, units %u)", p->n_rows, p->n_sites, p->row_stride, p->n_units);
    return PFA_ERR_ARG;
  }
  return PFA_OK;
}

}  // namespace pfa

extern "C" {

int pfa_synth_fill_dev(uint8_t* d_bases, const pfa_synth_params* p, const uint32_t* cdf, const uint32_t* units) {
  PFA_NEED_INIT();
  int rc = pfa::check_params(p, cdf, units);
  if (rc != PFA_OK) return rc;
  if (!d_bases) { pfa::set_error("pfa_synth_fill_dev: d_bases is NULL"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  uint32_t *d_cdf = nullptr, *d_units = nullptr;
  const size_t ncdf = p->n_rows >= 2 ? p->n_rows - 1 : 1;
  PFA_HIP(hipMalloc((void**)&d_cdf, ncdf * sizeof(uint32_t)));
  PFA_HIP(hipMalloc((void**)&d_units, p->n_units * sizeof(uint32_t)));
  if (p->n_rows >= 2) PFA_HIP(hipMemcpyAsync(d_cdf, cdf, ncdf * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  PFA_HIP(hipMemcpyAsync(d_units, units, p->n_units * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  const uint64_t groups = (uint64_t)((p->n_sites + 3u) / 4u) * p->n_aln;
  const uint64_t blocks = (groups + 255u) / 256u;
  if (blocks > 0x7FFFFFFFull) { pfa::set_error("pfa_synth_fill_dev: batch too large for one launch"); return PFA_ERR_ARG; }
  if (blocks) hipLaunchKernelGGL(pfa::synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_bases, *p, d_cdf, d_units);
  PFA_HIP(hipGetLastError());
  PFA_HIP(hipStreamSynchronize(st));
  PFA_HIP(hipFree(d_cdf));
  PFA_HIP(hipFree(d_units));
  return PFA_OK;
}

int pfa_synth_fill_host(uint8_t* bases, const pfa_synth_params* p, const uint32_t* cdf, const uint32_t* units) {
  int rc = pfa::check_params(p, cdf, units);
  if (rc != PFA_OK) return rc;
  if (!bases) { pfa::set_error("pfa_synth_fill_host: bases is NULL"); return PFA_ERR_ARG; }
  const pfa_synth_params P = *p;
  auto work = [&](uint32_t m0, uint32_t m1) {
    std::vector<pfa::SiteParams> sp(P.n_sites);
    std::vector<uint32_t> rank(P.n_sites);
    for (uint32_t m = m0; m < m1; ++m) {
      const uint64_t akey = pfa::alignment_key(P.seed, P.first_aln + m);
      for (uint32_t s = 0; s < P.n_sites; ++s) {
        sp[s] = pfa::decode_site(P, akey, s, cdf, units);
        rank[s] = sp[s].shift;
      }
      uint8_t* mat = bases + (uint64_t)m * P.n_rows * P.row_stride;
      for (uint32_t r = 0; r < P.n_rows; ++r) {
        uint8_t* row = mat + (uint64_t)r * P.row_stride;
        for (uint32_t s = 0; s < P.n_sites; ++s) {
          uint32_t code = sp[s].anc;
          if (sp[s].seg) {
            const uint32_t rk = rank[s];
            if (rk < sp[s].k) code = rk < sp[s].k2 ? sp[s].der2 : sp[s].der1;
            const uint32_t nx = rk + sp[s].unit;
            rank[s] = nx >= P.n_rows ? nx - P.n_rows : nx;
          }
          row[s] = pfa::ascii_of(code);
        }
      }
    }
  };
  const uint32_t hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const uint32_t nt = std::min<uint32_t>(hw, std::max<uint32_t>(1u, P.n_aln));
  if (nt <= 1) { work(0, P.n_aln); return PFA_OK; }
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < nt; ++t) {
    const uint32_t a = (uint32_t)((uint64_t)P.n_aln * t / nt), b = (uint32_t)((uint64_t)P.n_aln * (t + 1) / nt);
    th.emplace_back(work, a, b);
  }
  for (auto& t : th) t.join();
  return PFA_OK;
}

}  // extern "C"
