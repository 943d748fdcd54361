// pfa_launch.h — what the plan code (pfa_plan.hip) knows of the kernel translation units:
// one launch = one bucket of a plan (units sharing a kernel instantiation and an LDS size).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfa {

struct ScanArgs;

struct LaunchCfg {
  uint32_t n_units = 0;      // grid.x
  uint32_t n_chunks = 1;     // grid.y: chunks of consecutive column tiles
  uint32_t lds_bytes = 0;    // dynamic LDS per wavefront
  uint32_t tw = 256;         // sites per tile: 256, or 128 / 64 (scan_kernel, non-coding, no bitmaps)
  int stage = 2;             // 2 = non-temporal tile loads (default), 0 = default cache policy (A/B)
  bool cds = false;          // the bucket holds coding units
  bool mapped = false;       // ... units with a row-index list
  bool bitmaps = false;      // the plan wants per-site bitmaps
  bool wide = false;         // tall coding units above 2 048 rows: 56-bit per-site H
};

int launch_classic(const LaunchCfg& c, const ScanArgs& a, hipStream_t st);
int launch_tall(const LaunchCfg& c, const ScanArgs& a, hipStream_t st);
// load the translation unit's code object now (pfa_prewarm): hipFuncGetAttributes on one of its kernels
void warm_classic();
void warm_ingest();
int launch_classify_masks(const unsigned long long* d_masks, uint32_t n, uint32_t* d_labels, hipStream_t st);

}  // namespace pfa
