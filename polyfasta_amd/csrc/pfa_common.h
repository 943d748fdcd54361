// pfa_common.h — context, error plumbing and small helpers shared by the
// translation units of libpfa_hip.so.
#pragma once
#ifdef PFA_INGEST_HOST_TEST
#include "hip_mock.h"   // tests/: the HIP runtime as threads and memcpy, for the CPU sanitizer build of the ingest
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>

#include "../../include/pfa.h"

namespace pfa {

struct Context {
  bool ready = false;
  int device = -1;
  int n_cu = 0;
  size_t lds_per_block = 0;
  hipStream_t stream = nullptr;
  double cpu_share = 1e9;    // CPUs this process may keep busy (pfa_host.h: cpu_share)
  bool spin_waits = true;    // host waits spin (enough CPUs) or sleep
};

Context& ctx();
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void ingest_release_cache();   // pfa_ingest.hip: pinned slots and device buffers kept between ingests

}  // namespace pfa

#define PFA_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      pfa::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                     __LINE__);                                                        \
      return PFA_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define PFA_NEED_INIT()                                              \
  do {                                                               \
    if (!pfa::ctx().ready) {                                         \
      pfa::set_error("pfa_init() has not been called successfully"); \
      return PFA_ERR_NOT_INIT;                                       \
    }                                                                \
  } while (0)
