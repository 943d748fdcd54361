// init_probe — where a fresh process's HIP start-up goes (bench.py `cold_start`: pfa_init is 0.09-0.15 s of a 0.2 s one-file run)
// build: hipcc --offload-arch=gfx950 -O2 -o init_probe init_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { if (p) *p = 1; }
int main(int argc, char**) {
  double t = now(), t0 = t;
  auto lap = [&](const char* what) { const double n = now(); printf("%-44s %8.2f ms\n", what, (n - t) * 1e3); t = n; };
  int n = 0;
  (void)hipInit(0); lap("hipInit");
  (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount");
  (void)hipSetDevice(0); lap("hipSetDevice");
  (void)hipFree(nullptr); lap("hipFree(0) (context)");
  (void)hipSetDeviceFlags(hipDeviceScheduleSpin); lap("hipSetDeviceFlags");
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
  int cu = 0; (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0); lap("hipDeviceGetAttribute");
  hipStream_t st = nullptr;
  if (argc > 1) {      // any argument: stay on the NULL stream (is its queue cheaper than a created stream's?)
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, nullptr, (int*)nullptr); (void)hipStreamSynchronize(nullptr); lap("first kernel launch on the NULL stream + sync");
  } else {
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); lap("hipStreamCreateWithFlags");
  }
  void* d = nullptr; (void)hipMalloc(&d, 1 << 20); lap("hipMalloc 1 MiB");
  void* h = nullptr; (void)hipHostMalloc(&h, 1 << 20, 0); lap("hipHostMalloc 1 MiB");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, (int*)d); (void)hipStreamSynchronize(st); lap("first kernel launch + sync (code object)");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, (int*)d); (void)hipStreamSynchronize(st); lap("second kernel launch + sync");
  (void)hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); lap("first H2D 1 MiB + sync");
  hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); lap("hipEventCreate");
  printf("%-44s %8.2f ms (%s, %d CUs)\n", "total", (now() - t0) * 1e3, prop.gcnArchName, cu);
  return 0;
}
