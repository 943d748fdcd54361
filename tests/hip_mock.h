// hip_mock.h — the slice of the HIP runtime that csrc/pfa_ingest.hip uses, as plain threads and memcpy.
// TEST INFRASTRUCTURE (tests/ingest_tsan.cpp): the ingest's host side — loader threads, the pinned ring, the
// uploader, batch assembly — is built with g++ -fsanitize=thread / address against this header and run on the CPU
// (GPU sanitizers are not available on the pool).  A stream is a thread that executes its operations in order and
// LATER than the call that enqueued them, like the real thing: a pinned slot handed to the loaders before its copy
// has run, or a copy started before its files have landed, is a data race the sanitizer reports.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

struct MockEvent {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, done = 0;   // record calls issued / executed by their stream
};
typedef MockEvent* hipEvent_t;

struct MockStream {
  std::mutex mu;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  std::thread th;
  MockStream() {
    th = std::thread([this]() {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        while (q.empty() && !stop) cv.wait(lk);
        if (q.empty()) return;
        std::function<void()> f = std::move(q.front());
        q.pop_front();
        busy = true;
        lk.unlock();
        f();
        lk.lock();
        busy = false;
        if (q.empty()) idle.notify_all();
      }
    });
  }
  ~MockStream() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    th.join();
  }
  void push(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
    cv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    while (!q.empty() || busy) idle.wait(lk);
  }
};
typedef MockStream* hipStream_t;

inline const char* hipGetErrorString(hipError_t) { return "mock HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new MockStream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { s->drain(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new MockEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
  s->push([=]() { memcpy(dst, src, n); });
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  uint64_t ticket;
  { std::lock_guard<std::mutex> lk(e->mu); ticket = ++e->recorded; }
  s->push([=]() { { std::lock_guard<std::mutex> lk(e->mu); if (e->done < ticket) e->done = ticket; } e->cv.notify_all(); });
  return hipSuccess;
}
// waits for the record calls issued so far (HIP: the most recent record before the wait)
inline hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(e->mu);
  const uint64_t want = e->recorded;
  while (e->done < want) e->cv.wait(lk);
  return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  uint64_t want;
  { std::lock_guard<std::mutex> lk(e->mu); want = e->recorded; }
  s->push([=]() { std::unique_lock<std::mutex> lk(e->mu); while (e->done < want) e->cv.wait(lk); });
  return hipSuccess;
}
