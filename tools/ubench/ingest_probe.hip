// ingest_probe — what the host can feed the card from page-cached FASTA files (tmpfs):
//   mode 0: read() every file into a pageable buffer (thread-local, reused)
//   mode 1: read() every file straight into one pinned buffer (sequential offsets)
//   mode 2: mode 1 + one end-of-line scan of the bytes (memchr for '\n' per line)
//   mode 3: mode 2 + H2D of the pinned bytes in chunks as the files land (the whole pipeline)
//   mode 4: H2D alone (pinned -> device), by copy size
// usage: ingest_probe DIR THREADS MODE [chunk MiB]
// build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <dirent.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s DIR THREADS MODE [chunkMiB]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const int nt = atoi(argv[2]), mode = atoi(argv[3]);
  const size_t chunk = (argc > 4 ? (size_t)atoi(argv[4]) : 32) << 20;
  CK(hipSetDevice(0));
  if (mode == 4) {
    const size_t cap = size_t(1) << 30;
    void *h, *d;
    CK(hipHostMalloc(&h, cap, hipHostMallocDefault));
    CK(hipMalloc(&d, cap));
    memset(h, 1, cap);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t sz = 1 << 20; sz <= cap; sz <<= 1) {
      // the whole GiB in copies of `sz`
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        for (size_t off = 0; off < cap; off += sz) CK(hipMemcpyAsync((char*)d + off, (char*)h + off, sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        best = std::min(best, now() - t0);
      }
      printf("h2d 1 GiB in copies of %5zu MiB: %6.2f GB/s\n", sz >> 20, cap / best / 1e9);
    }
    return 0;
  }
  std::vector<std::string> paths;
  std::vector<size_t> sizes, offs;
  {
    DIR* dp = opendir(dir.c_str());
    if (!dp) { perror("opendir"); return 1; }
    while (dirent* e = readdir(dp)) if (e->d_name[0] != '.') paths.push_back(dir + "/" + e->d_name);
    closedir(dp);
    std::sort(paths.begin(), paths.end());
  }
  size_t total = 0;
  for (auto& p : paths) {
    struct stat sb;
    stat(p.c_str(), &sb);
    sizes.push_back((size_t)sb.st_size);
    offs.push_back(total);
    total += ((size_t)sb.st_size + 255) & ~size_t(255);
  }
  uint8_t *pinned = nullptr, *dev = nullptr;
  if (mode >= 1) {
    const double t0 = now();
    CK(hipHostMalloc((void**)&pinned, total, hipHostMallocDefault));
    printf("hipHostMalloc %.1f MiB: %.1f ms\n", total / 1048576.0, (now() - t0) * 1e3);
  }
  if (mode >= 3) CK(hipMalloc((void**)&dev, total));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t nf = paths.size();
  for (int rep = 0; rep < 4; ++rep) {
    std::atomic<size_t> next{0};
    std::vector<std::atomic<uint8_t>> done(nf);
    for (auto& x : done) x.store(0);
    std::atomic<uint64_t> lines{0};
    const double t0 = now();
    auto worker = [&]() {
      std::vector<uint8_t> local;
      uint64_t nl = 0;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= nf) break;
        uint8_t* dst;
        if (mode == 0) { if (local.size() < sizes[i]) local.resize(sizes[i]); dst = local.data(); } else dst = pinned + offs[i];
        const int fd = open(paths[i].c_str(), O_RDONLY);
        size_t got = 0;
        while (got < sizes[i]) { const ssize_t r = read(fd, dst + got, sizes[i] - got); if (r <= 0) break; got += (size_t)r; }
        close(fd);
        if (mode >= 2) {
          const uint8_t *p = dst, *e = dst + got;
          while (p < e) { const uint8_t* q = (const uint8_t*)memchr(p, '\n', e - p); ++nl; if (!q) break; p = q + 1; }
        }
        done[i].store(1, std::memory_order_release);
      }
      lines += nl;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(worker);
    double t_first = 0;
    if (mode >= 3) {   // this thread uploads the completed prefix in chunks
      size_t up = 0, f = 0;
      while (f < nf) {
        size_t g = f;
        while (g < nf && done[g].load(std::memory_order_acquire)) ++g;
        const size_t end = g < nf ? offs[g] : total;
        if (g == nf || end - up >= chunk) {
          if (end > up) { if (up == 0) t_first = now() - t0; CK(hipMemcpyAsync(dev + up, pinned + up, end - up, hipMemcpyHostToDevice, st)); up = end; }
          f = g;
        } else {
          f = g;
          std::this_thread::yield();
        }
      }
    }
    for (auto& t : th) t.join();
    const double t_read = now() - t0;
    if (mode >= 3) CK(hipStreamSynchronize(st));
    const double dt = now() - t0;
    printf("mode %d threads %3d chunk %3zu MiB rep %d: %zu files %.1f MB  read done %.2f ms  all done %.2f ms  %.2f GB/s  %.0f files/s  (first h2d at %.2f ms, %llu lines)\n",
           mode, nt, chunk >> 20, rep, nf, total / 1e6, t_read * 1e3, dt * 1e3, total / dt / 1e9, nf / dt, t_first * 1e3, (unsigned long long)lines.load());
  }
  return 0;
}
