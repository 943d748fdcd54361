// Host micro-benchmark: how fast can T threads get the bytes of many page-cached files in front of a parser — read() into a
// thread-local buffer against mmap(PROT_READ, MAP_POPULATE) + munmap — each followed by one pass over the bytes (memchr for '\n',
// what the indexer does).  usage: read_vs_mmap DIR n_files file_bytes threads [processes]   (writes the files first if DIR is empty)
// With `processes` P > 1 every (mode, repeat) is run by P forked processes of `threads` threads each, process p taking the files
// i = p (mod P): P address spaces, as P ranks of a multi-GPU job have (the mmap / munmap / page-fault paths take a per-process lock).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s DIR n_files file_bytes threads\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const int n = atoi(argv[2]);
  const size_t fb = (size_t)atoll(argv[3]);
  const int T = atoi(argv[4]);
  const int P = argc > 5 ? atoi(argv[5]) : 1;
  mkdir(dir.c_str(), 0755);
  std::vector<std::string> paths(n);
  std::vector<char> line(fb, 'A');
  for (size_t i = 10000; i < fb; i += 10001) line[i] = '\n';
  for (int i = 0; i < n; ++i) {
    paths[i] = dir + "/f" + std::to_string(i) + ".fa";
    if (access(paths[i].c_str(), R_OK) != 0) {
      int fd = open(paths[i].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0 || write(fd, line.data(), fb) != (ssize_t)fb) { perror("write"); return 1; }
      close(fd);
    }
  }
  for (int mode = 0; mode < 3; ++mode) {   // 0 read(), 1 mmap populate, 2 mmap without populate
    for (int rep = 0; rep < 3; ++rep) {
      std::atomic<int> next{0};
      std::atomic<uint64_t> lines{0};
      const double t0 = now();
      int me = 0;
      std::vector<pid_t> kids;
      if (P > 1) {
        for (me = 0; me < P; ++me) {
          const pid_t k = fork();
          if (k == 0) break;
          kids.push_back(k);
        }
        if (me == P) {    // the parent: wait for the P workers
          for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); }
          const double dt = now() - t0;
          printf("%-22s procs %2d x threads %3d  %8.0f files/s  %7.1f GB/s\n", mode == 0 ? "read()" : mode == 1 ? "mmap MAP_POPULATE" : "mmap, faults on touch", P, T,
                 n / dt, n * (double)fb / dt / 1e9);
          fflush(stdout);
          continue;
        }
      }
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&]() {
          std::vector<char> buf(fb + 64);
          uint64_t nl = 0;
          for (;;) {
            const int i = next.fetch_add(1) * P + me;
            if (i >= n) break;
            const int fd = open(paths[i].c_str(), O_RDONLY);
            if (fd < 0) continue;
            const char* p;
            void* m = nullptr;
            if (mode == 0) {
              size_t got = 0;
              while (got < fb) { const ssize_t r = read(fd, buf.data() + got, fb - got); if (r <= 0) break; got += (size_t)r; }
              p = buf.data();
            } else {
              m = mmap(nullptr, fb, PROT_READ, MAP_PRIVATE | (mode == 1 ? MAP_POPULATE : 0), fd, 0);
              if (m == MAP_FAILED) { close(fd); continue; }
              p = (const char*)m;
            }
            close(fd);
            for (const char* q = p; (q = (const char*)memchr(q, '\n', p + fb - q)) != nullptr; ++q) ++nl;
            if (m) munmap(m, fb);
          }
          lines += nl;
        });
      for (auto& x : th) x.join();
      if (P > 1) _exit(0);
      const double dt = now() - t0;
      printf("%-22s threads %3d  %8.0f files/s  %7.1f GB/s  (lines %llu)\n", mode == 0 ? "read()" : mode == 1 ? "mmap MAP_POPULATE" : "mmap, faults on touch", T,
             n / dt, n * (double)fb / dt / 1e9, (unsigned long long)lines.load());
    }
  }
  return 0;
}
