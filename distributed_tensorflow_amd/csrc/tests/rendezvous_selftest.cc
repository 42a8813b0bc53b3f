// Self-test of the TCP rendezvous (csrc/runtime/rendezvous.cc), meant to run under
// AddressSanitizer / UBSan and ThreadSanitizer: the ranks of a job are threads of this process,
// each with its own handle, so every byte the collectives move and every socket they share is
// seen by the sanitizers.
//   1. 5 ranks: all_gather of rank-dependent payloads (two sizes, one of them 1 MiB),
//      broadcast from rank 0 and from a non-zero root, barriers, a zero-length all_gather
//   2. world 1: no sockets, the collectives are copies
//   3. a job one rank short: create() on the others fails with the time-out code within the
//      bound instead of hanging
//   4. a peer that leaves before a collective: the others get an error code, no hang, no crash
//   5. argument errors
//   6. strangers: two connections that say nothing (one stays open, one closes at once) and one
//      that sends half an introduction arrive before the real peers; the job still forms well
//      inside its deadline
//   7. two processes claim the same rank: the second is told so (refused code) at create()
#include <arpa/inet.h>
#include <netinet/in.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

extern "C" {
void* dtf_rdzv_create(const char* addr, int port, int rank, int world, double timeout_s, int* err);
int dtf_rdzv_all_gather(void* handle, const void* mine, long n, void* out);
int dtf_rdzv_broadcast(void* handle, void* buf, long n, int root);
int dtf_rdzv_barrier(void* handle);
void dtf_rdzv_set_timeout(void* handle, double timeout_s);
void dtf_rdzv_destroy(void* handle);
}

static std::atomic<int> failures{0};
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond);        \
      failures.fetch_add(1);                                                 \
    }                                                                        \
  } while (0)

static int free_port() {
  const int fd = socket(AF_INET, SOCK_STREAM, 0);
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  a.sin_port = 0;
  bind(fd, reinterpret_cast<sockaddr*>(&a), sizeof(a));
  socklen_t len = sizeof(a);
  getsockname(fd, reinterpret_cast<sockaddr*>(&a), &len);
  close(fd);
  return ntohs(a.sin_port);
}

static unsigned char pattern(int rank, long i) { return (unsigned char)(rank * 37 + i * 11 + 5); }

static void full_job(int port, int rank, int world) {
  int err = -1;
  void* h = dtf_rdzv_create("127.0.0.1", port, rank, world, 20.0, &err);
  CHECK(h != nullptr && err == 0);
  if (!h) return;
  for (long n : {7L, 1L << 20}) {
    std::vector<unsigned char> mine((size_t)n), table((size_t)n * world);
    for (long i = 0; i < n; ++i) mine[(size_t)i] = pattern(rank, i);
    CHECK(dtf_rdzv_all_gather(h, mine.data(), n, table.data()) == 0);
    bool ok = true;
    for (int r = 0; r < world && ok; ++r)
      for (long i = 0; i < n; i += (n > 64 ? 4093 : 1))
        if (table[(size_t)r * n + i] != pattern(r, i)) { ok = false; break; }
    CHECK(ok);
  }
  for (int root : {0, world - 2}) {
    std::vector<unsigned char> buf(100001);
    if (rank == root)
      for (size_t i = 0; i < buf.size(); ++i) buf[i] = pattern(root + 100, (long)i);
    CHECK(dtf_rdzv_broadcast(h, buf.data(), (long)buf.size(), root) == 0);
    bool ok = true;
    for (size_t i = 0; i < buf.size(); i += 97)
      if (buf[i] != pattern(root + 100, (long)i)) { ok = false; break; }
    CHECK(ok);
  }
  CHECK(dtf_rdzv_barrier(h) == 0);
  CHECK(dtf_rdzv_all_gather(h, nullptr, 0, nullptr) == 0);
  CHECK(dtf_rdzv_barrier(h) == 0);
  dtf_rdzv_destroy(h);
}

int main() {
  // 1. a complete job
  {
    const int world = 5, port = free_port();
    std::vector<std::thread> ts;
    for (int r = world - 1; r >= 0; --r) ts.emplace_back(full_job, port, r, world);   // rank 0 last
    for (auto& t : ts) t.join();
  }
  // 2. world 1
  {
    int err = -1;
    void* h = dtf_rdzv_create(nullptr, 0, 0, 1, 1.0, &err);
    CHECK(h && err == 0);
    char mine[3] = {1, 2, 3}, out[3] = {0, 0, 0};
    CHECK(dtf_rdzv_all_gather(h, mine, 3, out) == 0 && out[2] == 3);
    CHECK(dtf_rdzv_broadcast(h, mine, 3, 0) == 0);
    CHECK(dtf_rdzv_barrier(h) == 0);
    dtf_rdzv_destroy(h);
  }
  // 3. one rank never comes
  {
    const int world = 3, port = free_port();
    std::atomic<int> errs[2];
    errs[0] = errs[1] = 0;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    for (int r = 0; r < 2; ++r)
      ts.emplace_back([&, r] {
        int err = 0;
        void* h = dtf_rdzv_create("127.0.0.1", port, r, world, 1.0, &err);
        if (h) {                       // rank 1 connects; its first collective must fail
          char one = 1, table[3];
          dtf_rdzv_set_timeout(h, 1.5);
          err = dtf_rdzv_all_gather(h, &one, 1, table);
          dtf_rdzv_destroy(h);
        }
        errs[r] = err;
      });
    for (auto& t : ts) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(errs[0] != 0 && errs[1] != 0);
    CHECK(secs < 10.0);
  }
  // 4. a peer leaves before the collective
  {
    const int world = 3, port = free_port();
    std::atomic<int> rc[3];
    rc[0] = rc[1] = rc[2] = -100;
    std::vector<std::thread> ts;
    for (int r = 0; r < world; ++r)
      ts.emplace_back([&, r] {
        int err = 0;
        void* h = dtf_rdzv_create("127.0.0.1", port, r, world, 10.0, &err);
        CHECK(h != nullptr);
        if (!h) return;
        CHECK(dtf_rdzv_barrier(h) == 0);
        if (r == 2) {                  // leaves
          dtf_rdzv_destroy(h);
          rc[r] = 0;
          return;
        }
        dtf_rdzv_set_timeout(h, 2.0);
        char one = 1, table[3];
        rc[r] = dtf_rdzv_all_gather(h, &one, 1, table);
        dtf_rdzv_destroy(h);
      });
    for (auto& t : ts) t.join();
    CHECK(rc[0] != 0 && rc[1] != 0);
  }
  // 5. arguments
  {
    int err = 0;
    CHECK(dtf_rdzv_create("127.0.0.1", 1234, 3, 3, 1.0, &err) == nullptr && err != 0);
    CHECK(dtf_rdzv_create(nullptr, 0, 0, 2, 1.0, &err) == nullptr && err != 0);
    CHECK(dtf_rdzv_barrier(nullptr) != 0);
    dtf_rdzv_destroy(nullptr);
  }
  // 6. silent strangers do not hold the rendezvous up
  {
    const int world = 3, port = free_port();
    std::atomic<bool> stop{false};
    std::thread strangers([&] {
      sockaddr_in a{};
      a.sin_family = AF_INET;
      a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
      a.sin_port = htons((uint16_t)port);
      int fds[3] = {-1, -1, -1};
      for (int i = 0; i < 3; ++i) {
        for (int tries = 0; tries < 200 && fds[i] < 0; ++tries) {     // rank 0 may not listen yet
          const int fd = socket(AF_INET, SOCK_STREAM, 0);
          if (connect(fd, reinterpret_cast<sockaddr*>(&a), sizeof(a)) == 0) fds[i] = fd;
          else { close(fd); std::this_thread::sleep_for(std::chrono::milliseconds(10)); }
        }
      }
      if (fds[1] >= 0) { close(fds[1]); fds[1] = -1; }               // leaves at once
      const char half[2] = {1, 0};
      if (fds[2] >= 0) (void)!write(fds[2], half, 2);                 // half an introduction
      while (!stop.load()) std::this_thread::sleep_for(std::chrono::milliseconds(20));
      for (int fd : fds) if (fd >= 0) close(fd);
    });
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    ts.emplace_back(full_job, port, 0, world);
    std::this_thread::sleep_for(std::chrono::milliseconds(300));      // strangers get in first
    for (int r = 1; r < world; ++r) ts.emplace_back(full_job, port, r, world);
    for (auto& t : ts) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(secs < 15.0);               // full_job's deadline is 20 s: nobody waited for a stranger
    stop = true;
    strangers.join();
  }
  // 7. a duplicate rank is refused by name, the job forms with the first claimant
  {
    const int world = 3, port = free_port();
    std::atomic<int> dup_err{0};
    std::vector<std::thread> ts;
    ts.emplace_back(full_job, port, 0, world);
    ts.emplace_back(full_job, port, 1, world);
    std::this_thread::sleep_for(std::chrono::milliseconds(500));      // rank 1 is in
    ts.emplace_back([&] {
      int err = 0;
      void* h = dtf_rdzv_create("127.0.0.1", port, 1, world, 5.0, &err);
      CHECK(h == nullptr);
      if (h) dtf_rdzv_destroy(h);
      dup_err = err;
    });
    ts[2].join();
    CHECK(dup_err.load() == -4);
    ts.emplace_back(full_job, port, 2, world);
    for (size_t i = 0; i < ts.size(); ++i) if (i != 2) ts[i].join();
  }
  if (failures.load() == 0) printf("rendezvous selftest ok\n");
  return failures.load() == 0 ? 0 : 1;
}
