// Self-test of the C++ parameter-server task, built with -fsanitize=thread by
// tests/test_native_sanitizers.py. Worker threads talk to a real server over localhost TCP:
// each owns one variable (PULL / APPLY on it) while all of them ASSIGN new variables, list
// names and poll counters concurrently, so the variable table's shared lock, the atomics, the
// done-set and the shutdown path run under the race detector. (Updates of ONE variable by
// several workers are deliberately lock-free in the server -- the reference's asynchronous
// update rule -- and are therefore not part of this test.) Values are small integers, so the
// final state is exact.
//
// Two more phases run against the same server: (a) one client re-ASSIGNs a variable with
// alternating shapes (each time the table entry is replaced by new storage) while other
// clients PULL it and the hosting process reads it -- the readers must keep the storage they
// found alive (a use-after-free otherwise, which both sanitizer builds report); (b) malformed
// frames -- negative, fractional, NaN and overflowing dimensions, a header that is nothing but
// opening brackets, an unterminated number at the end of the header, non-array "sgd" -- must
// each come back as an error reply (or a closed connection), never a crash or an over-read.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

extern "C" {
const char* dtf_ps_last_error();
void* dtf_ps_start(const char* host, int port, int expected_workers);
int dtf_ps_port(void* h);
int dtf_ps_stopped(void* h);
long dtf_ps_requests(void* h);
long dtf_ps_updates(void* h);
int dtf_ps_wait(void* h, long timeout_ms);
void dtf_ps_stop(void* h);
long dtf_ps_read(void* h, const char* name, void* out, long cap, char* meta, long meta_cap);
long dtf_ps_names(void* h, char* out, long cap);
}

static int failures = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      __atomic_fetch_add(&failures, 1, __ATOMIC_RELAXED);                \
    }                                                                    \
  } while (0)

enum { PULL = 1, ASSIGN = 2, APPLY = 3, UNINIT = 4, SHUTDOWN = 5, NAMES = 6, PING = 7, DONE = 8,
       OK = 100, ERR = 101 };

#pragma pack(push, 1)
struct FrameHeader { char magic[4]; uint8_t op; uint32_t hlen; uint64_t plen; };
#pragma pack(pop)

static bool send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    p += k; n -= (size_t)k;
  }
  return true;
}
static bool recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t k = recv(fd, p, n, 0);
    if (k <= 0) return false;
    p += k; n -= (size_t)k;
  }
  return true;
}

struct Client {
  int fd = -1;
  bool connect_to(int port) {
    fd = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    inet_pton(AF_INET, "127.0.0.1", &addr.sin_addr);
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    return fd >= 0 && connect(fd, (sockaddr*)&addr, sizeof addr) == 0;
  }
  // returns the reply op; header text in `rh`, payload in `rp`
  int call(int op, const std::string& header, const void* payload, size_t plen, std::string* rh,
           std::vector<char>* rp) {
    FrameHeader fh;
    memcpy(fh.magic, "DTF1", 4);
    fh.op = (uint8_t)op;
    fh.hlen = (uint32_t)header.size();
    fh.plen = plen;
    if (!send_all(fd, &fh, sizeof fh) || !send_all(fd, header.data(), header.size()) ||
        (plen && !send_all(fd, payload, plen)))
      return -1;
    FrameHeader r;
    if (!recv_all(fd, &r, sizeof r) || memcmp(r.magic, "DTF1", 4) != 0) return -1;
    std::string h(r.hlen, '\0');
    std::vector<char> p(r.plen);
    if (r.hlen && !recv_all(fd, &h[0], r.hlen)) return -1;
    if (r.plen && !recv_all(fd, p.data(), r.plen)) return -1;
    if (rh) *rh = h;
    if (rp) rp->swap(p);
    return r.op;
  }
  ~Client() { if (fd >= 0) close(fd); }
};

static const int N = 64;           // floats per variable
static const int STEPS = 150;

static void worker(int port, int id) {
  Client c;
  CHECK(c.connect_to(port));
  const std::string name = "w" + std::to_string(id);
  const std::string spec = "{\"dtype\": \"float32\", \"shape\": [" + std::to_string(N) + "]}";
  std::vector<float> init(N, 1.0f), grad(N, 1.0f);
  std::string rh;
  std::vector<char> rp;
  CHECK(c.call(ASSIGN, "{\"names\": [\"" + name + "\"], \"specs\": [" + spec + "]}", init.data(),
               N * 4, &rh, &rp) == OK);
  for (int s = 0; s < STEPS; ++s) {
    CHECK(c.call(PULL, "{\"names\": [\"" + name + "\"]}", nullptr, 0, &rh, &rp) == OK);
    CHECK(rp.size() == (size_t)N * 4);
    if (rp.size() == (size_t)N * 4) {
      float first;
      memcpy(&first, rp.data(), 4);
      CHECK(first == 1.0f - 0.5f * (float)s);          // nobody else touches this variable
    }
    // w -= 0.5 * grad, and the updated value comes back
    CHECK(c.call(APPLY, "{\"specs\": [" + spec + "], \"sgd\": [[\"" + name +
                            "\", 0.5]], \"return\": [\"" + name + "\"]}",
                 grad.data(), N * 4, &rh, &rp) == OK);
    if (s % 10 == 0) {
      // grow the table while others read it
      const std::string extra = "x" + std::to_string(id) + "_" + std::to_string(s);
      CHECK(c.call(ASSIGN, "{\"names\": [\"" + extra + "\"], \"specs\": [" + spec + "]}",
                   init.data(), N * 4, &rh, &rp) == OK);
      CHECK(c.call(NAMES, "{}", nullptr, 0, &rh, &rp) == OK);
      CHECK(rh.find(name) != std::string::npos);
      CHECK(c.call(UNINIT, "{\"names\": [\"" + extra + "\", \"nope\"]}", nullptr, 0, &rh, &rp) ==
            OK);
      CHECK(rh.find("nope") != std::string::npos && rh.find(extra) == std::string::npos);
    }
    if (s % 25 == 0) CHECK(c.call(PING, "{}", nullptr, 0, &rh, &rp) == OK);
  }
  // errors keep the connection usable
  CHECK(c.call(PULL, "{\"names\": [\"missing\"]}", nullptr, 0, &rh, &rp) == ERR);
  CHECK(c.call(PING, "{}", nullptr, 0, &rh, &rp) == OK);
  CHECK(c.call(DONE, "{\"worker\": " + std::to_string(id) + "}", nullptr, 0, &rh, &rp) == OK);
}

static void reshaper(int port, int rounds) {
  Client c;
  CHECK(c.connect_to(port));
  std::string rh;
  std::vector<char> rp;
  std::vector<float> big(4096, 2.0f);
  for (int i = 0; i < rounds; ++i) {
    const int n = (i & 1) ? 4096 : 1024;
    const std::string spec = (i % 3 == 0)
        ? "{\"dtype\": \"float32\", \"shape\": [" + std::to_string(n) + "]}"
        : "{\"dtype\": \"float32\", \"shape\": [" + std::to_string(n / 4) + ", 4]}";
    CHECK(c.call(ASSIGN, "{\"names\": [\"shared\"], \"specs\": [" + spec + "]}", big.data(),
                 (size_t)n * 4, &rh, &rp) == OK);
  }
}

static void puller(int port, int rounds) {
  Client c;
  CHECK(c.connect_to(port));
  std::string rh;
  std::vector<char> rp;
  for (int i = 0; i < rounds; ++i) {
    const int op = c.call(PULL, "{\"names\": [\"shared\"]}", nullptr, 0, &rh, &rp);
    CHECK(op == OK);
    CHECK(rp.size() == 4096u * 4 || rp.size() == 1024u * 4);
    if (!rp.empty()) {
      float first;
      memcpy(&first, rp.data(), 4);
      CHECK(first == 2.0f);
    }
  }
}

static void malformed(int port) {
  const char* bad_specs[] = {
      "{\"dtype\": \"float32\", \"shape\": [-1]}",
      "{\"dtype\": \"float32\", \"shape\": [2.5]}",
      "{\"dtype\": \"float32\", \"shape\": [1e300]}",
      "{\"dtype\": \"float32\", \"shape\": [4294967296, 4294967296]}",
      "{\"dtype\": \"float32\", \"shape\": [4611686018427387904]}",
      "{\"dtype\": \"float32\", \"shape\": [\"8\"]}",
      "{\"dtype\": \"float128\", \"shape\": [8]}",
  };
  float payload[8] = {0};
  for (const char* spec : bad_specs) {
    Client c;
    CHECK(c.connect_to(port));
    std::string rh;
    std::vector<char> rp;
    const std::string h = std::string("{\"names\": [\"m\"], \"specs\": [") + spec + "]}";
    CHECK(c.call(ASSIGN, h, payload, sizeof payload, &rh, &rp) == ERR);
    CHECK(c.call(PING, "{}", nullptr, 0, &rh, &rp) == OK);       // the connection survives
  }
  {
    Client c;                                  // 1 MiB of '[': bounded recursion, error reply
    CHECK(c.connect_to(port));
    std::string rh;
    std::vector<char> rp;
    CHECK(c.call(PING, std::string(1 << 20, '['), nullptr, 0, &rh, &rp) == ERR);
    CHECK(c.call(PING, "{\"x\": 12345", nullptr, 0, &rh, &rp) == ERR);    // number at the very end
    CHECK(c.call(PING, "{\"x\": tru", nullptr, 0, &rh, &rp) == ERR);
    const std::string spec = "{\"dtype\": \"float32\", \"shape\": [8]}";
    CHECK(c.call(ASSIGN, "{\"names\": [\"m\"], \"specs\": [" + spec + "]}", payload,
                 sizeof payload, &rh, &rp) == OK);
    CHECK(c.call(APPLY, "{\"specs\": [" + spec + "], \"sgd\": 7}", payload, sizeof payload, &rh,
                 &rp) == ERR);
    CHECK(c.call(APPLY, "{\"specs\": [" + spec + "], \"sgd\": [[7, \"m\"]]}", payload,
                 sizeof payload, &rh, &rp) == ERR);
    CHECK(c.call(APPLY, "{\"specs\": [" + spec + "], \"sgd\": [[\"m\", 0.5]]}", payload,
                 sizeof payload - 4, &rh, &rp) == ERR);             // payload shorter than the spec
    CHECK(c.call(PING, "{}", nullptr, 0, &rh, &rp) == OK);
  }
}

int main() {
  const int workers = 6;
  void* h = dtf_ps_start("127.0.0.1", 0, workers);
  if (!h) {
    fprintf(stderr, "start failed: %s\n", dtf_ps_last_error());
    return 2;
  }
  const int port = dtf_ps_port(h);
  CHECK(port > 0);
  {
    // phase (a): storage of "shared" is replaced 400 times under three readers
    std::vector<float> first(1024, 2.0f);
    Client c;
    CHECK(c.connect_to(port));
    std::string rh;
    std::vector<char> rp;
    CHECK(c.call(ASSIGN, "{\"names\": [\"shared\"], \"specs\": [{\"dtype\": \"float32\", "
                         "\"shape\": [1024]}]}", first.data(), 1024 * 4, &rh, &rp) == OK);
    std::vector<std::thread> rs;
    rs.emplace_back(reshaper, port, 400);
    for (int i = 0; i < 3; ++i) rs.emplace_back(puller, port, 300);
    for (int i = 0; i < 200; ++i) {
      float out[4096];
      char meta[64];
      const long n = dtf_ps_read(h, "shared", out, sizeof out, meta, sizeof meta);
      CHECK(n == 4096 * 4 || n == 1024 * 4);
      CHECK(out[0] == 2.0f);
    }
    for (auto& t : rs) t.join();
  }
  malformed(port);                                  // phase (b)
  std::vector<std::thread> ts;
  for (int i = 0; i < workers; ++i) ts.emplace_back(worker, port, i);
  // the hosting process polls counters and the table while requests are served
  for (int i = 0; i < 50 && !dtf_ps_stopped(h); ++i) {
    char names[8192];
    CHECK(dtf_ps_names(h, names, sizeof names) >= 0);
    CHECK(dtf_ps_requests(h) >= 0 && dtf_ps_updates(h) >= 0);
    usleep(2000);
  }
  for (auto& t : ts) t.join();
  CHECK(dtf_ps_wait(h, 5000) == 1);                 // every expected worker reported DONE
  CHECK(dtf_ps_updates(h) == (long)workers * STEPS);
  for (int i = 0; i < workers; ++i) {
    float out[N];
    char meta[256];
    const std::string name = "w" + std::to_string(i);
    CHECK(dtf_ps_read(h, name.c_str(), out, sizeof out, meta, sizeof meta) == (long)sizeof out);
    CHECK(out[0] == 1.0f - 0.5f * STEPS && out[N - 1] == out[0]);
  }
  dtf_ps_stop(h);
  if (failures) fprintf(stderr, "%d check(s) failed\n", failures);
  else printf("ps_server selftest ok\n");
  return failures ? 1 : 0;
}
