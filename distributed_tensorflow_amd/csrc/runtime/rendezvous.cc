// Control-plane rendezvous of a job's ranks over TCP: how the native RCCL communicator
// (rccl_comm.cc, parallel/rccl_native.py) exchanges its ncclUniqueId and the small host-side
// objects a training job passes around (variable names, split colours, barriers).
//
// Star topology: rank 0 listens on (address, port) and keeps one connection per peer; every other
// rank connects to it (retrying until the deadline -- ranks start in any order). Three
// collectives, all blocking, all to be called by every rank in the same order:
//   all_gather(mine[n]) -> out[world][n]     peers send their n bytes, rank 0 returns the table
//   broadcast(buf[n], root)                  root's bytes reach everyone (through rank 0)
//   barrier                                  an all_gather of one byte
// Every socket operation is bounded by the handle's time-out; a peer that died or never came
// makes the call fail with an error code (the Python side raises CommError), never hang.
#include <arpa/inet.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <vector>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

using Clock = std::chrono::steady_clock;

struct Rdzv {
  int rank = 0, world = 1;
  double timeout_s = 60.0;
  int listen_fd = -1;
  std::vector<int> peers;     // rank 0: fd of every rank (index = rank; [0] unused); others: [0] = rank 0
};

enum { RDZV_OK = 0, RDZV_ERR_ARG = -1, RDZV_ERR_IO = -2, RDZV_ERR_TIMEOUT = -3 };

int wait_fd(int fd, short events, Clock::time_point deadline) {
  for (;;) {
    const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - Clock::now());
    if (left.count() <= 0) return RDZV_ERR_TIMEOUT;
    pollfd p{fd, events, 0};
    const int rc = poll(&p, 1, (int)(left.count() > 1000 ? 1000 : left.count()));
    if (rc < 0 && errno == EINTR) continue;
    if (rc < 0) return RDZV_ERR_IO;
    if (rc == 0) continue;
    if (p.revents & (POLLERR | POLLNVAL)) return RDZV_ERR_IO;
    return RDZV_OK;            // readable / writable (or hung up: the read / write reports it)
  }
}

int send_all(int fd, const void* buf, size_t n, Clock::time_point deadline) {
  const char* p = static_cast<const char*>(buf);
  while (n > 0) {
    const int w = wait_fd(fd, POLLOUT, deadline);
    if (w != RDZV_OK) return w;
    const ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue;
    if (k <= 0) return RDZV_ERR_IO;
    p += k;
    n -= (size_t)k;
  }
  return RDZV_OK;
}

int recv_all(int fd, void* buf, size_t n, Clock::time_point deadline) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    const int w = wait_fd(fd, POLLIN, deadline);
    if (w != RDZV_OK) return w;
    const ssize_t k = recv(fd, p, n, 0);
    if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue;
    if (k <= 0) return RDZV_ERR_IO;           // 0: the peer closed the connection
    p += k;
    n -= (size_t)k;
  }
  return RDZV_OK;
}

Clock::time_point deadline_of(const Rdzv* h) {
  return Clock::now() + std::chrono::milliseconds((long)(h->timeout_s * 1000.0));
}

void close_all(Rdzv* h) {
  for (int fd : h->peers)
    if (fd >= 0) close(fd);
  h->peers.clear();
  if (h->listen_fd >= 0) close(h->listen_fd);
  h->listen_fd = -1;
}

}  // namespace

// Returns a handle or null (the error code, one of RDZV_*, in *err). world == 1 opens nothing.
DTF_API void* dtf_rdzv_create(const char* addr, int port, int rank, int world, double timeout_s,
                              int* err) {
  int dummy;
  if (!err) err = &dummy;
  *err = RDZV_OK;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!addr || port <= 0))) {
    *err = RDZV_ERR_ARG;
    return nullptr;
  }
  Rdzv* h = new Rdzv;
  h->rank = rank;
  h->world = world;
  h->timeout_s = timeout_s > 0 ? timeout_s : 60.0;
  if (world == 1) return h;
  const auto deadline = deadline_of(h);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char port_s[16];
  snprintf(port_s, sizeof(port_s), "%d", port);
  if (getaddrinfo(addr, port_s, &hints, &res) != 0 || !res) {
    *err = RDZV_ERR_IO;
    delete h;
    return nullptr;
  }
  const int one = 1;
  if (rank == 0) {
    h->peers.assign(world, -1);
    h->listen_fd = socket(AF_INET, SOCK_STREAM, 0);
    setsockopt(h->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (h->listen_fd < 0 || bind(h->listen_fd, res->ai_addr, res->ai_addrlen) != 0 ||
        listen(h->listen_fd, world) != 0) {
      *err = RDZV_ERR_IO;
    } else {
      for (int got = 1; got < world && *err == RDZV_OK;) {
        const int w = wait_fd(h->listen_fd, POLLIN, deadline);
        if (w != RDZV_OK) { *err = w; break; }
        const int fd = accept(h->listen_fd, nullptr, nullptr);
        if (fd < 0) continue;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        int32_t who = -1;                               // the peer introduces itself
        if (recv_all(fd, &who, sizeof(who), deadline) != RDZV_OK || who <= 0 || who >= world ||
            h->peers[who] >= 0) {
          close(fd);
          continue;
        }
        h->peers[who] = fd;
        ++got;
      }
    }
  } else {
    h->peers.assign(1, -1);
    int rc = RDZV_ERR_TIMEOUT;
    while (Clock::now() < deadline) {                   // rank 0 may not be listening yet
      const int fd = socket(AF_INET, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        const int32_t who = rank;
        rc = send_all(fd, &who, sizeof(who), deadline);
        if (rc == RDZV_OK) h->peers[0] = fd;
        else close(fd);
        break;
      }
      if (fd >= 0) close(fd);
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    *err = rc;
  }
  freeaddrinfo(res);
  if (*err != RDZV_OK) {
    close_all(h);
    delete h;
    return nullptr;
  }
  return h;
}

// out[r * n .. r * n + n) = rank r's `mine`
DTF_API int dtf_rdzv_all_gather(void* handle, const void* mine, long n, void* out) {
  Rdzv* h = static_cast<Rdzv*>(handle);
  if (!h || n < 0 || (n > 0 && (!mine || !out))) return RDZV_ERR_ARG;
  char* table = static_cast<char*>(out);
  if (h->world == 1) {
    if (n) memcpy(table, mine, (size_t)n);
    return RDZV_OK;
  }
  const auto deadline = deadline_of(h);
  if (h->rank == 0) {
    if (n) memcpy(table, mine, (size_t)n);
    for (int r = 1; r < h->world; ++r) {
      const int rc = recv_all(h->peers[r], table + (size_t)r * n, (size_t)n, deadline);
      if (rc != RDZV_OK) return rc;
    }
    for (int r = 1; r < h->world; ++r) {
      const int rc = send_all(h->peers[r], table, (size_t)n * h->world, deadline);
      if (rc != RDZV_OK) return rc;
    }
    return RDZV_OK;
  }
  int rc = send_all(h->peers[0], mine, (size_t)n, deadline);
  if (rc == RDZV_OK) rc = recv_all(h->peers[0], table, (size_t)n * h->world, deadline);
  return rc;
}

DTF_API int dtf_rdzv_broadcast(void* handle, void* buf, long n, int root) {
  Rdzv* h = static_cast<Rdzv*>(handle);
  if (!h || n < 0 || (n > 0 && !buf) || root < 0 || root >= h->world) return RDZV_ERR_ARG;
  if (h->world == 1 || n == 0) return RDZV_OK;
  const auto deadline = deadline_of(h);
  if (h->rank == 0) {
    if (root != 0) {
      const int rc = recv_all(h->peers[root], buf, (size_t)n, deadline);
      if (rc != RDZV_OK) return rc;
    }
    for (int r = 1; r < h->world; ++r) {
      if (r == root) continue;
      const int rc = send_all(h->peers[r], buf, (size_t)n, deadline);
      if (rc != RDZV_OK) return rc;
    }
    return RDZV_OK;
  }
  if (h->rank == root) return send_all(h->peers[0], buf, (size_t)n, deadline);
  return recv_all(h->peers[0], buf, (size_t)n, deadline);
}

DTF_API int dtf_rdzv_barrier(void* handle) {
  Rdzv* h = static_cast<Rdzv*>(handle);
  if (!h) return RDZV_ERR_ARG;
  std::vector<char> table((size_t)h->world);
  const char one = 1;
  return dtf_rdzv_all_gather(handle, &one, 1, table.data());
}

DTF_API void dtf_rdzv_set_timeout(void* handle, double timeout_s) {
  Rdzv* h = static_cast<Rdzv*>(handle);
  if (h && timeout_s > 0) h->timeout_s = timeout_s;
}

DTF_API void dtf_rdzv_destroy(void* handle) {
  Rdzv* h = static_cast<Rdzv*>(handle);
  if (!h) return;
  close_all(h);
  delete h;
}
