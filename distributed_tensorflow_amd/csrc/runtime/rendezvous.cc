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
// Every socket is non-blocking and every operation on it is a poll() against the handle's
// deadline (connect included: EINPROGRESS, then POLLOUT and SO_ERROR), so a peer that died or
// never came makes the call fail with an error code (the Python side raises CommError), never
// hang. A connection has HELLO_LIMIT_MS of its own to introduce itself: rank 0 keeps accepting
// meanwhile, so a stranger that connects and stays silent (a port scan, a health probe) costs
// the job nothing. Rank 0 answers each introduction with one status word, so a peer with a
// duplicate or impossible rank learns it was refused (RDZV_ERR_REFUSED) instead of failing
// later on a closed socket.
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
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

enum { RDZV_OK = 0, RDZV_ERR_ARG = -1, RDZV_ERR_IO = -2, RDZV_ERR_TIMEOUT = -3,
       RDZV_ERR_REFUSED = -4 };
constexpr int HELLO_LIMIT_MS = 2000;

bool set_nonblocking(int fd) {
  const int fl = fcntl(fd, F_GETFL, 0);
  return fl >= 0 && fcntl(fd, F_SETFL, fl | O_NONBLOCK) == 0;
}

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
    if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
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
    if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
    if (k <= 0) return RDZV_ERR_IO;           // 0: the peer closed the connection
    p += k;
    n -= (size_t)k;
  }
  return RDZV_OK;
}

Clock::time_point deadline_of(const Rdzv* h) {
  return Clock::now() + std::chrono::milliseconds((long)(h->timeout_s * 1000.0));
}

// Non-blocking connect bounded by `deadline`: RDZV_OK and *out = fd, or an error (fd closed).
int connect_bounded(const sockaddr* addr, socklen_t len, Clock::time_point deadline, int* out) {
  const int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return RDZV_ERR_IO;
  if (!set_nonblocking(fd)) { close(fd); return RDZV_ERR_IO; }
  int rc = connect(fd, addr, len);
  if (rc != 0 && errno == EINPROGRESS) {
    rc = wait_fd(fd, POLLOUT, deadline);
    if (rc == RDZV_OK) {
      int soerr = 0;
      socklen_t sl = sizeof(soerr);
      if (getsockopt(fd, SOL_SOCKET, SO_ERROR, &soerr, &sl) != 0 || soerr != 0) rc = RDZV_ERR_IO;
    }
  } else if (rc != 0) {
    rc = RDZV_ERR_IO;
  }
  if (rc != RDZV_OK) { close(fd); return rc; }
  *out = fd;
  return RDZV_OK;
}

struct Pending {             // an accepted connection that has not finished its introduction
  int fd;
  Clock::time_point limit;
  int32_t who;
  size_t got;
};

// Rank 0: accept until every peer has introduced itself. One poll() over the listening socket
// and all connections still owing their 4-byte rank.
int accept_peers(Rdzv* h, Clock::time_point deadline) {
  const int one = 1;
  std::vector<Pending> pending;
  int got = 1, result = RDZV_OK;
  while (got < h->world) {
    const auto now = Clock::now();
    if (now >= deadline) { result = RDZV_ERR_TIMEOUT; break; }
    for (size_t i = 0; i < pending.size();) {          // strangers that stayed silent
      if (now >= pending[i].limit) {
        close(pending[i].fd);
        pending.erase(pending.begin() + (long)i);
      } else {
        ++i;
      }
    }
    std::vector<pollfd> fds;
    fds.push_back(pollfd{h->listen_fd, POLLIN, 0});
    auto wake = deadline;
    for (const Pending& p : pending) {
      fds.push_back(pollfd{p.fd, POLLIN, 0});
      if (p.limit < wake) wake = p.limit;
    }
    long ms = std::chrono::duration_cast<std::chrono::milliseconds>(wake - now).count() + 1;
    if (ms > 1000) ms = 1000;
    const int rc = poll(fds.data(), fds.size(), (int)ms);
    if (rc < 0 && errno == EINTR) continue;
    if (rc < 0) { result = RDZV_ERR_IO; break; }
    if (rc == 0) continue;
    for (size_t i = fds.size(); i-- > 1;) {             // back to front: erase keeps indices valid
      if (!fds[i].revents) continue;
      Pending& p = pending[i - 1];
      const ssize_t k = recv(p.fd, reinterpret_cast<char*>(&p.who) + p.got, sizeof(p.who) - p.got, 0);
      if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
      bool drop = k <= 0;
      if (!drop) {
        p.got += (size_t)k;
        if (p.got < sizeof(p.who)) continue;
        const bool ok = p.who > 0 && p.who < h->world && h->peers[p.who] < 0;
        const int32_t status = ok ? RDZV_OK : RDZV_ERR_REFUSED;
        const auto limit = p.limit < deadline ? p.limit : deadline;
        if (send_all(p.fd, &status, sizeof(status), limit) == RDZV_OK && ok) {
          h->peers[p.who] = p.fd;
          ++got;
          pending.erase(pending.begin() + (long)(i - 1));
          continue;
        }
        drop = true;
      }
      if (drop) {
        close(p.fd);
        pending.erase(pending.begin() + (long)(i - 1));
      }
    }
    if (fds[0].revents & POLLIN) {
      for (;;) {
        const int fd = accept(h->listen_fd, nullptr, nullptr);
        if (fd < 0) break;
        if (!set_nonblocking(fd)) { close(fd); continue; }
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        auto limit = Clock::now() + std::chrono::milliseconds(HELLO_LIMIT_MS);
        if (deadline < limit) limit = deadline;
        pending.push_back(Pending{fd, limit, -1, 0});
      }
    }
  }
  for (const Pending& p : pending) close(p.fd);
  return result;
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
        listen(h->listen_fd, world + 16) != 0 || !set_nonblocking(h->listen_fd)) {
      *err = RDZV_ERR_IO;
    } else {
      *err = accept_peers(h, deadline);
    }
  } else {
    h->peers.assign(1, -1);
    int rc = RDZV_ERR_TIMEOUT;
    while (Clock::now() < deadline) {                   // rank 0 may not be listening yet
      int fd = -1;
      auto attempt = Clock::now() + std::chrono::milliseconds(5000);
      if (deadline < attempt) attempt = deadline;
      if (connect_bounded(res->ai_addr, res->ai_addrlen, attempt, &fd) == RDZV_OK) {
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        const int32_t who = rank;
        int32_t status = RDZV_ERR_IO;
        rc = send_all(fd, &who, sizeof(who), deadline);
        if (rc == RDZV_OK) rc = recv_all(fd, &status, sizeof(status), deadline);
        if (rc == RDZV_OK && status != RDZV_OK) rc = RDZV_ERR_REFUSED;
        if (rc == RDZV_OK) h->peers[0] = fd;
        else close(fd);
        // a connection that was accepted by the backlog of a dying rank 0 and then reset is
        // worth another try; a refusal or the deadline is final
        if (rc == RDZV_ERR_IO && Clock::now() < deadline) {
          rc = RDZV_ERR_TIMEOUT;
          std::this_thread::sleep_for(std::chrono::milliseconds(50));
          continue;
        }
        break;
      }
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
