// Parameter-server task in C++: the variable host behind Server(job_name="ps").
//
// The reference's ps process is TensorFlow's C++ gRPC worker service: it owns the variables,
// serves the workers' reads and applies their gradient pushes, unsynchronised across workers
// (tf_distributed.py:31,34-36,75-76). This is that task for the framework's TCP transport:
// one thread per worker connection, variables as raw little-endian buffers, and the same framed
// protocol the Python client speaks (parallel/ps_transport.py):
//
//   frame   = "DTF1" | u8 op | u32 header_len | u64 payload_len | header JSON | payload
//   PULL    {"names":[..]}                       -> {"specs":[{dtype,shape}..]} + values
//   ASSIGN  {"names":[..],"specs":[..]} + values -> {}
//   APPLY   {"sgd":[[name,lr]..],"add":[name..],"specs":[..],"return":[..]} + grads/deltas
//           -> {"specs":[..]} + returned values;  w -= lr * g,  v += d, applied on arrival
//   UNINIT  {"names":[..]} -> {"names":[those not yet assigned]}
//   NAMES   {} -> {"names":[..]}      PING {} -> {"updates":n}
//   DONE    {"worker":k} -> {"done":n} (service stops once `expected_workers` reported)
//   SHUTDOWN {} -> {}
//
// Updates from different connections interleave freely on the variable bytes (asynchronous,
// lock-free SGD as in the reference); only the variable TABLE is guarded, by a shared mutex.
#include <arpa/inet.h>
#include <math.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

enum Op : uint8_t { PULL = 1, ASSIGN = 2, APPLY = 3, UNINIT = 4, SHUTDOWN = 5, NAMES = 6, PING = 7,
                    DONE = 8, OK = 100, ERR = 101 };

// ---- a JSON subset: objects, arrays, strings, numbers, true/false/null -------------------------
struct Json {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const char* key) const {
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct JsonParser {
  const char* p;
  const char* end;          // the buffer is NUL-terminated AT `end` (strtod stops there)
  bool ok = true;
  int depth = 0;            // nesting is capped: a header of '[' * 2^26 must not recurse 2^26 deep
  static constexpr int kMaxDepth = 32;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
  Json parse() {
    Json j;
    if (++depth > kMaxDepth) { ok = false; return j; }
    j = parse_value();
    --depth;
    return j;
  }
  Json parse_value() {
    Json j;
    ws();
    if (p >= end) { ok = false; return j; }
    if (*p == '{') {
      ++p; j.kind = Json::OBJ;
      if (eat('}')) return j;
      do {
        Json k = parse();
        if (!ok || k.kind != Json::STR || !eat(':')) { ok = false; return j; }
        j.obj.emplace_back(k.str, parse());
        if (!ok) return j;
      } while (eat(','));
      if (!eat('}')) ok = false;
    } else if (*p == '[') {
      ++p; j.kind = Json::ARR;
      if (eat(']')) return j;
      do { j.arr.push_back(parse()); if (!ok) return j; } while (eat(','));
      if (!eat(']')) ok = false;
    } else if (*p == '"') {
      ++p; j.kind = Json::STR;
      while (p < end && *p != '"') {
        if (*p == '\\' && p + 1 < end) {
          ++p;
          switch (*p) {
            case 'n': j.str.push_back('\n'); break;
            case 't': j.str.push_back('\t'); break;
            case 'u': {                      // \uXXXX: names are ASCII; keep the low byte
              if (p + 4 < end) { j.str.push_back((char)strtol(std::string(p + 1, 4).c_str(), nullptr, 16)); p += 4; }
              break;
            }
            default: j.str.push_back(*p);
          }
          ++p;
        } else {
          j.str.push_back(*p++);
        }
      }
      if (p >= end) ok = false; else ++p;
    } else if (end - p >= 4 && !memcmp(p, "true", 4)) { j.kind = Json::BOOL; j.b = true; p += 4; }
    else if (end - p >= 5 && !memcmp(p, "false", 5)) { j.kind = Json::BOOL; p += 5; }
    else if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; }
    else {
      char* e = nullptr;
      j.kind = Json::NUM;
      j.num = strtod(p, &e);            // safe: *end == '\0'
      if (e == p || e > end) ok = false;
      p = e;
    }
    return j;
  }
};

void json_escape(std::string& out, const std::string& s) {
  out.push_back('"');
  for (char c : s) {
    if (c == '"' || c == '\\') { out.push_back('\\'); out.push_back(c); }
    else if (c == '\n') out += "\\n";
    else out.push_back(c);
  }
  out.push_back('"');
}

// ---- variables -------------------------------------------------------------------------------
struct Var {
  std::string dtype;                // numpy name: float32, float64, int32, int64, bool, float16
  std::vector<int64_t> shape;
  std::vector<char> data;
};

size_t dtype_size(const std::string& d) {
  if (d == "float32" || d == "int32") return 4;
  if (d == "float64" || d == "int64") return 8;
  if (d == "float16") return 2;
  if (d == "bool") return 1;
  return 0;
}

struct Spec { std::string dtype; std::vector<int64_t> shape; size_t nbytes = 0; };

bool parse_specs(const Json* j, std::vector<Spec>& out) {
  if (!j || j->kind != Json::ARR) return false;
  for (auto& s : j->arr) {
    const Json* d = s.get("dtype");
    const Json* sh = s.get("shape");
    if (!d || !sh || d->kind != Json::STR || sh->kind != Json::ARR) return false;
    Spec sp;
    sp.dtype = d->str;
    const size_t es = dtype_size(sp.dtype);
    if (!es || sh->arr.size() > 32) return false;
    size_t count = 1;
    for (auto& dim : sh->arr) {
      // dimensions come off the network: whole numbers in [0, 2^40], products checked
      if (dim.kind != Json::NUM || !(dim.num >= 0.0) || dim.num > 1099511627776.0 ||
          dim.num != floor(dim.num))
        return false;
      sp.shape.push_back((int64_t)dim.num);
      if (__builtin_mul_overflow(count, (size_t)dim.num, &count)) return false;
    }
    if (__builtin_mul_overflow(count, es, &sp.nbytes)) return false;
    out.push_back(std::move(sp));
  }
  return true;
}

// total payload bytes of `specs`, false on overflow
bool specs_total(const std::vector<Spec>& specs, size_t& total) {
  total = 0;
  for (auto& sp : specs)
    if (__builtin_add_overflow(total, sp.nbytes, &total)) return false;
  return true;
}

void specs_json(std::string& out, const std::vector<const Var*>& vars) {
  out += "{\"specs\": [";
  for (size_t i = 0; i < vars.size(); ++i) {
    if (i) out += ", ";
    out += "{\"dtype\": \"" + vars[i]->dtype + "\", \"shape\": [";
    for (size_t k = 0; k < vars[i]->shape.size(); ++k) {
      if (k) out += ", ";
      out += std::to_string(vars[i]->shape[k]);
    }
    out += "]}";
  }
  out += "]}";
}

struct Server {
  int listen_fd = -1;
  int port = 0;
  int expected_workers = 0;
  std::atomic<bool> stop{false};
  std::atomic<long> requests{0}, updates{0}, bytes_in{0}, bytes_out{0};
  std::shared_mutex table_mu;
  std::map<std::string, std::shared_ptr<Var>> vars;
  std::mutex misc_mu;
  std::condition_variable stop_cv;
  std::set<long> done_workers;
  std::vector<int> conns;
  std::vector<std::thread> threads;
  std::thread accept_thread;
  std::string last_error;

  // A reference taken under the table lock: a variable replaced by ASSIGN (new dtype or size)
  // stays alive until the last connection thread still reading or updating it lets go.
  std::shared_ptr<Var> find(const std::string& name) {
    std::shared_lock<std::shared_mutex> lk(table_mu);
    auto it = vars.find(name);
    return it == vars.end() ? nullptr : it->second;
  }
  void request_stop() {
    stop.store(true);
    std::lock_guard<std::mutex> lk(misc_mu);
    stop_cv.notify_all();
  }
};

bool recv_exact(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t k = recv(fd, p, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    p += k; n -= (size_t)k;
  }
  return true;
}

bool send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    p += k; n -= (size_t)k;
  }
  return true;
}

#pragma pack(push, 1)
struct FrameHeader { char magic[4]; uint8_t op; uint32_t hlen; uint64_t plen; };
#pragma pack(pop)

bool send_frame(int fd, uint8_t op, const std::string& header,
                const std::vector<std::pair<const char*, size_t>>& parts) {
  FrameHeader fh;
  memcpy(fh.magic, "DTF1", 4);
  fh.op = op;
  fh.hlen = (uint32_t)header.size();
  fh.plen = 0;
  for (auto& p : parts) fh.plen += p.second;
  std::string head((const char*)&fh, sizeof fh);
  head += header;
  if (!send_all(fd, head.data(), head.size())) return false;
  for (auto& p : parts)
    if (p.second && !send_all(fd, p.first, p.second)) return false;
  return true;
}

bool send_error(int fd, const char* type, const std::string& message) {
  std::string h = "{\"type\": \"";
  h += type;
  h += "\", \"message\": ";
  json_escape(h, message);
  h += "}";
  return send_frame(fd, ERR, h, {});
}

template <typename T>
void axpy(T* w, const T* g, size_t n, double alpha) {
  const T a = (T)alpha;
  for (size_t i = 0; i < n; ++i) w[i] += a * g[i];
}

// apply `var += alpha * delta` for the numeric dtypes
bool add_scaled(const std::shared_ptr<Var>& v, const char* delta, const Spec& sp, double alpha) {
  if (sp.nbytes != v->data.size()) return false;
  if (v->dtype == "float32" && sp.dtype == "float32")
    axpy((float*)v->data.data(), (const float*)delta, sp.nbytes / 4, alpha);
  else if (v->dtype == "float64" && sp.dtype == "float64")
    axpy((double*)v->data.data(), (const double*)delta, sp.nbytes / 8, alpha);
  else if (v->dtype == "int32" && sp.dtype == "int32")
    axpy((int32_t*)v->data.data(), (const int32_t*)delta, sp.nbytes / 4, alpha);
  else if (v->dtype == "int64" && sp.dtype == "int64")
    axpy((int64_t*)v->data.data(), (const int64_t*)delta, sp.nbytes / 8, alpha);
  else
    return false;
  return true;
}

// returns false when the connection should be closed
bool handle(Server* s, int fd, uint8_t op, const Json& hdr, const std::vector<char>& payload) {
  auto names_of = [&](const char* key, std::vector<std::string>& out) {
    const Json* j = hdr.get(key);
    if (!j) return true;
    if (j->kind != Json::ARR) return false;
    for (auto& e : j->arr) {
      if (e.kind != Json::STR) return false;
      out.push_back(e.str);
    }
    return true;
  };
  switch (op) {
    case PULL: {
      std::vector<std::string> names;
      if (!names_of("names", names)) return send_error(fd, "ValueError", "bad PULL header");
      std::vector<std::shared_ptr<Var>> held;
      std::vector<const Var*> vs;
      std::vector<std::pair<const char*, size_t>> parts;
      for (auto& n : names) {
        auto v = s->find(n);
        if (!v) return send_error(fd, "FailedPreconditionError",
                                  "Attempting to use uninitialized value " + n);
        vs.push_back(v.get());
        parts.emplace_back(v->data.data(), v->data.size());
        held.push_back(std::move(v));
      }
      std::string h;
      specs_json(h, vs);
      for (auto& p : parts) s->bytes_out += (long)p.second;
      return send_frame(fd, OK, h, parts);
    }
    case ASSIGN: {
      std::vector<std::string> names;
      std::vector<Spec> specs;
      if (!names_of("names", names) || !parse_specs(hdr.get("specs"), specs) ||
          names.size() != specs.size())
        return send_error(fd, "ValueError", "bad ASSIGN header");
      size_t off = 0;
      if (!specs_total(specs, off) || off != payload.size())
        return send_error(fd, "ValueError", "ASSIGN payload size");
      off = 0;
      for (size_t i = 0; i < names.size(); ++i) {
        auto v = s->find(names[i]);
        if (v && v->dtype == specs[i].dtype && v->data.size() == specs[i].nbytes &&
            v->shape == specs[i].shape) {
          memcpy(v->data.data(), payload.data() + off, specs[i].nbytes);   // keep the storage
        } else {
          auto nv = std::make_shared<Var>();
          nv->dtype = specs[i].dtype;
          nv->shape = specs[i].shape;
          nv->data.assign(payload.data() + off, payload.data() + off + specs[i].nbytes);
          std::unique_lock<std::shared_mutex> lk(s->table_mu);
          s->vars[names[i]] = std::move(nv);
        }
        off += specs[i].nbytes;
      }
      return send_frame(fd, OK, "{}", {});
    }
    case APPLY: {
      std::vector<Spec> specs;
      if (!parse_specs(hdr.get("specs"), specs))
        return send_error(fd, "ValueError", "bad APPLY header");
      size_t total = 0;
      if (!specs_total(specs, total) || total != payload.size())
        return send_error(fd, "ValueError", "APPLY payload size");
      size_t k = 0, off = 0;
      if (const Json* sgd = hdr.get("sgd")) {
        if (sgd->kind != Json::ARR) return send_error(fd, "ValueError", "bad sgd list");
        for (auto& item : sgd->arr) {
          if (item.kind != Json::ARR || item.arr.size() != 2 || k >= specs.size() ||
              item.arr[0].kind != Json::STR || item.arr[1].kind != Json::NUM)
            return send_error(fd, "ValueError", "bad sgd entry");
          auto v = s->find(item.arr[0].str);
          if (!v) return send_error(fd, "FailedPreconditionError",
                                    "Attempting to use uninitialized value " + item.arr[0].str);
          if (!add_scaled(v, payload.data() + off, specs[k], -item.arr[1].num))
            return send_error(fd, "ValueError", "gradient does not match " + item.arr[0].str);
          s->updates++;
          off += specs[k++].nbytes;
        }
      }
      if (const Json* add = hdr.get("add")) {
        if (add->kind != Json::ARR) return send_error(fd, "ValueError", "bad add list");
        for (auto& item : add->arr) {
          if (item.kind != Json::STR || k >= specs.size())
            return send_error(fd, "ValueError", "bad add entry");
          auto v = s->find(item.str);
          if (!v) return send_error(fd, "FailedPreconditionError",
                                    "Attempting to use uninitialized value " + item.str);
          if (!add_scaled(v, payload.data() + off, specs[k], 1.0))
            return send_error(fd, "ValueError", "delta does not match " + item.str);
          off += specs[k++].nbytes;
        }
      }
      std::vector<std::string> want;
      if (!names_of("return", want)) return send_error(fd, "ValueError", "bad return list");
      std::vector<std::shared_ptr<Var>> held;
      std::vector<const Var*> vs;
      std::vector<std::pair<const char*, size_t>> parts;
      for (auto& n : want) {
        auto v = s->find(n);
        if (!v) return send_error(fd, "FailedPreconditionError",
                                  "Attempting to use uninitialized value " + n);
        vs.push_back(v.get());
        parts.emplace_back(v->data.data(), v->data.size());
        held.push_back(std::move(v));
      }
      std::string h;
      specs_json(h, vs);
      return send_frame(fd, OK, h, parts);
    }
    case UNINIT: {
      std::vector<std::string> names;
      if (!names_of("names", names)) return send_error(fd, "ValueError", "bad UNINIT header");
      std::string h = "{\"names\": [";
      bool first = true;
      for (auto& n : names)
        if (!s->find(n)) {
          if (!first) h += ", ";
          json_escape(h, n);
          first = false;
        }
      h += "]}";
      return send_frame(fd, OK, h, {});
    }
    case NAMES: {
      std::string h = "{\"names\": [";
      {
        std::shared_lock<std::shared_mutex> lk(s->table_mu);
        bool first = true;
        for (auto& kv : s->vars) {
          if (!first) h += ", ";
          json_escape(h, kv.first);
          first = false;
        }
      }
      h += "]}";
      return send_frame(fd, OK, h, {});
    }
    case PING:
      return send_frame(fd, OK, "{\"updates\": " + std::to_string(s->updates.load()) + "}", {});
    case DONE: {
      size_t n;
      bool finished = false;
      {
        std::lock_guard<std::mutex> lk(s->misc_mu);
        const Json* w = hdr.get("worker");
        s->done_workers.insert(w && w->kind == Json::NUM ? (long)w->num : -1);
        n = s->done_workers.size();
        finished = s->expected_workers > 0 && (int)n >= s->expected_workers;
      }
      const bool sent = send_frame(fd, OK, "{\"done\": " + std::to_string(n) + "}", {});
      if (finished) s->request_stop();
      return sent;
    }
    case SHUTDOWN: {
      send_frame(fd, OK, "{}", {});
      s->request_stop();
      return false;
    }
    default:
      return send_error(fd, "ValueError", "unknown ps op " + std::to_string((int)op));
  }
}

void serve(Server* s, int fd) {
  std::vector<char> hbuf, payload;
  while (!s->stop.load()) {
    FrameHeader fh;
    if (!recv_exact(fd, &fh, sizeof fh) || memcmp(fh.magic, "DTF1", 4) != 0) break;
    if (fh.hlen > (64u << 20) || fh.plen > (64ull << 30)) break;      // nonsense sizes
    hbuf.resize((size_t)fh.hlen + 1);
    hbuf[fh.hlen] = '\0';               // the number parser (strtod) needs a terminator
    payload.resize(fh.plen);
    if (fh.hlen && !recv_exact(fd, hbuf.data(), fh.hlen)) break;
    if (fh.plen && !recv_exact(fd, payload.data(), fh.plen)) break;
    s->requests++;
    s->bytes_in += (long)fh.plen;
    Json hdr;
    hdr.kind = Json::OBJ;
    if (fh.hlen) {
      JsonParser jp{hbuf.data(), hbuf.data() + fh.hlen};
      hdr = jp.parse();
      if (!jp.ok || hdr.kind != Json::OBJ) {
        if (!send_error(fd, "ValueError", "malformed header")) break;
        continue;
      }
    }
    if (!handle(s, fd, fh.op, hdr, payload)) break;
  }
  shutdown(fd, SHUT_RDWR);
  close(fd);
}

void accept_loop(Server* s) {
  while (!s->stop.load()) {
    pollfd pfd{s->listen_fd, POLLIN, 0};
    const int r = poll(&pfd, 1, 200);
    if (r <= 0) continue;
    const int fd = accept(s->listen_fd, nullptr, nullptr);
    if (fd < 0) continue;
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    std::lock_guard<std::mutex> lk(s->misc_mu);
    s->conns.push_back(fd);
    s->threads.emplace_back(serve, s, fd);
  }
}

thread_local std::string g_error;

}  // namespace

DTF_API const char* dtf_ps_last_error() { return g_error.c_str(); }

// Bind host:port (port 0 = ephemeral) and start serving. Returns a handle or null.
DTF_API void* dtf_ps_start(const char* host, int port, int expected_workers) {
  auto s = std::make_unique<Server>();
  s->expected_workers = expected_workers;
  s->listen_fd = socket(AF_INET, SOCK_STREAM, 0);
  if (s->listen_fd < 0) { g_error = "socket() failed"; return nullptr; }
  int one = 1;
  setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)port);
  // bind the configured interface only: a name is resolved, never widened to every interface
  // ("0.0.0.0" has to be asked for explicitly)
  if (!host || !*host || !strcmp(host, "localhost")) host = "127.0.0.1";
  if (inet_pton(AF_INET, host, &addr.sin_addr) != 1) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host, nullptr, &hints, &res) != 0 || !res) {
      g_error = std::string("cannot resolve host ") + host;
      close(s->listen_fd);
      return nullptr;
    }
    addr.sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr;
    freeaddrinfo(res);
  }
  if (bind(s->listen_fd, (sockaddr*)&addr, sizeof addr) != 0 || listen(s->listen_fd, 64) != 0) {
    g_error = std::string("cannot bind/listen on ") + host + ":" + std::to_string(port) + ": " +
              strerror(errno);
    close(s->listen_fd);
    return nullptr;
  }
  socklen_t len = sizeof addr;
  getsockname(s->listen_fd, (sockaddr*)&addr, &len);
  s->port = ntohs(addr.sin_port);
  Server* raw = s.release();
  raw->accept_thread = std::thread(accept_loop, raw);
  return raw;
}

DTF_API int dtf_ps_port(void* h) { return h ? ((Server*)h)->port : -1; }
DTF_API int dtf_ps_stopped(void* h) { return h && ((Server*)h)->stop.load() ? 1 : 0; }
DTF_API long dtf_ps_requests(void* h) { return h ? ((Server*)h)->requests.load() : -1; }
DTF_API long dtf_ps_updates(void* h) { return h ? ((Server*)h)->updates.load() : -1; }

// Block until the service has been asked to stop (remote SHUTDOWN / all workers DONE / stop()).
// Returns 1 when stopped, 0 on timeout (timeout_ms < 0 waits forever).
DTF_API int dtf_ps_wait(void* h, long timeout_ms) {
  auto* s = (Server*)h;
  if (!s) return 1;
  std::unique_lock<std::mutex> lk(s->misc_mu);
  if (timeout_ms < 0) {
    s->stop_cv.wait(lk, [&] { return s->stop.load(); });
    return 1;
  }
  return s->stop_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms),
                             [&] { return s->stop.load(); }) ? 1 : 0;
}

// Stop serving, close every connection, join every thread and free the server.
DTF_API void dtf_ps_stop(void* h) {
  auto* s = (Server*)h;
  if (!s) return;
  s->request_stop();
  if (s->accept_thread.joinable()) s->accept_thread.join();
  std::vector<std::thread> threads;
  {
    std::lock_guard<std::mutex> lk(s->misc_mu);
    for (int fd : s->conns) shutdown(fd, SHUT_RDWR);
    threads.swap(s->threads);
  }
  for (auto& t : threads)
    if (t.joinable()) t.join();
  close(s->listen_fd);
  delete s;
}

// Local access for the hosting process (checkpointing a ps task, tests): copy a variable out.
// Returns its byte size (copying min(size, cap) bytes), or -1 when it does not exist; dtype and
// shape come back as "float32:784,100".
DTF_API long dtf_ps_read(void* h, const char* name, void* out, long cap, char* meta, long meta_cap) {
  auto* s = (Server*)h;
  if (!s) return -1;
  auto v = s->find(name);
  if (!v) return -1;
  if (out && cap > 0) memcpy(out, v->data.data(), std::min((size_t)cap, v->data.size()));
  if (meta && meta_cap > 0) {
    std::string m = v->dtype + ":";
    for (size_t i = 0; i < v->shape.size(); ++i) m += (i ? "," : "") + std::to_string(v->shape[i]);
    snprintf(meta, (size_t)meta_cap, "%s", m.c_str());
  }
  return (long)v->data.size();
}

// Names joined by '\n' into out (truncated at cap); returns the full length needed.
DTF_API long dtf_ps_names(void* h, char* out, long cap) {
  auto* s = (Server*)h;
  if (!s) return -1;
  std::string all;
  {
    std::shared_lock<std::shared_mutex> lk(s->table_mu);
    for (auto& kv : s->vars) { all += kv.first; all.push_back('\n'); }
  }
  if (out && cap > 0) snprintf(out, (size_t)cap, "%s", all.c_str());
  return (long)all.size();
}
