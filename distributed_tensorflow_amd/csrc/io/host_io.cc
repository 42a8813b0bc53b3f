// Host-side native runtime (libdtf_host.so): CRC-32C, TFRecord event-file writer and the
// SSTable reader/writer behind Saver-V2 checkpoint index files.
//
// These are the pieces of the runtime that sit under summaries (tf.summary.FileWriter at
// tf_distributed.py:97,112) and under the Saver a Supervisor owns (tf_distributed.py:92) in the
// reference's stack. Plain C ABI, loaded with ctypes; no GPU involved.
#include <nmmintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

// ---- CRC-32C (Castagnoli) ------------------------------------------------------------------
uint32_t g_table[8][256];
bool g_table_ready = false;

void init_table() {
  if (g_table_ready) return;
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t)
      g_table[t][i] = (g_table[t - 1][i] >> 8) ^ g_table[0][g_table[t - 1][i] & 0xFF];
  g_table_ready = true;
}

bool have_sse42() { return __builtin_cpu_supports("sse4.2"); }

uint32_t crc32c_sw(uint32_t crc, const uint8_t* p, size_t n) {
  init_table();
  while (n >= 8) {   // slice-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = g_table[7][lo & 0xFF] ^ g_table[6][(lo >> 8) & 0xFF] ^ g_table[5][(lo >> 16) & 0xFF] ^
          g_table[4][lo >> 24] ^ g_table[3][hi & 0xFF] ^ g_table[2][(hi >> 8) & 0xFF] ^
          g_table[1][(hi >> 16) & 0xFF] ^ g_table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = g_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return crc;
}

__attribute__((target("sse4.2"))) uint32_t crc32c_hw(uint32_t crc, const uint8_t* p, size_t n) {
  uint64_t c = crc;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = _mm_crc32_u64(c, v);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = _mm_crc32_u8(c32, *p++);
  return c32;
}

uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n) {
  static const bool hw = have_sse42();
  uint32_t c = crc ^ 0xFFFFFFFFu;
  c = hw ? crc32c_hw(c, (const uint8_t*)data, n) : crc32c_sw(c, (const uint8_t*)data, n);
  return c ^ 0xFFFFFFFFu;
}

uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xA282EAD8u; }

void put_fixed32(std::string* dst, uint32_t v) {
  char b[4];
  memcpy(b, &v, 4);
  dst->append(b, 4);
}
void put_fixed64(std::string* dst, uint64_t v) {
  char b[8];
  memcpy(b, &v, 8);
  dst->append(b, 8);
}
void put_varint64(std::string* dst, uint64_t v) {
  while (v >= 0x80) {
    dst->push_back((char)(v | 0x80));
    v >>= 7;
  }
  dst->push_back((char)v);
}
bool get_varint64(const uint8_t*& p, const uint8_t* end, uint64_t* out) {
  uint64_t r = 0;
  for (int shift = 0; shift <= 63 && p < end; shift += 7) {
    uint64_t b = *p++;
    r |= (b & 0x7F) << shift;
    if (!(b & 0x80)) {
      *out = r;
      return true;
    }
  }
  return false;
}

// ---- TFRecord event file ---------------------------------------------------------------------
struct EventsFile {
  FILE* fh = nullptr;
};

// ---- SSTable (LevelDB table format, no compression) -------------------------------------------
constexpr int kRestartInterval = 16;
constexpr size_t kBlockSize = 4096;
constexpr uint64_t kTableMagic = 0xdb4775248b80fb57ull;

struct BlockBuilder {
  std::string buf;
  std::vector<uint32_t> restarts{0};
  std::string last_key;
  int counter = 0;

  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter < kRestartInterval) {
      const size_t m = std::min(last_key.size(), key.size());
      while (shared < m && last_key[shared] == key[shared]) ++shared;
    } else {
      restarts.push_back((uint32_t)buf.size());
      counter = 0;
    }
    put_varint64(&buf, shared);
    put_varint64(&buf, key.size() - shared);
    put_varint64(&buf, value.size());
    buf.append(key, shared, std::string::npos);
    buf.append(value);
    last_key = key;
    ++counter;
  }
  bool empty() const { return buf.empty(); }
  size_t estimate() const { return buf.size() + restarts.size() * 4 + 4; }
  std::string finish() {
    std::string out = buf;
    for (uint32_t r : restarts) put_fixed32(&out, r);
    put_fixed32(&out, (uint32_t)restarts.size());
    return out;
  }
  void reset() {
    buf.clear();
    restarts.assign(1, 0);
    last_key.clear();
    counter = 0;
  }
};

struct TableBuilder {
  FILE* fh = nullptr;
  uint64_t offset = 0;
  BlockBuilder data, index;
  std::string last_key;
  bool pending_index = false;
  uint64_t pending_off = 0, pending_size = 0;
  bool failed = false;

  // writes block contents + 1-byte type (0 = uncompressed) + masked crc32c(contents + type)
  void write_block(const std::string& contents, uint64_t* off, uint64_t* size) {
    *off = offset;
    *size = contents.size();
    char trailer[5];
    trailer[0] = 0;
    uint32_t crc = crc32c_extend(0, contents.data(), contents.size());
    crc = crc32c_extend(crc, trailer, 1);
    uint32_t masked = mask_crc(crc);
    memcpy(trailer + 1, &masked, 4);
    if (fwrite(contents.data(), 1, contents.size(), fh) != contents.size() ||
        fwrite(trailer, 1, 5, fh) != 5)
      failed = true;
    offset += contents.size() + 5;
  }
  void flush_data() {
    if (data.empty()) return;
    write_block(data.finish(), &pending_off, &pending_size);
    data.reset();
    pending_index = true;
  }
  void emit_pending_index(const std::string& separator) {
    if (!pending_index) return;
    std::string handle;
    put_varint64(&handle, pending_off);
    put_varint64(&handle, pending_size);
    index.add(separator, handle);
    pending_index = false;
  }
};

// shortest key k with a <= k < b (LevelDB BytewiseComparator::FindShortestSeparator)
std::string shortest_separator(const std::string& a, const std::string& b) {
  size_t m = std::min(a.size(), b.size()), d = 0;
  while (d < m && a[d] == b[d]) ++d;
  if (d < m) {
    uint8_t byte = (uint8_t)a[d];
    if (byte < 0xFF && byte + 1 < (uint8_t)b[d]) {
      std::string out = a.substr(0, d + 1);
      out[d] = (char)(byte + 1);
      return out;
    }
  }
  return a;
}

// shortest key >= a (FindShortSuccessor)
std::string short_successor(const std::string& a) {
  for (size_t i = 0; i < a.size(); ++i)
    if ((uint8_t)a[i] != 0xFF) {
      std::string out = a.substr(0, i + 1);
      out[i] = (char)((uint8_t)a[i] + 1);
      return out;
    }
  return a;
}

struct Table {
  std::vector<std::pair<std::string, std::string>> entries;
};

bool read_block(const std::string& file, uint64_t off, uint64_t size, std::string* out) {
  if (off + size + 5 > file.size()) return false;
  const char* p = file.data() + off;
  uint32_t stored;
  memcpy(&stored, p + size + 1, 4);
  uint32_t crc = crc32c_extend(0, p, size + 1);
  if (mask_crc(crc) != stored) return false;
  if (p[size] != 0) return false;   // compressed blocks are not produced by this writer
  out->assign(p, size);
  return true;
}

bool parse_block(const std::string& block, std::vector<std::pair<std::string, std::string>>* out) {
  if (block.size() < 4) return false;
  uint32_t num_restarts;
  memcpy(&num_restarts, block.data() + block.size() - 4, 4);
  if ((uint64_t)num_restarts * 4 + 4 > block.size()) return false;
  const uint8_t* p = (const uint8_t*)block.data();
  const uint8_t* end = p + block.size() - 4 - (size_t)num_restarts * 4;
  std::string key;
  while (p < end) {
    uint64_t shared, non_shared, vlen;
    if (!get_varint64(p, end, &shared) || !get_varint64(p, end, &non_shared) ||
        !get_varint64(p, end, &vlen))
      return false;
    if (shared > key.size() || p + non_shared + vlen > end) return false;
    key.resize(shared);
    key.append((const char*)p, non_shared);
    p += non_shared;
    out->emplace_back(key, std::string((const char*)p, vlen));
    p += vlen;
  }
  return true;
}

}  // namespace

DTF_API const char* dtf_last_error() { return g_last_error.c_str(); }

DTF_API uint32_t dtf_crc32c_extend(uint32_t crc, const char* data, size_t n) {
  return crc32c_extend(crc, data, n);
}
DTF_API int dtf_crc32c_hw() { return have_sse42() ? 1 : 0; }

DTF_API void* dtf_events_open(const char* path) {
  FILE* fh = fopen(path, "ab");
  if (!fh) {
    set_error(std::string("cannot open ") + path);
    return nullptr;
  }
  setvbuf(fh, nullptr, _IOFBF, 1 << 16);
  auto* f = new EventsFile;
  f->fh = fh;
  return f;
}

// one TFRecord: u64 len | u32 masked_crc(len) | payload | u32 masked_crc(payload)
DTF_API int dtf_events_write(void* handle, const char* payload, size_t n) {
  auto* f = (EventsFile*)handle;
  if (!f || !f->fh) return -1;
  char header[12];
  uint64_t len = n;
  memcpy(header, &len, 8);
  uint32_t hcrc = mask_crc(crc32c_extend(0, header, 8));
  memcpy(header + 8, &hcrc, 4);
  uint32_t pcrc = mask_crc(crc32c_extend(0, payload, n));
  if (fwrite(header, 1, 12, f->fh) != 12 || fwrite(payload, 1, n, f->fh) != n ||
      fwrite(&pcrc, 1, 4, f->fh) != 4) {
    set_error("short write to event file");
    return -1;
  }
  return 0;
}
DTF_API int dtf_events_flush(void* handle) {
  auto* f = (EventsFile*)handle;
  return (f && f->fh) ? fflush(f->fh) : -1;
}
DTF_API int dtf_events_close(void* handle) {
  auto* f = (EventsFile*)handle;
  if (!f) return -1;
  int rc = f->fh ? fclose(f->fh) : 0;
  delete f;
  return rc;
}

DTF_API void* dtf_table_builder_new(const char* path) {
  FILE* fh = fopen(path, "wb");
  if (!fh) {
    set_error(std::string("cannot create ") + path);
    return nullptr;
  }
  auto* b = new TableBuilder;
  b->fh = fh;
  return b;
}

// keys must arrive in strictly increasing bytewise order
DTF_API int dtf_table_builder_add(void* handle, const char* key, size_t klen, const char* value,
                                  size_t vlen) {
  auto* b = (TableBuilder*)handle;
  if (!b) return -1;
  std::string k(key, klen), v(value, vlen);
  if (!b->last_key.empty() || b->pending_index || !b->data.empty()) {
    if (!(b->last_key < k)) {
      set_error("table keys must be added in increasing order");
      return -1;
    }
  }
  b->emit_pending_index(shortest_separator(b->last_key, k));
  b->data.add(k, v);
  b->last_key = k;
  if (b->data.estimate() >= kBlockSize) b->flush_data();
  return b->failed ? -1 : 0;
}

DTF_API int dtf_table_builder_finish(void* handle) {
  auto* b = (TableBuilder*)handle;
  if (!b) return -1;
  b->flush_data();
  b->emit_pending_index(short_successor(b->last_key));
  uint64_t meta_off, meta_size, idx_off, idx_size;
  BlockBuilder meta;   // empty metaindex block
  b->write_block(meta.finish(), &meta_off, &meta_size);
  b->write_block(b->index.finish(), &idx_off, &idx_size);
  std::string footer;
  put_varint64(&footer, meta_off);
  put_varint64(&footer, meta_size);
  put_varint64(&footer, idx_off);
  put_varint64(&footer, idx_size);
  footer.resize(40, '\0');
  put_fixed64(&footer, kTableMagic);
  bool ok = !b->failed && fwrite(footer.data(), 1, footer.size(), b->fh) == footer.size();
  ok = (fclose(b->fh) == 0) && ok;
  delete b;
  if (!ok) set_error("failed writing table");
  return ok ? 0 : -1;
}

DTF_API void* dtf_table_open(const char* path) {
  FILE* fh = fopen(path, "rb");
  if (!fh) {
    set_error(std::string("cannot open ") + path);
    return nullptr;
  }
  std::string file;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fh)) > 0) file.append(buf, n);
  fclose(fh);
  if (file.size() < 48) {
    set_error("file too short to be a table");
    return nullptr;
  }
  uint64_t magic;
  memcpy(&magic, file.data() + file.size() - 8, 8);
  if (magic != kTableMagic) {
    set_error("bad table magic");
    return nullptr;
  }
  const uint8_t* p = (const uint8_t*)file.data() + file.size() - 48;
  const uint8_t* end = p + 40;
  uint64_t meta_off, meta_size, idx_off, idx_size;
  if (!get_varint64(p, end, &meta_off) || !get_varint64(p, end, &meta_size) ||
      !get_varint64(p, end, &idx_off) || !get_varint64(p, end, &idx_size)) {
    set_error("bad table footer");
    return nullptr;
  }
  std::string index_block;
  if (!read_block(file, idx_off, idx_size, &index_block)) {
    set_error("corrupt index block");
    return nullptr;
  }
  std::vector<std::pair<std::string, std::string>> index_entries;
  if (!parse_block(index_block, &index_entries)) {
    set_error("malformed index block");
    return nullptr;
  }
  auto table = std::make_unique<Table>();
  for (auto& e : index_entries) {
    const uint8_t* hp = (const uint8_t*)e.second.data();
    const uint8_t* hend = hp + e.second.size();
    uint64_t off, size;
    if (!get_varint64(hp, hend, &off) || !get_varint64(hp, hend, &size)) {
      set_error("bad block handle");
      return nullptr;
    }
    std::string block;
    if (!read_block(file, off, size, &block) || !parse_block(block, &table->entries)) {
      set_error("corrupt data block");
      return nullptr;
    }
  }
  return table.release();
}

DTF_API int64_t dtf_table_count(void* handle) {
  auto* t = (Table*)handle;
  return t ? (int64_t)t->entries.size() : -1;
}

DTF_API int dtf_table_entry(void* handle, int64_t i, const void** key, size_t* klen,
                            const void** value, size_t* vlen) {
  auto* t = (Table*)handle;
  if (!t || i < 0 || (size_t)i >= t->entries.size()) return -1;
  *key = t->entries[i].first.data();
  *klen = t->entries[i].first.size();
  *value = t->entries[i].second.data();
  *vlen = t->entries[i].second.size();
  return 0;
}

DTF_API void dtf_table_close(void* handle) { delete (Table*)handle; }
