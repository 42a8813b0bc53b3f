// Self-test of the C++ IO runtime, built with -fsanitize=address,undefined by
// tests/test_native_sanitizers.py: CRC-32C known answers and incremental use, event-file
// framing, SSTable write -> read round trips (multi-block, empty values, long shared prefixes),
// and the error paths (missing files, truncated tables). Any heap misuse or undefined behaviour
// aborts the process; logic errors return a non-zero exit code.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

extern "C" {
const char* dtf_last_error();
uint32_t dtf_crc32c_extend(uint32_t crc, const char* data, size_t n);
void* dtf_events_open(const char* path);
int dtf_events_write(void* handle, const char* payload, size_t n);
int dtf_events_flush(void* handle);
int dtf_events_close(void* handle);
void* dtf_table_builder_new(const char* path);
int dtf_table_builder_add(void* handle, const char* key, size_t klen, const char* value, size_t vlen);
int dtf_table_builder_finish(void* handle);
void* dtf_table_open(const char* path);
int64_t dtf_table_count(void* handle);
int dtf_table_entry(void* handle, int64_t i, const void** key, size_t* klen, const void** value,
                    size_t* vlen);
void dtf_table_close(void* handle);
}

static int failures = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

static uint32_t masked(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];

  // CRC-32C (Castagnoli) known answers
  CHECK(dtf_crc32c_extend(0, "123456789", 9) == 0xe3069283u);
  CHECK(dtf_crc32c_extend(0, "", 0) == 0u);
  std::vector<char> zeros(32, 0), ones(32, (char)0xff);
  CHECK(dtf_crc32c_extend(0, zeros.data(), 32) == 0x8a9136aau);
  CHECK(dtf_crc32c_extend(0, ones.data(), 32) == 0x62a8ab43u);
  std::string blob;
  for (int i = 0; i < 5000; ++i) blob.push_back((char)(i * 131 + 7));
  for (size_t cut : {0ul, 1ul, 7ul, 8ul, 63ul, 4096ul, 4999ul, 5000ul}) {   // any split, any alignment
    uint32_t part = dtf_crc32c_extend(0, blob.data(), cut);
    CHECK(dtf_crc32c_extend(part, blob.data() + cut, blob.size() - cut) ==
          dtf_crc32c_extend(0, blob.data(), blob.size()));
  }

  // event file: [u64 len][u32 masked crc(len)][payload][u32 masked crc(payload)]
  const std::string ev = dir + "/events.out.tfevents.selftest";
  void* w = dtf_events_open(ev.c_str());
  CHECK(w != nullptr);
  const std::string p1 = "hello", p2(70000, 'x'), p3;
  CHECK(dtf_events_write(w, p1.data(), p1.size()) == 0);
  CHECK(dtf_events_write(w, p2.data(), p2.size()) == 0);
  CHECK(dtf_events_write(w, p3.data(), 0) == 0);
  CHECK(dtf_events_flush(w) == 0);
  CHECK(dtf_events_close(w) == 0);
  FILE* fh = fopen(ev.c_str(), "rb");
  CHECK(fh != nullptr);
  for (const std::string* p : {&p1, &p2, &p3}) {
    uint64_t len = 0;
    uint32_t c1 = 0, c2 = 0;
    CHECK(fread(&len, 8, 1, fh) == 1 && len == p->size());
    CHECK(fread(&c1, 4, 1, fh) == 1 && c1 == masked(dtf_crc32c_extend(0, (const char*)&len, 8)));
    std::string body(len, '\0');
    CHECK(len == 0 || fread(&body[0], 1, len, fh) == len);
    CHECK(body == *p);
    CHECK(fread(&c2, 4, 1, fh) == 1 && c2 == masked(dtf_crc32c_extend(0, body.data(), len)));
  }
  fclose(fh);
  CHECK(dtf_events_open((dir + "/no/such/dir/file").c_str()) == nullptr);
  CHECK(strlen(dtf_last_error()) > 0);

  // SSTable: enough entries for many 4 KiB blocks, shared prefixes, empty key and empty values
  std::vector<std::pair<std::string, std::string>> rows;
  rows.emplace_back("", std::string("\x08\x01", 2));
  for (int i = 0; i < 3000; ++i) {
    char key[64];
    snprintf(key, sizeof key, "layer_%04d/weights/Variable_%d", i / 7, i);
    rows.emplace_back(key, i % 5 == 0 ? std::string() : std::string((size_t)(i % 97), (char)('a' + i % 26)));
  }
  std::sort(rows.begin() + 1, rows.end());
  const std::string tp = dir + "/selftest.index";
  void* b = dtf_table_builder_new(tp.c_str());
  CHECK(b != nullptr);
  for (auto& kv : rows)
    CHECK(dtf_table_builder_add(b, kv.first.data(), kv.first.size(), kv.second.data(), kv.second.size()) == 0);
  CHECK(dtf_table_builder_finish(b) == 0);
  void* t = dtf_table_open(tp.c_str());
  CHECK(t != nullptr);
  CHECK(dtf_table_count(t) == (int64_t)rows.size());
  for (size_t i = 0; i < rows.size(); ++i) {
    const void *k, *v;
    size_t kl, vl;
    CHECK(dtf_table_entry(t, (int64_t)i, &k, &kl, &v, &vl) == 0);
    CHECK(std::string((const char*)k, kl) == rows[i].first);
    CHECK(std::string((const char*)v, vl) == rows[i].second);
  }
  const void *k, *v;
  size_t kl, vl;
  CHECK(dtf_table_entry(t, -1, &k, &kl, &v, &vl) != 0);
  CHECK(dtf_table_entry(t, (int64_t)rows.size(), &k, &kl, &v, &vl) != 0);
  dtf_table_close(t);

  // corrupt / truncated tables must be rejected, not read out of bounds
  fh = fopen(tp.c_str(), "rb");
  std::string whole;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fh)) > 0) whole.append(buf, n);
  fclose(fh);
  for (size_t keep : {0ul, 10ul, 47ul, whole.size() / 2, whole.size() - 1}) {
    const std::string cut = dir + "/cut.index";
    fh = fopen(cut.c_str(), "wb");
    fwrite(whole.data(), 1, keep, fh);
    fclose(fh);
    CHECK(dtf_table_open(cut.c_str()) == nullptr);
  }
  std::string flipped = whole;
  flipped[100] ^= 0x40;                               // payload bit flip -> block checksum
  fh = fopen((dir + "/flip.index").c_str(), "wb");
  fwrite(flipped.data(), 1, flipped.size(), fh);
  fclose(fh);
  CHECK(dtf_table_open((dir + "/flip.index").c_str()) == nullptr);
  CHECK(dtf_table_open((dir + "/missing.index").c_str()) == nullptr);

  if (failures) {
    fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  printf("host_io selftest ok\n");
  return 0;
}
