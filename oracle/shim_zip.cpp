// TEST INFRASTRUCTURE ONLY -- see oracle/shim/zip.h.  Whole archive in memory, central directory parsed from the back,
// entries inflated with zlib on zip_fopen.
#include "zip.h"

#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct zip {
  std::vector<unsigned char> bytes;
  struct Entry {
    std::string name;
    unsigned method, csize, usize, offset;
  };
  std::vector<Entry> entries;
};
struct zip_file {
  std::vector<unsigned char> data;
  size_t pos;
};

static unsigned rd16(const unsigned char* p) { return p[0] | p[1] << 8; }
static unsigned rd32(const unsigned char* p) { return rd16(p) | rd16(p + 2) << 16; }

zip_t* zip_open(const char* path, int, int* err) {
  if (err) *err = 0;
  FILE* f = fopen(path, "rb");
  if (!f) {
    if (err) *err = 9;  // ZIP_ER_NOENT
    return 0;
  }
  zip_t* a = new zip;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  a->bytes.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n > 22 && fread(a->bytes.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  long eocd = -1;
  if (ok)
    for (long p = n - 22; p >= 0 && p >= n - 22 - 65535; p--)
      if (rd32(&a->bytes[(size_t)p]) == 0x06054b50u) {
        eocd = p;
        break;
      }
  if (eocd < 0) {
    if (err) *err = 19;  // ZIP_ER_NOZIP
    delete a;
    return 0;
  }
  const unsigned count = rd16(&a->bytes[(size_t)eocd + 10]);
  size_t p = rd32(&a->bytes[(size_t)eocd + 16]);
  for (unsigned i = 0; i < count && p + 46 <= a->bytes.size() && rd32(&a->bytes[p]) == 0x02014b50u; i++) {
    zip::Entry e;
    e.method = rd16(&a->bytes[p + 10]);
    e.csize = rd32(&a->bytes[p + 20]);
    e.usize = rd32(&a->bytes[p + 24]);
    const unsigned nl = rd16(&a->bytes[p + 28]), xl = rd16(&a->bytes[p + 30]), cl = rd16(&a->bytes[p + 32]);
    e.offset = rd32(&a->bytes[p + 42]);
    e.name.assign((const char*)&a->bytes[p + 46], nl);
    a->entries.push_back(e);
    p += 46 + nl + xl + cl;
  }
  return a;
}

long long zip_get_num_entries(zip_t* a, unsigned) { return a ? (long long)a->entries.size() : -1; }
const char* zip_get_name(zip_t* a, unsigned long long i, unsigned) { return (a && i < a->entries.size()) ? a->entries[(size_t)i].name.c_str() : 0; }

zip_file_t* zip_fopen(zip_t* a, const char* name, unsigned) {
  if (!a) return 0;
  for (const zip::Entry& e : a->entries) {
    if (e.name != name) continue;
    if ((size_t)e.offset + 30 > a->bytes.size()) return 0;
    const unsigned char* lh = &a->bytes[e.offset];
    const size_t data = (size_t)e.offset + 30 + rd16(lh + 26) + rd16(lh + 28);
    if (data + e.csize > a->bytes.size()) return 0;
    zip_file_t* f = new zip_file;
    f->pos = 0;
    f->data.resize(e.usize);
    if (e.method == 0) {
      memcpy(f->data.data(), &a->bytes[data], e.usize);
    } else {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      inflateInit2(&zs, -15);
      zs.next_in = &a->bytes[data];
      zs.avail_in = e.csize;
      zs.next_out = f->data.data();
      zs.avail_out = e.usize;
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END) {
        delete f;
        return 0;
      }
    }
    return f;
  }
  return 0;
}

long long zip_fread(zip_file_t* f, void* buf, unsigned long long n) {
  if (!f) return -1;
  const size_t k = (size_t)std::min<unsigned long long>(n, f->data.size() - f->pos);
  memcpy(buf, f->data.data() + f->pos, k);
  f->pos += k;
  return (long long)k;
}

int zip_close(zip_t* a) {
  delete a;
  return 0;
}
