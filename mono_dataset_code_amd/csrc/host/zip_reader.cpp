#include "zip_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <cstring>

namespace mdc_host {
namespace {
uint32_t le16(const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
uint32_t le32(const unsigned char* p) { return le16(p) | le16(p + 2) << 16; }
uint64_t le64(const unsigned char* p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }
bool fail(std::string* err, const std::string& msg) {
  if (err) *err = msg;
  return false;
}
}  // namespace

ZipArchive::ZipArchive() : map_(0), size_(0), fd_(-1) {}
ZipArchive::~ZipArchive() {
  if (map_) munmap(const_cast<unsigned char*>(map_), size_);
  if (fd_ >= 0) close(fd_);
}

bool ZipArchive::open(const std::string& path, std::string* err) {
  fd_ = ::open(path.c_str(), O_RDONLY);
  if (fd_ < 0) return fail(err, "cannot open " + path);
  struct stat st;
  if (fstat(fd_, &st) != 0 || st.st_size < 22) return fail(err, path + " is not a zip archive");
  size_ = (size_t)st.st_size;
  void* m = mmap(0, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
  if (m == MAP_FAILED) return fail(err, "mmap failed for " + path);
  map_ = (const unsigned char*)m;
  // end-of-central-directory record: last 22 .. 22+65535 bytes
  size_t eocd = size_;
  for (size_t back = 22; back <= size_ && back <= 22 + 65535; back++)
    if (le32(map_ + size_ - back) == 0x06054b50u) {
      eocd = size_ - back;
      break;
    }
  if (eocd == size_) return fail(err, path + ": no end-of-central-directory record");
  uint64_t n = le16(map_ + eocd + 10), cd_size = le32(map_ + eocd + 12), cd_off = le32(map_ + eocd + 16);
  if (eocd >= 20 && le32(map_ + eocd - 20) == 0x07064b50u) {  // ZIP64 locator -> ZIP64 EOCD
    const uint64_t z = le64(map_ + eocd - 20 + 8);
    if (size_ >= 56 && z <= size_ - 56 && le32(map_ + z) == 0x06064b50u) {
      n = le64(map_ + z + 32);
      cd_size = le64(map_ + z + 40);
      cd_off = le64(map_ + z + 48);
    }
  }
  // untrusted 64-bit fields: every range check is written without an addition that could wrap
  if (cd_off > size_ || cd_size > size_ - cd_off) return fail(err, path + ": central directory out of range");
  size_t p = (size_t)cd_off;
  dir_.clear();
  for (uint64_t i = 0; i < n; i++) {
    if (size_ < 46 || p > size_ - 46 || le32(map_ + p) != 0x02014b50u) return fail(err, path + ": bad central directory entry");
    Entry e;
    e.method = (int)le16(map_ + p + 10);
    e.csize = le32(map_ + p + 20);
    e.usize = le32(map_ + p + 24);
    const size_t nl = le16(map_ + p + 28), xl = le16(map_ + p + 30), cl = le16(map_ + p + 32);
    e.local_off = le32(map_ + p + 42);
    if (nl + xl + cl > size_ - 46 - p) return fail(err, path + ": truncated central directory");  // p + 46 <= size_ holds
    e.name.assign((const char*)map_ + p + 46, nl);
    // ZIP64 extended information: the fields that read 0xffffffff, in this order
    const size_t xend = p + 46 + nl + xl;  // <= size_ (checked above)
    for (size_t x = p + 46 + nl; x + 4 <= xend;) {
      const uint32_t id = le16(map_ + x), sz = le16(map_ + x + 2);
      if (sz > xend - (x + 4)) break;  // a field that claims to run past the extra area: ignore the rest
      if (id == 1) {
        size_t q = x + 4;
        const size_t fend = x + 4 + sz;
        if (e.usize == 0xffffffffu && q + 8 <= fend) { e.usize = le64(map_ + q); q += 8; }
        if (e.csize == 0xffffffffu && q + 8 <= fend) { e.csize = le64(map_ + q); q += 8; }
        if (e.local_off == 0xffffffffu && q + 8 <= fend) { e.local_off = le64(map_ + q); q += 8; }
      }
      x += 4 + sz;
    }
    dir_.push_back(e);
    p += 46 + nl + xl + cl;
  }
  return true;
}

int ZipArchive::find(const std::string& name) const {
  for (size_t i = 0; i < dir_.size(); i++)
    if (dir_[i].name == name) return (int)i;
  return -1;
}

bool ZipArchive::read(int i, std::vector<unsigned char>& out, std::string* err) const {
  if (i < 0 || i >= (int)dir_.size()) return fail(err, "zip: no such entry");
  const Entry& e = dir_[(size_t)i];
  if (size_ < 30 || e.local_off > size_ - 30 || le32(map_ + e.local_off) != 0x04034b50u) return fail(err, "zip: bad local header of " + e.name);
  const size_t data = (size_t)e.local_off + 30 + le16(map_ + e.local_off + 26) + le16(map_ + e.local_off + 28);
  if (data > size_ || e.csize > size_ - data) return fail(err, "zip: entry " + e.name + " runs past the end of the archive");
  // a corrupted size field must not turn into a giant allocation: deflate expands at most 1032 : 1, stored entries 1 : 1,
  // and both zlib counters are 32-bit
  if (e.usize > (uint64_t)e.csize * 1032u + 64u || e.usize > 0xffffffffull || e.csize > 0xffffffffull)
    return fail(err, "zip: entry " + e.name + " declares an implausible size");
  out.resize((size_t)e.usize);
  if (e.method == 0) {
    if (e.csize != e.usize) return fail(err, "zip: stored entry with differing sizes");
    memcpy(out.data(), map_ + data, (size_t)e.usize);
    return true;
  }
  if (e.method != 8) return fail(err, "zip: entry " + e.name + " uses an unsupported compression method");
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return fail(err, "zip: inflateInit2 failed");
  zs.next_in = const_cast<unsigned char*>(map_ + data);
  zs.avail_in = (uInt)e.csize;
  zs.next_out = out.data();
  zs.avail_out = (uInt)out.size();
  const int rc = inflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && zs.total_out == e.usize;
  inflateEnd(&zs);
  return ok ? true : fail(err, "zip: corrupt deflate stream in " + e.name);
}

}  // namespace mdc_host
