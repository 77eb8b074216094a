// Read-only access to the entries of images.zip (host side) -- stands in for the libzip calls of the
// reference's reader (zip_open / zip_get_num_entries / zip_get_name / zip_fopen / zip_fread,
// src/BenchmarkDatasetReader.h:107-125,256-258).  The archive is memory-mapped; entries are "stored"
// or "deflate" (zlib), ZIP64 sizes and offsets are understood.  read() is re-entrant: the decode pool
// inflates many entries at once.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mdc_host {

class ZipArchive {
 public:
  ZipArchive();
  ~ZipArchive();
  // false (with *err set) if the file cannot be opened or is not a zip archive
  bool open(const std::string& path, std::string* err);
  int entries() const { return (int)dir_.size(); }
  const std::string& name(int i) const { return dir_[(size_t)i].name; }
  int find(const std::string& name) const;  // index or -1
  // Uncompressed bytes of entry i into `out`; false on a corrupt entry / unsupported method.
  bool read(int i, std::vector<unsigned char>& out, std::string* err) const;

 private:
  ZipArchive(const ZipArchive&);
  ZipArchive& operator=(const ZipArchive&);
  struct Entry {
    std::string name;
    uint64_t csize, usize, local_off;
    int method;
  };
  std::vector<Entry> dir_;
  const unsigned char* map_;
  size_t size_;
  int fd_;
};

}  // namespace mdc_host
